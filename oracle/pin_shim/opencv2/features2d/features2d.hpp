// declarations-only stand-in for the syntax check of oracle/pin_opencv.cpp (see ../../pin_decls.h)
#pragma once
#include "pin_decls.h"
