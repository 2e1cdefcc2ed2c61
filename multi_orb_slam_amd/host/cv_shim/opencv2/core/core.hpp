// Build without OpenCV: the reference headers ask for <opencv2/core/core.hpp>; hand them the stand-in (host/cv_compat.h).
#pragma once
#include "cv_compat.h"
