// Build without OpenCV: the reference headers ask for <opencv2/features2d/features2d.hpp>; hand them the stand-in (host/cv_compat.h).
#pragma once
#include "cv_compat.h"
