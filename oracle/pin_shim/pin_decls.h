// DECLARATIONS ONLY -- the OpenCV entry points oracle/pin_opencv.cpp calls, so that the harness can be checked with
// `g++ -fsyntax-only` in an image that has no OpenCV (this one).  Nothing here is defined: a link against it fails, which is the point.
// Where OpenCV exists the harness is built with the real headers (`make -C oracle pin`) and this directory is not on the include path.
// Signatures as in OpenCV 2.4 / 3.2 (imgproc.hpp, features2d.hpp, core/core.hpp, types_c.h).
#pragma once
#include "cv_compat.h"
namespace cv {
enum { INTER_NEAREST = 0, INTER_LINEAR = 1 };
enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4, BORDER_DEFAULT = 4, BORDER_ISOLATED = 16 };
struct Scalar { double val[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : val{a, b, c, d} {} };
void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void copyMakeBorder(InputArray src, OutputArray dst, int top, int bottom, int left, int right, int borderType, const Scalar& value = Scalar());
void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_DEFAULT);
void FAST(InputArray image, std::vector<KeyPoint>& keypoints, int threshold, bool nonmaxSuppression = true);
float fastAtan2(float y, float x);
}  // namespace cv
int cvRound(double value);
#ifndef CV_VERSION
#define CV_VERSION "none (declarations only)"
#endif
