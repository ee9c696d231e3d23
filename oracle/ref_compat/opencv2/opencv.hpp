// ORACLE — test infrastructure only.  Stand-in for <opencv2/opencv.hpp> so that reference headers which merely
// MENTION cv:: types in prototypes (yolov8/include/postprocess.h, preprocess.h) parse.  The only function with a body is
// cv::invertAffineTransform on 2x3 CV_32F matrices (used by yolov8/src/preprocess.cu:106-111): OpenCV is an un-vendored
// third-party dependency of the reference (absent here), so its published algorithm (opencv/modules/imgproc/src/
// imgwarp.cpp, double-precision cofactors of the 2x2 part, results rounded to float) is restated.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
namespace cv {
struct Rect { int x = 0, y = 0, width = 0, height = 0; Rect() {} Rect(int a, int b, int c, int d) : x(a), y(b), width(c), height(d) {} };
struct Scalar { double v[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {} };
struct Size { int width = 0, height = 0; };
struct Point { int x = 0, y = 0; };
#define CV_32F 5
struct Mat {
    int rows = 0, cols = 0;
    uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int /*type*/, void* d) : rows(r), cols(c), data(static_cast<uint8_t*>(d)) {}
    uint8_t* ptr() const { return data; }
    template <typename T>
    T* ptr(int row = 0) const { return reinterpret_cast<T*>(data) + (size_t)row * cols; }
};
inline void invertAffineTransform(const Mat& m, Mat& im) {
    const float* M = m.ptr<float>();
    float* iM = im.ptr<float>();
    const int step = m.cols, istep = im.cols;
    double D = (double)M[0] * M[step + 1] - (double)M[1] * M[step];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[step + 1] * D, A22 = M[0] * D, A12 = -M[1] * D, A21 = -M[step] * D;
    const double b1 = -A11 * M[2] - A12 * M[step + 2];
    const double b2 = -A21 * M[2] - A22 * M[step + 2];
    iM[0] = (float)A11; iM[1] = (float)A12; iM[2] = (float)b1;
    iM[istep] = (float)A21; iM[istep + 1] = (float)A22; iM[istep + 2] = (float)b2;
}
}  // namespace cv
