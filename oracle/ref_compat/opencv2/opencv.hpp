// ORACLE — test infrastructure only.  Stand-in for <opencv2/opencv.hpp> (OpenCV is an un-vendored third-party dependency of the
// reference, absent here) so that reference translation units that USE cv:: only in their demo / pre- / post-processing code (imread,
// resize, drawing: lenet/lenet.cpp:301-306, retinaface/common.hpp:31-41, rcnn/common.hpp, rcnn/rcnn.cpp main) compile, and their network
// BUILDERS - plain nvinfer1 API consumers in the same files - can be run by oracle/ref_harness/build_*.cpp.  Nothing image-related
// works: the image functions are declarations that abort if a test ever reaches them.  The one function with a body is
// cv::invertAffineTransform on 2x3 CV_32F matrices (used by yolov8/src/preprocess.cu:106-111): its published algorithm (opencv/modules/
// imgproc/src/imgwarp.cpp, double-precision cofactors of the 2x2 part, results rounded to float) is restated.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32F 5
#define CV_32FC1 5
#define CV_32FC3 21
namespace cv {
[[noreturn]] inline void stub_abort(const char* what) {
    fprintf(stderr, "[oracle/ref_compat] cv::%s is a compile-only stand-in (OpenCV is not available here)\n", what);
    abort();
}
struct Size { int width = 0, height = 0; Size() {} Size(int w, int h) : width(w), height(h) {} };
struct Point { int x = 0, y = 0; Point() {} Point(int a, int b) : x(a), y(b) {} };
struct Scalar { double v[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {} };
struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() {}
    Rect(int a, int b, int c, int d) : x(a), y(b), width(c), height(d) {}
    Rect(const Point& a, const Point& b) : x(a.x), y(a.y), width(b.x - a.x), height(b.y - a.y) {}
    Point tl() const { return Point(x, y); }
    Point br() const { return Point(x + width, y + height); }
};
struct Vec3b { uint8_t v[3]; uint8_t& operator[](int i) { return v[i]; } const uint8_t& operator[](int i) const { return v[i]; } };
struct Mat {
    int rows = 0, cols = 0;
    uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int /*type*/) : rows(r), cols(c) {}
    Mat(int r, int c, int /*type*/, void* d) : rows(r), cols(c), data(static_cast<uint8_t*>(d)) {}
    Mat(int r, int c, int /*type*/, const Scalar&) : rows(r), cols(c) {}
    Mat(Size s, int /*type*/) : rows(s.height), cols(s.width) {}
    Mat(Size s, int /*type*/, const Scalar&) : rows(s.height), cols(s.width) {}
    static Mat zeros(int r, int c, int t) { return Mat(r, c, t); }
    static Mat zeros(Size s, int t) { return Mat(s, t); }
    uint8_t* ptr() const { return data; }
    template <typename T>
    T* ptr(int row = 0) const { return reinterpret_cast<T*>(data) + (size_t)row * cols; }
    template <typename T>
    T& at(int) const { stub_abort("Mat::at"); }
    template <typename T>
    T& at(int, int) const { stub_abort("Mat::at"); }
    Size size() const { return Size(cols, rows); }
    bool empty() const { return data == nullptr; }
    int channels() const { return 3; }
    size_t total() const { return (size_t)rows * cols; }
    Mat clone() const { return *this; }
    void copyTo(const Mat&) const { stub_abort("Mat::copyTo"); }
    void convertTo(Mat&, int, double = 1, double = 0) const { stub_abort("Mat::convertTo"); }
    Mat operator()(const Rect&) const { stub_abort("Mat::operator()"); }
    Mat& setTo(const Scalar&) { return *this; }
    size_t elemSize() const { return 4; }
    Mat& operator+=(const Mat&) { stub_abort("Mat::operator+="); }
};
inline Mat operator/(const Mat&, const Scalar&) { stub_abort("Mat / Scalar"); }
enum { IMREAD_GRAYSCALE = 0, IMREAD_COLOR = 1, INTER_LINEAR = 1, INTER_NEAREST = 0, FONT_HERSHEY_PLAIN = 1, THRESH_BINARY = 0, RETR_EXTERNAL = 0,
       CHAIN_APPROX_NONE = 1, COLOR_BGR2RGB = 4 };
template <typename... A> Mat imread(A&&...) { stub_abort("imread"); }
template <typename... A> bool imwrite(A&&...) { stub_abort("imwrite"); }
template <typename... A> void resize(A&&...) { stub_abort("resize"); }
template <typename... A> void rectangle(A&&...) { stub_abort("rectangle"); }
template <typename... A> void putText(A&&...) { stub_abort("putText"); }
template <typename... A> void circle(A&&...) { stub_abort("circle"); }
template <typename... A> double threshold(A&&...) { stub_abort("threshold"); }
template <typename... A> void findContours(A&&...) { stub_abort("findContours"); }
template <typename... A> void drawContours(A&&...) { stub_abort("drawContours"); }
template <typename... A> void cvtColor(A&&...) { stub_abort("cvtColor"); }
namespace dnn {
template <typename... A> Mat blobFromImages(A&&...) { stub_abort("dnn::blobFromImages"); }
}
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
