// TEST-ONLY functional stand-in for the handful of OpenCV *types* that include/ivfront_orbslam.hpp names (cv::Mat of
// CV_8U / CV_32F, cv::KeyPoint, cv::Point2f, Input/OutputArray), so the adapter can be COMPILED AND RUN in an image without
// OpenCV (tests/adapter/).  Not used by the product and not used to build anything of the reference.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_Assert(x) do { if (!(x)) std::abort(); } while (0)
namespace cv {
struct Size { int width, height; bool operator==(const Size& o) const { return width == o.width && height == o.height; } };
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float a, float b) : x(a), y(b) {} };
struct KeyPoint {
    Point2f pt; float size, angle, response; int octave;
    KeyPoint() : size(0), angle(-1), response(0), octave(0) {}
    KeyPoint(float x, float y, float s, float a, float r, int o) : pt(x, y), size(s), angle(a), response(r), octave(o) {}
};
class Mat {
public:
    int rows = 0, cols = 0; size_t step = 0; uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int t) { create(r, c, t); }
    void create(int r, int c, int t)
    {
        rows = r; cols = c; type_ = t; step = (size_t)c * (t == CV_32F ? 4 : 1);
        store = std::make_shared<std::vector<uint8_t>>((size_t)r * step, 0);
        data = store->data();
    }
    bool empty() const { return rows == 0 || cols == 0 || !data; }
    int type() const { return type_; }
    Size size() const { return Size{cols, rows}; }
    Mat rowRange(int a, int b) const { Mat m(*this); m.rows = b - a; m.data = data + a * step; return m; }
    Mat row(int y) const { return rowRange(y, y + 1); }
    void copyTo(Mat& o) const { o.create(rows, cols, type_); for (int y = 0; y < rows; y++) std::memcpy(o.data + y * o.step, data + y * step, o.step); }
    Mat clone() const { Mat o; copyTo(o); return o; }
    void release() { rows = cols = 0; data = nullptr; store.reset(); }
    template <class T> T* ptr(int y = 0) { return (T*)(data + y * step); }
    template <class T> const T* ptr(int y = 0) const { return (const T*)(data + y * step); }
    template <class T> T& at(int y, int x) { return ((T*)(data + y * step))[x]; }
    template <class T> const T& at(int y, int x) const { return ((const T*)(data + y * step))[x]; }
    Mat getMat() const { return *this; }
private:
    int type_ = CV_8U;
    std::shared_ptr<std::vector<uint8_t>> store;
};
typedef const Mat& InputArray;
typedef Mat& OutputArray;
}  // namespace cv
