// TEST-ONLY minimal stand-in for the handful of OpenCV *types* that include/ivfront_orbslam.hpp names in its
// signatures, so the adapter can be syntax-checked in an image without OpenCV.  Not used by the product and
// not used to build anything of the reference.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_Assert(x) do { if (!(x)) std::abort(); } while (0)
namespace cv {
struct Size { int width, height; bool operator==(const Size& o) const { return width == o.width && height == o.height; } };
struct Point2f { float x, y; };
struct KeyPoint {
    Point2f pt; float size, angle, response; int octave;
    KeyPoint() {}
    KeyPoint(float x, float y, float s, float a, float r, int o) : pt{x, y}, size(s), angle(a), response(r), octave(o) {}
};
class Mat {
public:
    int rows = 0, cols = 0; size_t step = 0; uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int) { create(r, c, 0); }
    void create(int r, int c, int) { rows = r; cols = c; step = c; store.assign((size_t)r * c, 0); data = store.data(); }
    bool empty() const { return rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    Size size() const { return {cols, rows}; }
    Mat rowRange(int a, int b) const { Mat m; m.rows = b - a; m.cols = cols; m.step = step; m.data = data + a * step; return m; }
    void copyTo(Mat& o) const { o.create(rows, cols, 0); for (int y = 0; y < rows; y++) std::memcpy(o.data + y * o.step, data + y * step, cols); }
    void release() { rows = cols = 0; data = nullptr; store.clear(); }
    template <class T> T* ptr() { return (T*)data; }
    template <class T> const T* ptr() const { return (const T*)data; }
    Mat getMat() const { return *this; }
private:
    std::vector<uint8_t> store;
};
typedef const Mat& InputArray;
typedef Mat& OutputArray;
}  // namespace cv
