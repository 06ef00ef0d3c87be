// svx_reference_api.hpp -- header-only C++ mirror of the reference's interface for the census
// stereo-matching path, implemented on libsvx (include/svx.h).  A caller of the reference switches by
// including this header instead of the sparsestereo / highPerfStereoLib headers for this path: the class
// and function names, argument meaning and error behaviour are the reference's; the arithmetic runs on
// the GPU.  Every entry cites the reference declaration it replaces (file:line in the reference tree).
//
// Image arguments are templates: any type with public `rows`, `cols`, `step` (bytes per row) and `data`
// members works -- cv::Mat_<T> does, and so does the minimal svx::Image<T> below, so this header needs
// no OpenCV.  Host images are staged to the device, processed and copied back by each call (the C ABI
// itself works on device pointers; pipelines that keep data on the device should call it directly).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <exception>
#include <string>
#include <type_traits>
#include <unordered_set>
#include <vector>
#include "../../include/svx.h"

namespace svx {

// Minimal owning image with the member names the reference uses on cv::Mat_.
template <typename T>
struct Image {
    typedef T value_type;
    int rows = 0, cols = 0;
    size_t step = 0;   // bytes per row
    unsigned char *data = nullptr;
    std::vector<T> store;
    Image() {}
    Image(int r, int c, T fill = T()) : rows(r), cols(c), step(sizeof(T) * (size_t)c), store((size_t)r * c, fill) {
        data = reinterpret_cast<unsigned char *>(store.data());
    }
    Image(const Image &o) : rows(o.rows), cols(o.cols), step(o.step), store(o.store) { data = reinterpret_cast<unsigned char *>(store.data()); }
    Image &operator=(const Image &o) {
        rows = o.rows; cols = o.cols; step = o.step; store = o.store;
        data = reinterpret_cast<unsigned char *>(store.data());
        return *this;
    }
    T &operator()(int y, int x) { return store[(size_t)y * cols + x]; }
    const T &operator()(int y, int x) const { return store[(size_t)y * cols + x]; }
};

struct Point { int x = 0, y = 0; Point() {} Point(int x_, int y_) : x(x_), y(y_) {} };
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Point3f { float x = 0, y = 0, z = 0; Point3f() {} Point3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {} };
struct KeyPoint { Point2f pt; float size = 0, angle = -1, response = 0; KeyPoint() {} KeyPoint(float x, float y) : pt(x, y) {} };

}  // namespace svx

namespace sparsestereo {

// sparsestereo/exception.h:14-27
class Exception : public std::exception {
public:
    explicit Exception(const char *msg) : message(msg) {}
    explicit Exception(const std::string &msg) : message(msg) {}
    ~Exception() throw() {}
    const char *what() const throw() { return message.c_str(); }
private:
    std::string message;
};

}  // namespace sparsestereo

namespace svx {

// One process-wide context on the current default device (created on first use).
class Runtime {
public:
    static svx_ctx *ctx() {
        static Runtime rt;
        return rt.c;
    }
    static void check(int rc) {
        if (rc != SVX_OK) {
            const char *m = svx_last_error(ctx());
            throw sparsestereo::Exception(m && *m ? std::string(m) : std::string(svx_strerror(rc)));
        }
    }
private:
    svx_ctx *c = nullptr;
    Runtime() {
        const int rc = svx_create(0, &c);
        if (rc != SVX_OK) throw sparsestereo::Exception(std::string("libsvx: ") + svx_strerror(rc));
    }
    ~Runtime() { if (c) svx_destroy(c); }
};

// Device buffer that uploads a host image (row by row when it is not contiguous) and downloads it back.
template <typename T>
class DeviceImage {
public:
    int rows, cols;
    T *d = nullptr;
    DeviceImage(int r, int c) : rows(r), cols(c) { Runtime::check(svx_malloc(Runtime::ctx(), bytes(), (void **)&d)); }
    ~DeviceImage() { if (d) svx_free(Runtime::ctx(), d); }
    DeviceImage(const DeviceImage &) = delete;
    DeviceImage &operator=(const DeviceImage &) = delete;
    size_t bytes() const { return sizeof(T) * (size_t)rows * cols; }
    template <class M>
    void upload(const M &m) {
        if ((size_t)m.step == sizeof(T) * (size_t)cols) { Runtime::check(svx_memcpy_h2d(Runtime::ctx(), d, m.data, bytes())); return; }
        std::vector<T> tmp((size_t)rows * cols);
        for (int y = 0; y < rows; ++y) std::memcpy(&tmp[(size_t)y * cols], m.data + (size_t)y * m.step, sizeof(T) * cols);
        Runtime::check(svx_memcpy_h2d(Runtime::ctx(), d, tmp.data(), bytes()));
    }
    template <class M>
    void download(M &m) const {
        if ((size_t)m.step == sizeof(T) * (size_t)cols) { Runtime::check(svx_memcpy_d2h(Runtime::ctx(), m.data, d, bytes())); return; }
        std::vector<T> tmp((size_t)rows * cols);
        Runtime::check(svx_memcpy_d2h(Runtime::ctx(), tmp.data(), d, bytes()));
        for (int y = 0; y < rows; ++y) std::memcpy(m.data + (size_t)y * m.step, &tmp[(size_t)y * cols], sizeof(T) * cols);
    }
};

template <typename T>
class DeviceArray {
public:
    size_t n;
    T *d = nullptr;
    explicit DeviceArray(size_t n_) : n(n_) { Runtime::check(svx_malloc(Runtime::ctx(), sizeof(T) * (n ? n : 1), (void **)&d)); }
    ~DeviceArray() { if (d) svx_free(Runtime::ctx(), d); }
    DeviceArray(const DeviceArray &) = delete;
    DeviceArray &operator=(const DeviceArray &) = delete;
    void upload(const T *h) { if (n) Runtime::check(svx_memcpy_h2d(Runtime::ctx(), d, h, sizeof(T) * n)); }
    void download(T *h) const { if (n) Runtime::check(svx_memcpy_d2h(Runtime::ctx(), h, d, sizeof(T) * n)); }
};

template <class PointVec>
inline std::vector<int32_t> flatten_points(const PointVec &pts) {
    std::vector<int32_t> xy(2 * pts.size());
    for (size_t i = 0; i < pts.size(); ++i) { xy[2 * i] = pts[i].x; xy[2 * i + 1] = pts[i].y; }
    return xy;
}

}  // namespace svx

namespace sparsestereo {

// sparsestereo/imageconversion.h:24  static void unsignedToSigned(const Mat_<uchar>& src, Mat_<char>* dst)
class ImageConversion {
public:
    template <class MatU8, class MatS8>
    static void unsignedToSigned(const MatU8 &src, MatS8 *dst) {
        svx::DeviceImage<uint8_t> s(src.rows, src.cols);
        svx::DeviceImage<int8_t> d(src.rows, src.cols);
        s.upload(src);
        svx::Runtime::check(svx_u8_to_s8(svx::Runtime::ctx(), s.d, d.d, src.rows, src.cols, 1));
        d.download(*dst);
    }
};

// sparsestereo/census.h:17-18  template<T> static void transform5x5(const Mat_<T>& input, Mat_<unsigned int>* output)
// `char` input selects the 24-bit layout of the SSE specialisation (census.cpp:12-15), any other element
// type the 25-bit layout of the generic template (census-inl.h:13-53), as in the reference.  Pixels the
// reference does not write keep the caller's content.
class Census {
public:
    template <class MatIn, class MatU32>
    static void transform5x5(const MatIn &input, MatU32 *output) {
        typedef typename MatIn::value_type T;   // cv::Mat_<T> and svx::Image<T> both declare value_type
        static_assert(sizeof(T) == 1, "the accelerated census works on 8-bit pixels");
        const bool is_char = std::is_same<T, char>::value || std::is_same<T, signed char>::value;
        svx::DeviceImage<int8_t> in(input.rows, input.cols);
        svx::DeviceImage<uint32_t> out(input.rows, input.cols);
        in.upload(input);
        out.upload(*output);   // untouched pixels keep what the caller had
        if (is_char)           // census.cpp:12-15: the SSE specialisation, 24-bit words
            svx::Runtime::check(svx_census5x5_s8(svx::Runtime::ctx(), in.d, out.d, input.rows, input.cols, 1, SVX_CENSUS_SSE24));
        else                   // generic template on unsigned pixels: unsigned compare == signed compare of (p ^ 0x80)
            svx::Runtime::check(svx_census5x5_u8(svx::Runtime::ctx(), (const uint8_t *)in.d, out.d, input.rows, input.cols, 1, SVX_CENSUS_GENERIC25));
        out.download(*output);
    }
};

// sparsestereo/censuswindow.h:22-64  CensusWindow<SIZE>: set images, set the reference point, match.
// SIZE == 5 only (the reference has no other instantiation that does anything, censuswindow.h:66-67).
template <int SIZE>
class CensusWindow {
public:
    CensusWindow() : rows(0), cols(0) {}
    template <class MatU32> void setReferenceImage(const MatU32 &image) { set(ref, image); }
    template <class MatU32> void setComparisonImage(const MatU32 &image) { set(cmp, image); }
    int getWindowSize() const { return SIZE; }
    void setReferencePoint(const svx::Point &point) { refPoint = point; }
    short match(svx::Point point) const {
        static_assert(SIZE == 5, "only the 5x5 census window exists");
        const int32_t r[2] = {refPoint.x, refPoint.y}, p[2] = {point.x, point.y};
        svx::DeviceArray<int32_t> dr(2), dp(2);
        svx::DeviceArray<int16_t> dc(1);
        dr.upload(r); dp.upload(p);
        svx::Runtime::check(svx_census_window_cost(svx::Runtime::ctx(), ref->d, cmp->d, rows, cols, dr.d, dp.d, 1, dc.d, SVX_WINDOW_5X6));
        int16_t c = 0;
        dc.download(&c);
        return c;
    }
    ~CensusWindow() { delete ref; delete cmp; }
private:
    svx::DeviceImage<uint32_t> *ref = nullptr, *cmp = nullptr;
    int rows, cols;
    svx::Point refPoint;
    template <class MatU32>
    void set(svx::DeviceImage<uint32_t> *&dst, const MatU32 &image) {
        delete dst;
        dst = new svx::DeviceImage<uint32_t>(image.rows, image.cols);
        dst->upload(image);
        rows = image.rows; cols = image.cols;
    }
};

// sparsestereo/sparsematch.h:15-32
struct SparseMatch {
    const svx::KeyPoint *imgLeft, *imgRight;
    svx::Point2f rectLeft, rectRight;
    short cost;
    SparseMatch() : imgLeft(0), imgRight(0), cost(0) {}
    SparseMatch(const svx::KeyPoint *l, const svx::KeyPoint *r, const svx::Point2f &rl, const svx::Point2f &rr, short c = 0)
        : imgLeft(l), imgRight(r), rectLeft(rl), rectRight(rr), cost(c) {}
    float disparity() const { return rectLeft.x - rectRight.x; }

    // sparsestereo/sparsematch.h:35-50  projectMatches(matches, points, rect).  The reference takes Q from the
    // rectification's calibration; rectification is outside this path, so Q is passed itself (any 4x4 matrix of doubles
    // with rows/cols/step/data); the reference's signature stays for rect == NULL and throws as the reference does.
    template <class Point3Vec, class MatD>
    static void projectMatches(const std::vector<SparseMatch> &matches, Point3Vec *points, const MatD &Q) {
        if (Q.rows != 4 || Q.cols != 4) throw Exception("projectMatches: Q must be 4x4");
        double q[16];
        for (int i = 0; i < 4; ++i) std::memcpy(q + 4 * i, Q.data + (size_t)i * Q.step, sizeof(double) * 4);
        std::vector<float> xyd(3 * matches.size()), xyz(3 * matches.size());
        for (size_t i = 0; i < matches.size(); ++i) {
            xyd[3 * i] = matches[i].rectLeft.x; xyd[3 * i + 1] = matches[i].rectLeft.y; xyd[3 * i + 2] = matches[i].disparity();
        }
        svx::DeviceArray<float> din(xyd.size()), dout(xyz.size());
        din.upload(xyd.data());
        svx::Runtime::check(svx_project_matches(svx::Runtime::ctx(), din.d, (int)matches.size(), q, dout.d));
        dout.download(xyz.data());
        typedef typename Point3Vec::value_type P3;
        points->reserve(points->size() + matches.size());
        for (size_t i = 0; i < matches.size(); ++i) points->push_back(P3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
    }
    template <class Point3Vec>
    static void projectMatches(const std::vector<SparseMatch> &, Point3Vec *, const class StereoRectification *rect) {
        if (rect == 0) throw Exception("Projection of 3d points is only possible for calibrated cameras.");
        throw Exception("libsvx: projectMatches takes Q itself (rectification is outside the accelerated path)");
    }
};

class StereoRectification;   // rectification maps are outside the accelerated path: only rect == NULL is supported

// sparsestereo/sparsestereo.h:13-56  SparseStereo<CORRELATION, COST_TYPE>
template <class CORRELATION, typename COST_TYPE>
class SparseStereo {
public:
    SparseStereo(int maxDisparity, float yTolerance = 1, float uniqueness = 0.6, StereoRectification *rect = 0,
                 bool subpixelFeatures = false, bool storeUnmatched = false, int leftRightStep = 1)
        : maxDisparity(maxDisparity), yTolerance(yTolerance), uniqueness(uniqueness), storeUnmatched(storeUnmatched),
          leftRightStep(leftRightStep) {
        (void)subpixelFeatures;
        if (rect != 0) throw Exception("libsvx: SparseStereo supports pre-rectified input only (rect must be NULL)");
    }
    // match(): sparsestereo-inl.h:37-81.  `left` / `right` are the census images; matches are appended.
    template <class MatU32>
    void match(const MatU32 &left, const MatU32 &right, const std::vector<svx::KeyPoint> &leftFeat,
               const std::vector<svx::KeyPoint> &rightFeat, std::vector<SparseMatch> *matches) {
        if (left.rows != right.rows || left.cols != right.cols) throw Exception("Mismatching image sizes");
        if (leftFeat.size() == 0 || rightFeat.size() == 0) return;
        svx::DeviceImage<uint32_t> dl(left.rows, left.cols), dr(left.rows, left.cols);
        dl.upload(left); dr.upload(right);
        std::vector<float> lxy(2 * leftFeat.size()), rxy(2 * rightFeat.size());
        for (size_t i = 0; i < leftFeat.size(); ++i) { lxy[2 * i] = leftFeat[i].pt.x; lxy[2 * i + 1] = leftFeat[i].pt.y; }
        for (size_t i = 0; i < rightFeat.size(); ++i) { rxy[2 * i] = rightFeat[i].pt.x; rxy[2 * i + 1] = rightFeat[i].pt.y; }
        std::vector<int32_t> ol(leftFeat.size()), orr(leftFeat.size());
        std::vector<int16_t> oc(leftFeat.size());
        int n = 0;
        svx::Runtime::check(svx_sparse_match(svx::Runtime::ctx(), dl.d, dr.d, left.rows, left.cols, lxy.data(), (int)leftFeat.size(),
                                             rxy.data(), (int)rightFeat.size(), maxDisparity, yTolerance, uniqueness, leftRightStep,
                                             ol.data(), orr.data(), oc.data(), &n));
        for (int i = 0; i < n; ++i) {   // compose loop, sparsestereo-inl.h:66-80
            const svx::KeyPoint *lk = &leftFeat[ol[i]];
            if (orr[i] >= 0) {
                const svx::KeyPoint *rk = &rightFeat[orr[i]];
                matches->push_back(SparseMatch(lk, rk, lk->pt, rk->pt, oc[i]));
            } else if (storeUnmatched) {
                matches->push_back(SparseMatch(lk, 0, lk->pt, svx::Point2f(lk->pt.x + 1, lk->pt.y), 0));
            }
        }
    }
private:
    int maxDisparity;
    float yTolerance, uniqueness;
    bool storeUnmatched;
    int leftRightStep;
};

}  // namespace sparsestereo

// ---- highPerfStereo/highPerfStereoLib.h:72-151 -------------------------------------------------------
class ConfidentSupport {
public:
    int x, y; float disparity; char cost;
    ConfidentSupport() : x(0), y(0), disparity(0), cost(0) {}
    ConfidentSupport(int x_, int y_, float d, char c) : x(x_), y(y_), disparity(d), cost(c) {}
};
class InvalidMatch {
public:
    int x, y; char cost;
    InvalidMatch() : x(0), y(0), cost(0) {}
    InvalidMatch(int x_, int y_, char c) : x(x_), y(y_), cost(c) {}
};
class PotentialSupports {
    unsigned int rows, cols;
    std::vector<ConfidentSupport> confidentSupports;
    std::vector<InvalidMatch> invalidMatches;
public:
    PotentialSupports(int height, int width, char tLow, char tHigh)
        : rows(height), cols(width), confidentSupports((size_t)height * width, ConfidentSupport(0, 0, 0, tLow)),
          invalidMatches((size_t)height * width, InvalidMatch(0, 0, tHigh)) {}
    unsigned int getOccGridHeight() { return rows; }
    unsigned int getOccGridWidth() { return cols; }
    ConfidentSupport getConfidentSupport(int u, int v) { return confidentSupports[(size_t)v * cols + u]; }
    InvalidMatch getInvalidMatch(int u, int v) { return invalidMatches[(size_t)v * cols + u]; }
    void setConfidentSupport(int u, int v, int x, int y, float d, char cost) { confidentSupports[(size_t)v * cols + u] = ConfidentSupport(x, y, d, cost); }
    void setInvalidMatch(int u, int v, int x, int y, char cost) { invalidMatches[(size_t)v * cols + u] = InvalidMatch(x, y, cost); }
};

// highPerfStereoLib.h:196-200  costEvaluation(censusLeft, censusRight, highGradPts, disparities, matchingCosts)
template <class MatU32, class PointVec, class MatF32, class MatS8>
inline void costEvaluation(const MatU32 &censusLeft, const MatU32 &censusRight, const PointVec &highGradPts,
                           const MatF32 &disparities, MatS8 &matchingCosts) {
    const int rows = censusLeft.rows, cols = censusLeft.cols;
    svx::DeviceImage<uint32_t> dl(rows, cols), dr(rows, cols);
    svx::DeviceImage<float> dd(rows, cols);
    svx::DeviceImage<int8_t> dc(rows, cols);
    dl.upload(censusLeft); dr.upload(censusRight); dd.upload(disparities); dc.upload(matchingCosts);
    std::vector<int32_t> xy = svx::flatten_points(highGradPts);
    svx::DeviceArray<int32_t> dp(xy.size());
    dp.upload(xy.data());
    svx::Runtime::check(svx_cost_eval(svx::Runtime::ctx(), dl.d, dr.d, rows, cols, dp.d, (int)highGradPts.size(), dd.d, dc.d));
    dc.download(matchingCosts);
}

// highPerfStereoLib.h:203-209  disparityRefinement(...)
template <class PointVec, class MatF32, class MatS8>
inline PotentialSupports disparityRefinement(const PointVec &highGradPts, const MatF32 &disparities, const MatS8 &matchingCosts,
                                             const char tLow, const char tHigh, const unsigned int occGridSize,
                                             MatF32 &finalDisparities, MatS8 &finalCosts) {
    const int rows = disparities.rows, cols = disparities.cols;
    const int gh = rows / (int)occGridSize + 1, gw = cols / (int)occGridSize + 1;
    const size_t nc = (size_t)gh * gw;
    svx::DeviceImage<float> dd(rows, cols), dfd(rows, cols);
    svx::DeviceImage<int8_t> dc(rows, cols), dfc(rows, cols);
    dd.upload(disparities); dc.upload(matchingCosts); dfd.upload(finalDisparities); dfc.upload(finalCosts);
    std::vector<int32_t> xy = svx::flatten_points(highGradPts);
    svx::DeviceArray<int32_t> dp(xy.size()), cx(nc), cy(nc), ix(nc), iy(nc);
    svx::DeviceArray<float> cd(nc);
    svx::DeviceArray<int8_t> cc(nc), ic(nc);
    dp.upload(xy.data());
    svx::Runtime::check(svx_refine(svx::Runtime::ctx(), dp.d, (int)highGradPts.size(), dd.d, dc.d, rows, cols, tLow, tHigh, occGridSize,
                                   dfd.d, dfc.d, cx.d, cy.d, cd.d, cc.d, ix.d, iy.d, ic.d));
    dfd.download(finalDisparities); dfc.download(finalCosts);
    std::vector<int32_t> hcx(nc), hcy(nc), hix(nc), hiy(nc);
    std::vector<float> hcd(nc);
    std::vector<int8_t> hcc(nc), hic(nc);
    cx.download(hcx.data()); cy.download(hcy.data()); cd.download(hcd.data()); cc.download(hcc.data());
    ix.download(hix.data()); iy.download(hiy.data()); ic.download(hic.data());
    PotentialSupports ps(gh, gw, tLow, tHigh);
    for (int j = 0; j < gh; ++j)
        for (int i = 0; i < gw; ++i) {
            const size_t c = (size_t)j * gw + i;
            ps.setConfidentSupport(i, j, hcx[c], hcy[c], hcd[c], (char)hcc[c]);
            ps.setInvalidMatch(i, j, hix[c], hiy[c], (char)hic[c]);
        }
    return ps;
}

// highPerfStereoLib.h:211-214  epipolarMatching(censusLeft, censusRight, censusSize, costAggrWindowSize, leftPoint, maxDisparity)
template <class MatU32>
inline ConfidentSupport epipolarMatching(const MatU32 &censusLeft, const MatU32 &censusRight, int censusSize, int costAggrWindowSize,
                                         InvalidMatch leftPoint, int maxDisparity) {
    const int rows = censusLeft.rows, cols = censusLeft.cols;
    svx::DeviceImage<uint32_t> dl(rows, cols), dr(rows, cols);
    dl.upload(censusLeft); dr.upload(censusRight);
    const int32_t p[2] = {leftPoint.x, leftPoint.y};
    svx::DeviceArray<int32_t> dp(2);
    svx::DeviceArray<float> dd(1);
    svx::DeviceArray<int8_t> dc(1);
    dp.upload(p);
    svx::Runtime::check(svx_epipolar_match_batch(svx::Runtime::ctx(), dl.d, dr.d, rows, cols, censusSize, costAggrWindowSize, dp.d, 1,
                                                 maxDisparity, dd.d, dc.d));
    float d = 0; int8_t c = 0;
    dd.download(&d); dc.download(&c);
    return ConfidentSupport(leftPoint.x, leftPoint.y, d, (char)c);
}

// Support-point mesh with the two operations the pipeline needs from Fade_2D (highPerfStereoLib.cpp:434-445, :457):
// insert a point, list the triangles.  Points are kept in insertion order without repeats; triangles() runs libsvx's
// exact Delaunay triangulation (index triples into points()); every insert goes into the mesh that is there (svx_mesh_*).
namespace svx {
class Mesh {
public:
    Mesh() { if (svx_mesh_create(&m) != SVX_OK) throw sparsestereo::Exception("libsvx: cannot create a mesh"); }
    ~Mesh() { svx_mesh_destroy(m); }
    Mesh(const Mesh &) = delete;
    Mesh &operator=(const Mesh &) = delete;
    void insert(const Point &p) {
        const uint64_t key = ((uint64_t)(uint32_t)p.x << 32) | (uint32_t)p.y;
        if (!seen.insert(key).second) return;
        const int32_t xy[2] = {p.x, p.y};
        if (svx_mesh_insert(m, xy, 1) != SVX_OK) throw sparsestereo::Exception("libsvx: mesh point out of range");
        pts.push_back(p);
    }
    const std::vector<Point> &points() const { return pts; }
    std::vector<int32_t> triangles() const {
        std::vector<int32_t> tri(6 * pts.size() + 3);
        int n = 0;
        if (svx_mesh_triangles(m, tri.data(), (int)(tri.size() / 3), &n) != SVX_OK) throw sparsestereo::Exception("libsvx: mesh");
        tri.resize(3 * (size_t)n);
        return tri;
    }
    const svx_mesh *handle() const { return m; }
private:
    svx_mesh *m = nullptr;
    std::vector<Point> pts;
    std::unordered_set<uint64_t> seen;
};
// How supportResampling hands a new support point to the mesh.  The default fits svx::Mesh and any type with
// insert(svx::Point); a caller that keeps Fade2D adds, before including this header's users,
//   namespace svx { inline void mesh_insert(GEOM_FADE2D::Fade_2D &m, int x, int y) { GEOM_FADE2D::Point2 p(x, y); m.insert(p); } }
template <class MeshT>
inline void mesh_insert(MeshT &mesh, int x, int y) { mesh.insert(Point(x, y)); }
}  // namespace svx

// highPerfStereoLib.h:216-222  supportResampling(mesh, ps, censusLeft, censusRight, censusSize, costAggrWindowSize,
//                                                disparities, tLow, tHigh, maxDisp)
// The reference's first parameter is a Fade_2D&; here it is any mesh type (see svx::mesh_insert).  The new points are
// written into `disparities` and inserted in the reference's order (row-major cells; re-matched invalid point first,
// then the confident support, highPerfStereoLib.cpp:151-173).
template <class MeshT, class MatU32, class MatF32>
inline void supportResampling(MeshT &mesh, PotentialSupports &ps, const MatU32 &censusLeft, const MatU32 &censusRight,
                              int censusSize, int costAggrWindowSize, MatF32 &disparities, char tLow, char tHigh, int maxDisp) {
    const int rows = censusLeft.rows, cols = censusLeft.cols;
    const int gw = (int)ps.getOccGridWidth(), gh = (int)ps.getOccGridHeight();
    const size_t nc = (size_t)gw * gh;
    std::vector<int32_t> cx(nc), cy(nc), ix(nc), iy(nc);
    std::vector<float> cd(nc);
    std::vector<int8_t> cc(nc), ic(nc);
    for (int j = 0; j < gh; ++j)
        for (int i = 0; i < gw; ++i) {
            const size_t c = (size_t)j * gw + i;
            const ConfidentSupport s = ps.getConfidentSupport(i, j);
            const InvalidMatch m = ps.getInvalidMatch(i, j);
            cx[c] = s.x; cy[c] = s.y; cd[c] = s.disparity; cc[c] = (int8_t)s.cost;
            ix[c] = m.x; iy[c] = m.y; ic[c] = (int8_t)m.cost;
        }
    svx::DeviceImage<uint32_t> dl(rows, cols), dr(rows, cols);
    svx::DeviceImage<float> dd(rows, cols);
    dl.upload(censusLeft); dr.upload(censusRight); dd.upload(disparities);
    svx::DeviceArray<int32_t> dcx(nc), dcy(nc), dix(nc), diy(nc), oxy(4 * nc);
    svx::DeviceArray<float> dcd(nc), od(2 * nc);
    svx::DeviceArray<int8_t> dcc(nc), dic(nc);
    svx::DeviceArray<uint8_t> ov(2 * nc);
    dcx.upload(cx.data()); dcy.upload(cy.data()); dcd.upload(cd.data()); dcc.upload(cc.data());
    dix.upload(ix.data()); diy.upload(iy.data()); dic.upload(ic.data());
    svx::Runtime::check(svx_support_resampling(svx::Runtime::ctx(), dl.d, dr.d, rows, cols, censusSize, costAggrWindowSize, gw, gh,
                                               dcx.d, dcy.d, dcd.d, dcc.d, dix.d, diy.d, dic.d, tLow, tHigh, maxDisp, dd.d, oxy.d, od.d, ov.d));
    dd.download(disparities);
    std::vector<int32_t> hxy(4 * nc);
    std::vector<uint8_t> hv(2 * nc);
    oxy.download(hxy.data()); ov.download(hv.data());
    for (size_t s = 0; s < 2 * nc; ++s)
        if (hv[s]) svx::mesh_insert(mesh, hxy[2 * s], hxy[2 * s + 1]);
}

// denseCensusStereo/denseCensusStereo.cpp:257-266: the dense winner-takes-all driver loop as one call
// (census images in, float disparity map out; only interior pixels are written, as in the reference).
template <class MatU32, class MatF32>
inline void denseEpipolarMatching(const MatU32 &censusLeft, const MatU32 &censusRight, int maxDisparity, MatF32 &finalDisp) {
    const int rows = censusLeft.rows, cols = censusLeft.cols;
    if (censusRight.rows != rows || censusRight.cols != cols || finalDisp.rows != rows || finalDisp.cols != cols)
        throw sparsestereo::Exception("Mismatching image sizes");
    svx::DeviceImage<uint32_t> dl(rows, cols), dr(rows, cols);
    svx::DeviceImage<float> dd(rows, cols);
    dl.upload(censusLeft); dr.upload(censusRight); dd.upload(finalDisp);
    svx::Runtime::check(svx_dense_wta5x5(svx::Runtime::ctx(), dl.d, dr.d, rows, cols, 1, maxDisparity, dd.d));
    dd.download(finalDisp);
}

// ---- highPerfStereo/highPerfStereoLib.h:35-70 ---------------------------------------------------------
class StereoParameters {
public:
    // Stereo matching parameters
    double uniqueness;
    int maxDisp;
    int leftRightStep;
    int costAggrWindowSize;
    unsigned char gradThreshold;
    char tLow;
    char tHigh;
    int nIters;
    double resizeFactor;
    bool applyBlur;
    bool applyHistEqualization;
    int blurSize;
    // Laplace parameters (gradient calculation)
    int kernelSize;
    int scale;
    int delta;
    int ddepth;
    // Feature detection parameters
    double adaptivity;
    int minThreshold;
    bool traceLines;
    int nbLines;
    int lineSize;
    bool invertRows;
    int nbRows;
    // Misc. parameters
    bool recordFullDisp;
    bool showImages;
    int colorMapSliding;
};

// highPerfStereoLib.h:224-228  highPerfStereo(leftImg, rightImg, parameters, disparityMap, computedPoints)
// Errors as highPerfStereoLib.cpp:184-191.  The computed points are appended, as in the reference.
template <class MatU8, class MatF32, class PointVec>
inline void highPerfStereo(const MatU8 &leftImg, const MatU8 &rightImg, const StereoParameters &parameters,
                           MatF32 &disparityMap, PointVec &highGradPoints) {
    if (leftImg.data == 0 || rightImg.data == 0) throw sparsestereo::Exception("Input images are empty!");
    if (leftImg.cols != rightImg.cols || leftImg.rows != rightImg.rows)
        throw sparsestereo::Exception("Input images do not have the same dimensions!");
    if (leftImg.cols != disparityMap.cols || leftImg.rows != disparityMap.rows)
        throw sparsestereo::Exception("disparity map do not have the same dimensions as the input images!");
    const int rows = leftImg.rows, cols = leftImg.cols;
    svx_stereo_parameters p;
    p.uniqueness = parameters.uniqueness; p.maxDisp = parameters.maxDisp; p.leftRightStep = parameters.leftRightStep;
    p.costAggrWindowSize = parameters.costAggrWindowSize; p.gradThreshold = parameters.gradThreshold;
    p.tLow = (signed char)parameters.tLow; p.tHigh = (signed char)parameters.tHigh; p.nIters = parameters.nIters;
    p.resizeFactor = parameters.resizeFactor; p.applyBlur = parameters.applyBlur; p.applyHistEqualization = parameters.applyHistEqualization;
    p.blurSize = parameters.blurSize; p.kernelSize = parameters.kernelSize; p.scale = parameters.scale; p.delta = parameters.delta;
    p.ddepth = parameters.ddepth; p.adaptivity = parameters.adaptivity; p.minThreshold = parameters.minThreshold;
    p.traceLines = parameters.traceLines; p.nbLines = parameters.nbLines; p.lineSize = parameters.lineSize;
    p.invertRows = parameters.invertRows; p.nbRows = parameters.nbRows; p.recordFullDisp = parameters.recordFullDisp;
    p.showImages = parameters.showImages; p.colorMapSliding = parameters.colorMapSliding;
    svx::DeviceImage<uint8_t> dl(rows, cols), dr(rows, cols);
    svx::DeviceImage<float> dd(rows, cols);
    dl.upload(leftImg); dr.upload(rightImg); dd.upload(disparityMap);
    std::vector<int32_t> xy(4 * (size_t)rows * cols);
    int n = 0;
    svx::Runtime::check(svx_high_perf_stereo(svx::Runtime::ctx(), dl.d, dr.d, rows, cols, &p, dd.d, xy.data(), (int)(xy.size() / 2), &n));
    dd.download(disparityMap);
    typedef typename PointVec::value_type P;
    for (int i = 0; i < n; ++i) highGradPoints.push_back(P(xy[2 * i], xy[2 * i + 1]));
}

// PerformanceEvaluator.h:14-85  PerformanceEvaluator(rawDepth, disparities, consideredDisparities, camMatrix, distortion,
//                                                     perspTransform, rotation, translation, xOffset, yOffset)
// Runs at construction, as the reference's does (PerformanceEvaluator.cpp:16-130), without its text files and console
// output; the results the reference keeps private are readable.  `distortion` is accepted and unused: the reference never
// stores it (its member stays empty, :16-27, :109).  Matrices are any type with rows/cols/step/data holding doubles
// (cv::Mat_<double> or svx::Image<double>), the translation anything indexable with three doubles (cv::Vec3d, double[3]).
class PerformanceEvaluator {
public:
    template <class MatI32, class MatF32, class PointVec, class MatD, class Vec3>
    PerformanceEvaluator(const MatI32 &rawDepth, const MatF32 &disparities, const PointVec &consideredDisparities, const MatD &camMatrix,
                         const MatD & /*distortion*/, const MatD &perspTransform, const MatD &rotation, const Vec3 &translation,
                         int xOffset, int yOffset)
        : distErrorMeans(2048), xErrorMeans(2048), yErrorMeans(2048), zErrorMeans(2048), distErrorSigmas(2048), xErrorSigmas(2048),
          yErrorSigmas(2048), zErrorSigmas(2048), binCounts(2048) {
        double q[16], k[9], r[9], t[3] = {translation[0], translation[1], translation[2]};
        flatten(perspTransform, 4, 4, q); flatten(camMatrix, 3, 3, k); flatten(rotation, 3, 3, r);
        svx::DeviceImage<int32_t> raw(rawDepth.rows, rawDepth.cols);
        svx::DeviceImage<float> disp(disparities.rows, disparities.cols);
        raw.upload(rawDepth); disp.upload(disparities);
        const std::vector<int32_t> xy = svx::flatten_points(consideredDisparities);
        svx::DeviceArray<int32_t> pts(xy.size());
        pts.upload(xy.data());
        std::vector<float> mean(4 * 2048), sigma(4 * 2048);
        float totals[10];
        svx::Runtime::check(svx_kinect_evaluate(svx::Runtime::ctx(), raw.d, rawDepth.rows, rawDepth.cols, disp.d, disparities.rows,
                                                disparities.cols, xy.empty() ? nullptr : pts.d, (int)(xy.size() / 2), xOffset, yOffset, q, k, r, t,
                                                mean.data(), sigma.data(), binCounts.data(), totals, &nbCorrespondences));
        std::copy(mean.begin(), mean.begin() + 2048, xErrorMeans.begin());
        std::copy(mean.begin() + 2048, mean.begin() + 4096, yErrorMeans.begin());
        std::copy(mean.begin() + 4096, mean.begin() + 6144, zErrorMeans.begin());
        std::copy(mean.begin() + 6144, mean.end(), distErrorMeans.begin());
        std::copy(sigma.begin(), sigma.begin() + 2048, xErrorSigmas.begin());
        std::copy(sigma.begin() + 2048, sigma.begin() + 4096, yErrorSigmas.begin());
        std::copy(sigma.begin() + 4096, sigma.begin() + 6144, zErrorSigmas.begin());
        std::copy(sigma.begin() + 6144, sigma.end(), distErrorSigmas.begin());
        meanXError = totals[0]; meanYError = totals[1]; meanZError = totals[2]; meanDistError = totals[3];
        sigmaXError = totals[4]; sigmaYError = totals[5]; sigmaZError = totals[6]; sigmaDistError = totals[7];
        meanRelDistError = totals[8]; sigmaRelDistError = totals[9];
    }

    int nbCorrespondences = 0;
    std::vector<float> distErrorMeans, xErrorMeans, yErrorMeans, zErrorMeans, distErrorSigmas, xErrorSigmas, yErrorSigmas, zErrorSigmas;
    std::vector<int32_t> binCounts;       // correspondences per raw depth value (the sizes of the reference's per-bin error lists)
    float meanDistError = 0, meanXError = 0, meanYError = 0, meanZError = 0, sigmaDistError = 0, sigmaXError = 0, sigmaYError = 0, sigmaZError = 0;
    float meanRelDistError = 0, sigmaRelDistError = 0;

private:
    template <class MatD>
    static void flatten(const MatD &m, int rows, int cols, double *out) {
        if (m.rows != rows || m.cols != cols) throw sparsestereo::Exception("PerformanceEvaluator: matrix of the wrong size");
        for (int i = 0; i < rows; ++i) std::memcpy(out + (size_t)i * cols, m.data + (size_t)i * m.step, sizeof(double) * cols);
    }
};
