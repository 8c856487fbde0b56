// orb_module.cpp -- pybind11 module `orb_project`: the reference's Python boundary
// (reference src/main_interface.cpp:104-110) re-hosted on liborbamd.so.
//
// extract(image) keeps the reference's contract: [H, W, 3] uint8 BGR in, tuple
// (keypoints float32 [N, 7], descriptors uint8 [Nv, 32]) out, both fresh NumPy-owned arrays, and
// RuntimeError for inputs the reference rejects (src/main_interface.cpp:17-21).  With zero
// keypoints the descriptor array has shape (0, 0), as an empty cv::Mat converts
// (src/main_interface.cpp:57-65, :89-98).
//
// Added entry points (the reference has no matcher in its module; it calls OpenCV from Python at
// views.py:162): extract_gray, extract_indexed, knn_match2, ratio_filter, match_ratio; profile / profile_read for measurement.
//
// This translation unit contains no image arithmetic; it only moves buffers through the C ABI.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/orbamd.h"

namespace py = pybind11;

namespace {

std::mutex g_mu;
orbamd_ctx* g_ctx = nullptr;

// The module's one context is shared by all Python threads and guarded by g_mu.  The mutex is only
// ever WAITED for with the GIL released: a thread that blocked on it while holding the GIL would
// deadlock against the owner, which needs the GIL back at the end of its own nogil section.
std::unique_lock<std::mutex> lock_context()
{
    std::unique_lock<std::mutex> lock(g_mu, std::defer_lock);
    {
        py::gil_scoped_release nogil;
        lock.lock();
    }
    return lock;
}

orbamd_ctx* ctx()
{
    if (!g_ctx) {
        int dev = 0;
        if (const char* e = std::getenv("ORBAMD_DEVICE")) dev = std::atoi(e);
        int st = orbamd_ctx_create(dev, &g_ctx);
        if (st != ORBAMD_OK)
            throw std::runtime_error(std::string("orb_project: cannot create a GPU context (") +
                                     orbamd_status_string(st) + "); this module has no CPU path");
    }
    return g_ctx;
}

void check(int st, const char* what)
{
    if (st == ORBAMD_OK) return;
    std::string msg = std::string("orb_project: ") + what + ": " + orbamd_status_string(st);
    if (g_ctx && st == ORBAMD_E_DEVICE) msg += std::string(" [") + orbamd_last_error(g_ctx) + "]";
    throw std::runtime_error(msg);
}

// Same acceptance rules and messages as the reference's numpy_uint8_or_float_to_mat
// (src/main_interface.cpp:14-21).  Returns a C-contiguous uint8 view/copy.
py::array_t<uint8_t> accept_image(const py::array& arr, bool want_bgr)
{
    if (!(arr.ndim() == 2 || (arr.ndim() == 3 && arr.shape(2) == 3)))
        throw std::runtime_error("Expect HxW (grey) or HxWx3 (BGR) image");
    const bool is_u8 = arr.dtype().is(py::dtype::of<uint8_t>());
    const bool is_f32 = arr.dtype().is(py::dtype::of<float>());
    if (!is_u8 && !is_f32) throw std::runtime_error("Image must be uint8 or float32");
    // The reference accepts these two at the door but cannot process them: a 2-D array fails
    // inside cvtColor(BGR2GRAY) (src/main_interface.cpp:83) and float32 pixels are then read as
    // bytes (src/orb.cpp:53).  Both are reported as errors here.
    if (want_bgr && arr.ndim() != 3)
        throw std::runtime_error("extract expects an HxWx3 BGR image (BGR2GRAY needs 3 channels)");
    if (!want_bgr && arr.ndim() != 2) throw std::runtime_error("extract_gray expects an HxW image");
    if (!is_u8) throw std::runtime_error("float32 images are not supported by the ORB kernels; pass uint8");
    return py::array_t<uint8_t, py::array::c_style>::ensure(arr);
}

struct Extracted {
    py::array_t<float> kp;
    py::array_t<uint8_t> desc;
    py::array_t<int32_t> vidx;
};

Extracted run_extract(const py::array& image, bool bgr)
{
    py::array_t<uint8_t> img = accept_image(image, bgr);
    const int H = (int)img.shape(0), W = (int)img.shape(1);
    std::unique_lock<std::mutex> lock = lock_context();
    orbamd_ctx* c = ctx();
    int n = 0, nv = 0, st;
    {
        py::gil_scoped_release nogil;
        st = bgr ? orbamd_extract_bgr(c, img.data(), H, W, &n, &nv)
                 : orbamd_extract_gray(c, img.data(), H, W, &n, &nv);
    }
    check(st, "extract");
    Extracted out;
    using Shape = std::vector<py::ssize_t>;
    out.kp = py::array_t<float>(Shape{n, 7});
    out.desc = py::array_t<uint8_t>(n == 0 ? Shape{0, 0} : Shape{nv, 32});
    out.vidx = py::array_t<int32_t>(Shape{nv});
    {
        py::gil_scoped_release nogil;
        st = orbamd_fetch(c, out.kp.mutable_data(), out.desc.mutable_data(), out.vidx.mutable_data());
    }
    check(st, "fetch");
    return out;
}

py::tuple extract(const py::array& image)
{
    Extracted e = run_extract(image, true);
    return py::make_tuple(e.kp, e.desc);
}

py::tuple extract_gray(const py::array& image)
{
    Extracted e = run_extract(image, false);
    return py::make_tuple(e.kp, e.desc);
}

py::tuple extract_indexed(const py::array& image)
{
    Extracted e = run_extract(image, image.ndim() == 3);
    return py::make_tuple(e.kp, e.desc, e.vidx);
}

py::array_t<uint8_t> as_desc(const py::array& a, const char* name)
{
    if (!a.dtype().is(py::dtype::of<uint8_t>()))
        throw std::runtime_error(std::string(name) + " must be uint8");
    if (a.ndim() != 2) throw std::runtime_error(std::string(name) + " must be 2-D");
    // an empty descriptor array from extract() has shape (0, 0)
    if (a.shape(0) != 0 && a.shape(1) != 32) throw std::runtime_error(std::string(name) + " must be [N, 32]");
    return py::array_t<uint8_t, py::array::c_style>::ensure(a);
}

py::tuple knn_match2(const py::array& d1, const py::array& d2)
{
    py::array_t<uint8_t> q = as_desc(d1, "query descriptors");
    py::array_t<uint8_t> t = as_desc(d2, "train descriptors");
    const int nq = (int)q.shape(0), nt = (int)t.shape(0);
    using Shape = std::vector<py::ssize_t>;
    py::array_t<int32_t> idx(Shape{nq, 2});
    py::array_t<int32_t> dist(Shape{nq, 2});
    std::unique_lock<std::mutex> lock = lock_context();
    orbamd_ctx* c = ctx();
    int st;
    {
        py::gil_scoped_release nogil;
        st = orbamd_knn2(c, q.data(), nq, t.data(), nt, idx.mutable_data(), dist.mutable_data());
    }
    check(st, "knn_match2");
    return py::make_tuple(idx, dist);
}

// knn_match2 + ratio_filter with one trip to the device: the knnMatch call and the ratio loop of reference
// views.py:162-171 together.
py::tuple match_ratio(const py::array& d1, const py::array& d2)
{
    py::array_t<uint8_t> q = as_desc(d1, "query descriptors");
    py::array_t<uint8_t> t = as_desc(d2, "train descriptors");
    const int nq = (int)q.shape(0), nt = (int)t.shape(0);
    using Shape = std::vector<py::ssize_t>;
    py::array_t<int32_t> idx(Shape{nq, 2}), dist(Shape{nq, 2});
    std::vector<int32_t> gq(nq), g2(nq), g1(nq);
    int m = 0, st;
    std::unique_lock<std::mutex> lock = lock_context();
    orbamd_ctx* c = ctx();
    {
        py::gil_scoped_release nogil;
        st = orbamd_knn2_ratio(c, q.data(), nq, t.data(), nt, idx.mutable_data(), dist.mutable_data(), gq.data(),
                               g2.data(), g1.data(), &m);
    }
    if (st == ORBAMD_E_BADARG && m == -1)
        throw py::value_error("not enough values to unpack (expected 2)");  // as views.py:167 would
    check(st, "match_ratio");
    auto head = [m](const std::vector<int32_t>& a) {
        py::array_t<int32_t> r(Shape{m});
        std::copy(a.data(), a.data() + m, r.mutable_data());
        return r;
    };
    return py::make_tuple(idx, dist, head(gq), head(g2), head(g1));
}

py::tuple ratio_filter(const py::array_t<int32_t, py::array::c_style | py::array::forcecast>& idx,
                       const py::array_t<int32_t, py::array::c_style | py::array::forcecast>& dist)
{
    if (idx.ndim() != 2 || dist.ndim() != 2 || idx.shape(1) != 2 || dist.shape(1) != 2 ||
        idx.shape(0) != dist.shape(0))
        throw std::runtime_error("ratio_filter expects idx and dist of shape [N, 2]");
    const int nq = (int)idx.shape(0);
    using Shape = std::vector<py::ssize_t>;
    py::array_t<int32_t> gq(Shape{nq}), g2(Shape{nq}), g1(Shape{nq});
    int m = 0, st;
    std::unique_lock<std::mutex> lock = lock_context();
    orbamd_ctx* c = ctx();
    {
        py::gil_scoped_release nogil;
        st = orbamd_ratio_filter(c, idx.data(), dist.data(), nq, gq.mutable_data(), g2.mutable_data(),
                                 g1.mutable_data(), &m);
    }
    if (st == ORBAMD_E_BADARG)
        throw py::value_error("not enough values to unpack (expected 2)");  // as views.py:167 would
    check(st, "ratio_filter");
    // hand back exactly m rows (fresh arrays, like everything else this module returns)
    auto head = [m](const py::array_t<int32_t>& a) {
        py::array_t<int32_t> r(Shape{m});
        std::copy(a.data(), a.data() + m, r.mutable_data());
        return r;
    };
    return py::make_tuple(head(gq), head(g2), head(g1));
}

// Per-kernel device time and the host's time shares inside the calls of this module (measurement aid; the
// reference only wraps its call in time.perf_counter, views.py:269-279).
void profile(bool enable)
{
    std::unique_lock<std::mutex> lock = lock_context();
    orbamd_ctx* c = ctx();
    check(orbamd_ctx_timing_enable(c, enable ? 1 : 0), "profile");
    check(orbamd_ctx_timing_reset(c), "profile");
}

py::dict profile_read()
{
    std::unique_lock<std::mutex> lock = lock_context();
    orbamd_ctx* c = ctx();
    double ms[ORBAMD_K_COUNT];
    long launches[ORBAMD_K_COUNT];
    double us[ORBAMD_HT_COUNT];
    check(orbamd_ctx_timing_read(c, ms, launches), "profile_read");
    check(orbamd_ctx_host_timing_read(c, us), "profile_read");
    py::dict kernels;
    for (int k = 0; k < ORBAMD_K_COUNT; ++k)
        if (launches[k]) kernels[orbamd_kernel_name(k)] = py::make_tuple(ms[k], launches[k]);
    py::dict host;
    host["h2d"] = us[ORBAMD_HT_H2D];
    host["enqueue"] = us[ORBAMD_HT_ENQUEUE];
    host["wait"] = us[ORBAMD_HT_WAIT];
    host["d2h"] = us[ORBAMD_HT_D2H];
    py::dict out;
    out["kernel_ms"] = kernels;
    out["host_us"] = host;
    return out;
}

}  // namespace

PYBIND11_MODULE(orb_project, m)
{
    m.doc() = "Feature extraction via ORB implementation (MI355X / HIP back end)";
    m.def("extract", &extract, py::arg("image"),
          "Extract keypoints (Nx7 float32) and descriptors (NxD uint8/float32) from image frame");
    m.def("extract_gray", &extract_gray, py::arg("image"),
          "extract() for an HxW uint8 frame that is already gray");
    m.def("extract_indexed", &extract_indexed, py::arg("image"),
          "extract() plus valid_idx int32 [Nv]: the keypoint row of each descriptor row");
    m.def("knn_match2", &knn_match2, py::arg("d1"), py::arg("d2"),
          "Hamming 2-nearest neighbours of each row of d1 in d2: (idx int32 [N,2], dist int32 [N,2])");
    m.def("ratio_filter", &ratio_filter, py::arg("idx"), py::arg("dist"),
          "Lowe ratio 0.75 filter: (queryIdx, second trainIdx, first trainIdx) of the kept rows");
    m.def("match_ratio", &match_ratio, py::arg("d1"), py::arg("d2"),
          "knn_match2 + ratio_filter in one call: (idx, dist, queryIdx, second trainIdx, first trainIdx)");
    m.def("profile", &profile, py::arg("enable"),
          "Switch per-kernel / host-phase timing of this module's calls on or off (and clear the totals)");
    m.def("profile_read", &profile_read,
          "{'kernel_ms': {kernel: (ms, launches)}, 'host_us': {h2d, enqueue, wait, d2h}} since profile(True)");
}
