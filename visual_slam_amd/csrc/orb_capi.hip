// orb_capi.hip -- the extern "C" surface of liborbamd.so (declared in include/orbamd.h): context,
// grow-only device work buffers, host<->device staging and kernel sequencing.  No computation on
// image or descriptor data happens on the host in this file.
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/orbamd.h"
#include "orb_internal.h"

using namespace orbamd;

// Scratch of orbamd_knn2_dev_noalloc: the partial top-2 keys of the train-axis split, which every
// other matching entry point keeps in a context-owned hipMalloc buffer.  Every GPU the library runs
// on holds its own copy of this array until the process ends.  In host code the NAME of the array is
// only a handle for hipGetSymbolAddress; the address kernels get is looked up once per device in
// orbamd_ctx_create (noalloc_scratch_of) and kept in the context.
#define ORBAMD_NOALLOC_SCRATCH_BYTES (4u << 20)
static __device__ __align__(256) unsigned char scratch[ORBAMD_NOALLOC_SCRATCH_BYTES];

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct TimedLaunch {
    int kind;
    hipEvent_t a, b;
};

struct PinnedBuf {
    uint8_t* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct orbamd_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;

    // work buffers
    DevBuf gray, bgr, mask, rowcnt, rowoff, kpxy, kpslot, counts, kp7, desc, vidx, angle;
    DevBuf q, t, idx, dist, good, partial, rq, rt2, rt1, rcount, risk;
    bool audit = false;       // parity audit on: the describe kernels count near-midpoint narrowings into `risk`
    int risk_frames = 0;      // frames the last audited call covered
    int matcher = ORBAMD_MATCHER_MATRIX;
    // detector parameters; the defaults are the reference's (src/orb.cpp:39, :70-96) and select its kernel
    int det_thr = 20, det_arc = 12, det_wrap = 0, det_nms = 0;
    int32_t* h_counts = nullptr;  // pinned, 4 ints
    void* noalloc_scratch = nullptr;  // device address of `scratch` on this context's GPU

    // state of the last single-frame extract
    bool have_extract = false;
    int last_n = 0, last_nv = 0;
    // Single-frame calls launch at a capacity instead of reading the keypoint count back first; the
    // capacity grows when a frame exceeds it.  The first `spec_n` keypoint rows / `spec_nv` descriptor rows of
    // the last extract were copied to pinned host memory speculatively, ahead of the call's one wait.
    int single_cap = 16384;
    int guess_rows = 8192;
    int spec_n = 0, spec_nv = 0;
    size_t off_kp = 0, off_desc = 0, off_vidx = 0;   // where those rows sit in out_block
    PinnedBuf in_stage, out_block, mt_stage;
    bool stage_h2d = true;
    double host_us[ORBAMD_HT_COUNT] = {0};

    // per-kernel timing
    bool timing = false;
    std::vector<TimedLaunch> pending;
    std::vector<hipEvent_t> pool;
    double ms[ORBAMD_K_COUNT] = {0};
    long launches[ORBAMD_K_COUNT] = {0};
};

namespace {

int fail(orbamd_ctx* c, hipError_t e, const char* what)
{
    c->err = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? ORBAMD_E_NOMEM : ORBAMD_E_DEVICE;
}

#define HIPCHK(c, expr)                                   \
    do {                                                  \
        hipError_t e_ = (expr);                           \
        if (e_ != hipSuccess) return fail((c), e_, #expr); \
    } while (0)

#define CHK(expr)                       \
    do {                                \
        int s_ = (expr);                \
        if (s_ != ORBAMD_OK) return s_; \
    } while (0)

int ensure(orbamd_ctx* c, DevBuf& b, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return ORBAMD_OK;
    if (b.p) {
        // buffers may still be in use by queued work
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    size_t want = bytes + bytes / 4;  // head room against repeated regrowth
    HIPCHK(c, hipMalloc(&b.p, want));
    b.bytes = want;
    return ORBAMD_OK;
}

int ensure_pinned(orbamd_ctx* c, PinnedBuf& b, size_t bytes)
{
    if (bytes == 0) bytes = 64;
    if (b.bytes >= bytes) return ORBAMD_OK;
    if (b.p) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipHostFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    const size_t want = bytes + bytes / 4;
    HIPCHK(c, hipHostMalloc((void**)&b.p, want, hipHostMallocDefault));
    b.bytes = want;
    return ORBAMD_OK;
}

// Host wall time of a region, added to one of the ORBAMD_HT_* shares while timing is on.
struct HostPhase {
    orbamd_ctx* c;
    int kind;
    std::chrono::steady_clock::time_point t0;
    HostPhase(orbamd_ctx* c_, int kind_) : c(c_), kind(kind_)
    {
        if (c->timing) t0 = std::chrono::steady_clock::now();
    }
    ~HostPhase()
    {
        if (c->timing)
            c->host_us[kind] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
};

hipEvent_t get_event(orbamd_ctx* c)
{
    if (!c->pool.empty()) {
        hipEvent_t e = c->pool.back();
        c->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// Brackets one kernel launch with events when timing is on.
struct Timed {
    orbamd_ctx* c;
    int kind;
    hipEvent_t a = nullptr, b = nullptr;
    Timed(orbamd_ctx* c_, int kind_) : c(c_), kind(kind_)
    {
        if (c->timing) {
            a = get_event(c);
            b = get_event(c);
            if (a && b) (void)hipEventRecord(a, c->stream);
            else {
                // no pair of events: this launch goes untimed, and a lone event goes back to the pool
                if (a) c->pool.push_back(a);
                if (b) c->pool.push_back(b);
                a = b = nullptr;
            }
        }
    }
    ~Timed()
    {
        if (c->timing && a && b) {
            (void)hipEventRecord(b, c->stream);
            c->pending.push_back({kind, a, b});
        }
    }
};

int post_launch(orbamd_ctx* c, const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, e, what);
    return ORBAMD_OK;
}

// Host -> device for a caller's (pageable) array: copied piece by piece into pinned memory owned by the
// context, each piece picked up from there by a copy KERNEL (pinned memory is mapped into the device's address
// space) while the host copies the next piece.  No copy-engine transfer is involved, so nothing on the stream
// waits for another queue.  The pinned buffer is reused by the next call, which is why every host-pointer
// entry point ends with a wait for the stream.  dst must be 16-byte aligned.
constexpr size_t INGEST_PIECE = 3u << 18;    // 768 KiB

int upload(orbamd_ctx* c, void* dst, const void* src, size_t bytes, PinnedBuf& stage, size_t stage_off = 0)
{
    HostPhase hp(c, ORBAMD_HT_H2D);
    if (!c->stage_h2d) {
        HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        return ORBAMD_OK;
    }
    CHK(ensure_pinned(c, stage, stage_off + bytes));
    for (size_t o = 0; o < bytes; o += INGEST_PIECE) {
        const size_t nb = bytes - o < INGEST_PIECE ? bytes - o : INGEST_PIECE;
        memcpy(stage.p + stage_off + o, (const uint8_t*)src + o, nb);
        launch_copy_bytes(stage.p + stage_off + o, (uint8_t*)dst + o, nb, c->stream);
    }
    return post_launch(c, "copy16_kernel");
}

// BGR frame -> gray in device memory: staged like upload(), but each piece is converted straight out of
// pinned memory, so the BGR pixels never exist in device memory.
int upload_bgr_as_gray(orbamd_ctx* c, uint8_t* d_gray, const uint8_t* bgr, size_t npix)
{
    if (!c->stage_h2d) {
        CHK(ensure(c, c->bgr, npix * 3));
        {
            HostPhase hp(c, ORBAMD_HT_H2D);
            HIPCHK(c, hipMemcpyAsync(c->bgr.p, bgr, npix * 3, hipMemcpyHostToDevice, c->stream));
        }
        Timed t(c, ORBAMD_K_BGR2GRAY);
        launch_bgr2gray((const uint8_t*)c->bgr.p, d_gray, (long)npix, c->stream);
        return ORBAMD_OK;
    }
    HostPhase hp(c, ORBAMD_HT_H2D);
    CHK(ensure_pinned(c, c->in_stage, npix * 3));
    const size_t piece = INGEST_PIECE / 3;              // pixels per piece, a multiple of 4
    static_assert((INGEST_PIECE / 3) % 4 == 0, "pieces start on a dword of both the BGR and the gray array");
    for (size_t o = 0; o < npix; o += piece) {
        const size_t np = npix - o < piece ? npix - o : piece;
        memcpy(c->in_stage.p + 3 * o, bgr + 3 * o, 3 * np);
        Timed t(c, ORBAMD_K_BGR2GRAY);
        launch_bgr2gray(c->in_stage.p + 3 * o, d_gray + o, (long)np, c->stream);
    }
    return ORBAMD_OK;
}

int drain_timing(orbamd_ctx* c)
{
    if (c->pending.empty()) return ORBAMD_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto& t : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            c->ms[t.kind] += ms;
            c->launches[t.kind] += 1;
        }
        c->pool.push_back(t.a);
        c->pool.push_back(t.b);
    }
    c->pending.clear();
    return ORBAMD_OK;
}

int enter(orbamd_ctx* c)
{
    if (!c) return ORBAMD_E_BADARG;
    HIPCHK(c, hipSetDevice(c->device));
    return ORBAMD_OK;
}

// Descriptor rows are fetched 16 bytes at a time.
bool misaligned16(const void* p) { return ((uintptr_t)p & 15) != 0; }

int check_image(int H, int W)
{
    // H * MW and H * W must stay well inside 32-bit grid and index arithmetic per frame.
    if (H <= 0 || W <= 0 || H > 32768 || W > 32768) return ORBAMD_E_BADARG;
    return ORBAMD_OK;
}

// FAST mask -> row counts -> row offsets + per-frame totals, for `frames` frames at d_gray.
int run_detect_counts(orbamd_ctx* c, const uint8_t* d_gray, int frames, int H, int W, int32_t* d_counts)
{
    const size_t MW = mask_words(W);
    CHK(ensure(c, c->mask, (size_t)frames * H * MW * 4));
    CHK(ensure(c, c->rowcnt, (size_t)frames * H * sizeof(int2)));
    CHK(ensure(c, c->rowoff, (size_t)frames * H * sizeof(int2)));
    FrameSet fs{d_gray, frames, H, W};
    const bool reference_detector = c->det_thr == 20 && c->det_arc == 12 && !c->det_wrap && !c->det_nms;
    {
        // the row totals start from zero; the fill is timed under its own kind, not as part of the detector
        Timed t(c, ORBAMD_K_ROWCOUNT);
        HIPCHK(c, hipMemsetAsync(c->rowcnt.p, 0, (size_t)frames * H * sizeof(int2), c->stream));
    }
    {
        Timed t(c, ORBAMD_K_FAST);
        if (reference_detector) launch_fast_mask(fs, (uint32_t*)c->mask.p, (int2*)c->rowcnt.p, c->stream);
        else launch_fast_ext(fs, c->det_thr, c->det_arc, c->det_wrap, c->det_nms, (uint32_t*)c->mask.p,
                             (int2*)c->rowcnt.p, c->stream);
    }
    CHK(post_launch(c, reference_detector ? "fast_mask_kernel" : "fastx_tile_kernel"));
    {
        Timed t(c, ORBAMD_K_ROWSCAN);
        launch_rowscan(fs, (const int2*)c->rowcnt.p, (int2*)c->rowoff.p, d_counts, c->stream);
    }
    CHK(post_launch(c, "rowscan_kernel"));
    return ORBAMD_OK;
}

int run_emit(orbamd_ctx* c, const uint8_t* d_gray, int frames, int H, int W, int cap, int32_t* d_counts)
{
    CHK(ensure(c, c->kpxy, (size_t)frames * cap * sizeof(int2)));
    CHK(ensure(c, c->kpslot, (size_t)frames * cap * sizeof(int32_t)));
    FrameSet fs{d_gray, frames, H, W};
    {
        Timed t(c, ORBAMD_K_EMIT);
        launch_emit(fs, (const uint32_t*)c->mask.p, (const int2*)c->rowcnt.p, (const int2*)c->rowoff.p, cap,
                    (int2*)c->kpxy.p, (int32_t*)c->kpslot.p, d_counts, c->stream);
    }
    return post_launch(c, "emit_kernel");
}

int read_counts(orbamd_ctx* c, const int32_t* d_counts, int nints)
{
    HIPCHK(c, hipMemcpyAsync(c->h_counts, d_counts, sizeof(int32_t) * nints, hipMemcpyDeviceToHost, c->stream));
    HostPhase hp(c, ORBAMD_HT_WAIT);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ORBAMD_OK;
}

// Train-axis splits so that small jobs still fill the chip: aim for >= `target` workgroups of
// `rows_per_block` queries, keep every train chunk >= 256 rows.
int choose_splits(int nq, int nt, int npairs, int rows_per_block, int target)
{
    long blocks = (long)((nq + rows_per_block - 1) / rows_per_block) * npairs;
    if (blocks <= 0) return 1;
    long s = (target + blocks - 1) / blocks;
    long smax = nt / 256;
    if (s > smax) s = smax;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return (int)s;
}

// Device address of `scratch` on `device` (which must be current), looked up once per device.
int noalloc_scratch_of(int device, void** out)
{
    static std::mutex mu;
    static void* ptr_of[64] = {nullptr};
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0 || device >= 64) return ORBAMD_E_BADARG;
    if (!ptr_of[device]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return ORBAMD_E_DEVICE;
        ptr_of[device] = p;
    }
    *out = ptr_of[device];
    return ORBAMD_OK;
}

// scratch / scratch_bytes: where the partial keys of a train-axis split go.  NULL = the context's own
// grow-only buffer (may allocate); otherwise the given memory is used as it is, the split count is
// lowered until it fits, and the launches are not bracketed by timing events (creating one allocates).
int run_knn2(orbamd_ctx* c, Knn2Args& a, void* scratch_mem = nullptr, size_t scratch_bytes = 0)
{
    const int nqmax = a.qcounts ? a.cap : a.nq_fixed;
    const int ntmax = a.tcounts ? a.cap : a.nt_fixed;
    const bool matrix = c->matcher == ORBAMD_MATCHER_MATRIX;
    a.splits = matrix ? choose_splits(nqmax, ntmax, a.npairs, 512, 1024)
                      : choose_splits(nqmax, ntmax, a.npairs, 256, 2048);
    if (scratch_mem) {
        const size_t per_split = (size_t)a.npairs * a.cap * 2 * sizeof(uint32_t);
        while (a.splits > 1 && per_split * a.splits > scratch_bytes) --a.splits;
        a.partial = (uint32_t*)scratch_mem;
        if (matrix) launch_knn2_mfma(a, c->stream);
        else launch_knn2(a, c->stream);
        CHK(post_launch(c, matrix ? "knn2_mfma_kernel" : "knn2_kernel"));
        if (a.splits > 1) {
            launch_knn2_merge(a, c->stream);
            CHK(post_launch(c, "knn2_merge_kernel"));
        }
        return ORBAMD_OK;
    }
    if (a.splits > 1) {
        CHK(ensure(c, c->partial, (size_t)a.npairs * a.splits * a.cap * 2 * sizeof(uint32_t)));
        a.partial = (uint32_t*)c->partial.p;
    }
    {
        Timed t(c, ORBAMD_K_KNN2);
        if (matrix) launch_knn2_mfma(a, c->stream);
        else launch_knn2(a, c->stream);
    }
    CHK(post_launch(c, matrix ? "knn2_mfma_kernel" : "knn2_kernel"));
    if (a.splits > 1) {
        {
            Timed t(c, ORBAMD_K_KNN2_MERGE);
            launch_knn2_merge(a, c->stream);
        }
        CHK(post_launch(c, "knn2_merge_kernel"));
    }
    return ORBAMD_OK;
}

// Parity audit: zeroed counters for `frames` frames when the audit is on, NULL otherwise.
int audit_buffer(orbamd_ctx* c, int frames, int32_t** out)
{
    *out = nullptr;
    c->risk_frames = 0;
    if (!c->audit) return ORBAMD_OK;
    CHK(ensure(c, c->risk, (size_t)frames * 4 * sizeof(int32_t)));
    HIPCHK(c, hipMemsetAsync(c->risk.p, 0, (size_t)frames * 4 * sizeof(int32_t), c->stream));
    c->risk_frames = frames;
    *out = (int32_t*)c->risk.p;
    return ORBAMD_OK;
}

// Orientation + descriptors for the scan-ordered lists in c->kpxy / c->kpslot (as run_emit leaves them).
// Frames made of whole aligned 16-byte units are described tile by tile from LDS, the others keypoint by
// keypoint; the results are the same.
int run_describe(orbamd_ctx* c, const FrameSet& fs, const int32_t* d_counts, int cap, float* d_kp7,
                 uint8_t* d_desc, int32_t* d_valid_idx)
{
    static const bool per_keypoint = getenv("ORBAMD_DESCRIBE_PER_KEYPOINT") != nullptr;   // (for A/B timing)
    const bool tiles = !per_keypoint && (fs.W & 15) == 0 && ((uintptr_t)fs.gray & 15) == 0;
    int32_t* risk = nullptr;
    CHK(audit_buffer(c, fs.frames, &risk));
    {
        Timed t(c, ORBAMD_K_DESCRIBE);
        if (tiles)
            launch_describe_tiles(fs, (const int2*)c->kpxy.p, (const int32_t*)c->kpslot.p, d_counts, cap,
                                  (const int2*)c->rowoff.p, d_kp7, d_desc, d_valid_idx, risk, c->stream);
        else
            launch_orient_describe(fs, (const int2*)c->kpxy.p, (const int32_t*)c->kpslot.p, d_counts, cap, nullptr,
                                   d_kp7, nullptr, d_desc, d_valid_idx, risk, c->stream);
    }
    return post_launch(c, tiles ? "describe_tile_kernel" : "orient_describe_kernel");
}

// Single-frame extract on a gray frame already in (or on its way into) c->gray.  Everything is launched
// at the context's single-frame capacity, so the keypoint count is not needed on the host before the
// last kernel; the counts and (speculatively, sized by the previous frame) the result rows travel to
// pinned host memory behind the kernels, and the call waits for the device once.
int extract_resident(orbamd_ctx* c, int H, int W, int* n_out, int* nv_out)
{
    c->have_extract = false;
    const uint8_t* d_gray = (const uint8_t*)c->gray.p;
    int n = 0, nv = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int cap = c->single_cap;
        {
            HostPhase hp(c, ORBAMD_HT_ENQUEUE);
            CHK(ensure(c, c->counts, 16));
            CHK(ensure(c, c->kp7, (size_t)cap * 7 * sizeof(float)));
            CHK(ensure(c, c->desc, (size_t)cap * 32));
            CHK(ensure(c, c->vidx, (size_t)cap * sizeof(int32_t)));
            CHK(run_detect_counts(c, d_gray, 1, H, W, (int32_t*)c->counts.p));
            CHK(run_emit(c, d_gray, 1, H, W, cap, (int32_t*)c->counts.p));
            FrameSet fs{d_gray, 1, H, W};
            CHK(run_describe(c, fs, (const int32_t*)c->counts.p, cap, (float*)c->kp7.p, (uint8_t*)c->desc.p,
                             (int32_t*)c->vidx.p));
        }
        const int g = c->guess_rows < cap ? c->guess_rows : cap;
        {
            // one kernel writes the counts and the live rows (at most g of each) into pinned host memory
            HostPhase hp(c, ORBAMD_HT_D2H);
            c->off_kp = 64;
            c->off_desc = c->off_kp + (((size_t)g * 28 + 15) & ~(size_t)15);
            c->off_vidx = c->off_desc + (size_t)g * 32;
            CHK(ensure_pinned(c, c->out_block, c->off_vidx + (size_t)g * 4));
            const int32_t* dc = (const int32_t*)c->counts.p;
            uint8_t* const h = c->out_block.p;
            EgressArgs e{};
            e.nseg = 4;
            e.seg[0] = {(const uint32_t*)c->counts.p, (uint32_t*)h, nullptr, 0, 2};
            e.seg[1] = {(const uint32_t*)c->kp7.p, (uint32_t*)(h + c->off_kp), dc, 7, g * 7};
            e.seg[2] = {(const uint32_t*)c->desc.p, (uint32_t*)(h + c->off_desc), dc + 1, 8, g * 8};
            e.seg[3] = {(const uint32_t*)c->vidx.p, (uint32_t*)(h + c->off_vidx), dc + 1, 1, g};
            launch_egress(e, c->stream);
            CHK(post_launch(c, "egress_kernel"));
        }
        {
            HostPhase hp(c, ORBAMD_HT_WAIT);
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        n = reinterpret_cast<const int32_t*>(c->out_block.p)[0];
        nv = reinterpret_cast<const int32_t*>(c->out_block.p)[1];
        if (n > ORBAMD_MAX_CAP) return ORBAMD_E_CAPACITY;
        c->spec_n = g;
        c->spec_nv = g;
        if (n <= cap) break;
        // more corners than the capacity: grow it and run the frame again (the second pass always fits)
        c->single_cap = n + n / 4 < ORBAMD_MAX_CAP ? n + n / 4 : ORBAMD_MAX_CAP;
        c->guess_rows = c->single_cap;
    }
    c->guess_rows = n + n / 8 + 64;
    c->last_n = n;
    c->last_nv = nv;
    c->have_extract = true;
    if (n_out) *n_out = n;
    if (nv_out) *nv_out = nv;
    return ORBAMD_OK;
}

}  // namespace

extern "C" {

int orbamd_version(void) { return 100; }

const char* orbamd_status_string(int status)
{
    switch (status) {
        case ORBAMD_OK: return "ok";
        case ORBAMD_E_BADARG: return "bad argument";
        case ORBAMD_E_CAPACITY: return "output capacity too small";
        case ORBAMD_E_DEVICE: return "HIP call failed";
        case ORBAMD_E_NOMEM: return "out of memory";
        case ORBAMD_E_STATE: return "call sequence wrong";
        default: return "unknown status";
    }
}

const char* orbamd_kernel_name(int kind)
{
    static const char* names[ORBAMD_K_COUNT] = {"bgr2gray", "fast_mask", "rowcount", "rowscan", "emit",
                                                "orient_describe", "knn2", "knn2_merge", "synth", "ratio_compact"};
    return (kind >= 0 && kind < ORBAMD_K_COUNT) ? names[kind] : "?";
}

int orbamd_ctx_create(int device, orbamd_ctx** out)
{
    if (!out) return ORBAMD_E_BADARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ORBAMD_E_DEVICE;
    if (device < 0 || device >= ndev) return ORBAMD_E_BADARG;
    orbamd_ctx* c = new (std::nothrow) orbamd_ctx();
    if (!c) return ORBAMD_E_NOMEM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void**)&c->h_counts, 64, hipHostMallocDefault) != hipSuccess) {
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
        delete c;
        return ORBAMD_E_DEVICE;
    }
    c->stream = c->own_stream;
    if (noalloc_scratch_of(device, &c->noalloc_scratch) != ORBAMD_OK) {
        (void)hipHostFree(c->h_counts);
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return ORBAMD_E_DEVICE;
    }
    if (const char* e = getenv("ORBAMD_MATCHER"))
        if (strcmp(e, "valu") == 0) c->matcher = ORBAMD_MATCHER_VALU;
    if (const char* e = getenv("ORBAMD_STAGE_COPIES"))      // (0: hand the caller's pageable arrays to HIP as they are)
        c->stage_h2d = atoi(e) != 0;
    *out = c;
    return ORBAMD_OK;
}

int orbamd_ctx_set_matcher(orbamd_ctx* c, int matcher)
{
    if (!c || (matcher != ORBAMD_MATCHER_MATRIX && matcher != ORBAMD_MATCHER_VALU)) return ORBAMD_E_BADARG;
    c->matcher = matcher;
    return ORBAMD_OK;
}

int orbamd_ctx_set_detector(orbamd_ctx* c, int threshold, int arc, int wrap, int nms)
{
    if (!c || threshold < 1 || threshold > 255 || arc < 9 || arc > 16) return ORBAMD_E_BADARG;
    c->det_thr = threshold;
    c->det_arc = arc;
    c->det_wrap = wrap != 0;
    c->det_nms = nms != 0;
    return ORBAMD_OK;
}

int orbamd_ctx_set_audit(orbamd_ctx* c, int on)
{
    if (!c) return ORBAMD_E_BADARG;
    c->audit = on != 0;
    c->risk_frames = 0;
    return ORBAMD_OK;
}

int orbamd_ctx_audit_read(orbamd_ctx* c, int32_t* out, int frames)
{
    CHK(enter(c));
    if (!out || frames <= 0) return ORBAMD_E_BADARG;
    if (!c->audit || c->risk_frames <= 0 || frames > c->risk_frames) return ORBAMD_E_STATE;
    HIPCHK(c, hipMemcpyAsync(out, c->risk.p, (size_t)frames * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ORBAMD_OK;
}

int orbamd_ctx_destroy(orbamd_ctx* c)
{
    if (!c) return ORBAMD_E_BADARG;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->gray, &c->bgr, &c->mask, &c->rowcnt, &c->rowoff, &c->kpxy, &c->kpslot, &c->counts,
                      &c->kp7, &c->desc, &c->vidx, &c->angle, &c->q, &c->t, &c->idx, &c->dist, &c->good,
                      &c->partial, &c->rq, &c->rt2, &c->rt1, &c->rcount, &c->risk};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (auto& t : c->pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto e : c->pool) (void)hipEventDestroy(e);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    PinnedBuf* pins[] = {&c->in_stage, &c->out_block, &c->mt_stage};
    for (PinnedBuf* b : pins)
        if (b->p) (void)hipHostFree(b->p);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return ORBAMD_OK;
}

int orbamd_ctx_reserve(orbamd_ctx* c, int frames, int H, int W, int cap)
{
    CHK(enter(c));
    if (frames <= 0 || frames > 65535 || cap <= 0 || cap > ORBAMD_MAX_CAP) return ORBAMD_E_BADARG;
    CHK(check_image(H, W));
    const size_t MW = mask_words(W);
    CHK(ensure(c, c->mask, (size_t)frames * H * MW * 4));
    CHK(ensure(c, c->rowcnt, (size_t)frames * H * sizeof(int2)));
    CHK(ensure(c, c->rowoff, (size_t)frames * H * sizeof(int2)));
    CHK(ensure(c, c->kpxy, (size_t)frames * cap * sizeof(int2)));
    CHK(ensure(c, c->kpslot, (size_t)frames * cap * sizeof(int32_t)));
    // the matcher's partial keys, for the consecutive pairs of such a batch (either kernel family)
    // (and for the single pair of orbamd_knn2_cross_dev, which is split further than the pairs of a large batch)
    if (cap <= ORBAMD_MAX_TRAIN) {
        size_t need = 0;
        const int batch_pairs = frames > 1 ? frames - 1 : 1;
        for (int npairs : {batch_pairs, 1}) {
            int splits = choose_splits(cap, cap, npairs, 512, 1024);
            const int sv = choose_splits(cap, cap, npairs, 256, 2048);
            splits = splits > sv ? splits : sv;
            const size_t b = splits > 1 ? (size_t)npairs * splits * cap * 2 * sizeof(uint32_t) : 0;
            need = b > need ? b : need;
        }
        if (need) CHK(ensure(c, c->partial, need));
    }
    return ORBAMD_OK;
}

int orbamd_ctx_set_stream(orbamd_ctx* c, void* hip_stream)
{
    CHK(enter(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    CHK(drain_timing(c));
    // The handle is used as given: NULL is HIP's default (null) stream, which is what a host framework
    // whose current stream is the default one hands over.
    c->stream = (hipStream_t)hip_stream;
    return ORBAMD_OK;
}

int orbamd_ctx_use_own_stream(orbamd_ctx* c)
{
    CHK(enter(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    CHK(drain_timing(c));
    c->stream = c->own_stream;
    return ORBAMD_OK;
}

int orbamd_ctx_sync(orbamd_ctx* c)
{
    CHK(enter(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ORBAMD_OK;
}

const char* orbamd_last_error(orbamd_ctx* c) { return c ? c->err.c_str() : "null context"; }

int orbamd_ctx_timing_enable(orbamd_ctx* c, int on)
{
    CHK(enter(c));
    CHK(drain_timing(c));
    c->timing = on != 0;
    return ORBAMD_OK;
}

int orbamd_ctx_timing_reset(orbamd_ctx* c)
{
    CHK(enter(c));
    CHK(drain_timing(c));
    for (int k = 0; k < ORBAMD_K_COUNT; ++k) { c->ms[k] = 0; c->launches[k] = 0; }
    for (int k = 0; k < ORBAMD_HT_COUNT; ++k) c->host_us[k] = 0;
    return ORBAMD_OK;
}

int orbamd_ctx_host_timing_read(orbamd_ctx* c, double us_out[ORBAMD_HT_COUNT])
{
    CHK(enter(c));
    if (!us_out) return ORBAMD_E_BADARG;
    for (int k = 0; k < ORBAMD_HT_COUNT; ++k) us_out[k] = c->host_us[k];
    return ORBAMD_OK;
}

int orbamd_ctx_timing_read(orbamd_ctx* c, double ms_out[ORBAMD_K_COUNT], long launches_out[ORBAMD_K_COUNT])
{
    CHK(enter(c));
    CHK(drain_timing(c));
    for (int k = 0; k < ORBAMD_K_COUNT; ++k) {
        if (ms_out) ms_out[k] = c->ms[k];
        if (launches_out) launches_out[k] = c->launches[k];
    }
    return ORBAMD_OK;
}

// ---- single frame, host pointers ---------------------------------------------------------------

int orbamd_bgr2gray(orbamd_ctx* c, const uint8_t* bgr, int H, int W, uint8_t* gray_out)
{
    CHK(enter(c));
    if (!bgr || !gray_out) return ORBAMD_E_BADARG;
    CHK(check_image(H, W));
    const size_t npix = (size_t)H * W;
    CHK(ensure(c, c->bgr, npix * 3));
    CHK(ensure(c, c->gray, npix));
    HIPCHK(c, hipMemcpyAsync(c->bgr.p, bgr, npix * 3, hipMemcpyHostToDevice, c->stream));
    {
        Timed t(c, ORBAMD_K_BGR2GRAY);
        launch_bgr2gray((const uint8_t*)c->bgr.p, (uint8_t*)c->gray.p, (long)npix, c->stream);
    }
    CHK(post_launch(c, "bgr2gray_kernel"));
    HIPCHK(c, hipMemcpyAsync(gray_out, c->gray.p, npix, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ORBAMD_OK;
}

int orbamd_detect(orbamd_ctx* c, const uint8_t* gray, int H, int W, int32_t* out_xy, int cap, int* n_out)
{
    CHK(enter(c));
    if (!gray || !n_out || cap < 0 || (cap > 0 && !out_xy)) return ORBAMD_E_BADARG;
    CHK(check_image(H, W));
    c->have_extract = false;
    const size_t npix = (size_t)H * W;
    CHK(ensure(c, c->gray, npix));
    CHK(ensure(c, c->counts, 16));
    HIPCHK(c, hipMemcpyAsync(c->gray.p, gray, npix, hipMemcpyHostToDevice, c->stream));
    CHK(run_detect_counts(c, (const uint8_t*)c->gray.p, 1, H, W, (int32_t*)c->counts.p));
    CHK(read_counts(c, (const int32_t*)c->counts.p, 2));
    const int n = c->h_counts[0];
    *n_out = n;
    int m = n < cap ? n : cap;
    m = m < ORBAMD_MAX_CAP ? m : ORBAMD_MAX_CAP;
    if (m > 0) {
        CHK(run_emit(c, (const uint8_t*)c->gray.p, 1, H, W, m, (int32_t*)c->counts.p));
        HIPCHK(c, hipMemcpyAsync(out_xy, c->kpxy.p, (size_t)m * sizeof(int2), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return n > cap ? ORBAMD_E_CAPACITY : ORBAMD_OK;
}

static int upload_keypoints(orbamd_ctx* c, const uint8_t* gray, int H, int W, const int32_t* xy, int n)
{
    const size_t npix = (size_t)H * W;
    CHK(ensure(c, c->gray, npix));
    CHK(ensure(c, c->kpxy, (size_t)n * sizeof(int2)));
    CHK(ensure(c, c->kpslot, (size_t)n * sizeof(int32_t)));
    CHK(ensure(c, c->counts, 16));
    HIPCHK(c, hipMemcpyAsync(c->gray.p, gray, npix, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->kpxy.p, xy, (size_t)n * sizeof(int2), hipMemcpyHostToDevice, c->stream));
    return ORBAMD_OK;
}

int orbamd_orient(orbamd_ctx* c, const uint8_t* gray, int H, int W, const int32_t* xy, int n, float* angle_out)
{
    CHK(enter(c));
    if (!gray || n < 0 || n > ORBAMD_MAX_CAP || (n > 0 && (!xy || !angle_out))) return ORBAMD_E_BADARG;
    CHK(check_image(H, W));
    if (n == 0) return ORBAMD_OK;
    c->have_extract = false;
    CHK(upload_keypoints(c, gray, H, W, xy, n));
    CHK(ensure(c, c->angle, (size_t)n * sizeof(float)));
    c->h_counts[0] = n;
    c->h_counts[1] = 0;
    HIPCHK(c, hipMemcpyAsync(c->counts.p, c->h_counts, 8, hipMemcpyHostToDevice, c->stream));
    FrameSet fs{(const uint8_t*)c->gray.p, 1, H, W};
    int32_t* risk = nullptr;
    CHK(audit_buffer(c, 1, &risk));
    {
        Timed t(c, ORBAMD_K_DESCRIBE);
        launch_orient_describe(fs, (const int2*)c->kpxy.p, (const int32_t*)c->kpslot.p,
                               (const int32_t*)c->counts.p, n, nullptr, nullptr, (float*)c->angle.p, nullptr,
                               nullptr, risk, c->stream);
    }
    CHK(post_launch(c, "orient_describe_kernel"));
    HIPCHK(c, hipMemcpyAsync(angle_out, c->angle.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ORBAMD_OK;
}

int orbamd_describe(orbamd_ctx* c, const uint8_t* gray, int H, int W, const int32_t* xy, const float* angle,
                    int n, uint8_t* desc_out, int32_t* valid_idx_out, int* nv_out)
{
    CHK(enter(c));
    if (!gray || !nv_out || n < 0 || n > ORBAMD_MAX_CAP || (n > 0 && (!xy || !angle || !desc_out)))
        return ORBAMD_E_BADARG;
    CHK(check_image(H, W));
    *nv_out = 0;
    if (n == 0) return ORBAMD_OK;
    c->have_extract = false;
    CHK(upload_keypoints(c, gray, H, W, xy, n));
    CHK(ensure(c, c->angle, (size_t)n * sizeof(float)));
    CHK(ensure(c, c->desc, (size_t)n * 32));
    CHK(ensure(c, c->vidx, (size_t)n * sizeof(int32_t)));
    HIPCHK(c, hipMemcpyAsync(c->angle.p, angle, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    launch_slots((const int2*)c->kpxy.p, n, H, W, (int32_t*)c->kpslot.p, (int32_t*)c->counts.p, c->stream);
    CHK(post_launch(c, "slots_kernel"));
    FrameSet fs{(const uint8_t*)c->gray.p, 1, H, W};
    int32_t* risk = nullptr;
    CHK(audit_buffer(c, 1, &risk));
    {
        Timed t(c, ORBAMD_K_DESCRIBE);
        launch_orient_describe(fs, (const int2*)c->kpxy.p, (const int32_t*)c->kpslot.p,
                               (const int32_t*)c->counts.p, n, (const float*)c->angle.p, nullptr, nullptr,
                               (uint8_t*)c->desc.p, (int32_t*)c->vidx.p, risk, c->stream);
    }
    CHK(post_launch(c, "orient_describe_kernel"));
    CHK(read_counts(c, (const int32_t*)c->counts.p, 2));
    const int nv = c->h_counts[1];
    *nv_out = nv;
    if (nv > 0) {
        HIPCHK(c, hipMemcpyAsync(desc_out, c->desc.p, (size_t)nv * 32, hipMemcpyDeviceToHost, c->stream));
        if (valid_idx_out)
            HIPCHK(c, hipMemcpyAsync(valid_idx_out, c->vidx.p, (size_t)nv * sizeof(int32_t),
                                     hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return ORBAMD_OK;
}

int orbamd_extract_gray(orbamd_ctx* c, const uint8_t* gray, int H, int W, int* n_out, int* nv_out)
{
    CHK(enter(c));
    if (!gray) return ORBAMD_E_BADARG;
    CHK(check_image(H, W));
    const size_t npix = (size_t)H * W;
    CHK(ensure(c, c->gray, npix));
    CHK(upload(c, c->gray.p, gray, npix, c->in_stage));
    return extract_resident(c, H, W, n_out, nv_out);
}

int orbamd_extract_bgr(orbamd_ctx* c, const uint8_t* bgr, int H, int W, int* n_out, int* nv_out)
{
    CHK(enter(c));
    if (!bgr) return ORBAMD_E_BADARG;
    CHK(check_image(H, W));
    const size_t npix = (size_t)H * W;
    CHK(ensure(c, c->gray, npix));
    CHK(upload_bgr_as_gray(c, (uint8_t*)c->gray.p, bgr, npix));
    CHK(post_launch(c, "bgr2gray_kernel"));
    return extract_resident(c, H, W, n_out, nv_out);
}

int orbamd_fetch(orbamd_ctx* c, float* kp7_out, uint8_t* desc_out, int32_t* valid_idx_out)
{
    CHK(enter(c));
    if (!c->have_extract) return ORBAMD_E_STATE;
    const int n = c->last_n, nv = c->last_nv;
    HostPhase hp(c, ORBAMD_HT_D2H);
    // rows the extract call already brought to pinned host memory are copied from there; a frame that
    // outgrew the guess is fetched from the device
    bool wait = false;
    if (n > 0 && kp7_out) {
        if (n <= c->spec_n) memcpy(kp7_out, c->out_block.p + c->off_kp, (size_t)n * 7 * sizeof(float));
        else {
            HIPCHK(c, hipMemcpyAsync(kp7_out, c->kp7.p, (size_t)n * 7 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
            wait = true;
        }
    }
    if (nv > 0 && nv <= c->spec_nv) {
        if (desc_out) memcpy(desc_out, c->out_block.p + c->off_desc, (size_t)nv * 32);
        if (valid_idx_out) memcpy(valid_idx_out, c->out_block.p + c->off_vidx, (size_t)nv * sizeof(int32_t));
    } else if (nv > 0) {
        if (desc_out)
            HIPCHK(c, hipMemcpyAsync(desc_out, c->desc.p, (size_t)nv * 32, hipMemcpyDeviceToHost, c->stream));
        if (valid_idx_out)
            HIPCHK(c, hipMemcpyAsync(valid_idx_out, c->vidx.p, (size_t)nv * sizeof(int32_t), hipMemcpyDeviceToHost,
                                     c->stream));
        wait = true;
    }
    if (wait) HIPCHK(c, hipStreamSynchronize(c->stream));
    return ORBAMD_OK;
}

int orbamd_knn2(orbamd_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx_out,
                int32_t* dist_out)
{
    CHK(enter(c));
    if (nq < 0 || nt < 0 || nt >= ORBAMD_MAX_TRAIN) return ORBAMD_E_BADARG;
    if (nq == 0) return ORBAMD_OK;
    if (!q || !idx_out || !dist_out || (nt > 0 && !t)) return ORBAMD_E_BADARG;
    CHK(ensure(c, c->q, (size_t)nq * 32));
    CHK(ensure(c, c->t, (size_t)nt * 32));
    CHK(ensure(c, c->idx, (size_t)nq * 2 * sizeof(int32_t)));
    CHK(ensure(c, c->dist, (size_t)nq * 2 * sizeof(int32_t)));
    // both descriptor sets through one pinned staging buffer (query rows first)
    CHK(ensure_pinned(c, c->mt_stage, (size_t)(nq + nt) * 32));
    CHK(upload(c, c->q.p, q, (size_t)nq * 32, c->mt_stage, 0));
    if (nt > 0) CHK(upload(c, c->t.p, t, (size_t)nt * 32, c->mt_stage, (size_t)nq * 32));
    Knn2Args a{};
    a.qdesc = (const uint8_t*)c->q.p;
    a.tdesc = (const uint8_t*)c->t.p;
    a.npairs = 1;
    a.cap = nq;
    a.nq_fixed = nq;
    a.nt_fixed = nt;
    a.idx = (int32_t*)c->idx.p;
    a.dist = (int32_t*)c->dist.p;
    {
        HostPhase hp(c, ORBAMD_HT_ENQUEUE);
        CHK(run_knn2(c, a));
    }
    // results to pinned memory by a kernel behind the matcher, one wait, then into the caller's arrays
    const size_t ob = (size_t)nq * 2 * sizeof(int32_t);
    CHK(ensure_pinned(c, c->out_block, 2 * ob));
    c->spec_n = c->spec_nv = 0;          // the block no longer holds the last extract's rows
    {
        HostPhase hp(c, ORBAMD_HT_D2H);
        EgressArgs e{};
        e.nseg = 2;
        e.seg[0] = {(const uint32_t*)c->idx.p, (uint32_t*)c->out_block.p, nullptr, 0, nq * 2};
        e.seg[1] = {(const uint32_t*)c->dist.p, (uint32_t*)(c->out_block.p + ob), nullptr, 0, nq * 2};
        launch_egress(e, c->stream);
        CHK(post_launch(c, "egress_kernel"));
    }
    {
        HostPhase hp(c, ORBAMD_HT_WAIT);
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    HostPhase hp(c, ORBAMD_HT_D2H);
    memcpy(idx_out, c->out_block.p, ob);
    memcpy(dist_out, c->out_block.p + ob, ob);
    return ORBAMD_OK;
}

int orbamd_knn2_ratio(orbamd_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx_out,
                      int32_t* dist_out, int32_t* good_q, int32_t* good_t2, int32_t* good_t1, int* m_out)
{
    CHK(enter(c));
    if (nq < 0 || nt < 0 || nt >= ORBAMD_MAX_TRAIN || !m_out) return ORBAMD_E_BADARG;
    *m_out = 0;
    if (nq == 0) return ORBAMD_OK;
    if (!q || !idx_out || !dist_out || !good_q || !good_t2 || !good_t1 || (nt > 0 && !t)) return ORBAMD_E_BADARG;
    const size_t b1 = (size_t)nq * sizeof(int32_t), b2 = 2 * b1;
    CHK(ensure(c, c->q, (size_t)nq * 32));
    CHK(ensure(c, c->t, (size_t)nt * 32));
    CHK(ensure(c, c->idx, b2));
    CHK(ensure(c, c->dist, b2));
    CHK(ensure(c, c->rq, b1));
    CHK(ensure(c, c->rt2, b1));
    CHK(ensure(c, c->rt1, b1));
    CHK(ensure(c, c->rcount, 16));
    CHK(ensure_pinned(c, c->mt_stage, (size_t)(nq + nt) * 32));
    CHK(ensure_pinned(c, c->out_block, 64 + 2 * b2 + 3 * b1));
    CHK(upload(c, c->q.p, q, (size_t)nq * 32, c->mt_stage, 0));
    if (nt > 0) CHK(upload(c, c->t.p, t, (size_t)nt * 32, c->mt_stage, (size_t)nq * 32));
    {
        HostPhase hp(c, ORBAMD_HT_ENQUEUE);
        Knn2Args a{};
        a.qdesc = (const uint8_t*)c->q.p;
        a.tdesc = (const uint8_t*)c->t.p;
        a.npairs = 1;
        a.cap = nq;
        a.nq_fixed = nq;
        a.nt_fixed = nt;
        a.idx = (int32_t*)c->idx.p;
        a.dist = (int32_t*)c->dist.p;
        CHK(run_knn2(c, a));
        {
            Timed tm(c, ORBAMD_K_RATIO);
            launch_ratio_compact((const int32_t*)c->idx.p, (const int32_t*)c->dist.p, nq, (int32_t*)c->rq.p,
                                 (int32_t*)c->rt2.p, (int32_t*)c->rt1.p, (int32_t*)c->rcount.p, c->stream);
        }
        CHK(post_launch(c, "ratio_compact_kernel"));
    }
    uint8_t* const h = c->out_block.p;   // count | idx | dist | good_q | good_t2 | good_t1
    c->spec_n = c->spec_nv = 0;          // the block no longer holds the last extract's rows
    {
        HostPhase hp(c, ORBAMD_HT_D2H);
        const int32_t* rc = (const int32_t*)c->rcount.p;
        EgressArgs e{};
        e.nseg = 6;
        e.seg[0] = {(const uint32_t*)c->rcount.p, (uint32_t*)h, nullptr, 0, 1};
        e.seg[1] = {(const uint32_t*)c->idx.p, (uint32_t*)(h + 64), nullptr, 0, nq * 2};
        e.seg[2] = {(const uint32_t*)c->dist.p, (uint32_t*)(h + 64 + b2), nullptr, 0, nq * 2};
        e.seg[3] = {(const uint32_t*)c->rq.p, (uint32_t*)(h + 64 + 2 * b2), rc, 1, nq};
        e.seg[4] = {(const uint32_t*)c->rt2.p, (uint32_t*)(h + 64 + 2 * b2 + b1), rc, 1, nq};
        e.seg[5] = {(const uint32_t*)c->rt1.p, (uint32_t*)(h + 64 + 2 * b2 + 2 * b1), rc, 1, nq};
        launch_egress(e, c->stream);
        CHK(post_launch(c, "egress_kernel"));
    }
    {
        HostPhase hp(c, ORBAMD_HT_WAIT);
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    HostPhase hp(c, ORBAMD_HT_D2H);
    memcpy(idx_out, h + 64, b2);
    memcpy(dist_out, h + 64 + b2, b2);
    const int m = reinterpret_cast<const int32_t*>(h)[0];
    *m_out = m;
    if (m < 0) return ORBAMD_E_BADARG;   // some query has fewer than two neighbours
    memcpy(good_q, h + 64 + 2 * b2, (size_t)m * 4);
    memcpy(good_t2, h + 64 + 2 * b2 + b1, (size_t)m * 4);
    memcpy(good_t1, h + 64 + 2 * b2 + 2 * b1, (size_t)m * 4);
    return ORBAMD_OK;
}

int orbamd_ratio_filter(orbamd_ctx* c, const int32_t* idx, const int32_t* dist, int nq, int32_t* good_q,
                        int32_t* good_t2, int32_t* good_t1, int* m_out)
{
    CHK(enter(c));
    if (nq < 0 || !m_out) return ORBAMD_E_BADARG;
    *m_out = 0;
    if (nq == 0) return ORBAMD_OK;
    if (!idx || !dist || !good_q || !good_t2 || !good_t1) return ORBAMD_E_BADARG;
    const size_t b2 = (size_t)nq * 2 * sizeof(int32_t), b1 = (size_t)nq * sizeof(int32_t);
    CHK(ensure(c, c->idx, b2));
    CHK(ensure(c, c->dist, b2));
    CHK(ensure(c, c->rq, b1));
    CHK(ensure(c, c->rt2, b1));
    CHK(ensure(c, c->rt1, b1));
    CHK(ensure(c, c->rcount, 16));
    HIPCHK(c, hipMemcpyAsync(c->idx.p, idx, b2, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dist.p, dist, b2, hipMemcpyHostToDevice, c->stream));
    {
        Timed t(c, ORBAMD_K_RATIO);
        launch_ratio_compact((const int32_t*)c->idx.p, (const int32_t*)c->dist.p, nq, (int32_t*)c->rq.p,
                             (int32_t*)c->rt2.p, (int32_t*)c->rt1.p, (int32_t*)c->rcount.p, c->stream);
    }
    CHK(post_launch(c, "ratio_compact_kernel"));
    CHK(read_counts(c, (const int32_t*)c->rcount.p, 1));
    const int m = c->h_counts[0];
    if (m < 0) return ORBAMD_E_BADARG;
    *m_out = m;
    if (m > 0) {
        HIPCHK(c, hipMemcpyAsync(good_q, c->rq.p, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(good_t2, c->rt2.p, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(good_t1, c->rt1.p, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return ORBAMD_OK;
}

// ---- frame batches, device pointers ------------------------------------------------------------

int orbamd_synth_frames_dev(orbamd_ctx* c, uint8_t* d_gray, int frames, int H, int W, int first_frame,
                            uint32_t seed, int cell, int density_thr)
{
    CHK(enter(c));
    if (!d_gray || frames <= 0 || cell < 10 || density_thr < 0 || first_frame < 0) return ORBAMD_E_BADARG;
    CHK(check_image(H, W));
    {
        Timed t(c, ORBAMD_K_SYNTH);
        launch_synth(d_gray, frames, H, W, first_frame, seed, cell, density_thr, c->stream);
    }
    return post_launch(c, "synth_kernel");
}

int orbamd_synth_textured_dev(orbamd_ctx* c, uint8_t* d_gray, int frames, int H, int W, int first_frame,
                              uint32_t seed)
{
    CHK(enter(c));
    if (!d_gray || frames <= 0 || first_frame < 0) return ORBAMD_E_BADARG;
    CHK(check_image(H, W));
    {
        Timed t(c, ORBAMD_K_SYNTH);
        launch_synth(d_gray, frames, H, W, first_frame, seed, 0, 0, c->stream);
    }
    return post_launch(c, "synth_kernel");
}

int orbamd_trig_selftest_dev(orbamd_ctx* c, const int32_t* d_m01, const int32_t* d_m10, int n, double* d_deg,
                             uint8_t* d_unsure, const double* d_rad, int m, double* d_sin, double* d_cos)
{
    CHK(enter(c));
    if (n < 0 || m < 0 || (n > 0 && (!d_m01 || !d_m10 || !d_deg || !d_unsure)) || (m > 0 && (!d_rad || !d_sin || !d_cos)))
        return ORBAMD_E_BADARG;
    if (n == 0 && m == 0) return ORBAMD_OK;
    launch_trig_selftest(d_m01, d_m10, n, d_deg, d_unsure, d_rad, m, d_sin, d_cos, c->stream);
    return post_launch(c, "trig_selftest_kernel");
}

int orbamd_bgr2gray_dev(orbamd_ctx* c, const uint8_t* d_bgr, int frames, int H, int W, uint8_t* d_gray)
{
    CHK(enter(c));
    if (!d_bgr || !d_gray || frames <= 0) return ORBAMD_E_BADARG;
    CHK(check_image(H, W));
    {
        Timed t(c, ORBAMD_K_BGR2GRAY);
        launch_bgr2gray(d_bgr, d_gray, (long)frames * H * W, c->stream);
    }
    return post_launch(c, "bgr2gray_kernel");
}

int orbamd_extract_batch_dev(orbamd_ctx* c, const uint8_t* d_gray, int frames, int H, int W, int cap,
                             float* d_kp7, uint8_t* d_desc, int32_t* d_valid_idx, int32_t* d_counts)
{
    CHK(enter(c));
    // (cap: keypoint indices are turned into 32-bit byte offsets inside a frame's arrays)
    if (!d_gray || !d_kp7 || !d_desc || !d_counts || frames <= 0 || frames > 65535 || cap <= 0 ||
        cap > ORBAMD_MAX_CAP)
        return ORBAMD_E_BADARG;
    CHK(check_image(H, W));
    c->have_extract = false;
    CHK(run_detect_counts(c, d_gray, frames, H, W, d_counts));
    CHK(run_emit(c, d_gray, frames, H, W, cap, d_counts));
    FrameSet fs{d_gray, frames, H, W};
    return run_describe(c, fs, d_counts, cap, d_kp7, d_desc, d_valid_idx);
}

// MEASUREMENT entry point (not part of include/orbamd.h): orbamd_extract_batch_dev for one batch whose
// orientation + descriptor kernel is fused with orbamd_knn2_batch_dev over the consecutive pairs of ANOTHER,
// already extracted batch (d_mdesc, d_mcounts) into one launch with role-specialised workgroups.
int orbamd_x_describe_match_dev(orbamd_ctx* c, const uint8_t* d_gray, int frames, int H, int W, int cap, float* d_kp7,
                                uint8_t* d_desc, int32_t* d_valid_idx, int32_t* d_counts, const uint8_t* d_mdesc,
                                const int32_t* d_mcounts, int mframes, int mcap, int32_t* d_idx, int32_t* d_dist,
                                uint8_t* d_good, int qt)
{
    CHK(enter(c));
    if (!d_gray || !d_kp7 || !d_desc || !d_counts || !d_mdesc || !d_mcounts || !d_idx || !d_dist || frames <= 0 ||
        frames > 65535 || cap <= 0 || cap > ORBAMD_MAX_CAP || mframes < 2 || mcap <= 0 || mcap > ORBAMD_MAX_TRAIN ||
        (qt != 2 && qt != 4) || (W & 15) || ((uintptr_t)d_gray & 15) || misaligned16(d_mdesc))
        return ORBAMD_E_BADARG;
    CHK(check_image(H, W));
    c->have_extract = false;
    CHK(run_detect_counts(c, d_gray, frames, H, W, d_counts));
    CHK(run_emit(c, d_gray, frames, H, W, cap, d_counts));
    FrameSet fs{d_gray, frames, H, W};
    {
        Timed t(c, ORBAMD_K_DESCRIBE);
        launch_describe_match(fs, (const int2*)c->kpxy.p, (const int32_t*)c->kpslot.p, d_counts, cap,
                              (const int2*)c->rowoff.p, d_kp7, d_desc, d_valid_idx, d_mdesc, d_mcounts, mframes, mcap,
                              d_idx, d_dist, d_good, qt, c->stream);
    }
    return post_launch(c, "describe_match_kernel");
}

int orbamd_knn2_batch_dev(orbamd_ctx* c, const uint8_t* d_desc, const int32_t* d_counts, int frames, int cap,
                          const int32_t* d_pairs, int npairs, int32_t* d_idx, int32_t* d_dist, uint8_t* d_good)
{
    CHK(enter(c));
    if (!d_desc || !d_counts || !d_idx || !d_dist || frames <= 0 || cap <= 0 || cap > ORBAMD_MAX_TRAIN ||
        npairs < 0 || npairs > 65535)
        return ORBAMD_E_BADARG;
    if (!d_pairs && npairs > frames - 1) return ORBAMD_E_BADARG;
    if (misaligned16(d_desc)) return ORBAMD_E_BADARG;
    if (npairs == 0) return ORBAMD_OK;
    Knn2Args a{};
    a.qdesc = d_desc;
    a.tdesc = d_desc;
    a.qcounts = d_counts;
    a.tcounts = d_counts;
    a.pairs = d_pairs;
    a.npairs = npairs;
    a.nframes = frames;
    a.cap = cap;
    a.idx = d_idx;
    a.dist = d_dist;
    a.good = d_good;
    return run_knn2(c, a);
}

int orbamd_ratio_filter_batch_dev(orbamd_ctx* c, const int32_t* d_idx, const uint8_t* d_good,
                                  const int32_t* d_counts, int frames, int cap, const int32_t* d_pairs,
                                  int npairs, int32_t* d_good_q, int32_t* d_good_t2, int32_t* d_good_t1,
                                  int32_t* d_kept)
{
    CHK(enter(c));
    if (!d_idx || !d_good || !d_counts || !d_good_q || !d_good_t2 || !d_good_t1 || !d_kept || frames <= 0 ||
        cap <= 0 || npairs < 0 || npairs > 65535)
        return ORBAMD_E_BADARG;
    if (!d_pairs && npairs > frames - 1) return ORBAMD_E_BADARG;
    if (npairs == 0) return ORBAMD_OK;
    {
        Timed t(c, ORBAMD_K_RATIO);
        launch_ratio_compact_batch(d_idx, d_good, d_counts, d_pairs, npairs, frames, cap, d_good_q, d_good_t2,
                                   d_good_t1, d_kept, c->stream);
    }
    return post_launch(c, "ratio_compact_batch_kernel");
}

int orbamd_knn2_cross_dev(orbamd_ctx* c, const uint8_t* d_qdesc, const int32_t* d_qcounts, int qframe,
                          const uint8_t* d_tdesc, const int32_t* d_tcounts, int tframe, int cap, int32_t* d_idx,
                          int32_t* d_dist, uint8_t* d_good)
{
    CHK(enter(c));
    if (!d_qdesc || !d_qcounts || !d_tdesc || !d_tcounts || !d_idx || !d_dist || qframe < 0 || tframe < 0 ||
        cap <= 0 || cap > ORBAMD_MAX_TRAIN || misaligned16(d_qdesc) || misaligned16(d_tdesc))
        return ORBAMD_E_BADARG;
    Knn2Args a{};
    a.qdesc = d_qdesc;
    a.tdesc = d_tdesc;
    a.qcounts = d_qcounts;
    a.tcounts = d_tcounts;
    a.qframe0 = qframe - 1;  // the kernel's consecutive-pair rule is (p + 1 + qframe0, p + tframe0)
    a.tframe0 = tframe;
    a.npairs = 1;
    a.cap = cap;
    a.idx = d_idx;
    a.dist = d_dist;
    a.good = d_good;
    return run_knn2(c, a);
}

int orbamd_knn2_dev(orbamd_ctx* c, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx,
                    int32_t* d_dist, uint8_t* d_good)
{
    CHK(enter(c));
    if (nq < 0 || nt < 0 || nt >= ORBAMD_MAX_TRAIN) return ORBAMD_E_BADARG;
    if (nq == 0) return ORBAMD_OK;
    if (!d_q || !d_idx || !d_dist || (nt > 0 && !d_t)) return ORBAMD_E_BADARG;
    if (misaligned16(d_q) || misaligned16(d_t)) return ORBAMD_E_BADARG;
    Knn2Args a{};
    a.qdesc = d_q;
    a.tdesc = d_t ? d_t : d_q;
    a.npairs = 1;
    a.cap = nq;
    a.nq_fixed = nq;
    a.nt_fixed = nt;
    a.idx = d_idx;
    a.dist = d_dist;
    a.good = d_good;
    return run_knn2(c, a);
}

int orbamd_knn2_dev_noalloc(orbamd_ctx* c, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt,
                            int32_t* d_idx, int32_t* d_dist, uint8_t* d_good)
{
    CHK(enter(c));
    if (nq < 0 || nt < 0 || nt >= ORBAMD_MAX_TRAIN) return ORBAMD_E_BADARG;
    if (nq > ORBAMD_NOALLOC_MAX_QUERIES) {
        c->err = "orbamd_knn2_dev_noalloc: nq exceeds ORBAMD_NOALLOC_MAX_QUERIES";
        return ORBAMD_E_CAPACITY;
    }
    if (nq == 0) return ORBAMD_OK;
    if (!d_q || !d_idx || !d_dist || (nt > 0 && !d_t)) return ORBAMD_E_BADARG;
    if (misaligned16(d_q) || misaligned16(d_t)) return ORBAMD_E_BADARG;
    if (!c->noalloc_scratch) return ORBAMD_E_STATE;
    Knn2Args a{};
    a.qdesc = d_q;
    a.tdesc = d_t ? d_t : d_q;
    a.npairs = 1;
    a.cap = nq;
    a.nq_fixed = nq;
    a.nt_fixed = nt;
    a.idx = d_idx;
    a.dist = d_dist;
    a.good = d_good;
    // Every (split, query) slot of the scratch that the merge reads is written by the matcher kernel
    // of the same call first, so nothing has to be cleared between calls.
    return run_knn2(c, a, c->noalloc_scratch, ORBAMD_NOALLOC_SCRATCH_BYTES);
}

}  // extern "C"
