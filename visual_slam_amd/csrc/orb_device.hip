// orb_device.hip -- gfx950 kernels of liborbamd.so and their launchers.
//
// Kernels (one wavefront = 64 lanes throughout):
//   bgr2gray_kernel         byte stream, 4 pixels per lane
//   synth_kernel            seeded synthetic frames, 4 pixels per lane
//   fast_mask_kernel        FAST segment test on LDS-staged tiles -> 1 bit per pixel, and per image
//                           row the number of corners / of corners that get a descriptor
//   rowscan_kernel          per frame: exclusive prefix over the row counts
//   emit_kernel             ordered expansion of the bit mask into (x, y) lists (ballot-free
//                           popcount + wave prefix; ascending y then x by construction)
//   slots_kernel            descriptor row numbers for caller-supplied keypoint lists
//   orient_describe_kernel  256 keypoints per workgroup in x order, 64 per wavefront: 31x31 moments
//                           (four keypoints per step), atan2 / cos / sin once per lane, steered
//                           BRIEF-256 from an LDS-staged window
//   knn2_mfma_kernel        Hamming top-2 on the matrix cores (default); descriptor bits become +-1 FP4
//                           operands on the way into registers / LDS
//   knn2_kernel             Hamming top-2 by XOR + popcount, one query per lane, train rows broadcast
//   knn2_merge_kernel       merges train-axis partial results
//   ratio_compact_kernel, ratio_compact_batch_kernel   ordered lists of the matches that pass the ratio test
//
// Floating point on the parity path is compiled without contraction: the reference computes
// x*c - y*s as two rounded products and one rounded difference (reference src/orb.cpp:195-199).
#include "orb_internal.h"
#include "orb_pattern.h"
#include "orb_trig.h"

#include <stdlib.h>

#pragma clang fp contract(off)

namespace orbamd {

// Pair-major sampling pattern in constant memory: bytes 4k .. 4k+3 = {x1, y1, x2, y2} of pair k.
__constant__ __attribute__((aligned(16))) int8_t c_pairs[ORBAMD_PATTERN_PAIRS * 4] = ORBAMD_PATTERN_QUADS_INIT;
// The same table as doubles (built at compile time): a lane's four operands of a round are 32 contiguous bytes,
// fetched by two loads instead of being unpacked and converted by a dozen vector instructions.
struct PairsAsDoubles { double v[ORBAMD_PATTERN_PAIRS * 4]; };
constexpr PairsAsDoubles pairs_as_doubles()
{
    constexpr int8_t q[ORBAMD_PATTERN_PAIRS * 4] = ORBAMD_PATTERN_QUADS_INIT;
    PairsAsDoubles t{};
    for (int i = 0; i < ORBAMD_PATTERN_PAIRS * 4; ++i) t.v[i] = (double)q[i];
    return t;
}
__constant__ __attribute__((aligned(32))) PairsAsDoubles c_pairs_d = pairs_as_doubles();

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Index of this wave inside its workgroup as a SCALAR.  threadIdx.x >> 6 is the same in all 64
// lanes, but the compiler cannot prove it and would keep everything derived from it (row numbers,
// row pointers, LDS slab addresses) in vector registers with per-lane 64-bit address arithmetic.
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Work-item number of this workgroup in a 1-D grid, remapped so that each XCD walks a contiguous
// run of items.  Workgroups are dealt round-robin to the 8 XCDs (and, inside an XCD, round-robin
// to its shader engines), so with `id % 8` fixed per XCD the items id/8 of one XCD are consecutive
// logical items.  Callers whose items come in groups where only a leading part has work (a frame's
// keypoint batches, a pair's query blocks) make the group size ODD so that the idle items do not
// fall on the same engines for every group.  Affects speed only, never results.
__device__ __forceinline__ uint32_t xcd_contiguous_item()
{
    const uint32_t total = gridDim.x, id = blockIdx.x;
    const uint32_t qd = total >> 3, rm = total & 7u, xcd = id & 7u, pos = id >> 3;
    return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + pos;
}

// v + (v of the lane a DPP control selects); lanes the control leaves without a source add 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add(int v)
{
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true);
}

// Inclusive prefix sum over the 64 lanes in seven DPP adds: three shifted copies of the input give
// sums of 4 inside each row of 16, row_shr:4 / :8 (restricted to the lanes that have a source) finish
// the rows, and the row totals are broadcast into the following rows.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_take(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, true);
}
__device__ __forceinline__ int wave_incl_scan(int v, int /*lane*/)
{
    int s = v + dpp_take<0x111, 0xf, 0xf>(v);       // row_shr:1
    s += dpp_take<0x112, 0xf, 0xf>(v);              // row_shr:2
    s += dpp_take<0x113, 0xf, 0xf>(v);              // row_shr:3
    s += dpp_take<0x114, 0xf, 0xe>(s);              // row_shr:4 into lanes 4 .. 15 of each row
    s += dpp_take<0x118, 0xf, 0xc>(s);              // row_shr:8 into lanes 8 .. 15
    s += dpp_take<0x142, 0xa, 0xf>(s);              // row_bcast:15 into rows 1 and 3
    s += dpp_take<0x143, 0xc, 0xf>(s);              // row_bcast:31 into rows 2 and 3
    return s;
}

// Measurement knob: extra dynamic LDS per workgroup (bytes, from an environment variable) for one kernel,
// which lowers how many of its workgroups fit a CU.  Never changes results.
static unsigned lds_pad_from_env(const char* name, const void* kernel)
{
    const char* e = getenv(name);
    const unsigned pad = e ? (unsigned)atoi(e) : 0u;
    if (pad) (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
    return pad;
}

// Bits b of mask word w whose pixel x = 32w + b satisfies x - 16 >= 0 && x + 16 < W.
__device__ __forceinline__ uint32_t desc_column_bits(int w, int W)
{
    int lo = 16 - 32 * w, hi = (W - 17) - 32 * w;
    if (hi < 0 || lo > 31) return 0u;
    lo = lo < 0 ? 0 : lo;
    hi = hi > 31 ? 31 : hi;
    uint32_t upto_hi = hi == 31 ? 0xFFFFFFFFu : ((1u << (hi + 1)) - 1u);
    return upto_hi & ~((1u << lo) - 1u);
}

// ---------------------------------------------------------------------------------------------
// BGR -> gray, OpenCV 4.x 8-bit fixed point (reference src/main_interface.cpp:83).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t gray_of(uint32_t b, uint32_t g, uint32_t r)
{
    return (b * 3735u + g * 19235u + r * 9798u + (1u << 14)) >> 15;
}

__global__ __launch_bounds__(256) void bgr2gray_kernel(const uint8_t* __restrict__ bgr,
                                                       uint8_t* __restrict__ gray, long npix,
                                                       int aligned)
{
    long quad = (long)blockIdx.x * 256 + threadIdx.x;  // 4 pixels = 12 input bytes
    long nquad = npix >> 2;
    long stride = (long)gridDim.x * 256;
    for (; quad < nquad; quad += stride) {
        uint32_t w0, w1, w2;
        if (aligned) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(bgr) + quad * 3;
            w0 = p[0]; w1 = p[1]; w2 = p[2];
        } else {
            const uint8_t* p = bgr + quad * 12;
            w0 = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
            w1 = p[4] | (p[5] << 8) | (p[6] << 16) | ((uint32_t)p[7] << 24);
            w2 = p[8] | (p[9] << 8) | (p[10] << 16) | ((uint32_t)p[11] << 24);
        }
        uint32_t g0 = gray_of(w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255);
        uint32_t g1 = gray_of(w0 >> 24, w1 & 255, (w1 >> 8) & 255);
        uint32_t g2 = gray_of((w1 >> 16) & 255, w1 >> 24, w2 & 255);
        uint32_t g3 = gray_of((w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24);
        uint32_t out = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
        if (aligned) reinterpret_cast<uint32_t*>(gray)[quad] = out;
        else { uint8_t* o = gray + quad * 4; o[0] = g0; o[1] = g1; o[2] = g2; o[3] = g3; }
    }
    // tail pixels (npix not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (npix & 3)) {
        long i = (npix & ~3L) + threadIdx.x;
        gray[i] = (uint8_t)gray_of(bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2]);
    }
}

void launch_bgr2gray(const uint8_t* bgr, uint8_t* gray, long npix, hipStream_t s)
{
    long nquad = npix >> 2;
    long blocks = (nquad + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    int aligned = (((uintptr_t)bgr | (uintptr_t)gray) & 3) == 0;
    hipLaunchKernelGGL(bgr2gray_kernel, dim3((unsigned)blocks), dim3(256), 0, s, bgr, gray, npix, aligned);
}

// ---------------------------------------------------------------------------------------------
// Ingest / egress for the single-frame, host-pointer calls.  Pinned host memory is mapped into the device's
// address space, so a kernel can read a staged input or write results there directly; compared with
// copy-engine transfers this removes a queue-to-queue synchronisation (10 - 20 us each on MI355X) per
// transfer from a call that is all latency.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16,
                                                     long nbytes)
{
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
    // tail bytes (nbytes not a multiple of 16)
    const long done = n16 * 16;
    if (blockIdx.x == 0 && done + threadIdx.x < nbytes)
        reinterpret_cast<uint8_t*>(dst)[done + threadIdx.x] = reinterpret_cast<const uint8_t*>(src)[done + threadIdx.x];
}

void launch_copy_bytes(const void* src, void* dst, size_t nbytes, hipStream_t s)
{
    const long n16 = (long)(nbytes >> 4);
    long blocks = (n16 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, n16,
                       (long)nbytes);
}

__global__ __launch_bounds__(256) void egress_kernel(EgressArgs a)
{
    const long stride = (long)gridDim.x * 256;
    for (int k = 0; k < a.nseg; ++k) {
        const EgressSeg sg = a.seg[k];
        long n = sg.max_dwords;
        if (sg.count) {
            long c = sg.count[0];
            c = c < 0 ? 0 : c * sg.count_mul;
            n = c < n ? c : n;
        }
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) sg.dst[i] = sg.src[i];
    }
}

void launch_egress(const EgressArgs& a, hipStream_t s)
{
    long total = 0;
    for (int k = 0; k < a.nseg; ++k) total += a.seg[k].max_dwords;
    long blocks = (total + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(egress_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------------
// Synthetic frames (definition: visual_slam_amd/synth.py).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t hash3(uint32_t seed, uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t h = mix32(seed + 0x9E3779B9u * a);
    h = mix32(h ^ (0x85EBCA6Bu * b));
    h = mix32(h + 0xC2B2AE35u * c);
    return h;
}
__device__ __forceinline__ uint32_t synth_pixel(uint32_t x, uint32_t y, uint32_t f, uint32_t seed,
                                                uint32_t cell, uint32_t thr)
{
    uint32_t wx = x + f, wy = y;
    uint32_t cx = wx / cell, lx = wx % cell, cy = wy / cell, ly = wy % cell;
    uint32_t phase = hash3(seed ^ 0x5EEDu, cx, cy, 7u) % 20u;
    uint32_t epoch = (f + phase) / 20u;
    uint32_t hc = hash3(seed ^ 0xB10Bu, cx, cy, epoch);
    bool has = (hc & 0xFFFFu) < thr;
    uint32_t span = cell - 9u;
    uint32_t bx = 4u + ((hc >> 16) & 0xFFu) % span;
    uint32_t by = 4u + ((hc >> 24) & 0xFFu) % span;
    uint32_t h2 = mix32(hc ^ 0xA5A5A5A5u);
    uint32_t size = 1u + (h2 & 1u);
    bool bright = ((h2 >> 1) & 1u) != 0;
    bool inblob = has && lx >= bx && lx < bx + size && ly >= by && ly < by + size;
    uint32_t bg = 96u + hash3(seed ^ 0xBAC6u, wx, wy, 0u) % 9u;
    uint32_t hf = hash3(seed ^ 0xF11Cu, x, y, f);
    if ((hf & 15u) == 0u) bg = bg < 104u ? bg + 1u : bg - 1u;
    return inblob ? (bright ? 230u : 10u) : bg;
}

// Textured scene (definition: synth_textured_frame in visual_slam_amd/synth.py).
__device__ __forceinline__ uint32_t synth_textured_pixel(uint32_t x, uint32_t y, uint32_t f, uint32_t seed)
{
    const uint32_t wx = x + f, wy = y;
    const uint32_t a = hash3(seed ^ 0x7E01u, wx >> 4, wy >> 4, 0u) & 255u;
    const uint32_t b = hash3(seed ^ 0x7E02u, wx / 7u, wy / 7u, 0u) % 81u;
    const uint32_t n = hash3(seed ^ 0x7E03u, wx, wy, 0u) % 9u;
    uint32_t v = 30u + (a >> 1) + b + n;
    const uint32_t hf = hash3(seed ^ 0xF11Cu, x, y, f);
    if ((hf & 15u) == 0u) v = ((hf >> 4) & 1u) ? v + 1u : v - 1u;
    return v;
}

// cell == 0 selects the textured scene.
__global__ __launch_bounds__(256) void synth_kernel(uint8_t* __restrict__ gray, int frames, int H,
                                                    int W, int first_frame, uint32_t seed,
                                                    uint32_t cell, uint32_t thr)
{
    const long per_frame = (long)H * W;
    const long total = per_frame * frames;
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 256 * 4;
    const bool al = ((uintptr_t)gray & 3) == 0;
    for (; i < total; i += stride) {
        uint32_t out = 0;
        int nb = total - i < 4 ? (int)(total - i) : 4;
        for (int b = 0; b < nb; ++b) {
            long j = i + b;
            uint32_t f = (uint32_t)(j / per_frame);
            long r = j - (long)f * per_frame;
            uint32_t y = (uint32_t)(r / W), x = (uint32_t)(r - (long)y * W);
            const uint32_t px = cell ? synth_pixel(x, y, f + (uint32_t)first_frame, seed, cell, thr)
                                     : synth_textured_pixel(x, y, f + (uint32_t)first_frame, seed);
            out |= px << (8 * b);
        }
        if (al && nb == 4) *reinterpret_cast<uint32_t*>(gray + i) = out;
        else for (int b = 0; b < nb; ++b) gray[i + b] = (uint8_t)(out >> (8 * b));
    }
}

void launch_synth(uint8_t* gray, int frames, int H, int W, int first_frame, uint32_t seed, int cell,
                  int density_thr, hipStream_t s)
{
    long total = (long)frames * H * W;
    long blocks = (total / 4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, s, gray, frames, H, W,
                       first_frame, seed, (uint32_t)cell, (uint32_t)density_thr);
}

// ---------------------------------------------------------------------------------------------
// FAST segment test (reference src/orb.cpp:24-108).
//
// A pixel with its radius-3 circle inside the image is a corner iff 12 consecutive circle pixels
// k = s .. s+11, s in 0..4 (the reference's run counter never wraps past k = 15), are all
// >= centre + 20 or all <= centre - 20.  Every such run contains k = 4 .. 11, hence k = 4 (E) and
// k = 8 (S), and it contains k = 0 (N) when s = 0 and k = 12 (W) when s >= 1.  So a corner needs
// E and S and (N or W) of one class.  That is the cheap screen used here; it follows from the run rule
// alone, so screening with it cannot lose a corner (it is stronger than the reference's own
// four-point pre-test, src/orb.cpp:65-68, which the run rule implies as well).
//
// Tile: 256 x 64 outputs per 256-thread workgroup, staged in LDS with a 3-row halo above and below
// and a 16-byte margin left and right (which holds the 3 halo columns).  A wave owns a strip of 16
// rows and walks it in two passes of 8; a lane owns one dword = 4 horizontally adjacent pixels.
//   Phase 1, per dword: v_sad_u8 gives sum |N_k - C_k| over the dword's 4 pixels; a pixel can only
//     pass the compass screen if the E sum and the S sum and one of the N / W sums are >= 20.
//     Two ors, one and, one compare decide for 4 pixels; passing dwords are pushed to list A.
//   Phase 2a, per listed dword (one per lane): the compass screen proper, per byte position with
//     SDWA byte selects; results are 64-bit lane masks combined on the scalar unit; passing
//     pixels are pushed to list B.  Skipped when list A is short: its pixels go straight to 2b.
//   Phase 2b, per listed pixel (one per lane): the full 16-pixel test; hits are OR-ed into the
//     tile's bit mask in LDS, which is written out coalesced at the end.
// Lists live in LDS per wave and are filled with mbcnt ranks, so every stage after phase 1 runs
// with all lanes busy however sparse the corners are.
// ---------------------------------------------------------------------------------------------
constexpr int FT_W = 256, FT_H = 64;
constexpr int FT_PITCH = FT_W / 4 + 8;     // LDS row pitch in dwords: 16 bytes of left margin, the 256-byte
                                           // body, 16 bytes of right margin; pixel x of a tile row sits at
                                           // byte 16 + (x - x0), the halo columns x0 - 3 .. x0 - 1 in bytes
                                           // 13 .. 15 and x0 + 256 .. x0 + 258 in bytes 272 .. 274
constexpr int FT_X0 = 4;                   // dword index of pixel x0 within an LDS row
constexpr int FT_LW = FT_PITCH * 4;        // LDS row pitch in bytes
constexpr int FT_LH = FT_H + 6;            // LDS rows
constexpr int FT_ROWS_PER_WAVE = FT_H / 4;
constexpr int FT_EVAL_ROWS = 8;           // rows of one evaluation pass (bounds list A: 64 entries a row)
static_assert(FT_ROWS_PER_WAVE % FT_EVAL_ROWS == 0, "a wave's rows are whole evaluation passes");

// Circle offsets in LDS bytes, k = 0 at (0,-3), clockwise (reference src/orb.cpp:43-48).
__device__ __forceinline__ constexpr int circ_off(int k)
{
    constexpr int dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    constexpr int dy[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
    return dy[k] * FT_LW + dx[k];
}

// Lane masks of "byte K of a >= b" (unsigned) and "byte K of a <= b" (signed; b may be negative).
#define ORB_CMP_GE_BYTE(K, dst, a, b) \
    asm("v_cmp_ge_u32_sdwa %0, %1, %2 src0_sel:BYTE_" #K " src1_sel:DWORD" : "=s"(dst) : "v"(a), "v"(b))
#define ORB_CMP_LE_BYTE(K, dst, a, b) \
    asm("v_cmp_le_i32_sdwa %0, %1, %2 src0_sel:BYTE_" #K " src1_sel:DWORD" : "=s"(dst) : "v"(a), "v"(b))

// Screen of byte position K of this lane's centre dword C against its compass neighbours.
#define ORB_SCREEN_BYTE(K, C, N, S, E, Wd, cand)                                   \
    {                                                                               \
        const uint32_t p_ = ((C) >> (8 * (K))) & 255u;                              \
        const uint32_t hi_ = p_ + 20u;                                              \
        const int lo_ = (int)p_ - 20;                                               \
        unsigned long long bn_, bs_, be_, bw_, dn_, ds_, de_, dw_;                  \
        ORB_CMP_GE_BYTE(K, bn_, N, hi_); ORB_CMP_GE_BYTE(K, bs_, S, hi_);           \
        ORB_CMP_GE_BYTE(K, be_, E, hi_); ORB_CMP_GE_BYTE(K, bw_, Wd, hi_);          \
        ORB_CMP_LE_BYTE(K, dn_, N, lo_); ORB_CMP_LE_BYTE(K, ds_, S, lo_);           \
        ORB_CMP_LE_BYTE(K, de_, E, lo_); ORB_CMP_LE_BYTE(K, dw_, Wd, lo_);          \
        cand = (be_ & bs_ & (bn_ | bw_)) | (de_ & ds_ & (dn_ | dw_));                \
    }

constexpr int FT_LIST_A = 576;   // entries of list A per wave (resolved before a pass could overflow it)
constexpr int FT_LIST_B = 320;   // entries of list B per wave (at most 63 + 4 * 64 are ever held)
constexpr int FT_SPARSE = 32;   // up to this many listed dwords skip the compass screen

// Push the lanes of `m` (a wave-uniform lane mask) onto a per-wave LDS list.
__device__ __forceinline__ void list_push(uint16_t* list, int& count, unsigned long long m, int lane,
                                          uint32_t value)
{
    if (m != 0ull) {   // wave-uniform
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                              __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if ((m >> lane) & 1ull) list[count + rank] = (uint16_t)value;
        count += __popcll(m);
    }
}

// Phase 2b: exact segment test for list entries [from, from + n), n <= 64, one per lane.
// Entry = (row within the wave's strip) << 8 | (x within the tile).
// Every comparison "circle pixel k is brighter / darker" is made for all 64 lanes at once into a
// 64-bit lane mask (one v_cmp each), and the run test is evaluated on those masks by the scalar
// unit: a run k = s .. s+11 (s = 0 .. 4) is the core k = 4 .. 11 plus a suffix of k = 0 .. 3 and a
// prefix of k = 12 .. 15.  Lanes without an entry evaluate entry 0 (in bounds) and are masked out.
__device__ __forceinline__ unsigned long long run12_lanes(const unsigned long long (&m)[16])
{
    const unsigned long long core = m[4] & m[5] & m[6] & m[7] & m[8] & m[9] & m[10] & m[11];
    if (core == 0ull) return 0ull;
    const unsigned long long l3 = m[3], l2 = l3 & m[2], l1 = l2 & m[1], l0 = l1 & m[0];
    const unsigned long long r12 = m[12], r13 = r12 & m[13], r14 = r13 & m[14], r15 = r14 & m[15];
    return core & (l0 | (l1 & r12) | (l2 & r13) | (l3 & r14) | r15);
}

__device__ __forceinline__ void fast_exact_batch(const uint8_t* tile, uint32_t* mtile, const uint16_t* list,
                                                 int from, int n, int lane, int r0, int x0, int y0, int H, int W)
{
    const bool act = lane < n;
    const uint32_t e = act ? (uint32_t)list[from + lane] : 0u;
    const int rr = (int)(e >> 8), xl = (int)(e & 255u);
    const int x = x0 + xl, y = y0 + r0 + rr;
    const bool ok = act && x >= 3 && x < W - 3 && y >= 3 && y < H - 3;
    const uint8_t* c = tile + (r0 + rr + 3) * FT_LW + xl + 4 * FT_X0;
    const uint32_t p = c[0];
    const uint32_t hi = p + 20u;
    const int lo = (int)p - 20;
    unsigned long long mb[16], md[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t v = c[circ_off(k)];
        mb[k] = __ballot(v >= hi);
        md[k] = __ballot((int)v <= lo);
    }
    const unsigned long long hit = (run12_lanes(mb) | run12_lanes(md)) & __ballot(ok);
    if ((hit >> lane) & 1ull) atomicOr(&mtile[(r0 + rr) * (FT_W / 32) + (xl >> 5)], 1u << (xl & 31));
}

// Staging of one tile (aligned path): lane l reads dword l of each of this wave's LDS rows (image
// x = x0 - 4 + 4l; wave w owns tile rows w, w + 4, ...); dwords 64 and 65 of all those rows come from
// one extra load (lane = 2 * row step + side).  Loads are unconditional from clamped coordinates
// and the value is zeroed afterwards where the true position is outside the image: no branches, and
// every load of the wave is in flight at once -- and stays in flight across the evaluation of the
// previous tile, because loading (to registers) and storing (to LDS) are separate steps.
constexpr int FT_STEPS = (FT_LH + 3) / 4;

struct FastStage {
    uint32_t va[FT_STEPS];
    uint32_t vb;
};

__device__ __forceinline__ void fast_stage_load(FastStage& st, const uint8_t* __restrict__ img, int H, int W,
                                                int x0, int y0, int lane, int wave)
{
    const int xa = x0 - 4 + 4 * lane;
    const int xac = xa < 0 ? 0 : (xa < W ? xa : W - 4);
    const int hy = y0 - 3 + wave + 4 * (lane >> 1);
    const int hyc = hy < 0 ? 0 : (hy < H ? hy : H - 1);
    const int xb = x0 + 252 + 4 * (lane & 1);
    const int xbc = xb < W ? xb : W - 4;
#pragma unroll
    for (int it = 0; it < FT_STEPS; ++it) {
        const int y = y0 - 3 + wave + 4 * it;                      // wave-uniform
        const int yc = y < 0 ? 0 : (y < H ? y : H - 1);
        st.va[it] = *reinterpret_cast<const uint32_t*>(img + (uint32_t)(yc * W) + (uint32_t)xac);
    }
    st.vb = *reinterpret_cast<const uint32_t*>(img + (uint32_t)(hyc * W) + (uint32_t)xbc);
}

__device__ __forceinline__ void fast_stage_store(const FastStage& st, uint32_t* tile32, int H, int W, int x0,
                                                 int y0, int lane, int wave)
{
    const int xa = x0 - 4 + 4 * lane;
    const bool oka = xa >= 0 && xa < W;
    const int hr = wave + 4 * (lane >> 1), hy = y0 - 3 + hr;
    const int xb = x0 + 252 + 4 * (lane & 1);
    const bool okb = hy >= 0 && hy < H && xb < W;
#pragma unroll
    for (int it = 0; it < FT_STEPS; ++it) {
        const int r = wave + 4 * it, y = y0 - 3 + r;
        const bool rowok = y >= 0 && y < H;                        // wave-uniform
        if (r < FT_LH) tile32[r * FT_PITCH + FT_X0 - 1 + lane] = (rowok && oka) ? st.va[it] : 0u;
    }
    if (lane < 2 * FT_STEPS && hr < FT_LH) tile32[hr * FT_PITCH + FT_X0 + 63 + (lane & 1)] = okb ? st.vb : 0u;
}

// Staging when rows are whole aligned 16-byte groups (W % 16 == 0, 16-byte-aligned frame).  A tile
// row is 18 segments of 16 bytes: the 16 of the body plus one on each side, which hold the three
// halo columns (image x = x0 - 16 + 16 * segment).  The segments of all rows are numbered row-major
// (idx = 18 * row + segment) and thread t moves segments t, t + 256, ...; LDS rows are exactly 18
// segments long, so segment idx lands at byte 16 * idx.  The halo columns ride in the same load
// instructions as the body: fetching them with loads of their own (one dword per lane and row, 64
// different cache lines per instruction) cost a third of the kernel's run time on MI355X.
// The loads themselves are in fast_mask_kernel: bounds-checked buffer loads that write LDS directly.
constexpr int FT_SEGS = FT_LH * (FT_PITCH / 4);          // 16-byte segments of a tile
constexpr int FT_SPT = (FT_SEGS + 255) / 256;            // segments per thread
static_assert(FT_PITCH % 4 == 0 && FT_PITCH / 4 == FT_W / 16 + 2, "a tile row is the body plus one segment per side");

// Staging for images whose rows are not whole aligned dwords: bytes assembled one by one.
__device__ __forceinline__ void fast_stage_bytes(uint32_t* tile32, const uint8_t* __restrict__ img, int H,
                                                 int W, int x0, int y0, int lane, int wave)
{
    for (int r = wave; r < FT_LH; r += 4) {
        const int y = y0 - 3 + r;
        const bool rowok = y >= 0 && y < H;
        const uint8_t* row = img + (size_t)(rowok ? y : 0) * W;
        for (int j = lane; j < FT_W / 4 + 2; j += 64) {
            const int x = x0 - 4 + 4 * j;
            uint32_t v = 0;
            if (rowok) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int xx = x + b;
                    if (xx >= 0 && xx < W) v |= (uint32_t)row[xx] << (8 * b);
                }
            }
            tile32[r * FT_PITCH + FT_X0 - 1 + j] = v;
        }
    }
}

// Phase 1 for FT_EVAL_ROWS output rows of the staged tile, starting at tile row r0 = strip + rr0
// (strip = first row of this wave's strip): SAD screen, one dword (4 pixels) per lane; passing dwords
// are appended to list A as (row within the strip) << 6 | lane.
__device__ __forceinline__ void fast_screen_rows(const uint32_t* tile32, uint16_t* listA, int& countA, int lane,
                                                 int r0, int rr0)
{
    {
        // rw[i] = dword of tile row r0 + i at this lane's column (output row rr uses rw[rr .. rr + 6])
        uint32_t rw[FT_EVAL_ROWS + 6];
#pragma unroll
        for (int i = 0; i < FT_EVAL_ROWS + 6; ++i) rw[i] = tile32[(r0 + i) * FT_PITCH + FT_X0 + lane];
        // All LDS reads first, then straight-line arithmetic for the 8 rows (the lane masks stay in
        // scalar registers), then the pushes: no LDS latency and no branch between rows.  (Reading
        // the dwords three pixels to the right and left at their unaligned LDS addresses instead of
        // assembling them with v_alignbyte made the kernel three times slower on MI355X.)
        uint32_t e[FT_EVAL_ROWS], w[FT_EVAL_ROWS];
#pragma unroll
        for (int rr = 0; rr < FT_EVAL_ROWS; ++rr) {
            const uint32_t cp = tile32[(r0 + rr + 3) * FT_PITCH + FT_X0 - 1 + lane];
            const uint32_t cn = tile32[(r0 + rr + 3) * FT_PITCH + FT_X0 + 1 + lane];
            e[rr] = __builtin_amdgcn_alignbyte(cn, rw[rr + 3], 3);    // pixels x+3 .. x+6
            w[rr] = __builtin_amdgcn_alignbyte(rw[rr + 3], cp, 1);    // pixels x-3 .. x
        }
        // sv[i] = SAD of tile rows r0 + i and r0 + i + 3, minus 20 (the accumulator input of v_sad_u8
        // starts at -20): the north term of output row i and the south term of output row i - 3.
        // A term is negative iff its SAD is below 20, so "E and S and (N or W) pass" is the sign of
        // e | s | (n & w) being clear: or / and issue at twice the rate of min / max.
        int sv[FT_EVAL_ROWS + 3];
#pragma unroll
        for (int i = 0; i < FT_EVAL_ROWS + 3; ++i) sv[i] = (int)__builtin_amdgcn_sad_u8(rw[i], rw[i + 3], (uint32_t)-20);
        unsigned long long m[FT_EVAL_ROWS];
#pragma unroll
        for (int rr = 0; rr < FT_EVAL_ROWS; ++rr) {
            const uint32_t C = rw[rr + 3];
            const int se = (int)__builtin_amdgcn_sad_u8(e[rr], C, (uint32_t)-20);
            const int sw = (int)__builtin_amdgcn_sad_u8(w[rr], C, (uint32_t)-20);
            m[rr] = __ballot((se | sv[rr + 3] | (sv[rr] & sw)) >= 0);
        }
#pragma unroll
        for (int rr = 0; rr < FT_EVAL_ROWS; ++rr)
            list_push(listA, countA, m[rr], lane, (uint32_t)(((rr0 + rr) << 6) | lane));
    }
}

// Phases 2a, 2b for the dwords on list A; r0 = first tile row of this wave's strip.
__device__ __forceinline__ void fast_resolve(const uint32_t* tile32, uint32_t* mtile, const uint16_t* listA,
                                             int countA, uint16_t* listB, int lane, int r0, int x0, int y0, int H, int W)
{
    const uint8_t* tile = reinterpret_cast<const uint8_t*>(tile32);
    int countB = 0;
    // Few listed dwords: skip the compass screen and give every pixel of the listed dwords straight
    // to the exact test, 16 dwords = 64 pixels per batch (a batch costs less than screening it).
    if (countA <= FT_SPARSE) {
        for (int d0 = 0; d0 < countA; d0 += 16) {
            const int n = 4 * ((countA - d0) < 16 ? (countA - d0) : 16);
            if (lane < n) {
                const uint32_t e = listA[d0 + (lane >> 2)];
                listB[lane] = (uint16_t)(((e >> 6) << 8) | (4 * (e & 63u) + (uint32_t)(lane & 3)));
            }
            fast_exact_batch(tile, mtile, listB, 0, n, lane, r0, x0, y0, H, W);
        }
        return;
    }

    // ---- phase 2a: compass screen on the listed dwords, one per lane; 2b whenever B holds 64 ----
    for (int i0 = 0; i0 < countA; i0 += 64) {
        const int i = i0 + lane;
        const bool act = i < countA;
        const uint32_t e = act ? listA[i] : 0u;
        const int rr = (int)(e >> 6), dw = (int)(e & 63u);
        const uint32_t* rc = tile32 + (r0 + rr + 3) * FT_PITCH + FT_X0 - 1 + dw;   // -> the dword left of C
        const uint32_t Cp = rc[0], C = rc[1], Cn = rc[2];
        const uint32_t N = rc[1 - 3 * FT_PITCH], S = rc[1 + 3 * FT_PITCH];
        const uint32_t E = __builtin_amdgcn_alignbyte(Cn, C, 3);
        const uint32_t Wd = __builtin_amdgcn_alignbyte(C, Cp, 1);
        const unsigned long long live = __ballot(act);
        unsigned long long cand[4];
        ORB_SCREEN_BYTE(0, C, N, S, E, Wd, cand[0]);
        ORB_SCREEN_BYTE(1, C, N, S, E, Wd, cand[1]);
        ORB_SCREEN_BYTE(2, C, N, S, E, Wd, cand[2]);
        ORB_SCREEN_BYTE(3, C, N, S, E, Wd, cand[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            list_push(listB, countB, cand[k] & live, lane, (uint32_t)((rr << 8) | (4 * dw + k)));
        while (countB >= 64) {
            countB -= 64;
            fast_exact_batch(tile, mtile, listB, countB, 64, lane, r0, x0, y0, H, W);
        }
    }
    if (countB > 0) fast_exact_batch(tile, mtile, listB, 0, countB, lane, r0, x0, y0, H, W);
}

// This wave's FT_ROWS_PER_WAVE rows of the tile, one evaluation pass after the other.
__device__ __forceinline__ void fast_eval_tile(const uint32_t* tile32, uint32_t* mtile, uint16_t* listA,
                                               uint16_t* listB, int lane, int wave, int x0, int y0, int H, int W)
{
    const int strip = wave * FT_ROWS_PER_WAVE;
    int countA = 0;
    for (int pass = 0; pass < FT_ROWS_PER_WAVE / FT_EVAL_ROWS; ++pass) {
        const int rr0 = pass * FT_EVAL_ROWS;
        if (y0 + strip + rr0 >= H) break;                     // rows below the image (wave-uniform)
        fast_screen_rows(tile32, listA, countA, lane, strip + rr0, rr0);
        // list A is resolved once per strip, earlier only if the next pass could overflow it
        if (countA > FT_LIST_A - 64 * FT_EVAL_ROWS) {
            fast_resolve(tile32, mtile, listA, countA, listB, lane, strip, x0, y0, H, W);
            countA = 0;
        }
    }
    if (countA > 0) fast_resolve(tile32, mtile, listA, countA, listB, lane, strip, x0, y0, H, W);
}

// One workgroup per tile.  The kernel is bound by vector-ALU issue, not by memory: loading and
// staging alone run at 5.6 TB/s on MI355X (a bare read of the same tiles reaches 6.7 TB/s), and the
// evaluation phases add their vector instructions on top.  Five workgroups are resident per CU;
// variants that spent more LDS (longer lists, taller tiles) or more registers (a persistent
// workgroup prefetching its next tile) lost more to the lower residency than they saved.
__global__ __launch_bounds__(256) void fast_mask_kernel(const uint8_t* __restrict__ gray, int H,
                                                        int W, uint32_t* __restrict__ mask, int MW,
                                                        int aligned, int ntc, int ntr,
                                                        unsigned long long* __restrict__ rowcnt)
{
    // (whole load steps: the last one writes zeros into the 20 segments behind the tile's 1260)
    __shared__ __attribute__((aligned(16))) uint32_t tile32[256 * FT_SPT * 4];                 // 20480 B
    __shared__ uint32_t mtile[FT_H * (FT_W / 32)];                // 2 KiB: the tile's corner bits
    __shared__ uint16_t lists[4][FT_LIST_A + FT_LIST_B];           // per wave: list A, then list B
    // 1-D grid, tile column fastest.  Neighbouring tiles read each other's edge cache lines (the
    // halo), so a tile's neighbours should run on the SAME XCD and find those lines in its L2: with
    // the plain mapping and 8 tile columns every column would be pinned to its own XCD and each tile
    // row would pull two extra lines through the fabric.
    const uint32_t item = xcd_contiguous_item();
    const int bx = (int)(item % (uint32_t)ntc);
    const uint32_t rest = item / (uint32_t)ntc;
    const int by = (int)(rest % (uint32_t)ntr);
    const int f = (int)(rest / (uint32_t)ntr);
    const uint8_t* img = gray + (size_t)f * H * W;
    uint32_t* mfr = mask + (size_t)f * H * MW;
    const int x0 = bx * FT_W, y0 = by * FT_H;
    const int lane = lane_id(), wave = wave_id();

    if (aligned == 16) {
        // Segments straight into LDS (no registers, no LDS store instructions): segment idx of the tile lands at
        // byte 16 * idx, which is where a wave's lanes write when the LDS base advances by 64 segments per wave
        // and 256 per step.  Rows above / below the frame fall outside the buffer and read as zero; segments left
        // or right of it (W is a multiple of 16: inside or outside as a whole) get an out-of-range offset.  No
        // pixel outside the frame is ever looked at by a pixel that can be a corner.
        constexpr int SPR = FT_PITCH / 4;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, H * W, 0x00020000);
        const int t = (int)threadIdx.x;
        const int row0 = t / SPR, seg0 = t - SPR * row0;
        const int xs0 = x0 - 16 + 16 * seg0;
        const int off0 = __mul24(y0 - 3 + row0, W) + xs0;
#pragma unroll
        for (int i = 0; i < FT_SPT; ++i) {
            const int drow = (256 * i) / SPR, dseg = (256 * i) % SPR;
            const bool wrap = seg0 + dseg >= SPR;
            const int x = xs0 + 16 * dseg - (wrap ? 16 * SPR : 0);
            const int off = off0 + drow * W + 16 * dseg + (wrap ? W - 16 * SPR : 0);
            const bool ok = (uint32_t)x < (uint32_t)W && t + 256 * i < FT_SEGS;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (__attribute__((address_space(3))) void*)(tile32 + 4 * (256 * i + 64 * wave)), 16,
                ok ? (uint32_t)off : 0x80000000u, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (aligned == 4) {
        FastStage st;
        fast_stage_load(st, img, H, W, x0, y0, lane, wave);
        fast_stage_store(st, tile32, H, W, x0, y0, lane, wave);
    } else {
        fast_stage_bytes(tile32, img, H, W, x0, y0, lane, wave);
    }
#pragma unroll
    for (int t = 0; t < FT_H * (FT_W / 32); t += 256) mtile[t + threadIdx.x] = 0u;
    __syncthreads();

    fast_eval_tile(tile32, mtile, lists[wave], lists[wave] + FT_LIST_A, lane, wave, x0, y0, H, W);
    __syncthreads();

    // write the tile's mask words (FT_H rows x 8 words) and add this tile's share to the row totals:
    // (corners, corners that get a descriptor = >= 16 px from every edge, reference src/orb.cpp:156)
    // as the two halves of one 64-bit add.  rowcnt is zeroed before the launch.
    const int wq = threadIdx.x & 7, wi = (x0 >> 5) + wq;
    const uint32_t dbits = desc_column_bits(wi, W);
    unsigned long long* const rc = rowcnt + (size_t)f * H;
#pragma unroll
    for (int t0 = 0; t0 < FT_H * (FT_W / 32); t0 += 256) {
        const int t = t0 + (int)threadIdx.x;
        const int y = y0 + (t >> 3);
        int packed = 0;
        if (y < H && wi < MW) {
            const uint32_t m = mtile[t];
            mfr[(uint32_t)(y * MW + wi)] = m;
            const bool yv = (y - 16 >= 0) && (y + 16 < H);
            packed = __popc(m) | ((yv ? __popc(m & dbits) : 0) << 16);
        }
        // sum over the row's 8 words = 8 consecutive lanes (at most 256 per half)
        packed = dpp_add<0xB1, 0xf>(packed);      // quad_perm [1,0,3,2]
        packed = dpp_add<0x4E, 0xf>(packed);      // quad_perm [2,3,0,1]
        packed = dpp_add<0x141, 0xf>(packed);     // row_half_mirror: the other quad of the 8
        if (wq == 0 && y < H && packed != 0)
            atomicAdd(rc + (uint32_t)y, (unsigned long long)((uint32_t)packed & 0xFFFFu) |
                                            ((unsigned long long)((uint32_t)packed >> 16) << 32));
    }
}

void launch_fast_mask(const FrameSet& fs, uint32_t* mask, int2* rowcnt, hipStream_t s)
{
    const int MW = mask_words(fs.W);
    const int ntc = (fs.W + FT_W - 1) / FT_W, ntr = (fs.H + FT_H - 1) / FT_H;
    dim3 grid((unsigned)((size_t)ntc * ntr * fs.frames));
    int aligned = 1;
    if ((fs.W & 15) == 0 && ((uintptr_t)fs.gray & 15) == 0) aligned = 16;
    else if ((fs.W & 3) == 0 && ((uintptr_t)fs.gray & 3) == 0) aligned = 4;
    static const unsigned pad = lds_pad_from_env("ORBAMD_PAD_LDS_FAST", (const void*)fast_mask_kernel);
    hipLaunchKernelGGL(fast_mask_kernel, grid, dim3(256), pad, s, fs.gray, fs.H, fs.W, mask, MW, aligned, ntc, ntr,
                       reinterpret_cast<unsigned long long*>(rowcnt));
}

// ---------------------------------------------------------------------------------------------
// Ordered compaction of the corner mask.  Output order is the reference's scan order (ascending
// y, then x; reference src/orb.cpp:51-52) by construction: rows are prefix-summed, and inside a
// row each lane owns 32 consecutive pixels.  A keypoint "has a descriptor" iff it is >= 16 px from
// every edge (reference src/orb.cpp:156); its descriptor row is its rank among those.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowscan_kernel(const int2* __restrict__ rowcnt, int H,
                                                      int2* __restrict__ rowoff,
                                                      int32_t* __restrict__ counts)
{
    __shared__ int2 wsum[4];
    const int f = blockIdx.x;
    const int2* rc = rowcnt + (size_t)f * H;
    int2* ro = rowoff + (size_t)f * H;
    const int per = (H + 255) / 256;
    const int r0 = threadIdx.x * per;
    int a = 0, b = 0;
    for (int r = r0; r < r0 + per && r < H; ++r) { a += rc[r].x; b += rc[r].y; }
    const int lane = lane_id(), wave = wave_id();
    int ia = wave_incl_scan(a, lane), ib = wave_incl_scan(b, lane);
    if (lane == 63) wsum[wave] = make_int2(ia, ib);
    __syncthreads();
    int basea = 0, baseb = 0;
    for (int w = 0; w < wave; ++w) { basea += wsum[w].x; baseb += wsum[w].y; }
    int ea = basea + ia - a, eb = baseb + ib - b;  // exclusive prefix of this thread's chunk
    for (int r = r0; r < r0 + per && r < H; ++r) {
        ro[r] = make_int2(ea, eb);
        ea += rc[r].x; eb += rc[r].y;
    }
    if (threadIdx.x == 255) {
        counts[2 * f] = basea + ia;
        counts[2 * f + 1] = baseb + ib;
    }
}

void launch_rowscan(const FrameSet& fs, const int2* rowcnt, int2* rowoff, int32_t* counts, hipStream_t s)
{
    hipLaunchKernelGGL(rowscan_kernel, dim3(fs.frames), dim3(256), 0, s, rowcnt, fs.H, rowoff, counts);
}

constexpr int EMIT_AHEAD = 4;       // rows whose mask words are fetched together
// EMIT_ROWS = rows per wave (one row's count and offset per lane): 64 for batches; small launches (a single
// frame) use 8 so that the rows spread over enough waves.
template <int EMIT_ROWS>
__global__ __launch_bounds__(256) void emit_kernel(const uint32_t* __restrict__ mask, int H, int W,
                                                   int MW, long nrows, const int2* __restrict__ rowcnt,
                                                   const int2* __restrict__ rowoff, int cap,
                                                   int2* __restrict__ kpxy, int32_t* __restrict__ kpslot,
                                                   int32_t* __restrict__ counts)
{
    // A wave owns 64 consecutive rows (of the whole batch: a band may straddle two frames).  Lane l
    // fetches the count and the offset of row l in one coalesced load each; only rows that hold a
    // corner are walked (many hold none), and the mask words of the next four such rows are fetched
    // together, so that four rows' load latencies overlap instead of adding up.
    const long r0 = ((long)blockIdx.x * 4 + wave_id()) * EMIT_ROWS;
    if (r0 >= nrows) return;
    const int lane = lane_id();
    const bool have = lane < EMIT_ROWS && r0 + lane < nrows;
    const int2 rc_l = have ? rowcnt[r0 + lane] : make_int2(0, 0);
    const int2 off_l = have ? rowoff[r0 + lane] : make_int2(0, 0);
    // frame and image row of every lane's row: one division for the 64 rows (nrows < 2^31)
    const uint32_t f_l = (uint32_t)(r0 + lane) / (uint32_t)H;
    const int y_l = (int)((uint32_t)(r0 + lane) - f_l * (uint32_t)H);
    const uint32_t colbits0 = desc_column_bits(lane, W);    // of this lane's word in the first 64 words
    unsigned long long todo = __ballot(rc_l.x != 0);
    while (todo) {
        int jr[EMIT_AHEAD];
        uint32_t first[EMIT_AHEAD];
#pragma unroll
        for (int u = 0; u < EMIT_AHEAD; ++u) {
            jr[u] = -1;
            first[u] = 0u;
            if (todo) {
                jr[u] = (int)__builtin_ctzll(todo);
                todo &= todo - 1;
                if (lane < MW) first[u] = mask[(size_t)(r0 + jr[u]) * MW + lane];
            }
        }
#pragma unroll
        for (int u = 0; u < EMIT_AHEAD; ++u) {
            if (jr[u] < 0) break;
            const long row = r0 + jr[u];
            const size_t f = (size_t)__builtin_amdgcn_readlane((int)f_l, jr[u]);
            const int y = __builtin_amdgcn_readlane(y_l, jr[u]);
            const bool yv = (y - 16 >= 0) && (y + 16 < H);
            const uint32_t* mrow = mask + (size_t)row * MW;
            // (frame base pointers are scalars; the keypoint index becomes a 32-bit byte offset)
            uint8_t* const xyb = reinterpret_cast<uint8_t*>(kpxy + f * cap);
            uint8_t* const slotb = reinterpret_cast<uint8_t*>(kpslot + f * cap);
            int base = __builtin_amdgcn_readlane(off_l.x, jr[u]);
            int basev = __builtin_amdgcn_readlane(off_l.y, jr[u]);
            for (int w0 = 0; w0 < MW; w0 += 64) {
                const int w = w0 + lane;
                uint32_t bits = w0 == 0 ? first[u] : (w < MW ? mrow[w] : 0u);
                const uint32_t vb = yv ? (bits & (w0 == 0 ? colbits0 : desc_column_bits(w, W))) : 0u;
                const int c = __popc(bits), cv = __popc(vb);
                // both counts in one scan (a row holds at most 32 * 64 corners: 16 bits each)
                const int icc = wave_incl_scan(c | (cv << 16), lane);
                const int ic = icc & 0xFFFF, icv = icc >> 16;
                int i = base + ic - c, iv = basev + icv - cv;
                while (bits) {
                    const int b = __ffs(bits) - 1;
                    bits &= bits - 1u;
                    const bool v = (vb >> b) & 1u;
                    if (i < cap) {
                        *reinterpret_cast<int2*>(xyb + (uint32_t)i * 8u) = make_int2(32 * w + b, y);
                        *reinterpret_cast<int32_t*>(slotb + (uint32_t)i * 4u) = v ? iv : -1;
                    } else if (i == cap) {
                        // The first keypoint that does not fit: iv descriptor rows belong to the cap
                        // keypoints that are stored.  The frame's second count becomes that number, so
                        // that the matchers never read a descriptor row nobody wrote.
                        counts[2 * f + 1] = iv;
                    }
                    ++i;
                    iv += v;
                }
                base += __builtin_amdgcn_readlane(ic, 63);
                basev += __builtin_amdgcn_readlane(icv, 63);
            }
        }
    }
}

void launch_emit(const FrameSet& fs, const uint32_t* mask, const int2* rowcnt, const int2* rowoff,
                 int cap, int2* kpxy, int32_t* kpslot, int32_t* counts, hipStream_t s)
{
    long nrows = (long)fs.frames * fs.H;
    if (nrows >= 16384)
        hipLaunchKernelGGL(emit_kernel<64>, dim3((unsigned)((nrows + 4 * 64 - 1) / (4 * 64))), dim3(256), 0, s, mask, fs.H,
                           fs.W, mask_words(fs.W), nrows, rowcnt, rowoff, cap, kpxy, kpslot, counts);
    else
        hipLaunchKernelGGL(emit_kernel<8>, dim3((unsigned)((nrows + 4 * 8 - 1) / (4 * 8))), dim3(256), 0, s, mask, fs.H,
                           fs.W, mask_words(fs.W), nrows, rowcnt, rowoff, cap, kpxy, kpslot, counts);
}

// ---------------------------------------------------------------------------------------------
// Detector extension (opt-in, NOT the reference's detector and not on the parity path): arc length
// 9 .. 16, runs that may wrap from k = 15 to k = 0, a threshold other than 20, and a corner score with
// 3x3 non-maximum suppression.  The reference hard-codes 12 / no wrap / 20 and has no score
// (src/orb.cpp:39, :70-96; the OpenCV call with suppression is commented out at :104-105).
//
// score(p) = largest t for which p passes the segment test with threshold t (inclusive compares) =
// max over the allowed runs of the smallest one-signed |difference| in the run; kept as a byte, 0
// where it is below the threshold.
// ---------------------------------------------------------------------------------------------
constexpr int FX_W = 64, FX_H = 16;                  // outputs per workgroup: one mask row of a tile = one wave's ballot
constexpr int FX_PW = FX_W + 8, FX_PH = FX_H + 8;    // pixel tile: columns x0 - 4 .. x0 + 67 (18 dwords), rows y0 - 4 .. y0 + 19
constexpr int FX_SW = FX_W + 2, FX_SH = FX_H + 2;    // score region: one pixel more on every side, for the suppression
constexpr int FX_SP = 68;                            // score row pitch in bytes

// Circle offsets in the pixel tile (pitch FX_PW), k = 0 at (0,-3), clockwise.
__device__ __forceinline__ constexpr int fx_off(int k)
{
    constexpr int dxk[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    constexpr int dyk[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
    return dyk[k] * FX_PW + dxk[k];
}

// Cheap screen: any run of >= 9 circle pixels holds two neighbouring compass points (k = 0, 4, 8, 12) of
// its sign, so a pixel without such a pair scores 0.
__device__ __forceinline__ bool fastx_screen(const uint8_t* c, int thr)
{
    const int p = c[0];
    const int d0 = (int)c[fx_off(0)] - p, d4 = (int)c[fx_off(4)] - p, d8 = (int)c[fx_off(8)] - p, d12 = (int)c[fx_off(12)] - p;
    const bool b0 = d0 >= thr, b4 = d4 >= thr, b8 = d8 >= thr, b12 = d12 >= thr;
    const bool k0 = d0 <= -thr, k4 = d4 <= -thr, k8 = d8 <= -thr, k12 = d12 <= -thr;
    return (b0 && b4) || (b4 && b8) || (b8 && b12) || (b12 && b0) || (k0 && k4) || (k4 && k8) || (k8 && k12) || (k12 && k0);
}

// max over the runs k = s .. s + ARC - 1 (s < nstarts, indices mod 16) of the smallest e[k] in the run: window
// minima of length 2, 4, 8 by doubling, and a run of 9 .. 16 as two overlapping windows of 8.
template <int ARC>
__device__ __forceinline__ int fastx_best_run(const int (&e)[16], int nstarts)
{
    int a[16], b[16], c[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = min(e[k], e[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) b[k] = min(a[k], a[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k] = min(b[k], b[(k + 4) & 15]);
    int best = -256;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int w = min(c[k], c[(k + ARC - 8) & 15]);
        if (k < nstarts) best = max(best, w);
    }
    return best;
}

// Score of the pixel at LDS position c (pixel tile, pitch FX_PW): see above.
template <int ARC>
__device__ __forceinline__ int fastx_score(const uint8_t* c, int thr, int nstarts)
{
    const int p = c[0];
    int d[16], nd[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { d[k] = (int)c[fx_off(k)] - p; nd[k] = -d[k]; }
    const int best = max(fastx_best_run<ARC>(d, nstarts), fastx_best_run<ARC>(nd, nstarts));
    return best < thr ? 0 : (best > 255 ? 255 : best);
}

__device__ __forceinline__ int fastx_score_arc(const uint8_t* c, int thr, int arc, int nstarts)
{
    switch (arc) {                       // (uniform over the launch)
        case 9: return fastx_score<9>(c, thr, nstarts);
        case 10: return fastx_score<10>(c, thr, nstarts);
        case 11: return fastx_score<11>(c, thr, nstarts);
        case 12: return fastx_score<12>(c, thr, nstarts);
        case 13: return fastx_score<13>(c, thr, nstarts);
        case 14: return fastx_score<14>(c, thr, nstarts);
        case 15: return fastx_score<15>(c, thr, nstarts);
        default: return fastx_score<16>(c, thr, nstarts);
    }
}

// One workgroup per 64 x 16 tile: pixels (with a 4-pixel halo) staged in LDS by dword loads, scores of the
// tile and of the ring of pixels around it computed into LDS, then one wave per mask row: the ballot of
// "kept" over the 64 lanes IS the row's two mask words.  The score map never exists in device memory.  Also
// adds the tile's (corners, corners with a descriptor) to the row totals (rowcnt zeroed before the launch).
__global__ __launch_bounds__(256) void fastx_tile_kernel(const uint8_t* __restrict__ gray, int H, int W, int thr,
                                                         int arc, int wrap, int nms, uint32_t* __restrict__ mask,
                                                         int MW, unsigned long long* __restrict__ rowcnt, int ntc,
                                                         int ntr, int dwords)
{
    __shared__ uint32_t pix32[FX_PH * FX_PW / 4];
    __shared__ uint8_t sc[FX_SH * FX_SP];
    __shared__ uint16_t cand[FX_SW * FX_SH];
    __shared__ int ncand;
    if (threadIdx.x == 0) ncand = 0;
    const uint32_t item = xcd_contiguous_item();
    const int bx = (int)(item % (uint32_t)ntc);
    const uint32_t rest = item / (uint32_t)ntc;
    const int by = (int)(rest % (uint32_t)ntr);
    const size_t f = rest / (uint32_t)ntr;
    const uint8_t* img = gray + f * H * W;
    const int x0 = bx * FX_W, y0 = by * FX_H;
    const int t = (int)threadIdx.x;
    for (int i = t; i < FX_PH * (FX_PW / 4); i += 256) {
        const int row = i / (FX_PW / 4), dw = i - row * (FX_PW / 4);
        const int y = y0 - 4 + row, x = x0 - 4 + 4 * dw;
        uint32_t v = 0u;
        if (y >= 0 && y < H) {
            if (dwords) {
                if (x >= 0 && x < W) v = *reinterpret_cast<const uint32_t*>(img + (size_t)y * W + x);
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (x + b >= 0 && x + b < W) v |= (uint32_t)img[(size_t)y * W + x + b] << (8 * b);
            }
        }
        pix32[i] = v;
    }
    __syncthreads();
    const uint8_t* pix = reinterpret_cast<const uint8_t*>(pix32);
    const int nstarts = wrap ? 16 : 16 - arc + 1;
    // screen every position of the score region; the few that pass are listed and scored with all lanes busy
    for (int i = t; i < FX_SW * FX_SH; i += 256) {
        const int r = i / FX_SW, c = i - r * FX_SW;
        const int x = x0 - 1 + c, y = y0 - 1 + r;
        sc[r * FX_SP + c] = 0;
        if (x >= 3 && x < W - 3 && y >= 3 && y < H - 3 && fastx_screen(pix + (r + 3) * FX_PW + c + 3, thr))
            cand[atomicAdd(&ncand, 1)] = (uint16_t)((r << 8) | c);
    }
    __syncthreads();
    for (int j = t; j < ncand; j += 256) {
        const int r = cand[j] >> 8, c = cand[j] & 255;
        sc[r * FX_SP + c] = (uint8_t)fastx_score_arc(pix + (r + 3) * FX_PW + c + 3, thr, arc, nstarts);
    }
    __syncthreads();
    const int lane = lane_id(), wave = wave_id();
    const int wi = x0 >> 5;
    uint32_t* mfr = mask + f * H * MW;
    unsigned long long* const rc = rowcnt + f * H;
    for (int row = wave; row < FX_H; row += 4) {
        const int y = y0 + row;
        if (y >= H) break;                                        // (wave-uniform)
        const uint8_t* sp = sc + (row + 1) * FX_SP + lane + 1;
        const int s = sp[0];
        bool keep = s > 0 && x0 + lane < W;
        if (keep && nms) {
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx)
                    if ((dx || dy) && (int)sp[dy * FX_SP + dx] >= s) keep = false;
        }
        const unsigned long long bits = __ballot(keep);
        const uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
        if (lane == 0) {
            if (wi < MW) mfr[(size_t)y * MW + wi] = lo;
            if (wi + 1 < MW) mfr[(size_t)y * MW + wi + 1] = hi;
            if (bits) {
                const bool yv = (y - 16 >= 0) && (y + 16 < H);
                const uint32_t v = yv ? (uint32_t)(__popc(lo & desc_column_bits(wi, W)) + __popc(hi & desc_column_bits(wi + 1, W))) : 0u;
                atomicAdd(rc + y, (unsigned long long)(uint32_t)__popcll(bits) | ((unsigned long long)v << 32));
            }
        }
    }
}

void launch_fast_ext(const FrameSet& fs, int thr, int arc, int wrap, int nms, uint32_t* mask, int2* rowcnt, hipStream_t s)
{
    const int ntc = (fs.W + FX_W - 1) / FX_W, ntr = (fs.H + FX_H - 1) / FX_H;
    const int dwords = ((fs.W & 3) == 0) && (((uintptr_t)fs.gray & 3) == 0);
    hipLaunchKernelGGL(fastx_tile_kernel, dim3((unsigned)((size_t)ntc * ntr * fs.frames)), dim3(256), 0, s, fs.gray, fs.H,
                       fs.W, thr, arc, wrap, nms, mask, mask_words(fs.W), reinterpret_cast<unsigned long long*>(rowcnt),
                       ntc, ntr, dwords);
}

// Descriptor rows for a caller-supplied keypoint list (any order): slot = rank among keypoints
// that pass the 16-px border filter.  One workgroup walks the list in chunks of 1024.
__global__ __launch_bounds__(1024) void slots_kernel(const int2* __restrict__ kpxy, int n, int H, int W,
                                                     int32_t* __restrict__ kpslot,
                                                     int32_t* __restrict__ counts)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = lane_id(), wave = wave_id();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        int v = 0;
        if (i < n) {
            const int2 p = kpxy[i];
            v = p.y >= 16 && p.y < H - 16 && p.x >= 16 && p.x < W - 16;     // (no sums: any int32 is a legal input)
        }
        const int iv = wave_incl_scan(v, lane);
        if (lane == 63) wsum[wave] = iv;
        __syncthreads();
        int base = carry;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        if (i < n) kpslot[i] = v ? base + iv - 1 : -1;
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + iv;
        __syncthreads();
    }
    if (threadIdx.x == 0) { counts[0] = n; counts[1] = carry; }
}

void launch_slots(const int2* kpxy, int n, int H, int W, int32_t* kpslot, int32_t* counts, hipStream_t s)
{
    hipLaunchKernelGGL(slots_kernel, dim3(1), dim3(1024), 0, s, kpxy, n, H, W, kpslot, counts);
}

// ---------------------------------------------------------------------------------------------
// Orientation + steered BRIEF.
//
// Orientation (reference src/orb.cpp:110-140, :236-260): raw moments over the 31x31 square
// clipped to the image.  The reference accumulates in double; all terms are integers below 2^23,
// so int32 sums are exact.  angle = (float)(atan2(m01, m10) * (180/pi)), +360 (in double) if
// negative, 0 when m00 == 0.
//
// Descriptor (reference src/orb.cpp:166-231): theta = angle * (pi/180) in double; each pattern
// point is rotated with double cos/sin (two rounded products, one rounded sum), narrowed to
// float, added to the float centre, rounded half-to-even; samples outside the image read 0.
//
// Work decomposition: a workgroup owns 4 * KPW consecutive keypoints and hands them to its four
// wavefronts in x order (see the kernel); each wavefront runs three phases over its KPW keypoints.
// A: patch moments, four keypoints per step with 16 lanes each (DPP row reduction); the totals are
// parked in the lane whose number is the keypoint's position in the batch.
// B: every lane computes atan2 / cos / sin for ITS keypoint -- the double-precision library calls
// cost thousands of cycles, so they run once per batch with 64 keypoints side by side instead of
// once per keypoint with 64 lanes doing the same thing.  C: for each keypoint in turn, cos / sin /
// centre are broadcast from its lane and lane l of round r evaluates test pair 64r + l; the wave's
// ballot is descriptor bytes 8r .. 8r+7.
// ---------------------------------------------------------------------------------------------
constexpr double ORB_PI = 3.1415926535897932384626433832795;

// Parity audit.  The reference narrows doubles that come out of libm calls (atan2; products with cos / sin)
// to float (src/orb.cpp:129, :195-199).  A libm whose results differ in the last bits gives the same float
// unless the double sits next to the midpoint of two neighbouring floats.  near_float_midpoint: the 29 bits
// the narrowing drops are within 4 of that midpoint pattern.
__device__ __forceinline__ bool near_float_midpoint(double v)
{
    const uint32_t lo = (uint32_t)__double2loint(v);
    return (((lo + 4u) - 0x10000000u) & 0x1FFFFFFFu) <= 8u;
}

// One rotated sample coordinate before narrowing, and the float centre it is added to.  Bit 0: near a
// float midpoint; bit 1: the two candidate floats would also round to different pixels.
__device__ __forceinline__ uint32_t audit_coord(double v, float centre)
{
    if (!near_float_midpoint(v)) return 0u;
    const float f = (float)v;
    if (f == 0.0f) return 1u;
    const float g = __uint_as_float(__float_as_uint(f) + (((double)f < v) == (f >= 0.0f) ? 1u : 0xFFFFFFFFu));
    return __float2int_rn(centre + f) != __float2int_rn(centre + g) ? 3u : 1u;
}

// The coefficient table of orb_trig.h behind a pointer the optimiser cannot trace: the table is then read by
// scalar loads where it is used.  (Seen through, the 25 doubles are kept in vector registers for the whole
// kernel, on top of the sampling pattern's 32.)
typedef const __attribute__((address_space(4))) double* TrigTable;
__device__ __forceinline__ TrigTable trig_table()
{
    TrigTable k = (TrigTable)orbtrig::COEF;
    asm volatile("" : "+s"(k));
    return k;
}

template <class QuarterTable>
__device__ __forceinline__ float angle_from_moments(int m00, int m10, int m01, TrigTable K, QuarterTable Q,
                                                    bool* near_midpoint = nullptr)
{
    // straight-line: an all-zero patch has m10 = m01 = 0, for which the series returns 0 like atan2(0, 0)
    bool unsure;
    const double deg = orbtrig::atan2_deg_moments(m01, m10, unsure, K, Q);
    if (near_midpoint) *near_midpoint = m00 != 0 && near_float_midpoint(deg);
    float ang = (float)deg;
    const float wrapped = (float)((double)ang + 360.0);
    ang = ang < 0 ? wrapped : ang;
    return m00 == 0 ? 0.0f : ang;
}

// Test hook: the two functions of orb_trig.h as the kernels call them (coefficient table through the opaque
// pointer, quarter table in LDS), one element per thread.
__global__ __launch_bounds__(256) void trig_selftest_kernel(const int32_t* __restrict__ m01, const int32_t* __restrict__ m10,
                                                            int n, double* __restrict__ deg, uint8_t* __restrict__ unsure,
                                                            const double* __restrict__ rad, int m,
                                                            double* __restrict__ sn, double* __restrict__ cs)
{
    __shared__ double quarter[8];
    if (threadIdx.x < 5) quarter[threadIdx.x] = orbtrig::COEF[orbtrig::K_ATAN_QUARTER + threadIdx.x];
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const TrigTable K = trig_table();
    if (i < n) {
        bool u;
        deg[i] = orbtrig::atan2_deg_moments(m01[i], m10[i], u, K, quarter);
        unsure[i] = u;
    }
    if (i < m) {
        double s, c;
        orbtrig::sincos_turn(rad[i], s, c, K);
        sn[i] = s;
        cs[i] = c;
    }
}

void launch_trig_selftest(const int32_t* m01, const int32_t* m10, int n, double* deg, uint8_t* unsure,
                          const double* rad, int m, double* sn, double* cs, hipStream_t s)
{
    const int big = n > m ? n : m;
    hipLaunchKernelGGL(trig_selftest_kernel, dim3((unsigned)((big + 255) / 256)), dim3(256), 0, s, m01, m10, n, deg, unsure,
                       rad, m, sn, cs);
}

// Sum over the 64 lanes with DPP adds; the total lands in lane 63 and is returned wave-uniform.
__device__ __forceinline__ int wave_total(int v)
{
    v = dpp_add<0x111, 0xf>(v);   // row_shr:1
    v = dpp_add<0x112, 0xf>(v);   // row_shr:2
    v = dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add<0x118, 0xf>(v);   // row_shr:8  -> lane 15 of each row of 16 holds the row sum
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __builtin_amdgcn_readlane(v, 63);
}

// 4 aligned bytes at a 32-bit byte offset from a wave-uniform base (scalar base + vector offset).
__device__ __forceinline__ uint32_t load_u32_at(const uint8_t* __restrict__ base, uint32_t off)
{
    return *reinterpret_cast<const uint32_t*>(base + off);
}

// 12 bytes from a 4-byte-aligned address in one load.
struct __attribute__((aligned(4))) Tri32 { uint32_t v[3]; };
__device__ __forceinline__ Tri32 load_tri_at(const uint8_t* __restrict__ base, uint32_t off)
{
    return *reinterpret_cast<const Tri32*>(base + off);
}

// The arithmetic of the 16-lanes-per-keypoint patch moments, shared by the kernel that fetches patch rows from
// global memory and the one that reads them from an LDS tile.  Lane q < 15 of a keypoint's group of 16 =
// (row slot rs = q / 3, chunk ch = q % 3): it holds dwords 3 ch .. 3 ch + 2 of patch rows rs, rs + 5, ...,
// rs + 30 (pix[it].v[j]; anything may be in a dword whose row is outside the patch or the image as long as the
// caller passes it as 0), the 31 columns starting `r` bytes into dword 0.  Per dword: m00 += sum of bytes
// (v_sad_u8, accumulating), m10' += bytes . (u + 16) (v_dot4_u32_u8); only dwords 0, 7 and 8 contain bytes
// outside the 31 columns.  m01 += v * rowsum per row.  The three sums are then reduced over the 16 lanes of a
// DPP row (4 steps); the totals come out in lane 15 of each row of 16.
struct MomentLane {
    uint32_t mk[3], w[3];
    int rs;
};
__device__ __forceinline__ MomentLane moment_lane(int r, int q, bool lane_on)
{
    MomentLane m;
    const int ch = q < 15 ? q % 3 : 0;
    m.rs = q < 15 ? q / 3 : 0;
    // column offset t = 4 c + byte is inside the patch iff r <= t <= 30 + r: partial dwords are
    // c = 0 (chunk 0, first), c = 7 and c = 8 (chunk 2, second and third)
    m.mk[0] = m.mk[1] = m.mk[2] = 0xFFFFFFFFu;
    if (ch == 0) m.mk[0] = 0xFFFFFFFFu << (8 * r);
    if (ch == 2) {
        m.mk[1] = r == 0 ? 0x00FFFFFFu : 0xFFFFFFFFu;
        m.mk[2] = r < 2 ? 0u : (r == 2 ? 0x000000FFu : 0x0000FFFFu);
    }
    if (!lane_on) { m.mk[0] = 0u; m.mk[1] = 0u; m.mk[2] = 0u; }
    // weights u + 16 = t - r + 1 per byte.  In dword 0 the bytes left of the patch would get
    // negative weights (their borrows would spill into valid bytes), so that word is the r = 0
    // pattern {1,2,3,4} shifted up by r bytes; from dword 1 on every weight is >= 2.
#pragma unroll
    for (int j = 0; j < 3; ++j)       // the byte 4 (3 ch + j) - r + 1 in all four lanes, plus 0 1 2 3
        m.w[j] = __builtin_amdgcn_perm(0u, (uint32_t)(4 * (3 * ch + j) - r + 1), 0u) + 0x03020100u;
    if (ch == 0) m.w[0] = 0x04030201u << (8 * r);
    return m;
}

// One row slot's dword j of step `it` into the running sums (rsum = this row's byte sum so far).
__device__ __forceinline__ void moment_dword(const MomentLane& m, uint32_t v, int j, uint32_t& rsum, uint32_t& t10)
{
    const uint32_t pm = v & m.mk[j];
    rsum = __builtin_amdgcn_sad_u8(pm, 0u, rsum);
    t10 = __builtin_amdgcn_udot4(pm, m.w[j], t10, false);
}

__device__ __forceinline__ void moment_row(const MomentLane& m, int it, uint32_t rsum, uint32_t& t00, int& t01)
{
    t00 += rsum;
    t01 += __mul24(m.rs + 5 * it - 15, (int)rsum);
}

// (t00, t10', t01) of one lane -> (m00, m10, m01) summed over the row of 16 lanes, in its lane 15.
__device__ __forceinline__ void moment_reduce(uint32_t t00, uint32_t t10, int t01, int& s00, int& s10, int& s01)
{
    s00 = (int)t00;
    s10 = (int)t10 - 16 * s00;
    s01 = t01;
    s00 = dpp_add<0x111, 0xf>(s00); s10 = dpp_add<0x111, 0xf>(s10); s01 = dpp_add<0x111, 0xf>(s01);
    s00 = dpp_add<0x112, 0xf>(s00); s10 = dpp_add<0x112, 0xf>(s10); s01 = dpp_add<0x112, 0xf>(s01);
    s00 = dpp_add<0x114, 0xf>(s00); s10 = dpp_add<0x114, 0xf>(s10); s01 = dpp_add<0x114, 0xf>(s01);
    s00 = dpp_add<0x118, 0xf>(s00); s10 = dpp_add<0x118, 0xf>(s10); s01 = dpp_add<0x118, 0xf>(s01);
}

// Patch moments on the dword path (rows are whole aligned dwords: W % 4 == 0 and a 4-byte-aligned
// frame): FOUR keypoints per step, 16 lanes each.  The 31 columns x-15 .. x+15 lie inside 9 aligned
// dwords starting at xa = (x-15) & ~3.  Lane q of a group = (row slot rs = q / 3, chunk ch = q % 3):
// it reads dwords 3 ch .. 3 ch + 2 of rows rs, rs + 5, ..., so one load instruction covers five
// whole 36-byte rows per keypoint (few cache lines per instruction; lane 15 idles).  Per dword:
// m00 += sum of bytes (v_sad_u8, accumulating), m10' += bytes . (u + 16) (v_dot4_u32_u8); only
// dwords 0, 7 and 8 contain bytes outside the 31 columns.  m01 += v * rowsum per row.  The three
// sums are reduced over the 16 lanes of a DPP row (4 steps, against 6 steps over 64 lanes for one
// keypoint at a time).  The totals come out in lane 15 of each row; (x, y) are per lane.
__device__ __forceinline__ void patch_moments_quad(const uint8_t* __restrict__ img, int H, int W, int x, int y,
                                                   bool group_valid, int q, int& s00, int& s10, int& s01)
{
    const int r = (x - 15) & 3;
    const int xa = (x - 15) - r;              // aligned start column (may be negative)
    const bool lane_on = group_valid && q < 15;
    const MomentLane m = moment_lane(r, q, lane_on);
    const int rs = m.rs, ch = q < 15 ? q % 3 : 0;

    const int gx = xa + 12 * ch, gy0 = y - 15 + rs;
    Tri32 pix[7];
    const bool interior = xa >= 0 && xa <= W - 36 && y >= 15 && y < H - 15;
    if (__all(interior || !group_valid)) {
        // Lanes without work read the frame's first bytes; their masks are zero.
        const uint32_t off = lane_on ? (uint32_t)(gy0 * W + gx) : 0u;
#pragma unroll
        for (int it = 0; it < 6; ++it) pix[it] = load_tri_at(img, off + (lane_on ? (uint32_t)(5 * it * W) : 0u));
        // rows 30 .. 34: only row slot 0 is still inside the patch (the others are masked out below)
        pix[6] = load_tri_at(img, (lane_on && rs == 0) ? off + (uint32_t)(30 * W) : off);
    } else if (__all((xa >= 0 && xa <= W - 36) || !group_valid)) {
        // clipped at the top or bottom edge only: whole 12-byte loads, rows outside the image read as 0
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int gy = gy0 + 5 * it;
            const bool rowok = lane_on && rs + 5 * it < 31 && gy >= 0 && gy < H;
            pix[it] = load_tri_at(img, rowok ? (uint32_t)(gy * W + gx) : 0u);
            if (!rowok) { pix[it].v[0] = 0u; pix[it].v[1] = 0u; pix[it].v[2] = 0u; }
        }
    } else {
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int gy = gy0 + 5 * it;
            const bool rowok = lane_on && rs + 5 * it < 31 && gy >= 0 && gy < H;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const bool ok = rowok && gx + 4 * j >= 0 && gx + 4 * j < W;
                const uint32_t v = load_u32_at(img, ok ? (uint32_t)(gy * W + gx + 4 * j) : 0u);
                pix[it].v[j] = ok ? v : 0u;
            }
        }
    }
    uint32_t t00 = 0, t10 = 0;
    int t01 = 0;
#pragma unroll
    for (int it = 0; it < 7; ++it) {
        uint32_t rsum = 0;
        const bool rowin = it < 6 || rs == 0;       // rows 30 .. 34: only row 30 is in the patch
#pragma unroll
        for (int j = 0; j < 3; ++j) moment_dword(m, rowin ? pix[it].v[j] : 0u, j, rsum, t10);
        moment_row(m, it, rsum, t00, t01);
    }
    moment_reduce(t00, t10, t01, s00, s10, s01);
}

// Any width / alignment: one byte per lane and step.
__device__ __forceinline__ void patch_moments_bytes(const uint8_t* __restrict__ img, int H, int W, int x,
                                                    int y, int lane, int& t00, int& t10, int& t01)
{
    int s00 = 0, s10 = 0, s01 = 0;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
        const int q = lane + 64 * j;      // 0 .. 960 walks the 31x31 square row-major
        const int v = q / 31, u = q - v * 31;
        const int yy = y + v - 15, xx = x + u - 15;
        if (q < 961 && yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const int I = img[(size_t)yy * W + xx];
            s00 += I;
            s10 += (u - 15) * I;
            s01 += (v - 15) * I;
        }
    }
    t00 = wave_total(s00);
    t10 = wave_total(s10);
    t01 = wave_total(s01);
}

// Staging of a 45-row x 12-dword window (aligned start column xa, even first row y0: the 43 rows /
// columns a keypoint's samples can reach, plus alignment, plus slack that lets neighbouring
// keypoints use the same window) into dense LDS rows of 48 bytes: lane = 12 * rs + dw, row = rs + 5 * it.
// Lanes 60..63 repeat lanes 0..3.  (Rows of 64 bytes would make the index a shift, but put every
// second row on the same LDS banks: the sample gather then takes 4.5 instead of 3.2 cycles per 32
// lanes, and the staging stores conflict as well.)
constexpr int WIN_ROWS = 45, WIN_DW = 12, WIN_PITCH_DW = 12;
constexpr int WIN_UINT4 = WIN_ROWS * WIN_PITCH_DW / 4;      // 135
constexpr int ROT_UINT4 = 32;                               // 31 rotation-table entries of 16 bytes

__device__ __forceinline__ void stage_load(const uint8_t* __restrict__ img, int H, int W, int xa, int y0,
                                           int dw, int rs, uint32_t (&regs)[9])
{
    const int gx = xa + 4 * dw, gy0 = y0 + rs;
    if (xa >= 0 && xa <= W - 4 * WIN_DW && y0 >= 0 && y0 <= H - WIN_ROWS) {
        const uint32_t off = (uint32_t)(gy0 * W + gx);
#pragma unroll
        for (int it = 0; it < 9; ++it) regs[it] = load_u32_at(img, off + (uint32_t)(5 * it * W));
    } else {
#pragma unroll
        for (int it = 0; it < 9; ++it) {
            const int row = rs + 5 * it, gy = gy0 + 5 * it;
            const bool ok = row < WIN_ROWS && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const uint32_t v = load_u32_at(img, ok ? (uint32_t)(gy * W + gx) : 0u);
            regs[it] = ok ? v : 0u;
        }
    }
}

__device__ __forceinline__ void stage_write(uint32_t* win, int dw, int rs, const uint32_t (&regs)[9])
{
    uint32_t* const p = win + rs * WIN_PITCH_DW + dw;
#pragma unroll
    for (int it = 0; it < 9; ++it) p[5 * it * WIN_PITCH_DW] = regs[it];
}

// 1.5 * 2^23: adding it to a float |t| < 2^22 leaves rint(t) (ties to even, the cvRound of the
// reference, src/orb.cpp:203-206) in the low mantissa bits: bits(t + magic) = 0x4B400000 + rint(t).
constexpr float RND_MAGIC = 12582912.0f;
constexpr uint32_t RND_MAGIC_BITS = 0x4B400000u;

// At most 5 waves per SIMD: with more, the windows and patches of the resident waves no longer fit
// the vector L1 cache and the kernel runs 10 - 15 % slower on MI355X.
template <int KPW, bool AUDIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void orient_describe_kernel(
    const uint8_t* __restrict__ gray, int H, int W, const int2* __restrict__ kpxy,
    const int32_t* __restrict__ kpslot, const int32_t* __restrict__ counts, int cap,
    const float* __restrict__ angle_in, float* __restrict__ kp7, float* __restrict__ angle_out,
    uint8_t* __restrict__ desc, int32_t* __restrict__ valid_idx, int dwords, int nbf, int32_t* __restrict__ risk)
{
    // 1-D grid of frames * nbf workgroups, nbf (workgroups per frame) odd: see xcd_contiguous_item().
    const uint32_t item = xcd_contiguous_item();
    const int f = (int)(item / (uint32_t)nbf);
    const int blk = (int)(item - (uint32_t)f * (uint32_t)nbf);
    const int wg_base = blk * (4 * KPW);
    int n = counts[2 * f];
    n = n < cap ? n : cap;
    if (wg_base >= n) return;                 // uniform over the workgroup
    const int wg_n = (n - wg_base) < 4 * KPW ? (n - wg_base) : 4 * KPW;
    const int lane = lane_id(), wave = wave_id();
    const uint8_t* img = gray + (size_t)f * H * W;
    const size_t k0 = (size_t)f * cap + wg_base;

    // The workgroup owns 4 * KPW consecutive keypoints of the scan order: a band of image rows
    // across the full width, in which consecutive keypoints are hundreds of pixels apart.  They
    // are processed in order of x instead (counting sort on 256 column buckets; the order inside a
    // bucket is whatever the atomics give and does not matter, every result goes to the keypoint's
    // own row).  The sorted sequence is dealt to the four waves in runs of 8 positions (lanes 8c ..
    // 8c + 7 of wave w hold run 4c + w): the waves sweep the band from left to right together, so a
    // cache line one wave fetched is an L2 hit for the other three, and the corners of one cluster,
    // which are neighbours in x, mostly fall into one run and share one staged window.  (Runs of 64
    // share a few more windows but triple the L2 misses; runs of 1 share none.)
    __shared__ int bucket_count[256];
    __shared__ int wave_sum_lds[4];
    __shared__ uint16_t order[256];
    __shared__ double atan_quarter_lds[8];    // atan(k / 4), k = 0 .. 4: looked up per lane (written before the barriers below)
    int kidx = -1;                            // this lane's keypoint, relative to wg_base
    {
        const int t = threadIdx.x;
        bucket_count[t] = 0;
        if (t < 5) atan_quarter_lds[t] = orbtrig::COEF[orbtrig::K_ATAN_QUARTER + t];
        __syncthreads();
        const int shift = W > 256 ? 32 - __builtin_clz((unsigned)(W - 1)) - 8 : 0;
        int bkt = 0, rank = 0;
        if (t < wg_n) {
            bkt = kpxy[k0 + t].x >> shift;
            bkt = bkt < 0 ? 0 : (bkt > 255 ? 255 : bkt);
            rank = atomicAdd(&bucket_count[bkt], 1);
        }
        __syncthreads();
        const int cnt = bucket_count[t];
        const int incl = wave_incl_scan(cnt, lane);
        if (lane == 63) wave_sum_lds[wave] = incl;
        __syncthreads();
        int before = incl - cnt;
        for (int w = 0; w < wave; ++w) before += wave_sum_lds[w];
        bucket_count[t] = before;             // exclusive prefix: first sorted position of bucket t
        __syncthreads();
        if (t < wg_n) order[bucket_count[bkt] + rank] = (uint16_t)t;
        __syncthreads();
        constexpr int C = 8;                  // run of sorted positions per wave (KPW is a multiple)
        const int pos = ((lane / C) * 4 + wave) * C + lane % C;
        if (lane < KPW && pos < wg_n) kidx = order[pos];
    }
    const int nk = (int)__builtin_popcountll(__ballot(kidx >= 0));   // lanes 0 .. nk-1 hold keypoints
    if (nk == 0) return;

    // this lane's keypoint
    const bool mine = kidx >= 0;
    const int2 p = mine ? kpxy[k0 + kidx] : make_int2(0, 0);
    const int slot = (mine && desc) ? kpslot[k0 + kidx] : -1;

    // ---- phase A: moments, one keypoint at a time ----
    // Caller-supplied points may be any int32.  A patch further than 15 pixels outside the image is
    // empty wherever it is, so the moment phase works on clamped coordinates and its index
    // arithmetic cannot overflow.
    const int mxc = p.x < -64 ? -64 : (p.x > W + 64 ? W + 64 : p.x);
    const int myc = p.y < -64 ? -64 : (p.y > H + 64 ? H + 64 : p.y);
    float angle = 0.0f;
    if (angle_in) {
        if (mine) angle = angle_in[k0 + kidx];
    } else {
        int m00 = 0, m10 = 0, m01 = 0;
        if (dwords) {
            // group g = lane / 16 takes batch position 4 * step + g; the totals of a group come out
            // in its lane 15 and are handed to the lane that owns the keypoint
            const int g = lane >> 4, q = lane & 15;
            const int total_lane = 16 * (lane & 3) + 15;
            for (int step = 0; 4 * step < nk; ++step) {
                const int src = 4 * step + g;
                const int x = __shfl(mxc, src), y = __shfl(myc, src);
                int s00, s10, s01;
                patch_moments_quad(img, H, W, x, y, src < nk, q, s00, s10, s01);
                const int t00 = __shfl(s00, total_lane), t10 = __shfl(s10, total_lane), t01 = __shfl(s01, total_lane);
                if ((lane >> 2) == step) { m00 = t00; m10 = t10; m01 = t01; }
            }
        } else {
            for (int j = 0; j < nk; ++j) {
                const int x = __builtin_amdgcn_readlane(mxc, j), y = __builtin_amdgcn_readlane(myc, j);
                int t00, t10, t01;
                patch_moments_bytes(img, H, W, x, y, lane, t00, t10, t01);
                if (lane == j) { m00 = t00; m10 = t10; m01 = t01; }
            }
        }
        // ---- phase B (first half): one atan2 per lane ----
        if (AUDIT) {
            bool near;
            angle = angle_from_moments(m00, m10, m01, trig_table(), atan_quarter_lds, &near);
            if (mine && near) atomicAdd(&risk[4 * f], 1);
        } else {
            angle = angle_from_moments(m00, m10, m01, trig_table(), atan_quarter_lds);
        }
    }
    if (mine) {
        if (angle_out) angle_out[k0 + kidx] = angle;
        if (kp7) {
            // [x, y, size, angle, response, octave, class_id] (reference src/main_interface.cpp:46-52)
            float* row = kp7 + (k0 + kidx) * 7;
            row[0] = (float)p.x; row[1] = (float)p.y; row[2] = 7.0f; row[3] = angle;
            row[4] = 0.0f; row[5] = 0.0f; row[6] = -1.0f;
        }
        if (slot >= 0 && valid_idx) valid_idx[(size_t)f * cap + slot] = wg_base + kidx;
    }
    if (!desc) return;
    if (__ballot(slot >= 0) == 0ull) return;   // (lanes beyond the batch carry slot = -1)

    // ---- phase B (second half): one cos / sin per lane ----
    const double rad = (double)angle * (ORB_PI / 180.0);
    double c, s;
    if (rad >= 0.0 && rad <= 6.5) {          // every angle this library computes; a caller's angle may be anything
        orbtrig::sincos_turn(rad, s, c, trig_table());
    } else {
        c = cos(rad);
        s = sin(rad);
    }
    const int c_lo = __double2loint(c), c_hi = __double2hiint(c);
    const int s_lo = __double2loint(s), s_hi = __double2hiint(s);

    // ---- phase C: descriptors, one keypoint at a time ----
    unsigned long long todo = __ballot(slot >= 0);
    int aud_mid = 0, aud_dec = 0;            // parity audit: this lane's coordinates near a float midpoint / decisive
    if (dwords) {
        // The 512 samples of a keypoint fall in the 43x43 square around it (|rotated offset| <=
        // 21.3).  Scattered byte gathers from global memory cost one cache line per lane; instead the
        // window is staged into this wave's LDS slab with coalesced dword reads (rows of 12 aligned
        // dwords, zero where outside the image -- which is also what an out-of-image sample must
        // read, reference src/orb.cpp:211-217) and the samples are gathered from LDS.  The next
        // keypoint's window is loaded into registers while this one is evaluated.
        //
        // Rotation: a pattern coordinate is an integer k in -15 .. 15 and the reference forms
        // k * cos and k * sin in double (src/orb.cpp:195-199).  Lanes 0 .. 30 compute those 31
        // product pairs once per keypoint into an LDS table; every sample then takes its four
        // products from the table (two 16-byte reads) instead of multiplying.
        //
        // One window and one table per wave are enough: a wave's LDS operations execute in program
        // order, so the next keypoint's stores cannot overtake this keypoint's reads.
        __shared__ uint4 slab[4][WIN_UINT4 + ROT_UINT4];
        const int wv = wave_id();
        uint32_t* const win = reinterpret_cast<uint32_t*>(slab[wv]);
        const uint8_t* const pb = reinterpret_cast<const uint8_t*>(slab[wv]);
        const uint8_t* const rot = reinterpret_cast<const uint8_t*>(slab[wv] + WIN_UINT4);
        const int dw = lane % WIN_DW, rs = (lane / WIN_DW) % 5;
        // this lane's table entries, one test pair per round
        const double2 *t1x[4], *t1y[4], *t2x[4], *t2y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t q = reinterpret_cast<const uint32_t*>(c_pairs)[64 * r + lane];
            t1x[r] = reinterpret_cast<const double2*>(rot) + ((int)(int8_t)(q & 255u) + 15);
            t1y[r] = reinterpret_cast<const double2*>(rot) + ((int)(int8_t)((q >> 8) & 255u) + 15);
            t2x[r] = reinterpret_cast<const double2*>(rot) + ((int)(int8_t)((q >> 16) & 255u) + 15);
            t2y[r] = reinterpret_cast<const double2*>(rot) + ((int)(int8_t)(q >> 24) + 15);
        }
        const double kd = (double)((lane < 31 ? lane : 30) - 15);
        // rounds >= ROT_TABLE_ROUNDS multiply directly (balances the LDS and vector pipes)
        constexpr int ROT_TABLE_ROUNDS = 2;
        double dpx1[4], dpy1[4], dpx2[4], dpy2[4];
#pragma unroll
        for (int r = ROT_TABLE_ROUNDS; r < 4; ++r) {
            const uint32_t q = reinterpret_cast<const uint32_t*>(c_pairs)[64 * r + lane];
            dpx1[r] = (double)(int8_t)(q & 255u);
            dpy1[r] = (double)(int8_t)((q >> 8) & 255u);
            dpx2[r] = (double)(int8_t)((q >> 16) & 255u);
            dpy2[r] = (double)(int8_t)(q >> 24);
        }
        // The window in LDS starts at column ox (a multiple of 4), row oy (even).  A keypoint whose
        // 43 x 43 box of reachable samples lies inside the window that is already staged uses it as
        // it is: corners come in clusters of adjacent pixels, and the wave meets them one after the
        // other because its keypoints are consecutive in x order.
        uint32_t regs[9];
        int j = __builtin_ctzll(todo);
        todo &= todo - 1;
        int ox, oy;
        {
            const int x = __builtin_amdgcn_readlane(p.x, j), y = __builtin_amdgcn_readlane(p.y, j);
            ox = (x - 21) & ~3;
            oy = (y - 21) & ~1;
            stage_load(img, H, W, ox, oy, dw, rs, regs);
        }
        bool fresh = true;                       // regs hold a window that is not in LDS yet
        while (true) {
            if (fresh) stage_write(win, dw, rs, regs);
            const int xj = __builtin_amdgcn_readlane(p.x, j), yj = __builtin_amdgcn_readlane(p.y, j);
            const int sj = __builtin_amdgcn_readlane(slot, j);
            const double cj = __hiloint2double(__builtin_amdgcn_readlane(c_hi, j), __builtin_amdgcn_readlane(c_lo, j));
            const double sn = __hiloint2double(__builtin_amdgcn_readlane(s_hi, j), __builtin_amdgcn_readlane(s_lo, j));
            if (lane < 31) {
                double2 e;
                e.x = kd * cj;
                e.y = kd * sn;
                reinterpret_cast<double2*>(slab[wv] + WIN_UINT4)[lane] = e;
            }
            const bool more = todo != 0ull;
            int jn = j, nox = ox, noy = oy;
            fresh = false;
            if (more) {
                jn = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int x = __builtin_amdgcn_readlane(p.x, jn), y = __builtin_amdgcn_readlane(p.y, jn);
                const bool covered = ox <= x - 21 && x + 21 < ox + 4 * WIN_DW && oy <= y - 21 && y + 21 < oy + WIN_ROWS;
                if (!covered) {
                    nox = (x - 21) & ~3;
                    noy = (y - 21) & ~1;
                    stage_load(img, H, W, nox, noy, dw, rs, regs);
                    fresh = true;
                }
            }
            unsigned long long bal[4] = {0ull, 0ull, 0ull, 0ull};
            // A non-finite angle (possible only for caller-supplied angles) makes every coordinate
            // NaN; the reference's cvRound then yields INT_MIN, outside the image: all samples read 0.
            if (fabs(cj) <= 1.0 && fabs(sn) <= 1.0) {
                // Rounding and window-relative indexing in one float add: for an integer m,
                // (t + (magic - m)) has the bits 0x4B400000 + rint(t) - m -- for EVEN m: the add rounds
                // a tie to an even sum, which is an even rint(t) only then.  With m = the window's
                // first column / row (multiples of 4 / 2) the low bits are the sample's column / row
                // in the window; the 0x4B400000 parts of row * 48 + column are a constant folded
                // into the address.
                float cx = (float)xj, cy = (float)yj;
                float mx = RND_MAGIC - (float)ox, my = RND_MAGIC - (float)oy;
                asm volatile("" : "+v"(cx), "+v"(cy), "+v"(mx), "+v"(my));   // vector copies: scalar operands halve the add rate
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double vx1, vy1, vx2, vy2;
                    if (r < ROT_TABLE_ROUNDS) {
                        const double2 e1x = *t1x[r], e1y = *t1y[r], e2x = *t2x[r], e2y = *t2y[r];
                        vx1 = e1x.x - e1y.y;     // px1 * c - py1 * s
                        vy1 = e1x.y + e1y.x;     // px1 * s + py1 * c
                        vx2 = e2x.x - e2y.y;
                        vy2 = e2x.y + e2y.x;
                    } else {
                        vx1 = dpx1[r] * cj - dpy1[r] * sn;
                        vy1 = dpx1[r] * sn + dpy1[r] * cj;
                        vx2 = dpx2[r] * cj - dpy2[r] * sn;
                        vy2 = dpx2[r] * sn + dpy2[r] * cj;
                    }
                    if (AUDIT) {
                        const uint32_t a4[4] = {audit_coord(vx1, cx), audit_coord(vy1, cy), audit_coord(vx2, cx),
                                                audit_coord(vy2, cy)};
#pragma unroll
                        for (int k = 0; k < 4; ++k) { aud_mid += a4[k] & 1u; aud_dec += a4[k] >> 1; }
                    }
                    const float x1r = (float)vx1, y1r = (float)vy1, x2r = (float)vx2, y2r = (float)vy2;
                    const uint32_t ax = __float_as_uint((cx + x1r) + mx);
                    const uint32_t ay = __float_as_uint((cy + y1r) + my);
                    const uint32_t bx = __float_as_uint((cx + x2r) + mx);
                    const uint32_t by = __float_as_uint((cy + y2r) + my);
                    // row * 48 + column in ONE instruction: the low 16 bits of both floats' bit patterns are
                    // the row / column themselves (0x4B400000 has none of them set), and v_mad_legacy_u16 multiplies
                    // and adds the low halves only and writes zeros to the upper half of its result (v_mad_u16 would
                    // keep the destination's old upper half).
                    static_assert(4 * WIN_PITCH_DW == 48, "the row pitch is the literal in the instruction");
                    static_assert((RND_MAGIC_BITS & 0xFFFFu) == 0u, "row and column sit in the low 16 bits");
                    uint32_t i1, i2;
                    asm("v_mad_legacy_u16 %0, %1, 48, %2" : "=v"(i1) : "v"(ay), "v"(ax));
                    asm("v_mad_legacy_u16 %0, %1, 48, %2" : "=v"(i2) : "v"(by), "v"(bx));
                    const int v1 = pb[i1], v2 = pb[i2];
                    bal[r] = __ballot(v1 < v2);
                }
            }
            // lanes 0..3 store one 8-byte word each: 32 contiguous bytes per descriptor
            if (lane < 4) {
                const unsigned long long w = lane == 0 ? bal[0] : lane == 1 ? bal[1] : lane == 2 ? bal[2] : bal[3];
                reinterpret_cast<unsigned long long*>(desc + ((size_t)f * cap + sj) * 32)[lane] = w;
            }
            if (!more) break;
            j = jn;
            ox = nox;
            oy = noy;
        }
    } else {
        // this lane's four test pairs (one per round), as doubles
        double px1[4], py1[4], px2[4], py2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t q = reinterpret_cast<const uint32_t*>(c_pairs)[64 * r + lane];
            px1[r] = (double)(int8_t)(q & 255u);
            py1[r] = (double)(int8_t)((q >> 8) & 255u);
            px2[r] = (double)(int8_t)((q >> 16) & 255u);
            py2[r] = (double)(int8_t)(q >> 24);
        }
        while (todo) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int sj = __builtin_amdgcn_readlane(slot, j);
            const double cj = __hiloint2double(__builtin_amdgcn_readlane(c_hi, j), __builtin_amdgcn_readlane(c_lo, j));
            const double sn = __hiloint2double(__builtin_amdgcn_readlane(s_hi, j), __builtin_amdgcn_readlane(s_lo, j));
            const bool finite = fabs(cj) <= 1.0 && fabs(sn) <= 1.0;
            const float cx = (float)__builtin_amdgcn_readlane(p.x, j), cy = (float)__builtin_amdgcn_readlane(p.y, j);
            unsigned long long bal[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double vx1 = px1[r] * cj - py1[r] * sn, vy1 = px1[r] * sn + py1[r] * cj;
                const double vx2 = px2[r] * cj - py2[r] * sn, vy2 = px2[r] * sn + py2[r] * cj;
                if (AUDIT && finite) {
                    const uint32_t a4[4] = {audit_coord(vx1, cx), audit_coord(vy1, cy), audit_coord(vx2, cx),
                                            audit_coord(vy2, cy)};
#pragma unroll
                    for (int k = 0; k < 4; ++k) { aud_mid += a4[k] & 1u; aud_dec += a4[k] >> 1; }
                }
                const float x1r = (float)vx1, y1r = (float)vy1, x2r = (float)vx2, y2r = (float)vy2;
                const int ax = __float2int_rn(cx + x1r), ay = __float2int_rn(cy + y1r);
                const int bx = __float2int_rn(cx + x2r), by = __float2int_rn(cy + y2r);
                int v1 = 0, v2 = 0;
                if (finite && ay >= 0 && ay < H && ax >= 0 && ax < W) v1 = img[(size_t)ay * W + ax];
                if (finite && by >= 0 && by < H && bx >= 0 && bx < W) v2 = img[(size_t)by * W + bx];
                bal[r] = __ballot(v1 < v2);
            }
            if (lane < 4) {
                const unsigned long long w = lane == 0 ? bal[0] : lane == 1 ? bal[1] : lane == 2 ? bal[2] : bal[3];
                reinterpret_cast<unsigned long long*>(desc + ((size_t)f * cap + sj) * 32)[lane] = w;
            }
        }
    }
    if (AUDIT) {
        const int tm = wave_total(aud_mid), td = wave_total(aud_dec);
        if (lane == 0) {
            if (tm) atomicAdd(&risk[4 * f + 1], tm);
            if (td) atomicAdd(&risk[4 * f + 2], td);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Tile-centric orientation + descriptors for keypoint lists in scan order.
//
// The per-keypoint kernel above fetches ~2 KB around every keypoint with row-strided loads (31 rows
// of 36 bytes for the moments, 45 rows of 48 bytes for the sampling window) and stages a window per
// keypoint.  Here a workgroup owns one 240 x 64 image tile, loads it ONCE with its halo (32 columns
// left and right, 22 rows above, 21 below) into LDS by coalesced 16-byte loads (zero outside the
// image, which is what both the clipped moments and the out-of-image samples need), and serves all
// keypoints whose centre lies in the tile from there: what is left is vector arithmetic.  The
// keypoints of a 64-row band are one contiguous range of the frame's list (scan order), known from
// the row prefix; every tile of the band scans that range for its columns.
// ---------------------------------------------------------------------------------------------
constexpr int DT_W = 240, DT_H = 64;                 // 1920 and 3840 are whole numbers of tiles
constexpr int DT_MX = 32, DT_MY = 22;                // left / top margin: >= 21, multiple of 16 / even
constexpr int DT_PITCH = DT_W + 2 * DT_MX;           // 304 bytes per LDS row
constexpr int DT_PITCH_DW = DT_PITCH / 4;            // 76
constexpr int DT_PITCH_Q = DT_PITCH / 16;            // 19 16-byte units
constexpr int DT_ROWS = DT_H + 2 * DT_MY - 1;        // 107: rows -22 .. +84 of the band
constexpr int DT_UNITS = DT_ROWS * DT_PITCH_Q;       // 2033
constexpr int DT_STEPS = (DT_UNITS + 255) / 256;     // tile load steps of 256 units (one per thread)
constexpr int DT_SEG = 256;                          // list entries scanned per pass (one per thread)
constexpr int DT_LCAP = 384;                         // keypoints gathered before they are processed
constexpr int DT_BATCH = 64;                         // keypoints per moments / trigonometry / descriptor round: one wave's lanes
static_assert(DT_W % 16 == 0 && DT_MX % 16 == 0 && DT_MY % 2 == 0 && DT_MX >= 22 && DT_MY >= 21, "halo, even origin, aligned units");
static_assert(DT_ROWS * DT_PITCH < 65536, "LDS byte index fits 16 bits");
static_assert(256 == 13 * DT_PITCH_Q + 9, "tile walk: 256 units on = 13 rows and 9 units");

struct DescribeTileLds {
    uint32_t tile32[DT_UNITS * 4];
    // the last load step writes whole waves: up to 256 * DT_STEPS - DT_UNITS units past the tile, into cs,
    // which holds nothing yet at that time
    double2 cs[DT_BATCH];                            // moments, then (cos, sin)
    double2 bxy[DT_BATCH];                           // biases of the fixed-grid rotation (x, y), see describe_tile_body
    uint32_t lst_xy[DT_LCAP];                        // (row in tile << 16) | column in tile
    int32_t lst_idx[DT_LCAP];                        // keypoint index in the frame
    int32_t lst_slot[DT_LCAP];                       // its descriptor row, or -1 (fetched with the list entry)
    double atan_quarter[8];                          // atan(k / 4), k = 0 .. 4: looked up per lane
    int lst_count;
};
static_assert((256 * DT_STEPS - DT_UNITS) * 16 <= (int)sizeof(double2) * DT_BATCH, "load overrun stays inside cs");
static_assert(sizeof(DescribeTileLds) <= 40960, "four workgroups per CU");

// One workgroup's tile: `item` numbers the (frame, band, tile column) triples, tile column fastest.
template <bool AUDIT>
__device__ __forceinline__ void describe_tile_body(
    DescribeTileLds& L, uint32_t item, const uint8_t* __restrict__ gray, int H, int W, const int2* __restrict__ kpxy,
    const int32_t* __restrict__ kpslot, const int32_t* __restrict__ counts, int cap,
    const int2* __restrict__ rowoff, float* __restrict__ kp7, uint8_t* __restrict__ desc,
    int32_t* __restrict__ valid_idx, int ntc, int nbands, int32_t* __restrict__ risk)
{
    uint32_t* const tile32 = L.tile32;
    uint32_t* const lst_xy = L.lst_xy;
    int32_t* const lst_idx = L.lst_idx;
    double2* const cs = L.cs;

    const int tc = (int)(item % (uint32_t)ntc);
    const uint32_t rest = item / (uint32_t)ntc;
    const int band = (int)(rest % (uint32_t)nbands);
    const int f = (int)(rest / (uint32_t)nbands);
    int n = counts[2 * f];
    n = n < cap ? n : cap;
    const int2* ro = rowoff + (size_t)f * H;
    int k0 = ro[band * DT_H].x;
    int k1 = (band + 1) * DT_H < H ? ro[(band + 1) * DT_H].x : n;
    k0 = k0 < n ? k0 : n;
    k1 = k1 < n ? k1 : n;
    if (k0 >= k1) return;                            // no keypoints in this band (uniform)

    const int t = (int)threadIdx.x, lane = lane_id(), wave = wave_id();
    const int X0 = tc * DT_W, tx0 = X0 - DT_MX, ty0 = band * DT_H - DT_MY;
    const uint8_t* img = gray + (size_t)f * H * W;
    const size_t kf = (size_t)f * cap;
    const uint8_t* const pb = reinterpret_cast<const uint8_t*>(tile32);

    // first pass of the list scan: in flight together with the tile
    int2 pnext = make_int2(-1, -1);
    int snext = -1;
    if (k0 + t < k1) { pnext = kpxy[kf + k0 + t]; snext = kpslot[kf + k0 + t]; }

    // ---- the tile: rows ty0 .. ty0 + 106, columns tx0 .. tx0 + 303, straight into LDS in 16-byte units.
    // Bounds-checked buffer loads return zero past either end of the frame (rows above and below the image);
    // units left or right of the image get an offset that is out of range on purpose (W is a multiple of
    // 16 and so is tx0: a unit is inside or outside as a whole). ----
    {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, H * W, 0x00020000);
        int row = t / DT_PITCH_Q;
        int gx = tx0 + 16 * (t - row * DT_PITCH_Q);
        int off = (ty0 + row) * W + gx;
#pragma unroll
        for (int k = 0; k < DT_STEPS; ++k) {
            const bool ok = (uint32_t)gx < (uint32_t)W && (k + 1 < DT_STEPS || t + 256 * k < DT_UNITS);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (__attribute__((address_space(3))) void*)(tile32 + 4 * (256 * k + 64 * wave)), 16,
                ok ? (uint32_t)off : 0x80000000u, 0, 0, 0);
            const bool wrap = gx + 16 * 9 >= tx0 + DT_PITCH;
            gx += wrap ? 16 * 9 - DT_PITCH : 16 * 9;
            off += wrap ? 14 * W + 16 * 9 - DT_PITCH : 13 * W + 16 * 9;
        }
    }
    if (t == 0) L.lst_count = 0;
    if (t < 5) L.atan_quarter[t] = orbtrig::COEF[orbtrig::K_ATAN_QUARTER + t];

    // this lane's four test pairs, as doubles.  (A rotation table in LDS, as the per-keypoint kernel keeps,
    // would cost 16 KB of LDS reads per keypoint: with the image tile in LDS as well, that is the limit.)
    double dpx1[4], dpy1[4], dpx2[4], dpy2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double2* pd = reinterpret_cast<const double2*>(c_pairs_d.v) + 2 * (64 * r + lane);
        const double2 a = pd[0], b = pd[1];
        dpx1[r] = a.x; dpy1[r] = a.y; dpx2[r] = b.x; dpy2[r] = b.y;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int aud_mid = 0, aud_dec = 0;            // parity audit: this lane's coordinates near a float midpoint / decisive

    // Fixed-grid rotation (the common case of the descriptor rounds).  The reference's sample coordinate is
    // RNE(fl32(c + fl32(d))) with d = fl64(fl64(px*cos) - fl64(py*sin)) (src/orb.cpp:195-206).  For S = c + px*cos
    // - py*sin in exact arithmetic, fl32(c + fl32(d)) lies within e = 2^(E-25) + 2^-20 + 2^-45 of S when every
    // such sum of this tile is below 2^E (half an ulp of the float sum, half an ulp of the narrowed d < 32, the
    // three double roundings).  So when S is farther than e from every half-integer, the reference's pixel is
    // round(S) whatever the intermediate roundings did.  Two dependent v_fma_f64 on top of the bias
    // 1.5 * 2^36 + (c - tile origin) + 1/2 + zs * 2^-16 leave S' = S + delta, |delta| <= 2^-16, on a grid of 2^-16
    // in the low dword of the result: high half = floor(S' + 1/2 + zs * 2^-16) in tile coordinates, low half = the
    // fraction.  A fraction below 2 zs + 1 means S' is within zs grid steps of a half-integer; otherwise S is at
    // least zs * 2^-16 > e away from it and the high half is the reference's pixel.  zs = 2^(E-9) + 1.  Rounds
    // with a flagged coordinate in any lane are redone with the reference's own arithmetic (about one
    // coordinate in 8 000).
    int coord_max = X0 + DT_W + 21;
    {
        const int ymax = (band + 1) * DT_H + 21;
        coord_max = coord_max > ymax ? coord_max : ymax;
    }
    const int ebits = 32 - __builtin_clz((unsigned)coord_max);           // every float sum of this tile is < 2^ebits
    const int zs = (1 << (ebits > 9 ? ebits - 9 : 0)) + 1;
    const uint32_t zone = 2u * (uint32_t)zs + 1u;
    const double bias0 = 103079215104.0 + 0.5 + (double)zs * (1.0 / 65536.0);      // 1.5 * 2^36 + ...

    for (int s0 = k0; s0 < k1; s0 += DT_SEG) {
        // ---- gather: entries of the band's range whose column falls into this tile ----
        const int2 p = pnext;
        const int pslot = snext;
        pnext = make_int2(-1, -1);
        if (s0 + DT_SEG + t < k1) { pnext = kpxy[kf + s0 + DT_SEG + t]; snext = kpslot[kf + s0 + DT_SEG + t]; }
        if (p.x >= X0 && p.x < X0 + DT_W) {
            const int pos = atomicAdd(&L.lst_count, 1);
            lst_xy[pos] = ((uint32_t)(p.y - ty0) << 16) | (uint32_t)(p.x - tx0);
            lst_idx[pos] = s0 + t;
            L.lst_slot[pos] = pslot;
        }
        __syncthreads();
        const int cnt = L.lst_count;
        __syncthreads();                   // everybody has read the count before the next pass can add to it
        const bool last = s0 + DT_SEG >= k1;
        if (!(cnt > DT_LCAP - DT_SEG || last)) continue;     // (uniform) room for another pass
        if (cnt > 0) {
            for (int bb = 0; bb < cnt; bb += DT_BATCH) {
                const int nb = (cnt - bb) < DT_BATCH ? (cnt - bb) : DT_BATCH;
                // ---- moments: 16 keypoints per step over the workgroup, 16 lanes each ----
                {
                    const int g = lane >> 4, q = lane & 15;
                    for (int e0 = 4 * wave; e0 < nb; e0 += 16) {      // a wave without keypoints left is done
                        const int e = e0 + g;
                        const bool valid = e < nb;
                        const uint32_t pxy = lst_xy[bb + (valid ? e : 0)];
                        const int lx = (int)(pxy & 0xFFFFu), ly = (int)(pxy >> 16);
                        const int r = (lx - 15) & 3;
                        const int xa = (lx - 15) - r;
                        const MomentLane ml = moment_lane(r, q, valid && q < 15);
                        const int rs = ml.rs, ch = q < 15 ? q % 3 : 0;
                        const uint32_t* src = tile32 + __mul24(ly - 15 + rs, DT_PITCH_DW) + (xa >> 2) + 3 * ch;
                        uint32_t t00 = 0, t10 = 0;
                        int t01 = 0;
#pragma unroll
                        for (int it = 0; it < 7; ++it) {
                            uint32_t rsum = 0;
                            const bool rowin = it < 6 || rs == 0;       // rows 30 .. 34: only row 30 is in the patch
#pragma unroll
                            for (int j = 0; j < 3; ++j)
                                moment_dword(ml, rowin ? src[5 * it * DT_PITCH_DW + j] : 0u, j, rsum, t10);
                            moment_row(ml, it, rsum, t00, t01);
                        }
                        int s00, s10, s01;
                        moment_reduce(t00, t10, t01, s00, s10, s01);
                        if (valid && q == 15) {
                            int* m = reinterpret_cast<int*>(&cs[e]);
                            m[0] = s00; m[1] = s10; m[2] = s01;
                        }
                    }
                }
                __syncthreads();
                // ---- one keypoint per lane of the first wave: angle, keypoint row, cos / sin.  Straight-line code
                // with nothing to wait for but LDS: the descriptor slot came in with the list entry. ----
                if (t < nb) {
                    const int* m = reinterpret_cast<const int*>(&cs[t]);
                    const int m00 = m[0], m10 = m[1], m01 = m[2];
                    const uint32_t pxy = lst_xy[bb + t];
                    const int i = lst_idx[bb + t];
                    const int slot = L.lst_slot[bb + t];
                    const TrigTable K = trig_table();
                    float angle;
                    if (AUDIT) {
                        bool near;
                        angle = angle_from_moments(m00, m10, m01, K, L.atan_quarter, &near);
                        if (near) atomicAdd(&risk[4 * f], 1);
                    } else {
                        angle = angle_from_moments(m00, m10, m01, K, L.atan_quarter);
                    }
                    float* row = kp7 + (kf + i) * 7;
                    row[0] = (float)((int)(pxy & 0xFFFFu) + tx0); row[1] = (float)((int)(pxy >> 16) + ty0);
                    row[2] = 7.0f; row[3] = angle; row[4] = 0.0f; row[5] = 0.0f; row[6] = -1.0f;
                    if (slot >= 0 && valid_idx) valid_idx[kf + slot] = i;
                    const double rad = (double)angle * (ORB_PI / 180.0);
                    double2 v;
                    orbtrig::sincos_turn(rad, v.y, v.x, K);
                    cs[t] = v;
                    v.x = bias0 + (double)(int)(pxy & 0xFFFFu);
                    v.y = bias0 + (double)(int)(pxy >> 16);
                    L.bxy[t] = v;
                }
                __syncthreads();
                // ---- descriptors: the waves take the keypoints of the batch in turn ----
                for (int e = wave; e < nb; e += 4) {
                    const int slot = __builtin_amdgcn_readfirstlane(L.lst_slot[bb + e]);
                    if (slot < 0) continue;
                    const uint32_t pxy = lst_xy[bb + e];
                    const double2 csv = cs[e];
                    const double cj = csv.x, sn = csv.y;
                    float cx = (float)((int)(pxy & 0xFFFFu) + tx0), cy = (float)((int)(pxy >> 16) + ty0);
                    const float mx = RND_MAGIC - (float)tx0, my = RND_MAGIC - (float)ty0;
                    const uint32_t pitch = DT_PITCH;
                    int v1[4], v2[4];
                    if (!AUDIT) {
                        const double2 bv = L.bxy[e];
                        uint32_t ia = 0, ib = 0, mn[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const uint32_t ax = (uint32_t)__double2loint(__builtin_fma(dpx1[r], cj, __builtin_fma(-dpy1[r], sn, bv.x)));
                            const uint32_t ay = (uint32_t)__double2loint(__builtin_fma(dpx1[r], sn, __builtin_fma(dpy1[r], cj, bv.y)));
                            const uint32_t bx = (uint32_t)__double2loint(__builtin_fma(dpx2[r], cj, __builtin_fma(-dpy2[r], sn, bv.x)));
                            const uint32_t by = (uint32_t)__double2loint(__builtin_fma(dpx2[r], sn, __builtin_fma(dpy2[r], cj, bv.y)));
                            // byte index = high half of y * pitch + high half of x; the destination's high half is
                            // kept by this instruction (zero from the start), so ia / ib are chained
                            asm("v_mad_u16 %0, %1, %2, %3 op_sel:[1,0,1,0]" : "+v"(ia) : "v"(ay), "s"(pitch), "v"(ax));
                            v1[r] = pb[ia];
                            asm("v_mad_u16 %0, %1, %2, %3 op_sel:[1,0,1,0]" : "+v"(ib) : "v"(by), "s"(pitch), "v"(bx));
                            v2[r] = pb[ib];
                            uint32_t m;
                            asm("v_min3_u16 %0, %1, %2, %3" : "=v"(m) : "v"(ax), "v"(ay), "v"(bx));
                            asm("v_min_u16 %0, %1, %2" : "=v"(mn[r]) : "v"(m), "v"(by));
                        }
                        uint32_t mall;
                        asm("v_min3_u16 %0, %1, %2, %3" : "=v"(mall) : "v"(mn[0]), "v"(mn[1]), "v"(mn[2]));
                        asm("v_min_u16 %0, %1, %2" : "=v"(mall) : "v"(mall), "v"(mn[3]));
                        if (__ballot((mall & 0xFFFFu) < zone) != 0ull) {        // (uniform) rare: the reference's own chain
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if (__ballot((mn[r] & 0xFFFFu) < zone) == 0ull) continue;
                                const double vx1 = dpx1[r] * cj - dpy1[r] * sn, vy1 = dpx1[r] * sn + dpy1[r] * cj;
                                const double vx2 = dpx2[r] * cj - dpy2[r] * sn, vy2 = dpx2[r] * sn + dpy2[r] * cj;
                                const float x1r = (float)vx1, y1r = (float)vy1, x2r = (float)vx2, y2r = (float)vy2;
                                const uint32_t ax = __float_as_uint((cx + x1r) + mx);
                                const uint32_t ay = __float_as_uint((cy + y1r) + my);
                                const uint32_t bx = __float_as_uint((cx + x2r) + mx);
                                const uint32_t by = __float_as_uint((cy + y2r) + my);
                                uint32_t j1, j2;
                                asm("v_mad_legacy_u16 %0, %1, %2, %3" : "=v"(j1) : "v"(ay), "s"(pitch), "v"(ax));
                                asm("v_mad_legacy_u16 %0, %1, %2, %3" : "=v"(j2) : "v"(by), "s"(pitch), "v"(bx));
                                v1[r] = pb[j1];
                                v2[r] = pb[j2];
                            }
                        }
                    } else {
                    uint32_t i1[4], i2[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double vx1 = dpx1[r] * cj - dpy1[r] * sn, vy1 = dpx1[r] * sn + dpy1[r] * cj;
                        const double vx2 = dpx2[r] * cj - dpy2[r] * sn, vy2 = dpx2[r] * sn + dpy2[r] * cj;
                        if (AUDIT) {
                            const uint32_t a4[4] = {audit_coord(vx1, cx), audit_coord(vy1, cy), audit_coord(vx2, cx),
                                                    audit_coord(vy2, cy)};
#pragma unroll
                            for (int k = 0; k < 4; ++k) { aud_mid += a4[k] & 1u; aud_dec += a4[k] >> 1; }
                        }
                        const float x1r = (float)vx1, y1r = (float)vy1, x2r = (float)vx2, y2r = (float)vy2;
                        const uint32_t ax = __float_as_uint((cx + x1r) + mx);
                        const uint32_t ay = __float_as_uint((cy + y1r) + my);
                        const uint32_t bx = __float_as_uint((cx + x2r) + mx);
                        const uint32_t by = __float_as_uint((cy + y2r) + my);
                        asm("v_mad_legacy_u16 %0, %1, %2, %3" : "=v"(i1[r]) : "v"(ay), "s"(pitch), "v"(ax));
                        asm("v_mad_legacy_u16 %0, %1, %2, %3" : "=v"(i2[r]) : "v"(by), "s"(pitch), "v"(bx));
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v1[r] = pb[i1[r]]; v2[r] = pb[i2[r]]; }
                    }
                    unsigned long long bal[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) bal[r] = __ballot(v1[r] < v2[r]);
                    if (lane < 4) {
                        const unsigned long long wd = lane == 0 ? bal[0] : lane == 1 ? bal[1] : lane == 2 ? bal[2] : bal[3];
                        reinterpret_cast<unsigned long long*>(desc + (kf + slot) * 32)[lane] = wd;
                    }
                }
                __syncthreads();
            }
        }
        if (t == 0) L.lst_count = 0;
        __syncthreads();
    }
    if (AUDIT) {
        const int tm = wave_total(aud_mid), td = wave_total(aud_dec);
        if (lane == 0) {
            if (tm) atomicAdd(&risk[4 * f + 1], tm);
            if (td) atomicAdd(&risk[4 * f + 2], td);
        }
    }
}

template <bool AUDIT>
__global__ __launch_bounds__(256, 4) void describe_tile_kernel(
    const uint8_t* __restrict__ gray, int H, int W, const int2* __restrict__ kpxy,
    const int32_t* __restrict__ kpslot, const int32_t* __restrict__ counts, int cap,
    const int2* __restrict__ rowoff, float* __restrict__ kp7, uint8_t* __restrict__ desc,
    int32_t* __restrict__ valid_idx, int ntc, int nbands, int32_t* __restrict__ risk)
{
    __shared__ __attribute__((aligned(16))) DescribeTileLds L;
    describe_tile_body<AUDIT>(L, xcd_contiguous_item(), gray, H, W, kpxy, kpslot, counts, cap, rowoff, kp7, desc,
                              valid_idx, ntc, nbands, risk);
}

void launch_describe_tiles(const FrameSet& fs, const int2* kpxy, const int32_t* kpslot, const int32_t* counts,
                           int cap, const int2* rowoff, float* kp7, uint8_t* desc, int32_t* valid_idx, int32_t* risk,
                           hipStream_t s)
{
    const int ntc = (fs.W + DT_W - 1) / DT_W, nbands = (fs.H + DT_H - 1) / DT_H;
    static const unsigned pad = lds_pad_from_env("ORBAMD_PAD_LDS_DESCRIBE", (const void*)describe_tile_kernel<false>);
    const dim3 grid((unsigned)((size_t)ntc * nbands * fs.frames));
    if (risk)
        hipLaunchKernelGGL(describe_tile_kernel<true>, grid, dim3(256), 0, s, fs.gray, fs.H, fs.W, kpxy, kpslot, counts,
                           cap, rowoff, kp7, desc, valid_idx, ntc, nbands, risk);
    else
        hipLaunchKernelGGL(describe_tile_kernel<false>, grid, dim3(256), pad, s, fs.gray, fs.H, fs.W, kpxy, kpslot,
                           counts, cap, rowoff, kp7, desc, valid_idx, ntc, nbands, risk);
}

template <int KPW>
static void launch_orient_describe_kpw(const FrameSet& fs, const int2* kpxy, const int32_t* kpslot,
                                       const int32_t* counts, int cap, const float* angle_in, float* kp7,
                                       float* angle_out, uint8_t* desc, int32_t* valid_idx, int32_t* risk, hipStream_t s)
{
    const int dwords = ((fs.W & 3) == 0) && (((uintptr_t)fs.gray & 3) == 0);
    const int nbf = ((cap + 4 * KPW - 1) / (4 * KPW)) | 1;
    const dim3 grid((unsigned)((size_t)nbf * fs.frames));
    if (risk)
        hipLaunchKernelGGL((orient_describe_kernel<KPW, true>), grid, dim3(256), 0, s, fs.gray, fs.H, fs.W, kpxy, kpslot,
                           counts, cap, angle_in, kp7, angle_out, desc, valid_idx, dwords, nbf, risk);
    else
        hipLaunchKernelGGL((orient_describe_kernel<KPW, false>), grid, dim3(256), 0, s, fs.gray, fs.H, fs.W, kpxy, kpslot,
                           counts, cap, angle_in, kp7, angle_out, desc, valid_idx, dwords, nbf, risk);
}

void launch_orient_describe(const FrameSet& fs, const int2* kpxy, const int32_t* kpslot,
                            const int32_t* counts, int cap, const float* angle_in, float* kp7,
                            float* angle_out, uint8_t* desc, int32_t* valid_idx, int32_t* risk, hipStream_t s)
{
    // Big launches: 64 keypoints per wave (library calls amortised 64-fold).  Small launches
    // (single frames): 16 per wave, so that a few thousand keypoints still spread over the chip.
    if ((long)fs.frames * cap >= 64L * 8192)
        launch_orient_describe_kpw<64>(fs, kpxy, kpslot, counts, cap, angle_in, kp7, angle_out, desc, valid_idx, risk, s);
    else
        launch_orient_describe_kpw<16>(fs, kpxy, kpslot, counts, cap, angle_in, kp7, angle_out, desc, valid_idx, risk, s);
}

// ---------------------------------------------------------------------------------------------
// Hamming top-2 (cv2.BFMatcher(NORM_HAMMING).knnMatch(q, t, 2); reference views.py:162).
//
// One query per lane, held in 8 VGPRs.  The train row is the same for all 64 lanes, so its 8
// dwords are wave-uniform operands.  Per (query, train) pair: 8 XOR + 8 popcount-accumulate,
// then key = (distance << 18) | trainIndex and a running top-2 of keys by unsigned min / med3.
// Unsigned key order is (distance, trainIndex) lexicographic order, which is the order in which
// OpenCV's insertion (strict '<' while scanning ascending train index) leaves its two slots.
// The train axis can be split over blockIdx.z; partial (m1, m2) keys are merged afterwards.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;

__device__ __forceinline__ void top2_insert(uint32_t& m1, uint32_t& m2, uint32_t key)
{
    // second smallest of {m1 <= m2, key} is their median; v_med3_u32 has no builtin.
    uint32_t med;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(med) : "v"(m1), "v"(m2), "v"(key));
    m2 = med;
    m1 = m1 < key ? m1 : key;
}

__device__ __forceinline__ void knn2_store(uint32_t m1, uint32_t m2, int32_t* idx, int32_t* dist,
                                           uint8_t* good, size_t o)
{
    const int i1 = m1 == KEY_NONE ? -1 : (int)(m1 & 0x3FFFFu);
    const int d1 = m1 == KEY_NONE ? -1 : (int)(m1 >> 18);
    const int i2 = m2 == KEY_NONE ? -1 : (int)(m2 & 0x3FFFFu);
    const int d2 = m2 == KEY_NONE ? -1 : (int)(m2 >> 18);
    reinterpret_cast<int2*>(idx)[o] = make_int2(i1, i2);
    reinterpret_cast<int2*>(dist)[o] = make_int2(d1, d2);
    // d1 < 0.75 * d2 for integers <= 256  <=>  4*d1 < 3*d2 (reference views.py:168)
    if (good) good[o] = (i2 >= 0 && 4 * d1 < 3 * d2) ? 1 : 0;
}

// popcount(q ^ t) over 256 bits as a chain of popcount-accumulate instructions
// (v_bcnt_u32_b32 dst = popcount(src0) + src1; the first link adds the inline constant 0).
__device__ __forceinline__ uint32_t hamming256(const uint32_t (&q)[8], const uint32_t* __restrict__ t)
{
    uint32_t d;
    {
        const uint32_t x = q[0] ^ t[0];
        asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d) : "v"(x));
    }
#pragma unroll
    for (int w = 1; w < 8; ++w) {
        const uint32_t x = q[w] ^ t[w];
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(x));
    }
    return d;
}

// The same for two rows held as two uint4 each.
__device__ __forceinline__ uint32_t hamming256_regs(const uint4& q0, const uint4& q1, const uint4& t0, const uint4& t1)
{
    return (uint32_t)(__popc(q0.x ^ t0.x) + __popc(q0.y ^ t0.y) + __popc(q0.z ^ t0.z) + __popc(q0.w ^ t0.w) +
                      __popc(q1.x ^ t1.x) + __popc(q1.y ^ t1.y) + __popc(q1.z ^ t1.z) + __popc(q1.w ^ t1.w));
}

// QPL queries per lane (register tiling over the query axis: one train row fetched per QPL * 64
// pairs), RPT train rows per trip (their scalar loads are issued together), BS threads per block.
template <int QPL, int RPT, int BS, bool PARTIAL>
__global__ __launch_bounds__(BS) void knn2_kernel(const uint8_t* __restrict__ qdesc,
                                                  const uint8_t* __restrict__ tdesc,
                                                  const int32_t* __restrict__ qcounts,
                                                  const int32_t* __restrict__ tcounts,
                                                  const int32_t* __restrict__ pairs, int qframe0,
                                                  int tframe0, int nframes, int cap, int nq_fixed, int nt_fixed,
                                                  int splits, int nbp, uint32_t* __restrict__ partial,
                                                  int32_t* __restrict__ idx, int32_t* __restrict__ dist,
                                                  uint8_t* __restrict__ good)
{
    // 1-D grid of npairs * splits * nbp work items (nbp = query blocks per pair, made odd by the
    // launcher, see xcd_contiguous_item()): a pair's blocks run on one XCD and share its train rows
    // in that XCD's L2.
    int pair, zsplit, qblk;
    {
        const uint32_t logical = xcd_contiguous_item();
        const uint32_t per_pair = (uint32_t)nbp * (uint32_t)splits;
        pair = (int)(logical / per_pair);
        const uint32_t rem = logical - (uint32_t)pair * per_pair;
        zsplit = (int)(rem / (uint32_t)nbp);
        qblk = (int)(rem - (uint32_t)zsplit * (uint32_t)nbp);
    }
    int qf = 0, tf = 0, nq = nq_fixed, nt = nt_fixed;
    if (qcounts) {
        if (pairs) {
            qf = pairs[2 * pair]; tf = pairs[2 * pair + 1];
            // a pair that names a frame outside the batch is skipped (uniform over the workgroup)
            if ((unsigned)qf >= (unsigned)nframes || (unsigned)tf >= (unsigned)nframes) return;
        } else { qf = pair + 1 + qframe0; tf = pair + tframe0; }
        nq = qcounts[2 * qf + 1]; nq = nq < cap ? nq : cap;
        nt = tcounts[2 * tf + 1]; nt = nt < cap ? nt : cap;
    }
    const int q0 = qblk * (BS * QPL);
    if (q0 >= nq) return;

    uint32_t q[QPL][8];
#pragma unroll
    for (int u = 0; u < QPL; ++u) {
        const int qi = q0 + u * BS + threadIdx.x;
        const uint4* qp = reinterpret_cast<const uint4*>(qdesc + ((size_t)qf * cap + (qi < nq ? qi : q0)) * 32);
        const uint4 a = qp[0], b = qp[1];
        q[u][0] = a.x; q[u][1] = a.y; q[u][2] = a.z; q[u][3] = a.w;
        q[u][4] = b.x; q[u][5] = b.y; q[u][6] = b.z; q[u][7] = b.w;
    }

    const int chunk = (nt + splits - 1) / splits;
    const int t0 = zsplit * chunk;
    int t1 = t0 + chunk; t1 = t1 < nt ? t1 : nt;
    const uint32_t* __restrict__ tp = reinterpret_cast<const uint32_t*>(tdesc + (size_t)tf * cap * 32) + (size_t)t0 * 8;

    uint32_t m1[QPL], m2[QPL];
#pragma unroll
    for (int u = 0; u < QPL; ++u) { m1[u] = KEY_NONE; m2[u] = KEY_NONE; }
    // A row can only change a lane's top-2 if its distance is below that lane's second-best
    // distance (an equal distance carries a larger train index, so its key is larger).  After the
    // first few hundred rows that is rare, so each trip tests min(distances) < second-best once
    // and skips the key build and the three-way inserts when no lane needs them.  The bound is
    // refreshed only inside the slow path; a stale (larger) bound merely takes that path, where
    // inserting a key that is not below m2 leaves (m1, m2) unchanged.
    uint32_t bound[QPL];
#pragma unroll
    for (int u = 0; u < QPL; ++u) bound[u] = KEY_NONE >> 18;

    int j = t0;
    for (; j + RPT <= t1; j += RPT, tp += 8 * RPT) {
        uint32_t d[RPT][QPL];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
#pragma unroll
            for (int u = 0; u < QPL; ++u) d[r][u] = hamming256(q[u], tp + 8 * r);
        }
        bool any = false;
#pragma unroll
        for (int u = 0; u < QPL; ++u) {
            uint32_t mn = d[0][u];
#pragma unroll
            for (int r = 1; r < RPT; ++r) mn = mn < d[r][u] ? mn : d[r][u];
            any |= mn < bound[u];
        }
        if (any) {
#pragma unroll
            for (int u = 0; u < QPL; ++u) {
#pragma unroll
                for (int r = 0; r < RPT; ++r)
                    top2_insert(m1[u], m2[u], (d[r][u] << 18) | (uint32_t)(j + r));
                bound[u] = m2[u] >> 18;
            }
        }
    }
    for (; j < t1; ++j, tp += 8) {
#pragma unroll
        for (int u = 0; u < QPL; ++u)
            top2_insert(m1[u], m2[u], (hamming256(q[u], tp) << 18) | (uint32_t)j);
    }
#pragma unroll
    for (int u = 0; u < QPL; ++u) {
        const int qi = q0 + u * BS + threadIdx.x;
        if (qi < nq) {
            if (PARTIAL) {
                const size_t o = (((size_t)pair * splits + zsplit) * cap + qi);
                reinterpret_cast<uint2*>(partial)[o] = make_uint2(m1[u], m2[u]);
            } else {
                knn2_store(m1[u], m2[u], idx, dist, good, (size_t)pair * cap + qi);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Hamming top-2 on the matrix cores.
//
// With every descriptor bit b stored as the FP4 (E2M1) value 1 - 2b, the dot product of two
// 256-element rows is 256 - 2 * hamming(q, t): sums of +-1 products, exact in the f32 accumulator.
// v_mfma_scale_f32_32x32x64_f8f6f4 (both operands FP4, both scales 2^0) multiplies a 32-row train
// tile by a 32-column query tile over 64 bits per instruction, four instructions per tile.
//
// The accumulator comes out with the column (query) on the lane and 16 train rows in the 16
// registers of a lane (row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).  Starting the
// accumulator at -256 - row / 4096 makes each register -(2 * distance + row / 4096): a complete
// (distance, train index) key as one exact float (10 integer bits + 12 fraction bits), larger =
// better, so no key has to be built: the 16 keys of a lane are reduced to the best of each group of
// eight rows and those two enter the lane's running top-2 (knn2_select_tile).  After every tile the
// two running keys are raised by 32 / 4096, which makes their index fields relative to the start of
// the next tile (negative for rows already
// seen); the fields stay inside the distance step of 2 for 8192 rows, after which (and at the
// end) the keys are decoded into the integer (distance << 18) | index form and merged there.
// Lanes l and l + 32 hold disjoint train rows of the same query and are merged at the end.
//
// Operands: a tile of 32 descriptor rows is 4096 bytes of FP4 values laid out as
// [piece 0..7][row 0..31][16 bytes]; piece p holds descriptor dword p as 32 nibbles.  A lane's
// operand for k-step ks is the 16 bytes at ks * 1024 + lane * 16, i.e. dword 2 ks + (lane / 32) of
// row lane % 32.  Query operands are built in registers from the descriptor dwords once per wave;
// train tiles are fetched as plain descriptor dwords (a quarter of the bytes of an expanded image),
// expanded by the thread that stages them and read from LDS with lane-linear 16-byte accesses.
// ---------------------------------------------------------------------------------------------
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// 32 descriptor bits -> 32 FP4 (E2M1) values +-1.0 in four dwords: nibble = sign bit | 0b0010.  Dword s
// of the result takes bits s, s + 4, ..., s + 28; both operands of a product use this assignment,
// which is all a dot product needs.
__device__ __forceinline__ uint4 fp4_of_bits(uint32_t w)
{
    // three independent shifts, then one and-or per dword.  (A chain of three additions, which issue
    // faster than shifts, measured 1 % slower: it sits on the path to the tile's barrier.)
    const uint32_t w2 = w << 1, w4 = w << 2, w8 = w << 3;
    uint4 o;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(o.x) : "v"(w8), "s"(0x88888888u), "v"(0x22222222u));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(o.y) : "v"(w4), "s"(0x88888888u), "v"(0x22222222u));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(o.z) : "v"(w2), "s"(0x88888888u), "v"(0x22222222u));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(o.w) : "v"(w), "s"(0x88888888u), "v"(0x22222222u));
    return o;
}

// Two new keys a, b into the running pair m1 >= m2 in three instructions: with s1 >= s2 >= s3 the
// sorted {m1, a, b}, the new pair is (s1, max(s2, m2)) because m2 <= m1 <= s1.
// max(x, y) is written med3(x, y, +inf): the median intrinsic needs no NaN canonicalisation of
// its inputs, which a float max of raw accumulator registers would get.
// The two-step fmaxf is selected as one v_max3_f32.  (No inline asm here: the compiler inserts the wait
// states between a matrix instruction and the first vector instruction that reads its result only for
// instructions it emitted itself; an asm v_max3_f32 in this place read stale accumulators.)
__device__ __forceinline__ void top2_insert2_f32(float& m1, float& m2, float a, float b, float pinf)
{
    const float s2 = __builtin_amdgcn_fmed3f(m1, a, b);
    m1 = __builtin_fmaxf(__builtin_fmaxf(m1, a), b);
    m2 = __builtin_amdgcn_fmed3f(s2, m2, pinf);
}

// The 16 keys a lane holds of one tile into its running pair.  The registers 8 j .. 8 j + 7 are the train
// rows 16 j + 4 (lane / 32) + {0 .. 3, 8 .. 11} of the tile: only the best key of each such group of eight
// enters the top-2 update, 11 instructions instead of 24.  The running pair is then exact except for one
// thing: the rows that share a group with the best row never competed for second place.  (Any other row's
// group-mates are beaten by that row, so they cannot be second without it being first.)  Those seven rows
// are compared once per query at the end of knn2_mfma_wave.
__device__ __forceinline__ void knn2_select_tile(float& m1, float& m2, const v16f& acc, float pinf)
{
    float q[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float r = __builtin_fmaxf(__builtin_fmaxf(acc[8 * j], acc[8 * j + 1]), acc[8 * j + 2]);      // v_max3_f32
        r = __builtin_fmaxf(__builtin_fmaxf(r, acc[8 * j + 3]), acc[8 * j + 4]);
        r = __builtin_fmaxf(__builtin_fmaxf(r, acc[8 * j + 5]), acc[8 * j + 6]);
        q[j] = __builtin_amdgcn_fmed3f(r, acc[8 * j + 7], pinf);                                     // max as med3(x, y, +inf)
    }
    top2_insert2_f32(m1, m2, q[0], q[1], pinf);
}

// Float key after `tiles` tiles of a run that started at train row row0 -> integer key.
__device__ __forceinline__ uint32_t knn2_decode_key(float f, int tiles, int row0)
{
    if (f == -INFINITY) return KEY_NONE;
    const int w = (int)(f * -4096.0f) + 8192;       // 8192 * distance + (index - 32 * tiles + 8192)
    const int d = w >> 13;
    const int i = (w & 8191) - 8192 + 32 * tiles + row0;
    return ((uint32_t)d << 18) | (uint32_t)i;
}

constexpr int KNN2_RUN_TILES = 256;   // 8192 rows: the index field of a float key must stay below 2.0
constexpr int KNN2_GROUP = 4;         // train tiles staged per workgroup barrier

// One wave's share of a workgroup: QT query tiles (32 queries each) -- tiles qtile_first,
// qtile_first + 4, ... (the four waves of a workgroup take the query tiles of their block round-robin)
// -- against the train tiles tbeg .. tend, which stream through a double-buffered 4 KiB LDS tile
// shared by the 4 waves.  A wave without queries (has_queries == false, QT == 1) only helps staging.
template <int QT, bool PARTIAL>
__device__ __forceinline__ void knn2_mfma_wave(uint4 (*stage)[256], const uint32_t* __restrict__ qrows,
                                               const uint32_t* __restrict__ trows, int nq, int nt,
                                               int qtile_first, bool wave_has_queries, int splits, int zsplit,
                                               int pair, int cap, uint32_t* __restrict__ partial,
                                               int32_t* __restrict__ idx, int32_t* __restrict__ dist,
                                               uint8_t* __restrict__ good)
{
    const int lane = lane_id();
    const int h = lane >> 5;

    // query operands: lane (h, r) of k-step ks holds dword 2 ks + h of query row r as 32 FP4 values
    v8i b[QT][4];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const int qrow = (qtile_first + 4 * u) * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint32_t w = qrow < nq ? qrows[(size_t)qrow * 8 + 2 * ks + h] : 0u;
            const uint4 v = fp4_of_bits(w);
            b[u][ks] = v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
        }
    }

    v16f cin;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        cin[i] = -256.0f - (float)((i & 3) + 8 * (i >> 2) + 4 * h) * (1.0f / 4096.0f);

    const int tiles_total = (nt + 31) >> 5;
    const int tiles_per_split = (tiles_total + splits - 1) / splits;
    const int tbeg = zsplit * tiles_per_split;
    int tend = tbeg + tiles_per_split; tend = tend < tiles_total ? tend : tiles_total;

    uint32_t k1[QT], k2[QT];
    float m1[QT], m2[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) { k1[u] = KEY_NONE; k2[u] = KEY_NONE; m1[u] = -INFINITY; m2[u] = -INFINITY; }

    const int scale_one = 0x7F;   // E8M0 exponent 127 = 2^0
    float pinf = INFINITY;
    asm volatile("" : "+v"(pinf));   // keep +inf in a register (not a literal operand)
    // staging: thread (piece = t / 32, row = t % 32) fetches dword `piece` of the tile's row `row` and
    // writes it to LDS as 32 FP4 values, which is the 16-byte operand of lane (piece & 1, row) at
    // k-step piece / 2
    // (a 32-bit byte offset from the pair's first train row: scalar base + vector offset addressing)
    const int srow = threadIdx.x & 31, spiece = threadIdx.x >> 5;
    const uint32_t soff = (uint32_t)(srow * 32 + spiece * 4);
    const int tfull_all = nt >> 5;                // tiles below this one are whole: no row check needed
    auto fetch = [&](int t) -> uint32_t {
        const uint32_t off = soff + (uint32_t)t * 1024u;
        if (t < tfull_all) return load_u32_at(reinterpret_cast<const uint8_t*>(trows), off);
        return t * 32 + srow < nt ? load_u32_at(reinterpret_cast<const uint8_t*>(trows), off) : 0u;
    };
    int run_row0 = tbeg * 32, run_tiles = 0;
    // One staged train tile (waves with queries): multiply and select.  c0 = accumulator start.
    // The query tiles go through the matrix pipe two at a time, their two chains of four dependent
    // instructions interleaved: a matrix instruction that has to wait for its predecessor's result holds
    // up the issue of everything behind it, and with an independent instruction in between it never
    // waits.  (The same kernel WITHOUT any selection work between the dependent instructions of single
    // chains ran 20 % slower than with it.)  The selection of a pair's results follows its chains; the
    // other waves of the SIMD keep the matrix pipe busy meanwhile.
    auto compute = [&](int t, const uint4* tile, const v16f& c0) {
        v8i a[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 v = tile[ks * 64 + lane];
            a[ks] = v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u + 1 < QT; u += 2) {
            v16f x = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b[u][0], c0, 4, 4, 0, scale_one, 0, scale_one);
            v16f y = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b[u + 1][0], c0, 4, 4, 0, scale_one, 0, scale_one);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) {
                x = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[ks], b[u][ks], x, 4, 4, 0, scale_one, 0, scale_one);
                y = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[ks], b[u + 1][ks], y, 4, 4, 0, scale_one, 0, scale_one);
            }
            knn2_select_tile(m1[u], m2[u], x, pinf);
            knn2_select_tile(m1[u + 1], m2[u + 1], y, pinf);
            m1[u] += 32.0f / 4096.0f;
            m2[u] += 32.0f / 4096.0f;
            m1[u + 1] += 32.0f / 4096.0f;
            m2[u + 1] += 32.0f / 4096.0f;
        }
        if (QT & 1) {
            constexpr int u = QT - 1;
            v16f x = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b[u][0], c0, 4, 4, 0, scale_one, 0, scale_one);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks)
                x = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[ks], b[u][ks], x, 4, 4, 0, scale_one, 0, scale_one);
            knn2_select_tile(m1[u], m2[u], x, pinf);
            m1[u] += 32.0f / 4096.0f;
            m2[u] += 32.0f / 4096.0f;
        }
        ++run_tiles;
        if (run_tiles == KNN2_RUN_TILES || t + 1 == tend) {       // end of a key run
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                top2_insert(k1[u], k2[u], knn2_decode_key(m1[u], run_tiles, run_row0));
                top2_insert(k1[u], k2[u], knn2_decode_key(m2[u], run_tiles, run_row0));
                m1[u] = -INFINITY;
                m2[u] = -INFINITY;
            }
            run_row0 += run_tiles * 32;
            run_tiles = 0;
        }
    };
    // The train tiles are staged KNN2_GROUP at a time, so that the four waves meet at one barrier per
    // group instead of one per tile; two groups of buffers alternate.  Whole tiles first; only the very
    // last tile of the train set can be partial (rows past nt never win).
    int tfull = nt >> 5;
    tfull = tfull < tend ? tfull : tend;
    uint32_t next[KNN2_GROUP];
#pragma unroll
    for (int j = 0; j < KNN2_GROUP; ++j) next[j] = tbeg + j < tfull ? fetch(tbeg + j) : 0u;
    int grp = 0, t = tbeg;
    for (; t < tfull; t += KNN2_GROUP, grp ^= 1) {
        uint4* const bufs = stage[grp * KNN2_GROUP];
#pragma unroll
        for (int j = 0; j < KNN2_GROUP; ++j) bufs[j * 256 + threadIdx.x] = fp4_of_bits(next[j]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KNN2_GROUP; ++j) next[j] = t + KNN2_GROUP + j < tfull ? fetch(t + KNN2_GROUP + j) : 0u;
        if (wave_has_queries) {
            int g = tfull - t;
            g = g < KNN2_GROUP ? g : KNN2_GROUP;
#pragma unroll 1
            for (int j = 0; j < g; ++j) compute(t + j, bufs + j * 256, cin);
        }
    }
    const int tlast = tbeg > tfull ? tbeg : tfull;    // (an empty range may lie past the last whole tile)
    if (tlast < tend) {                               // the partial tile, alone behind a barrier of its own
        uint4* const bufs = stage[grp * KNN2_GROUP];
        bufs[threadIdx.x] = fp4_of_bits(fetch(tlast));
        __syncthreads();
        if (wave_has_queries) {
            v16f c0 = cin;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (tlast * 32 + (i & 3) + 8 * (i >> 2) + 4 * h >= nt) c0[i] = -INFINITY;
            compute(tlast, bufs, c0);
        }
    }

    if (!wave_has_queries) return;
    // Lanes l and l + 32 hold disjoint train rows of the same query: merge them.
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const uint32_t p1 = (uint32_t)__shfl_xor((int)k1[u], 32);
        const uint32_t p2 = (uint32_t)__shfl_xor((int)k2[u], 32);
        top2_insert(k1[u], k2[u], p1);
        top2_insert(k1[u], k2[u], p2);
    }
    // The rows that shared a group of eight with the best row (see knn2_select_tile) now compete for second
    // place, by XOR and popcount: the group is best's tile + (best & 20) + {0 .. 3, 8 .. 11}; counting on from
    // best's own place in it, lane l takes the 1st, 3rd, 5th and 7th, lane l + 32 the 2nd, 4th and 6th.  The
    // loads of two query tiles go out together.
    constexpr int UB = QT >= 2 ? 2 : 1;
#pragma unroll
    for (int u0 = 0; u0 < QT; u0 += UB) {
        uint4 rq[UB][2], rt[UB][4][2];
        int rr[UB][4];
#pragma unroll
        for (int v = 0; v < UB; ++v) {
            const int u = u0 + v;
            if (u >= QT) break;
            const int qi = (qtile_first + 4 * u) * 32 + (lane & 31);
            const bool have = k1[u] != KEY_NONE && qi < nq;
            const int best = (int)(k1[u] & 0x3FFFFu);
            const int base = best & ~11, pos = ((best >> 1) & 4) | (best & 3);      // place 0 .. 7 within the group
            const uint4* qp = reinterpret_cast<const uint4*>(qrows + (size_t)(have ? qi : 0) * 8);
            rq[v][0] = qp[0]; rq[v][1] = qp[1];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int place = (pos + 1 + 2 * k + h) & 7;
                const int row = base + ((place & 4) << 1) + (place & 3);
                rr[v][k] = have && place != pos && row < nt ? row : -1;
                const uint4* tp = reinterpret_cast<const uint4*>(trows + (size_t)(rr[v][k] >= 0 ? rr[v][k] : 0) * 8);
                rt[v][k][0] = tp[0]; rt[v][k][1] = tp[1];
            }
        }
#pragma unroll
        for (int v = 0; v < UB; ++v) {
            const int u = u0 + v;
            if (u >= QT) break;
            uint32_t kk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t d = hamming256_regs(rq[v][0], rq[v][1], rt[v][k][0], rt[v][k][1]);
                kk[k] = rr[v][k] >= 0 ? (d << 18) | (uint32_t)rr[v][k] : KEY_NONE;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t other = (uint32_t)__shfl_xor((int)kk[k], 32);
                top2_insert(k1[u], k2[u], kk[k]);
                top2_insert(k1[u], k2[u], other);
            }
            const int qi = (qtile_first + 4 * u) * 32 + (lane & 31);
            if (h == 0 && qi < nq) {
                if (PARTIAL) {
                    const size_t o = (((size_t)pair * splits + zsplit) * cap + qi);
                    reinterpret_cast<uint2*>(partial)[o] = make_uint2(k1[u], k2[u]);
                } else {
                    knn2_store(k1[u], k2[u], idx, dist, good, (size_t)pair * cap + qi);
                }
            }
        }
    }
}

// Up to QTMAX query tiles per wave, 4 waves per workgroup: a workgroup owns a block of 128 * QTMAX
// queries.  More tiles per wave mean more matrix work per staged train tile and per barrier
// (QTMAX = 4 against 2: 2.12 against 2.22 ms per 1000 1080p frames on MI355X, at 167 registers and
// 3 waves per SIMD), but a coarser block: the last block of a pair is mostly empty.  So the query
// tiles of a block are dealt to the waves round-robin and every wave runs the instantiation for the
// number of tiles it really has; the waves of the last block then all work, each on fewer tiles,
// instead of one wave working on four.  (All waves pass the same barriers: the train loop is the same
// in every instantiation.)
// Three waves per SIMD (<= 168 registers per lane); the compiler keeps the
// accumulators in ordinary VGPRs, which the top-2 instructions can read directly.
// One workgroup's block of queries: `logical` numbers the (pair, train split, query block) triples, query block
// fastest.
template <int QTMAX, bool PARTIAL>
__device__ __forceinline__ void knn2_mfma_body(uint4 (*stage)[256], uint32_t logical, const uint8_t* __restrict__ qdesc,
                                               const uint8_t* __restrict__ tdesc,
                                               const int32_t* __restrict__ qcounts,
                                               const int32_t* __restrict__ tcounts,
                                               const int32_t* __restrict__ pairs, int qframe0,
                                               int tframe0, int nframes, int cap, int nq_fixed, int nt_fixed,
                                               int splits, int nbp, uint32_t* __restrict__ partial,
                                               int32_t* __restrict__ idx, int32_t* __restrict__ dist,
                                               uint8_t* __restrict__ good)
{
    int pair, zsplit, qblk;
    {
        const uint32_t per_pair = (uint32_t)nbp * (uint32_t)splits;
        pair = (int)(logical / per_pair);
        const uint32_t rem = logical - (uint32_t)pair * per_pair;
        zsplit = (int)(rem / (uint32_t)nbp);
        qblk = (int)(rem - (uint32_t)zsplit * (uint32_t)nbp);
    }
    int nq = nq_fixed, nt = nt_fixed, qf = 0, tf = 0;
    if (qcounts) {
        if (pairs) {
            qf = pairs[2 * pair]; tf = pairs[2 * pair + 1];
            // a pair that names a frame outside the batch is skipped (uniform over the workgroup)
            if ((unsigned)qf >= (unsigned)nframes || (unsigned)tf >= (unsigned)nframes) return;
        } else { qf = pair + 1 + qframe0; tf = pair + tframe0; }
        nq = qcounts[2 * qf + 1]; nq = nq < cap ? nq : cap;
        nt = tcounts[2 * tf + 1]; nt = nt < cap ? nt : cap;
    }
    if (qblk * (128 * QTMAX) >= nq) return;           // uniform over the workgroup

    const int wave = wave_id();
    // descriptor rows of this pair, 8 dwords each (rows past the counts are never read)
    const uint32_t* __restrict__ qrows = reinterpret_cast<const uint32_t*>(qdesc) + (size_t)qf * cap * 8;
    const uint32_t* __restrict__ trows = reinterpret_cast<const uint32_t*>(tdesc) + (size_t)tf * cap * 8;
    // this wave's query tiles: qtile_first + 4 u for u < nu
    const int qtile_first = qblk * (4 * QTMAX) + wave;
    const int qtiles = (nq + 31) >> 5;
    int nu = (qtiles - qtile_first + 3) >> 2;
    nu = nu < 0 ? 0 : (nu > QTMAX ? QTMAX : nu);
#define ORB_KNN2_WAVE(N, HAS)                                                                                  \
    knn2_mfma_wave<N, PARTIAL>(stage, qrows, trows, nq, nt, qtile_first, HAS, splits, zsplit, pair, cap, partial, \
                               idx, dist, good)
    static_assert(QTMAX == 4 || QTMAX == 2, "instantiations below");
    if (nu == QTMAX) ORB_KNN2_WAVE(QTMAX, true);
    else if (nu == 0) ORB_KNN2_WAVE(1, false);
    else if (nu == 1) ORB_KNN2_WAVE(1, true);
    else if constexpr (QTMAX == 4) {
        if (nu == 2) ORB_KNN2_WAVE(2, true);
        else ORB_KNN2_WAVE(3, true);
    }
#undef ORB_KNN2_WAVE
}

template <int QTMAX, bool PARTIAL>
__global__ __launch_bounds__(256, 3) void knn2_mfma_kernel(const uint8_t* __restrict__ qdesc,
                                                        const uint8_t* __restrict__ tdesc,
                                                        const int32_t* __restrict__ qcounts,
                                                        const int32_t* __restrict__ tcounts,
                                                        const int32_t* __restrict__ pairs, int qframe0,
                                                        int tframe0, int nframes, int cap, int nq_fixed, int nt_fixed,
                                                        int splits, int nbp, uint32_t* __restrict__ partial,
                                                        int32_t* __restrict__ idx, int32_t* __restrict__ dist,
                                                        uint8_t* __restrict__ good)
{
    __shared__ uint4 stage[2 * KNN2_GROUP][256];
    knn2_mfma_body<QTMAX, PARTIAL>(stage, xcd_contiguous_item(), qdesc, tdesc, qcounts, tcounts, pairs, qframe0, tframe0,
                                   nframes, cap, nq_fixed, nt_fixed, splits, nbp, partial, idx, dist, good);
}

// ---------------------------------------------------------------------------------------------
// MEASUREMENT KERNEL: orientation + descriptors of one frame batch and Hamming top-2 of ANOTHER batch's
// descriptors in one launch, with role-specialised workgroups, so that workgroups of both kinds are resident on
// every compute unit at the same time by construction (workgroups are dispatched in index order; groups of 8
// consecutive indices = one per XCD alternate between the roles at the ratio of the two item counts).  It
// exists to answer one question with counters: do the vector-bound extraction and the matrix-bound matcher
// gain from sharing SIMDs?  Same device code as the two kernels above, same results.
// ---------------------------------------------------------------------------------------------
constexpr size_t DM_LDS_BYTES = sizeof(DescribeTileLds) > sizeof(uint4) * 2 * KNN2_GROUP * 256
                                    ? sizeof(DescribeTileLds) : sizeof(uint4) * 2 * KNN2_GROUP * 256;

template <int QTMAX>
__global__ __launch_bounds__(256, QTMAX == 4 ? 3 : 4) void describe_match_kernel(
    // describe role
    const uint8_t* __restrict__ gray, int H, int W, const int2* __restrict__ kpxy,
    const int32_t* __restrict__ kpslot, const int32_t* __restrict__ counts, int cap,
    const int2* __restrict__ rowoff, float* __restrict__ kp7, uint8_t* __restrict__ desc,
    int32_t* __restrict__ valid_idx, int ntc, int nbands, uint32_t n_describe,
    // matcher role (consecutive pairs of another feature buffer)
    const uint8_t* __restrict__ mdesc, const int32_t* __restrict__ mcounts, int mframes, int mcap, int nbp,
    int32_t* __restrict__ idx, int32_t* __restrict__ dist, uint8_t* __restrict__ good, uint32_t n_match,
    uint32_t every)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds_raw[DM_LDS_BYTES];
    const uint32_t g = blockIdx.x >> 3, x = blockIdx.x & 7u;
    const uint32_t mgroups = (n_match + 7u) >> 3, dgroups = (n_describe + 7u) >> 3;
    const bool mrole = (g % every == 0u) && (g / every < mgroups);
    if (mrole) {
        const uint32_t item = x * mgroups + g / every;
        if (item >= n_match) return;
        knn2_mfma_body<QTMAX, false>(reinterpret_cast<uint4(*)[256]>(lds_raw), item, mdesc, mdesc, mcounts, mcounts,
                                     nullptr, 0, 0, mframes, mcap, 0, 0, 1, nbp, nullptr, idx, dist, good);
    } else {
        uint32_t mbefore = (g + every - 1u) / every;
        mbefore = mbefore < mgroups ? mbefore : mgroups;
        const uint32_t item = x * dgroups + (g - mbefore);
        if (item >= n_describe) return;
        describe_tile_body<false>(*reinterpret_cast<DescribeTileLds*>(lds_raw), item, gray, H, W, kpxy, kpslot, counts,
                                  cap, rowoff, kp7, desc, valid_idx, ntc, nbands, nullptr);
    }
}

void launch_describe_match(const FrameSet& fs, const int2* kpxy, const int32_t* kpslot, const int32_t* counts, int cap,
                           const int2* rowoff, float* kp7, uint8_t* desc, int32_t* valid_idx, const uint8_t* mdesc,
                           const int32_t* mcounts, int mframes, int mcap, int32_t* idx, int32_t* dist, uint8_t* good,
                           int qt, hipStream_t s)
{
    const int ntc = (fs.W + DT_W - 1) / DT_W, nbands = (fs.H + DT_H - 1) / DT_H;
    const uint32_t n_describe = (uint32_t)((size_t)ntc * nbands * fs.frames);
    const int nbp = (mcap + 128 * qt - 1) / (128 * qt);
    const uint32_t n_match = (uint32_t)((size_t)nbp * (mframes - 1));
    const uint32_t mgroups = (n_match + 7u) >> 3, dgroups = (n_describe + 7u) >> 3;
    const uint32_t every = mgroups ? (mgroups + dgroups) / mgroups : 1u;
    const dim3 grid(8u * (mgroups + dgroups));
    if (qt == 4)
        hipLaunchKernelGGL(describe_match_kernel<4>, grid, dim3(256), 0, s, fs.gray, fs.H, fs.W, kpxy, kpslot, counts, cap,
                           rowoff, kp7, desc, valid_idx, ntc, nbands, n_describe, mdesc, mcounts, mframes, mcap, nbp, idx,
                           dist, good, n_match, every);
    else
        hipLaunchKernelGGL(describe_match_kernel<2>, grid, dim3(256), 0, s, fs.gray, fs.H, fs.W, kpxy, kpslot, counts, cap,
                           rowoff, kp7, desc, valid_idx, ntc, nbands, n_describe, mdesc, mcounts, mframes, mcap, nbp, idx,
                           dist, good, n_match, every);
}

__global__ __launch_bounds__(256) void knn2_merge_kernel(const int32_t* __restrict__ qcounts,
                                                         const int32_t* __restrict__ pairs, int qframe0,
                                                         int nframes, int cap, int nq_fixed, int splits,
                                                         const uint32_t* __restrict__ partial,
                                                         int32_t* __restrict__ idx,
                                                         int32_t* __restrict__ dist,
                                                         uint8_t* __restrict__ good)
{
    const int pair = blockIdx.y;
    int nq = nq_fixed;
    if (qcounts) {
        const int qf = pairs ? pairs[2 * pair] : pair + 1 + qframe0;
        if (pairs && ((unsigned)qf >= (unsigned)nframes || (unsigned)pairs[2 * pair + 1] >= (unsigned)nframes)) return;
        nq = qcounts[2 * qf + 1]; nq = nq < cap ? nq : cap;
    }
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    uint32_t m1 = KEY_NONE, m2 = KEY_NONE;
    for (int z = 0; z < splits; ++z) {
        const uint2 k = reinterpret_cast<const uint2*>(partial)[((size_t)pair * splits + z) * cap + qi];
        if (k.x != KEY_NONE) top2_insert(m1, m2, k.x);
        if (k.y != KEY_NONE) top2_insert(m1, m2, k.y);
    }
    knn2_store(m1, m2, idx, dist, good, (size_t)pair * cap + qi);
}

// Lowe ratio filter with ordered compaction (reference views.py:163-171): query i is kept iff
// dist0 < 0.75 * dist1 (4*d0 < 3*d1 in integers); kept rows keep query order.  One workgroup
// walks the queries in chunks of 1024.  A query with a missing neighbour makes count[0] = -1.
__global__ __launch_bounds__(1024) void ratio_compact_kernel(const int32_t* __restrict__ idx,
                                                             const int32_t* __restrict__ dist, int nq,
                                                             int32_t* __restrict__ good_q,
                                                             int32_t* __restrict__ good_t2,
                                                             int32_t* __restrict__ good_t1,
                                                             int32_t* __restrict__ count)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    __shared__ int bad;
    if (threadIdx.x == 0) { carry = 0; bad = 0; }
    __syncthreads();
    const int lane = lane_id(), wave = wave_id();
    for (int i0 = 0; i0 < nq; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        int keep = 0;
        int2 id = make_int2(0, 0);
        if (i < nq) {
            id = reinterpret_cast<const int2*>(idx)[i];
            const int2 d = reinterpret_cast<const int2*>(dist)[i];
            if (id.x < 0 || id.y < 0) bad = 1;
            else keep = 4 * d.x < 3 * d.y;
        }
        const int ik = wave_incl_scan(keep, lane);
        if (lane == 63) wsum[wave] = ik;
        __syncthreads();
        int base = carry;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        if (keep) {
            const int o = base + ik - 1;
            good_q[o] = i;
            good_t2[o] = id.y;
            good_t1[o] = id.x;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + ik;
        __syncthreads();
    }
    if (threadIdx.x == 0) count[0] = bad ? -1 : carry;
}

// The same compaction for every pair of a batch: one workgroup per pair, flags from the matcher.
__global__ __launch_bounds__(1024) void ratio_compact_batch_kernel(const int32_t* __restrict__ idx,
                                                                   const uint8_t* __restrict__ good,
                                                                   const int32_t* __restrict__ qcounts,
                                                                   const int32_t* __restrict__ pairs, int nframes,
                                                                   int cap, int32_t* __restrict__ good_q,
                                                                   int32_t* __restrict__ good_t2,
                                                                   int32_t* __restrict__ good_t1,
                                                                   int32_t* __restrict__ kept)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    __shared__ int bad;
    const int pair = blockIdx.x;
    const int qf = pairs ? pairs[2 * pair] : pair + 1;
    if (pairs && ((unsigned)qf >= (unsigned)nframes || (unsigned)pairs[2 * pair + 1] >= (unsigned)nframes)) {
        if (threadIdx.x == 0) kept[pair] = -1;     // a pair that names a frame outside the batch
        return;
    }
    int nq = qcounts[2 * qf + 1];
    nq = nq < cap ? nq : cap;
    if (threadIdx.x == 0) { carry = 0; bad = 0; }
    __syncthreads();
    const int lane = lane_id(), wave = wave_id();
    const size_t row0 = (size_t)pair * cap;
    for (int i0 = 0; i0 < nq; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        int keep = 0;
        int2 id = make_int2(0, 0);
        if (i < nq) {
            id = reinterpret_cast<const int2*>(idx)[row0 + i];
            if (id.x < 0 || id.y < 0) bad = 1;
            else keep = good[row0 + i] != 0;
        }
        const int ik = wave_incl_scan(keep, lane);
        if (lane == 63) wsum[wave] = ik;
        __syncthreads();
        int base = carry;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        if (keep) {
            const size_t o = row0 + base + ik - 1;
            good_q[o] = i;
            good_t2[o] = id.y;
            good_t1[o] = id.x;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + ik;
        __syncthreads();
    }
    if (threadIdx.x == 0) kept[pair] = bad ? -1 : carry;
}

void launch_ratio_compact_batch(const int32_t* idx, const uint8_t* good, const int32_t* qcounts,
                                const int32_t* pairs, int npairs, int nframes, int cap, int32_t* good_q,
                                int32_t* good_t2, int32_t* good_t1, int32_t* kept, hipStream_t s)
{
    hipLaunchKernelGGL(ratio_compact_batch_kernel, dim3(npairs), dim3(1024), 0, s, idx, good, qcounts, pairs, nframes, cap,
                       good_q, good_t2, good_t1, kept);
}

void launch_ratio_compact(const int32_t* idx, const int32_t* dist, int nq, int32_t* good_q,
                          int32_t* good_t2, int32_t* good_t1, int32_t* count, hipStream_t s)
{
    hipLaunchKernelGGL(ratio_compact_kernel, dim3(1), dim3(1024), 0, s, idx, dist, nq, good_q, good_t2,
                       good_t1, count);
}

template <int QPL, int RPT, int BS>
static void launch_knn2_variant(const Knn2Args& a, hipStream_t s)
{
    const int nqmax = a.qcounts ? a.cap : a.nq_fixed;
    const int nbp = ((nqmax + BS * QPL - 1) / (BS * QPL)) | 1;   // odd: see the remap in the kernel
    dim3 grid((unsigned)((size_t)nbp * a.npairs * a.splits));
    if (a.splits > 1)
        hipLaunchKernelGGL((knn2_kernel<QPL, RPT, BS, true>), grid, dim3(BS), 0, s, a.qdesc, a.tdesc, a.qcounts,
                           a.tcounts, a.pairs, a.qframe0, a.tframe0, a.nframes, a.cap, a.nq_fixed, a.nt_fixed, a.splits, nbp,
                           a.partial, a.idx, a.dist, a.good);
    else
        hipLaunchKernelGGL((knn2_kernel<QPL, RPT, BS, false>), grid, dim3(BS), 0, s, a.qdesc, a.tdesc, a.qcounts,
                           a.tcounts, a.pairs, a.qframe0, a.tframe0, a.nframes, a.cap, a.nq_fixed, a.nt_fixed, a.splits, nbp,
                           a.partial, a.idx, a.dist, a.good);
}

void launch_knn2(const Knn2Args& a, hipStream_t s)
{
    // Measured on MI355X (1080p batch, ~4.7k x 4.7k descriptors per pair): one query per lane, four
    // train rows per trip and single-wave workgroups was the fastest of the tilings tried
    // (more queries per lane or 256-thread workgroups were 1-40 % slower).
    launch_knn2_variant<1, 4, 64>(a, s);
}

void launch_knn2_mfma(const Knn2Args& a, hipStream_t s)
{
    constexpr int QT = 4;
    const int nqmax = a.qcounts ? a.cap : a.nq_fixed;
    const int nbp = ((nqmax + 128 * QT - 1) / (128 * QT)) | 1;   // odd: see xcd_contiguous_item()
    dim3 grid((unsigned)((size_t)nbp * a.npairs * a.splits));
    static const unsigned pad = lds_pad_from_env("ORBAMD_PAD_LDS_KNN2", (const void*)knn2_mfma_kernel<QT, false>);
    if (a.splits > 1)
        hipLaunchKernelGGL((knn2_mfma_kernel<QT, true>), grid, dim3(256), 0, s, a.qdesc, a.tdesc,
                           a.qcounts, a.tcounts, a.pairs, a.qframe0, a.tframe0, a.nframes,
                           a.cap, a.nq_fixed, a.nt_fixed, a.splits, nbp, a.partial, a.idx, a.dist, a.good);
    else
        hipLaunchKernelGGL((knn2_mfma_kernel<QT, false>), grid, dim3(256), pad, s, a.qdesc, a.tdesc,
                           a.qcounts, a.tcounts, a.pairs, a.qframe0, a.tframe0, a.nframes,
                           a.cap, a.nq_fixed, a.nt_fixed, a.splits, nbp, a.partial, a.idx, a.dist, a.good);
}

void launch_knn2_merge(const Knn2Args& a, hipStream_t s)
{
    const int nqmax = a.qcounts ? a.cap : a.nq_fixed;
    dim3 grid((nqmax + 255) / 256, a.npairs);
    hipLaunchKernelGGL(knn2_merge_kernel, grid, dim3(256), 0, s, a.qcounts, a.pairs, a.qframe0, a.nframes, a.cap,
                       a.nq_fixed, a.splits, a.partial, a.idx, a.dist, a.good);
}

}  // namespace orbamd
