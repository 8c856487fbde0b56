// orb_internal.h -- declarations shared by the kernel file and the C-ABI file of liborbamd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbamd {

// Mask geometry: one bit per pixel, bit (x & 31) of 32-bit word (x >> 5); rows padded to whole words.
static inline int mask_words(int W) { return (W + 31) >> 5; }

// Frame-strided views used by the batch kernels (all device pointers).
struct FrameSet {
    const uint8_t* gray;   // [frames][H][W]
    int frames, H, W;
};

struct ExtractBuffers {
    uint32_t* mask;     // [frames][H][MW]
    int2* rowcnt;       // [frames][H]   (all, valid) keypoints in the row
    int2* rowoff;       // [frames][H]   exclusive prefix of rowcnt within the frame
    int2* kpxy;         // [frames][cap] (x, y)
    int32_t* kpslot;    // [frames][cap] descriptor row of the keypoint, -1 if it has none
};

// Kernel launchers (enqueue on `s`, no synchronisation, no allocation).
void launch_bgr2gray(const uint8_t* bgr, uint8_t* gray, long npix, hipStream_t s);
// nbytes from src to dst by a kernel; both 16-byte aligned, either may be pinned host memory.
void launch_copy_bytes(const void* src, void* dst, size_t nbytes, hipStream_t s);
// Up to 6 dword segments copied by one kernel (results -> pinned host memory): segment k moves
// min(max_dwords, count ? max(count[0], 0) * count_mul : max_dwords) dwords, count read on the device.
struct EgressSeg {
    const uint32_t* src;
    uint32_t* dst;
    const int32_t* count;
    int count_mul;
    int max_dwords;
};
struct EgressArgs {
    EgressSeg seg[6];
    int nseg;
};
void launch_egress(const EgressArgs& a, hipStream_t s);
void launch_synth(uint8_t* gray, int frames, int H, int W, int first_frame, uint32_t seed, int cell,
                  int density_thr, hipStream_t s);
// Also accumulates rowcnt[frame][y] = (corners, corners with a descriptor); rowcnt must be zero.
void launch_fast_mask(const FrameSet& fs, uint32_t* mask, int2* rowcnt, hipStream_t s);
// Detector extension (arc length, wrap, threshold, score + 3x3 non-maximum suppression): fills mask and
// rowcnt like launch_fast_mask.  rowcnt must be zero.
void launch_fast_ext(const FrameSet& fs, int thr, int arc, int wrap, int nms, uint32_t* mask, int2* rowcnt,
                     hipStream_t s);
void launch_rowscan(const FrameSet& fs, const int2* rowcnt, int2* rowoff, int32_t* counts, hipStream_t s);
// counts [frames][2] as written by launch_rowscan; a frame with more than cap keypoints gets its
// second entry replaced by the number of descriptor rows that belong to its first cap keypoints.
void launch_emit(const FrameSet& fs, const uint32_t* mask, const int2* rowcnt, const int2* rowoff,
                 int cap, int2* kpxy, int32_t* kpslot, int32_t* counts, hipStream_t s);
// Orientation (+ keypoint rows) and descriptors for keypoints listed in kpxy/kpslot.
//   counts  [frames][2] device; keypoints processed per frame = min(counts[f][0], cap)
//   kp7     [frames][cap][7] or NULL;  angle_out [frames][cap] or NULL
//   angle_in: when non-NULL the orientation step is skipped and these angles are used
//   desc    [frames][cap][32] or NULL (then no descriptors are computed)
//   risk    [frames][4] int32 or NULL: parity-audit counters that the kernel ADDS to (the caller zeroes them):
//           [0] keypoints whose angle, [1] sample coordinates whose value is within 4 units in the last place
//           of the midpoint of two floats before the reference narrows it to float, [2] those of [1] whose two
//           candidate floats also round to different pixels.  NULL runs the kernels without the audit code.
void launch_orient_describe(const FrameSet& fs, const int2* kpxy, const int32_t* kpslot,
                            const int32_t* counts, int cap, const float* angle_in, float* kp7,
                            float* angle_out, uint8_t* desc, int32_t* valid_idx, int32_t* risk, hipStream_t s);

// The same results for whole frames whose keypoint lists are in scan order (as launch_emit leaves them),
// computed tile by tile from LDS: rowoff [frames][H] as written by launch_rowscan.  Needs rows of whole
// aligned 16-byte units (W % 16 == 0, 16-byte-aligned frames); kp7 and desc must not be NULL.
void launch_describe_tiles(const FrameSet& fs, const int2* kpxy, const int32_t* kpslot, const int32_t* counts,
                           int cap, const int2* rowoff, float* kp7, uint8_t* desc, int32_t* valid_idx, int32_t* risk,
                           hipStream_t s);

struct Knn2Args {
    const uint8_t* qdesc;     // query descriptor base
    const uint8_t* tdesc;     // train descriptor base
    const int32_t* qcounts;   // [frames][2] of the query buffer, or NULL (then *_fixed sizes are used)
    const int32_t* tcounts;   // [frames][2] of the train buffer (== qcounts for one batch)
    const int32_t* pairs;     // [npairs][2] (query frame, train frame) or NULL = consecutive (p+1, p)
    int qframe0, tframe0;     // added to the pair's frame numbers (single cross pair: pairs == NULL, npairs == 1)
    int npairs;
    int nframes;              // frames in the buffers: pairs[] entries outside [0, nframes) are skipped
    int cap;                  // rows per frame in qdesc/tdesc and per pair in the outputs
    int nq_fixed, nt_fixed;   // used when qcounts == NULL
    int splits;               // train-axis splits (>= 1)
    uint32_t* partial;        // [npairs][splits][cap][2] packed keys when splits > 1
    int32_t* idx;             // [npairs][cap][2]
    int32_t* dist;            // [npairs][cap][2]
    uint8_t* good;            // [npairs][cap] or NULL
};
// Descriptor rows (rank among keypoints >= 16 px from every edge) for a caller-supplied list;
// counts[0] = n, counts[1] = number of descriptor rows.
// Test hook: orb_trig.h on the device, element-wise.
void launch_trig_selftest(const int32_t* m01, const int32_t* m10, int n, double* deg, uint8_t* unsure,
                          const double* rad, int m, double* sn, double* cs, hipStream_t s);
void launch_slots(const int2* kpxy, int n, int H, int W, int32_t* kpslot, int32_t* counts, hipStream_t s);
// Ordered compaction of the ratio test over one query set; count[0] = kept rows, or -1 when a
// query has fewer than two neighbours.
void launch_ratio_compact(const int32_t* idx, const int32_t* dist, int nq, int32_t* good_q,
                          int32_t* good_t2, int32_t* good_t1, int32_t* count, hipStream_t s);
// Per pair of a batch: ordered lists of the queries whose flag in good[pair][*] is set; kept[pair] =
// their number, or -1 when a query of the pair has fewer than two neighbours.
void launch_ratio_compact_batch(const int32_t* idx, const uint8_t* good, const int32_t* qcounts,
                                const int32_t* pairs, int npairs, int nframes, int cap, int32_t* good_q,
                                int32_t* good_t2, int32_t* good_t1, int32_t* kept, hipStream_t s);
void launch_knn2(const Knn2Args& a, hipStream_t s);
// FP4 image of descriptor rows: slot s takes frame (pairs ? pairs[2 s + side] : frame0 + s) of
// `desc` (rows per frame = cap, row counts from counts[2 f + 1]; or n_fixed rows when counts == NULL)
void launch_knn2_mfma(const Knn2Args& a, hipStream_t s);
void launch_knn2_merge(const Knn2Args& a, hipStream_t s);
// Measurement launch: launch_describe_tiles for one batch and launch_knn2_mfma over the consecutive pairs of
// another feature buffer (mdesc [mframes][mcap][32], mcounts [mframes][2]) as ONE kernel with role-specialised
// workgroups; qt = query tiles per wave of the matcher role (4 or 2).
void launch_describe_match(const FrameSet& fs, const int2* kpxy, const int32_t* kpslot, const int32_t* counts, int cap,
                           const int2* rowoff, float* kp7, uint8_t* desc, int32_t* valid_idx, const uint8_t* mdesc,
                           const int32_t* mcounts, int mframes, int mcap, int32_t* idx, int32_t* dist, uint8_t* good,
                           int qt, hipStream_t s);

}  // namespace orbamd
