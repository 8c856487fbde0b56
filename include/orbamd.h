/*
 * orbamd.h -- C ABI of liborbamd.so, the MI355X (gfx950) implementation of the reference's
 * custom ORB front-end: FAST segment-test detection, intensity-moment orientation, steered
 * BRIEF-256 description and brute-force Hamming top-2 matching.
 *
 * The reference has no C ABI; its boundary is the pybind11 module `orb_project`
 * (reference src/main_interface.cpp:104-110) over three static C++ methods
 * (reference include/orb.hpp:14-20).  Each entry point below names the reference interface it
 * stands in for.  Plain pointers and sizes only; no exceptions cross this boundary.
 *
 * Conventions
 *   - Every function returns an int status: 0 = ORBAMD_OK, negative = error (see below).
 *   - A context owns one HIP stream plus grow-only device work buffers.  One context per host
 *     thread; calls on one context must not overlap.
 *   - Functions without a suffix take HOST pointers, copy in, run, copy out and return after the
 *     stream has drained.  Functions ending in `_dev` take DEVICE pointers, only enqueue work on
 *     the context's stream and return at once; use orbamd_ctx_sync() before reading results.
 *   - Images are 8-bit, row-major, tightly packed (row stride = W bytes, frame stride = H*W).
 *   - Keypoint coordinates are (x = column, y = row).  Keypoints come in the reference's scan
 *     order: ascending y, then ascending x (reference src/orb.cpp:51-52, :100).
 *   - Descriptors are 32 bytes; bit (k % 8) of byte (k / 8) is test pair k
 *     (reference src/orb.cpp:188, :223, :228).
 *   - This library has no CPU code path: without a usable GPU, orbamd_ctx_create() fails.
 */
#ifndef ORBAMD_H
#define ORBAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBAMD_OK            0
#define ORBAMD_E_BADARG     (-1)  /* null pointer, non-positive size, train rows >= 2^18, ... */
#define ORBAMD_E_CAPACITY   (-2)  /* caller's output capacity too small; needed size reported */
#define ORBAMD_E_DEVICE     (-3)  /* a HIP call failed; text via orbamd_last_error() */
#define ORBAMD_E_NOMEM      (-4)  /* device or host allocation failed */
#define ORBAMD_E_STATE      (-5)  /* call sequence wrong (e.g. fetch before extract) */

/* Largest train set a matcher call accepts (train indices are packed into 18 bits; OpenCV's
 * BFMatcher has the same limit). */
#define ORBAMD_MAX_TRAIN    (1 << 18)
/* Largest per-frame keypoint capacity (keypoint rows are addressed by 32-bit byte offsets inside
 * a frame's arrays).  A single frame with more corners than this makes the single-frame entry
 * points return ORBAMD_E_CAPACITY. */
#define ORBAMD_MAX_CAP      (1 << 26)

typedef struct orbamd_ctx orbamd_ctx;

/* Library version as major*10000 + minor*100 + patch. */
int orbamd_version(void);
/* Static text for a status code. */
const char* orbamd_status_string(int status);

/* ---- context ---------------------------------------------------------------------------- */

/* Create a context on HIP device `device` with its own non-blocking stream. */
int orbamd_ctx_create(int device, orbamd_ctx** out);
int orbamd_ctx_destroy(orbamd_ctx* ctx);
/* Use a caller-owned hipStream_t (passed as void*) for all later work.  The handle is taken as it
 * is: NULL means HIP's default (null) stream, the stream a host framework reports when its current
 * stream is the default one.  Work queued before the call is waited for. */
int orbamd_ctx_set_stream(orbamd_ctx* ctx, void* hip_stream);
/* Go back to the context's own non-blocking stream (the state after orbamd_ctx_create). */
int orbamd_ctx_use_own_stream(orbamd_ctx* ctx);
/* Size the context's work buffers for batches of up to `frames` frames of H x W pixels with per-frame
 * capacity `cap` (orbamd_extract_batch_dev, and orbamd_knn2_batch_dev over the consecutive pairs of
 * such a batch).  The `_dev` calls only enqueue work, EXCEPT that a call which needs a larger work
 * buffer than the context holds first waits for the stream, frees the buffer and allocates a bigger
 * one; after a reserve that covers the sizes in use that never happens.  Optional. */
int orbamd_ctx_reserve(orbamd_ctx* ctx, int frames, int H, int W, int cap);
/* Block until everything enqueued on the context's stream has finished. */
int orbamd_ctx_sync(orbamd_ctx* ctx);
/* Text of the last HIP failure seen by this context ("" if none). */
const char* orbamd_last_error(orbamd_ctx* ctx);
/* Which kernel family computes the Hamming top-2 (all orbamd_knn2* calls of this context).  Both
 * give the results of cv2.BFMatcher(NORM_HAMMING).knnMatch(q, t, 2) (reference views.py:65, :162)
 * bit for bit.  MATRIX (default): descriptors as +-1 FP4 values on the matrix cores, distance =
 * (256 - dot) / 2, exact.  VALU: XOR + popcount on the vector ALUs.  The default can also be
 * chosen with the environment variable ORBAMD_MATCHER=valu|matrix read at context creation.
 * No counterpart in the reference. */
#define ORBAMD_MATCHER_MATRIX 0
#define ORBAMD_MATCHER_VALU   1
int orbamd_ctx_set_matcher(orbamd_ctx* ctx, int matcher);

/* Detector EXTENSION, off by default; no counterpart in the reference, whose detector is fixed at
 * threshold 20, 12 contiguous pixels, no wrap-around, no score and no suppression (src/orb.cpp:39,
 * :70-96; its OpenCV call with suppression is commented out at :104-105).  Applies to every later
 * detect / extract call of the context:
 *   threshold 1 .. 255     circle pixel counts when >= centre + threshold or <= centre - threshold
 *   arc       9 .. 16      contiguous circle pixels required (9 = the usual "FAST-9")
 *   wrap      0 / 1        whether a run may continue from k = 15 to k = 0
 *   nms       0 / 1        keep a corner only if its score (the largest threshold it would still pass)
 *                          is strictly greater than the scores of its 8 neighbours
 * (20, 12, 0, 0) is the reference's detector and runs its bit-exact kernel; anything else runs a separate
 * tile kernel (scores and suppression in LDS, no scratch memory).  Keypoint order, orientation and
 * descriptors are computed as always. */
int orbamd_ctx_set_detector(orbamd_ctx* ctx, int threshold, int arc, int wrap, int nms);

/* Parity audit, off by default; no counterpart in the reference.  The reference narrows doubles that come out
 * of libm calls to float: the angle atan2(m01, m10) * (180 / pi) (src/orb.cpp:127-129) and every rotated sample
 * coordinate x * cos - y * sin (src/orb.cpp:195-199).  The device's arc tangent, sine and cosine are the
 * library's own series, not glibc's code; a difference in the last bits of such a double changes the float
 * only if the double lies next to the midpoint of two neighbouring floats.  With the audit on, the orientation / descriptor kernels count exactly those cases, per frame:
 *   [0] keypoints whose angle (before narrowing) is within 4 units in the last place of such a midpoint
 *   [1] sample coordinates (before narrowing) within 4 units in the last place of such a midpoint
 *   [2] those of [1] for which the two candidate floats, added to the keypoint's coordinate, also round to
 *       different pixels (src/orb.cpp:202-206) -- only these could change a descriptor bit
 *   [3] reserved (0)
 * Expected values are about 1.5e-8 per keypoint / per coordinate for [0] / [1] and practically never for [2];
 * a frame with a non-zero [0] or [2] is one to re-check on the host if bit-exactness with a glibc build is
 * required.  The audited kernels issue about a quarter more instructions; results are unchanged.
 * orbamd_ctx_audit_read copies the counters of the LAST orient / describe / extract call (batch or single
 * frame; `frames` <= the frames of that call) to host memory, after waiting for the stream:
 * out = int32 [frames][4].  ORBAMD_E_STATE when the audit is off or no audited call has run. */
int orbamd_ctx_set_audit(orbamd_ctx* ctx, int on);
int orbamd_ctx_audit_read(orbamd_ctx* ctx, int32_t* out, int frames);

/* Per-kernel timing: when enabled, every kernel launch is bracketed by HIP events on the
 * context's stream.  orbamd_ctx_timing_read() drains the stream and returns, for kernel kind k
 * (ORBAMD_K_*), the summed milliseconds and launch count since the last reset. */
#define ORBAMD_K_BGR2GRAY   0
#define ORBAMD_K_FAST       1
#define ORBAMD_K_ROWCOUNT   2   /* the fill that zeroes the per-row corner totals before the detector runs */
#define ORBAMD_K_ROWSCAN    3
#define ORBAMD_K_EMIT       4
#define ORBAMD_K_DESCRIBE   5
#define ORBAMD_K_KNN2       6
#define ORBAMD_K_KNN2_MERGE 7
#define ORBAMD_K_SYNTH      8
#define ORBAMD_K_RATIO      9
#define ORBAMD_K_COUNT      10
int orbamd_ctx_timing_enable(orbamd_ctx* ctx, int on);
int orbamd_ctx_timing_reset(orbamd_ctx* ctx);
int orbamd_ctx_timing_read(orbamd_ctx* ctx, double ms_out[ORBAMD_K_COUNT], long launches_out[ORBAMD_K_COUNT]);
const char* orbamd_kernel_name(int kind);

/* Where the HOST's wall time goes inside the single-frame, host-pointer calls (orbamd_extract_*,
 * orbamd_fetch, orbamd_knn2), accumulated in microseconds while timing is enabled and cleared by
 * orbamd_ctx_timing_reset():
 *   H2D      copying the caller's arrays towards the device (staging into pinned memory + enqueueing the DMA)
 *   ENQUEUE  sizing work buffers and launching kernels
 *   WAIT     blocked until the device has finished
 *   D2H      bringing results back (enqueueing the DMA + copying from pinned memory into the caller's arrays)
 * No counterpart in the reference (its only timing is time.perf_counter around extract, views.py:269-279). */
#define ORBAMD_HT_H2D       0
#define ORBAMD_HT_ENQUEUE   1
#define ORBAMD_HT_WAIT      2
#define ORBAMD_HT_D2H       3
#define ORBAMD_HT_COUNT     4
int orbamd_ctx_host_timing_read(orbamd_ctx* ctx, double us_out[ORBAMD_HT_COUNT]);

/* ---- single frame, host pointers (the drop-in path) -------------------------------------- */

/* cv::cvtColor(frame, gray, COLOR_BGR2GRAY) for 8-bit input (reference
 * src/main_interface.cpp:83): gray = (B*3735 + G*19235 + R*9798 + 2^14) >> 15. */
int orbamd_bgr2gray(orbamd_ctx* ctx, const uint8_t* bgr, int H, int W, uint8_t* gray_out);

/* ORBDescriptor::detectKeypoints (reference src/orb.cpp:24-108; include/orb.hpp:14).
 * Writes up to `cap` (x, y) pairs to out_xy and the total found to *n_out; returns
 * ORBAMD_E_CAPACITY (with *n_out set) when the total exceeds cap. */
int orbamd_detect(orbamd_ctx* ctx, const uint8_t* gray, int H, int W,
                  int32_t* out_xy, int cap, int* n_out);

/* ORBDescriptor::computeOrientation (reference src/orb.cpp:110-140; include/orb.hpp:17) for n
 * caller-supplied integer keypoints; angle in degrees in [0, 360] as float.  Any int32 is a legal
 * coordinate: a patch that lies outside the image has no pixels and gets angle 0, one that
 * straddles an edge is clipped (src/orb.cpp:250).  n <= ORBAMD_MAX_CAP. */
int orbamd_orient(orbamd_ctx* ctx, const uint8_t* gray, int H, int W,
                  const int32_t* xy, int n, float* angle_out);

/* ORBDescriptor::computeDescriptors (reference src/orb.cpp:142-234; include/orb.hpp:20).
 * Keypoints closer than 16 px to an edge get no descriptor; desc_out receives *nv_out rows of
 * 32 bytes in keypoint order and valid_idx_out (may be NULL) the keypoint index of each row.
 * desc_out and valid_idx_out must hold n rows.  n <= ORBAMD_MAX_CAP. */
int orbamd_describe(orbamd_ctx* ctx, const uint8_t* gray, int H, int W,
                    const int32_t* xy, const float* angle, int n,
                    uint8_t* desc_out, int32_t* valid_idx_out, int* nv_out);

/* extract() of the pybind module (reference src/main_interface.cpp:80-101), in two steps so the
 * caller can size its arrays exactly: orbamd_extract_* runs detect + orient + describe on the
 * device and reports the counts; orbamd_fetch copies the results of the LAST extract out:
 *   kp7_out       float32 [n, 7] = [x, y, 7, angle, 0, 0, -1] (reference :39-55)
 *   desc_out      uint8   [nv, 32]
 *   valid_idx_out int32   [nv]  keypoint index of each descriptor row (NULL to skip)
 * orbamd_extract_bgr takes the [H, W, 3] frame the reference's extract() takes. */
int orbamd_extract_gray(orbamd_ctx* ctx, const uint8_t* gray, int H, int W, int* n_out, int* nv_out);
int orbamd_extract_bgr(orbamd_ctx* ctx, const uint8_t* bgr, int H, int W, int* n_out, int* nv_out);
int orbamd_fetch(orbamd_ctx* ctx, float* kp7_out, uint8_t* desc_out, int32_t* valid_idx_out);

/* cv2.BFMatcher(NORM_HAMMING, crossCheck=False).knnMatch(q, t, 2) (call site: reference
 * views.py:162; src/main.cpp:75).  For each of nq query rows the two train rows with the
 * smallest Hamming distance, ties broken towards the lower train index, best first.
 *   idx_out  int32 [nq, 2]   train index, -1 where the train set has fewer rows than needed
 *   dist_out int32 [nq, 2]   distance in bits, -1 likewise
 * nt must be < ORBAMD_MAX_TRAIN. */
int orbamd_knn2(orbamd_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt,
                int32_t* idx_out, int32_t* dist_out);

/* Lowe ratio filter as reference views.py:163-171 applies it to knnMatch output: query i is kept
 * iff dist[i][0] < 0.75 * dist[i][1].  Kept rows, in query order:
 *   good_q[m]  = i              (m1.queryIdx)
 *   good_t2[m] = idx[i][1]      (m2.trainIdx -- the SECOND neighbour, as the reference stores it)
 *   good_t1[m] = idx[i][0]      (m1.trainIdx, used for the point pairs)
 * Returns ORBAMD_E_BADARG if any query has fewer than two neighbours (the reference's tuple
 * unpacking raises there).  Arrays must hold nq entries. */
int orbamd_ratio_filter(orbamd_ctx* ctx, const int32_t* idx, const int32_t* dist, int nq,
                        int32_t* good_q, int32_t* good_t2, int32_t* good_t1, int* m_out);

/* orbamd_knn2 followed by orbamd_ratio_filter on its output, in one call with one wait for the device: the
 * knnMatch + ratio loop of reference views.py:162-171.  idx_out / dist_out as orbamd_knn2 (always
 * written); good_q / good_t2 / good_t1 as orbamd_ratio_filter, *m_out rows.  When some query has fewer
 * than two neighbours (nt < 2) *m_out is -1 and the status ORBAMD_E_BADARG, as orbamd_ratio_filter
 * reports it.  The three lists must hold nq entries. */
int orbamd_knn2_ratio(orbamd_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt,
                      int32_t* idx_out, int32_t* dist_out, int32_t* good_q, int32_t* good_t2,
                      int32_t* good_t1, int* m_out);

/* ---- frame batches, device pointers ------------------------------------------------------ */

/* Alignment: image pointers may have any alignment (16-byte-aligned frames whose width is a multiple of 16
 * take the fastest kernels).  Descriptor pointers handed to the orbamd_knn2*_dev calls must be 16-byte
 * aligned (rows are fetched 16 bytes at a time); a misaligned one is ORBAMD_E_BADARG. */

/* Synthetic gray frames written straight into device memory: frame i of the call is frame
 * (first_frame + i) of the seeded scene (see visual_slam_amd/synth.py for the definition).
 * density_thr = round(density * 65536). */
int orbamd_synth_frames_dev(orbamd_ctx* ctx, uint8_t* d_gray, int frames, int H, int W,
                            int first_frame, uint32_t seed, int cell, int density_thr);

/* The textured synthetic scene (synth_textured_frame in visual_slam_amd/synth.py): piecewise-constant
 * texture with local contrast far above the FAST threshold, about 11 k keypoints per 1080p frame in
 * clusters of adjacent pixels.  Measurement input; no counterpart in the reference. */
int orbamd_synth_textured_dev(orbamd_ctx* ctx, uint8_t* d_gray, int frames, int H, int W,
                              int first_frame, uint32_t seed);

/* BGR -> gray for `frames` frames of H*W pixels (device pointers). */
int orbamd_bgr2gray_dev(orbamd_ctx* ctx, const uint8_t* d_bgr, int frames, int H, int W,
                        uint8_t* d_gray);

/* detect + orient + describe for `frames` gray frames resident in device memory.  Outputs are
 * frame-strided with a fixed per-frame capacity `cap`:
 *   d_kp7       float32 [frames, cap, 7]
 *   d_desc      uint8   [frames, cap, 32]
 *   d_valid_idx int32   [frames, cap]
 *   d_counts    int32   [frames, 2] = (n, nd): n = keypoints the frame holds (the TRUE total, may
 *                                      exceed cap); nd = descriptor rows stored for it.
 * Rows past the counts are left untouched.  When n <= cap, nd is the reference's descriptor count
 * (keypoints >= 16 px from every edge, src/orb.cpp:156).  A frame with n > cap is an overflow: only
 * its first cap keypoints are stored, nd counts the descriptors of exactly those keypoints (the
 * first nd rows of what a larger cap would give), and every matching entry point works on those nd
 * rows.  The caller detects it by n > cap and extracts again with a larger cap if it needs the rest.
 * The call only enqueues work, but the first call at a given batch size (and any later, larger one)
 * grows the context's work buffers, which waits for the stream and allocates device memory. */
int orbamd_extract_batch_dev(orbamd_ctx* ctx, const uint8_t* d_gray, int frames, int H, int W,
                             int cap, float* d_kp7, uint8_t* d_desc, int32_t* d_valid_idx,
                             int32_t* d_counts);

/* Top-2 matching for frame pairs inside a batch produced by orbamd_extract_batch_dev.
 * d_pairs is int32 [npairs, 2] = (query frame, train frame); NULL means the consecutive pairs
 * (p + 1, p) for p in [0, npairs) -- current frame as query, previous as train
 * (reference views.py:88-90).  Descriptor counts are read from d_counts on the device (clamped
 * to cap), so no host round trip separates extraction from matching.  A d_pairs entry that names
 * a frame outside [0, frames) is skipped: its output rows stay untouched (and the ratio filter
 * reports -1 kept for it).
 *   d_idx  int32 [npairs, cap, 2], d_dist int32 [npairs, cap, 2]  as orbamd_knn2
 *   d_good uint8 [npairs, cap]  1 iff both neighbours exist and dist0 < 0.75 * dist1
 *                               (reference views.py:168); may be NULL. */
int orbamd_knn2_batch_dev(orbamd_ctx* ctx, const uint8_t* d_desc, const int32_t* d_counts,
                          int frames, int cap, const int32_t* d_pairs, int npairs,
                          int32_t* d_idx, int32_t* d_dist, uint8_t* d_good);

/* The ratio-filter loop (reference views.py:163-171) for every pair of a batch matched by
 * orbamd_knn2_batch_dev, on the device: per pair, in ascending query order, the kept matches as
 * three lists with the reference's contents -- m1.queryIdx, m2.trainIdx (sic, views.py:170) and
 * m1.trainIdx.  d_pairs / frame numbering as in orbamd_knn2_batch_dev (the query frame's descriptor
 * count comes from d_counts).
 *   d_good_q, d_good_t2, d_good_t1  int32 [npairs, cap]   first d_kept[p] entries of row p are valid
 *   d_kept                          int32 [npairs]        number kept; -1 when some query of the pair
 *                                   has fewer than two neighbours (the reference's loop raises there) */
int orbamd_ratio_filter_batch_dev(orbamd_ctx* ctx, const int32_t* d_idx, const uint8_t* d_good,
                                  const int32_t* d_counts, int frames, int cap, const int32_t* d_pairs,
                                  int npairs, int32_t* d_good_q, int32_t* d_good_t2, int32_t* d_good_t1,
                                  int32_t* d_kept);

/* One frame pair whose two frames live in different frame-strided buffers with device-side
 * counts: query frame `qframe` of (d_qdesc, d_qcounts) against train frame `tframe` of
 * (d_tdesc, d_tcounts), both with per-frame capacity `cap`.  This is the pair that spans two
 * GPUs' frame ranges: the train side is the boundary frame received from the neighbouring GPU.
 * Outputs as one pair of orbamd_knn2_batch_dev: d_idx/d_dist int32 [cap, 2], d_good uint8 [cap].
 * The call is given no frame counts of the two buffers, so it cannot range-check qframe and
 * tframe beyond >= 0: the caller guarantees that both frames exist. */
int orbamd_knn2_cross_dev(orbamd_ctx* ctx, const uint8_t* d_qdesc, const int32_t* d_qcounts, int qframe,
                          const uint8_t* d_tdesc, const int32_t* d_tcounts, int tframe, int cap,
                          int32_t* d_idx, int32_t* d_dist, uint8_t* d_good);

/* Top-2 matching of two plain descriptor sets in device memory with host-side sizes. */
int orbamd_knn2_dev(orbamd_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt,
                    int32_t* d_idx, int32_t* d_dist, uint8_t* d_good);

/* orbamd_knn2_dev without any allocation.  No counterpart in the reference; it is for callers that
 * cannot allocate device memory at call time: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  Same arguments and the same results as orbamd_knn2_dev, but the
 * partial results of the train-axis split live in a fixed 4 MiB array inside the library (one copy
 * per GPU, held until the process ends) instead of a context buffer that grows on demand, the split
 * count is lowered until they fit, and nothing is timed.  The call makes no hipMalloc / hipFree and
 * creates no events; it only enqueues kernels on the context's stream.
 *   nq <= ORBAMD_NOALLOC_MAX_QUERIES, else ORBAMD_E_CAPACITY (text via orbamd_last_error()).
 * All calls on one GPU share that GPU's array, from every context: calls that overlap in time on
 * one GPU (two contexts, two streams) race on it and must be serialised by the caller.  The array
 * keeps the previous call's contents; the kernels overwrite every slot they read, so no clearing is
 * needed between calls. */
#define ORBAMD_NOALLOC_MAX_QUERIES 65536
int orbamd_knn2_dev_noalloc(orbamd_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt,
                            int32_t* d_idx, int32_t* d_dist, uint8_t* d_good);

/* Test hook, no counterpart in the reference: the library's own arc tangent / sine / cosine (what the
 * orientation step uses in place of the libm calls of reference src/orb.cpp:127 and :173-174), evaluated
 * element-wise on the device, so that a test can compare them bit for bit with a host build of the same source
 * (csrc/orb_trig.h) and so carry the host-side accuracy measurements over to the device.
 *   d_deg[i]    = atan2(d_m01[i], d_m10[i]) * (180 / pi) as a double, i < n (|moments| < 2^22)
 *   d_unsure[i] = 1 when that double is within 2^-43 (relative) of a float rounding midpoint
 *   d_sin[j], d_cos[j] of d_rad[j], j < m (0 <= d_rad[j] <= 6.5)
 * Device pointers; asynchronous on the context's stream. */
int orbamd_trig_selftest_dev(orbamd_ctx* ctx, const int32_t* d_m01, const int32_t* d_m10, int n, double* d_deg,
                             uint8_t* d_unsure, const double* d_rad, int m, double* d_sin, double* d_cos);

#ifdef __cplusplus
}
#endif
#endif /* ORBAMD_H */
