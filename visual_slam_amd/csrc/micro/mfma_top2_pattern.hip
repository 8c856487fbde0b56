// mfma_top2_pattern.hip -- the matcher's inner pattern in isolation, registers only: per "unit" four
// dependent FP4 MFMAs into one accumulator, then 24 vector instructions (v_med3_f32 x2 + v_max3_f32 per
// element pair) that consume the 16 results.  WAVES waves per SIMD, two units per loop trip.
// Variants: mfma only, valu only (consuming a constant accumulator), both.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void ins2(float& m1, float& m2, float a, float b, float pinf)
{
    const float s2 = __builtin_amdgcn_fmed3f(m1, a, b);
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(m1), "v"(a), "v"(b), "v"(s2));
    m2 = __builtin_amdgcn_fmed3f(s2, m2, pinf);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters)
{
    v8i a[4], b[2][4];
    for (int i = 0; i < 4; ++i) {
        a[i] = v8i{(int)threadIdx.x + i, 2, 3, 4, 0, 0, 0, 0};
        b[0][i] = v8i{5, 6 + i, 7, (int)blockIdx.x, 0, 0, 0, 0};
        b[1][i] = v8i{9, 6 + i, 1, (int)blockIdx.x, 0, 0, 0, 0};
    }
    v16f c0;
    for (int i = 0; i < 16; ++i) c0[i] = -256.0f - i;
    float m1[2] = {-1e30f, -1e30f}, m2[2] = {-1e30f, -1e30f};
    float pinf = __builtin_inff();
    asm volatile("" : "+v"(pinf));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            v16f acc = c0;
            if (MODE & 1) {
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b[u][0], c0, 4, 4, 0, 0x7F, 0, 0x7F);
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[1], b[u][1], acc, 4, 4, 0, 0x7F, 0, 0x7F);
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[2], b[u][2], acc, 4, 4, 0, 0x7F, 0, 0x7F);
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[3], b[u][3], acc, 4, 4, 0, 0x7F, 0, 0x7F);
            } else {
                asm volatile("" : "+v"(acc));
            }
            if (MODE & 2) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) ins2(m1[u], m2[u], acc[i], acc[i + 1], pinf);
                m1[u] += 0.0078125f;
                m2[u] += 0.0078125f;
            } else {
                m1[u] += acc[0];
            }
        }
        a[0][0] += 1;   // keep the operands changing
    }
    out[blockIdx.x * 256 + threadIdx.x] = m1[0] + m2[0] + m1[1] + m2[1];
}

// Software-pipelined order: every chain of four MFMAs is issued together with the selection of the
// accumulators produced by the chain before it (one MFMA, then a quarter of the vector work, ...).
__device__ __forceinline__ void ins2b(float& m1, float& m2, float a, float b, float pinf)
{
    const float s2 = __builtin_amdgcn_fmed3f(m1, a, b);
    m1 = __builtin_fmaxf(__builtin_fmaxf(m1, a), b);
    m2 = __builtin_amdgcn_fmed3f(s2, m2, pinf);
}
__global__ __launch_bounds__(256, 2) void kpipe(float* out, int iters)
{
    v8i a[4], b[2][4];
    for (int i = 0; i < 4; ++i) {
        a[i] = v8i{(int)threadIdx.x + i, 2, 3, 4, 0, 0, 0, 0};
        b[0][i] = v8i{5, 6 + i, 7, (int)blockIdx.x, 0, 0, 0, 0};
        b[1][i] = v8i{9, 6 + i, 1, (int)blockIdx.x, 0, 0, 0, 0};
    }
    v16f c0, prev;
    for (int i = 0; i < 16; ++i) { c0[i] = -256.0f - i; prev[i] = -1e30f; }
    float m1[2] = {-1e30f, -1e30f}, m2[2] = {-1e30f, -1e30f};
    float pinf = __builtin_inff();
    asm volatile("" : "+v"(pinf));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pu = u ^ 1;
            v16f acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b[u][0], c0, 4, 4, 0, 0x7F, 0, 0x7F);
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[1], b[u][1], acc, 4, 4, 0, 0x7F, 0, 0x7F);
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[2], b[u][2], acc, 4, 4, 0, 0x7F, 0, 0x7F);
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[3], b[u][3], acc, 4, 4, 0, 0x7F, 0, 0x7F);
#pragma unroll
            for (int i = 0; i < 16; i += 2) ins2b(m1[pu], m2[pu], prev[i], prev[i + 1], pinf);
            m1[pu] += 0.0078125f;
            m2[pu] += 0.0078125f;
            prev = acc;
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        }
        a[0][0] += 1;
    }
    out[blockIdx.x * 256 + threadIdx.x] = m1[0] + m2[0] + m1[1] + m2[1] + prev[0];
}

// Both query tiles' chains interleaved (A0 B0 A1 B1 ...), so that no matrix instruction waits for the
// one just before it, with the selection of the previous trip's two accumulators between them.
__global__ __launch_bounds__(256, 2) void kpipe2(float* out, int iters)
{
    v8i a[4], b[2][4];
    for (int i = 0; i < 4; ++i) {
        a[i] = v8i{(int)threadIdx.x + i, 2, 3, 4, 0, 0, 0, 0};
        b[0][i] = v8i{5, 6 + i, 7, (int)blockIdx.x, 0, 0, 0, 0};
        b[1][i] = v8i{9, 6 + i, 1, (int)blockIdx.x, 0, 0, 0, 0};
    }
    v16f c0, p0, p1;
    for (int i = 0; i < 16; ++i) { c0[i] = -256.0f - i; p0[i] = -1e30f; p1[i] = -1e30f; }
    float m1[2] = {-1e30f, -1e30f}, m2[2] = {-1e30f, -1e30f};
    float pinf = __builtin_inff();
    asm volatile("" : "+v"(pinf));
    for (int it = 0; it < iters; ++it) {
        v16f x = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b[0][0], c0, 4, 4, 0, 0x7F, 0, 0x7F);
        v16f y = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b[1][0], c0, 4, 4, 0, 0x7F, 0, 0x7F);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            x = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[k], b[0][k], x, 4, 4, 0, 0x7F, 0, 0x7F);
            y = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[k], b[1][k], y, 4, 4, 0, 0x7F, 0, 0x7F);
        }
#pragma unroll
        for (int i = 0; i < 16; i += 2) ins2b(m1[0], m2[0], p0[i], p0[i + 1], pinf);
#pragma unroll
        for (int i = 0; i < 16; i += 2) ins2b(m1[1], m2[1], p1[i], p1[i + 1], pinf);
        m1[0] += 0.0078125f; m2[0] += 0.0078125f; m1[1] += 0.0078125f; m2[1] += 0.0078125f;
        p0 = x; p1 = y;
#define GRP(N) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, N, 0);
        GRP(7) GRP(7) GRP(7) GRP(7) GRP(6) GRP(6) GRP(6) GRP(6)
#undef GRP
        a[0][0] += 1;
    }
    out[blockIdx.x * 256 + threadIdx.x] = m1[0] + m2[0] + m1[1] + m2[1] + p0[0] + p1[0];
}

typedef float v4f __attribute__((ext_vector_type(4)));
// Same work per unit with the 16x16x128 shape: 4 output tiles x 2 k-steps, 4 results per tile and lane.
template <int MODE>
__global__ __launch_bounds__(256, 2) void k16(float* out, int iters)
{
    v8i a[2][2], b[2][2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            a[i][j] = v8i{(int)threadIdx.x + i, 2 + j, 3, 4, 0, 0, 0, 0};
            for (int u = 0; u < 2; ++u) b[u][i][j] = v8i{5 + u, 6 + i, 7 + j, (int)blockIdx.x, 0, 0, 0, 0};
        }
    v4f c0 = {-256.f, -257.f, -258.f, -259.f};
    float m1[2] = {-1e30f, -1e30f}, m2[2] = {-1e30f, -1e30f};
    float pinf = __builtin_inff();
    asm volatile("" : "+v"(pinf));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    v4f acc = c0;
                    if (MODE & 1) {
                        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[ti][0], b[u][tj][0], c0, 4, 4, 0, 0x7F, 0, 0x7F);
                        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[ti][1], b[u][tj][1], acc, 4, 4, 0, 0x7F, 0, 0x7F);
                    } else {
                        asm volatile("" : "+v"(acc));
                    }
                    if (MODE & 2) {
                        ins2(m1[u], m2[u], acc[0], acc[1], pinf);
                        ins2(m1[u], m2[u], acc[2], acc[3], pinf);
                    } else {
                        m1[u] += acc[0];
                    }
                }
            }
            m1[u] += 0.0078125f;
            m2[u] += 0.0078125f;
        }
        a[0][0][0] += 1;
    }
    out[blockIdx.x * 256 + threadIdx.x] = m1[0] + m2[0] + m1[1] + m2[1];
}

template <int MODE, int SHAPE>
float run(float* d, int wgs_per_cu, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        if (MODE == 4) hipLaunchKernelGGL(kpipe, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
        else if (MODE == 5) hipLaunchKernelGGL(kpipe2, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
        else if (SHAPE == 32) hipLaunchKernelGGL(k<MODE>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
        else hipLaunchKernelGGL(k16<MODE>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main()
{
    float* d;
    if (hipMalloc(&d, 256 * 8 * 256 * sizeof(float)) != hipSuccess) return 1;
    const int iters = 20000;
    for (int w = 1; w <= 4; ++w) {
        const float t1 = run<1, 32>(d, w, iters), t2 = run<2, 32>(d, w, iters), t3 = run<3, 32>(d, w, iters);
        // per SIMD and trip: w waves x 2 units x (4 MFMA = 128 cycles, 26 VALU = 104 cycles)
        const float t4 = run<4, 32>(d, w, iters), t5 = run<5, 32>(d, w, iters);
        printf("%d waves/SIMD: ns per unit  mfma %.1f  valu %.1f  both %.1f   (sum %.1f)  both, software-pipelined %.1f, with the two chains interleaved %.1f\n", w,
               t1 * 1e6 / iters / (2 * w), t2 * 1e6 / iters / (2 * w), t3 * 1e6 / iters / (2 * w),
               (t1 + t2) * 1e6 / iters / (2 * w), t4 * 1e6 / iters / (2 * w), t5 * 1e6 / iters / (2 * w));
        const float s1 = run<1, 16>(d, w, iters), s2 = run<2, 16>(d, w, iters), s3 = run<3, 16>(d, w, iters);
        printf("   16x16x128 shape:          mfma %.1f  valu %.1f  both %.1f\n", s1 * 1e6 / iters / (2 * w),
               s2 * 1e6 / iters / (2 * w), s3 * 1e6 / iters / (2 * w));
    }
    return 0;
}
