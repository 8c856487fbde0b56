// mfma_valu_coexec.hip -- do matrix and vector instructions of DIFFERENT waves on one SIMD overlap?
// A 512-thread workgroup puts two waves on each SIMD; waves 0..3 run an FP4 MFMA loop, waves 4..7 a
// v_med3_f32 loop.  Times: MFMA waves alone, VALU waves alone, both.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, int valu_from_acc)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float r = 0;
    if (wave < 4) {
        if (mode & 1) {
            v8i a = {(int)threadIdx.x, 2, 3, 4, 0, 0, 0, 0}, b = {5, 6, 7, (int)blockIdx.x, 0, 0, 0, 0};
            v16f c0 = {0}, c1 = {0};
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0x7F, 0, 0x7F);
                    c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 0x7F, 0, 0x7F);
                }
            }
            for (int i = 0; i < 16; ++i) r += c0[i] + c1[i];
        }
    } else {
        if (mode & 2) {
            float m[8];
            for (int i = 0; i < 8; ++i) m[i] = (float)(threadIdx.x * (i + 1));
            float x = (float)blockIdx.x, y = 3.0f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
#pragma unroll
                    for (int w = 0; w < 8; ++w) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(m[w]) : "v"(x), "v"(y));
                }
            }
            for (int i = 0; i < 8; ++i) r += m[i];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main()
{
    float* d;
    if (hipMalloc(&d, 512 * 256 * 4 * sizeof(float)) != hipSuccess) return 1;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    for (int mode = 1; mode <= 3; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode, 0);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep && ms < best) best = ms;
        }
        // per SIMD: MFMA wave issues 8 MFMA per iter (32 cyc each = 256); VALU wave 64 med3 (4 cyc each = 256)
        printf("mode %d (%s): %.3f ms  = %.1f ns per iteration (256 MFMA-cycles and/or 256 VALU-cycles)\n", mode,
               mode == 1 ? "mfma only" : mode == 2 ? "valu only" : "both", best, best * 1e6 / iters);
    }
    return 0;
}
