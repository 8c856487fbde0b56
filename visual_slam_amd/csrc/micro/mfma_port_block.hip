// mfma_port_block.hip -- for how many cycles does one matrix instruction keep the vector-issue port of
// its SIMD from issuing other waves' vector instructions?
//
// A 1024-thread workgroup puts four waves on each SIMD.  Wave 0 of every SIMD runs a loop of dependent
// matrix instructions (two chains), NV of the other three run independent v_add_f32 at the full issue
// rate.  Times for the vector waves alone, the matrix wave alone, and both; from them
//   blocked cycles per matrix instruction = (both - vector alone) / matrix instructions, in cycles of the
//   clock the chip held in the "both" run (taken from the same run: see below).
//
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_port_block mfma_port_block.hip && build/mfma_port_block
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum { M_FP4_32, M_FP8_32, M_FP4_16, M_COUNT };
static const char* const mname[M_COUNT] = {"32x32x64 fp4 (A,B 4 regs used, C 16)", "32x32x64 fp8 (A,B 8 regs, C 16)",
                                           "16x16x128 fp4 (A,B 4 regs used, C 4)"};
static const int mcycles[M_COUNT] = {32, 64, 16};   // passes x 4 at the nominal rates

template <int MK>
__global__ __launch_bounds__(1024) void k(float* out, int iters, int mode, int nv)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float r = 0;
    if (wave < 4) {
        if (mode & 1) {
            v8i a = {(int)threadIdx.x, 2, 3, 4, 5, 6, 7, 8}, b = {5, 6, 7, (int)blockIdx.x, 1, 2, 3, 4};
            if (MK == M_FP4_16) {
                v4f c0 = {0}, c1 = {0};
                for (int i = 0; i < iters; ++i) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 4, 4, 0, 0x7F, 0, 0x7F);
                        c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 4, 4, 0, 0x7F, 0, 0x7F);
                    }
                }
                for (int i = 0; i < 4; ++i) r += c0[i] + c1[i];
            } else {
                v16f c0 = {0}, c1 = {0};
                for (int i = 0; i < iters; ++i) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (MK == M_FP4_32) {
                            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0x7F, 0, 0x7F);
                            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 0x7F, 0, 0x7F);
                        } else {
                            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 0x7F, 0, 0x7F);
                            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 0x7F, 0, 0x7F);
                        }
                    }
                }
                for (int i = 0; i < 16; ++i) r += c0[i] + c1[i];
            }
        }
    } else if ((mode & 2) && (wave >> 2) <= nv) {
        float m[8];
        for (int i = 0; i < 8; ++i) m[i] = (float)(threadIdx.x * (i + 1));
        float x = (float)blockIdx.x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int w = 0; w < 8; ++w) asm volatile("v_add_f32 %0, %0, %1" : "+v"(m[w]) : "v"(x));
            }
        }
        for (int i = 0; i < 8; ++i) r += m[i];
    }
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}

template <int MK>
static void run(float* d, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](int mode, int nv) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k<MK>, dim3(256), dim3(1024), 0, 0, d, iters, mode, nv);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep && ms < best) best = ms;
        }
        return best * 1e6f / iters;     // ns per iteration
    };
    const float tm = time(1, 0);
    printf("%s: matrix wave alone %.1f ns per 8 instructions = %.2f GHz at %d cycles each\n", mname[MK], tm,
           8.0f * mcycles[MK] / tm, mcycles[MK]);
    for (int nv = 1; nv <= 3; ++nv) {
        const float tv = time(2, nv), tb = time(3, nv);
        // vector alone: nv * 64 instructions per iteration and SIMD
        const float ns_per_vinst = tv / (64.0f * nv);
        printf("   %d vector wave(s) per SIMD: alone %.1f ns (%.2f ns per instruction), with the matrix wave %.1f ns"
               " -> %.1f ns = %.1f vector-instruction slots lost per matrix instruction; sum of the parts %.1f ns\n",
               nv, tv, ns_per_vinst, tb, (tb - tv) / 8.0f, (tb - tv) / 8.0f / ns_per_vinst, tv + tm);
    }
    hipEventDestroy(a); hipEventDestroy(b);
}

int main()
{
    float* d;
    if (hipMalloc(&d, 1024 * 256 * sizeof(float)) != hipSuccess) return 1;
    const int iters = 20000;
    run<M_FP4_32>(d, iters);
    run<M_FP8_32>(d, iters);
    run<M_FP4_16>(d, iters);
    return 0;
}
