// coexec_classes.hip -- which classes of vector instructions run beside FP4 matrix instructions of ANOTHER
// wave on the same SIMD, and which take turns with them?
//
// A 512-thread workgroup puts two waves on each SIMD.  Waves 0..3 run a loop of dependent
// v_mfma_scale_f32_32x32x64_f8f6f4 (two chains), waves 4..7 a loop of 64 independent instructions of one
// class.  For every class: the matrix waves alone, the vector waves alone, both.  "both" near max(alone)
// means the two pipes overlap; near the sum means they exclude each other.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/coexec_classes coexec_classes.hip && build/coexec_classes
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

enum { C_MED3_F32, C_ADD_F32, C_MUL_F64, C_ADD_F64, C_FMA_F64, C_CVT_F32_F64, C_SAD_U8, C_DOT4, C_BCNT, C_XOR,
       C_MAX3_F32, C_PK_ADD_F32, C_LDS_READ, C_COUNT };
static const char* const class_name[C_COUNT] = {
    "v_med3_f32", "v_add_f32", "v_mul_f64", "v_add_f64", "v_fma_f64", "v_cvt_f32_f64", "v_sad_u8", "v_dot4_u32_u8",
    "v_bcnt_u32_b32", "v_xor_b32", "v_max3_f32", "v_pk_add_f32", "ds_read_b32"};

template <int CLS>
__device__ __forceinline__ float valu_loop(int iters, uint32_t* lds)
{
    float m[8];
    double d[8];
    uint32_t u[8];
    for (int i = 0; i < 8; ++i) {
        m[i] = (float)(threadIdx.x * (i + 1));
        d[i] = (double)(threadIdx.x + i) * 1.0000001;
        u[i] = threadIdx.x * 2654435761u + i;
    }
    float x = (float)blockIdx.x, y = 3.0f;
    double dx = 1.0000000001 + blockIdx.x * 1e-12;
    uint32_t ux = 0x01020304u + blockIdx.x;
    float2 pk[8];
    for (int i = 0; i < 8; ++i) pk[i] = make_float2(m[i], m[i] + 1.0f);
    const uint32_t lds_addr = (threadIdx.x & 63) * 4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                if (CLS == C_MED3_F32) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(m[w]) : "v"(x), "v"(y));
                if (CLS == C_ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(m[w]) : "v"(x));
                if (CLS == C_MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[w]) : "v"(dx));
                if (CLS == C_ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[w]) : "v"(dx));
                if (CLS == C_FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[w]) : "v"(dx));
                if (CLS == C_CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(m[w]) : "v"(d[w]));
                if (CLS == C_SAD_U8) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(u[w]) : "v"(ux));
                if (CLS == C_DOT4) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(u[w]) : "v"(ux));
                if (CLS == C_BCNT) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(u[w]) : "v"(ux));
                if (CLS == C_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[w]) : "v"(ux));
                if (CLS == C_MAX3_F32) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m[w]) : "v"(x), "v"(y));
                if (CLS == C_PK_ADD_F32) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[w]) : "v"(pk[(w + 1) & 7]));
                if (CLS == C_LDS_READ) asm volatile("ds_read_b32 %0, %1" : "=v"(u[w]) : "v"(lds_addr));
            }
        }
        if (CLS == C_LDS_READ) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += m[i] + (float)d[i] + (float)u[i] + pk[i].x;
    (void)lds;
    return r;
}

template <int CLS>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode)
{
    __shared__ uint32_t lds[256];
    if (threadIdx.x < 256) lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
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
    } else if (mode & 2) {
        r = valu_loop<CLS>(iters, lds);
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int CLS>
static void run(float* d, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float t[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k<CLS>, dim3(256), dim3(512), 0, 0, d, iters, mode);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep && ms < best) best = ms;
        }
        t[mode] = best * 1e6f / iters;
    }
    // per iteration and SIMD: 8 matrix instructions (32 cycles each) and / or 64 vector instructions
    const float sum = t[1] + t[2], mx = t[1] > t[2] ? t[1] : t[2];
    printf("%-16s mfma alone %7.1f ns  (%.2f GHz if 256 cycles)   vector alone %7.1f ns (%.2f cycles each at 2.4 GHz)   "
           "both %7.1f ns   overlap %3.0f %%\n",
           class_name[CLS], t[1], 256.0f / t[1], t[2], t[2] * 2.4f / 64.0f, t[3], 100.0f * (sum - t[3]) / (sum - mx));
    hipEventDestroy(a); hipEventDestroy(b);
}

int main()
{
    float* d;
    if (hipMalloc(&d, 512 * 256 * sizeof(float)) != hipSuccess) return 1;
    const int iters = 20000;
    printf("two waves per SIMD, one issues matrix instructions only, the other 64 vector instructions per iteration;\n"
           "overlap 100 %% = both take max(alone), 0 %% = both take the sum\n");
    run<C_MED3_F32>(d, iters);
    run<C_MAX3_F32>(d, iters);
    run<C_ADD_F32>(d, iters);
    run<C_PK_ADD_F32>(d, iters);
    run<C_XOR>(d, iters);
    run<C_BCNT>(d, iters);
    run<C_SAD_U8>(d, iters);
    run<C_DOT4>(d, iters);
    run<C_MUL_F64>(d, iters);
    run<C_ADD_F64>(d, iters);
    run<C_FMA_F64>(d, iters);
    run<C_CVT_F32_F64>(d, iters);
    run<C_LDS_READ>(d, iters);
    return 0;
}
