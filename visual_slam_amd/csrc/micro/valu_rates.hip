// valu_rates.hip -- issue rate of individual VALU instructions on gfx950 (register-only loops,
// 8 waves per SIMD on every CU).  Used to choose the instruction mix of the integer kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define DEFKERNEL(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, int iters, uint32_t seed) \
    {                                                                                     \
        uint32_t a[8];                                                                    \
        for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u;    \
        uint32_t t = seed ^ blockIdx.x, u = seed * 3u;                                    \
        for (int it = 0; it < iters; ++it) {                                              \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                               \
                _Pragma("unroll") for (int w = 0; w < 8; ++w)                             \
                    asm volatile(ASM : "+v"(a[w]) : "v"(t), "v"(u));                      \
            }                                                                             \
        }                                                                                 \
        uint32_t s = 0;                                                                   \
        for (int i = 0; i < 8; ++i) s += a[i];                                            \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                          \
    }

DEFKERNEL(xor_e32, "v_xor_b32_e32 %0, %1, %0")
DEFKERNEL(xor_e64, "v_xor_b32_e64 %0, %1, %0")
DEFKERNEL(add_e32, "v_add_u32_e32 %0, %1, %0")
DEFKERNEL(min_u32, "v_min_u32_e32 %0, %1, %0")
DEFKERNEL(bcnt, "v_bcnt_u32_b32 %0, %1, %0")
DEFKERNEL(bcnt0, "v_bcnt_u32_b32 %0, %0, 0")
DEFKERNEL(add3, "v_add3_u32 %0, %1, %2, %0")
DEFKERNEL(lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
DEFKERNEL(bfi, "v_bfi_b32 %0, %1, %2, %0")
DEFKERNEL(and_or, "v_and_or_b32 %0, %0, %1, %2")
DEFKERNEL(med3, "v_med3_u32 %0, %1, %2, %0")
DEFKERNEL(min3, "v_min3_u32 %0, %1, %2, %0")
DEFKERNEL(max3, "v_max3_u32 %0, %1, %2, %0")
DEFKERNEL(perm, "v_perm_b32 %0, %1, %0, %2")
DEFKERNEL(alignbyte, "v_alignbyte_b32 %0, %1, %0, 1")
DEFKERNEL(mad_u24, "v_mad_u32_u24 %0, %1, %2, %0")
DEFKERNEL(mul_u24, "v_mul_u32_u24_e32 %0, %1, %0")
DEFKERNEL(sad_u8, "v_sad_u8 %0, %1, %2, %0")
DEFKERNEL(pk_add_u16, "v_pk_add_u16 %0, %1, %0")
DEFKERNEL(pk_min_u16, "v_pk_min_u16 %0, %1, %0")
DEFKERNEL(pk_max_u16, "v_pk_max_u16 %0, %1, %0")
DEFKERNEL(pk_sub_i16, "v_pk_sub_i16 %0, %1, %0")
DEFKERNEL(dot4_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
DEFKERNEL(cmp_gt, "v_cmp_gt_u32_e32 vcc, %1, %0\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc")
DEFKERNEL(cmp_sgpr, "v_cmp_gt_u32_e64 s[20:21], %1, %0")
DEFKERNEL(cndmask, "v_cndmask_b32_e32 %0, %1, %0, vcc")
DEFKERNEL(fma_f32, "v_fma_f32 %0, %1, %2, %0")
DEFKERNEL(bfe, "v_bfe_u32 %0, %0, 3, 8")
DEFKERNEL(mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEFKERNEL(add_sdwa, "v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
DEFKERNEL(cmp_sdwa, "v_cmp_gt_u32_sdwa s[20:21], %1, %0 src0_sel:BYTE_1 src1_sel:BYTE_2")

DEFKERNEL(and_b32, "v_and_b32_e32 %0, %1, %0")
DEFKERNEL(or_b32, "v_or_b32_e32 %0, %1, %0")
DEFKERNEL(sub_u32, "v_sub_u32_e32 %0, %1, %0")
DEFKERNEL(lshlrev, "v_lshlrev_b32_e32 %0, 3, %0")
DEFKERNEL(lshrrev, "v_lshrrev_b32_e32 %0, 3, %0")
DEFKERNEL(mov_b32, "v_mov_b32_e32 %0, %1")
DEFKERNEL(max_u32, "v_max_u32_e32 %0, %1, %0")
DEFKERNEL(mul_lo_u32, "v_mul_lo_u32 %0, %1, %0")
DEFKERNEL(add_f32, "v_add_f32_e32 %0, %1, %0")
DEFKERNEL(mul_f32, "v_mul_f32_e32 %0, %1, %0")
DEFKERNEL(rndne_f32, "v_rndne_f32_e32 %0, %0")
DEFKERNEL(cvt_i32_f32, "v_cvt_i32_f32_e32 %0, %0")
DEFKERNEL(cvt_f32_i32, "v_cvt_f32_i32_e32 %0, %0")
DEFKERNEL(lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
DEFKERNEL(add_lshl, "v_add_lshl_u32 %0, %0, %1, 2")

#define DEFKERNEL64(NAME, ASM)                                                            \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, int iters, uint32_t seed) \
    {                                                                                     \
        double a[8];                                                                      \
        for (int i = 0; i < 8; ++i) a[i] = 1.0 + 1e-9 * (seed * (threadIdx.x + 1) + i);   \
        double t = 1.0 + 1e-12 * (seed ^ blockIdx.x), u = 1e-13;                          \
        for (int it = 0; it < iters; ++it) {                                              \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                               \
                _Pragma("unroll") for (int w = 0; w < 8; ++w)                             \
                    asm volatile(ASM : "+v"(a[w]) : "v"(t), "v"(u));                      \
            }                                                                             \
        }                                                                                 \
        double s = 0;                                                                     \
        for (int i = 0; i < 8; ++i) s += a[i];                                            \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s;                                \
    }
DEFKERNEL64(mul_f64, "v_mul_f64 %0, %1, %0")
DEFKERNEL64(add_f64, "v_add_f64 %0, %1, %0")
DEFKERNEL64(fma_f64, "v_fma_f64 %0, %1, %2, %0")

__global__ __launch_bounds__(256) void k_cvt_f32_f64(uint32_t* out, int iters, uint32_t seed)
{
    double a[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0 + 1e-9 * (seed * (threadIdx.x + 1) + i); f[i] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int w = 0; w < 8; ++w) asm volatile("v_cvt_f32_f64_e32 %0, %1" : "=v"(f[w]) : "v"(a[w]));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s;
}
__global__ __launch_bounds__(256) void k_cvt_f64_i32(uint32_t* out, int iters, uint32_t seed)
{
    double a[8];
    int f[8];
    for (int i = 0; i < 8; ++i) { f[i] = seed * (threadIdx.x + 1) + i; a[i] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int w = 0; w < 8; ++w) asm volatile("v_cvt_f64_i32_e32 %0, %1" : "=v"(a[w]) : "v"(f[w]));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s;
}


// One operand in a scalar register (wave-uniform), as the Hamming kernel's train row is.
#define DEFKERNEL_S(NAME, ASM)                                                            \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, int iters, uint32_t seed) \
    {                                                                                     \
        uint32_t a[8];                                                                    \
        for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u;    \
        uint32_t t = __builtin_amdgcn_readfirstlane(seed ^ blockIdx.x);                   \
        for (int it = 0; it < iters; ++it) {                                              \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                               \
                _Pragma("unroll") for (int w = 0; w < 8; ++w)                             \
                    asm volatile(ASM : "+v"(a[w]) : "s"(t));                              \
            }                                                                             \
        }                                                                                 \
        uint32_t s = 0;                                                                   \
        for (int i = 0; i < 8; ++i) s += a[i];                                            \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                          \
    }
DEFKERNEL_S(xor_sgpr, "v_xor_b32_e32 %0, %1, %0")
DEFKERNEL_S(add_sgpr, "v_add_u32_e32 %0, %1, %0")
DEFKERNEL_S(mov_sgpr, "v_mov_b32_e32 %0, %1")
DEFKERNEL_S(xor_lit, "v_xor_b32_e32 %0, 0x12345678, %0")
DEFKERNEL_S(xor_inl, "v_xor_b32_e32 %0, 17, %0")
DEFKERNEL_S(bcnt_sgpr, "v_bcnt_u32_b32 %0, %1, %0")

typedef void (*kern_t)(uint32_t*, int, uint32_t);

int run(const char* name, kern_t k, int ops_per_stmt, uint32_t* d)
{
    const int blocks = 256 * 8, iters = 10000;
    hipEvent_t a, b;
    HIPOK(hipEventCreate(&a));
    HIPOK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        HIPOK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
        HIPOK(hipEventRecord(b));
        HIPOK(hipEventSynchronize(b));
        float ms = 0;
        HIPOK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    double instr = (double)blocks * 4 * iters * 32.0 * ops_per_stmt;  // wave-instructions
    double per_simd_ns = best * 1e6 / (instr / 1024.0);
    printf("%-14s %8.3f ms  %7.2f T lane-ops/s  %6.3f ns per wave-instr per SIMD\n", name, best,
           instr * 64 / best / 1e9, per_simd_ns);
    return 0;
}

#define RUN(NAME, N) if (run(#NAME, k_##NAME, N, d)) return 1;

int main()
{
    uint32_t* d;
    HIPOK(hipMalloc(&d, (size_t)256 * 8 * 256 * 4));
    RUN(xor_e32, 1) RUN(xor_e64, 1) RUN(add_e32, 1) RUN(min_u32, 1) RUN(bcnt, 1) RUN(bcnt0, 1) RUN(add3, 1)
    RUN(lshl_or, 1) RUN(bfi, 1) RUN(and_or, 1) RUN(med3, 1) RUN(min3, 1) RUN(max3, 1) RUN(perm, 1)
    RUN(alignbyte, 1) RUN(mad_u24, 1) RUN(mul_u24, 1) RUN(sad_u8, 1) RUN(pk_add_u16, 1) RUN(pk_min_u16, 1)
    RUN(pk_max_u16, 1) RUN(pk_sub_i16, 1) RUN(dot4_u8, 1) RUN(cmp_gt, 2) RUN(cmp_sgpr, 1) RUN(cndmask, 1)
    RUN(fma_f32, 1) RUN(bfe, 1) RUN(mov_dpp, 1) RUN(add_sdwa, 1) RUN(cmp_sdwa, 1)
    RUN(and_b32, 1) RUN(or_b32, 1) RUN(sub_u32, 1) RUN(lshlrev, 1) RUN(lshrrev, 1) RUN(mov_b32, 1) RUN(max_u32, 1)
    RUN(mul_lo_u32, 1) RUN(add_f32, 1) RUN(mul_f32, 1) RUN(rndne_f32, 1) RUN(cvt_i32_f32, 1) RUN(cvt_f32_i32, 1)
    RUN(lshl_add, 1) RUN(add_lshl, 1) RUN(mul_f64, 1) RUN(add_f64, 1) RUN(fma_f64, 1)
    RUN(cvt_f32_f64, 1) RUN(cvt_f64_i32, 1)
    RUN(xor_sgpr, 1) RUN(add_sgpr, 1) RUN(mov_sgpr, 1) RUN(xor_lit, 1) RUN(xor_inl, 1) RUN(bcnt_sgpr, 1)
    return 0;
}
