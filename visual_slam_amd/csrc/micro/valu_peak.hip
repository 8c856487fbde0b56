// valu_peak.hip -- measures VALU issue rates that bound the Hamming matcher: register-only loops
// of one instruction kind (or the matcher's xor + popcount-accumulate mix), every CU busy with 8
// waves per SIMD.  Prints lane-operations per second; the nominal figure for a full-rate wave64
// instruction is 256 CUs x 4 SIMDs x 32 lanes x clock.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum { K_XOR, K_BCNT, K_ADD, K_FMA, K_MIX, K_MED3, K_COUNT_ };
static const char* kname[] = {"v_xor_b32", "v_bcnt_u32_b32", "v_add_u32", "v_fma_f32", "xor+bcnt mix", "v_med3_u32"};

template <int KIND>
__global__ __launch_bounds__(256) void loop(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u;
    uint32_t t = seed ^ blockIdx.x;
    float fa[8];
    for (int i = 0; i < 8; ++i) fa[i] = (float)a[i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                if (KIND == K_XOR) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[w]) : "v"(t));
                if (KIND == K_BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[w]) : "v"(t));
                if (KIND == K_ADD) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[w]) : "v"(t));
                if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(fa[w]) : "v"(t));
                if (KIND == K_MED3) asm volatile("v_med3_u32 %0, %1, %0, %0" : "+v"(a[w]) : "v"(t));
                if (KIND == K_MIX) {
                    uint32_t x;
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(t), "v"(a[w]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[(w + 4) & 7]) : "v"(x));
                }
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (uint32_t)fa[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
int run(uint32_t* d, int blocks, int iters)
{
    hipEvent_t a, b;
    HIPOK(hipEventCreate(&a));
    HIPOK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        HIPOK(hipEventRecord(a));
        hipLaunchKernelGGL(loop<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
        HIPOK(hipEventRecord(b));
        HIPOK(hipEventSynchronize(b));
        float ms = 0;
        HIPOK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    double per_iter = 32.0 * (KIND == K_MIX ? 2 : 1);
    double ops = (double)blocks * 256 * iters * per_iter;
    printf("%-16s %8.3f ms  %7.2f T lane-ops/s\n", kname[KIND], best, ops / best / 1e9);
    return 0;
}

int main()
{
    const int blocks = 256 * 8, iters = 20000;
    uint32_t* d;
    HIPOK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    if (run<K_XOR>(d, blocks, iters)) return 1;
    if (run<K_BCNT>(d, blocks, iters)) return 1;
    if (run<K_ADD>(d, blocks, iters)) return 1;
    if (run<K_FMA>(d, blocks, iters)) return 1;
    if (run<K_MED3>(d, blocks, iters)) return 1;
    if (run<K_MIX>(d, blocks, iters)) return 1;
    return 0;
}
