// Which SIMD does wave w of a 256-thread workgroup run on?  Histogram [wave in workgroup][SIMD id] over a grid
// that keeps several workgroups resident per CU (LDS-limited like the tile kernels), with uneven run times so
// that workgroups retire out of order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ __launch_bounds__(256) void probe(unsigned* hist, int spin, int lds_dwords)
{
    extern __shared__ uint32_t lds[];
    const int wave = threadIdx.x >> 6;
    uint32_t id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    const int simd = (id >> 4) & 3;
    // uneven work: a hash of the workgroup index decides how long it spins
    uint32_t h = blockIdx.x * 2654435761u;
    int n = spin * (1 + (h >> 29));
    float a = threadIdx.x;
    for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
    if (a == 12345.0f) lds[threadIdx.x % lds_dwords] = 1;
    if ((threadIdx.x & 63) == 0) atomicAdd(&hist[wave * 4 + simd], 1u);
}

int main(int argc, char** argv)
{
    const int lds_bytes = argc > 1 ? atoi(argv[1]) : 40960;
    unsigned* d;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    hipLaunchKernelGGL(probe, dim3(136000), dim3(256), lds_bytes, 0, d, 200, lds_bytes / 4);
    hipDeviceSynchronize();
    unsigned h[16];
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("LDS per workgroup %d bytes; rows = wave in workgroup, columns = SIMD 0..3\n", lds_bytes);
    for (int w = 0; w < 4; ++w) printf("wave %d: %8u %8u %8u %8u\n", w, h[4 * w], h[4 * w + 1], h[4 * w + 2], h[4 * w + 3]);
    return 0;
}
