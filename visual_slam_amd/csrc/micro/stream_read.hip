// Read-bandwidth ceilings on MI355X for the access shapes the FAST kernel could use.
//   flat     every workgroup reads one contiguous block, 16 bytes per lane and load
//   tile     every workgroup reads ROWS rows of 256 bytes at a row stride of W bytes (a 256-pixel-wide
//            image tile with halo rows re-read by the neighbour), 16 bytes per lane and load
// Prints GB/s of unique bytes.  Build: make micro; run: build/stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LOADS>
__global__ __launch_bounds__(256) void flat_kernel(const uint4* __restrict__ src, uint32_t* __restrict__ sink)
{
    const size_t base = (size_t)blockIdx.x * (256 * LOADS) + threadIdx.x;
    uint4 v[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) v[i] = src[base + (size_t)i * 256];
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) acc |= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (acc == 0x12345678u) sink[0] = acc;
}

// tile: W bytes per image row, tiles of 256 x TH outputs, TH + 6 rows read.
//   MODE 0: loads only.  MODE 1: the workgroup also owns LDSB bytes of LDS (limits the resident
//   workgroups) but stores one word.  MODE 2: every loaded value is stored to LDS, then a barrier and
//   one LDS read per lane (the shape of "stage the tile, then evaluate").
template <int TH, int MODE = 0, int LDSB = 0>
__global__ __launch_bounds__(256) void tile_kernel(const uint8_t* __restrict__ img, int H, int W, int ntc, int ntr,
                                                   uint32_t* __restrict__ sink)
{
    constexpr int ROWS = TH + 6, GROUPS = (ROWS + 3) / 4, GPW = (GROUPS + 3) / 4;
    __shared__ uint4 lds[MODE == 0 ? 1 : (LDSB > GPW * 4096 ? LDSB : GPW * 4096) / 16];
    const uint32_t item = blockIdx.x;
    const int bx = item % ntc, by = (item / ntc) % ntr, f = item / (ntc * ntr);
    const uint8_t* fr = img + (size_t)f * H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = bx * 256 + 16 * (lane & 15);
    if (x >= W) x = W - 16;
    uint4 v[GPW];
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
        int y = by * TH - 3 + 4 * (wave + 4 * gi) + (lane >> 4);
        y = y < 0 ? 0 : (y < H ? y : H - 1);
        v[gi] = *reinterpret_cast<const uint4*>(fr + (size_t)y * W + x);
    }
    uint32_t acc = 0;
    if (MODE == 2) {
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) lds[(wave + 4 * gi) * 64 + lane] = v[gi];
        __syncthreads();
        acc = lds[(threadIdx.x * 7) % (GPW * 256)].x;
    } else {
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) acc |= v[gi].x ^ v[gi].y ^ v[gi].z ^ v[gi].w;
        if (MODE == 1) { lds[threadIdx.x].x = acc; acc = lds[threadIdx.x ^ 1].y; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}


// Halo variants (all store what they load to LDS and pass one barrier):
//   HALO 0  body only: 256-byte rows
//   HALO 1  body + one extra dword load per lane in waves 1..3: the dwords left and right of each row
//           (64 different cache lines per instruction)
//   HALO 2  288-byte rows (the 16-byte segments left and right of the body ride in the same loads)
//   HALO 3  256-byte rows that overlap the neighbour tiles: 224 outputs per tile, 9 tile columns
__device__ __forceinline__ uint32_t xcd_item()
{
    const uint32_t total = gridDim.x, id = blockIdx.x;
    const uint32_t qd = total >> 3, rm = total & 7u, xcd = id & 7u, pos = id >> 3;
    return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + pos;
}

template <int TH, int HALO, bool XCD>
__global__ __launch_bounds__(256) void halo_kernel(const uint8_t* __restrict__ img, int H, int W, int ntc, int ntr,
                                                   uint32_t* __restrict__ sink)
{
    constexpr int ROWS = TH + 6, GROUPS = (ROWS + 3) / 4, GPW = (GROUPS + 3) / 4;
    constexpr int SEGS = ROWS * 18, SPT = (SEGS + 255) / 256;
    __shared__ uint4 lds[(HALO == 2 ? SPT * 256 : GPW * 256) + 64];
    const uint32_t item = XCD ? xcd_item() : blockIdx.x;
    const int bx = item % ntc, by = (item / ntc) % ntr, f = item / (ntc * ntr);
    const uint8_t* fr = img + (size_t)f * H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    if (HALO == 2) {
        uint4 v[SPT];
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int row = idx / 18, seg = idx - 18 * row;
            int x = bx * 256 - 16 + 16 * seg;
            x = x < 0 ? 0 : (x < W ? x : W - 16);
            int y = by * TH - 3 + row;
            y = y < 0 ? 0 : (y < H ? y : H - 1);
            v[i] = *reinterpret_cast<const uint4*>(fr + (size_t)y * W + x);
        }
#pragma unroll
        for (int i = 0; i < SPT; ++i) lds[threadIdx.x + 256 * i] = v[i];
    } else {
        const int xs = HALO == 3 ? bx * 224 - 16 : bx * 256;
        int x = xs + 16 * (lane & 15);
        x = x < 0 ? 0 : (x < W ? x : W - 16);
        uint4 v[GPW];
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) {
            int y = by * TH - 3 + 4 * (wave + 4 * gi) + (lane >> 4);
            y = y < 0 ? 0 : (y < H ? y : H - 1);
            v[gi] = *reinterpret_cast<const uint4*>(fr + (size_t)y * W + x);
        }
        uint32_t h = 0;
        if (HALO == 1) {
            int hy = by * TH - 3 + 32 * (wave > 0 ? wave - 1 : 0) + (lane >> 1);
            hy = hy < 0 ? 0 : (hy < H ? hy : H - 1);
            int hx = (lane & 1) ? bx * 256 + 256 : bx * 256 - 4;
            hx = hx < 0 ? 0 : (hx < W ? hx : W - 4);
            h = *reinterpret_cast<const uint32_t*>(fr + (size_t)hy * W + hx);
        }
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) lds[(wave + 4 * gi) * 64 + lane] = v[gi];
        if (HALO == 1) lds[GPW * 256 + (threadIdx.x >> 2)].x = h;
    }
    __syncthreads();
    acc = lds[(threadIdx.x * 7) % (GPW * 256)].x;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F>
static float time_ms(F launch, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const int H = 1080, W = 1920, frames = 1000;
    const size_t bytes = (size_t)H * W * frames;
    uint8_t* img;
    uint32_t* sink;
    CK(hipMalloc(&img, bytes + 65536));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(img, 1, bytes + 65536));
    const double gb = bytes / 1e9;
    {
        const unsigned nb = (unsigned)(bytes / (256 * 16 * 2));
        float ms = time_ms([&] { hipLaunchKernelGGL(flat_kernel<2>, dim3(nb), dim3(256), 0, 0, (const uint4*)img, sink); }, 10);
        printf("flat  2 x 16 B per lane   %8.3f ms  %7.1f GB/s\n", ms, gb / ms * 1e3);
    }
    {
        const unsigned nb = (unsigned)(bytes / (256 * 16 * 4));
        float ms = time_ms([&] { hipLaunchKernelGGL(flat_kernel<4>, dim3(nb), dim3(256), 0, 0, (const uint4*)img, sink); }, 10);
        printf("flat  4 x 16 B per lane   %8.3f ms  %7.1f GB/s\n", ms, gb / ms * 1e3);
    }
    {
        const unsigned nb = (unsigned)(bytes / (256 * 16 * 8));
        float ms = time_ms([&] { hipLaunchKernelGGL(flat_kernel<8>, dim3(nb), dim3(256), 0, 0, (const uint4*)img, sink); }, 10);
        printf("flat  8 x 16 B per lane   %8.3f ms  %7.1f GB/s\n", ms, gb / ms * 1e3);
    }
    {
        const unsigned nb = (unsigned)(bytes / (256 * 16 * 16));
        float ms = time_ms([&] { hipLaunchKernelGGL(flat_kernel<16>, dim3(nb), dim3(256), 0, 0, (const uint4*)img, sink); }, 10);
        printf("flat 16 x 16 B per lane   %8.3f ms  %7.1f GB/s\n", ms, gb / ms * 1e3);
    }
    const int ntc = (W + 255) / 256;
#define TILE(TH, MODE, LDSB)                                                                                        \
    {                                                                                                               \
        const int ntr = (H + TH - 1) / TH;                                                                          \
        const unsigned nb = (unsigned)(ntc * ntr * frames);                                                         \
        float ms = time_ms([&] { hipLaunchKernelGGL((tile_kernel<TH, MODE, LDSB>), dim3(nb), dim3(256), 0, 0, img, H, W, ntc, ntr, sink); }, 10); \
        printf("tile 256 x %3d (+6 rows) mode %d lds %6d  %8.3f ms  %7.1f GB/s\n", TH, MODE, LDSB, ms, gb / ms * 1e3); \
    }
    TILE(32, 0, 0) TILE(64, 0, 0) TILE(128, 0, 0) TILE(256, 0, 0)
    TILE(32, 1, 16000) TILE(64, 1, 25600) TILE(64, 1, 32000) TILE(64, 1, 40000)
    TILE(64, 1, 60000) TILE(64, 1, 81000) TILE(32, 1, 81000) TILE(32, 1, 60000) TILE(32, 1, 40000)
    TILE(32, 2, 16000) TILE(64, 2, 25600) TILE(64, 2, 20000) TILE(128, 2, 40000)
#define HALO(TH, HL, XCD)                                                                                           \
    {                                                                                                               \
        const int tw = HL == 3 ? 224 : 256;                                                                         \
        const int ntc2 = (W + tw - 1) / tw, ntr = (H + TH - 1) / TH;                                                \
        const unsigned nb = (unsigned)(ntc2 * ntr * frames);                                                        \
        float ms = time_ms([&] { hipLaunchKernelGGL((halo_kernel<TH, HL, XCD>), dim3(nb), dim3(256), 0, 0, img, H, W, ntc2, ntr, sink); }, 10); \
        printf("halo variant %d, tile rows %3d, xcd-contiguous %d  %8.3f ms  %7.1f GB/s\n", HL, TH, (int)XCD, ms, gb / ms * 1e3); \
    }
    HALO(64, 0, false) HALO(64, 0, true) HALO(64, 1, false) HALO(64, 1, true) HALO(64, 2, false) HALO(64, 2, true)
    HALO(64, 3, false) HALO(64, 3, true) HALO(32, 1, true) HALO(32, 2, true) HALO(32, 3, true)
    return 0;
}
