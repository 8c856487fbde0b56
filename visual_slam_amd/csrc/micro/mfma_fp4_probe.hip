#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// A: 32 rows x 64 nibbles (32 bytes per row), B: 32 cols x 64 nibbles; both [row][k] nibble-packed.
__global__ void probe(const uint8_t* A, const uint8_t* B, const float* Cin, float* D, int scale)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v8i a = {0,0,0,0,0,0,0,0}, b = {0,0,0,0,0,0,0,0};
    const int* ap = (const int*)(A + r * 32 + h * 16);
    const int* bp = (const int*)(B + r * 32 + h * 16);
    for (int i = 0; i < 4; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
    v16f c;
    for (int i = 0; i < 16; ++i) c[i] = Cin[l * 16 + i];
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, scale, 0, scale);
    for (int i = 0; i < 16; ++i) D[l * 16 + i] = c[i];
}

int main()
{
    uint8_t hA[32 * 32], hB[32 * 32];
    int bitA[32][64], bitB[32][64];
    srand(5);
    for (int r = 0; r < 32; ++r)
        for (int k = 0; k < 64; ++k) { bitA[r][k] = rand() & 1; bitB[r][k] = (rand() >> 3) & 1; }
    for (int r = 0; r < 32; ++r)
        for (int k = 0; k < 64; k += 2) {
            hA[r * 32 + k / 2] = (0x2 | (bitA[r][k] << 3)) | ((0x2 | (bitA[r][k + 1] << 3)) << 4);
            hB[r * 32 + k / 2] = (0x2 | (bitB[r][k] << 3)) | ((0x2 | (bitB[r][k + 1] << 3)) << 4);
        }
    float hC[64 * 16], hD[64 * 16];
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 16; ++i) {
            int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
            hC[l * 16 + i] = -256.0f + (float)(8191 - row * 37 - (l & 31)) / 4096.0f;
        }
    uint8_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    const int scales[3] = {0x7F7F7F7F, 0, 0x7F};
    for (int s = 0; s < 3; ++s) {
        probe<<<1, 64>>>(dA, dB, dC, dD, scales[s]);
        if (hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost) != hipSuccess) { printf("fail\n"); return 1; }
        int bad = 0, badT = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 16; ++i) {
                int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
                int dot = 0, dotT = 0;
                for (int k = 0; k < 64; ++k) {
                    dot += (bitA[row][k] ? -1 : 1) * (bitB[col][k] ? -1 : 1);
                    dotT += (bitA[col][k] ? -1 : 1) * (bitB[row][k] ? -1 : 1);
                }
                float e = hC[l * 16 + i] + (float)dot, eT = hC[l * 16 + i] + (float)dotT;
                if (hD[l * 16 + i] != e) ++bad;
                if (hD[l * 16 + i] != eT) ++badT;
            }
        printf("scale %08x: mismatches %d (transposed hypothesis %d); D[0]=%.6f expect-in %.6f\n", scales[s], bad, badT, hD[0], hC[0]);
    }
    return 0;
}
