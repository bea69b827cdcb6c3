// Probe (tools only): operand lane map of v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 (e4m3) operands and unit block
// scales, checked with exact small-integer data against a host reference.  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// e4m3fn encode of small integers / halves (exact)
__host__ __device__ inline uint8_t e4m3(float v) {
    if (v == 0.f) return 0;
    uint8_t s = v < 0 ? 0x80 : 0;
    float a = fabsf(v);
    int e = (int)floorf(log2f(a));
    float m = a / exp2f((float)e) - 1.0f;          // [0,1)
    int mi = (int)roundf(m * 8.0f);
    if (mi == 8) { mi = 0; e += 1; }
    return s | (uint8_t)(((e + 7) & 15) << 3) | (uint8_t)mi;
}

__global__ void probe(const uint8_t* A, const uint8_t* B, float* C, int variant, unsigned sa, unsigned sb) {
    // A [16][128], B [16][128] (both K-contiguous); C[m][n] = sum_k A[m][k] B[n][k]
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    i32x8 a, b;
    for (int j = 0; j < 8; ++j) {
        int k0 = variant == 0 ? q * 32 + j * 4                       // lane holds 32 consecutive k: block q
                              : (j >> 2) * 64 + q * 16 + (j & 3) * 4; // two 16-byte halves, 64 apart
        a[j] = *(const int*)(A + r * 128 + k0);
        b[j] = *(const int*)(B + r * 128 + k0);
    }
    f32x4 c = {0, 0, 0, 0};
    // first operand = the one whose row index lands on the lane of the result (we pass B first, like the bf16 kernels)
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, c, 0, 0, 0, sb, 0, sa);
    for (int e = 0; e < 4; ++e) C[lane * 4 + e] = c[e];
}

int main() {
    uint8_t hA[16 * 128], hB[16 * 128];
    float fA[16 * 128], fB[16 * 128];
    for (int i = 0; i < 16 * 128; ++i) {
        fA[i] = (float)((i * 7 + (i >> 7) * 3) % 9 - 4);           // -4..4
        fB[i] = (float)((i * 5 + (i >> 7) * 11) % 7 - 3) * 0.5f;   // -1.5..1.5
        hA[i] = e4m3(fA[i]); hB[i] = e4m3(fB[i]);
    }
    float ref[16][16];
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 128; ++k) s += fA[m * 128 + k] * fB[n * 128 + k]; ref[m][n] = s; }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 64 * 4 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 2; ++variant)
        for (unsigned sc : {0x7F7F7F7Fu, 0x80808080u, 0x7F7F7F80u}) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, variant, sc, 0x7F7F7F7Fu);
            float hC[256];
            hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
            // try both output maps: (row m = lane&15 | col n = (lane>>4)*4+e) and the transpose
            int ok1 = 1, ok2 = 1; double r1 = 0, r2 = 0;
            for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
                int i = l & 15, j = (l >> 4) * 4 + e;
                if (hC[l * 4 + e] != ref[i][j]) ok1 = 0;
                if (hC[l * 4 + e] != ref[j][i]) ok2 = 0;
                if (ref[i][j] != 0) r1 = hC[l * 4 + e] / ref[i][j];
                if (ref[j][i] != 0) r2 = hC[l * 4 + e] / ref[j][i];
            }
            printf("variant %d scaleA 0x%08x: C[m=lane&15][n=4q+e] %s (ratio %.3f) | C[m=4q+e][n=lane&15] %s (ratio %.3f)\n", variant, sc,
                   ok1 ? "EXACT" : "no", r1, ok2 ? "EXACT" : "no", r2);
        }
    return 0;
}
