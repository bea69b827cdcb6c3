// Semantics check of v_permlane32_swap / v_permlane16_swap as used by gemm_epilogue.h::epi_swap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    f32x4 a, b;
    for (int e = 0; e < 4; ++e) { a[e] = lane * 10 + e; b[e] = 1000 + lane * 10 + e; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = a[e], y = b[e];
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
        a[e] = x;
        b[e] = y;
    }
    for (int e = 0; e < 4; ++e) { out[lane * 8 + e] = a[e]; out[lane * 8 + 4 + e] = b[e]; }
}
int main() {
    float* d; hipMalloc(&d, 64 * 8 * 4);
    k<<<1, 64>>>(d);
    float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 16, 17, 32, 33, 48, 49}) {
        printf("lane %2d:", l);
        for (int i = 0; i < 8; ++i) printf(" %5.0f", h[l * 8 + i]);
        printf("\n");
    }
    return 0;
}
