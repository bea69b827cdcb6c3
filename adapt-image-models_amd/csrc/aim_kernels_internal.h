// Internal glue between the public C ABI (include/aim_kernels.h) and the kernel translation units.
#pragma once
#include <stdlib.h>
#include "../../include/aim_kernels.h"
#include <hip/hip_runtime.h>

typedef aim_gemm_args GemmArgs;
enum { EPI_BF16 = AIM_EPI_BF16, EPI_ACT = AIM_EPI_ACT, EPI_DACT = AIM_EPI_DACT, EPI_F32 = AIM_EPI_F32, EPI_EXPSUM = AIM_EPI_EXPSUM, EPI_ACT8 = AIM_EPI_ACT8, EPI_RES16 = AIM_EPI_RES16 };
enum { ACT_QGELU = AIM_ACT_QGELU, ACT_GELU = AIM_ACT_GELU };

// CUs of the current device (per-call query, no cache; CU-masked caller streams are not supported: capi.hip).  Persistent
// kernels size their grids by it.
int aim_device_cus();
int aim_gemm_launch(const GemmArgs& g, int epi, int batch, hipStream_t st);
int aim_gemm256_launch(const GemmArgs& g, int epi, int nbatch, hipStream_t st);
int aim_gemm256_fp8_launch(const GemmArgs& g, int epi, hipStream_t st);
int aim_gemm_small_launch(const GemmArgs& g, int epi, int batch, hipStream_t st);
int aim_gemm_small_fp8_launch(const GemmArgs& g, int epi, hipStream_t st);
// rows of a thin last tile round that go to the small-tile kernel (0: no peel); fills M0 = rows of the whole rounds
int aim_gemm_peel_rows(const GemmArgs& g, int* M0);
// EXPSUM problems of one 256x256 tile per batch item run on the persistent kernel (8 partial slots per item)
static inline bool aim_expsum_use256(int M, int N) {
    static const bool on = [] { const char* e = getenv("AIM_EXPSUM_256"); return !e || atoi(e) != 0; }();
    return on && (M > 128 || N > 128) && M <= 256 && N <= 256;
}
