// Internal glue between the public C ABI (include/aim_kernels.h) and the kernel translation units.
#pragma once
#include "../../include/aim_kernels.h"
#include <hip/hip_runtime.h>

typedef aim_gemm_args GemmArgs;
enum { EPI_BF16 = AIM_EPI_BF16, EPI_ACT = AIM_EPI_ACT, EPI_DACT = AIM_EPI_DACT, EPI_F32 = AIM_EPI_F32, EPI_EXPSUM = AIM_EPI_EXPSUM };
enum { ACT_QGELU = AIM_ACT_QGELU, ACT_GELU = AIM_ACT_GELU };

int aim_gemm_launch(const GemmArgs& g, int epi, int batch, hipStream_t st);
int aim_gemm256_launch(const GemmArgs& g, int epi, hipStream_t st);
