// C-ABI surface that is not a kernel: versioning, thread-local error text, GEMM argument checks.
#include "aim_common.h"
#include "aim_kernels_internal.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void aim_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int aim_version(void) { return AIM_ABI_VERSION; }
extern "C" const char* aim_last_error(void) { return g_err; }

extern "C" int aim_gemm_bf16(const aim_gemm_args* args, int epilogue, int batch, void* stream) {
    AIM_CHECK_ARG(args != nullptr, "gemm: null args");
    AIM_CHECK_ARG(batch >= 1, "gemm: batch must be >= 1");
    return aim_gemm_launch(*args, epilogue, batch, (hipStream_t)stream);
}

extern "C" int aim_gemm_expsum_tiles(int M, int N) {
    return aim_expsum_use256(M, N) ? 8 : ((M + 127) / 128) * ((N + 127) / 128);
}
