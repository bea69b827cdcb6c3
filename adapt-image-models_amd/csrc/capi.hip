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

// CUs of the device the calling thread has current: what a persistent kernel sizes its grid by.  Queried per call (a
// table lookup inside the runtime, ~0.1 us), nothing cached: the library keeps no state between calls.  CU-MASKED caller
// streams (hipExtStreamCreateWithCUMask) are NOT part of the ABI: measured in round 2, a masked main stream makes the
// persistent GEMM 41 % slower (the dispatcher still stripes workgroups over engines that now have fewer CUs), and the one
// abort in that experiment's records sat in the masked stream's teardown under the profiler -- a caller that wants CUs kept
// out of a launch passes aim_gemm_args.reserve_cus instead.
int aim_device_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 256;
    }
    return cus > 0 ? cus : 256;
}

extern "C" int aim_version(void) { return AIM_ABI_VERSION; }
extern "C" const char* aim_last_error(void) { return g_err; }

extern "C" int aim_gemm_bf16(const aim_gemm_args* args, int epilogue, int batch, void* stream) {
    AIM_CHECK_ARG(args != nullptr, "gemm: null args");
    AIM_CHECK_ARG(batch >= 1, "gemm: batch must be >= 1");
    return aim_gemm_launch(*args, epilogue, batch, (hipStream_t)stream);
}

extern "C" int aim_gemm_fp8(const aim_gemm_args* args, int epilogue, void* stream) {
    AIM_CHECK_ARG(args != nullptr, "gemm_fp8: null args");
    const GemmArgs& g = *args;
    AIM_CHECK_ARG(g.M >= 1024 && g.N >= 64 && g.K > 0, "gemm_fp8: large-M problems only (M=%d N=%d K=%d)", g.M, g.N, g.K);
    AIM_CHECK_ARG((g.K % 16) == 0 && (g.lda % 16) == 0 && (g.ldw % 16) == 0, "gemm_fp8: K/lda/ldw must be multiples of 16 (K=%d lda=%d ldw=%d)", g.K, g.lda, g.ldw);
    AIM_CHECK_ARG((g.N % 8) == 0 && (g.n_split % 8) == 0 && (g.ldo % 8) == 0, "gemm_fp8: N, n_split and ldo must be multiples of 8");
    AIM_CHECK_ARG((((g.K + 127) / 128) & 1) == 0, "gemm_fp8: ceil(K / 128) must be even (K=%d)", g.K);
    AIM_CHECK_ARG(g.A && g.W && g.out && !g.xrow, "gemm_fp8: null operand / unsupported xrow");
    if (epilogue == EPI_RES16)
        AIM_CHECK_ARG(g.resid && (g.ldr % 8) == 0 && (g.ldv % 4) == 0 && (!g.vec || g.ntok >= 128), "gemm_fp8: RES16 needs a bf16 residual (ldr %% 8 == 0) and ntok >= 128 with vec");
    if (epilogue == EPI_F32) AIM_CHECK_ARG((g.ldr % 4) == 0 && (g.ldv % 4) == 0 && (!g.vec || g.ntok >= 128), "gemm_fp8: F32 epilogue strides");
    if (g.af || g.at || g.vec) AIM_CHECK_ARG(g.ntok > 0, "gemm_fp8: ntok required with row factors");
    AIM_CHECK_ARG(g.row0 == 0, "gemm_fp8: row0 is set by the library's own tail launches only");
    // a thin last tile round goes to the small-tile kernel (gemm.hip, aim_gemm_peel_rows): 12 views x 32 frames of ViT-L/14 are
    // 385.5 row tiles -- 6.03 rounds for out_proj / c_proj, 18.09 for QKV
    int M0 = 0;
    if ((epilogue == EPI_BF16 || epilogue == EPI_RES16) && (g.K % 128) == 0 && (g.N % 4) == 0 && aim_gemm_peel_rows(g, &M0) > 0) {
        GemmArgs head = g, tail = g;
        head.M = M0;
        tail.M = g.M - M0;
        tail.row0 = M0;
        tail.A = (const aim_bf16*)((const char*)g.A + (long long)M0 * g.lda);
        tail.out = (char*)g.out + (long long)M0 * g.ldo * 2;
        if (g.resid) tail.resid = (const float*)((const char*)g.resid + (long long)M0 * g.ldr * 2);      // (RES16: bf16 rows)
        tail.reserve_cus = 0;
        if (int rc = aim_gemm256_fp8_launch(head, epilogue, (hipStream_t)stream)) return rc;
        return aim_gemm_small_fp8_launch(tail, epilogue, (hipStream_t)stream);
    }
    return aim_gemm256_fp8_launch(g, epilogue, (hipStream_t)stream);
}

extern "C" int aim_gemm_expsum_tiles(int M, int N) {
    return aim_expsum_use256(M, N) ? 8 : ((M + 127) / 128) * ((N + 127) / 128);
}
