// bf16 MFMA GEMM  C[m][n] = sum_k A[m][k] * W[n][k]   (both operands K-contiguous, fp32 accumulate)
// with the fused epilogues the AIM block needs.  gfx950 only.
//
// Replaces, on the hot path of reference mmaction/models/backbones/vit_clip.py:
//   q/k/v projections (:132-138), attn.out_proj (:157), mlp.c_fc + QuickGELU + c_proj (:93-97,286),
//   Adapter D_fc1 + GELU + D_fc2 (:62-64), the residual combines (:275,286) and, for autograd,
//   the dgrad of each frozen Linear (weights stored transposed so dgrad is the same NT kernel).
//
// Tiling (v1): 128x128x64 block tile, 256 threads = 4 waves as 2(M) x 2(N), each wave a 64x64
// output tile = 4x4 MFMA 16x16x32 bf16 tiles (64 accumulator VGPRs).  A and W tiles are staged
// HBM -> LDS directly (buffer_load ... lds, 16 B/lane, 1 KiB per wave-instruction) into a
// double-buffered, XOR-swizzled image (aim_common.h: swz_off); the swizzle is applied to the
// per-lane SOURCE address so the LDS side stays lane-linear.  Ragged M/N/K edges are zero-filled
// by the buffer bounds check.  The MFMA is issued "transposed" (first operand = W fragment) so a
// lane owns 4 consecutive output columns of one row: 8-byte bf16 / 16-byte f32 stores.
#include "aim_common.h"
#include "aim_kernels_internal.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;   // 16 KiB per operand per stage

template <int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    AIM_LDS char* smem = (AIM_LDS char*)smem_raw;

    const int tiles_n = (g.N + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int batch = blockIdx.y;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int m0 = tm * BM, n0 = tn * BN;
    const bf16_t* Ab = (const bf16_t*)g.A + (long long)batch * g.strideA + (long long)m0 * g.lda;
    const bf16_t* Wb = (const bf16_t*)g.W + (long long)batch * g.strideW + (long long)n0 * g.ldw;
    const int rowsA = g.M - m0, rowsW = g.N - n0;
    __amdgpu_buffer_rsrc_t rA = make_rsrc(Ab, ((long long)(rowsA - 1) * g.lda + g.K) * 2);
    __amdgpu_buffer_rsrc_t rW = make_rsrc(Wb, ((long long)(rowsW - 1) * g.ldw + g.K) * 2);

    // per-lane staging constants: 4 pieces (8 rows x 128 B) of A and of W per wave per K-step
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    unsigned voffA[4], voffW[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + srow;
        voffA[j] = (r < rowsA) ? (unsigned)((r * g.lda + schunk * 8) * 2) : AIM_OOB;
        voffW[j] = (r < rowsW) ? (unsigned)((r * g.ldw + schunk * 8) * 2) : AIM_OOB;
    }
    const int nk = (g.K + BK - 1) / BK;

    auto stage = [&](int buf, int kt) {
        const int k0 = kt * BK;
        const bool kin = (k0 + schunk * 8) < g.K;
        AIM_LDS char* dA = smem + buf * (2 * TILE_BYTES) + wave * 4096;
        AIM_LDS char* dW = dA + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned va = (kin && voffA[j] != AIM_OOB) ? voffA[j] + (unsigned)(k0 * 2) : AIM_OOB;
            unsigned vw = (kin && voffW[j] != AIM_OOB) ? voffW[j] + (unsigned)(k0 * 2) : AIM_OOB;
            stage_piece(rA, dA + j * 1024, va);
            stage_piece(rW, dW + j * 1024, vw);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    // EXPSUM problems of N = 257 tokens (ViT-L/14) are cut into 3 x 3 tiles of which five hold ONE valid row or column:
    // a wave multiplies only the 16-row / 16-column MFMA tiles that contain a valid element (wave-uniform limits)
    const int ilim = EPI == EPI_EXPSUM ? min(4, (rowsA - wm * 64 + 15) >> 4) : 4;
    const int jlim = EPI == EPI_EXPSUM ? min(4, (rowsW - wn * 64 + 15) >> 4) : 4;

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const AIM_LDS char* sA = smem + (kt & 1) * (2 * TILE_BYTES);
        const AIM_LDS char* sW = sA + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = lds_read8(sA + swz_off(wm * 64 + i * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = lds_read8(sW + swz_off(wn * 64 + j * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (EPI == EPI_EXPSUM) {
                        if (i < ilim && j < jlim)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
                    }
                }
        }
    }

    // ---- epilogue: lane owns row m = .. + (lane&15), columns n = .. + 4*(lane>>4) + {0..3} ----
    if constexpr (EPI == EPI_EXPSUM) {
        // masked (max, sum exp) of scale*acc over the tile -> partial[batch][tile][2]
        __syncthreads();
        AIM_LDS float* red = (AIM_LDS float*)smem;
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + frow;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + fq * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = (m < g.M && n + e < g.N) ? acc[i][j][e] * g.scale : -INFINITY;
                    acc[i][j][e] = v;
                    mx = fmaxf(mx, v);
                }
            }
        }
        mx = wave_max(mx);
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float s = 0.f;
        if (mx > -INFINITY) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s += expf(acc[i][j][e] - mx);
        }
        s = wave_sum(s);
        if (lane == 0) red[4 + wave] = s;
        __syncthreads();
        if (tid == 0) {
            float* po = (float*)g.out + ((long long)batch * gridDim.x + (tm * tiles_n + tn)) * 2;
            po[0] = mx;
            po[1] = red[4] + red[5] + red[6] + red[7];
        }
        return;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + frow;
            if (m >= g.M) continue;
            const RowFactors rf = row_factors(g, m);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + fq * 4;
                if (n >= g.N) continue;
                store_frag<EPI>(g, acc[i][j], m, n, rf);
            }
        }
    }
}

template <int EPI>
int launch(const GemmArgs& g, int batch, hipStream_t st) {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    dim3 grid(tiles, batch), block(256);
    hipLaunchKernelGGL(gemm_kernel<EPI>, grid, block, 4 * TILE_BYTES, st, g);
    AIM_CHECK_LAUNCH("aim_gemm_bf16");
    return 0;
}

}  // namespace

int aim_gemm_peel_rows(const GemmArgs& g, int* M0) {
    static const bool peel_on = [] { const char* e = getenv("AIM_GEMM_PEEL"); return !e || atoi(e) != 0; }();
    if (!peel_on) return 0;
    const int ncu = aim_device_cus();
    const int res = g.reserve_cus > 0 ? g.reserve_cus : 0;
    const int cus = ncu - res > 8 ? ncu - res : 8;
    const int tn = (g.N + 255) / 256, tm = (g.M + 255) / 256;
    const int tiles = tm * tn, full = tiles / cus, left = tiles - full * cus;
    const int lrt = (left + tn - 1) / tn;                    // row tiles that hold the left-over tiles
    if (full >= 2 && left > 0 && lrt <= 2 && left * 8 <= cus && (tm - lrt) * tn <= full * cus) {
        *M0 = (tm - lrt) * 256;
        return g.M - *M0;
    }
    return 0;
}

int aim_gemm_launch(const GemmArgs& g, int epi, int batch, hipStream_t st) {
    AIM_CHECK_ARG(g.M > 0 && g.N > 0 && g.K > 0, "gemm: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
    AIM_CHECK_ARG((g.K % 8) == 0 && (g.lda % 8) == 0 && (g.ldw % 8) == 0, "gemm: K/lda/ldw must be multiples of 8 (K=%d lda=%d ldw=%d)", g.K, g.lda, g.ldw);
    AIM_CHECK_ARG(epi == EPI_EXPSUM || ((g.N % 4) == 0 && (g.ldo % 4) == 0), "gemm: N and ldo must be multiples of 4 (N=%d ldo=%d)", g.N, g.ldo);
    AIM_CHECK_ARG(g.A && g.W && g.out, "gemm: null operand");
    AIM_CHECK_ARG((long long)128 * g.lda * 2 < 0x7fffffffLL && (long long)128 * g.ldw * 2 < 0x7fffffffLL, "gemm: leading dimension too large");
    if (g.af || g.at || g.vec) AIM_CHECK_ARG(g.ntok > 0, "gemm: ntok required with row factors");
    // large-M problems run on the 256x256 pipelined kernel (gemm256.hip); AIM_GEMM_TILE=128 forces this file's.
    // Its epilogue moves 16 bytes per lane: bf16 outputs need N and the leading dimensions in multiples of 8.
    static const int pick = [] { const char* e = getenv("AIM_GEMM_TILE"); return e ? atoi(e) : 0; }();
    // (and K in whole 64-element K-tiles: the 256x256 kernel's staging does not mask a row's K tail)
    const bool wide_ok = (g.N % 8) == 0 && (g.n_split % 8) == 0 && (g.K % 64) == 0 &&
                         (epi == EPI_F32 ? ((g.ldr % 4) == 0 && (g.ldv % 4) == 0 && (!g.vec || g.ntok >= 128))
                                         : ((g.ldo % 8) == 0 && (epi != EPI_ACT || (g.ldo2 % 8) == 0) &&
                                            (epi != EPI_DACT || (g.ldaux % 8) == 0)));
    if (pick != 128 && batch == 1 && epi != EPI_EXPSUM && g.M >= 1024 && g.N >= 64 && wide_ok) {
        AIM_CHECK_ARG(!g.aux_frag || epi == EPI_ACT || epi == EPI_DACT, "gemm: aux_frag is an ACT / DACT option");
        AIM_CHECK_ARG(g.row0 == 0, "gemm: row0 != 0 is for launches of fewer than 1024 rows");
        // A THIN last tile round (ViT-L/14: 2 056 tiles = 8.03 rounds of 256 CUs; every N = 1 024 GEMM of the block) costs the
        // persistent kernel a whole tile time (tools/tail_probe.py: 60-69 of 975 us for c_proj).  Such a launch is cut in two:
        // the rows that fill whole rounds, and the last one or two row tiles on the latency-oriented 64 x 64 kernel
        // (gemm_small.hip: same K order, same epilogue arithmetic -- bit-identical outputs), with row0 carrying the rows'
        // place in the whole problem for the per-frame / per-token factors.  AIM_GEMM_PEEL=0: never.
        int M0 = 0;
        if ((epi == EPI_BF16 || epi == EPI_F32) && aim_gemm_peel_rows(g, &M0) > 0) {
            GemmArgs head = g, tail = g;
            head.M = M0;
            const long long osz = epi == EPI_F32 ? 4 : 2;
            tail.M = g.M - M0;
            tail.row0 = M0;
            tail.A = (const aim_bf16*)((const char*)g.A + (long long)M0 * g.lda * 2);
            tail.out = (char*)g.out + (long long)M0 * g.ldo * osz;
            if (g.resid) tail.resid = g.resid + (long long)M0 * g.ldr;
            tail.reserve_cus = 0;
            if (int rc = aim_gemm256_launch(head, epi, 1, st)) return rc;
            return aim_gemm_small_launch(tail, epi, 1, st);
        }
        return aim_gemm256_launch(g, epi, 1, st);
    }
    if (epi == EPI_EXPSUM && aim_expsum_use256(g.M, g.N) && (g.K % 64) == 0) {
        AIM_CHECK_ARG(!g.xrow || (g.N < 256 && (g.ldx % 8) == 0), "gemm: EXPSUM extra key needs N < 256 and ldx %% 8 == 0");
        AIM_CHECK_ARG(g.ldo == 0 || g.ldo >= (g.xrow ? 32 : 16), "gemm: EXPSUM slot stride ldo=%d is smaller than the tile's own slots", g.ldo);
        return aim_gemm256_launch(g, epi, batch, st);
    }
    AIM_CHECK_ARG(!g.aux_frag, "gemm: aux_frag needs the large-tile kernel (ACT / DACT, batch 1, M >= 1024, N %% 8 == 0, K %% 64 == 0)");
    AIM_CHECK_ARG(!g.xrow, "gemm: `xrow` is only supported by the one-tile-per-item EXPSUM path");
    // very few rows (the class-token chain at one sample x 3 views per call: B*T = 96 rows): the latency-oriented 64 x 64
    // kernel (gemm_small.hip).  Measured (tools/inf_batch_probe.py, interleaved): ViT-L/14 3-view inference 179 -> 210 views/s
    // (fp8), the main stream's wait for the chain 2.7 -> 0.75 ms per step; at B*T = 512 (training, 64 clips) the chain is not on
    // the critical path and the kernel is neutral (1 274 vs 1 277 clips/s), so this file's 128 x 128 kernel keeps those.
    // AIM_GEMM_SMALL=rows moves the threshold (0: never).
    static const int small_rows = [] { const char* e = getenv("AIM_GEMM_SMALL"); return e ? atoi(e) : 256; }();
    if ((g.M <= small_rows || g.row0 != 0) && epi != EPI_EXPSUM && (g.K % 64) == 0) {
        if (epi == EPI_ACT) AIM_CHECK_ARG(g.out2 && (g.ldo2 % 4) == 0, "gemm: ACT epilogue needs out2");
        if (epi == EPI_DACT) AIM_CHECK_ARG(g.aux && (g.ldaux % 4) == 0, "gemm: DACT epilogue needs aux");
        return aim_gemm_small_launch(g, epi, batch, st);
    }
    switch (epi) {
        case EPI_BF16: return launch<EPI_BF16>(g, batch, st);
        case EPI_ACT:
            AIM_CHECK_ARG(g.out2 && (g.ldo2 % 4) == 0, "gemm: ACT epilogue needs out2");
            return launch<EPI_ACT>(g, batch, st);
        case EPI_DACT:
            AIM_CHECK_ARG(g.aux && (g.ldaux % 4) == 0, "gemm: DACT epilogue needs aux");
            return launch<EPI_DACT>(g, batch, st);
        case EPI_F32: return launch<EPI_F32>(g, batch, st);
        case EPI_EXPSUM: return launch<EPI_EXPSUM>(g, batch, st);
    }
    aim_set_error("gemm: unknown epilogue %d", epi);
    return 1;
}
