// bf16 MFMA GEMM for FEW rows (M < 1024): the class-token chain of the AIM block -- temporal attention projections,
// T_Adapter / S_Adapter, the collapsed cross-attention (reference vit_clip.py:220-229,265,272-275) and their dgrads -- a
// dozen dependent launches per block on B*T rows (512 in training at 64 clips, 96 at one sample x 3 views of ViT-L/14).
// gfx950 only.  Same contract and epilogues as gemm.hip (C = A W^T, both K-contiguous; gemm_epilogue.h::store_frag).
//
// These launches are LATENCY, not throughput: a 128 x 128 tile with a `vmcnt(0)` + barrier per K-step gives a 512 x 768 x 768
// problem 24 workgroups that each wait for 12 dependent loads (~20 us; tools/inf_batch_probe.py: the chain's span is 318 us per
// ViT-L/14 block and the main stream waits for it at small batches).  Here the tile is 64 x 64 (4x the workgroups), and the
// K-loop keeps THREE K-steps in flight in a four-stage LDS ring with counted waits and raw barriers, so a workgroup's time
// is ~ one load latency + K/64 short steps instead of K/64 latencies.
//
// 4 waves as 2 (M) x 2 (N), wave tile 32 x 32 = 2 x 2 MFMA 16x16x32 tiles.  Stage = A 64 rows + W 64 rows of 128 B (the
// XOR-swizzled image of aim_common.h), 16 KiB; 4 stages = 64 KiB, two workgroups per CU.  Per K-step and wave: 4 LDS-DMA
// pieces, 8 fragment reads, 8 MFMAs.  K must be a multiple of 64 (the dispatcher sends other K to gemm.hip).
#include "aim_common.h"
#include "aim_kernels_internal.h"
#include "gemm_epilogue.h"

namespace {

constexpr int GS_STAGE = 2 * 64 * 128;      // A tile + W tile
constexpr int GS_LDS = 4 * GS_STAGE;

// fp8 outputs of the tail launches of aim_gemm_fp8 (capi.hip): acc * wscale[n] + bias, then the row factor (BF16) or the bf16
// residual stream's update (RES16), the arithmetic of wave_epilogue<EPI, true> spelled out as the fmas it compiles to.
template <int EPI>
__device__ __forceinline__ void store_frag_f8(const GemmArgs& g, const f32x4& acc, int m, int n, const RowFactors& rf) {
    const f32x4 ws = g.wscale ? *(const f32x4*)(g.wscale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 b = g.bias ? *(const f32x4*)(g.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc[e], ws[e], b[e]);
    bf16_t* op = (bf16_t*)g.out + (long long)m * g.ldo + n;
    if constexpr (EPI == EPI_BF16) {
        *(bf16x4*)op = pack4(rf.rs * v[0], rf.rs * v[1], rf.rs * v[2], rf.rs * v[3]);
    } else {      // EPI_RES16: out = resid + rs (acc + bias) + bt[tok] vec[frame], rounded to bf16 once
        const bf16x4 res = *(const bf16x4*)((const bf16_t*)g.resid + (long long)m * g.ldr + n);
        const f32x4 vv = g.vec ? *(const f32x4*)(g.vec + (long long)rf.frame * g.ldv + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf(rf.vs, vv[e], __builtin_fmaf(rf.rs, v[e], (float)res[e]));
        *(bf16x4*)op = pack4(y[0], y[1], y[2], y[3]);
    }
}

// F8: operands are fp8 e4m3 bytes -- a K-step is still one 128-byte LDS row per operand row (128 elements), multiplied by ONE
// v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 tile (unit block scales), like gemm256_kernel<EPI, true>.
template <int EPI, bool F8 = false>
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmArgs g) {
    constexpr int ES = F8 ? 1 : 2;                // operand element bytes
    constexpr int KT = 128 / ES;                  // elements per K-step
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    AIM_LDS char* smem = (AIM_LDS char*)smem_raw;
    const int tiles_n = (g.N + 63) >> 6;
    const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int batch = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;
    const int m0 = tm * 64, n0 = tn * 64;
    const char* Ab = (const char*)g.A + ((long long)batch * g.strideA + (long long)m0 * g.lda) * ES;
    const char* Wb = (const char*)g.W + ((long long)batch * g.strideW + (long long)n0 * g.ldw) * ES;
    const int rowsA = g.M - m0, rowsW = g.N - n0;
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(Ab, ((long long)(rowsA - 1) * g.lda + g.K) * ES);
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(Wb, ((long long)(rowsW - 1) * g.ldw + g.K) * ES);
    // staging: pieces of 8 rows x 128 B; a wave stages pieces {2 wave, 2 wave + 1} of A and of W
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    unsigned voA[2], voW[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave * 2 + j) * 8 + srow;
        voA[j] = r < rowsA ? (unsigned)(r * g.lda * ES + schunk * 16) : AIM_OOB;
        voW[j] = r < rowsW ? (unsigned)(r * g.ldw * ES + schunk * 16) : AIM_OOB;
    }
    const int nk = g.K / KT;
    auto stage = [&](int slot, int kt) {
        const bool live = kt < nk;                      // steps past K: zero fill, still counted
        const unsigned k0b = (unsigned)(kt * 128);
        AIM_LDS char* dA = smem + slot * GS_STAGE + wave * 2048;
        AIM_LDS char* dW = dA + 64 * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            stage_piece(rA, dA + j * 1024, live ? voA[j] + k0b : AIM_OOB);
            stage_piece(rW, dW + j * 1024, live ? voW[j] + k0b : AIM_OOB);
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    stage(1, 1);
    stage(2, 2);
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_waitcnt(0x0F78);             // vmcnt(8): all but the two newest stages -- step kt has landed
        __builtin_amdgcn_s_barrier();
        stage((kt + 3) & 3, kt + 3);                    // the slot step kt - 1 was read from (every wave is past those reads)
        const AIM_LDS char* sA = smem + (kt & 3) * GS_STAGE;
        const AIM_LDS char* sW = sA + 64 * 128;
        if constexpr (F8) {
            typedef __attribute__((ext_vector_type(8))) int i32x8;
            typedef __attribute__((ext_vector_type(4))) int i32x4;
            i32x8 af8[2], wf8[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af8[i].lo = *(const AIM_LDS i32x4*)(sA + swz_off(wm * 32 + i * 16 + frow, fq));
                af8[i].hi = *(const AIM_LDS i32x4*)(sA + swz_off(wm * 32 + i * 16 + frow, 4 + fq));
                wf8[i].lo = *(const AIM_LDS i32x4*)(sW + swz_off(wn * 32 + i * 16 + frow, fq));
                wf8[i].hi = *(const AIM_LDS i32x4*)(sW + swz_off(wn * 32 + i * 16 + frow, 4 + fq));
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf8[j], af8[i], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0,
                                                                                 0x7F7F7F7F);
        } else {
            bf16x8 af[2][2], wf[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i][ks] = lds_read8(sA + swz_off(wm * 32 + i * 16 + frow, ks * 4 + fq));
                    wf[i][ks] = lds_read8(sW + swz_off(wn * 32 + i * 16 + frow, ks * 4 + fq));
                }
            __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], af[i][ks], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill stages past K
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 32 + i * 16 + frow;
        if (m >= g.M) continue;
        const RowFactors rf = row_factors(g, m);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 32 + j * 16 + fq * 4;
            if (n >= g.N) continue;
            if constexpr (F8) store_frag_f8<EPI>(g, acc[i][j], m, n, rf);
            else store_frag<EPI>(g, acc[i][j], m, n, rf);
        }
    }
}

template <int EPI, bool F8 = false>
int launch_small(const GemmArgs& g, int batch, hipStream_t st) {
    const int tiles = ((g.M + 63) / 64) * ((g.N + 63) / 64);
    hipLaunchKernelGGL((gemm_small_kernel<EPI, F8>), dim3(tiles, batch), dim3(256), GS_LDS, st, g);
    AIM_CHECK_LAUNCH(F8 ? "aim_gemm_fp8(small)" : "aim_gemm_bf16(small)");
    return 0;
}

}  // namespace

int aim_gemm_small_launch(const GemmArgs& g, int epi, int batch, hipStream_t st) {
    AIM_CHECK_ARG((long long)64 * g.lda * 2 < 0x7fffffffLL && (long long)64 * g.ldw * 2 < 0x7fffffffLL, "gemm(small): leading dimension too large");
    switch (epi) {
        case EPI_BF16: return launch_small<EPI_BF16>(g, batch, st);
        case EPI_ACT: return launch_small<EPI_ACT>(g, batch, st);
        case EPI_DACT: return launch_small<EPI_DACT>(g, batch, st);
        case EPI_F32: return launch_small<EPI_F32>(g, batch, st);
    }
    aim_set_error("gemm(small): unsupported epilogue %d", epi);
    return 1;
}

// fp8 operands (K a multiple of 128): the tail launches of aim_gemm_fp8's thin-last-round peel
int aim_gemm_small_fp8_launch(const GemmArgs& g, int epi, hipStream_t st) {
    AIM_CHECK_ARG((g.K % 128) == 0 && (long long)64 * g.lda < 0x7fffffffLL && (long long)64 * g.ldw < 0x7fffffffLL,
                  "gemm_fp8(small): K must be a multiple of 128 (K=%d)", g.K);
    switch (epi) {
        case EPI_BF16: return launch_small<EPI_BF16, true>(g, 1, st);
        case EPI_RES16: return launch_small<EPI_RES16, true>(g, 1, st);
    }
    aim_set_error("gemm_fp8(small): unsupported epilogue %d", epi);
    return 1;
}
