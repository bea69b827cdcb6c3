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

template <int EPI>
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmArgs g) {
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
    const bf16_t* Ab = (const bf16_t*)g.A + (long long)batch * g.strideA + (long long)m0 * g.lda;
    const bf16_t* Wb = (const bf16_t*)g.W + (long long)batch * g.strideW + (long long)n0 * g.ldw;
    const int rowsA = g.M - m0, rowsW = g.N - n0;
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(Ab, ((long long)(rowsA - 1) * g.lda + g.K) * 2);
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(Wb, ((long long)(rowsW - 1) * g.ldw + g.K) * 2);
    // staging: pieces of 8 rows x 128 B; a wave stages pieces {2 wave, 2 wave + 1} of A and of W
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    unsigned voA[2], voW[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave * 2 + j) * 8 + srow;
        voA[j] = r < rowsA ? (unsigned)((r * g.lda + schunk * 8) * 2) : AIM_OOB;
        voW[j] = r < rowsW ? (unsigned)((r * g.ldw + schunk * 8) * 2) : AIM_OOB;
    }
    const int nk = g.K >> 6;
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
        bf16x8 af[2][2], wf[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i][ks] = lds_read8(sA + swz_off(wm * 32 + i * 16 + frow, ks * 4 + fq));
                wf[i][ks] = lds_read8(sW + swz_off(wn * 32 + i * 16 + frow, ks * 4 + fq));
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], af[i][ks], acc[i][j], 0, 0, 0);
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
            store_frag<EPI>(g, acc[i][j], m, n, rf);
        }
    }
}

template <int EPI>
int launch_small(const GemmArgs& g, int batch, hipStream_t st) {
    const int tiles = ((g.M + 63) / 64) * ((g.N + 63) / 64);
    hipLaunchKernelGGL(gemm_small_kernel<EPI>, dim3(tiles, batch), dim3(256), GS_LDS, st, g);
    AIM_CHECK_LAUNCH("aim_gemm_bf16(small)");
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
