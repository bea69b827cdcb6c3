// Weight-gradient GEMM for the trainable adapters:  dW[n][k] += sum_m G[m][n] * A[m][k].  gfx950 only.
//
// Autograd counterpart of Adapter.D_fc1 / D_fc2 (reference vit_clip.py:57-58, 62-64); the reference
// gets it from torch autograd (addmm backward).  Both operands are row-major with the REDUCTION index
// m as the slow dimension, so neither can be read as a k-contiguous MFMA fragment.  The tiles are
// staged untransposed ([m][n] and [m][k], coalesced) into swizzled LDS images and the fragments are
// fetched with ds_read_b64_tr_b16, the hardware transposing read: no transposed copies in HBM.
//
// Block tile 128(n) x 128(k), 4 waves as 2x2, wave tile 64x64 = 4x4 MFMA 16x16x32 tiles.
// The M dimension is split over blockIdx.y in chunks; partial tiles are combined with fp32 atomics
// (order-dependent in the last bits; the reference's cuBLAS split-K has the same property).
#include "aim_common.h"
#include "aim_kernels_internal.h"

namespace {

constexpr int MSTEP = 64;                 // reduction rows per stage
constexpr int HALF_BYTES = MSTEP * 128;   // one [64 m][64 col] image = 8 KiB
constexpr int OPER_BYTES = 2 * HALF_BYTES;  // [64 m][128 col] = two images

__global__ __launch_bounds__(256) void wgrad_kernel(const bf16_t* __restrict__ G, int ldg, const bf16_t* __restrict__ A,
                                                    int lda, float* __restrict__ dW, int lddw, int M, int Nw, int Kw,
                                                    int chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    AIM_LDS char* smem = (AIM_LDS char*)smem_raw;
    const int tiles_k = (Kw + 127) / 128;
    const int tn = blockIdx.x / tiles_k, tk = blockIdx.x - tn * tiles_k;
    const int n0 = tn * 128, k0 = tk * 128;
    const int mbeg = blockIdx.y * chunk;
    const int mend = min(M, mbeg + chunk);
    if (mbeg >= mend) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;

    const bf16_t* Gb = G + (long long)mbeg * ldg + n0;
    const bf16_t* Ab = A + (long long)mbeg * lda + k0;
    const int rows = mend - mbeg;
    __amdgpu_buffer_rsrc_t rG = make_rsrc(Gb, ((long long)(rows - 1) * ldg + min(128, Nw - n0)) * 2);
    __amdgpu_buffer_rsrc_t rA = make_rsrc(Ab, ((long long)(rows - 1) * lda + min(128, Kw - k0)) * 2);

    // staging: per operand 16 pieces (2 column halves x 8 row groups of 8); wave takes 4 of each
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    auto stage = [&](int buf, int ms) {
        AIM_LDS char* dG = smem + buf * (2 * OPER_BYTES);
        AIM_LDS char* dA = dG + OPER_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = wave * 4 + j;           // 0..15
            const int half = piece >> 3, rg = piece & 7;
            const int r = ms * MSTEP + rg * 8 + srow;  // row inside the chunk
            const int col = half * 64 + schunk * 8;
            const unsigned vg = (r < rows && n0 + col < Nw) ? (unsigned)((r * ldg + col) * 2) : AIM_OOB;
            const unsigned va = (r < rows && k0 + col < Kw) ? (unsigned)((r * lda + col) * 2) : AIM_OOB;
            stage_piece(rG, dG + half * HALF_BYTES + rg * 1024, vg);
            stage_piece(rA, dA + half * HALF_BYTES + rg * 1024, va);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposing fragment read: 16 columns c0..c0+15 (inside one 64-col image), 8 reduction rows
    // {mb + 4*fq + 0..3} and {mb + 16 + 4*fq + 0..3}; lane i = lane&15 supplies row (i>>2), cols 4*(i&3)..
    auto frag = [&](const AIM_LDS char* img, int c0, int mb) -> bf16x8 {
        const int ch = (c0 >> 3) + ((frow & 3) >> 1), half8 = (frow & 1) * 8;
        const int r0 = mb + fq * 4 + (frow >> 2);
        const bf16x4 a = lds_read_tr4(img + swz_off(r0, ch) + half8);
        const bf16x4 b = lds_read_tr4(img + swz_off(r0 + 16, ch) + half8);
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[e] = a[e];
            f[4 + e] = b[e];
        }
        return f;
    };

    const int nsteps = (rows + MSTEP - 1) / MSTEP;
    stage(0, 0);
    for (int ms = 0; ms < nsteps; ++ms) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ms + 1 < nsteps) stage((ms + 1) & 1, ms + 1);
        const AIM_LDS char* sG = smem + (ms & 1) * (2 * OPER_BYTES) + wn * HALF_BYTES;  // wave's 64 n columns
        const AIM_LDS char* sA = smem + (ms & 1) * (2 * OPER_BYTES) + OPER_BYTES + wk * HALF_BYTES;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            bf16x8 gf[4], af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) gf[i] = frag(sG, i * 16, sub * 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = frag(sA, j * 16, sub * 32);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[i], af[j], acc[i][j], 0, 0, 0);
        }
    }
    // D[i = n][j = k]: lane holds k = .. + (lane&15), n = .. + 4*(lane>>4) + e
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + wk * 64 + j * 16 + frow;
            if (k >= Kw) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + wn * 64 + i * 16 + fq * 4 + e;
                if (n < Nw) atomicAdd(dW + (long long)n * lddw + k, acc[i][j][e]);
            }
        }
}

}  // namespace

extern "C" int aim_wgrad_bf16(const aim_bf16* G, int ldg, const aim_bf16* A, int lda, float* dW, int lddw, float* db,
                              int M, int Nw, int Kw, void* stream) {
    AIM_CHECK_ARG(M > 0 && Nw > 0 && Kw > 0 && (Nw % 8) == 0 && (Kw % 8) == 0, "wgrad: Nw/Kw must be positive multiples of 8 (Nw=%d Kw=%d)", Nw, Kw);
    AIM_CHECK_ARG((ldg % 8) == 0 && (lda % 8) == 0, "wgrad: ldg/lda must be multiples of 8");
    AIM_CHECK_ARG(G && A && dW, "wgrad: null pointer");
    const int tiles = ((Nw + 127) / 128) * ((Kw + 127) / 128);
    int nchunks = (1024 + tiles - 1) / tiles;
    const int maxchunks = (M + MSTEP - 1) / MSTEP;
    if (nchunks > maxchunks) nchunks = maxchunks;
    int chunk = (M + nchunks - 1) / nchunks;
    chunk = ((chunk + MSTEP - 1) / MSTEP) * MSTEP;
    nchunks = (M + chunk - 1) / chunk;
    AIM_CHECK_ARG((long long)chunk * (ldg > lda ? ldg : lda) * 2 < 0x7fffffffLL, "wgrad: chunk too large");
    hipLaunchKernelGGL(wgrad_kernel, dim3(tiles, nchunks), dim3(256), 4 * OPER_BYTES, (hipStream_t)stream,
                       (const bf16_t*)G, ldg, (const bf16_t*)A, lda, dW, lddw, M, Nw, Kw, chunk);
    AIM_CHECK_LAUNCH("aim_wgrad_bf16");
    if (db) return aim_colsum_bf16(G, ldg, nullptr, nullptr, 0, db, M, Nw, nullptr, 0, stream);
    return 0;
}
