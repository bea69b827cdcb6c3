// Weight-gradient GEMM for the trainable adapters:  dW[n][k] += sum_m G[m][n] * A[m][k].  gfx950 only.
//
// Autograd counterpart of Adapter.D_fc1 / D_fc2 (reference vit_clip.py:57-58, 62-64); the reference
// gets it from torch autograd (addmm backward).  Both operands are row-major with the REDUCTION index
// m as the slow dimension, so neither can be read as a k-contiguous MFMA fragment.  The tiles are
// staged untransposed ([m][n] and [m][k], coalesced) into swizzled LDS images and the fragments are
// fetched with ds_read_b64_tr_b16, the hardware transposing read: no transposed copies in HBM.
//
// Block tile 256(n) x 256(k), 8 waves as 2x4, wave tile 128x64 = 8x4 MFMA 16x16x32 tiles, 4-stage LDS ring.
// The M dimension is split over blockIdx.y in chunks; partial tiles go to a caller-provided scratch slab and
// are summed by a second kernel (no atomics), or are combined with fp32 atomics when no scratch is given.
#include "aim_common.h"
#include <stdlib.h>
#include "aim_kernels_internal.h"

namespace {

constexpr int MSTEP = 32;                 // reduction rows per stage
constexpr int NST = 4;                    // LDS ring depth: three stages in flight hide the HBM latency
constexpr int IMG = MSTEP * 128;          // one [32 m][64 col] image = 4 KiB
constexpr int OPER_BYTES = 4 * IMG;       // [32 m][256 col] = four images = 16 KiB
constexpr int STAGE_BYTES = 2 * OPER_BYTES;
constexpr int TILE = 256;                 // output tile 256 (n) x 256 (k)

// One workgroup = 8 waves as 2 (n) x 4 (k); a wave owns 128 x 64 outputs = 8 x 4 MFMA 16x16x32 tiles.
// The kernel is bound by the bytes a CU can pull per cycle (every output tile re-reads its M-chunk of G and
// A), so the tile is as large as the accumulators allow: [768 x 192] needs 3 tiles, 271 MB of operand reads
// for M = 100 864 instead of 620 MB with 128 x 128 tiles.
//
// Bias gradient in the same pass (db != nullptr): db[n] += sum_m rs(m) G[m][n], rs(m) = at[m % ntok] (1 without `at`).  The
// workgroups of the first k-tile column (tk == 0) already hold every G row of their chunk in LDS: thread t sums column
// t & 255 over rows 16 (t >> 8) .. +15 of each stage beside the MFMAs (the kernel is bound by the CU's load path, not by LDS
// or VALU issue); the per-chunk column sums go to their own slab and are added up, in chunk order, by the finish kernel.
// This replaces a separate column-sum launch that re-read all of G from HBM (155 MB for the MLP_Adapter's D_fc2 bias).
__global__ __launch_bounds__(512) void wgrad_kernel(const bf16_t* __restrict__ G, int ldg, const bf16_t* __restrict__ A,
                                                    int lda, float* __restrict__ dW, int lddw, float* __restrict__ partial,
                                                    int M, int Nw, int Kw, int chunk, float* __restrict__ db,
                                                    float* __restrict__ bias_partial, const float* __restrict__ at, int ntok) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    AIM_LDS char* smem = (AIM_LDS char*)smem_raw;
    AIM_LDS float* sAt = (AIM_LDS float*)(smem + NST * STAGE_BYTES);     // [ntok] row factors (only with `at`)
    const int tiles_k = (Kw + TILE - 1) / TILE;
    const int tiles = ((Nw + TILE - 1) / TILE) * tiles_k;
    const int cidx = blockIdx.x / tiles, tix = blockIdx.x - cidx * tiles;
    const int tn = tix / tiles_k, tk = tix - tn * tiles_k;
    const int n0 = tn * TILE, k0 = tk * TILE;
    const int mbeg = cidx * chunk;
    const int mend = min(M, mbeg + chunk);
    if (mbeg >= mend) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wk = wave & 3;
    const int frow = lane & 15, fq = lane >> 4;

    const bf16_t* Gb = G + (long long)mbeg * ldg + n0;
    const bf16_t* Ab = A + (long long)mbeg * lda + k0;
    const int rows = mend - mbeg;
    aim_rsrc_words rG = make_rsrc_words(Gb, ((long long)(rows - 1) * ldg + min(TILE, Nw - n0)) * 2);
    aim_rsrc_words rA = make_rsrc_words(Ab, ((long long)(rows - 1) * lda + min(TILE, Kw - k0)) * 2);

    // staging: per operand 16 pieces (4 column images x 4 row groups of 8); a wave takes 2 of each.  A piece's offset is its
    // stage-0 offset plus ms * (32 rows): one v_add per piece and step.  Rows past the chunk need no test -- the resources end
    // with the chunk's last row, so their offsets are out of range by themselves (zero fill); columns past Nw / Kw carry
    // AIM_OOB from the start (AIM_OOB + ms * step stays out of range).
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    unsigned vg0[2], va0[2];
    AIM_LDS char* pdst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = wave * 2 + j;           // 0..15
        const int img = piece >> 2, rg = piece & 3;
        const int r = rg * 8 + srow;              // row inside stage 0
        const int col = img * 64 + schunk * 8;
        vg0[j] = n0 + col < Nw ? (unsigned)((r * ldg + col) * 2) : AIM_OOB;
        va0[j] = k0 + col < Kw ? (unsigned)((r * lda + col) * 2) : AIM_OOB;
        pdst[j] = smem + img * IMG + rg * 1024;
    }
    const unsigned gstep = (unsigned)(MSTEP * ldg * 2), astep = (unsigned)(MSTEP * lda * 2);
    auto stage = [&](int buf, int ms) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            stage_piece_asm(rG, pdst[j] + buf * STAGE_BYTES, vg0[j] + (unsigned)ms * gstep);
            stage_piece_asm(rA, pdst[j] + buf * STAGE_BYTES + OPER_BYTES, va0[j] + (unsigned)ms * astep);
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposing fragment read: 16 columns c0..c0+15 of the operand's [32 m][256] stage, reduction rows
    // {4*fq + 0..3} and {16 + 4*fq + 0..3}; lane i = lane&15 supplies row (i>>2), cols 4*(i&3)..
    auto frag = [&](const AIM_LDS char* oper, int c0) -> bf16x8 {
        const AIM_LDS char* img = oper + (c0 >> 6) * IMG;
        const int ch = ((c0 & 63) >> 3) + ((frow & 3) >> 1), half8 = (frow & 1) * 8;
        const int r0 = fq * 4 + (frow >> 2);
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

    // ring of NST stages; each wave issues 4 LDS-DMA loads per stage.  Iteration ms: vmcnt(8) retires
    // stage ms (stages ms+1, ms+2 stay in flight), one barrier, re-stage buffer (ms+3)%4 -- read in
    // iteration ms-1, which every wave finished before this barrier -- then 24 tr-reads + 32 MFMA.
    const int nsteps = (rows + MSTEP - 1) / MSTEP;
    const bool do_bias = db != nullptr && tk == 0;      // workgroup-uniform
    // bias column sums: thread t owns the 8 columns of 16-byte chunk (t & 31) in rows (t >> 5) and (t >> 5) + 16 of every stage
    // (two ds_read_b128 per step; the first version read sixteen 2-byte values per thread and step: +30 us per launch)
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int bcg = tid & 31, brow = tid >> 5;
    int btok0 = 0, btok1 = 0;
    if (do_bias && at) {
        for (int i = tid; i < ntok; i += 512) sAt[i] = at[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // written before the loop's first barrier publishes it
        btok0 = (mbeg + brow) % ntok;
        btok1 = (mbeg + brow + 16) % ntok;
    }
    const int bstep = MSTEP % (ntok > 0 ? ntok : 1);
    stage(0, 0); stage(1, 1); stage(2, 2);            // stages past the chunk are zero-fill, still counted
    for (int ms = 0; ms < nsteps; ++ms) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage((ms + 3) % NST, ms + 3);
        const AIM_LDS char* sG = smem + (ms % NST) * STAGE_BYTES;
        const AIM_LDS char* sA = sG + OPER_BYTES;
        if (do_bias) {
            // chunk bcg of the [32 m][256 n] stage: image bcg >> 3, 16-byte chunk bcg & 7 of rows brow and brow + 16
            const AIM_LDS char* img = sG + (bcg >> 3) * IMG;
            const bf16x8 v0 = lds_read8(img + swz_off(brow, bcg & 7));
            const bf16x8 v1 = lds_read8(img + swz_off(brow + 16, bcg & 7));
            float f0 = 1.0f, f1 = 1.0f;
            if (at) {
                f0 = sAt[btok0];
                f1 = sAt[btok1];
                btok0 += bstep; if (btok0 >= ntok) btok0 -= ntok;
                btok1 += bstep; if (btok1 >= ntok) btok1 -= ntok;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) bs[e] += f0 * (float)v0[e] + f1 * (float)v1[e];      // rows past the chunk are zero-filled
        }
        bf16x8 gf[8], af[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = frag(sA, wk * 64 + j * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) gf[i] = frag(sG, wn * 128 + i * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], gf[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (do_bias) {          // the sixteen row groups of a column meet in LDS (the stage images are dead now), summed in order
        __syncthreads();
        AIM_LDS float* red = (AIM_LDS float*)smem;
#pragma unroll
        for (int e = 0; e < 8; ++e) red[brow * 256 + bcg * 8 + e] = bs[e];
        __syncthreads();
        if (tid < 256 && n0 + tid < Nw) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v += red[r * 256 + tid];
            if (bias_partial) bias_partial[(long long)cidx * Nw + n0 + tid] = v;
            else db[n0 + tid] += v;                    // one chunk: this workgroup is the column's only writer
        }
    }
    // The MFMA is issued with the A fragment FIRST (D[k][n]): the lane holds n = .. + (lane & 15) and the four consecutive
    // k = .. + 4 * (lane >> 4) + e, so a partial tile leaves as 16-byte stores (32 per wave; the 4-byte form, 128 per wave,
    // took 7.6 us of a 46 us workgroup).
    // With a scratch slab the chunk's partial tile is stored plainly (slab [chunk][Nw][Kw], summed by
    // wgrad_finish_kernel: no atomics, bitwise reproducible); otherwise fp32 atomics into dW.
    float* slab = partial ? partial + (long long)cidx * Nw * Kw : nullptr;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 128 + i * 16 + frow;
            const int k = k0 + wk * 64 + j * 16 + fq * 4;
            if (n >= Nw || k >= Kw) continue;          // Kw is a multiple of 8: a lane's four k are in range together
            if (slab) *(f32x4*)(slab + (long long)n * Kw + k) = acc[i][j];
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(dW + (long long)n * lddw + k + e, acc[i][j][e]);
        }
}

// dW[n][k] += sum_c slab[c][n][k]  and  db[n] += sum_c bias_slab[c][n], both in chunk order (bitwise reproducible).
// A thread owns four consecutive outputs and walks the chunks with eight independent 16-byte loads in flight (the first
// version issued one dependent 4-byte load per chunk: latency-bound at 80 us for 50 MB).
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ partial, float* __restrict__ dW,
                                                           int lddw, int nchunks, int Nw, int Kw,
                                                           const float* __restrict__ bias_partial, float* __restrict__ db) {
    const long long total = (long long)Nw * Kw;
    const long long idx = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx < total) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        int c = 0;
        for (; c + 8 <= nchunks; c += 8) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const f32x4*)(partial + (long long)(c + j) * total + idx);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
        for (; c < nchunks; ++c) acc += *(const f32x4*)(partial + (long long)c * total + idx);
        const int n = (int)(idx / Kw), k = (int)(idx - (long long)n * Kw);      // Kw % 8 == 0: the four outputs share a row
        f32x4* o = (f32x4*)(dW + (long long)n * lddw + k);
        *o = *o + acc;
    }
    if (bias_partial) {          // a wave per bias column, four columns per block: chunk-strided loads, fixed-order wave sum
        const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (n < Nw) {
            float a = 0.f;
            for (int c = lane; c < nchunks; c += 64) a += bias_partial[(long long)c * Nw + n];
            a = wave_sum(a);
            if (lane == 0) db[n] += a;
        }
    }
}

// (A/B reference, AIM_WGRAD_FINISH_V=0) one output per thread, one dependent load per chunk
__global__ __launch_bounds__(256) void wgrad_finish1_kernel(const float* __restrict__ partial, float* __restrict__ dW,
                                                            int lddw, int nchunks, int Nw, int Kw) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)Nw * Kw;
    if (idx >= total) return;
    float acc = 0.f;
    for (int c = 0; c < nchunks; ++c) acc += partial[(long long)c * total + idx];
    const int n = (int)(idx / Kw), k = (int)(idx - (long long)n * Kw);
    dW[(long long)n * lddw + k] += acc;
}

}  // namespace

static int wgrad_chunks(int M, int tiles, int* chunk_out) {
    // Target workgroup count.  256 = one per CU is fastest stand-alone; the framework runs these launches on a stream
    // of their own beside MFMA-bound GEMMs, where fewer, longer workgroups mean fewer partial slabs to write and sum.
    static const int target = [] { const char* e = getenv("AIM_WGRAD_WGS"); const int v = e ? atoi(e) : 256; return v > 0 ? v : 256; }();
    int nchunks = (target + tiles - 1) / tiles;
    const int maxchunks = (M + MSTEP - 1) / MSTEP;
    if (nchunks > maxchunks) nchunks = maxchunks;
    int chunk = (M + nchunks - 1) / nchunks;
    chunk = ((chunk + MSTEP - 1) / MSTEP) * MSTEP;
    if (chunk_out) *chunk_out = chunk;
    return (M + chunk - 1) / chunk;
}

extern "C" int64_t aim_wgrad_workspace_bytes(int M, int Nw, int Kw) {
    const int tiles = ((Nw + TILE - 1) / TILE) * ((Kw + TILE - 1) / TILE);
    const int nchunks = wgrad_chunks(M, tiles, nullptr);
    return nchunks > 1 ? (int64_t)nchunks * Nw * (Kw + 1) * 4 : 0;      // weight slabs + one bias row per chunk
}

extern "C" int aim_wgrad_bias_bf16(const aim_bf16* G, int ldg, const aim_bf16* A, int lda, float* dW, int lddw, float* db,
                                   const float* at, int ntok, int M, int Nw, int Kw, float* workspace, int64_t workspace_bytes,
                                   void* stream) {
    AIM_CHECK_ARG(M > 0 && Nw > 0 && Kw > 0 && (Nw % 8) == 0 && (Kw % 8) == 0, "wgrad: Nw/Kw must be positive multiples of 8 (Nw=%d Kw=%d)", Nw, Kw);
    AIM_CHECK_ARG((ldg % 8) == 0 && (lda % 8) == 0 && (lddw % 4) == 0, "wgrad: ldg/lda must be multiples of 8, lddw of 4");
    AIM_CHECK_ARG(G && A && dW, "wgrad: null pointer");
    AIM_CHECK_ARG(!at || (db && ntok > 0 && ntok <= 4096), "wgrad: row factors need db and 0 < ntok <= 4096");
    const int tiles = ((Nw + TILE - 1) / TILE) * ((Kw + TILE - 1) / TILE);
    int chunk = 0;
    const int nchunks = wgrad_chunks(M, tiles, &chunk);
    const int lds_bytes = NST * STAGE_BYTES + (at ? ((ntok * 4 + 15) & ~15) : 0);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    AIM_CHECK_ARG((long long)chunk * (ldg > lda ? ldg : lda) * 2 < 0x7fffffffLL, "wgrad: chunk too large");
    float* slab = (workspace && nchunks > 1 && workspace_bytes >= (int64_t)nchunks * Nw * (Kw + 1) * 4) ? workspace : nullptr;
    static const bool fuse_bias = [] { const char* e = getenv("AIM_WGRAD_FUSE_BIAS"); return !e || atoi(e) != 0; }();
    static const bool finish_v = [] { const char* e = getenv("AIM_WGRAD_FINISH_V"); return !e || atoi(e) != 0; }();
    float* bias_slab = (slab && db && fuse_bias) ? slab + (long long)nchunks * Nw * Kw : nullptr;
    if (db && nchunks > 1 && !bias_slab) {       // no scratch: the bias through the stand-alone column sum (atomics inside)
        const int rc = aim_colsum_bf16(G, ldg, nullptr, at, ntok, db, M, Nw, nullptr, 0, stream);
        if (rc) return rc;
        db = nullptr;
    }
    hipLaunchKernelGGL(wgrad_kernel, dim3(tiles * nchunks), dim3(512), lds_bytes, (hipStream_t)stream,
                       (const bf16_t*)G, ldg, (const bf16_t*)A, lda, dW, lddw, slab, M, Nw, Kw, chunk, db, bias_slab, at, ntok);
    AIM_CHECK_LAUNCH("aim_wgrad_bf16");
    if (slab) {
        const long long total = (long long)Nw * Kw;
        unsigned fgrid = (unsigned)((total / 4 + 255) / 256);
        if (bias_slab && (unsigned)((Nw + 3) / 4) > fgrid) fgrid = (unsigned)((Nw + 3) / 4);
        if (finish_v || bias_slab)
            hipLaunchKernelGGL(wgrad_finish_kernel, dim3(fgrid), dim3(256), 0, (hipStream_t)stream,
                               slab, dW, lddw, nchunks, Nw, Kw, bias_slab, db);
        else
            hipLaunchKernelGGL(wgrad_finish1_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                               slab, dW, lddw, nchunks, Nw, Kw);
        AIM_CHECK_LAUNCH("aim_wgrad_bf16(finish)");
    }
    return 0;
}

extern "C" int aim_wgrad_bf16(const aim_bf16* G, int ldg, const aim_bf16* A, int lda, float* dW, int lddw, float* db,
                              int M, int Nw, int Kw, float* workspace, int64_t workspace_bytes, void* stream) {
    return aim_wgrad_bias_bf16(G, ldg, A, lda, dW, lddw, db, nullptr, 0, M, Nw, Kw, workspace, workspace_bytes, stream);
}
