// Spatial multi-head self-attention backward (dQ, dK, dV), probabilities recomputed from the
// forward's log-sum-exp.  gfx950 only.
//
// Autograd counterpart of reference vit_clip.py:139-156 (the reference relies on torch autograd
// through bmm/softmax/bmm and keeps the [BT,H,N,N] probabilities alive for it).
//
// Two kernels per call, one workgroup per (frame, head) each (dq: 8 waves; dkv: 4 waves, 120 VGPRs):
//   dq : query on the MFMA lane (same orientation as the forward).  Per 16-query tile and per pair
//        of 16-key tiles:  S^T = K Q^T,  dP^T = V dO^T,  dS^T = P^T o (dP^T - delta) / 8 and
//        dQ^T += K^T dS^T with the dS^T accumulators used directly as the MFMA's second operand
//        (K^T fragments by ds_read_b64_tr_b16).  Also writes delta = rowsum(dO o O).
//        (Prefetching the K / V row fragments one key pair ahead was measured: 128 VGPRs + spill, 6 % slower.)
//   dkv: key on the MFMA lane.  A wave owns 32 keys (K/V fragments in registers) and sweeps the
//        queries 32 at a time:  S = Q K^T,  dP = dO V^T,  then  dV^T += dO^T P  and  dK^T += Q^T dS
//        with P / dS accumulators as the second operand and Q^T / dO^T fragments by transposing
//        reads of the row-major Q / dO images.  dK, dV need no cross-workgroup reduction.
// Scores are recomputed twice (7 MFMA products instead of 5) in exchange for no dS exchange through
// LDS and no atomics; attention is ~4 % of the block's FLOPs.
#include "aim_common.h"
#include "aim_kernels_internal.h"
#include <stdlib.h>

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float C2 = 0.125f * LOG2E;      // 1/sqrt(dh) * log2(e): probabilities are recomputed in base 2

__global__ __launch_bounds__(512, 4) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                          const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ delta, bf16_t* __restrict__ dqkv, int N,
                                                          int H, int nkt) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    AIM_LDS char* sK = (AIM_LDS char*)smem_raw;
    AIM_LDS char* sV = sK + nkt * 16 * 128;

    const int bt = blockIdx.x / H, h = blockIdx.x - bt * H;
    const int D = H * 64, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const bf16_t* base = qkv + (long long)bt * N * ld + h * 64;
    {
        __amdgpu_buffer_rsrc_t rK = make_rsrc(base + D, ((long long)(N - 1) * ld + 64) * 2);
        __amdgpu_buffer_rsrc_t rV = make_rsrc(base + 2 * D, ((long long)(N - 1) * ld + 64) * 2);
        const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
        for (int p = wave; p < nkt * 2; p += 8) {
            const int key = p * 8 + srow;
            const unsigned voff = key < N ? (unsigned)((key * ld + schunk * 8) * 2) : AIM_OOB;
            stage_piece(rK, sK + p * 1024, voff);
            stage_piece(rV, sV + p * 1024, voff);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int nqt = (N + 15) >> 4;
    for (int qt = wave; qt < nqt; qt += 8) {
        asm volatile("" ::: "memory");          // keep the (qt-invariant) K/V fragment reads inside the loop
        const int q = qt * 16 + frow;
        const int qc = q < N ? q : N - 1;
        bf16x8 qf[2], dof[2];
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = (ks * 4 + fq) * 8;
            qf[ks] = *(const bf16x8*)(base + (long long)qc * ld + c);
            const long long orow = ((long long)bt * N + qc) * D + h * 64 + c;
            dof[ks] = *(const bf16x8*)(dout + orow);
            const bf16x8 of = *(const bf16x8*)(out + orow);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += (float)dof[ks][e] * (float)of[e];
        }
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        // exp(s/8 - L) = exp2(s * C2 - L2): one fma + one v_exp per score
        const float L2 = lse[((long long)bt * H + h) * N + qc] * LOG2E;
        if (fq == 0 && q < N) delta[((long long)bt * H + h) * N + q] = dl;

        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < nkt / 2; ++kk) {
            bf16x8 dsf;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * kk + u;
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 kf = lds_read8(sK + swz_off(t * 16 + frow, ks * 4 + fq));
                    const bf16x8 vf = lds_read8(sV + swz_off(t * 16 + frow, ks * 4 + fq));
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[ks], dp, 0, 0, 0);
                }
                // No key mask: rows of K and V past N are zero-filled in LDS, so such a key has a finite p and its dS
                // meets a zero K^T row in the dQ product.  The 1/sqrt(dh) factor of dS is applied once to dQ.
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __builtin_amdgcn_exp2f(s[e] * C2 - L2);
                    dsf[u * 4 + e] = (bf16_t)(p * (dp[e] - dl));
                }
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int r0 = (2 * kk) * 16 + fq * 4 + (frow >> 2);
                const int ch = dt * 2 + ((frow & 3) >> 1), half = (frow & 1) * 8;
                const bf16x4 a = lds_read_tr4(sK + swz_off(r0, ch) + half);
                const bf16x4 b = lds_read_tr4(sK + swz_off(r0 + 16, ch) + half);
                bf16x8 ktf;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ktf[e] = a[e];
                    ktf[4 + e] = b[e];
                }
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf, dq[dt], 0, 0, 0);
            }
        }
        {   // 16-byte stores: tiles (dt, dt+1) paired across even / odd 16-lane rows (aim_common.h pair_rows16)
            bf16_t* op = dqkv + ((long long)bt * N + (q < N ? q : 0)) * ld + h * 64 + ((fq & 1) ? 16 + (fq - 1) * 4 : fq * 4);
#pragma unroll
            for (int dt = 0; dt < 4; dt += 2) {
                const bf16x8 v = pair_rows16(
                    pack4(dq[dt][0] * 0.125f, dq[dt][1] * 0.125f, dq[dt][2] * 0.125f, dq[dt][3] * 0.125f),
                    pack4(dq[dt + 1][0] * 0.125f, dq[dt + 1][1] * 0.125f, dq[dt + 1][2] * 0.125f, dq[dt + 1][3] * 0.125f));
                if (q < N) *(bf16x8*)(op + dt * 16) = v;
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           bf16_t* __restrict__ dqkv, int N, int H, int nq32) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    AIM_LDS char* sQ = (AIM_LDS char*)smem_raw;
    AIM_LDS char* sO = sQ + nq32 * 128;
    AIM_LDS float* sL = (AIM_LDS float*)(sO + nq32 * 128);
    AIM_LDS float* sD = sL + nq32;

    const int bt = blockIdx.x / H, h = blockIdx.x - bt * H;
    const int D = H * 64, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const bf16_t* base = qkv + (long long)bt * N * ld + h * 64;
    const bf16_t* dob = dout + (long long)bt * N * D + h * 64;
    {
        __amdgpu_buffer_rsrc_t rQ = make_rsrc(base, ((long long)(N - 1) * ld + 64) * 2);
        __amdgpu_buffer_rsrc_t rO = make_rsrc(dob, ((long long)(N - 1) * D + 64) * 2);
        const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
        for (int p = wave; p < nq32 / 8; p += 4) {
            const int qr = p * 8 + srow;
            stage_piece(rQ, sQ + p * 1024, qr < N ? (unsigned)((qr * ld + schunk * 8) * 2) : AIM_OOB);
            stage_piece(rO, sO + p * 1024, qr < N ? (unsigned)((qr * D + schunk * 8) * 2) : AIM_OOB);
        }
        for (int i = tid; i < nq32; i += 256) {
            sL[i] = i < N ? lse[((long long)bt * H + h) * N + i] * LOG2E : 0.f;
            sD[i] = i < N ? delta[((long long)bt * H + h) * N + i] : 0.f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int nkp = (N + 31) >> 5;
    for (int kp = wave; kp < nkp; kp += 4) {
        bf16x8 kf[2][2], vf[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int key = kp * 32 + u * 16 + frow;
            const int kc = key < N ? key : N - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[u][ks] = *(const bf16x8*)(base + (long long)kc * ld + D + (ks * 4 + fq) * 8);
                vf[u][ks] = *(const bf16x8*)(base + (long long)kc * ld + 2 * D + (ks * 4 + fq) * 8);
            }
        }
        f32x4 dk[4][2], dv[4][2];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                dk[dt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
                dv[dt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        // LDS operands of a 32-query step: row fragments + (L, delta) for S / dP, transposed fragments for dV / dK.
        // The transposed fragments of step qs are requested at the TOP of the step (they are used in its second half) and
        // the row fragments of step qs+1 in its MIDDLE, into the registers the S / dP MFMAs have just released: with
        // 2 waves per SIMD nobody else hides an LDS round trip per MFMA group.
        struct RowSet {
            bf16x8 qa[2][2], oa[2][2];
            float Lr[2][4], Dr[2][4];
        };
        auto load_rows = [&](int qs, RowSet& rs) {
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int qrow = (2 * qs + w) * 16;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    rs.qa[w][ks] = lds_read8(sQ + swz_off(qrow + frow, ks * 4 + fq));
                    rs.oa[w][ks] = lds_read8(sO + swz_off(qrow + frow, ks * 4 + fq));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rs.Lr[w][e] = sL[qrow + fq * 4 + e];
                    rs.Dr[w][e] = sD[qrow + fq * 4 + e];
                }
            }
        };
        auto step = [&](int qs, RowSet& cur, bool more) {
            bf16x4 ta[4], tb[4], tc[4], td[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int r0 = qs * 32 + fq * 4 + (frow >> 2);
                const int ch = dt * 2 + ((frow & 3) >> 1), half = (frow & 1) * 8;
                ta[dt] = lds_read_tr4(sQ + swz_off(r0, ch) + half);
                tb[dt] = lds_read_tr4(sQ + swz_off(r0 + 16, ch) + half);
                tc[dt] = lds_read_tr4(sO + swz_off(r0, ch) + half);
                td[dt] = lds_read_tr4(sO + swz_off(r0 + 16, ch) + half);
            }
            __builtin_amdgcn_sched_barrier(0);          // keep the requests up here: the scheduler would sink them to their uses
            bf16x8 pf[2], dsf[2];
#pragma unroll
            for (int w = 0; w < 2; ++w) {  // the two 16-query tiles of this step
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.qa[w][ks], kf[u][ks], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.oa[w][ks], vf[u][ks], dp, 0, 0, 0);
                    }
                    // No masks: a key past N is a clamped duplicate whose dK / dV rows are never stored; a query past N
                    // has zero-filled Q and dO rows (and L = delta = 0), so it adds nothing to dK or dV.
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float p = __builtin_amdgcn_exp2f(s[e] * C2 - cur.Lr[w][e]);
                        pf[u][w * 4 + e] = (bf16_t)p;
                        dsf[u][w * 4 + e] = (bf16_t)(p * (dp[e] - cur.Dr[w][e]));
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) load_rows(qs + 1, cur);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x8 qt8, ot8;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    qt8[e] = ta[dt][e];
                    qt8[4 + e] = tb[dt][e];
                    ot8[e] = tc[dt][e];
                    ot8[4 + e] = td[dt][e];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    dv[dt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ot8, pf[u], dv[dt][u], 0, 0, 0);
                    dk[dt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt8, dsf[u], dk[dt][u], 0, 0, 0);
                }
            }
        };
        {
            const int nsteps = nq32 / 32;
            RowSet rows;
            load_rows(0, rows);
            for (int qs = 0; qs < nsteps; ++qs) step(qs, rows, qs + 1 < nsteps);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int key = kp * 32 + u * 16 + frow;
            // 16-byte stores: tiles (dt, dt+1) paired across even / odd 16-lane rows (aim_common.h pair_rows16)
            bf16_t* op = dqkv + ((long long)bt * N + (key < N ? key : 0)) * ld + h * 64 + ((fq & 1) ? 16 + (fq - 1) * 4 : fq * 4);
#pragma unroll
            for (int dt = 0; dt < 4; dt += 2) {
                const bf16x8 vk = pair_rows16(
                    pack4(dk[dt][u][0] * 0.125f, dk[dt][u][1] * 0.125f, dk[dt][u][2] * 0.125f, dk[dt][u][3] * 0.125f),
                    pack4(dk[dt + 1][u][0] * 0.125f, dk[dt + 1][u][1] * 0.125f, dk[dt + 1][u][2] * 0.125f, dk[dt + 1][u][3] * 0.125f));
                const bf16x8 vv = pair_rows16(pack4(dv[dt][u][0], dv[dt][u][1], dv[dt][u][2], dv[dt][u][3]),
                                              pack4(dv[dt + 1][u][0], dv[dt + 1][u][1], dv[dt + 1][u][2], dv[dt + 1][u][3]));
                if (key < N) {
                    *(bf16x8*)(op + D + dt * 16) = vk;
                    *(bf16x8*)(op + 2 * D + dt * 16) = vv;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Fused backward (N <= 224: ViT-B/16's 197 tokens): ONE pass over Q, K, V, dO per (frame, head) and the five products
// of flash attention's backward instead of seven.  One workgroup (8 waves, one per CU) per (frame, head), waves in two
// ROLES so that nothing waits for the dS exchange:
//   prologue   Q, dO, K -> swizzled LDS images (LDS-DMA, zero-filled past N); delta = rowsum(dO o O) and L -> LDS.
//   producers  (waves 0 .. nkb-1; key on the MFMA lane, as attn_bwd_dkv_kernel) wave w owns keys 32w .. 32w+31 with its K / V
//              fragments in registers and sweeps the queries in blocks of 64:  S = Q K^T, dP = dO V^T, P = exp2(S c - L),
//              dS = P o (dP - delta);  dV^T += dO^T P and dK^T += Q^T dS with the P / dS accumulators as second operand;
//              dS is ALSO written, as bf16, to a [key][64 query] LDS image (8-byte writes: a lane holds 4 consecutive
//              queries of its key), double-buffered over the query blocks;
//   consumer   (wave 7: N <= 224 leaves it no key block) dQ^T += K^T dS^T for the PREVIOUS block's 64 queries while the
//              producers work on the next one: dS^T fragments out of the [key][query] image by ds_read_b64_tr_b16 (query
//              on the lane), K^T fragments by transposing reads of the K image (as attn_bwd_dq_kernel);
//   one workgroup barrier per query block; epilogue: dK, dV of each producer's 32 keys.
// dS crosses LDS once; nothing is reduced across waves; no atomics; HBM traffic = read qkv + out + dout, write dqkv.
__global__ __launch_bounds__(512) void attn_bwd_fused_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                             const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                             bf16_t* __restrict__ dqkv, int N, int H, int nkb, int stagger
#ifdef AIM_X_STAMPS
                                                             , unsigned long long* stamps
#endif
) {
#ifdef AIM_X_STAMPS
    unsigned long long tstamp[6];
    tstamp[0] = __builtin_amdgcn_s_memrealtime();
#define FST(i) tstamp[i] = __builtin_amdgcn_s_memrealtime()
#else
#define FST(i)
#endif
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // The first workgroup of every CU starts staggered (0 .. 7 x ~2 us by blockIdx % 8): all workgroups do equal work, so
    // without it every CU of the chip sits in its load-heavy prologue at the same time (HBM-bound, ~5.6 us) and in its
    // load-free main loop at the same time (HBM idle); the offsets persist through the CU's chain of workgroups.
    if (stagger > 0 && blockIdx.x < 256) {
        const int k = (int)(blockIdx.x & 7) * stagger;
        for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(16);      // 16 x 64 clocks ~ 0.5 us
    }
    const int nrow = nkb * 32;                                  // rows of every image (queries and keys padded alike)
    AIM_LDS char* sQ = (AIM_LDS char*)smem_raw;
    AIM_LDS char* sO = sQ + nrow * 128;
    AIM_LDS char* sK = sO + nrow * 128;
    AIM_LDS char* sDS = sK + nrow * 128;                        // 2 x [key][64 queries] bf16
    AIM_LDS float* sL = (AIM_LDS float*)(sDS + 2 * nrow * 128);
    AIM_LDS float* sD = sL + nrow;

    const int bt = blockIdx.x / H, h = blockIdx.x - bt * H;
    const int D = H * 64, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const bf16_t* base = qkv + (long long)bt * N * ld + h * 64;
    const bf16_t* dob = dout + (long long)bt * N * D + h * 64;
    const bf16_t* ob = out + (long long)bt * N * D + h * 64;
    const bool producer = wave < nkb;
    const bool consumer = wave == 7;                             // (nkb <= 7)
    bf16x8 kf[2][2], vf[2][2];
    if (producer) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int key = wave * 32 + u * 16 + frow;
            const int kc = key < N ? key : N - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[u][ks] = *(const bf16x8*)(base + (long long)kc * ld + D + (ks * 4 + fq) * 8);
                vf[u][ks] = *(const bf16x8*)(base + (long long)kc * ld + 2 * D + (ks * 4 + fq) * 8);
            }
        }
    }
    {
        __amdgpu_buffer_rsrc_t rQ = make_rsrc(base, ((long long)(N - 1) * ld + 64) * 2);
        __amdgpu_buffer_rsrc_t rK = make_rsrc(base + D, ((long long)(N - 1) * ld + 64) * 2);
        __amdgpu_buffer_rsrc_t rO = make_rsrc(dob, ((long long)(N - 1) * D + 64) * 2);
        const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
        for (int p = wave; p < nrow / 8; p += 8) {
            const int r = p * 8 + srow;
            const unsigned vq = r < N ? (unsigned)((r * ld + schunk * 8) * 2) : AIM_OOB;
            stage_piece(rQ, sQ + p * 1024, vq);
            stage_piece(rK, sK + p * 1024, vq);
            stage_piece(rO, sO + p * 1024, r < N ? (unsigned)((r * D + schunk * 8) * 2) : AIM_OOB);
        }
        // delta = rowsum(dO o O): 4 lanes per query, 16 head-dim elements each.  ALL loads of the prologue are issued before
        // anything waits (one HBM round trip for the LDS-DMA images, the dO / O rows, L and the K / V fragments below)
        bf16x8 da[2][2], oa2[2][2];
        float lq[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = it * 128 + (tid >> 2), part = tid & 3;
            const int qc = q < N ? q : N - 1;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                da[it][c] = *(const bf16x8*)(dob + (long long)qc * D + part * 16 + c * 8);
                oa2[it][c] = *(const bf16x8*)(ob + (long long)qc * D + part * 16 + c * 8);
            }
            lq[it] = lse[((long long)bt * H + h) * N + qc];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = it * 128 + (tid >> 2), part = tid & 3;
            float dl = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += (float)da[it][c][e] * (float)oa2[it][c][e];
            dl += __shfl_xor(dl, 1, 64);
            dl += __shfl_xor(dl, 2, 64);
            if (part == 0 && q < nrow) {
                sD[q] = q < N ? dl : 0.f;
                sL[q] = q < N ? lq[it] * LOG2E : 0.f;
            }
        }
    }
    // producers: dK / dV of their 32 keys; consumer: dQ of one 64-query block (the same registers serve either role)
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    FST(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    FST(2);

    const int nqb = (nrow + 63) >> 6;
    for (int qb = 0; qb <= nqb; ++qb) {
        if (producer && qb < nqb) {
            AIM_LDS char* ds_img = sDS + (qb & 1) * nrow * 128;
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                const int qs = qb * 2 + hs;                      // 32-query step
                if (qs * 32 >= nrow) break;
                bf16x4 ta[4], tb[4], tc[4], td[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int r0 = qs * 32 + fq * 4 + (frow >> 2);
                    const int ch = dt * 2 + ((frow & 3) >> 1), half = (frow & 1) * 8;
                    ta[dt] = lds_read_tr4(sQ + swz_off(r0, ch) + half);
                    tb[dt] = lds_read_tr4(sQ + swz_off(r0 + 16, ch) + half);
                    tc[dt] = lds_read_tr4(sO + swz_off(r0, ch) + half);
                    td[dt] = lds_read_tr4(sO + swz_off(r0 + 16, ch) + half);
                }
                bf16x8 pf[2], dsf[2];
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int qrow = (2 * qs + w) * 16;
                    bf16x8 qa[2], oa[2];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        qa[ks] = lds_read8(sQ + swz_off(qrow + frow, ks * 4 + fq));
                        oa[ks] = lds_read8(sO + swz_off(qrow + frow, ks * 4 + fq));
                    }
                    float Lr[4], Dr[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        Lr[e] = sL[qrow + fq * 4 + e];
                        Dr[e] = sD[qrow + fq * 4 + e];
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[ks], kf[u][ks], s, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa[ks], vf[u][ks], dp, 0, 0, 0);
                        }
                        bf16x4 ds4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float p = __builtin_amdgcn_exp2f(s[e] * C2 - Lr[e]);
                            pf[u][w * 4 + e] = (bf16_t)p;
                            const bf16_t d = (bf16_t)(p * (dp[e] - Dr[e]));
                            dsf[u][w * 4 + e] = d;
                            ds4[e] = d;
                        }
                        // dS[q = qrow + 4 fq + e][key = 32 wave + 16 u + frow] -> image row = key, 4 consecutive queries
                        const int ql = hs * 32 + w * 16 + fq * 4;                    // query inside the 64-query block
                        *(AIM_LDS bf16x4*)(ds_img + swz_off(wave * 32 + u * 16 + frow, ql >> 3) + (ql & 7) * 2) = ds4;
                    }
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    bf16x8 qt8, ot8;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        qt8[e] = ta[dt][e];
                        qt8[4 + e] = tb[dt][e];
                        ot8[e] = tc[dt][e];
                        ot8[4 + e] = td[dt][e];
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        acc[8 + dt * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ot8, pf[u], acc[8 + dt * 2 + u], 0, 0, 0);   // dV
                        acc[dt * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt8, dsf[u], acc[dt * 2 + u], 0, 0, 0);         // dK
                    }
                }
            }
        }
        if (consumer && qb > 0) {
            // dQ of block qb-1 (its dS image was completed before the barrier that ended the previous iteration)
            const int pb = qb - 1;
            const AIM_LDS char* ds_img = sDS + (pb & 1) * nrow * 128;
            const int nqt = min(4, (N - pb * 64 + 15) >> 4);              // query tiles of this block that hold a query < N
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kk = 0; kk < nkb; ++kk) {
                const int r0 = kk * 32 + fq * 4 + (frow >> 2);
                bf16x8 ktf[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int ch = dt * 2 + ((frow & 3) >> 1), half = (frow & 1) * 8;
                    const bf16x4 a = lds_read_tr4(sK + swz_off(r0, ch) + half);
                    const bf16x4 b = lds_read_tr4(sK + swz_off(r0 + 16, ch) + half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ktf[dt][e] = a[e];
                        ktf[dt][4 + e] = b[e];
                    }
                }
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) {
                    if (qt < nqt) {                                         // wave-uniform
                        const int ch = qt * 2 + ((frow & 3) >> 1), half = (frow & 1) * 8;
                        const bf16x4 a = lds_read_tr4(ds_img + swz_off(r0, ch) + half);
                        const bf16x4 b = lds_read_tr4(ds_img + swz_off(r0 + 16, ch) + half);
                        bf16x8 dsT;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            dsT[e] = a[e];
                            dsT[4 + e] = b[e];
                        }
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
                            acc[qt * 4 + dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[dt], dsT, acc[qt * 4 + dt], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const int q = pb * 64 + qt * 16 + frow;
                if (qt < nqt) {
                    // lane holds dQ[q][d = 16 dt + 4 fq + e]; tiles (dt, dt+1) paired across even / odd 16-lane rows: 16-byte stores
                    bf16_t* op = dqkv + ((long long)bt * N + (q < N ? q : 0)) * ld + h * 64 + ((fq & 1) ? 16 + (fq - 1) * 4 : fq * 4);
#pragma unroll
                    for (int dt = 0; dt < 4; dt += 2) {
                        const f32x4 a = acc[qt * 4 + dt] * 0.125f, b = acc[qt * 4 + dt + 1] * 0.125f;
                        const bf16x8 v = pair_rows16(pack4(a[0], a[1], a[2], a[3]), pack4(b[0], b[1], b[2], b[3]));
                        if (q < N) *(bf16x8*)(op + dt * 16) = v;
                    }
                }
            }
        }
        __syncthreads();
#ifdef AIM_X_STAMPS
        if (qb == 0) FST(3);
#endif
    }
    FST(4);
    if (producer) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int key = wave * 32 + u * 16 + frow;
            bf16_t* op = dqkv + ((long long)bt * N + (key < N ? key : 0)) * ld + h * 64 + ((fq & 1) ? 16 + (fq - 1) * 4 : fq * 4);
#pragma unroll
            for (int dt = 0; dt < 4; dt += 2) {
                const f32x4 k0 = acc[dt * 2 + u] * 0.125f, k1 = acc[(dt + 1) * 2 + u] * 0.125f;
                const f32x4 v0 = acc[8 + dt * 2 + u], v1 = acc[8 + (dt + 1) * 2 + u];
                const bf16x8 vk = pair_rows16(pack4(k0[0], k0[1], k0[2], k0[3]), pack4(k1[0], k1[1], k1[2], k1[3]));
                const bf16x8 vv = pair_rows16(pack4(v0[0], v0[1], v0[2], v0[3]), pack4(v1[0], v1[1], v1[2], v1[3]));
                if (key < N) {
                    *(bf16x8*)(op + D + dt * 16) = vk;
                    *(bf16x8*)(op + 2 * D + dt * 16) = vv;
                }
            }
        }
    }
#ifdef AIM_X_STAMPS
    FST(5);
    if (stamps && blockIdx.x % 997 == 0 && lane == 0 && (wave == 0 || wave == 7)) {
        unsigned long long* o = stamps + ((blockIdx.x / 997) * 2 + (wave == 7)) * 6;
        for (int i = 0; i < 6; ++i) o[i] = tstamp[i];
    }
#endif
}

}  // namespace

// workspace-free: delta is written into the caller-provided `delta` buffer ([BT, H, N] f32)
extern "C" int aim_attn_bwd(const aim_bf16* qkv, const aim_bf16* out, const aim_bf16* dout, const float* lse,
                            float* delta, aim_bf16* dqkv, int BT, int N, int H, void* stream) {
    AIM_CHECK_ARG(BT > 0 && N > 0 && H > 0 && N <= 288, "attn_bwd: unsupported shape BT=%d N=%d H=%d (N <= 288)", BT, N, H);
    AIM_CHECK_ARG(qkv && out && dout && lse && delta && dqkv, "attn_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    // AIM_ATTN_BWD_FUSED=1 (64 <= N <= 224, ViT-B/16): the fused single-pass kernel.  Measured on MI355X (512 x 12 x 197 x 64):
    // 0.56-0.60 ms against 0.54-0.58 ms for the two kernels stand-alone, and the same 57.85 ms whole step (the backward GEMMs
    // beside it run 5 % faster, the attention itself slower): it moves 1/3 less HBM traffic and issues 5 products instead of
    // 7, but runs one workgroup per CU (145 KB of LDS), so its 5.6 us load-bound prologue and its VALU-latency-bound
    // 1.4 us per 32-query step are exposed (tools/bench_attn.py STAMPS=1).  Kept as the starting point for a persistent,
    // software-pipelined version; the two-kernel form stays the default.
    static const bool fused_on = [] { const char* e = getenv("AIM_ATTN_BWD_FUSED"); return e && atoi(e) != 0; }();
    if (fused_on && N <= 224 && N >= 64) {
        const int nkb = (N + 31) / 32;
        const int nrow = nkb * 32;
        const int lds = 5 * nrow * 128 + nrow * 8;
        static const int stagger = [] { const char* e = getenv("AIM_ATTN_STAGGER"); return e ? atoi(e) : 0; }();
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL(attn_bwd_fused_kernel, dim3(BT * H), dim3(512), lds, st, (const bf16_t*)qkv, (const bf16_t*)out,
                           (const bf16_t*)dout, lse, (bf16_t*)dqkv, N, H, nkb, stagger
#ifdef AIM_X_STAMPS
                           , (unsigned long long*)delta      // diagnostic build: the (unused) delta scratch receives time stamps
#endif
        );
        AIM_CHECK_LAUNCH("aim_attn_bwd(fused)");
        return 0;
    }
    const int nkt = ((N + 31) / 32) * 2;   // 16-key tiles, even
    const int nq32 = ((N + 31) / 32) * 32;
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(BT * H), dim3(512), nkt * 16 * 128 * 2, st, (const bf16_t*)qkv,
                       (const bf16_t*)out, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, N, H, nkt);
    AIM_CHECK_LAUNCH("aim_attn_bwd(dq)");
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(BT * H), dim3(256), nq32 * 128 * 2 + nq32 * 8, st, (const bf16_t*)qkv,
                       (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, N, H, nq32);
    AIM_CHECK_LAUNCH("aim_attn_bwd(dkv)");
    return 0;
}
