// Spatial multi-head self-attention backward (dQ, dK, dV), probabilities recomputed from the
// forward's log-sum-exp.  gfx950 only.
//
// Autograd counterpart of reference vit_clip.py:139-156 (the reference relies on torch autograd
// through bmm/softmax/bmm and keeps the [BT,H,N,N] probabilities alive for it).
//
// 65 <= N <= 224 tokens (ViT-B/16): ONE persistent, software-pipelined kernel (attn_bwd_pipe_kernel, below): a single pass
// over the operands and five MFMA products.  Other N (ViT-L/14's 257, tiny test shapes), or AIM_ATTN_BWD_PIPE=0:
// two kernels per call, one workgroup per (frame, head) each (dq: 8 waves; dkv: 4 waves):
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
#include <type_traits>

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

    const int bid = (int)AIM_REV_BLOCK;          // (frame, head) items from the last one down: aim_common.h
    const int bt = bid / H, h = bid - bt * H;
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
                // the dP accumulators start from -delta (this lane's query): dS = P o (dP - delta) without the subtraction
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{-dl, -dl, -dl, -dl};
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
                    dsf[u * 4 + e] = (bf16_t)(p * dp[e]);
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

    const int bid = (int)AIM_REV_BLOCK;          // (frame, head) items from the last one down: aim_common.h
    const int bt = bid / H, h = bid - bt * H;
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
            sD[i] = i < N ? -delta[((long long)bt * H + h) * N + i] : 0.f;      // negated: the dP accumulators start from it
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
                    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
                    f32x4 dp = f32x4{cur.Dr[w][0], cur.Dr[w][1], cur.Dr[w][2], cur.Dr[w][3]};       // -delta
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
                        dsf[u][w * 4 + e] = (bf16_t)(p * dp[e]);
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
// Pipelined fused backward (65 <= N <= 224: ViT-B/16's 197 tokens): ONE pass over Q, K, V, dO, O per (frame, head) and the five
// products of flash attention's backward instead of the two kernels' seven, by a PERSISTENT workgroup per CU (8 waves) whose
// loads run one to two 64-query blocks ahead of the arithmetic, across (frame, head) items.  An item moves 200 KB (read q, k,
// v, dO, O; write dq, dk, dv) for 25 MFLOP: at 1/256 of the chip's HBM rate that is ~9.4 us per item, so the kernel is worth
// as much as its loads overlap its arithmetic.  (A one-shot fused form, one workgroup per item with whole Q / dO / K images
// in LDS, was measured first: 0.58 ms against the two kernels' 0.54 -- its 5.6 us load-bound prologue was exposed.)  This one
// keeps only 64-query CHUNKS of Q and dO in a 2-slot LDS ring; dS crosses LDS once as bf16, nothing is reduced across waves,
// no atomics:
//   tick T = (item k, query block qb), flattened over the workgroup's items.  In tick T, after every wave has waited for its
//   own loads of the previous tick (vmcnt(0)) and the workgroup barrier,
//     every wave   ISSUES its share of chunk T+1's LDS-DMA (one Q and one dO piece per wave); at an item's second tick also
//                  of the next item's K image, at its last tick the next item's V row fragments (producers, into registers);
//     wave 7       (it owns no keys) finishes delta / L of chunk T+1 from registers loaded a tick earlier (dO and O row parts
//                  straight from global memory: the ring is not involved) and requests the ones of chunk T+2;
//     every wave   dQ of chunk T-1: wave w owns query tile w >> 1 and the two 16-wide d tiles 2 (w & 1), 2 (w & 1) + 1 and
//                  sums over all keys: dS^T fragments out of dS image (T-1) & 1 and K^T fragments out of the item's K image,
//                  both by transposing reads (one consumer wave doing all sixteen tiles was the critical path: 3.7 us a tick);
//     producers    (waves 0 .. nkb-1, 32 keys each; V fragments in registers, K fragments re-read from the item's K image)
//                  chunk T:  S, dP, P, dS, dV^T += dO^T P, dK^T += Q^T dS, dS -> [key][64 query] image T & 1;  dK / dV of an
//                  item are stored at the next item's first tick, so the stores have a whole tick before the next vmcnt(0).
// The tick loop is unrolled by two so that ring slot, L/delta slot and dS image are compile-time constants (with a run-time
// parity the compiler cannot tell the slot being filled from the slot being read and drains the LDS-DMA before every read).
// Item positions (frame, head) advance incrementally in SGPRs: no division in the loop.
// LDS: ring 2 x 16 KiB + K images 2 x 28 KiB + dS 2 x 28 KiB + L/delta 1 KiB = 145 KiB at N = 197.  One barrier per tick.
constexpr int PF_SLOT = 2 * 64 * 128;            // one ring slot: Q chunk [64][128 B] + dO chunk [64][128 B]

struct PfPos {                                   // a tick's item: workgroup-local index, query block, frame, head
    int k, qb, bt, h;
};

// XT ("extra tile"): N = 64 j + r with 1 <= r <= 16 and j >= 2 (197 = 3 x 64 + 5).  The r left-over queries would cost a
// whole tick of their own (barrier, loads, a 64-row chunk that is 92 % padding: 3.0 of an item's 15 us); instead they ride
// with the item's LAST chunk as a fifth 16-query tile -- their Q / dO rows, L / delta and dS image live in small
// single-buffered LDS areas beside the ring (written and read once per item, three ticks apart), the producers run one more
// half step on them, and their dQ goes to waves 2 and 3.
template <bool XT>
__global__ __launch_bounds__(512) void attn_bwd_pipe_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                            const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                            bf16_t* __restrict__ dqkv, int N, int H, int nkb, int items
#ifdef AIM_X_STAMPS
                                                            , unsigned long long* stamps
#endif
) {
#ifdef AIM_X_STAMPS      // diagnostic build: 100 MHz stamps of ticks 8..15 of workgroup 0, waves 0 and 7 (tools/bench_attn.py STAMPS=2)
    unsigned long long tst[8][5];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) tst[a][b] = 0;
#define PST(b) do { if (T >= 8 && T < 16) { __builtin_amdgcn_sched_barrier(0); const unsigned long long v_ = __builtin_amdgcn_s_memrealtime(); _Pragma("unroll") for (int a_ = 0; a_ < 8; ++a_) if (a_ == T - 8) tst[a_][b] = v_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define PST(b)
#endif
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int nrow = nkb * 32;                                  // keys (and query rows) padded alike
    AIM_LDS char* sRing = (AIM_LDS char*)smem_raw;              // 2 slots
    AIM_LDS float* sLD = (AIM_LDS float*)(sRing + 2 * PF_SLOT); // 2 x {L[64], delta[64]}
    AIM_LDS char* sKimg = sRing + 2 * PF_SLOT + 1024;           // 2 x [nrow][128 B]: item k in buffer k & 1
    AIM_LDS char* sDS = sKimg + 2 * nrow * 128;                 // 2 x [key][64 queries] bf16
    AIM_LDS char* sXQ = sDS + 2 * nrow * 128;                   // XT: Q [16][128 B] + dO [16][128 B] of the extra tile
    AIM_LDS char* sXDS = sXQ + 4096;                            // XT: [key][16 queries] bf16, 32-byte rows
    AIM_LDS float* sXLD = (AIM_LDS float*)(sXDS + nrow * 32);   // XT: L[16], -delta[16]

    const int D = H * 64, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    const bool producer = wave < nkb;
    const bool deltaw = wave == 7;                               // (nkb <= 7: wave 7 owns no keys)
    const int nqb = XT ? N >> 6 : (nrow + 63) >> 6;              // ticks per item (>= 2: N >= 65; XT: N >= 129)
    const int xq0 = nqb * 64;                                    // XT: first query of the extra tile
    const int nit = ((int)items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nt = nit * nqb;
    const int gbt = (int)gridDim.x / H, gh = (int)gridDim.x - gbt * H;      // item stride of the workgroup as (frames, heads)

    auto advance = [&](PfPos& p) {
        if (++p.qb == nqb) {
            p.qb = 0;
            ++p.k;
#ifdef AIM_X_FWDROWS
            p.bt += gbt;
            p.h += gh;
            if (p.h >= H) {
                p.h -= H;
                ++p.bt;
            }
#else
            p.bt -= gbt;        // the items are walked from the LAST (frame, head) down (aim_common.h, AIM_REV_BLOCK)
            p.h -= gh;
            if (p.h < 0) {
                p.h += H;
                --p.bt;
            }
#endif
        }
    };
    f32x4 acc[16];                     // producers: dK / dV of their 32 keys
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // LDS-DMA of a chunk into a ring slot: one Q piece and one dO piece per wave
    auto issue_chunk = [&](const PfPos& p, AIM_LDS char* slot) {
        const bf16_t* base = qkv + (long long)p.bt * N * ld + p.h * 64;
        const bf16_t* dob = dout + (long long)p.bt * N * D + p.h * 64;
        const aim_rsrc_words rQ = make_rsrc_words(base, ((long long)(N - 1) * ld + 64) * 2);
        const aim_rsrc_words rO = make_rsrc_words(dob, ((long long)(N - 1) * D + 64) * 2);
        const int r = p.qb * 64 + wave * 8 + srow;
        stage_piece_asm(rQ, slot + wave * 1024, r < N ? (unsigned)((r * ld + schunk * 8) * 2) : AIM_OOB);
        stage_piece_asm(rO, slot + 8192 + wave * 1024, r < N ? (unsigned)((r * D + schunk * 8) * 2) : AIM_OOB);
        if (XT && p.qb == nqb - 1 && wave < 4) {      // the extra tile: waves 0, 1 its two Q pieces, waves 2, 3 the dO pieces
            const int rx = xq0 + (wave & 1) * 8 + srow;
            if (wave < 2) stage_piece_asm(rQ, sXQ + (wave & 1) * 1024, rx < N ? (unsigned)((rx * ld + schunk * 8) * 2) : AIM_OOB);
            else stage_piece_asm(rO, sXQ + 2048 + (wave & 1) * 1024, rx < N ? (unsigned)((rx * D + schunk * 8) * 2) : AIM_OOB);
        }
    };
    // delta = rowsum(dO o O) and L of a chunk, by the wave that owns no keys (wave 7), straight from global memory: 8 lanes
    // per query row, 8 rows per load, the operands parked in that wave's otherwise unused accumulator registers.  (With every
    // wave loading its own share, each paid ~0.5 us per tick for three loads, the shuffles and the address arithmetic.)
    float lreg = 0.f, lregx = 0.f;
    bf16x8 vf[2][2] = {}, vfn[2][2] = {};          // producers: V fragments of the item / of the next one.  XT: wave 7 (no
                                                   // keys) parks the extra tile's O / dO row parts in vfn[g][0] / vfn[g][1]
    auto issue_delta = [&](const PfPos& p) {
        const bf16_t* ob = out + ((long long)p.bt * N * D + p.h * 64);       // wave-uniform bases, 32-bit lane offsets
        const bf16_t* dob = dout + ((long long)p.bt * N * D + p.h * 64);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int q = p.qb * 64 + g * 8 + srow;
            const unsigned off = (unsigned)((q < N ? q : N - 1) * D + (lane & 7) * 8);
            acc[g] = __builtin_bit_cast(f32x4, *(const bf16x8*)(ob + off));
            acc[8 + g] = __builtin_bit_cast(f32x4, *(const bf16x8*)(dob + off));
        }
        const int ql = p.qb * 64 + lane;
        lreg = (lse + ((long long)p.bt * H + p.h) * N)[ql < N ? ql : N - 1];
        if (XT && p.qb == nqb - 1) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int q = xq0 + g * 8 + srow;
                const unsigned off = (unsigned)((q < N ? q : N - 1) * D + (lane & 7) * 8);
                vfn[g][0] = *(const bf16x8*)(ob + off);
                vfn[g][1] = *(const bf16x8*)(dob + off);
            }
            const int qx = xq0 + (lane & 15);
            lregx = (lse + ((long long)p.bt * H + p.h) * N)[qx < N ? qx : N - 1];
        }
    };
    auto finish_delta = [&](const PfPos& p, AIM_LDS float* sl) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const bf16x8 o8 = __builtin_bit_cast(bf16x8, acc[g]), d8 = __builtin_bit_cast(bf16x8, acc[8 + g]);
            float dl = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += (float)d8[e] * (float)o8[e];
            // sum over the row's 8 lanes with DPP (quad_perm xor 1, xor 2, then row_half_mirror: the other quad of the eight);
            // the ds_bpermute form of __shfl_xor made this wave the tick's critical path
            dl += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, dl), 0xB1, 0xF, 0xF, true));
            dl += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, dl), 0x4E, 0xF, 0xF, true));
            dl += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, dl), 0x141, 0xF, 0xF, true));
            // stored NEGATED: the producers start the dP accumulators from -delta, so dS = P o dP needs no subtraction
            if ((lane & 7) == 0) sl[64 + g * 8 + srow] = p.qb * 64 + g * 8 + srow < N ? -dl : 0.f;
        }
        sl[lane] = p.qb * 64 + lane < N ? lreg * LOG2E : 0.f;
        if (XT && p.qb == nqb - 1) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const bf16x8 o8 = vfn[g][0], d8 = vfn[g][1];
                float dl = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += (float)d8[e] * (float)o8[e];
                dl += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, dl), 0xB1, 0xF, 0xF, true));
                dl += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, dl), 0x4E, 0xF, 0xF, true));
                dl += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, dl), 0x141, 0xF, 0xF, true));
                if ((lane & 7) == 0) sXLD[16 + g * 8 + srow] = xq0 + g * 8 + srow < N ? -dl : 0.f;
            }
            if (lane < 16) sXLD[lane] = xq0 + lane < N ? lregx * LOG2E : 0.f;
        }
    };
    auto issue_kimg = [&](const PfPos& p) {
        const bf16_t* base = qkv + (long long)p.bt * N * ld + p.h * 64;
        const aim_rsrc_words rK = make_rsrc_words(base + D, ((long long)(N - 1) * ld + 64) * 2);
        AIM_LDS char* img = sKimg + (p.k & 1) * nrow * 128;
        for (int pc = wave; pc < nrow / 8; pc += 8) {
            const int r = pc * 8 + srow;
            stage_piece_asm(rK, img + pc * 1024, r < N ? (unsigned)((r * ld + schunk * 8) * 2) : AIM_OOB);
        }
    };
    auto issue_v = [&](const PfPos& p) {        // producers: V row fragments of an item -> vfn
        const bf16_t* base = qkv + (long long)p.bt * N * ld + p.h * 64;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int key = wave * 32 + u * 16 + frow;
            const int kc = key < N ? key : N - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) vfn[u][ks] = *(const bf16x8*)(base + (long long)kc * ld + 2 * D + (ks * 4 + fq) * 8);
        }
    };

    auto store_dkv = [&](const PfPos& p) {      // producers: dK, dV of an item from acc
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int key = wave * 32 + u * 16 + frow;
            bf16_t* op = dqkv + ((long long)p.bt * N + (key < N ? key : 0)) * ld + p.h * 64 + ((fq & 1) ? 16 + (fq - 1) * 4 : fq * 4);
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
    };

    // dQ tile pair += sum over the item's keys of dS^T K: fragments of 32 keys at a time by transposing reads.  The key loop
    // is latency-bound (six dependent-free reads, two MFMAs, repeat: every wave spent ~1 us of a 3.7 us tick in it), so it
    // runs THREE key blocks per trip: eighteen reads in flight, then six MFMAs.  ds_at(row) -> the lane's dS bytes in key row `row`.
    auto dq_keys = [&](auto ds_at, const AIM_LDS char* sK, int chk0, int chk1, int half, int rl, f32x4& dq0, f32x4& dq1) {
        auto step = [&](int kk, bf16x4 (&f)[6]) {
            const int r0 = kk * 32 + rl;
            f[0] = lds_read_tr4(ds_at(r0));
            f[1] = lds_read_tr4(ds_at(r0 + 16));
            f[2] = lds_read_tr4(sK + swz_off(r0, chk0) + half);
            f[3] = lds_read_tr4(sK + swz_off(r0 + 16, chk0) + half);
            f[4] = lds_read_tr4(sK + swz_off(r0, chk1) + half);
            f[5] = lds_read_tr4(sK + swz_off(r0 + 16, chk1) + half);
        };
        auto mul = [&](const bf16x4 (&f)[6]) {
            bf16x8 dsT, kt0, kt1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dsT[e] = f[0][e];
                dsT[4 + e] = f[1][e];
                kt0[e] = f[2][e];
                kt0[4 + e] = f[3][e];
                kt1[e] = f[4][e];
                kt1[4 + e] = f[5][e];
            }
            dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt0, dsT, dq0, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt1, dsT, dq1, 0, 0, 0);
        };
        int kk = 0;
        for (; kk + 3 <= nkb; kk += 3) {
            bf16x4 fa[6], fb[6], fc[6];
            step(kk, fa);
            step(kk + 1, fb);
            step(kk + 2, fc);
            mul(fa);
            mul(fb);
            mul(fc);
        }
        for (; kk < nkb; ++kk) {
            bf16x4 fa[6];
            step(kk, fa);
            mul(fa);
        }
    };

    // positions of ticks T-1, T, T+1, T+2
    PfPos prv{0, 0, 0, 0}, cur, nx1, nx2;
    cur.k = 0;
    cur.qb = 0;
#ifdef AIM_X_FWDROWS
    const int first = (int)blockIdx.x;
#else
    const int first = items - 1 - (int)blockIdx.x;
#endif
    cur.bt = first / H;
    cur.h = first - cur.bt * H;
    nx1 = cur;
    advance(nx1);
    nx2 = nx1;
    advance(nx2);

    // ---- prologue: chunk 0 and its delta, item 0's K image and V fragments; then the delta operands of chunk 1
    issue_chunk(cur, sRing);
    if (deltaw) issue_delta(cur);
    issue_kimg(cur);
    if (producer) issue_v(cur);
    __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0) lgkmcnt(0), as a builtin: see the tick's wait
    if (deltaw) {
        finish_delta(cur, sLD);
        if (nt > 1) issue_delta(nx1);
    }

    // one tick; P = T & 1 as a compile-time constant
    auto tick = [&](int T, auto PAR) {
        constexpr int P = decltype(PAR)::value;
        PST(0);
        // vmcnt(0) lgkmcnt(0) as a BUILTIN so that the compiler's scoreboard sees it: behind an asm wait it still counts the
        // register loads of the previous tick (O / dO parts, V fragments) as outstanding and waits for them with a count that
        // ignores the LDS-DMA issued since -- i.e. for the DMA this tick has just started
        __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_sched_barrier(0);
        PST(1);
        __builtin_amdgcn_s_barrier();
        PST(2);
        const bool live = T < nt;
        // item switch first: the loads these registers come from were issued a tick ago, nothing else is in flight yet
        if (producer && live && cur.qb == 0) {
            if (T > 0) {
                store_dkv(prv);
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) vf[u][ks] = vfn[u][ks];
        }
        if (producer && !live && nt > 0) store_dkv(prv);         // the last item's dK / dV (T == nt)
        // The tick's loads
        if (T + 1 < nt) {
            if (deltaw) finish_delta(nx1, sLD + (1 - P) * 128);
            issue_chunk(nx1, sRing + (1 - P) * PF_SLOT);
        }
        if (deltaw && T + 2 < nt) issue_delta(nx2);
        if (live) {
            if (cur.qb == 1 && cur.k + 1 < nit) {        // the next item's K image (nx1 / nx2 may still be inside this item)
                PfPos nk = cur;
                nk.qb = nqb - 1;
                advance(nk);
                issue_kimg(nk);
            }
            if (cur.qb == nqb - 1 && cur.k + 1 < nit && producer) issue_v(nx1);
        }
        PST(3);
        if (T > 0) {
            // dQ of chunk T-1 (its dS image was completed before this tick's barrier): query tile wave >> 1, d tiles dt0, dt0 + 1
            const int qt = wave >> 1, dt0 = (wave & 1) * 2;
            const int nqt = min(4, (N - prv.qb * 64 + 15) >> 4);          // query tiles of the chunk that hold a query < N
            if (qt < nqt) {
                // lane-derived LDS offsets are recomputed per section: kept as loop invariants they (and the producer
                // section's) cost ~40 VGPRs across the tick loop and spill, and a scratch reload waits on vmcnt
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const int frow = ln & 15, fq = ln >> 4;
                const AIM_LDS char* sK = sKimg + (prv.k & 1) * nrow * 128;
                const AIM_LDS char* ds_img = sDS + (1 - P) * nrow * 128;
                f32x4 dq0 = f32x4{0.f, 0.f, 0.f, 0.f}, dq1 = dq0;
                const int chq = qt * 2 + ((frow & 3) >> 1), half = (frow & 1) * 8;
                const int chk0 = dt0 * 2 + ((frow & 3) >> 1), chk1 = chk0 + 2;
                dq_keys([&](int r) { return ds_img + swz_off(r, chq) + half; }, sK, chk0, chk1, half, fq * 4 + (frow >> 2), dq0, dq1);
                // lane holds dQ[q][d = 16 dt + 4 fq + e]; tiles (dt0, dt0+1) paired across even / odd 16-lane rows: 16-byte stores
                const int q = prv.qb * 64 + qt * 16 + frow;
                bf16_t* op = dqkv + ((long long)prv.bt * N + (q < N ? q : 0)) * ld + prv.h * 64 + ((fq & 1) ? 16 + (fq - 1) * 4 : fq * 4);
                const f32x4 a = dq0 * 0.125f, b = dq1 * 0.125f;
                const bf16x8 v = pair_rows16(pack4(a[0], a[1], a[2], a[3]), pack4(b[0], b[1], b[2], b[3]));
                if (q < N) *(bf16x8*)(op + dt0 * 16) = v;
            }
            if (XT && prv.qb == nqb - 1 && (wave >> 1) == 1) {
                // dQ of the extra tile of chunk T-1: waves 2 and 3, two d tiles each.  (Not wave 7: at an item's first tick its
                // delta work queues behind the producers' dK / dV stores and it is the tick's critical path already.)
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const int frow = ln & 15, fq = ln >> 4;
                const int dt0 = (wave & 1) * 2;
                const AIM_LDS char* sK = sKimg + (prv.k & 1) * nrow * 128;
                f32x4 dq0 = f32x4{0.f, 0.f, 0.f, 0.f}, dq1 = dq0;
                const int xoff = ((frow & 3) >> 1) * 16 + (frow & 1) * 8, half = (frow & 1) * 8;
                const int chk0 = dt0 * 2 + ((frow & 3) >> 1), chk1 = chk0 + 2;
                dq_keys([&](int r) { return sXDS + r * 32 + xoff; }, sK, chk0, chk1, half, fq * 4 + (frow >> 2), dq0, dq1);
                const int q = xq0 + frow;
                bf16_t* op = dqkv + ((long long)prv.bt * N + (q < N ? q : 0)) * ld + prv.h * 64 + ((fq & 1) ? 16 + (fq - 1) * 4 : fq * 4);
                const f32x4 a = dq0 * 0.125f, b = dq1 * 0.125f;
                const bf16x8 v = pair_rows16(pack4(a[0], a[1], a[2], a[3]), pack4(b[0], b[1], b[2], b[3]));
                if (q < N) *(bf16x8*)(op + dt0 * 16) = v;
            }
        }
        if (producer && live) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int frow = ln & 15, fq = ln >> 4;
            const AIM_LDS char* sQ = sRing + P * PF_SLOT;
            const AIM_LDS char* sO = sQ + 8192;
            const AIM_LDS float* sL = sLD + P * 128;
            const AIM_LDS float* sD = sL + 64;
            const AIM_LDS char* sK = sKimg + (cur.k & 1) * nrow * 128;
            AIM_LDS char* ds_img = sDS + P * nrow * 128;
            bf16x8 kf[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) kf[u][ks] = lds_read8(sK + swz_off(wave * 32 + u * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                if ((cur.qb * 2 + hs) * 32 >= nrow) break;       // the item's last 32-query step may be its chunk's first
                bf16x4 ta[4], tb[4], tc[4], td[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int r0 = hs * 32 + fq * 4 + (frow >> 2);
                    const int ch = dt * 2 + ((frow & 3) >> 1), half = (frow & 1) * 8;
                    ta[dt] = lds_read_tr4(sQ + swz_off(r0, ch) + half);
                    tb[dt] = lds_read_tr4(sQ + swz_off(r0 + 16, ch) + half);
                    tc[dt] = lds_read_tr4(sO + swz_off(r0, ch) + half);
                    td[dt] = lds_read_tr4(sO + swz_off(r0 + 16, ch) + half);
                }
                bf16x8 pf[2] = {}, dsf[2] = {};
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int qrow = (2 * hs + w) * 16;
                    // a 16-query tile wholly past N (the second half of the item's last 32-query step at N = 197) has zero
                    // Q / dO rows and L = delta = 0: its P and dS contribute nothing, so its products and exps are skipped
                    if (cur.qb * 64 + qrow >= N) continue;
                    bf16x8 qa[2], oa[2];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        qa[ks] = lds_read8(sQ + swz_off(qrow + frow, ks * 4 + fq));
                        oa[ks] = lds_read8(sO + swz_off(qrow + frow, ks * 4 + fq));
                    }
                    const f32x4 Lr = *(const AIM_LDS f32x4*)(sL + qrow + fq * 4);
                    const f32x4 Dr = *(const AIM_LDS f32x4*)(sD + qrow + fq * 4);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = Dr;        // Dr = -delta: dP - delta comes out of the MFMA
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
                            const bf16_t d = (bf16_t)(p * dp[e]);
                            dsf[u][w * 4 + e] = d;
                            ds4[e] = d;
                        }
                        // dS[q = qrow + 4 fq + e][key = 32 wave + 16 u + frow] -> image row = key, 4 consecutive queries
                        const int ql = qrow + fq * 4;                                    // query inside the 64-query chunk
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
            if (XT && cur.qb == nqb - 1) {
                // the extra tile: one 16-query half step (the upper 16 k-positions of the dV / dK products are zero)
                const AIM_LDS char* xQ = sXQ;
                const AIM_LDS char* xO = sXQ + 2048;
                bf16x4 ta[4], tc[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int r0 = fq * 4 + (frow >> 2);
                    const int ch = dt * 2 + ((frow & 3) >> 1), half = (frow & 1) * 8;
                    ta[dt] = lds_read_tr4(xQ + swz_off(r0, ch) + half);
                    tc[dt] = lds_read_tr4(xO + swz_off(r0, ch) + half);
                }
                bf16x8 pf[2] = {}, dsf[2] = {};
                bf16x8 qa[2], oa[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    qa[ks] = lds_read8(xQ + swz_off(frow, ks * 4 + fq));
                    oa[ks] = lds_read8(xO + swz_off(frow, ks * 4 + fq));
                }
                const f32x4 Lr = *(const AIM_LDS f32x4*)(sXLD + fq * 4);
                const f32x4 Dr = *(const AIM_LDS f32x4*)(sXLD + 16 + fq * 4);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = Dr;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[ks], kf[u][ks], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa[ks], vf[u][ks], dp, 0, 0, 0);
                    }
                    bf16x4 ds4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float p = __builtin_amdgcn_exp2f(s[e] * C2 - Lr[e]);
                        pf[u][e] = (bf16_t)p;
                        const bf16_t d = (bf16_t)(p * dp[e]);
                        dsf[u][e] = d;
                        ds4[e] = d;
                    }
                    *(AIM_LDS bf16x4*)(sXDS + (wave * 32 + u * 16 + frow) * 32 + fq * 8) = ds4;
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    bf16x8 qt8 = {}, ot8 = {};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        qt8[e] = ta[dt][e];
                        ot8[e] = tc[dt][e];
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        acc[8 + dt * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ot8, pf[u], acc[8 + dt * 2 + u], 0, 0, 0);
                        acc[dt * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt8, dsf[u], acc[dt * 2 + u], 0, 0, 0);
                    }
                }
            }
        }
        PST(4);
        prv = cur;
        cur = nx1;
        nx1 = nx2;
        advance(nx2);
    };
    for (int T = 0; T <= nt; T += 2) {
        tick(T, std::integral_constant<int, 0>{});
        if (T + 1 <= nt) tick(T + 1, std::integral_constant<int, 1>{});
    }
#ifdef AIM_X_STAMPS
    if (stamps && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7)) {
        unsigned long long* o = stamps + (wave == 7 ? 40 : 0);
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) o[a * 5 + b] = tst[a][b];
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
    // 65 <= N <= 224 (ViT-B/16's 197 tokens): the pipelined fused kernel; AIM_ATTN_BWD_PIPE=0 selects the two-kernel form.
    // Measured on MI355X (512 x 12 x 197 x 64, stand-alone): 0.39 ms against 0.54 ms; whole training step 53.8 against 54.7 ms.
    static const bool pipe_on = [] { const char* e = getenv("AIM_ATTN_BWD_PIPE"); return !e || atoi(e) != 0; }();
    if (pipe_on && N <= 224 && N >= 65) {
        const int nkb = (N + 31) / 32;
        const int nrow = nkb * 32;
        static const bool xt_on = [] { const char* e = getenv("AIM_ATTN_PIPE_XT"); return !e || atoi(e) != 0; }();
        const bool xt = xt_on && N >= 129 && ((N - 1) & 63) < 16;        // 64 j + 1 .. 64 j + 16, j >= 2
        const int lds = 2 * PF_SLOT + 4 * nrow * 128 + 2 * 128 * 4 + (xt ? 4096 + nrow * 32 + 128 : 0);
        static bool attr_set2 = false;
        if (!attr_set2) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_pipe_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)attn_bwd_pipe_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set2 = true;
        }
        const int items = BT * H;
        int cus = aim_device_cus();
        // The grid is persistent (one workgroup per CU for the whole launch), and the hardware deals workgroups to XCDs and
        // shader engines in launch order (workgroup i -> XCD i % 8 -> engine (i / 8) % 4) and makes a workgroup WAIT for a CU of
        // its engine even when other engines have one free (tools/cumask_probe.hip, tools/cumask_gemm.py).  With every CU
        // taken, the class-token chain's dozen small kernels on the side stream did not start until this launch ended
        // (tools/chain_probe.py: 1.4 ms of join stall per step); they run freely once EVERY engine keeps a CU free, i.e. at most
        // 7 x 4 = 28 workgroups per XCD: 32 CUs stay out of the grid (+12 % attention time, -1.3 ms join stall: net +0.9 %).
        // Fewer (10, 24) do not help: the first engines are still full.  AIM_ATTN_PIPE_RESERVE overrides the count.
        static const int reserve = [] { const char* e = getenv("AIM_ATTN_PIPE_RESERVE"); return e ? atoi(e) : 32; }();
        if (reserve > 0 && items > cus - reserve && cus - reserve >= 8) cus -= reserve;
        static const int grid_cap = [] { const char* e = getenv("AIM_ATTN_PIPE_GRID"); return e ? atoi(e) : 0; }();   // tests: few workgroups, many items each
        if (grid_cap > 0 && grid_cap < cus) cus = grid_cap;
        const int grid = items < cus ? items : cus;
        if (xt)
            hipLaunchKernelGGL(attn_bwd_pipe_kernel<true>, dim3(grid), dim3(512), lds, st, (const bf16_t*)qkv, (const bf16_t*)out,
                               (const bf16_t*)dout, lse, (bf16_t*)dqkv, N, H, nkb, items
#ifdef AIM_X_STAMPS
                               , (unsigned long long*)delta
#endif
            );
        else
            hipLaunchKernelGGL(attn_bwd_pipe_kernel<false>, dim3(grid), dim3(512), lds, st, (const bf16_t*)qkv, (const bf16_t*)out,
                               (const bf16_t*)dout, lse, (bf16_t*)dqkv, N, H, nkb, items
#ifdef AIM_X_STAMPS
                               , (unsigned long long*)delta
#endif
            );
        AIM_CHECK_LAUNCH("aim_attn_bwd(pipelined)");
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
