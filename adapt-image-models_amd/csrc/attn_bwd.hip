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

}  // namespace

// workspace-free: delta is written into the caller-provided `delta` buffer ([BT, H, N] f32)
extern "C" int aim_attn_bwd(const aim_bf16* qkv, const aim_bf16* out, const aim_bf16* dout, const float* lse,
                            float* delta, aim_bf16* dqkv, int BT, int N, int H, void* stream) {
    AIM_CHECK_ARG(BT > 0 && N > 0 && H > 0 && N <= 288, "attn_bwd: unsupported shape BT=%d N=%d H=%d (N <= 288)", BT, N, H);
    AIM_CHECK_ARG(qkv && out && dout && lse && delta && dqkv, "attn_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
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
