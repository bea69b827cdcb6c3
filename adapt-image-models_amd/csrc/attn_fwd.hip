// Spatial multi-head self-attention forward over the N tokens of one frame.  gfx950 only.
//
// Replaces reference vit_clip.py:139-156 (view/permute to [Nb,H,S,dh], q@k^T / sqrt(dh), softmax,
// @v, merge heads) for the spatial call at :264.  The [BT,H,N,N] score tensor the reference
// materialises (0.95 GB fp32 per layer at config 2) never leaves the CU.
//
// One workgroup (8 waves, two workgroups per CU) per (frame, head).  K and V of that head (N <= 288 keys x 64) are staged
// once into swizzled LDS images by buffer_load...lds (zero-filled past N).  Each wave takes 16-row
// query tiles round-robin:  S^T = K Q^T with the KEY on the MFMA row and the QUERY on the lane, so a
// query's whole score row sits in one lane quartet (2 shuffles per reduction) and the normalised
// P^T accumulators are, unchanged, the second operand of  O^T = V^T P^T  (V^T fragments come from
// the row-major V image through ds_read_b64_tr_b16).  O^T puts 4 consecutive head-dim elements of a
// query in one lane: 8-byte stores into the merged-heads [M, D] layout.
#include "aim_common.h"
#include "aim_kernels_internal.h"

#ifdef AIM_X_STAMPS      // diagnostic build only (tools/probe_attn.py): shader-clock stamps of one workgroup's waves
static unsigned long long* g_attn_probe = nullptr;
extern "C" int aim_attn_probe(void* buf) { g_attn_probe = (unsigned long long*)buf; return 0; }
#define ATT_STAMP(i) do { if (probe && bid == 3000) { __builtin_amdgcn_sched_barrier(0); stamps[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define ATT_STAMP(i)
#endif

namespace {

// NKT: number of 16-key tiles of the LDS images (even).  NFULL: key tiles below NFULL are known to lie entirely below N,
// so only tiles >= NFULL carry masking code (N = 197: NKT = 14, NFULL = 12; masking every tile costs ~1.5x the
// softmax's useful vector instructions in compares, selects and spilled condition masks).
// OUT8: the output is written as fp8 e4m3 bytes (inference: operand of the fp8 out_proj GEMM) instead of bf16
template <int NKT, int NFULL, bool OUT8 = false>
__global__ __launch_bounds__(512, 4) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                          float* __restrict__ lse, int N, int H
#ifdef AIM_X_STAMPS
                                                          , unsigned long long* probe
#endif
) {
#ifdef AIM_X_STAMPS
    unsigned long long stamps[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) stamps[i] = 0;
#endif
    const int bid = (int)AIM_REV_BLOCK;
    ATT_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    AIM_LDS char* sK = (AIM_LDS char*)smem_raw;
    AIM_LDS char* sV = sK + NKT * 16 * 128;

    const int bt = bid / H, h = bid - bt * H;
    const int D = H * 64, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    constexpr int NW = 8;                      // waves per workgroup
    const float C2 = 0.125f * 1.4426950408889634f;   // 1/sqrt(dh) * log2(e): softmax runs in base 2

    const bf16_t* base = qkv + (long long)bt * N * ld + h * 64;
    const int nqt = (N + 15) >> 4;
    // first query tile's fragments are requested before the K/V staging so their latency overlaps it
    bf16x8 qf[2];
    {
        const int q0 = wave * 16 + frow;
        const int qc = q0 < N ? q0 : N - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const bf16x8*)(base + (long long)qc * ld + (ks * 4 + fq) * 8);
    }
    {
        __amdgpu_buffer_rsrc_t rK = make_rsrc(base + D, ((long long)(N - 1) * ld + 64) * 2);
        __amdgpu_buffer_rsrc_t rV = make_rsrc(base + 2 * D, ((long long)(N - 1) * ld + 64) * 2);
        const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
        for (int p = wave; p < NKT * 2; p += NW) {
            const int key = p * 8 + srow;
            const unsigned voff = key < N ? (unsigned)((key * ld + schunk * 8) * 2) : AIM_OOB;
            stage_piece(rK, sK + p * 1024, voff);
            stage_piece(rV, sV + p * 1024, voff);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT_STAMP(1);
    __syncthreads();
    ATT_STAMP(2);

    for (int qt = wave; qt < nqt; qt += NW) {
        // the K/V fragments do not depend on qt: without this fence LICM hoists all 28 K fragments
        // (112 VGPRs) out of the loop and the kernel spills
        asm volatile("" ::: "memory");
        const int q = qt * 16 + frow;
#ifdef AIM_X_STAMPS
        const int sb = qt < NW ? 3 : 9;
#define ATT_STAMPQ(i) do { if (sb == 3) ATT_STAMP(3 + i); else ATT_STAMP(9 + i); } while (0)
#else
#define ATT_STAMPQ(i)
#endif
        ATT_STAMPQ(0);
        f32x4 s[NKT];
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 kf = lds_read8(sK + swz_off(t * 16 + frow, ks * 4 + fq));
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[t], 0, 0, 0);
            }
        }
        ATT_STAMPQ(1);
        // prefetch the next query tile of this wave
        if (qt + NW < nqt) {
            const int qn = (qt + NW) * 16 + frow;
            const int qc = qn < N ? qn : N - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const bf16x8*)(base + (long long)qc * ld + (ks * 4 + fq) * 8);
        }
        // softmax over keys (lane holds keys t*16 + fq*4 + e of query lane&15); only tiles that straddle N mask
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            if (t >= NFULL) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (t * 16 + fq * 4 + e >= N) s[t][e] = -INFINITY;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[t][e]);
        }
        mx = quad_max(mx);          // over the lane quartet of this query (permlane swaps, no LDS crossbar)
        const float mc = mx * C2;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = __builtin_amdgcn_exp2f(s[t][e] * C2 - mc);   // unnormalised, max = 1
                s[t][e] = p;
                sum += p;
            }
        sum = quad_sum(sum);
        const float inv = 1.0f / sum;
        if (fq == 0 && q < N && lse) lse[((long long)bt * H + h) * N + q] = mx * 0.125f + __logf(sum);

        ATT_STAMPQ(2);
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NKT / 2; ++kk) {
            bf16x8 pf;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pf[e] = (bf16_t)s[2 * kk][e];
                pf[4 + e] = (bf16_t)s[2 * kk + 1][e];
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                // V^T fragment: rows key0 + (i>>2), columns dt*16 + 4*(i&3) .. +3 of the V image (i = lane&15)
                const int r0 = (2 * kk) * 16 + fq * 4 + (frow >> 2);
                const int r1 = r0 + 16;
                const int ch = dt * 2 + ((frow & 3) >> 1), half = (frow & 1) * 8;
                const bf16x4 v0 = lds_read_tr4(sV + swz_off(r0, ch) + half);
                const bf16x4 v1 = lds_read_tr4(sV + swz_off(r1, ch) + half);
                bf16x8 vf;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vf[e] = v0[e];
                    vf[4 + e] = v1[e];
                }
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
            }
        }
        ATT_STAMPQ(3);
        if constexpr (OUT8) {
            // fp8 bytes: the lane owns 4 consecutive head-dim elements of tile dt -> one 4-byte store each
            unsigned char* op8 = (unsigned char*)out + ((long long)bt * N + (q < N ? q : 0)) * D + h * 64 + fq * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                if (q < N) *(unsigned*)(op8 + dt * 16) = pack4_fp8(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
        } else {
            // 16-byte stores: tiles (dt, dt+1) are paired across the even / odd 16-lane rows (aim_common.h pair_rows16), so a
            // lane writes 8 consecutive head-dim elements and a row's four lanes cover 64 contiguous bytes
            bf16_t* op = out + ((long long)bt * N + (q < N ? q : 0)) * D + h * 64 + ((fq & 1) ? 16 + (fq - 1) * 4 : fq * 4);
#pragma unroll
            for (int dt = 0; dt < 4; dt += 2) {
                const bf16x8 v = pair_rows16(pack4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv),
                                             pack4(o[dt + 1][0] * inv, o[dt + 1][1] * inv, o[dt + 1][2] * inv, o[dt + 1][3] * inv));
                if (q < N) *(bf16x8*)(op + dt * 16) = v;
            }
        }
        ATT_STAMPQ(4);
    }
#ifdef AIM_X_STAMPS
    stamps[15] = __builtin_amdgcn_s_memtime();
    if (probe && bid == 3000 && lane == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) probe[wave * 16 + i] = stamps[i];
    }
#endif
}

template <int NKT, int NFULL, bool OUT8 = false>
int launch_nf(const aim_bf16* qkv, aim_bf16* out, float* lse, int BT, int N, int H, hipStream_t st) {
    hipLaunchKernelGGL((attn_fwd_kernel<NKT, NFULL, OUT8>), dim3(BT * H), dim3(512), NKT * 16 * 128 * 2, st, (const bf16_t*)qkv,
                       (bf16_t*)out, lse, N, H
#ifdef AIM_X_STAMPS
                       , g_attn_probe
#endif
    );
    AIM_CHECK_LAUNCH("aim_attn_fwd");
    return 0;
}

// masking code only on the last two key tiles when N reaches into them, on every tile otherwise
template <int NKT, bool OUT8 = false>
int launch(const aim_bf16* qkv, aim_bf16* out, float* lse, int BT, int N, int H, hipStream_t st) {
    if ((N >> 4) >= NKT - 2) return launch_nf<NKT, NKT - 2, OUT8>(qkv, out, lse, BT, N, H, st);
    return launch_nf<NKT, 0, OUT8>(qkv, out, lse, BT, N, H, st);
}

}  // namespace

extern "C" int aim_attn_fwd(const aim_bf16* qkv, aim_bf16* out, float* lse, int BT, int N, int H, void* stream) {
    AIM_CHECK_ARG(BT > 0 && N > 0 && H > 0 && N <= 288, "attn_fwd: unsupported shape BT=%d N=%d H=%d (N <= 288)", BT, N, H);
    AIM_CHECK_ARG(qkv && out && lse, "attn_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (N <= 32) return launch<2>(qkv, out, lse, BT, N, H, st);
    if (N <= 64) return launch<4>(qkv, out, lse, BT, N, H, st);
    if (N <= 224) return launch<14>(qkv, out, lse, BT, N, H, st);
    return launch<18>(qkv, out, lse, BT, N, H, st);
}

extern "C" int aim_attn_fwd_fp8(const aim_bf16* qkv, uint8_t* out_fp8, float* lse, int BT, int N, int H, void* stream) {
    AIM_CHECK_ARG(BT > 0 && N > 0 && H > 0 && N <= 288, "attn_fwd_fp8: unsupported shape BT=%d N=%d H=%d (N <= 288)", BT, N, H);
    AIM_CHECK_ARG(qkv && out_fp8, "attn_fwd_fp8: null pointer");
    hipStream_t st = (hipStream_t)stream;
    aim_bf16* out = (aim_bf16*)out_fp8;
    if (N <= 32) return launch<2, true>(qkv, out, lse, BT, N, H, st);
    if (N <= 64) return launch<4, true>(qkv, out, lse, BT, N, H, st);
    if (N <= 224) return launch<14, true>(qkv, out, lse, BT, N, H, st);
    return launch<18, true>(qkv, out, lse, BT, N, H, st);
}
