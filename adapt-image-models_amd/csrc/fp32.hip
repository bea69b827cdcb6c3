// Reference-precision (fp32) forward path of the AIM ViT-CLIP block: north_star's "within 1e-5 (fp32)" bar, met ON THE
// GPU against the real reference's fixtures.  gfx950 only.
//
// Everything here computes in fp32 from fp32 operands, like the reference's un-autocast run (vit_clip.py:433-458): the
// GEMMs on v_mfma_f32_16x16x4_f32 (exact f32 products, a k-ordered fmaf chain per output element: 157 TF/s peak, 1/16 of
// the bf16 rate -- this is the verification mode, `ViT_CLIP.set_precision('fp32')`, not the product's fast path), exact
// erf / exp in the activations (libdevice, not the fast approximations of the bf16 epilogues), attention probabilities
// normalised BEFORE the PV product as the reference does (:153-155).
//
//   aim_gemm_f32        q/k/v, out_proj, c_fc / c_proj, Adapter.D_fc1 / D_fc2, conv1-as-GEMM (:93-97,132-138,157,436)
//   aim_attn_fwd_f32    spatial attention per (frame, head) (:139-156)
//   aim_cls_attn_fwd_f32  temporal attention over the T class tokens of a clip (:220-229)
//   aim_tattn_fwd_f32     temporal attention over the T frames of every token (stock AIM, vitclip_aim.py:199-204)
//   aim_lambda_f32      lamda = cw / (cw + ow) from the head-summed logits (:149-151,184-186,272)
//   aim_patchify_f32, aim_embed_ln_f32   (:434-447)
// and the BACKWARD of the same steps (the reference gets it from torch autograd), so that the hand-written backward's
// algebra is held to the real reference's autograd gradients at fp32 noise instead of bf16 noise:
//   aim_gemm_f32(AIM_EPI_DACT)   dgrad x activation derivative (exact erf / exp) from the saved fp32 pre-activation
//   aim_attn_bwd_f32             dq / dk / dv of the spatial attention, probabilities recomputed (softmax backward as
//                                autograd writes it: dS = P o (dP - rowsum(P o dP)), then the 1/sqrt(dh))
//   aim_cls_attn_bwd_f32 / aim_tattn_bwd_f32   the same over sequences of T rows, ACCUMULATED into their rows of d(qkv)
//   aim_wgrad_f32                adapter weight / bias gradients, fixed summation order (chunk partials + finish)
#include "aim_common.h"
#include "aim_kernels_internal.h"

namespace {

// ---- GEMM: C = A W^T, both K-contiguous fp32 -------------------------------------------------------------------------
// No LDS: for v_mfma_f32_16x16x4_f32 a lane supplies ONE f32 of each operand per instruction (row l & 15, k = l >> 4), and
// any assignment of k to lanes works as long as both operands use the same one.  So a lane loads 16 bytes of "its" row at
// k0 + 4 (l >> 4) and the four elements feed four consecutive MFMAs (K = 16 per step): every global load is a 64-byte
// segment per row, reuse across the wave's 4 x 4 MFMA tiles is in registers, reuse across waves in L1 / L2.
// Workgroup 128 x 128 (4 waves as 2 x 2), wave 64 x 64.  The weight fragment goes FIRST, so a lane ends with 4 consecutive
// output columns of one row (16-byte epilogue accesses).
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int z = blockIdx.z;
    const int m0 = blockIdx.y * 128 + (wave >> 1) * 64, n0 = blockIdx.x * 128 + (wave & 1) * 64;
    const float* A = (const float*)g.A + (long long)z * g.strideA;
    const float* W = (const float*)g.W + (long long)z * g.strideW;
    const float* ap[4];
    const float* wp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = min(m0 + i * 16 + r, g.M - 1), rw = min(n0 + i * 16 + r, g.N - 1);      // (clamped rows are discarded at the store)
        ap[i] = A + (long long)ra * g.lda + 4 * q;
        wp[i] = W + (long long)rw * g.ldw + 4 * q;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 av[4], wv[4], an[4], wn[4];
    const bool in0 = 4 * q < g.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        av[i] = in0 ? *(const f32x4*)ap[i] : zero;
        wv[i] = in0 ? *(const f32x4*)wp[i] : zero;
    }
    for (int k0 = 0; k0 < g.K; k0 += 16) {
        const bool inn = k0 + 16 + 4 * q < g.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            an[i] = inn ? *(const f32x4*)(ap[i] + k0 + 16) : zero;
            wn[i] = inn ? *(const f32x4*)(wp[i] + k0 + 16) : zero;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j][e], av[i][e], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            av[i] = an[i];
            wv[i] = wn[i];
        }
    }
    // epilogue: lane holds out[row = m0 + 16 i + r][n0 + 16 j + 4 q + (0..3)]
    float* out = (float*)g.out + (long long)z * g.M * g.ldo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + i * 16 + r;
        if (row >= g.M) continue;
        int frame = 0, tok = 0;
        float rs = 1.f, btf = 1.f;
        if (g.ntok > 0) {
            frame = row / g.ntok;
            tok = row - frame * g.ntok;
            if (g.af) rs *= g.af[frame];
            if (g.at) rs *= g.at[tok];
            if (g.bt) btf = g.bt[tok];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c0 = n0 + j * 16 + q * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = c0 + e;
                if (c >= g.N) continue;
                const float b = g.bias ? g.bias[c] : 0.f;
                float v = acc[i][j][e];
                if constexpr (EPI == EPI_BF16) {                  // linear: out = rs * (acc + bias)
                    v = rs * (v + b);
                } else if constexpr (EPI == EPI_ACT) {            // out = [rs *] act(acc + bias), column split as aim_gemm_bf16
                    const float pre = v + b;
                    const bool second = g.n_split > 0 && c >= g.n_split;
                    const int act = second ? g.act2 : g.act;
                    const float y = act == ACT_QGELU ? pre / (1.0f + expf(-1.702f * pre))
                                                     : 0.5f * pre * (1.0f + erff(pre * 0.70710678118654752f));
                    v = (g.n_split > 0 && !second) ? y : rs * y;
                    if (g.out2) ((float*)g.out2)[(long long)row * g.ldo2 + c] = pre;      // saved for AIM_EPI_DACT (f32)
                } else if constexpr (EPI == EPI_DACT) {           // out = [rs *] acc * act'(pre), same column split
                    const float pre = ((const float*)g.aux)[(long long)row * g.ldaux + c];
                    const bool second = g.n_split > 0 && c >= g.n_split;
                    const int act = second ? g.act2 : g.act;
                    float d;
                    if (act == ACT_QGELU) {                       // d/dx x sigmoid(1.702 x)
                        const float sg = 1.0f / (1.0f + expf(-1.702f * pre));
                        d = sg * (1.0f + 1.702f * pre * (1.0f - sg));
                    } else {                                      // d/dx x Phi(x) = Phi(x) + x phi(x)
                        d = 0.5f * (1.0f + erff(pre * 0.70710678118654752f)) + pre * 0.3989422804014327f * expf(-0.5f * pre * pre);
                    }
                    v = v * d * ((g.n_split > 0 && !second) ? 1.0f : rs);
                } else {                                          // EPI_F32: resid + rs * (acc + bias) + bt[tok] * vec[frame][c]
                    float o = g.rs_bias_only ? v + rs * b : rs * (v + b);
                    if (g.resid) o += g.resid[(long long)row * g.ldr + c];
                    if (g.vec) o += btf * g.vec[(long long)frame * g.ldv + c];
                    v = o;
                }
                out[(long long)row * g.ldo + c] = v;
            }
        }
    }
}

// ---- spatial attention, one workgroup (8 waves) per (frame, head) ------------------------------------------------------
// K [N][65] and V [N][64] of the head in LDS (fp32); a wave takes queries w, w + 8, ...: lane j scores key j (+64, +128, ...)
// with q broadcast from registers (v_readlane), softmax by wave reductions, probabilities normalised, then lane d
// accumulates out[d] = sum_j p_j V[j][d] with p_j broadcast by v_readlane.
constexpr int AF_MAXG = 5;      // key groups of 64: N <= 320
__global__ __launch_bounds__(512) void attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int H) {
    extern __shared__ float sm[];
    float* sK = sm;                       // [N][65]
    float* sV = sm + (size_t)N * 65;      // [N][64]
    const int D = H * 64, ld = 3 * D;
    const int bt = blockIdx.x / H, h = blockIdx.x - bt * H;
    const float* base = qkv + (long long)bt * N * ld + h * 64;
    for (int idx = threadIdx.x; idx < N * 16; idx += 512) {
        const int j = idx >> 4, c = (idx & 15) * 4;
        const f32x4 kv = *(const f32x4*)(base + (long long)j * ld + D + c);
        const f32x4 vv = *(const f32x4*)(base + (long long)j * ld + 2 * D + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) sK[j * 65 + c + e] = kv[e];
        *(f32x4*)(sV + j * 64 + c) = vv;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ng = (N + 63) >> 6;
    for (int i = wave; i < N; i += 8) {
        const float qd = base[(long long)i * ld + lane];
        float s[AF_MAXG];
#pragma unroll
        for (int gI = 0; gI < AF_MAXG; ++gI) s[gI] = 0.f;
#pragma unroll
        for (int d = 0; d < 64; ++d) {
            const float qv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qd), d));
#pragma unroll
            for (int gI = 0; gI < AF_MAXG; ++gI) {
                const int j = gI * 64 + lane;
                if (gI < ng && j < N) s[gI] = fmaf(qv, sK[j * 65 + d], s[gI]);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int gI = 0; gI < AF_MAXG; ++gI) {
            s[gI] = (gI < ng && gI * 64 + lane < N) ? s[gI] * 0.125f : -INFINITY;      // aff = q k^T / sqrt(dh) (:147)
            mx = fmaxf(mx, s[gI]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int gI = 0; gI < AF_MAXG; ++gI) {
            s[gI] = (gI < ng && gI * 64 + lane < N) ? expf(s[gI] - mx) : 0.f;
            sum += s[gI];
        }
        sum = wave_sum(sum);
#pragma unroll
        for (int gI = 0; gI < AF_MAXG; ++gI) s[gI] = s[gI] / sum;                        // softmax (:153), then aff @ v (:155)
        float o = 0.f;
#pragma unroll
        for (int gI = 0; gI < AF_MAXG; ++gI) {
            if (gI >= ng) break;
            const int jn = min(64, N - gI * 64);
            for (int jj = 0; jj < jn; ++jj) {
                const float pj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s[gI]), jj));
                o = fmaf(pj, sV[(gI * 64 + jj) * 64 + lane], o);
            }
        }
        out[((long long)bt * N + i) * D + h * 64 + lane] = o;
    }
}

// ---- attention over sequences of T rows (the T class tokens of a clip, :220-229; the T frames of EVERY token in the stock
// AIM block, vitclip_aim.py:199-204): one wave per (clip, sequence, head), lane = head dimension.  Sequence `n` of clip `b`
// starts at b * clip + n * seq floats of qkv and its rows are `row` floats apart (out: oclip / oseq / orow).
struct SeqLayout { long long clip, seq, row, oclip, oseq, orow; int nper; };
__global__ __launch_bounds__(64) void seq_attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, SeqLayout L, int T,
                                                          int H) {
    const int D = H * 64, lane = threadIdx.x;
    const int h = blockIdx.x % H, sq = blockIdx.x / H;
    const int b = sq / L.nper, n = sq - b * L.nper;
    const float* base = qkv + b * L.clip + n * L.seq + h * 64 + lane;
    float* ob = out + b * L.oclip + n * L.oseq + h * 64 + lane;
    for (int tq = 0; tq < T; ++tq) {
        const float qv = base[tq * L.row];
        float s[32];
        float mx = -INFINITY;
        for (int tk = 0; tk < T; ++tk) {
            s[tk] = wave_sum(qv * base[tk * L.row + D]) * 0.125f;
            mx = fmaxf(mx, s[tk]);
        }
        float sum = 0.f;
        for (int tk = 0; tk < T; ++tk) {
            s[tk] = expf(s[tk] - mx);
            sum += s[tk];
        }
        float o = 0.f;
        for (int tk = 0; tk < T; ++tk) o = fmaf(s[tk] / sum, base[tk * L.row + 2 * D], o);
        ob[tq * L.orow] = o;
    }
}

// ---- lamda: one workgroup per frame -----------------------------------------------------------------------------------
// scores[frame][i][j] = q_i . k_j over the FULL width (= the sum over heads of the per-head logits times sqrt(dh)), raw;
// ss_i = q_i . kx (the cross-attention's single key).  ow = sum_ij exp(scale s_ij), cw = sum_i exp(scale ss_i), both under
// one shared max shift (the ratio is exact; the reference's un-shifted form overflows where this does not).
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}
__global__ __launch_bounds__(256) void lambda_f32_kernel(const float* __restrict__ scores, int lds_, const float* __restrict__ qkv,
                                                         const float* __restrict__ kx, int ldkx, float* __restrict__ lam,
                                                         float* __restrict__ oml, int N, int D, float scale) {
    extern __shared__ float ssm[];          // [N] cross scores + [4] reduction slots
    float* red = ssm + N;
    const int f = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* kxf = kx + (long long)f * ldkx;
    for (int i = wave; i < N; i += 4) {
        const float* qi = qkv + ((long long)f * N + i) * 3 * D;
        float a = 0.f;
        for (int d = lane; d < D; d += 64) a = fmaf(qi[d], kxf[d], a);
        a = wave_sum(a);
        if (lane == 0) ssm[i] = a * scale;
    }
    __syncthreads();
    const float* S = scores + (long long)f * N * lds_;
    float mx = -INFINITY;
    for (int idx = threadIdx.x; idx < N * N; idx += 256) mx = fmaxf(mx, S[(idx / N) * lds_ + idx % N] * scale);
    for (int i = threadIdx.x; i < N; i += 256) mx = fmaxf(mx, ssm[i]);
    mx = block_reduce(mx, red, true);
    float ow = 0.f, cw = 0.f;
    for (int idx = threadIdx.x; idx < N * N; idx += 256) ow += expf(S[(idx / N) * lds_ + idx % N] * scale - mx);
    for (int i = threadIdx.x; i < N; i += 256) cw += expf(ssm[i] - mx);
    ow = block_reduce(ow, red, false);
    cw = block_reduce(cw, red, false);
    if (threadIdx.x == 0) {
        const float l = cw / (cw + ow);
        lam[f] = l;
        if (oml) oml[f] = 1.0f - l;
    }
}

// ---- patch matrix in fp32 (conv1 as a GEMM, :436); optional fused uint8 GPUNormalize like aim_patchify -----------------
template <typename TIN>
__global__ __launch_bounds__(256) void patchify_f32_kernel(const TIN* __restrict__ img, const float* __restrict__ mean3,
                                                           const float* __restrict__ std3, float* __restrict__ A, int B, int T,
                                                           int H, int W, int p, int Kp) {
    const int G = W / p, Gy = H / p, K = 3 * p * p;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * T * Gy * G * Kp) return;
    const long long row = idx / Kp;
    const int k = (int)(idx - row * Kp);
    float v = 0.f;
    if (k < K) {
        const int gx = (int)(row % G), gy = (int)((row / G) % Gy);
        const long long bt = row / ((long long)G * Gy);
        const int t = (int)(bt % T);
        const long long b = bt / T;
        const int c = k / (p * p), rem = k - c * p * p, py = rem / p, px = rem - py * p;
        v = (float)img[(((b * 3 + c) * T + t) * H + (gy * p + py)) * (long long)W + gx * p + px];
        if (mean3) v = (v - mean3[c]) / std3[c];
    }
    A[idx] = v;
}

// ---- class token + positional + temporal embedding + ln_pre (:439-447), tokens in fp32: a wave per row -----------------
__global__ __launch_bounds__(256) void embed_ln_f32_kernel(const float* __restrict__ tok, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, const float* __restrict__ tmp,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ x, float* __restrict__ pre,
                                                           float* __restrict__ mean, float* __restrict__ rstd, int B, int T,
                                                           int N, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)B * T * N) return;
    const long long bt = row / N;
    const int n = (int)(row - bt * N), t = (int)(bt % T);
    const float* src = n == 0 ? cls : tok + (bt * (N - 1) + (n - 1)) * (long long)D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += src[d] + pos[(long long)n * D + d] + tmp[(long long)t * D + d];
    const float mu = wave_sum(s) / (float)D;
    float qv = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float v = src[d] + pos[(long long)n * D + d] + tmp[(long long)t * D + d] - mu;
        qv += v * v;
    }
    const float rs = 1.0f / sqrtf(wave_sum(qv) / (float)D + eps);
    for (int d = lane; d < D; d += 64) {
        const float v = src[d] + pos[(long long)n * D + d] + tmp[(long long)t * D + d];
        x[row * D + d] = (v - mu) * rs * gamma[d] + beta[d];
        if (pre) pre[row * D + d] = v;            // what ln_pre's backward needs (temporal_embedding is trainable, :344)
    }
    if (pre && lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
}

// ---- spatial attention backward, two kernels, one workgroup (8 waves) per (frame, head) each ---------------------------
// dq: K [N][65] and V [N][65] in LDS; a wave takes queries w, w + 8, ...; lane j holds key j (+64, ...): s_j = q . k_j and
// dP_j = dO . v_j with q / dO broadcast by v_readlane, P as the forward computes it, dS = P o (dP - sum_j P_j dP_j) / 8
// (torch's softmax backward, then the division of :147), dq[d] = sum_j dS_j K[j][d] on lane d.  Leaves the row's
// log-sum-exp and sum_j P_j dP_j in `stats` for the dk / dv kernel.
__global__ __launch_bounds__(512) void attn_bwd_dq_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                              float* __restrict__ dqkv, float* __restrict__ stats, int N, int H) {
    extern __shared__ float sm[];
    float* sK = sm;                       // [N][65]
    float* sV = sm + (size_t)N * 65;      // [N][65]
    const int D = H * 64, ld = 3 * D;
    const int bt = blockIdx.x / H, h = blockIdx.x - bt * H;
    const float* base = qkv + (long long)bt * N * ld + h * 64;
    for (int idx = threadIdx.x; idx < N * 16; idx += 512) {
        const int j = idx >> 4, c = (idx & 15) * 4;
        const f32x4 kv = *(const f32x4*)(base + (long long)j * ld + D + c);
        const f32x4 vv = *(const f32x4*)(base + (long long)j * ld + 2 * D + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sK[j * 65 + c + e] = kv[e];
            sV[j * 65 + c + e] = vv[e];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ng = (N + 63) >> 6;
    for (int i = wave; i < N; i += 8) {
        const float qd = base[(long long)i * ld + lane];
        const float dod = dout[((long long)bt * N + i) * D + h * 64 + lane];
        float s[AF_MAXG], dp[AF_MAXG];
#pragma unroll
        for (int gI = 0; gI < AF_MAXG; ++gI) s[gI] = dp[gI] = 0.f;
#pragma unroll
        for (int d = 0; d < 64; ++d) {
            const float qv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qd), d));
            const float dv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dod), d));
#pragma unroll
            for (int gI = 0; gI < AF_MAXG; ++gI) {
                const int j = gI * 64 + lane;
                if (gI < ng && j < N) {
                    s[gI] = fmaf(qv, sK[j * 65 + d], s[gI]);
                    dp[gI] = fmaf(dv, sV[j * 65 + d], dp[gI]);
                }
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int gI = 0; gI < AF_MAXG; ++gI) {
            s[gI] = (gI < ng && gI * 64 + lane < N) ? s[gI] * 0.125f : -INFINITY;
            mx = fmaxf(mx, s[gI]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int gI = 0; gI < AF_MAXG; ++gI) {
            s[gI] = (gI < ng && gI * 64 + lane < N) ? expf(s[gI] - mx) : 0.f;
            sum += s[gI];
        }
        sum = wave_sum(sum);
        float del = 0.f;
#pragma unroll
        for (int gI = 0; gI < AF_MAXG; ++gI) {
            s[gI] = s[gI] / sum;
            del = fmaf(s[gI], dp[gI], del);
        }
        del = wave_sum(del);
#pragma unroll
        for (int gI = 0; gI < AF_MAXG; ++gI) s[gI] = s[gI] * (dp[gI] - del) * 0.125f;       // dS / sqrt(dh)
        if (lane == 0) {
            float* st = stats + (((long long)bt * H + h) * N + i) * 2;
            st[0] = mx + logf(sum);
            st[1] = del;
        }
        float o = 0.f;
#pragma unroll
        for (int gI = 0; gI < AF_MAXG; ++gI) {
            if (gI >= ng) break;
            const int jn = min(64, N - gI * 64);
            for (int jj = 0; jj < jn; ++jj) {
                const float dsj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s[gI]), jj));
                o = fmaf(dsj, sK[(gI * 64 + jj) * 65 + lane], o);
            }
        }
        dqkv[((long long)bt * N + i) * ld + h * 64 + lane] = o;
    }
}

// dk / dv: Q [N][65] and dO [N][65] in LDS with the rows' log-sum-exp and sum_j P_j dP_j; a wave takes keys w, w + 8, ...;
// lane i holds query i (+64, ...): P_i = exp(q_i . k / 8 - L_i), dS_i as above; dv[d] = sum_i P_i dO[i][d] and
// dk[d] = sum_i dS_i Q[i][d] on lane d.
__global__ __launch_bounds__(512) void attn_bwd_dkv_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                               float* __restrict__ dqkv, const float* __restrict__ stats, int N, int H) {
    extern __shared__ float sm[];
    float* sQ = sm;                          // [N][65]
    float* sO = sm + (size_t)N * 65;         // [N][65]  dO
    float* sL = sm + (size_t)N * 130;        // [N]
    float* sD = sL + N;                      // [N]
    const int D = H * 64, ld = 3 * D;
    const int bt = blockIdx.x / H, h = blockIdx.x - bt * H;
    const float* base = qkv + (long long)bt * N * ld + h * 64;
    const float* dob = dout + (long long)bt * N * D + h * 64;
    for (int idx = threadIdx.x; idx < N * 16; idx += 512) {
        const int i = idx >> 4, c = (idx & 15) * 4;
        const f32x4 qv = *(const f32x4*)(base + (long long)i * ld + c);
        const f32x4 ov = *(const f32x4*)(dob + (long long)i * D + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sQ[i * 65 + c + e] = qv[e];
            sO[i * 65 + c + e] = ov[e];
        }
    }
    for (int i = threadIdx.x; i < N; i += 512) {
        const float* st = stats + (((long long)bt * H + h) * N + i) * 2;
        sL[i] = st[0];
        sD[i] = st[1];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ng = (N + 63) >> 6;
    for (int j = wave; j < N; j += 8) {
        const float kd = base[(long long)j * ld + D + lane];
        const float vd = base[(long long)j * ld + 2 * D + lane];
        float s[AF_MAXG], dp[AF_MAXG];
#pragma unroll
        for (int gI = 0; gI < AF_MAXG; ++gI) s[gI] = dp[gI] = 0.f;
#pragma unroll
        for (int d = 0; d < 64; ++d) {
            const float kv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, kd), d));
            const float vv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vd), d));
#pragma unroll
            for (int gI = 0; gI < AF_MAXG; ++gI) {
                const int i = gI * 64 + lane;
                if (gI < ng && i < N) {
                    s[gI] = fmaf(sQ[i * 65 + d], kv, s[gI]);
                    dp[gI] = fmaf(sO[i * 65 + d], vv, dp[gI]);
                }
            }
        }
#pragma unroll
        for (int gI = 0; gI < AF_MAXG; ++gI) {
            const int i = gI * 64 + lane;
            const bool in = gI < ng && i < N;
            const float p = in ? expf(s[gI] * 0.125f - sL[i]) : 0.f;
            s[gI] = p;                                                  // P_i
            dp[gI] = in ? p * (dp[gI] - sD[i]) * 0.125f : 0.f;          // dS_i / sqrt(dh)
        }
        float dv = 0.f, dk = 0.f;
#pragma unroll
        for (int gI = 0; gI < AF_MAXG; ++gI) {
            if (gI >= ng) break;
            const int in_ = min(64, N - gI * 64);
            for (int ii = 0; ii < in_; ++ii) {
                const float pi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s[gI]), ii));
                const float di = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dp[gI]), ii));
                dv = fmaf(pi, sO[(gI * 64 + ii) * 65 + lane], dv);
                dk = fmaf(di, sQ[(gI * 64 + ii) * 65 + lane], dk);
            }
        }
        dqkv[((long long)bt * N + j) * ld + D + h * 64 + lane] = dk;
        dqkv[((long long)bt * N + j) * ld + 2 * D + h * 64 + lane] = dv;
    }
}

// ---- the same, backward: one wave per (clip, sequence, head).  ACCUMULATES into the sequence's rows of d(qkv) (for the class
// tokens those rows already hold the spatial attention's share); the wave is the only writer of its 64 columns of those rows.
__global__ __launch_bounds__(64) void seq_attn_bwd_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                              float* __restrict__ dqkv, SeqLayout L, int T, int H) {
    const int D = H * 64, lane = threadIdx.x;
    const int h = blockIdx.x % H, sq = blockIdx.x / H;
    const int b = sq / L.nper, n = sq - b * L.nper;
    const long long off = b * L.clip + n * L.seq + h * 64 + lane;
    const float* base = qkv + off;
    float* dbase = dqkv + off;
    const float* dob = dout + b * L.oclip + n * L.oseq + h * 64 + lane;
    for (int tq = 0; tq < T; ++tq) {
        const float qv = base[tq * L.row];
        const float dov = dob[tq * L.orow];
        float s[32], dp[32];
        float mx = -INFINITY;
        for (int tk = 0; tk < T; ++tk) {
            s[tk] = wave_sum(qv * base[tk * L.row + D]) * 0.125f;
            dp[tk] = wave_sum(dov * base[tk * L.row + 2 * D]);
            mx = fmaxf(mx, s[tk]);
        }
        float sum = 0.f;
        for (int tk = 0; tk < T; ++tk) {
            s[tk] = expf(s[tk] - mx);
            sum += s[tk];
        }
        float del = 0.f;
        for (int tk = 0; tk < T; ++tk) {
            s[tk] = s[tk] / sum;
            del = fmaf(s[tk], dp[tk], del);
        }
        float dq = 0.f;
        for (int tk = 0; tk < T; ++tk) {
            const float ds = s[tk] * (dp[tk] - del) * 0.125f;
            dq = fmaf(ds, base[tk * L.row + D], dq);
            dbase[tk * L.row + D] += ds * qv;
            dbase[tk * L.row + 2 * D] += s[tk] * dov;
        }
        dbase[tq * L.row] += dq;
    }
}

// ---- adapter weight gradients: dW[n][k] += sum_m G[m][n] A[m][k], db[n] += sum_m at[m % ntok] G[m][n] --------------------
// (autograd's addmm backward of Adapter.D_fc1 / D_fc2, :57-58,62-64.)  Plain fp32 FMAs, 64 x 64 output tile per workgroup, a
// 4 x 4 register tile per thread, 16 reduction rows per LDS stage; the M range is cut into chunks whose partial tiles are
// summed in chunk order by reduce_slabs (no atomics: the same bits every run).
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ A, int lda,
                                                        float* __restrict__ partial, int M, int Nw, int Kw, int chunk) {
    __shared__ float sG[16][64], sA[16][64];
    const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 64, z = blockIdx.z;
    const int mbeg = z * chunk, mend = min(M, mbeg + chunk);
    const int tid = threadIdx.x, tn = tid >> 4, tk = tid & 15;
    const int lr = tid >> 4, lc = (tid & 15) * 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int m0 = mbeg; m0 < mend; m0 += 16) {
        const int m = m0 + lr;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sG[lr][lc + e] = (m < mend && n0 + lc + e < Nw) ? G[(long long)m * ldg + n0 + lc + e] : 0.f;
            sA[lr][lc + e] = (m < mend && k0 + lc + e < Kw) ? A[(long long)m * lda + k0 + lc + e] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < 16; ++mm) {
            float gv[4], av[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gv[e] = sG[mm][tn * 4 + e];
                av[e] = sA[mm][tk * 4 + e];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(gv[i], av[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* P = partial + (long long)z * Nw * Kw;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tn * 4 + i, k = k0 + tk * 4 + j;
            if (n < Nw && k < Kw) P[(long long)n * Kw + k] = acc[i][j];
        }
}

__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ at, int ntok,
                                                         float* __restrict__ partial, int M, int C, int chunk) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int mbeg = blockIdx.y * chunk, mend = min(M, mbeg + chunk);
    float acc = 0.f;
    for (int m = mbeg; m < mend; ++m) acc = fmaf(at ? at[m % ntok] : 1.0f, G[(long long)m * ldg + c], acc);
    partial[(long long)blockIdx.y * C + c] = acc;
}

__global__ __launch_bounds__(256) void reduce_slabs_f32_kernel(const float* __restrict__ partial, float* __restrict__ out, int slabs,
                                                               long long numel) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= numel) return;
    float acc = 0.f;
    for (int s = 0; s < slabs; ++s) acc += partial[(long long)s * numel + i];
    out[i] += acc;
}

}  // namespace

extern "C" int aim_gemm_f32(const aim_gemm_args* args, int epilogue, int batch, void* stream) {
    AIM_CHECK_ARG(args != nullptr && batch >= 1, "gemm_f32: null args / bad batch");
    const GemmArgs& g = *args;
    AIM_CHECK_ARG(g.A && g.W && g.out && g.M > 0 && g.N > 0 && g.K > 0, "gemm_f32: null operand or empty problem");
    AIM_CHECK_ARG((g.K % 4) == 0 && (g.lda % 4) == 0 && (g.ldw % 4) == 0 && ((g.strideA | g.strideW) % 4) == 0,
                  "gemm_f32: K, lda, ldw and the batch strides must be multiples of 4 (K=%d lda=%d ldw=%d)", g.K, g.lda, g.ldw);
    AIM_CHECK_ARG((((uintptr_t)g.A | (uintptr_t)g.W) & 15) == 0, "gemm_f32: operands must be 16-byte aligned");
    AIM_CHECK_ARG(g.ldo >= g.N, "gemm_f32: ldo < N");
    if (g.af || g.at || g.vec || g.bt) AIM_CHECK_ARG(g.ntok > 0, "gemm_f32: ntok required with row factors");
    AIM_CHECK_ARG(batch == 1 || epilogue == EPI_BF16, "gemm_f32: batched problems take the linear epilogue only");
    AIM_CHECK_ARG(!g.out2 || (epilogue == EPI_ACT && g.ldo2 >= g.N), "gemm_f32: out2 (the fp32 pre-activation) goes with AIM_EPI_ACT, ldo2 >= N");
    const dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, batch), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case EPI_BF16: hipLaunchKernelGGL(gemm_f32_kernel<EPI_BF16>, grid, block, 0, st, g); break;
        case EPI_ACT: hipLaunchKernelGGL(gemm_f32_kernel<EPI_ACT>, grid, block, 0, st, g); break;
        case EPI_F32: hipLaunchKernelGGL(gemm_f32_kernel<EPI_F32>, grid, block, 0, st, g); break;
        case EPI_DACT:
            AIM_CHECK_ARG(g.aux && g.ldaux >= g.N, "gemm_f32: AIM_EPI_DACT needs the saved fp32 pre-activation (aux, ldaux >= N)");
            hipLaunchKernelGGL(gemm_f32_kernel<EPI_DACT>, grid, block, 0, st, g);
            break;
        default: aim_set_error("gemm_f32: unsupported epilogue %d (BF16 = linear, ACT, DACT, F32)", epilogue); return 1;
    }
    AIM_CHECK_LAUNCH("aim_gemm_f32");
    return 0;
}

extern "C" int aim_attn_fwd_f32(const float* qkv, float* out, int BT, int N, int H, void* stream) {
    AIM_CHECK_ARG(qkv && out && BT > 0 && H > 0 && N > 0 && N <= 64 * AF_MAXG && (size_t)N * 129 * 4 <= 160 * 1024,
                  "attn_fwd_f32: unsupported shape BT=%d N=%d H=%d (N <= 317)", BT, N, H);
    const size_t lds = (size_t)N * 129 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_f32_kernel, dim3(BT * H), dim3(512), lds, (hipStream_t)stream, qkv, out, N, H);
    AIM_CHECK_LAUNCH("aim_attn_fwd_f32");
    return 0;
}

extern "C" int aim_cls_attn_fwd_f32(const float* qkv, int64_t row_stride, float* out_cls, int B, int T, int H, void* stream) {
    AIM_CHECK_ARG(qkv && out_cls && B > 0 && T > 0 && T <= 32 && H > 0, "cls_attn_fwd_f32: unsupported shape B=%d T=%d H=%d (T <= 32)", B, T, H);
    const long long D = (long long)H * 64;
    const SeqLayout L{T * (long long)row_stride, 0, (long long)row_stride, T * D, 0, D, 1};
    hipLaunchKernelGGL(seq_attn_f32_kernel, dim3(B * H), dim3(64), 0, (hipStream_t)stream, qkv, out_cls, L, T, H);
    AIM_CHECK_LAUNCH("aim_cls_attn_fwd_f32");
    return 0;
}

static SeqLayout tattn_layout(int T, int N, int H) {
    const long long D = (long long)H * 64;
    return SeqLayout{T * (long long)N * 3 * D, 3 * D, N * 3 * D, T * (long long)N * D, D, N * D, N};
}

extern "C" int aim_tattn_fwd_f32(const float* qkv, float* out, int B, int T, int N, int H, void* stream) {
    AIM_CHECK_ARG(qkv && out && B > 0 && T > 0 && T <= 32 && N > 0 && H > 0, "tattn_fwd_f32: unsupported shape B=%d T=%d N=%d H=%d (T <= 32)",
                  B, T, N, H);
    hipLaunchKernelGGL(seq_attn_f32_kernel, dim3(B * N * H), dim3(64), 0, (hipStream_t)stream, qkv, out, tattn_layout(T, N, H), T, H);
    AIM_CHECK_LAUNCH("aim_tattn_fwd_f32");
    return 0;
}

extern "C" int aim_tattn_bwd_f32(const float* qkv, const float* dout, float* dqkv, int B, int T, int N, int H, void* stream) {
    AIM_CHECK_ARG(qkv && dout && dqkv && B > 0 && T > 0 && T <= 32 && N > 0 && H > 0,
                  "tattn_bwd_f32: unsupported shape B=%d T=%d N=%d H=%d (T <= 32)", B, T, N, H);
    hipLaunchKernelGGL(seq_attn_bwd_f32_kernel, dim3(B * N * H), dim3(64), 0, (hipStream_t)stream, qkv, dout, dqkv, tattn_layout(T, N, H),
                       T, H);
    AIM_CHECK_LAUNCH("aim_tattn_bwd_f32");
    return 0;
}

extern "C" int aim_lambda_f32(const float* scores, int lds_, const float* qkv, const float* kx, int ldkx, float* lam,
                              float* one_minus_lam, int BT, int N, int D, float scale, void* stream) {
    AIM_CHECK_ARG(scores && qkv && kx && lam && BT > 0 && N > 0 && D > 0 && lds_ >= N, "lambda_f32: bad arguments");
    hipLaunchKernelGGL(lambda_f32_kernel, dim3(BT), dim3(256), (size_t)(N + 4) * 4, (hipStream_t)stream, scores, lds_, qkv, kx, ldkx,
                       lam, one_minus_lam, N, D, scale);
    AIM_CHECK_LAUNCH("aim_lambda_f32");
    return 0;
}

extern "C" int aim_patchify_f32(const void* imgs, int in_dtype, const float* mean3, const float* std3, float* A, int B, int T,
                                int H, int W, int p, int Kp, void* stream) {
    AIM_CHECK_ARG(B > 0 && T > 0 && p > 0 && H % p == 0 && W % p == 0, "patchify_f32: bad shape H=%d W=%d p=%d", H, W, p);
    AIM_CHECK_ARG(Kp >= 3 * p * p && (Kp % 4) == 0, "patchify_f32: Kp=%d must be >= 3*p*p and a multiple of 4", Kp);
    AIM_CHECK_ARG(imgs && A && ((!mean3) == (!std3)), "patchify_f32: null pointer");
    const long long total = (long long)B * T * (H / p) * (W / p) * Kp;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (in_dtype == 0)
        hipLaunchKernelGGL(patchify_f32_kernel<float>, grid, block, 0, st, (const float*)imgs, mean3, std3, A, B, T, H, W, p, Kp);
    else if (in_dtype == 1)
        hipLaunchKernelGGL(patchify_f32_kernel<uint8_t>, grid, block, 0, st, (const uint8_t*)imgs, mean3, std3, A, B, T, H, W, p, Kp);
    else {
        aim_set_error("patchify_f32: in_dtype must be 0 (f32) or 1 (uint8), got %d", in_dtype);
        return 1;
    }
    AIM_CHECK_LAUNCH("aim_patchify_f32");
    return 0;
}

extern "C" int aim_embed_ln_f32(const float* tok, const float* cls, const float* pos, const float* temporal, const float* gamma,
                                const float* beta, float* x, float* pre, float* mean, float* rstd, int B, int T, int N, int D,
                                float eps, void* stream) {
    AIM_CHECK_ARG(tok && cls && pos && temporal && gamma && beta && x && B > 0 && T > 0 && N > 1 && D > 0, "embed_ln_f32: bad arguments");
    AIM_CHECK_ARG((!pre) == (!mean) && (!pre) == (!rstd), "embed_ln_f32: pre, mean and rstd go together");
    const long long rows = (long long)B * T * N;
    hipLaunchKernelGGL(embed_ln_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, tok, cls, pos,
                       temporal, gamma, beta, x, pre, mean, rstd, B, T, N, D, eps);
    AIM_CHECK_LAUNCH("aim_embed_ln_f32");
    return 0;
}

extern "C" int64_t aim_attn_bwd_f32_workspace_bytes(int BT, int N, int H) { return (int64_t)BT * H * N * 2 * 4; }

extern "C" int aim_attn_bwd_f32(const float* qkv, const float* dout, float* dqkv, int BT, int N, int H, float* workspace,
                                int64_t workspace_bytes, void* stream) {
    AIM_CHECK_ARG(qkv && dout && dqkv && BT > 0 && H > 0 && N > 0 && N <= 64 * AF_MAXG && (size_t)N * 132 * 4 <= 160 * 1024,
                  "attn_bwd_f32: unsupported shape BT=%d N=%d H=%d (N <= 310)", BT, N, H);
    AIM_CHECK_ARG(workspace && workspace_bytes >= aim_attn_bwd_f32_workspace_bytes(BT, N, H), "attn_bwd_f32: workspace too small");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_bwd_dq_f32_kernel, dim3(BT * H), dim3(512), (size_t)N * 130 * 4, (hipStream_t)stream, qkv, dout, dqkv,
                       workspace, N, H);
    AIM_CHECK_LAUNCH("aim_attn_bwd_f32(dq)");
    hipLaunchKernelGGL(attn_bwd_dkv_f32_kernel, dim3(BT * H), dim3(512), (size_t)N * 132 * 4, (hipStream_t)stream, qkv, dout, dqkv,
                       (const float*)workspace, N, H);
    AIM_CHECK_LAUNCH("aim_attn_bwd_f32(dkv)");
    return 0;
}

extern "C" int aim_cls_attn_bwd_f32(const float* qkv, int64_t row_stride, const float* dout_cls, float* dqkv, int B, int T, int H,
                                    void* stream) {
    AIM_CHECK_ARG(qkv && dout_cls && dqkv && B > 0 && T > 0 && T <= 32 && H > 0,
                  "cls_attn_bwd_f32: unsupported shape B=%d T=%d H=%d (T <= 32)", B, T, H);
    const long long D = (long long)H * 64;
    const SeqLayout L{T * (long long)row_stride, 0, (long long)row_stride, T * D, 0, D, 1};
    hipLaunchKernelGGL(seq_attn_bwd_f32_kernel, dim3(B * H), dim3(64), 0, (hipStream_t)stream, qkv, dout_cls, dqkv, L, T, H);
    AIM_CHECK_LAUNCH("aim_cls_attn_bwd_f32");
    return 0;
}

static int wgrad_f32_chunks(int M) {
    int chunks = (M + 511) / 512;
    return chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
}

extern "C" int64_t aim_wgrad_f32_workspace_bytes(int M, int Nw, int Kw) {
    return (int64_t)wgrad_f32_chunks(M) * ((int64_t)Nw * Kw + Nw) * 4;
}

extern "C" int aim_wgrad_f32(const float* G, int ldg, const float* A, int lda, float* dW, int M, int Nw, int Kw, float* db,
                             const float* at, int ntok, float* workspace, int64_t workspace_bytes, void* stream) {
    AIM_CHECK_ARG(G && A && dW && M > 0 && Nw > 0 && Kw > 0 && ldg >= Nw && lda >= Kw, "wgrad_f32: bad arguments");
    AIM_CHECK_ARG(!at || ntok > 0, "wgrad_f32: ntok required with the bias row factor");
    AIM_CHECK_ARG(workspace && workspace_bytes >= aim_wgrad_f32_workspace_bytes(M, Nw, Kw), "wgrad_f32: workspace too small");
    const int chunks = wgrad_f32_chunks(M);
    const int chunk = ((M + chunks - 1) / chunks + 15) / 16 * 16;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wgrad_f32_kernel, dim3((Kw + 63) / 64, (Nw + 63) / 64, chunks), dim3(256), 0, st, G, ldg, A, lda, workspace, M,
                       Nw, Kw, chunk);
    AIM_CHECK_LAUNCH("aim_wgrad_f32");
    const long long numel = (long long)Nw * Kw;
    hipLaunchKernelGGL(reduce_slabs_f32_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, st, (const float*)workspace, dW,
                       chunks, numel);
    AIM_CHECK_LAUNCH("aim_wgrad_f32(finish)");
    if (db) {
        float* bp = workspace + (long long)chunks * numel;
        hipLaunchKernelGGL(colsum_f32_kernel, dim3((Nw + 255) / 256, chunks), dim3(256), 0, st, G, ldg, at, ntok, bp, M, Nw, chunk);
        AIM_CHECK_LAUNCH("aim_wgrad_f32(bias)");
        hipLaunchKernelGGL(reduce_slabs_f32_kernel, dim3((Nw + 255) / 256), dim3(256), 0, st, (const float*)bp, db, chunks, (long long)Nw);
        AIM_CHECK_LAUNCH("aim_wgrad_f32(bias finish)");
    }
    return 0;
}
