// Temporal attention over the T class tokens of each clip, and the per-frame lamda statistic.
// gfx950 only.
//
// cls_attn_*: reference vit_clip.py:220-224 -- attention() (:139-156) applied to
//   rearrange(x[:1], 'n (b t) d -> t (b n) d'), i.e. sequence = the T class tokens of a clip,
//   batch = B.  Sequence length is T <= 32, so this is a latency-bound scalar kernel: one wave per
//   (clip, head), lane = head-dim element, everything in LDS.  The q/k/v rows are the class rows
//   (token 0) of the frame-major fused qkv buffer (same in_proj weights, same ln_1 rows).
// lambda: reference vit_clip.py:149-151,184-186,272 -- lamda = cw / (cw + ow) with
//   ow = sum_{i,j} exp(sum_h aff_h[i,j]) = sum exp(q_i . k_j / sqrt(dh)) over the full width D and
//   cw = sum_i exp(q_i . kx / sqrt(dh)).  ow arrives as (max, sum) partials of the EXPSUM GEMM;
//   cw is computed here; one shared max shift keeps the ratio exact without fp32 overflow.
#include "aim_common.h"
#include "aim_kernels_internal.h"

namespace {

constexpr int TMAX = 32;

__global__ __launch_bounds__(64) void cls_attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                          float* __restrict__ probs, int T, int N, int H) {
    __shared__ float sq[TMAX][65], sk[TMAX][65], sv[TMAX][65], sp[TMAX][TMAX + 1];
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * 64, ld = 3 * D, lane = threadIdx.x;
    for (int t = 0; t < T; ++t) {
        const bf16_t* r = qkv + ((long long)(b * T + t) * N) * ld + h * 64 + lane;
        sq[t][lane] = (float)r[0];
        sk[t][lane] = (float)r[D];
        sv[t][lane] = (float)r[2 * D];
    }
    __syncthreads();
    for (int p = lane; p < T * T; p += 64) {
        const int tq = p / T, tk = p - tq * T;
        float acc = 0.f;
#pragma unroll 8
        for (int d = 0; d < 64; ++d) acc += sq[tq][d] * sk[tk][d];
        sp[tq][tk] = acc * 0.125f;
    }
    __syncthreads();
    for (int tq = lane; tq < T; tq += 64) {
        float mx = -INFINITY;
        for (int tk = 0; tk < T; ++tk) mx = fmaxf(mx, sp[tq][tk]);
        float sum = 0.f;
        for (int tk = 0; tk < T; ++tk) {
            const float e = expf(sp[tq][tk] - mx);
            sp[tq][tk] = e;
            sum += e;
        }
        const float inv = 1.0f / sum;
        for (int tk = 0; tk < T; ++tk) {
            const float p = sp[tq][tk] * inv;
            sp[tq][tk] = p;
            probs[(((long long)b * H + h) * T + tq) * T + tk] = p;
        }
    }
    __syncthreads();
    for (int tq = 0; tq < T; ++tq) {
        float acc = 0.f;
        for (int tk = 0; tk < T; ++tk) acc += sp[tq][tk] * sv[tk][lane];
        out[(long long)(b * T + tq) * D + h * 64 + lane] = (bf16_t)acc;
    }
}

__global__ __launch_bounds__(64) void cls_attn_bwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ probs,
                                                          const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv,
                                                          int T, int N, int H, int compact) {
    __shared__ float sq[TMAX][65], sk[TMAX][65], sv[TMAX][65], sdo[TMAX][65], sp[TMAX][TMAX + 1], sds[TMAX][TMAX + 1];
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * 64, ld = 3 * D, lane = threadIdx.x;
    for (int t = 0; t < T; ++t) {
        const bf16_t* r = qkv + ((long long)(b * T + t) * N) * ld + h * 64 + lane;
        sq[t][lane] = (float)r[0];
        sk[t][lane] = (float)r[D];
        sv[t][lane] = (float)r[2 * D];
        sdo[t][lane] = (float)dout[(long long)(b * T + t) * D + h * 64 + lane];
    }
    for (int p = lane; p < T * T; p += 64) {
        const int tq = p / T, tk = p - tq * T;
        sp[tq][tk] = probs[(((long long)b * H + h) * T + tq) * T + tk];
    }
    __syncthreads();
    // dP[tq][tk] = dO[tq] . V[tk]
    for (int p = lane; p < T * T; p += 64) {
        const int tq = p / T, tk = p - tq * T;
        float acc = 0.f;
#pragma unroll 8
        for (int d = 0; d < 64; ++d) acc += sdo[tq][d] * sv[tk][d];
        sds[tq][tk] = acc;
    }
    __syncthreads();
    for (int tq = lane; tq < T; tq += 64) {
        float dot = 0.f;
        for (int tk = 0; tk < T; ++tk) dot += sp[tq][tk] * sds[tq][tk];
        for (int tk = 0; tk < T; ++tk) sds[tq][tk] = sp[tq][tk] * (sds[tq][tk] - dot) * 0.125f;
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int u = 0; u < T; ++u) {
            dq += sds[t][u] * sk[u][lane];
            dk += sds[u][t] * sq[u][lane];
            dv += sp[u][t] * sdo[u][lane];
        }
        if (compact) {      // dqkv is [B*T, 3D]: the class rows' contribution alone
            bf16_t* r = dqkv + (long long)(b * T + t) * ld + h * 64 + lane;
            r[0] = (bf16_t)dq;
            r[D] = (bf16_t)dk;
            r[2 * D] = (bf16_t)dv;
        } else {            // dqkv is the spatial attention's [B*T*N, 3D]: add into its class rows
            bf16_t* r = dqkv + ((long long)(b * T + t) * N) * ld + h * 64 + lane;
            r[0] = (bf16_t)((float)r[0] + dq);
            r[D] = (bf16_t)((float)r[D] + dk);
            r[2 * D] = (bf16_t)((float)r[2 * D] + dv);
        }
    }
}

// ss[bt][i] = scale * q_i . kx[bt] over the full width D: the memory-bound half of lamda's cw statistic (one pass over q).
// Split from lambda_kernel so that it runs as soon as q exists, beside the ow GEMM, instead of after it.
__global__ __launch_bounds__(256) void qk_cross_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ kx, int ldkx,
                                                       float* __restrict__ ssg, int N, int D, float scale) {
    const int bt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = 3 * D;
    const bf16_t* kr = kx + (long long)bt * ldkx;
    for (int i = wave; i < N; i += 4) {
        const bf16_t* qr = qkv + ((long long)bt * N + i) * ld;
        float acc = 0.f;
        for (int c = lane * 8; c < D; c += 512) {
            const bf16x8 a = *(const bf16x8*)(qr + c);
            const bf16x8 b = *(const bf16x8*)(kr + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += (float)a[e] * (float)b[e];
        }
        acc = wave_sum(acc);
        if (lane == 0) ssg[(long long)bt * N + i] = acc * scale;
    }
}

// lamda statistics of a frame whose token count is ONE more than the 256-row tile of the persistent GEMM (ViT-L/14: N = 257).
// The 256 x 256 block of scores q_i . k_j (i, j < N - 1) runs as ONE full tile per frame on the large-tile EXPSUM kernel; this
// kernel adds the border in one pass over q and one over k (two workgroups per frame):
//   y = 0 (q pass): ss[i] = scale q_i . kx for every i (the cross scores, as qk_cross) and the (max, sum exp) of
//                   scale q_i . k_{N-1}, i < N - 1  -> slot0;
//   y = 1 (k pass): (max, sum exp) of scale q_{N-1} . k_j, j < N  -> slot0 + 1.
// Eight rows per wave are in flight at a time (the one-row-at-a-time qk_cross is latency-bound: 143 us for 270 MB).
__global__ __launch_bounds__(256) void qk_border_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ kx, int ldkx,
                                                        float* __restrict__ ssg, float* __restrict__ part, int slot0, int nslots,
                                                        int N, int D, float scale) {
    __shared__ float sc[320];
    __shared__ float red[8];
    const int bt = blockIdx.x, pass = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = 3 * D, nch = D >> 9;          // 512-element chunks per row (D = 512 c: 1 | 2)
    const bf16_t* frame = qkv + (long long)bt * N * ld;
    // the fixed vectors of this pass: v0 = k_{N-1} (q pass) | q_{N-1} (k pass); v1 = kx (q pass only)
    const bf16_t* p0 = frame + (long long)(N - 1) * ld + (pass == 0 ? D : 0);
    const bf16_t* p1 = kx + (long long)bt * ldkx;
    bf16x8 v0[2], v1[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        v0[c] = v1[c] = bf16x8{};
        if (c < nch) {
            v0[c] = *(const bf16x8*)(p0 + c * 512 + lane * 8);
            if (pass == 0) v1[c] = *(const bf16x8*)(p1 + c * 512 + lane * 8);
        }
    }
    const bf16_t* rows = frame + (pass == 0 ? 0 : D);          // q rows | k rows
    const int nrow = N;
    for (int i0 = wave * 8; i0 < nrow; i0 += 32) {
        bf16x8 x[8][2];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int i = min(i0 + r, nrow - 1);
                x[r][c] = c < nch ? *(const bf16x8*)(rows + (long long)i * ld + c * 512 + lane * 8) : bf16x8{};
            }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xv = (float)x[r][c][e];
                    a0 = fmaf(xv, (float)v0[c][e], a0);
                    a1 = fmaf(xv, (float)v1[c][e], a1);
                }
            a0 = wave_sum(a0);
            if (pass == 0) a1 = wave_sum(a1);
            const int i = i0 + r;
            if (lane == 0 && i < nrow) {
                sc[i] = a0 * scale;
                if (pass == 0) ssg[(long long)bt * N + i] = a1 * scale;
            }
        }
    }
    __syncthreads();
    const int cnt = pass == 0 ? N - 1 : N;       // (the corner q_{N-1} . k_{N-1} belongs to the k pass)
    float mx = -INFINITY;
    for (int i = tid; i < cnt; i += 256) mx = fmaxf(mx, sc[i]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sm = 0.f;
    for (int i = tid; i < cnt; i += 256) sm += expf(sc[i] - mx);
    sm = wave_sum(sm);
    if (lane == 0) red[4 + wave] = sm;
    __syncthreads();
    if (tid == 0) {
        float* po = part + ((long long)bt * nslots + slot0 + pass) * 2;
        po[0] = mx;
        po[1] = red[4] + red[5] + red[6] + red[7];
    }
}

// one block (4 waves) per frame; ssg == nullptr: compute the cross scores here (one-call form)
__global__ __launch_bounds__(256) void lambda_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ kx,
                                                     int ldkx, const float* __restrict__ ssg,
                                                     const float* __restrict__ partials, int ntiles,
                                                     float* __restrict__ lam, float* __restrict__ oml, int N, int D,
                                                     float scale) {
    __shared__ float ss[320];
    __shared__ float red[8];
    const int bt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = 3 * D;
    if (ssg) {
        for (int i = tid; i < N; i += 256) ss[i] = ssg[(long long)bt * N + i];
    } else {
        const bf16_t* kr = kx + (long long)bt * ldkx;
        for (int i = wave; i < N; i += 4) {
            const bf16_t* qr = qkv + ((long long)bt * N + i) * ld;
            float acc = 0.f;
            for (int c = lane * 4; c < D; c += 256) {
                const bf16x4 a = *(const bf16x4*)(qr + c);
                const bf16x4 b = *(const bf16x4*)(kr + c);
                acc += (float)a[0] * (float)b[0] + (float)a[1] * (float)b[1] + (float)a[2] * (float)b[2] + (float)a[3] * (float)b[3];
            }
            acc = wave_sum(acc);
            if (lane == 0) ss[i] = acc * scale;
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int i = tid; i < N; i += 256) mx = fmaxf(mx, ss[i]);
    for (int t = tid; t < ntiles; t += 256) mx = fmaxf(mx, partials[((long long)bt * ntiles + t) * 2]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float cw = 0.f, ow = 0.f;
    for (int i = tid; i < N; i += 256) cw += expf(ss[i] - mx);
    for (int t = tid; t < ntiles; t += 256) {
        const float* p = partials + ((long long)bt * ntiles + t) * 2;
        if (p[0] > -INFINITY) ow += p[1] * expf(p[0] - mx);
    }
    cw = wave_sum(cw);
    ow = wave_sum(ow);
    __syncthreads();
    if (lane == 0) {
        red[wave] = cw;
        red[4 + wave] = ow;
    }
    __syncthreads();
    if (tid == 0) {
        const float c = red[0] + red[1] + red[2] + red[3], o = red[4] + red[5] + red[6] + red[7];
        const float l = c / (c + o);
        lam[bt] = l;
        if (oml) oml[bt] = 1.0f - l;
    }
}

}  // namespace

extern "C" int aim_cls_attn_fwd(const aim_bf16* qkv, aim_bf16* out_cls, float* probs, int B, int T, int N, int H,
                                void* stream) {
    AIM_CHECK_ARG(B > 0 && T > 0 && T <= TMAX && N > 0 && H > 0, "cls_attn_fwd: unsupported shape B=%d T=%d (T <= 32)", B, T);
    AIM_CHECK_ARG(qkv && out_cls && probs, "cls_attn_fwd: null pointer");
    hipLaunchKernelGGL(cls_attn_fwd_kernel, dim3(B * H), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qkv,
                       (bf16_t*)out_cls, probs, T, N, H);
    AIM_CHECK_LAUNCH("aim_cls_attn_fwd");
    return 0;
}

extern "C" int aim_cls_attn_bwd(const aim_bf16* qkv, const float* probs, const aim_bf16* dout_cls, aim_bf16* dqkv,
                                int compact, int B, int T, int N, int H, void* stream) {
    AIM_CHECK_ARG(B > 0 && T > 0 && T <= TMAX && N > 0 && H > 0, "cls_attn_bwd: unsupported shape B=%d T=%d (T <= 32)", B, T);
    AIM_CHECK_ARG(qkv && probs && dout_cls && dqkv, "cls_attn_bwd: null pointer");
    hipLaunchKernelGGL(cls_attn_bwd_kernel, dim3(B * H), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qkv, probs,
                       (const bf16_t*)dout_cls, (bf16_t*)dqkv, T, N, H, compact);
    AIM_CHECK_LAUNCH("aim_cls_attn_bwd");
    return 0;
}

// one thread per frame: 8 ow partials + 8 cw partials -> lamda
__global__ void lambda_partials_kernel(const float* __restrict__ part, float* __restrict__ lam, float* __restrict__ oml, int BT) {
    const int bt = blockIdx.x * blockDim.x + threadIdx.x;
    if (bt >= BT) return;
    const float* p = part + (long long)bt * 32;
    float mx = -INFINITY;
    for (int t = 0; t < 16; ++t) mx = fmaxf(mx, p[2 * t]);
    float o = 0.f, c = 0.f;
    for (int t = 0; t < 8; ++t)
        if (p[2 * t] > -INFINITY) o += p[2 * t + 1] * expf(p[2 * t] - mx);
    for (int t = 8; t < 16; ++t)
        if (p[2 * t] > -INFINITY) c += p[2 * t + 1] * expf(p[2 * t] - mx);
    const float l = c / (c + o);
    lam[bt] = l;
    if (oml) oml[bt] = 1.0f - l;
}

extern "C" int aim_lambda_partials(const float* partials, float* lam, float* one_minus_lam, int BT, void* stream) {
    AIM_CHECK_ARG(BT > 0 && partials && lam, "lambda_partials: bad arguments");
    hipLaunchKernelGGL(lambda_partials_kernel, dim3((BT + 63) / 64), dim3(64), 0, (hipStream_t)stream, partials, lam,
                       one_minus_lam, BT);
    AIM_CHECK_LAUNCH("aim_lambda_partials");
    return 0;
}

extern "C" int aim_qk_cross(const aim_bf16* qkv, const aim_bf16* kx, int ldkx, float* ss, int BT, int N, int D, float scale,
                            void* stream) {
    AIM_CHECK_ARG(BT > 0 && N > 0 && D > 0 && (D % 8) == 0 && ldkx >= D && (ldkx % 8) == 0, "qk_cross: unsupported shape BT=%d N=%d D=%d", BT, N, D);
    AIM_CHECK_ARG(qkv && kx && ss, "qk_cross: null pointer");
    hipLaunchKernelGGL(qk_cross_kernel, dim3(BT), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (const bf16_t*)kx,
                       ldkx, ss, N, D, scale);
    AIM_CHECK_LAUNCH("aim_qk_cross");
    return 0;
}

extern "C" int aim_qk_border(const aim_bf16* qkv, const aim_bf16* kx, int ldkx, float* ss, float* partials, int slot0, int nslots,
                             int BT, int N, int D, float scale, void* stream) {
    AIM_CHECK_ARG(BT > 0 && N > 1 && N <= 320 && (D == 512 || D == 1024) && ldkx >= D && (ldkx % 8) == 0,
                  "qk_border: unsupported shape BT=%d N=%d D=%d (D = 512 | 1024, N <= 320)", BT, N, D);
    AIM_CHECK_ARG(qkv && kx && ss && partials && slot0 >= 0 && slot0 + 2 <= nslots, "qk_border: bad arguments");
    hipLaunchKernelGGL(qk_border_kernel, dim3(BT, 2), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (const bf16_t*)kx, ldkx,
                       ss, partials, slot0, nslots, N, D, scale);
    AIM_CHECK_LAUNCH("aim_qk_border");
    return 0;
}

extern "C" int aim_lambda(const aim_bf16* qkv, const aim_bf16* kx, int ldkx, const float* ss, const float* partials,
                          int ntiles, float* lam, float* one_minus_lam, int BT, int N, int D, float scale, void* stream) {
    AIM_CHECK_ARG(BT > 0 && N > 0 && N <= 320 && D > 0 && (D % 4) == 0, "lambda: unsupported shape BT=%d N=%d D=%d", BT, N, D);
    AIM_CHECK_ARG(partials && lam && ntiles > 0, "lambda: bad arguments");
    AIM_CHECK_ARG(ss || (qkv && kx && ldkx >= D && (ldkx % 4) == 0), "lambda: needs either the cross scores or q and kx");
    hipLaunchKernelGGL(lambda_kernel, dim3(BT), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv,
                       (const bf16_t*)kx, ldkx, ss, partials, ntiles, lam, one_minus_lam, N, D, scale);
    AIM_CHECK_LAUNCH("aim_lambda");
    return 0;
}
