// LayerNorm forward / backward (fp32 statistics) for the AIM block.  gfx950 only.
//
// Replaces reference vit_clip.py:71-77 (`LayerNorm.forward`: fp32 nn.LayerNorm, eps 1e-5) at its
// call sites ln_1 (:224,264,265), ln_2 (:285), ln_pre (:447), ln_post (:452).
// One 64-lane wave per row, 4 rows per 256-thread block; the row lives in registers (float4 chunks),
// two-pass mean / variance, wavefront shuffles for the reductions; HBM-bound by design:
// fwd moves 4 B (x) + 2 B (bf16 y) per element.
#include "aim_common.h"
#include "aim_kernels_internal.h"

namespace {

constexpr int MAXC = 8;  // float4 chunks per lane: D <= 8*4*64 = 2048 (template NC = chunks actually used)

// (row elements as f32x4 from an fp32 or a bf16 row: the fp8 inference path keeps its residual stream in bf16)
template <typename TX>
__device__ __forceinline__ f32x4 ln_load4(const TX* p) {
    if constexpr (sizeof(TX) == 4) {
        return *(const f32x4*)p;
    } else {
        const bf16x4 b = *(const bf16x4*)p;
        return f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
    }
}

template <int NC, typename TX = float>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x, long long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     bf16_t* __restrict__ yb, float* __restrict__ yf, long long ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int rows, int D, float eps, unsigned char* __restrict__ y8 = nullptr) {
    const int lane = threadIdx.x & 63;
    const int row = (int)AIM_REV_BLOCK * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = D >> 2;
    const TX* xr = x + (long long)row * ldx;
    f32x4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nch) {
            v[c] = ln_load4(xr + ch * 4);
            s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
        }
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[c][e] - mu;
                q += d * d;
            }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nch) {
            const f32x4 g = *(const f32x4*)(gamma + ch * 4);
            const f32x4 b = *(const f32x4*)(beta + ch * 4);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[c][e] - mu) * rs * g[e] + b[e];
            if (yb) *(bf16x4*)(yb + (long long)row * ldy + ch * 4) = pack4(y[0], y[1], y[2], y[3]);
            if (yf) *(f32x4*)(yf + (long long)row * ldy + ch * 4) = y;
            if (y8) *(unsigned*)(y8 + (long long)row * ldy + ch * 4) = pack4_fp8(y[0], y[1], y[2], y[3]);
        }
    }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * gamma
template <int NC, typename TDY, typename TDR>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDY* __restrict__ dy, long long lddy,
                                                     const float* __restrict__ x, long long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const TDR* __restrict__ dres,
                                                     float* __restrict__ dx, bf16_t* __restrict__ dxb, long long lddx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = (int)AIM_REV_BLOCK * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = D >> 2;
    const float mu = mean[row], rs = rstd[row];
    f32x4 g[NC], xh[NC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nch) {
            f32x4 d;
            if constexpr (sizeof(TDY) == 4) {
                d = *(const f32x4*)(dy + (long long)row * lddy + ch * 4);
            } else {
                const bf16x4 db = *(const bf16x4*)(dy + (long long)row * lddy + ch * 4);
                d = f32x4{(float)db[0], (float)db[1], (float)db[2], (float)db[3]};
            }
            const f32x4 xv = *(const f32x4*)(x + (long long)row * ldx + ch * 4);
            const f32x4 gm = *(const f32x4*)(gamma + ch * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[c][e] = (xv[e] - mu) * rs;
                g[c][e] = d[e] * gm[e];
                s1 += g[c][e];
                s2 += g[c][e] * xh[c][e];
            }
            if (dgamma) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    atomicAdd(dgamma + ch * 4 + e, d[e] * xh[c][e]);
                    atomicAdd(dbeta + ch * 4 + e, d[e]);
                }
            }
        }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nch) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (g[c][e] - m1 - xh[c][e] * m2);
            if (dres) {
                if constexpr (sizeof(TDR) == 4) {
                    o += *(const f32x4*)(dres + (long long)row * lddx + ch * 4);
                } else {
                    const bf16x4 rb = *(const bf16x4*)(dres + (long long)row * lddx + ch * 4);
                    o += f32x4{(float)rb[0], (float)rb[1], (float)rb[2], (float)rb[3]};
                }
            }
            if (dx) *(f32x4*)(dx + (long long)row * lddx + ch * 4) = o;
            if (dxb) *(bf16x4*)(dxb + (long long)row * lddx + ch * 4) = pack4(o[0], o[1], o[2], o[3]);
        }
    }
}


// ln_bwd_kernel for bf16 dy / dres / dx that ALSO emits per-frame weighted column sums of the dx it stores:
//   partial[frame][g][c] = sum over the frame's token group g of w[n] * dx_bf16[frame*ntok + n][c].
// The block backward needs sum_n dms1[n] * d(x1)[frame, n, :] for the per-frame S_Adapter vector path; taken here, from the
// values in registers, it replaces a separate pass over the 155 MB tensor (aim_frame_sum) that opened the class-token chain
// on the side stream and ran starved of CUs beside a persistent GEMM.  One workgroup = one (frame, token group); a wave walks
// every fourth row of the group (the per-row arithmetic is ln_bwd_kernel's, bit for bit); the four waves' sums meet in LDS
// in a fixed order, the G groups are summed by aim_frame_sum over the [frames][G][D] partials (fixed order: reproducible).
// launch_bounds (256, 6) + a wave-uniform row index (scalar address arithmetic): left alone the compiler spends 98 VGPRs
// (4 waves per SIMD) and the kernel runs at 4.8 TB/s; now 70 VGPRs, 7 waves per SIMD.
template <int NC>
__global__ __launch_bounds__(256, (NC <= 4 ? 6 : 2)) void ln_bwd_fsum_kernel(const bf16_t* __restrict__ dy, long long lddy,
                                                          const float* __restrict__ x, long long ldx,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
                                                          bf16_t* __restrict__ dxb, long long lddx, const float* __restrict__ w,
                                                          float* __restrict__ partial, int ntok, int G, int D) {
    __shared__ f32x4 red[4][NC * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform rows: scalar address arithmetic
    const int bid = (int)AIM_REV_BLOCK;
    const int frame = bid / G, g = bid - frame * G;
    const int rg = (ntok + G - 1) / G;
    const int n_end = min(ntok, (g + 1) * rg);
    const int nch = D >> 2;
    f32x4 fs[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) fs[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma nounroll
    for (int n = g * rg + wave; n < n_end; n += 4) {
        asm volatile("" ::: "memory");          // one row at a time: no loads of the next row hoisted into this one (registers)
        const long long row = (long long)frame * ntok + n;
        const float mu = mean[row], rs = rstd[row];
        const float wt = w ? w[n] : 1.0f;
        f32x4 gv[NC], xh[NC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = lane + c * 64;
            if (ch < nch) {
                const bf16x4 db = *(const bf16x4*)(dy + row * lddy + ch * 4);
                const f32x4 d = f32x4{(float)db[0], (float)db[1], (float)db[2], (float)db[3]};
                const f32x4 xv = *(const f32x4*)(x + row * ldx + ch * 4);
                const f32x4 gm = *(const f32x4*)(gamma + ch * 4);      // (L1-resident; held across rows it costs 12 VGPRs)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[c][e] = (xv[e] - mu) * rs;
                    gv[c][e] = d[e] * gm[e];
                    s1 += gv[c][e];
                    s2 += gv[c][e] * xh[c][e];
                }
            }
        }
        const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = lane + c * 64;
            if (ch < nch) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (gv[c][e] - m1 - xh[c][e] * m2);
                const bf16x4 rb = *(const bf16x4*)(dres + row * lddx + ch * 4);
                o += f32x4{(float)rb[0], (float)rb[1], (float)rb[2], (float)rb[3]};
                const bf16x4 ob = pack4(o[0], o[1], o[2], o[3]);
                *(bf16x4*)(dxb + row * lddx + ch * 4) = ob;
#pragma unroll
                for (int e = 0; e < 4; ++e) fs[c][e] += wt * (float)ob[e];      // the STORED (bf16) value, like a pass over dx
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) red[wave][lane + c * 64] = fs[c];
    __syncthreads();
    for (int ch = threadIdx.x; ch < nch; ch += 256)
        *(f32x4*)(partial + (long long)bid * D + ch * 4) = (red[0][ch] + red[1][ch]) + (red[2][ch] + red[3][ch]);
}

}  // namespace

extern "C" int aim_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta,
                                 aim_bf16* y_bf16, float* y_f32, int64_t ldy, float* mean, float* rstd,
                                 int rows, int D, float eps, void* stream) {
    AIM_CHECK_ARG(rows > 0 && D > 0 && (D % 4) == 0 && D <= MAXC * 256, "layernorm_fwd: bad shape rows=%d D=%d", rows, D);
    AIM_CHECK_ARG(x && gamma && beta && (y_bf16 || y_f32), "layernorm_fwd: null pointer");
    AIM_CHECK_ARG((ldx % 4) == 0 && (ldy % 4) == 0, "layernorm_fwd: strides must be multiples of 4");
#define AIM_LN_FWD(NC)                                                                                            \
    hipLaunchKernelGGL(ln_fwd_kernel<NC>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, \
                       gamma, beta, (bf16_t*)y_bf16, y_f32, (long long)ldy, mean, rstd, rows, D, eps)
    const int nc = (D + 255) / 256;
    if (nc <= 1) AIM_LN_FWD(1); else if (nc == 2) AIM_LN_FWD(2); else if (nc == 3) AIM_LN_FWD(3);
    else if (nc == 4) AIM_LN_FWD(4); else AIM_LN_FWD(8);
#undef AIM_LN_FWD
    AIM_CHECK_LAUNCH("aim_layernorm_fwd");
    return 0;
}

extern "C" int aim_layernorm_fwd_fp8(const float* x, int64_t ldx, const float* gamma, const float* beta, uint8_t* y_fp8,
                                     int64_t ldy, int rows, int D, float eps, void* stream) {
    AIM_CHECK_ARG(rows > 0 && D > 0 && (D % 4) == 0 && D <= MAXC * 256, "layernorm_fwd_fp8: bad shape rows=%d D=%d", rows, D);
    AIM_CHECK_ARG(x && gamma && beta && y_fp8, "layernorm_fwd_fp8: null pointer");
    AIM_CHECK_ARG((ldx % 4) == 0 && (ldy % 4) == 0, "layernorm_fwd_fp8: strides must be multiples of 4");
#define AIM_LN_FWD8(NC)                                                                                           \
    hipLaunchKernelGGL(ln_fwd_kernel<NC>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, \
                       gamma, beta, (bf16_t*)nullptr, (float*)nullptr, (long long)ldy, (float*)nullptr, (float*)nullptr, rows, D, eps, y_fp8)
    const int nc = (D + 255) / 256;
    if (nc <= 1) AIM_LN_FWD8(1); else if (nc == 2) AIM_LN_FWD8(2); else if (nc == 3) AIM_LN_FWD8(3);
    else if (nc == 4) AIM_LN_FWD8(4); else AIM_LN_FWD8(8);
#undef AIM_LN_FWD8
    AIM_CHECK_LAUNCH("aim_layernorm_fwd_fp8");
    return 0;
}

// bf16 rows in (the fp8 inference path's residual stream); y as bf16 and / or f32 and / or fp8
extern "C" int aim_layernorm_fwd_x16(const aim_bf16* x, int64_t ldx, const float* gamma, const float* beta, aim_bf16* y_bf16,
                                     float* y_f32, uint8_t* y_fp8, int64_t ldy, int rows, int D, float eps, void* stream) {
    AIM_CHECK_ARG(rows > 0 && D > 0 && (D % 4) == 0 && D <= MAXC * 256, "layernorm_fwd_x16: bad shape rows=%d D=%d", rows, D);
    AIM_CHECK_ARG(x && gamma && beta && (y_bf16 || y_f32 || y_fp8), "layernorm_fwd_x16: null pointer");
    AIM_CHECK_ARG((ldx % 4) == 0 && (ldy % 4) == 0, "layernorm_fwd_x16: strides must be multiples of 4");
#define AIM_LN_FWDX(NC)                                                                                              \
    hipLaunchKernelGGL((ln_fwd_kernel<NC, bf16_t>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, \
                       (long long)ldx, gamma, beta, (bf16_t*)y_bf16, y_f32, (long long)ldy, (float*)nullptr, (float*)nullptr, \
                       rows, D, eps, y_fp8)
    const int nc = (D + 255) / 256;
    if (nc <= 1) AIM_LN_FWDX(1); else if (nc == 2) AIM_LN_FWDX(2); else if (nc == 3) AIM_LN_FWDX(3);
    else if (nc == 4) AIM_LN_FWDX(4); else AIM_LN_FWDX(8);
#undef AIM_LN_FWDX
    AIM_CHECK_LAUNCH("aim_layernorm_fwd_x16");
    return 0;
}

namespace {
// dgamma[c] += sum_r dy[r][c] * xhat[r][c], dbeta[c] += sum_r dy[r][c] in a fixed order (bitwise reproducible).  The path trains ln_post only (B*T rows), so the walk is short; beyond AIM_LN_DPARAM_ROWS rows the
// main kernel's per-element atomics are used instead.
// (256 threads = 64 columns x 4 row groups, a thread walks rows g, g+4, ... with the loads of four rows in flight, the groups
//  meet in LDS in a fixed order: the first version walked all rows in one thread, one dependent row at a time -- 146 us for
//  512 rows of ln_post.)
template <typename TDY>
__global__ __launch_bounds__(256) void ln_dparam_kernel(const TDY* __restrict__ dy, long long lddy, const float* __restrict__ x,
                                                       long long ldx, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, int rows, int D) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool live = c < D;
    const int cc = live ? c : 0;
    float ag = 0.f, ab = 0.f;
    int r = g;
    for (; r + 12 < rows; r += 16) {
        float d[4], xv[4], mu[4], rs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + 4 * u;
            d[u] = (float)dy[(long long)rr * lddy + cc];
            xv[u] = x[(long long)rr * ldx + cc];
            mu[u] = mean[rr];
            rs[u] = rstd[rr];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ag += d[u] * ((xv[u] - mu[u]) * rs[u]);
            ab += d[u];
        }
    }
    for (; r < rows; r += 4) {
        const float d = (float)dy[(long long)r * lddy + cc];
        ag += d * ((x[(long long)r * ldx + cc] - mean[r]) * rstd[r]);
        ab += d;
    }
    red[0][g][cl] = ag;
    red[1][g][cl] = ab;
    __syncthreads();
    if (g == 0 && live) {
        dgamma[c] += (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        dbeta[c] += (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    }
}
constexpr int AIM_LN_DPARAM_ROWS = 8192;
}  // namespace

extern "C" int aim_layernorm_bwd(const void* dy, int dy_is_bf16, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                                 const float* mean, const float* rstd, const void* dres, int dres_is_bf16, float* dx,
                                 aim_bf16* dx_bf16, int64_t lddx, float* dgamma, float* dbeta, int rows, int D,
                                 void* stream) {
    AIM_CHECK_ARG(rows > 0 && D > 0 && (D % 4) == 0 && D <= MAXC * 256, "layernorm_bwd: bad shape rows=%d D=%d", rows, D);
    AIM_CHECK_ARG(dy && x && gamma && mean && rstd && (dx || dx_bf16), "layernorm_bwd: null pointer");
    AIM_CHECK_ARG((!dgamma) == (!dbeta), "layernorm_bwd: dgamma and dbeta go together");
    AIM_CHECK_ARG((ldx % 4) == 0 && (lddy % 4) == 0 && (lddx % 4) == 0, "layernorm_bwd: strides must be multiples of 4");
    if (dgamma && rows <= AIM_LN_DPARAM_ROWS) {      // parameter gradients apart, in a fixed order
        if (dy_is_bf16)
            hipLaunchKernelGGL(ln_dparam_kernel<bf16_t>, dim3((D + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                               (long long)lddy, x, (long long)ldx, mean, rstd, dgamma, dbeta, rows, D);
        else
            hipLaunchKernelGGL(ln_dparam_kernel<float>, dim3((D + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const float*)dy,
                               (long long)lddy, x, (long long)ldx, mean, rstd, dgamma, dbeta, rows, D);
        AIM_CHECK_LAUNCH("aim_layernorm_bwd(dparam)");
        dgamma = nullptr;
        dbeta = nullptr;
    }
#define AIM_LN_BWD_T(NC, TDY, TDR)                                                                                  \
    hipLaunchKernelGGL((ln_bwd_kernel<NC, TDY, TDR>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,        \
                       (const TDY*)dy, (long long)lddy, x, (long long)ldx, gamma, mean, rstd, (const TDR*)dres, dx,   \
                       (bf16_t*)dx_bf16, (long long)lddx, dgamma, dbeta, rows, D)
#define AIM_LN_BWD(NC)                                                                                              \
    if (dy_is_bf16 && dres_is_bf16) AIM_LN_BWD_T(NC, bf16_t, bf16_t);                                               \
    else if (dy_is_bf16) AIM_LN_BWD_T(NC, bf16_t, float);                                                           \
    else if (dres_is_bf16) AIM_LN_BWD_T(NC, float, bf16_t);                                                         \
    else AIM_LN_BWD_T(NC, float, float)
    const int nc = (D + 255) / 256;
    if (nc <= 1) AIM_LN_BWD(1); else if (nc == 2) AIM_LN_BWD(2); else if (nc == 3) AIM_LN_BWD(3);
    else if (nc == 4) AIM_LN_BWD(4); else AIM_LN_BWD(8);
#undef AIM_LN_BWD
#undef AIM_LN_BWD_T
    AIM_CHECK_LAUNCH("aim_layernorm_bwd");
    return 0;
}

extern "C" int aim_layernorm_bwd_fsum(const aim_bf16* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                                      const float* mean, const float* rstd, const aim_bf16* dres, aim_bf16* dx_bf16,
                                      int64_t lddx, const float* w, float* fsum_partial, int groups, int frames, int ntok,
                                      int D, void* stream) {
    AIM_CHECK_ARG(frames > 0 && ntok > 0 && groups > 0 && D > 0 && (D % 4) == 0 && D <= 8 * 256,
                  "layernorm_bwd_fsum: bad shape frames=%d ntok=%d groups=%d D=%d", frames, ntok, groups, D);
    AIM_CHECK_ARG(dy && x && gamma && mean && rstd && dres && dx_bf16 && fsum_partial, "layernorm_bwd_fsum: null pointer");
    AIM_CHECK_ARG((ldx % 4) == 0 && (lddy % 4) == 0 && (lddx % 4) == 0, "layernorm_bwd_fsum: strides must be multiples of 4");
#define AIM_LN_FS(NC)                                                                                                  \
    hipLaunchKernelGGL(ln_bwd_fsum_kernel<NC>, dim3(frames * groups), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, \
                       (long long)lddy, x, (long long)ldx, gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx_bf16,    \
                       (long long)lddx, w, fsum_partial, ntok, groups, D)
    const int nc = (D + 255) / 256;
    if (nc <= 1) AIM_LN_FS(1); else if (nc == 2) AIM_LN_FS(2); else if (nc == 3) AIM_LN_FS(3);
    else if (nc == 4) AIM_LN_FS(4); else AIM_LN_FS(8);
#undef AIM_LN_FS
    AIM_CHECK_LAUNCH("aim_layernorm_bwd_fsum");
    return 0;
}
