// Patch-embedding glue and the small reductions / casts around the AIM block.  gfx950 only.
//
// patchify / embed_ln / embed_bwd: reference vit_clip.py:434-447 -- rearrange 'b c t h w -> (b t) c h w',
//   conv1 (kernel = stride = patch, no bias; done as a GEMM over the patch matrix written here),
//   class_embedding concat (:439), + positional_embedding (:440), + temporal_embedding (:443-445),
//   ln_pre (:447).  Optional uint8 input fuses the GPUNormalize pre-hook
//   (mmaction/utils/module_hooks.py:73-85): x.float().sub_(mean).div_(std).
// frame_sum / colsum / cast / scale_rows: bandwidth-bound helpers for the hand-written backward
//   (per-frame token sums for the broadcast S_Adapter term, bias gradients, bf16 weight staging).
#include "aim_common.h"
#include "aim_kernels_internal.h"

namespace {

constexpr int MAXC = 8;

// A[(bt*G*G + gy*G + gx)][c*p*p + py*p + px] = img[b][c][t][gy*p+py][gx*p+px]; one thread per 8 columns
template <typename TIN>
__global__ __launch_bounds__(256) void patchify_kernel(const TIN* __restrict__ img, const float* __restrict__ mean3,
                                                       const float* __restrict__ std3, bf16_t* __restrict__ A, int B,
                                                       int T, int H, int W, int p, int Kp) {
    const int G = W / p, Gy = H / p, K = 3 * p * p;
    const long long rows = (long long)B * T * Gy * G;
    const int cpr = Kp / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * cpr) return;
    const long long row = idx / cpr;
    const int k0 = (int)(idx - row * cpr) * 8;
    const int gx = (int)(row % G), gy = (int)((row / G) % Gy);
    const long long bt = row / ((long long)G * Gy);
    const int t = (int)(bt % T);
    const long long b = bt / T;
    bf16x8 o;
    if ((p & 7) == 0 && (W & 7) == 0 && k0 + 8 <= K && ((unsigned long long)img & 15) == 0) {
        // patch width a multiple of 8 (ViT-B/16): the thread's 8 columns are 8 consecutive pixels of one image row -> one or
        // two 16-byte loads instead of eight scalar ones with their index arithmetic
        const int c = k0 / (p * p), rem = k0 - c * p * p, py = rem / p, px = rem - py * p;
        const TIN* src = img + (((b * 3 + c) * T + t) * H + (gy * p + py)) * (long long)W + gx * p + px;
        float v[8];
        if constexpr (sizeof(TIN) == 4) {
            const f32x4 a0 = *(const f32x4*)src, a1 = *(const f32x4*)(src + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = a0[e];
                v[4 + e] = a1[e];
            }
        } else if constexpr (sizeof(TIN) == 2) {
            const bf16x8 a0 = *(const bf16x8*)src;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)a0[e];
        } else {
            const uint2 a0 = *(const uint2*)src;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (float)((a0.x >> (8 * e)) & 0xffu);
                v[4 + e] = (float)((a0.y >> (8 * e)) & 0xffu);
            }
        }
        if (mean3) {
            const float m = mean3[c], sd = std3[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (v[e] - m) / sd;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            float v = 0.f;
            if (k < K) {
                const int c = k / (p * p), rem = k - c * p * p, py = rem / p, px = rem - py * p;
                const long long src = (((b * 3 + c) * T + t) * H + (gy * p + py)) * (long long)W + gx * p + px;
                v = (float)img[src];
                if (mean3) v = (v - mean3[c]) / std3[c];
            }
            o[e] = (bf16_t)v;
        }
    }
    *(bf16x8*)(A + row * Kp + k0) = o;
}

__device__ __forceinline__ f32x4 embed_value(const bf16_t* tok, const float* cls, const float* pos, const float* tmp,
                                             long long bt, int n, int t, int G2, int D, int c4) {
    f32x4 v;
    if (n == 0) {
        v = *(const f32x4*)(cls + c4);
    } else {
        const bf16x4 tk = *(const bf16x4*)(tok + (bt * G2 + (n - 1)) * D + c4);
        v = f32x4{(float)tk[0], (float)tk[1], (float)tk[2], (float)tk[3]};
    }
    v += *(const f32x4*)(pos + (long long)n * D + c4);
    v += *(const f32x4*)(tmp + (long long)t * D + c4);
    return v;
}

template <int NC>
__global__ __launch_bounds__(256) void embed_ln_kernel(const bf16_t* __restrict__ tok, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, const float* __restrict__ tmp,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ x, float* __restrict__ mean,
                                                       float* __restrict__ rstd, int B, int T, int N, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)B * T * N) return;
    const long long bt = row / N;
    const int n = (int)(row - bt * N), t = (int)(bt % T), nch = D >> 2;
    f32x4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nch) {
            v[c] = embed_value(tok, cls, pos, tmp, bt, n, t, N - 1, D, ch * 4);
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
        mean[row] = mu;
        rstd[row] = rs;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nch) {
            const f32x4 g = *(const f32x4*)(gamma + ch * 4), b = *(const f32x4*)(beta + ch * 4);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[c][e] - mu) * rs * g[e] + b[e];
            *(f32x4*)(x + row * D + ch * 4) = y;
        }
    }
}

// grid (T, chunks): every wave walks rows (b, n) of frame-time t, accumulates ln_pre's input gradient
// in registers and adds it into dtemporal[t] once at the end.
template <typename T>
__device__ __forceinline__ f32x4 load4f(const T* p) {
    if constexpr (sizeof(T) == 4) {
        return *(const f32x4*)p;
    } else {
        const bf16x4 b = *(const bf16x4*)p;
        return f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
    }
}

template <int NC, typename TDX>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const TDX* __restrict__ dx, const bf16_t* __restrict__ tok,
                                                        const float* __restrict__ cls, const float* __restrict__ pos,
                                                        const float* __restrict__ tmp, const float* __restrict__ gamma,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        float* __restrict__ dtemporal, float* __restrict__ partial, int B,
                                                        int T, int N, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x, nch = D >> 2;
    f32x4 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int total = B * N;
    for (int r = blockIdx.y * 4 + wave; r < total; r += gridDim.y * 4) {
        const int b = r / N, n = r - b * N;
        const long long bt = (long long)b * T + t, row = bt * N + n;
        const float mu = mean[row], rs = rstd[row];
        f32x4 g[NC], xh[NC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = lane + c * 64;
            if (ch < nch) {
                const f32x4 v = embed_value(tok, cls, pos, tmp, bt, n, t, N - 1, D, ch * 4);
                const f32x4 d = load4f(dx + row * D + ch * 4);
                const f32x4 gm = *(const f32x4*)(gamma + ch * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[c][e] = (v[e] - mu) * rs;
                    g[c][e] = d[e] * gm[e];
                    s1 += g[c][e];
                    s2 += g[c][e] * xh[c][e];
                }
            }
        }
        const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = lane + c * 64;
            if (ch < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[c][e] += rs * (g[c][e] - m1 - xh[c][e] * m2);
            }
        }
    }
    if (partial) {
        // deterministic form: the block's four waves are summed in a fixed order through LDS and the block's partial row goes
        // to partial[t][blockIdx.y][D]; embed_bwd_finish_kernel adds the rows up in order (no atomics anywhere)
        __shared__ float red[3][NC * 256];
        if (wave > 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int ch = lane + c * 64;
                if (ch < nch) *(f32x4*)(&red[wave - 1][ch * 4]) = acc[c];
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int ch = lane + c * 64;
                if (ch < nch) {
                    f32x4 a = acc[c];
                    for (int w = 0; w < 3; ++w) a += *(const f32x4*)(&red[w][ch * 4]);
                    *(f32x4*)(partial + ((long long)t * gridDim.y + blockIdx.y) * D + ch * 4) = a;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(dtemporal + (long long)t * D + ch * 4 + e, acc[c][e]);
        }
    }
}

// dtemporal[t][c] += sum over the blocks' partial rows, in a fixed order: 64 columns x 4 row groups per workgroup, eight loads in
// flight per thread (the first version walked the 256 partial rows one dependent load at a time: 114 us)
__global__ __launch_bounds__(256) void embed_bwd_finish_kernel(const float* __restrict__ partial, float* __restrict__ dtemporal,
                                                               int chunks, int D) {
    __shared__ float red[4][64];
    const int t = blockIdx.x, cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const bool live = c < D;
    const float* p = partial + (long long)t * chunks * D + (live ? c : 0);
    float a = 0.f;
    int y = g;
    for (; y + 28 < chunks; y += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(y + 4 * u) * D];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; y < chunks; y += 4) a += p[(long long)y * D];
    red[g][cl] = a;
    __syncthreads();
    if (g == 0 && live) dtemporal[(long long)t * D + c] += (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// out[frame][d] = sum_tok w[tok] x[frame][tok][d].  Block (frame, 64-column group): 4 token groups x 64 float4-columns; a
// thread walks tokens g, g+4, ... with eight loads in flight, the four partial sums meet in LDS in a fixed order.  (The first
// version walked all tokens of a column in one thread, four loads in flight: latency-bound at 0.6 TB/s, and it opens the
// backward's class-token chain.)
template <typename TX>
__global__ __launch_bounds__(256) void frame_sum_kernel(const TX* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ out, int ntok, int D) {
    __shared__ f32x4 part[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int d4 = blockIdx.y * 64 + cl;                   // float4 column
    const bool live = d4 * 4 < D;
    const long long f = blockIdx.x;
    const TX* p = x + f * ntok * (long long)D + (live ? d4 * 4 : 0);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    int t = g;
    for (; t + 28 < ntok; t += 32) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = load4f(p + (long long)(t + 4 * u) * D);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (w ? w[t + 4 * u] : 1.0f) * v[u];
    }
    for (; t < ntok; t += 4) acc += (w ? w[t] : 1.0f) * load4f(p + (long long)t * D);
    part[g][cl] = acc;
    __syncthreads();
    if (g == 0 && live) *(f32x4*)(out + f * D + d4 * 4) = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ X, int ldx, const float* __restrict__ af,
                                                     const float* __restrict__ at, int ntok, float* __restrict__ out,
                                                     int M, int C, int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) {
        float rs = 1.f;
        if (af || at) {
            const int f = r / ntok, tk = r - f * ntok;
            if (af) rs *= af[f];
            if (at) rs *= at[tk];
        }
        acc += rs * (float)X[(long long)r * ldx + c];
    }
    atomicAdd(out + c, acc);
}

// 8 columns per thread (16-byte loads); threads = (C/8 column groups) x (row slots); rows strided by slots
__global__ __launch_bounds__(256) void colsum8_kernel(const bf16_t* __restrict__ X, int ldx, const float* __restrict__ af,
                                                      const float* __restrict__ at, int ntok, float* __restrict__ out,
                                                      float* __restrict__ partial, int M, int C, int rows_per_block) {
    const int cg = C >> 3;
    const int nrs = 256 / cg > 0 ? 256 / cg : 1;
    const int t = threadIdx.x;
    const int col = t % cg;   // column group
    const int rslot = t / cg;
    const bool live = rslot < nrs;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // 4 independent 16-byte loads in flight per thread
    int r = r0 + rslot;
    for (; live && r + 3 * nrs < r1; r += 4 * nrs) {
        bf16x8 v[4];
        float rs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * nrs;
            v[u] = *(const bf16x8*)(X + (long long)rr * ldx + col * 8);
            rs[u] = 1.f;
            if (af || at) {
                const int f = rr / ntok, tk = rr - f * ntok;
                if (af) rs[u] *= af[f];
                if (at) rs[u] *= at[tk];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += rs[u] * (float)v[u][e];
    }
    for (; live && r < r1; r += nrs) {
        float rs = 1.f;
        if (af || at) {
            const int f = r / ntok, tk = r - f * ntok;
            if (af) rs *= af[f];
            if (at) rs *= at[tk];
        }
        const bf16x8 v = *(const bf16x8*)(X + (long long)r * ldx + col * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += rs * (float)v[e];
    }
    // reduce the row slots of this block through LDS, then ONE atomic per column per block
    __shared__ float red[256 * 8];
    if (live) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(rslot * cg + col) * 8 + e] = acc[e];
    }
    __syncthreads();
    if (rslot == 0) {
        for (int s2 = 1; s2 < nrs; ++s2)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += red[(s2 * cg + col) * 8 + e];
        if (partial) {       // stage 1 of a two-stage reduction: no atomics, no contention
            f32x4* p = (f32x4*)(partial + (long long)blockIdx.y * C + col * 8);
            p[0] = f32x4{acc[0], acc[1], acc[2], acc[3]};
            p[1] = f32x4{acc[4], acc[5], acc[6], acc[7]};
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(out + col * 8 + e, acc[e]);
        }
    }
}

// stage 2: out[c] += sum_p partial[p][c].  ONE block of 64 columns x 16 row-slots per column group: every sum has a fixed
// order (slot s adds partials s, s+16, ...; the slots are added 0..15) and the update is a plain read-modify-write --
// bitwise reproducible, no atomics; it runs off the critical path.
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                             int P, int C) {
    __shared__ float red[1024];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int slot = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < C)
        for (int p = slot; p < P; p += 16) acc += partial[(long long)p * C + c];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (slot == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) t += red[threadIdx.x + 64 * s2];
        out[c] += t;
    }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 v = *(const f32x4*)(src + i);
        *(bf16x4*)(dst + i) = pack4(v[0], v[1], v[2], v[3]);
    } else {
        for (long long j = i; j < n; ++j) dst[j] = (bf16_t)src[j];
    }
}

// row-strided destination: dst[r * ldd + c] = src[r * C + c]
__global__ __launch_bounds__(256) void cast_strided_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int R,
                                                           int C, int ldd) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)R * C) return;
    const int r = (int)(i / C), c = (int)(i - (long long)r * C);
    dst[(long long)r * ldd + c] = (bf16_t)src[i];
}

// dst[c][r] = src[r][c]
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                             int R, int C, int ldd) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < R && c < C) ? src[(long long)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < C && r < R) dst[(long long)c * ldd + r] = (bf16_t)tile[tx][j];
    }
}

__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                         bf16_t* __restrict__ y, float* __restrict__ yf, int R, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)R * C) return;
    const float v = x[i] * s[i / C];
    if (y) y[i] = (bf16_t)v;
    if (yf) yf[i] = v;
}

// One launch for a whole table of small casts (this step's adapter weights -> bf16 GEMM operands, both
// orientations).  grid (n descriptors, 16): block (d, y) strides over descriptor d's elements.
__global__ __launch_bounds__(256) void cast_multi_kernel(const aim_cast_desc* __restrict__ table) {
    const aim_cast_desc d = table[blockIdx.x];
    const float* src = (const float*)d.src;
    bf16_t* dst = (bf16_t*)d.dst;
    const int n = d.R * d.C;                        // (ops.CastTable checks R * C < 2^31)
    if (d.transpose == 0 && (d.C & 3) == 0 && (d.ldd & 3) == 0 && ((unsigned long long)src & 15) == 0 && ((unsigned long long)dst & 7) == 0) {
        // row-major cast, four elements per thread
        const int n4 = n >> 2, c4n = d.C >> 2;
        for (int i = (int)blockIdx.y * 256 + threadIdx.x; i < n4; i += (int)gridDim.y * 256) {
            const int r = i / c4n, c = (i - r * c4n) * 4;
            const f32x4 v = *(const f32x4*)(src + (long long)i * 4);
            *(bf16x4*)(dst + (long long)r * d.ldd + c) = pack4(v[0], v[1], v[2], v[3]);
        }
        return;
    }
    if (d.transpose == 1 && (d.R & 31) == 0 && (d.C & 31) == 0) {
        // transposed cast by 32 x 32 tiles through LDS: coalesced reads along C, coalesced writes along R
        __shared__ float tile[32][33];
        const int tc = d.C >> 5, ntile = (d.R >> 5) * tc;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 8 rows of 32 per pass
        for (int tl = blockIdx.y; tl < ntile; tl += gridDim.y) {
            const int r0 = (tl / tc) * 32, c0 = (tl - (tl / tc) * tc) * 32;
#pragma unroll
            for (int k = 0; k < 4; ++k) tile[ty + k * 8][tx] = src[(long long)(r0 + ty + k * 8) * d.C + c0 + tx];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) dst[(long long)(c0 + ty + k * 8) * d.ldd + r0 + tx] = (bf16_t)tile[tx][ty + k * 8];
            __syncthreads();
        }
        return;
    }
    // (r, c) advance incrementally: no division per element
    const int step = (int)gridDim.y * 256, dr = step / d.C, dc = step - dr * d.C;
    int i = (int)blockIdx.y * 256 + threadIdx.x;
    int r = i / d.C, c = i - r * d.C;
    for (; i < n; i += step) {
        const float v = src[i];
        if (d.transpose == 2) ((float*)d.dst)[(long long)r * d.ldd + c] = v;       // fp32 copy (bias staging)
        else if (d.transpose) dst[(long long)c * d.ldd + r] = (bf16_t)v;
        else dst[(long long)r * d.ldd + c] = (bf16_t)v;
        r += dr;
        c += dc;
        if (c >= d.C) {
            c -= d.C;
            ++r;
        }
    }
}

// AdamW (decoupled weight decay) on flat fp32 buffers, torch.optim.AdamW semantics.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2s, float gs) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        f32x4 pp = *(f32x4*)(p + i), mm = *(f32x4*)(m + i), vv = *(f32x4*)(v + i);
        const f32x4 gg = *(const f32x4*)(g + i) * gs;       // gs = 1 / world: the SUM all-reduce becomes the mean here
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pp[e] *= 1.0f - lr * wd;
            mm[e] = b1 * mm[e] + (1.0f - b1) * gg[e];
            vv[e] = b2 * vv[e] + (1.0f - b2) * gg[e] * gg[e];
            pp[e] -= (lr / bc1) * mm[e] / (sqrtf(vv[e]) / bc2s + eps);
        }
        *(f32x4*)(p + i) = pp;
        *(f32x4*)(m + i) = mm;
        *(f32x4*)(v + i) = vv;
    } else {
        for (long long j = i; j < n; ++j) {
            float pp = p[j] * (1.0f - lr * wd);
            const float gj = g[j] * gs;
            const float mm = b1 * m[j] + (1.0f - b1) * gj, vv = b2 * v[j] + (1.0f - b2) * gj * gj;
            pp -= (lr / bc1) * mm / (sqrtf(vv) / bc2s + eps);
            p[j] = pp; m[j] = mm; v[j] = vv;
        }
    }
}

}  // namespace

extern "C" int aim_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
    AIM_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adamw: bad arguments");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (long long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
    AIM_CHECK_LAUNCH("aim_adamw_flat");
    return 0;
}

extern "C" int aim_patchify(const void* imgs, int in_dtype, const float* mean3, const float* std3, aim_bf16* A, int B,
                            int T, int H, int W, int p, int Kp, void* stream) {
    AIM_CHECK_ARG(B > 0 && T > 0 && p > 0 && H % p == 0 && W % p == 0, "patchify: bad shape H=%d W=%d p=%d", H, W, p);
    AIM_CHECK_ARG(Kp >= 3 * p * p && (Kp % 8) == 0, "patchify: Kp=%d must be >= 3*p*p and a multiple of 8", Kp);
    AIM_CHECK_ARG(imgs && A && ((!mean3) == (!std3)), "patchify: null pointer");
    const long long total = (long long)B * T * (H / p) * (W / p) * (Kp / 8);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (in_dtype == 0)
        hipLaunchKernelGGL(patchify_kernel<float>, grid, block, 0, st, (const float*)imgs, mean3, std3, (bf16_t*)A, B, T, H, W, p, Kp);
    else if (in_dtype == 1)
        hipLaunchKernelGGL(patchify_kernel<uint8_t>, grid, block, 0, st, (const uint8_t*)imgs, mean3, std3, (bf16_t*)A, B, T, H, W, p, Kp);
    else if (in_dtype == 2)
        hipLaunchKernelGGL(patchify_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)imgs, mean3, std3, (bf16_t*)A, B, T, H, W, p, Kp);
    else {
        aim_set_error("patchify: in_dtype must be 0 (f32), 1 (uint8) or 2 (bf16), got %d", in_dtype);
        return 1;
    }
    AIM_CHECK_LAUNCH("aim_patchify");
    return 0;
}

extern "C" int aim_embed_ln(const aim_bf16* tok, const float* cls, const float* pos, const float* temporal,
                            const float* gamma, const float* beta, float* x, float* mean, float* rstd, int B, int T,
                            int N, int D, float eps, void* stream) {
    AIM_CHECK_ARG(B > 0 && T > 0 && N > 1 && D > 0 && (D % 4) == 0 && D <= MAXC * 256, "embed_ln: bad shape N=%d D=%d", N, D);
    AIM_CHECK_ARG(tok && cls && pos && temporal && gamma && beta && x && mean && rstd, "embed_ln: null pointer");
    const long long rows = (long long)B * T * N;
#define AIM_EL(NC)                                                                                                 \
    hipLaunchKernelGGL(embed_ln_kernel<NC>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,      \
                       (const bf16_t*)tok, cls, pos, temporal, gamma, beta, x, mean, rstd, B, T, N, D, eps)
    const int nc = (D + 255) / 256;
    if (nc <= 1) AIM_EL(1); else if (nc == 2) AIM_EL(2); else if (nc == 3) AIM_EL(3); else if (nc == 4) AIM_EL(4); else AIM_EL(8);
#undef AIM_EL
    AIM_CHECK_LAUNCH("aim_embed_ln");
    return 0;
}

extern "C" int aim_embed_bwd(const void* dx, int dx_is_bf16, const aim_bf16* tok, const float* cls, const float* pos,
                             const float* temporal, const float* gamma, const float* mean, const float* rstd,
                             float* dtemporal, int B, int T, int N, int D, float* workspace, int64_t workspace_bytes,
                             void* stream) {
    AIM_CHECK_ARG(B > 0 && T > 0 && N > 1 && D > 0 && (D % 4) == 0 && D <= MAXC * 256, "embed_bwd: bad shape N=%d D=%d", N, D);
    AIM_CHECK_ARG(dx && tok && cls && pos && temporal && gamma && mean && rstd && dtemporal, "embed_bwd: null pointer");
    int chunks = (B * N + 3) / 4;
    const int want = (2048 + T - 1) / T;
    if (chunks > want) chunks = want;
    // with scratch ([T][chunks][D] fp32, aim_embed_bwd_workspace_bytes) the reduction is two-stage and bitwise reproducible
    float* partial = (workspace && workspace_bytes >= (int64_t)T * chunks * D * 4) ? workspace : nullptr;
#define AIM_EB(NC)                                                                                                     \
    if (dx_is_bf16)                                                                                                    \
        hipLaunchKernelGGL((embed_bwd_kernel<NC, bf16_t>), dim3(T, chunks), dim3(256), 0, (hipStream_t)stream,          \
                           (const bf16_t*)dx, (const bf16_t*)tok, cls, pos, temporal, gamma, mean, rstd, dtemporal, partial, B, T, N, D); \
    else                                                                                                               \
        hipLaunchKernelGGL((embed_bwd_kernel<NC, float>), dim3(T, chunks), dim3(256), 0, (hipStream_t)stream,           \
                           (const float*)dx, (const bf16_t*)tok, cls, pos, temporal, gamma, mean, rstd, dtemporal, partial, B, T, N, D)
    const int nc = (D + 255) / 256;
    if (nc <= 1) AIM_EB(1); else if (nc == 2) AIM_EB(2); else if (nc == 3) AIM_EB(3); else if (nc == 4) AIM_EB(4); else AIM_EB(8);
#undef AIM_EB
    AIM_CHECK_LAUNCH("aim_embed_bwd");
    if (partial) {
        hipLaunchKernelGGL(embed_bwd_finish_kernel, dim3(T, (D + 63) / 64), dim3(256), 0, (hipStream_t)stream, partial, dtemporal,
                           chunks, D);
        AIM_CHECK_LAUNCH("aim_embed_bwd(finish)");
    }
    return 0;
}

extern "C" int64_t aim_embed_bwd_workspace_bytes(int B, int T, int N, int D) {
    int chunks = (B * N + 3) / 4;
    const int want = (2048 + T - 1) / T;
    if (chunks > want) chunks = want;
    return (int64_t)T * chunks * D * 4;
}

extern "C" int aim_frame_sum(const void* x, int x_is_bf16, const float* w, float* out, int frames, int ntok, int D, void* stream) {
    AIM_CHECK_ARG(frames > 0 && ntok > 0 && D > 0 && (D % 4) == 0 && x && out, "frame_sum: bad arguments");
    if (x_is_bf16)
        hipLaunchKernelGGL(frame_sum_kernel<bf16_t>, dim3(frames, (D / 4 + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, w, out, ntok, D);
    else
        hipLaunchKernelGGL(frame_sum_kernel<float>, dim3(frames, (D / 4 + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, w, out, ntok, D);
    AIM_CHECK_LAUNCH("aim_frame_sum");
    return 0;
}

extern "C" int aim_colsum_bf16(const aim_bf16* X, int ldx, const float* af, const float* at, int ntok, float* out,
                               int M, int C, float* workspace, int64_t workspace_bytes, void* stream) {
    AIM_CHECK_ARG(M > 0 && C > 0 && X && out, "colsum: bad arguments");
    if (af || at) AIM_CHECK_ARG(ntok > 0, "colsum: ntok required with row factors");
    if ((C % 8) == 0 && C <= 2048 && (ldx % 8) == 0) {
        int rpb8 = (M + 1023) / 1024;        // <= 1024 row blocks
        if (rpb8 < 64) rpb8 = 64;
        int P = (M + rpb8 - 1) / rpb8;
        // two-stage (partials + finish, fixed summation order) when the caller provides scratch; without scratch a small
        // problem runs as ONE block (the only writer of each column) and a large one falls back to one atomic per
        // column per block
        float* partial = (workspace && workspace_bytes >= (int64_t)P * C * 4 && P > 1) ? workspace : nullptr;
        if (!partial && M <= 2048) {
            rpb8 = M;
            P = 1;
        }
        hipLaunchKernelGGL(colsum8_kernel, dim3(1, P), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)X, ldx, af, at,
                           ntok, out, partial, M, C, rpb8);
        AIM_CHECK_LAUNCH("aim_colsum_bf16");
        if (partial) {
            hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, partial, out, P, C);
            AIM_CHECK_LAUNCH("aim_colsum_bf16(finish)");
        }
        return 0;
    }
    int rpb = (M + 511) / 512;
    if (rpb < 64) rpb = 64;
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 255) / 256, (M + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)X, ldx, af, at, ntok, out, M, C, rpb);
    AIM_CHECK_LAUNCH("aim_colsum_bf16");
    return 0;
}

extern "C" int aim_cast_bf16(const float* src, aim_bf16* dst, int R, int C, int transpose, int ldd, void* stream) {
    AIM_CHECK_ARG(R > 0 && C > 0 && src && dst, "cast: bad arguments");
    AIM_CHECK_ARG(ldd == 0 || ldd >= (transpose ? R : C), "cast: ldd=%d smaller than the destination row", ldd);
    hipStream_t st = (hipStream_t)stream;
    if (transpose) {
        hipLaunchKernelGGL(cast_transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, st, src, (bf16_t*)dst, R, C,
                           ldd ? ldd : R);
    } else if (ldd && ldd != C) {
        const long long n = (long long)R * C;
        hipLaunchKernelGGL(cast_strided_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, (bf16_t*)dst, R, C, ldd);
    } else {
        const long long n = (long long)R * C;
        hipLaunchKernelGGL(cast_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, src, (bf16_t*)dst, n);
    }
    AIM_CHECK_LAUNCH("aim_cast_bf16");
    return 0;
}

extern "C" int aim_cast_multi(const aim_cast_desc* table_dev, int n, void* stream) {
    AIM_CHECK_ARG(table_dev && n > 0, "cast_multi: bad arguments");
    hipLaunchKernelGGL(cast_multi_kernel, dim3(n, 16), dim3(256), 0, (hipStream_t)stream, table_dev);
    AIM_CHECK_LAUNCH("aim_cast_multi");
    return 0;
}

// dst[r * dst_row_stride + c] += src[r * C + c]: the class rows' share of a gradient, produced on the side stream,
// folded into the token-major bf16 tensor on the main one
__global__ __launch_bounds__(256) void add_rows_kernel(bf16_t* __restrict__ dst, long long dst_row_stride,
                                                       const float* __restrict__ src, int R, int C) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (long long)R * C) return;
    const int r = (int)(i / C), c = (int)(i - (long long)r * C);
    bf16x4* d = (bf16x4*)(dst + r * dst_row_stride + c);
    const f32x4 a = *(const f32x4*)(src + i);
    const bf16x4 o = *d;
    *d = pack4((float)o[0] + a[0], (float)o[1] + a[1], (float)o[2] + a[2], (float)o[3] + a[3]);
}

extern "C" int aim_add_rows_bf16(aim_bf16* dst, int64_t dst_row_stride, const float* src, int R, int C, void* stream) {
    AIM_CHECK_ARG(R > 0 && C > 0 && (C % 4) == 0 && (dst_row_stride % 4) == 0 && dst && src, "add_rows: bad arguments");
    const long long n = (long long)R * C / 4;
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)dst,
                       (long long)dst_row_stride, src, R, C);
    AIM_CHECK_LAUNCH("aim_add_rows_bf16");
    return 0;
}

extern "C" int aim_scale_rows(const float* x, const float* s, aim_bf16* y, float* y_f32, int R, int C, void* stream) {
    AIM_CHECK_ARG(R > 0 && C > 0 && x && s && (y || y_f32), "scale_rows: bad arguments");
    const long long n = (long long)R * C;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, s,
                       (bf16_t*)y, y_f32, R, C);
    AIM_CHECK_LAUNCH("aim_scale_rows");
    return 0;
}
