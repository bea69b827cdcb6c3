// Temporal attention over the T frames of EVERY token position (stock-AIM block), forward and backward.  gfx950 only.
//
// Replaces reference mmaction/models/backbones/vitclip_aim.py:199-205: attention() (:148-187) applied to
// rearrange(x, 'n (b t) d -> t (b n) d'): sequence = the T frames, batch = B * N (every token position of every clip).
// (vit_clip.py's own block runs this attention on the class tokens only: cls_attn.hip.)
// Sequence length is T <= 32 and the batch is huge (B * N * H = 151 296 problems at 64 clips), so this is a
// bandwidth-bound kernel: one wave per (clip, token, head), lane = head-dim element, 4 problems per 256-thread workgroup;
// every global access of a wave is one whole 128-byte row segment of the frame-major fused qkv buffer
// (row (b*T + t)*N + n), so no transposed copy of the activations is ever made (the reference rearranges x twice).
#include "aim_common.h"
#include "aim_kernels_internal.h"

namespace {

constexpr int TMAX = 32;

// LDS per wave: q, k, v [T][65] f32 + p [T][T+1]
__device__ __forceinline__ int tattn_lds_floats(int T) { return 3 * T * 65 + T * (T + 1); }

__global__ __launch_bounds__(256) void tattn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                        float* __restrict__ probs, int B, int T, int N, int H) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long prob = (long long)blockIdx.x * (blockDim.x >> 6) + wave;          // (b, n, h)
    const long long nprob = (long long)B * N * H;
    if (prob >= nprob) return;          // (whole waves only: no barrier below spans waves)
    const int h = (int)(prob % H);
    const long long bn = prob / H;
    const int n = (int)(bn % N), b = (int)(bn / N);
    const int D = H * 64, ld = 3 * D;
    float* sq = smem + wave * tattn_lds_floats(T);
    float* sk = sq + T * 65;
    float* sv = sk + T * 65;
    float* sp = sv + T * 65;
    for (int t = 0; t < T; ++t) {
        const bf16_t* r = qkv + ((long long)(b * T + t) * N + n) * ld + h * 64 + lane;
        sq[t * 65 + lane] = (float)r[0];
        sk[t * 65 + lane] = (float)r[D];
        sv[t * 65 + lane] = (float)r[2 * D];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): this wave's LDS writes are done
    for (int p = lane; p < T * T; p += 64) {
        const int tq = p / T, tk = p - tq * T;
        float acc = 0.f;
#pragma unroll 8
        for (int d = 0; d < 64; ++d) acc += sq[tq * 65 + d] * sk[tk * 65 + d];
        sp[tq * (T + 1) + tk] = acc * 0.125f;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int tq = lane; tq < T; tq += 64) {
        float mx = -INFINITY;
        for (int tk = 0; tk < T; ++tk) mx = fmaxf(mx, sp[tq * (T + 1) + tk]);
        float sum = 0.f;
        for (int tk = 0; tk < T; ++tk) {
            const float e = expf(sp[tq * (T + 1) + tk] - mx);
            sp[tq * (T + 1) + tk] = e;
            sum += e;
        }
        const float inv = 1.0f / sum;
        for (int tk = 0; tk < T; ++tk) {
            const float p = sp[tq * (T + 1) + tk] * inv;
            sp[tq * (T + 1) + tk] = p;
            probs[(prob * T + tq) * T + tk] = p;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int tq = 0; tq < T; ++tq) {
        float acc = 0.f;
        for (int tk = 0; tk < T; ++tk) acc += sp[tq * (T + 1) + tk] * sv[tk * 65 + lane];
        out[((long long)(b * T + tq) * N + n) * D + h * 64 + lane] = (bf16_t)acc;
    }
}

// LDS per wave: q, k, v, dO [T][65] + p, dS [T][T+1]
__device__ __forceinline__ int tattn_bwd_lds_floats(int T) { return 4 * T * 65 + 2 * T * (T + 1); }

__global__ __launch_bounds__(256) void tattn_bwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ probs,
                                                        const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv, int B,
                                                        int T, int N, int H) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long prob = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
    const long long nprob = (long long)B * N * H;
    if (prob >= nprob) return;
    const int h = (int)(prob % H);
    const long long bn = prob / H;
    const int n = (int)(bn % N), b = (int)(bn / N);
    const int D = H * 64, ld = 3 * D;
    float* sq = smem + wave * tattn_bwd_lds_floats(T);
    float* sk = sq + T * 65;
    float* sv = sk + T * 65;
    float* sdo = sv + T * 65;
    float* sp = sdo + T * 65;
    float* sds = sp + T * (T + 1);
    for (int t = 0; t < T; ++t) {
        const long long row = (long long)(b * T + t) * N + n;
        const bf16_t* r = qkv + row * ld + h * 64 + lane;
        sq[t * 65 + lane] = (float)r[0];
        sk[t * 65 + lane] = (float)r[D];
        sv[t * 65 + lane] = (float)r[2 * D];
        sdo[t * 65 + lane] = (float)dout[row * D + h * 64 + lane];
    }
    for (int p = lane; p < T * T; p += 64) {
        const int tq = p / T, tk = p - tq * T;
        sp[tq * (T + 1) + tk] = probs[(prob * T + tq) * T + tk];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int p = lane; p < T * T; p += 64) {          // dP[tq][tk] = dO[tq] . V[tk]
        const int tq = p / T, tk = p - tq * T;
        float acc = 0.f;
#pragma unroll 8
        for (int d = 0; d < 64; ++d) acc += sdo[tq * 65 + d] * sv[tk * 65 + d];
        sds[tq * (T + 1) + tk] = acc;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int tq = lane; tq < T; tq += 64) {
        float dot = 0.f;
        for (int tk = 0; tk < T; ++tk) dot += sp[tq * (T + 1) + tk] * sds[tq * (T + 1) + tk];
        for (int tk = 0; tk < T; ++tk) sds[tq * (T + 1) + tk] = sp[tq * (T + 1) + tk] * (sds[tq * (T + 1) + tk] - dot) * 0.125f;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int t = 0; t < T; ++t) {
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int u = 0; u < T; ++u) {
            dq += sds[t * (T + 1) + u] * sk[u * 65 + lane];
            dk += sds[u * (T + 1) + t] * sq[u * 65 + lane];
            dv += sp[u * (T + 1) + t] * sdo[u * 65 + lane];
        }
        bf16_t* r = dqkv + ((long long)(b * T + t) * N + n) * ld + h * 64 + lane;
        r[0] = (bf16_t)dq;
        r[D] = (bf16_t)dk;
        r[2 * D] = (bf16_t)dv;
    }
}

// out[r][c] = a[r][c] + b[r][c]  (bf16, row strides in elements; 8 elements per thread)
__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16_t* __restrict__ a, long long lda, const bf16_t* __restrict__ b,
                                                       long long ldb, bf16_t* __restrict__ out, long long ldo, int R, int C8) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)R * C8) return;
    const int r = (int)(i / C8), c = (int)(i - (long long)r * C8) * 8;
    const bf16x8 x = *(const bf16x8*)(a + r * lda + c), y = *(const bf16x8*)(b + r * ldb + c);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)x[e] + (float)y[e]);
    *(bf16x8*)(out + r * ldo + c) = o;
}

// x[r][c] += s[r][c]   (x f32 dense [R, C], s bf16 with row stride lds; 4 elements per thread)
__global__ __launch_bounds__(256) void acc_bf16_kernel(float* __restrict__ x, const bf16_t* __restrict__ s, long long lds, int R, int C4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)R * C4) return;
    const int r = (int)(i / C4), c = (int)(i - (long long)r * C4) * 4;
    f32x4* xp = (f32x4*)(x + (long long)r * C4 * 4 + c);
    const bf16x4 v = *(const bf16x4*)(s + r * lds + c);
    f32x4 o = *xp;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] += (float)v[e];
    *xp = o;
}

}  // namespace

extern "C" int aim_tattn_fwd(const aim_bf16* qkv, aim_bf16* out, float* probs, int B, int T, int N, int H, void* stream) {
    AIM_CHECK_ARG(B > 0 && T > 0 && T <= TMAX && N > 0 && H > 0, "tattn_fwd: unsupported shape B=%d T=%d (T <= 32)", B, T);
    AIM_CHECK_ARG(qkv && out && probs, "tattn_fwd: null pointer");
    const long long nprob = (long long)B * N * H;
    const int wpb = T <= 16 ? 4 : 2;                // waves (problems) per workgroup: LDS per problem grows with T
    const size_t lds = (size_t)wpb * (3 * T * 65 + T * (T + 1)) * 4;
    static bool attr_set_f = false;
    if (!attr_set_f) {
        (void)hipFuncSetAttribute((const void*)tattn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set_f = true;
    }
    hipLaunchKernelGGL(tattn_fwd_kernel, dim3((unsigned)((nprob + wpb - 1) / wpb)), dim3(64 * wpb), lds, (hipStream_t)stream, (const bf16_t*)qkv,
                       (bf16_t*)out, probs, B, T, N, H);
    AIM_CHECK_LAUNCH("aim_tattn_fwd");
    return 0;
}

extern "C" int aim_tattn_bwd(const aim_bf16* qkv, const float* probs, const aim_bf16* dout, aim_bf16* dqkv, int B, int T, int N,
                             int H, void* stream) {
    AIM_CHECK_ARG(B > 0 && T > 0 && T <= TMAX && N > 0 && H > 0, "tattn_bwd: unsupported shape B=%d T=%d (T <= 32)", B, T);
    AIM_CHECK_ARG(qkv && probs && dout && dqkv, "tattn_bwd: null pointer");
    const long long nprob = (long long)B * N * H;
    const int wpb = T <= 16 ? 4 : 2;
    const size_t lds = (size_t)wpb * (4 * T * 65 + 2 * T * (T + 1)) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)tattn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(tattn_bwd_kernel, dim3((unsigned)((nprob + wpb - 1) / wpb)), dim3(64 * wpb), lds, (hipStream_t)stream, (const bf16_t*)qkv,
                       probs, (const bf16_t*)dout, (bf16_t*)dqkv, B, T, N, H);
    AIM_CHECK_LAUNCH("aim_tattn_bwd");
    return 0;
}

extern "C" int aim_add_bf16(const aim_bf16* a, int64_t lda, const aim_bf16* b, int64_t ldb, aim_bf16* out, int64_t ldo, int R, int C,
                            void* stream) {
    AIM_CHECK_ARG(a && b && out && R > 0 && C > 0 && (C % 8) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && (ldo % 8) == 0, "add_bf16: bad arguments");
    const long long n = (long long)R * (C / 8);
    hipLaunchKernelGGL(add_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                       (long long)lda, (const bf16_t*)b, (long long)ldb, (bf16_t*)out, (long long)ldo, R, C / 8);
    AIM_CHECK_LAUNCH("aim_add_bf16");
    return 0;
}

extern "C" int aim_acc_bf16(float* x, const aim_bf16* s, int64_t lds, int R, int C, void* stream) {
    AIM_CHECK_ARG(x && s && R > 0 && C > 0 && (C % 4) == 0 && (lds % 4) == 0, "acc_bf16: bad arguments");
    const long long n = (long long)R * (C / 4);
    hipLaunchKernelGGL(acc_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)s,
                       (long long)lds, R, C / 4);
    AIM_CHECK_LAUNCH("aim_acc_bf16");
    return 0;
}
