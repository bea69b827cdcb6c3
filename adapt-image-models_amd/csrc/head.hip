// Classification tail right after the backbone: I3DHead (avg-pool over frames, dropout, fc_cls), hard-label
// cross-entropy and top-k accuracy, forward and backward, on the device.  gfx950 only.
//
// Replaces: mmaction/models/heads/i3d_head.py:53-73 (AdaptiveAvgPool3d -> Dropout -> Linear),
// mmaction/models/losses/cross_entropy_loss.py:78 (F.cross_entropy, mean over the batch) and
// mmaction/models/heads/base.py:90-95 (top_k_accuracy on .cpu().numpy() -- a host sync per iteration in the reference).
// The work is tiny (B x D x C = 64 x 768 x 400: 39 MFLOP): these kernels exist to take ~15 eager launches, one
// vendor-library GEMM and the host round trip out of the step, not to reach a roofline.  fp32 throughout.
#include "aim_common.h"
#include "aim_kernels_internal.h"

namespace {

// workgroup (b, ct): the sample's pooled row into LDS (T*D floats: cheap enough to redo per class tile), then 32 classes,
// a wave per class with coalesced 256-B reads of the weight row (L2-resident)
constexpr int HEAD_CT = 32;
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ drop,
                                                       const float* __restrict__ W, const float* __restrict__ bias,
                                                       float* __restrict__ pooled, float* __restrict__ score, int T,
                                                       int D, int C) {
    extern __shared__ float sx[];                 // [D]
    const int b = blockIdx.x, c0 = blockIdx.y * HEAD_CT, tid = threadIdx.x;
    const float invT = 1.0f / (float)T;
    for (int d = tid; d < D; d += 256) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += feat[((long long)b * T + t) * D + d];
        s *= invT;
        if (drop) s *= drop[(long long)b * D + d];
        sx[d] = s;
        if (blockIdx.y == 0) pooled[(long long)b * D + d] = s;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int c = c0 + wave; c < min(c0 + HEAD_CT, C); c += 4) {
        const float* w = W + (long long)c * D;
        float acc = 0.f;
        for (int d = lane; d < D; d += 64) acc += w[d] * sx[d];
        acc = wave_sum(acc);
        if (lane == 0) score[(long long)b * C + c] = acc + (bias ? bias[c] : 0.f);
    }
}

// per sample: softmax, loss term, rank of the label (numpy argsort tie order), dscore row.
// A label outside [0, C) is IGNORED the way F.cross_entropy ignores ignore_index = -100 (the call this replaces,
// cross_entropy_loss.py:78): no loss term, a zero dscore row, and the mean runs over the valid samples only (torch
// device-asserts on other out-of-range labels; here they are ignored as well instead of read out of bounds).  The two
// accuracy columns keep the reference's denominator B (heads/base.py:90-95 counts such a sample as a miss).
__global__ __launch_bounds__(256) void ce_topk_kernel(const float* __restrict__ score, const long long* __restrict__ label,
                                                      float* __restrict__ dscore, float* __restrict__ per_sample, int B,
                                                      int C, int k2) {
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* s = score + (long long)b * C;
    const long long lab64 = label[b];
    const bool valid = lab64 >= 0 && lab64 < (long long)C;
    const int lab = valid ? (int)lab64 : 0;
    // valid samples of the batch (every workgroup counts them itself: B is a few dozen labels)
    float nv = 0.f;
    for (int i = tid; i < B; i += 256) {
        const long long l = label[i];
        nv += (l >= 0 && l < (long long)C) ? 1.f : 0.f;
    }
    nv = wave_sum(nv);
    if (lane == 0) red[4 + wave] = nv;
    float mx = -INFINITY;
    for (int c = tid; c < C; c += 256) mx = fmaxf(mx, s[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    nv = red[4] + red[5] + red[6] + red[7];
    __syncthreads();
    const float tgt = s[lab];
    float sum = 0.f, ahead = 0.f;
    for (int c = tid; c < C; c += 256) {
        const float v = s[c];
        sum += __expf(v - mx);
        ahead += (v > tgt || (v == tgt && c > lab)) ? 1.f : 0.f;
    }
    sum = wave_sum(sum);
    ahead = wave_sum(ahead);
    if (lane == 0) { red[wave] = sum; red[4 + wave] = ahead; }
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    ahead = red[4] + red[5] + red[6] + red[7];
    const float lse = mx + __logf(sum), inv = valid ? 1.0f / nv : 0.f;
    if (dscore)
        for (int c = tid; c < C; c += 256)
            dscore[(long long)b * C + c] = valid ? (__expf(s[c] - lse) - (c == lab ? 1.f : 0.f)) * inv : 0.f;
    if (tid == 0) {
        per_sample[b * 4 + 0] = valid ? lse - tgt : 0.f;
        per_sample[b * 4 + 1] = (valid && ahead < 1.f) ? 1.f : 0.f;
        per_sample[b * 4 + 2] = (valid && ahead < (float)(k2 < C ? k2 : C)) ? 1.f : 0.f;
        per_sample[b * 4 + 3] = valid ? 1.f : 0.f;
    }
}

// ordered sum over the samples (bitwise reproducible): out3 = [CE mean over the valid samples, top-1, top-k2 over B]
__global__ __launch_bounds__(64) void ce_finish_kernel(const float* __restrict__ per_sample, float* __restrict__ out3, int B) {
    const int j = threadIdx.x;
    if (j < 3) {
        float s = 0.f, n = 0.f;
        for (int b = 0; b < B; ++b) { s += per_sample[b * 4 + j]; n += per_sample[b * 4 + 3]; }
        out3[j] = s / (j == 0 ? n : (float)B);
    }
}

// dW[c][d] += sum_b dscore[b][c] pooled[b][d]; db[c] += sum_b dscore[b][c]   (grid: C blocks; fixed summation order)
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ dscore, const float* __restrict__ pooled,
                                                         float* __restrict__ dW, float* __restrict__ db, int B, int D, int C) {
    const int c = blockIdx.x, tid = threadIdx.x;
    for (int d = tid; d < D; d += 256) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += dscore[(long long)b * C + c] * pooled[(long long)b * D + d];
        dW[(long long)c * D + d] += acc;
    }
    if (tid == 0 && db) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += dscore[(long long)b * C + c];
        db[c] += acc;
    }
}

// dfeat[b][t][d] = drop[b][d] / T * sum_c dscore[b][c] W[c][d]     (grid: B x ceil(D / 64); 4 class groups x 64 columns,
// partial sums combined through LDS in a fixed order)
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ dscore, const float* __restrict__ drop,
                                                         const float* __restrict__ W, float* __restrict__ dfeat, int T,
                                                         int D, int C) {
    extern __shared__ float sg[];                 // [C] + [4][64]
    float* part = sg + C;
    const int b = blockIdx.x, tid = threadIdx.x, dl = tid & 63, grp = tid >> 6;
    const int d = blockIdx.y * 64 + dl;
    for (int c = tid; c < C; c += 256) sg[c] = dscore[(long long)b * C + c];
    __syncthreads();
    float acc = 0.f;
    if (d < D)
        for (int c = grp; c < C; c += 4) acc += sg[c] * W[(long long)c * D + d];     // 256-B coalesced per wave
    part[grp * 64 + dl] = acc;
    __syncthreads();
    if (grp == 0 && d < D) {
        float v = ((part[dl] + part[64 + dl]) + (part[128 + dl] + part[192 + dl])) / (float)T;
        if (drop) v *= drop[(long long)b * D + d];
        for (int t = 0; t < T; ++t) dfeat[((long long)b * T + t) * D + d] = v;
    }
}

}  // namespace

extern "C" int aim_head_fwd(const float* feat, const float* drop, const float* W, const float* bias, float* pooled,
                            float* score, int B, int T, int D, int C, void* stream) {
    AIM_CHECK_ARG(feat && W && pooled && score && B > 0 && T > 0 && D > 0 && C > 0, "head_fwd: bad arguments");
    AIM_CHECK_ARG((size_t)D * 4 <= 64 * 1024, "head_fwd: D=%d too large", D);
    hipLaunchKernelGGL(head_fwd_kernel, dim3(B, (C + HEAD_CT - 1) / HEAD_CT), dim3(256), (size_t)D * 4, (hipStream_t)stream, feat,
                       drop, W, bias, pooled, score, T, D, C);
    AIM_CHECK_LAUNCH("aim_head_fwd");
    return 0;
}

extern "C" int aim_head_bwd(const float* dscore, const float* pooled, const float* drop, const float* W, float* dW,
                            float* db, float* dfeat, int B, int T, int D, int C, void* stream) {
    AIM_CHECK_ARG(dscore && pooled && W && B > 0 && T > 0 && D > 0 && C > 0, "head_bwd: bad arguments");
    AIM_CHECK_ARG((size_t)(C + 256) * 4 <= 64 * 1024, "head_bwd: C=%d too large", C);
    hipStream_t st = (hipStream_t)stream;
    if (dW) {
        hipLaunchKernelGGL(head_wgrad_kernel, dim3(C), dim3(256), 0, st, dscore, pooled, dW, db, B, D, C);
        AIM_CHECK_LAUNCH("aim_head_bwd(wgrad)");
    }
    if (dfeat) {
        hipLaunchKernelGGL(head_dgrad_kernel, dim3(B, (D + 63) / 64), dim3(256), (size_t)(C + 256) * 4, st, dscore, drop, W, dfeat, T, D, C);
        AIM_CHECK_LAUNCH("aim_head_bwd(dgrad)");
    }
    return 0;
}

extern "C" int aim_ce_topk(const float* score, const int64_t* label, float* dscore, float* per_sample, float* out3, int B,
                           int C, int k2, void* stream) {
    AIM_CHECK_ARG(score && label && per_sample && out3 && B > 0 && C > 0 && k2 >= 1, "ce_topk: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_topk_kernel, dim3(B), dim3(256), 0, st, score, (const long long*)label, dscore, per_sample, B, C, k2);
    AIM_CHECK_LAUNCH("aim_ce_topk");
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(64), 0, st, per_sample, out3, B);
    AIM_CHECK_LAUNCH("aim_ce_topk(finish)");
    return 0;
}
