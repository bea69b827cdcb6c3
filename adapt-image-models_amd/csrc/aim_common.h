// Shared device/host helpers for the AIM ViT-CLIP hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// Row-streaming kernels (LayerNorm, attention) walk their rows in DESCENDING order: workgroups are dispatched in ascending
// blockIdx order, the GEMMs before and after them walk their tiles in ascending row order, and a consumer that starts with
// the rows its producer wrote LAST finds more of them still in the memory-side cache (tools/mall_probe.py: ln_fwd over a
// just-written 310 MB tensor 92 -> 84 us).  -DAIM_X_FWDROWS restores the ascending walk (A/B builds only).
#ifdef AIM_X_FWDROWS
#define AIM_REV_BLOCK (blockIdx.x)
#else
#define AIM_REV_BLOCK (gridDim.x - 1 - blockIdx.x)
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define AIM_LDS __attribute__((address_space(3)))
#define AIM_WAVE 64

// ---- error plumbing (host) -------------------------------------------------
void aim_set_error(const char* fmt, ...);
#define AIM_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            aim_set_error(__VA_ARGS__);          \
            return 1;                            \
        }                                        \
    } while (0)
#define AIM_CHECK_LAUNCH(name)                                                    \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            aim_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return 2;                                                             \
        }                                                                         \
    } while (0)

// ---- device helpers --------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }

__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
    bf16x4 r;
    r[0] = (bf16_t)a; r[1] = (bf16_t)b; r[2] = (bf16_t)c; r[3] = (bf16_t)d;
    return r;
}

// four floats -> four OCP fp8 e4m3 bytes (saturating: |x| is clamped to the largest finite e4m3, 448; the inference
// path's activations are cast this way by their producing kernel)
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    a = __builtin_fminf(__builtin_fmaxf(a, -448.f), 448.f);
    b = __builtin_fminf(__builtin_fmaxf(b, -448.f), 448.f);
    c = __builtin_fminf(__builtin_fmaxf(c, -448.f), 448.f);
    d = __builtin_fminf(__builtin_fmaxf(d, -448.f), 448.f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

// QuickGELU x*sigmoid(1.702x) (reference vit_clip.py:80-82) and its derivative
// sigmoid through v_exp_f32 (base 2) + v_rcp_f32: no IEEE division sequence in the GEMM epilogues
__device__ __forceinline__ float sigmoid_1702(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
}
__device__ __forceinline__ float quick_gelu(float x) { return x * sigmoid_1702(x); }
__device__ __forceinline__ float quick_gelu_grad(float x) {
    const float s = sigmoid_1702(x);
    return s * (1.0f + 1.702f * x * (1.0f - s));
}
// two elements at a time: the multiplies and adds become v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (the activation
// arithmetic is ~5 us of a 31 us ACT / DACT GEMM tile; exp2 and rcp stay scalar)
__device__ __forceinline__ f32x2 sigmoid_1702_2(f32x2 x) {
    const f32x2 t = x * (-1.702f * 1.4426950408889634f);
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(t[0]);
    e[1] = __builtin_amdgcn_exp2f(t[1]);
    e += 1.0f;
    f32x2 r;
    r[0] = __builtin_amdgcn_rcpf(e[0]);
    r[1] = __builtin_amdgcn_rcpf(e[1]);
    return r;
}
__device__ __forceinline__ f32x2 quick_gelu2(f32x2 x) { return x * sigmoid_1702_2(x); }
__device__ __forceinline__ f32x2 quick_gelu_grad2(f32x2 x) {
    const f32x2 s = sigmoid_1702_2(x);
    return s * (1.0f + (1.702f * x) * (1.0f - s));
}
// activation and derivative from ONE sigmoid (aim_gemm_args.aux_grad: the forward epilogue stores the derivative instead of the
// pre-activation, so the dgrad epilogue is a multiply): d = s + 1.702 (y - y s) with y = x s
__device__ __forceinline__ void quick_gelu_both2(f32x2 x, f32x2& y, f32x2& d) {
    const f32x2 s = sigmoid_1702_2(x);
    y = x * s;
    d = s + 1.702f * (y - y * s);
}
__device__ __forceinline__ void quick_gelu_both(float x, float& y, float& d) {
    const float s = sigmoid_1702(x);
    y = x * s;
    d = s + 1.702f * (y - y * s);
}
// erf GELU (nn.GELU(), reference vit_clip.py:52) and its derivative.  The normal CDF comes from Abramowitz & Stegun 7.1.26
// (|error of erf| <= 1.5e-7, five FMAs + one rcp + one exp2) instead of libdevice's erff (~70 instructions with both of its
// branches executed): the adapter's column tile of the fused c_fc GEMM spent 18 us (forward) / 24 us (backward) in its
// epilogue against 6 / 8 us for a QuickGELU tile (tools/probe_adapter_tile.py).  1 + erf(y) is formed without cancellation
// (P for y < 0, 2 - P otherwise), and the derivative's Gaussian is the same exp2 value.
__device__ __forceinline__ float gelu_cdf_gauss(float x, float& gauss) {
    const float y = x * 0.70710678118654752f, ay = __builtin_fabsf(y);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ay, 1.0f));
    gauss = __builtin_amdgcn_exp2f(-1.4426950408889634f * ay * ay);                 // exp(-x^2 / 2)
    float P = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    P = __builtin_fmaf(t, P, 1.421413741f);
    P = __builtin_fmaf(t, P, -0.284496736f);
    P = __builtin_fmaf(t, P, 0.254829592f);
    P = P * t * gauss;                                                              // 1 - erf(|y|)
    return 0.5f * (y < 0.f ? P : 2.0f - P);
}
__device__ __forceinline__ void gelu_erf_both(float x, float& y, float& d) {
    float g;
    const float c = gelu_cdf_gauss(x, g);
    y = x * c;
    d = __builtin_fmaf(x * 0.3989422804014327f, g, c);
}
#ifdef AIM_X_ERFF      // A/B builds only: libdevice erff
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
#else
__device__ __forceinline__ float gelu_erf(float x) {
    float g;
    return x * gelu_cdf_gauss(x, g);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float g;
    const float c = gelu_cdf_gauss(x, g);
    return __builtin_fmaf(x * 0.3989422804014327f, g, c);
}
#endif

// LDS image shared by every MFMA operand tile in this library: rows of 64 bf16 (128 B = eight
// 16-byte chunks); chunk c of row r is stored at chunk position c ^ (r & 7).  Conflict-free for
// the 16x16x32 operand ds_read_b128 (lane -> row l&15, chunk 4*ks + (l>>4)) and for
// ds_read_b64_tr_b16 over 8 consecutive rows (derivation in DESIGN.md).
__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 128 + (((chunk ^ row) & 7) << 4); }

// Stage `rows8 x 64` bf16 (8 rows of 128 B) into LDS with ONE buffer_load ... lds per wave:
// lane l writes LDS bytes [16 l, 16 l + 16) of the 1 KiB piece, so it must fetch source chunk
// (l & 7) ^ (l >> 3) of row (l >> 3) -- the swizzle goes on the SOURCE address.  Lanes whose row
// or column is out of range pass an out-of-range voffset: the buffer bounds check returns 0 and
// zero lands in LDS (zero-fill for ragged M / K tails).
#define AIM_OOB 0x80000000u
__device__ __forceinline__ void stage_piece(__amdgpu_buffer_rsrc_t rsrc, AIM_LDS char* lds_piece, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (AIM_LDS void*)lds_piece, 16, voff, 0, 0, 0);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long long bytes) {
    unsigned n = bytes > 0x7fffffffLL ? 0x7fffffffu : (bytes < 0 ? 0u : (unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}

// The same LDS-DMA issued from inline assembly, for kernels that keep a staging ring in flight ACROSS transposing reads or
// reads of a run-time-indexed buffer (wgrad.hip, the pipelined attention backward).  Through the builtin the compiler knows
// that an LDS write is in flight and protects every LDS read it cannot tell apart from it with `s_waitcnt vmcnt(0)`:
// ds_read_b64_tr_b16 is an intrinsic without a memory operand, so EVERY transposing read qualifies.  That drained the whole
// ring once per step, right after the next stage had been issued.  These kernels order their reads behind the DMA themselves
// (counted s_waitcnt vmcnt + workgroup barrier: other waves' pieces need that anyway), so the protection buys nothing.
// Unknown to the compiler, the DMA makes the compiler's own vmcnt waits for ORDINARY loads conservative (the counter is in
// order): keep such loads out of the span where a DMA is in flight, or retire them with __builtin_amdgcn_s_waitcnt, which the
// compiler's scoreboard sees.  The kernel must reach vmcnt(0) before it ends or re-uses the LDS bytes.
// (The GEMMs keep the builtin: their reads are plain ds_read_b128 of compile-time buffers, which the compiler tells apart, and
// the opaque asm costs them 2 % in scheduling freedom.)
typedef __attribute__((ext_vector_type(4))) int aim_rsrc_words;      // raw buffer resource words, held in SGPRs
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // m0 is a reserved register: named in the clobber list all the same
__device__ __forceinline__ void stage_piece_asm(aim_rsrc_words rsrc, AIM_LDS char* lds_piece, unsigned voff) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(uintptr_t)lds_piece);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc) : "memory", "m0");
}
#pragma clang diagnostic pop
// stride 0, num_records = bytes, raw buffer (bounds-checked against num_records: out-of-range lanes read 0)
__device__ __forceinline__ aim_rsrc_words make_rsrc_words(const void* base, long long bytes) {
    const unsigned n = bytes > 0x7fffffffLL ? 0x7fffffffu : (bytes < 0 ? 0u : (unsigned)bytes);
    const unsigned long long a = (unsigned long long)(uintptr_t)base;
    aim_rsrc_words r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)n);
    r[3] = 0x00020000;
    return r;
}

__device__ __forceinline__ bf16x8 lds_read8(const AIM_LDS char* p) { return *(const AIM_LDS bf16x8*)p; }
__device__ __forceinline__ bf16x4 lds_read_tr4(const AIM_LDS char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((AIM_LDS bf16x4*)p);
}

// x op x[lane ^ 16] and x op x[lane ^ 32] without the LDS crossbar (ds_bpermute): v_permlane16_swap / v_permlane32_swap on
// two copies of x leave (own, partner) in the two registers on every lane, in an order that a commutative op ignores.
__device__ __forceinline__ void lane_pair16(float x, float& a, float& b) {
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lane_pair32(float x, float& a, float& b) {
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// reductions over the four lanes {l, l^16, l^32, l^48}
__device__ __forceinline__ float quad_max(float x) {
    float a, b;
    lane_pair16(x, a, b);
    x = fmaxf(a, b);
    lane_pair32(x, a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float quad_sum(float x) {
    float a, b;
    lane_pair16(x, a, b);
    x = a + b;
    lane_pair32(x, a, b);
    return a + b;
}

// Pair two packed 4 x bf16 groups (8 bytes each) that a lane holds for MFMA tiles dt and dt+1 into ONE 16-byte store:
// v_permlane16_swap exchanges the odd 16-lane rows of the first operand with the even rows of the second, so an even-row
// lane ends with 8 consecutive elements of tile dt (its own 4 + its odd neighbour's), an odd-row lane with 8 of tile dt+1.
// Inline asm: the clang builtin for this instruction is miscompiled by this hipcc (tools/swap_probe.hip); `s_nop 1` covers
// the VALU-write -> permlane-read hazard.  Returns the 16 bytes; the caller offsets the address by row parity.
__device__ __forceinline__ bf16x8 pair_rows16(bf16x4 a, bf16x4 b) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
    u32x2_t ua = __builtin_bit_cast(u32x2_t, a), ub = __builtin_bit_cast(u32x2_t, b);
    unsigned a0 = ua[0], a1 = ua[1], b0 = ub[0], b1 = ub[1];
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    return __builtin_bit_cast(bf16x8, u32x4_t{a0, a1, b0, b1});
}

// XCD-aware bijective remap of a linear block id (8 XCDs, blocks dealt round-robin): blocks that
// share an XCD get a contiguous range of logical ids (cdna guide T1, bijective form).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
