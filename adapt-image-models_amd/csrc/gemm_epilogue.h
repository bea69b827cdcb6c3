// Fused GEMM epilogues shared by the 128x128 and 256x256 kernels.  A lane owns, per 16x16 MFMA tile,
// row m and the 4 consecutive columns n..n+3 (the MFMA is issued with the weight fragment first).
#pragma once
#include "aim_common.h"
#include "aim_kernels_internal.h"
#include <type_traits>

struct RowFactors {
    float rs, vs;
    int frame;
};

// column-split epilogues (fused [frozen MLP | adapter] GEMMs): activation kind and row factor for column n
__device__ __forceinline__ int col_act(const GemmArgs& g, int n) { return (g.n_split > 0 && n >= g.n_split) ? g.act2 : g.act; }
__device__ __forceinline__ float col_rs(const GemmArgs& g, int n, float rs) {
    return (g.n_split > 0 && n < g.n_split) ? 1.0f : rs;
}

__device__ __forceinline__ RowFactors row_factors(const GemmArgs& g, int m) {
    RowFactors r{1.0f, 0.0f, 0};
    if (g.af || g.at || g.vec) {
        const int mg = m + g.row0;              // row of the caller's whole problem (aim_gemm_args.row0: a peeled tail launch)
        r.frame = mg / g.ntok;
        const int tok = mg - r.frame * g.ntok;
        if (g.af) r.rs *= g.af[r.frame];
        if (g.at) r.rs *= g.at[tok];
        if (g.vec) r.vs = g.bt ? g.bt[tok] : 1.0f;
    }
    return r;
}

// Epilogue inputs that come from memory, fetched ahead of the arithmetic so that a wave keeps many
// loads in flight (resid may alias out, so the compiler cannot hoist these loads across stores itself).
struct FragIn {
    f32x4 resid;
    bf16x4 aux;
};

template <int EPI>
__device__ __forceinline__ FragIn load_frag_in(const GemmArgs& g, int m, int n) {
    FragIn f;
    f.resid = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == EPI_F32) {
        if (g.resid) f.resid = *(const f32x4*)(g.resid + (long long)m * g.ldr + n);
    }
    if constexpr (EPI == EPI_DACT) f.aux = *(const bf16x4*)((const bf16_t*)g.aux + (long long)m * g.ldaux + n);
    return f;
}

template <int EPI>
__device__ __forceinline__ void store_frag(const GemmArgs& g, f32x4 v, int m, int n, const RowFactors& rf,
                                           const FragIn& fin) {
    const float rs = (EPI == EPI_ACT || EPI == EPI_DACT) ? col_rs(g, n, rf.rs) : rf.rs;
    const int act = col_act(g, n);
    if constexpr (EPI == EPI_F32) {
        // the SAME expression as the 256 x 256 kernel's residual epilogue (wave_epilogue: v * rs + rs * bias with row factors,
        // v + bias without), so that the rows a peeled tail launch computes are bit-identical to the whole launch's
        // (spelled out: t = v * rs; v = fma(rs, b, t) -- the contraction the compiler picks there)
        const f32x4 b = g.bias ? *(const f32x4*)(g.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (g.af || g.at || g.vec) {
            const float ms = g.rs_bias_only ? 1.0f : rs;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rs, b[e], v[e] * ms);
        } else {
            v += b;
        }
    } else if (g.bias) {
        v += *(const f32x4*)(g.bias + n);
    }
    if constexpr (EPI == EPI_BF16) {
        v *= rs;
        *(bf16x4*)((bf16_t*)g.out + (long long)m * g.ldo + n) = pack4(v[0], v[1], v[2], v[3]);
    } else if constexpr (EPI == EPI_ACT) {
        const bf16x4 pre = pack4(v[0], v[1], v[2], v[3]);
        float y[4], d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (act == ACT_QGELU) quick_gelu_both((float)pre[e], y[e], d[e]);
            else gelu_erf_both((float)pre[e], y[e], d[e]);
        }
        // out2: the pre-activation, or (aux_grad) the activation's derivative at it
        *(bf16x4*)((bf16_t*)g.out2 + (long long)m * g.ldo2 + n) = g.aux_grad ? pack4(d[0], d[1], d[2], d[3]) : pre;
        *(bf16x4*)((bf16_t*)g.out + (long long)m * g.ldo + n) = pack4(rs * y[0], rs * y[1], rs * y[2], rs * y[3]);
    } else if constexpr (EPI == EPI_DACT) {
        const bf16x4 pre = fin.aux;
        float d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            d[e] = g.aux_grad ? (float)pre[e] : (act == ACT_QGELU) ? quick_gelu_grad((float)pre[e]) : gelu_erf_grad((float)pre[e]);
        *(bf16x4*)((bf16_t*)g.out + (long long)m * g.ldo + n) =
            pack4(rs * v[0] * d[0], rs * v[1] * d[1], rs * v[2] * d[2], rs * v[3] * d[3]);
    } else if constexpr (EPI == EPI_F32) {
        if (g.vec) {
            const f32x4 w = *(const f32x4*)(g.vec + (long long)rf.frame * g.ldv + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(w[e], rf.vs, v[e]);
        }
        v += fin.resid;
        *(f32x4*)((float*)g.out + (long long)m * g.ldo + n) = v;
    }
}

template <int EPI>
__device__ __forceinline__ void store_frag(const GemmArgs& g, f32x4 v, int m, int n, const RowFactors& rf) {
    store_frag<EPI>(g, v, m, n, rf, load_frag_in<EPI>(g, m, n));
}

// 8-column form for the bf16-output epilogues: the lane owns columns n..n+7 of row m, so each output
// row segment leaves as ONE 16-byte store (the 8-byte form is store-issue-bound).
template <int EPI>
__device__ __forceinline__ void store_frag8(const GemmArgs& g, f32x4 v0, f32x4 v1, int m, int n, const RowFactors& rf,
                                            const bf16x8& aux8) {
    static_assert(EPI == EPI_BF16 || EPI == EPI_ACT || EPI == EPI_DACT, "bf16-output epilogues only");
    const float rs = (EPI == EPI_BF16) ? rf.rs : col_rs(g, n, rf.rs);
    const int act = col_act(g, n);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    if (g.bias) {
        const f32x4 b0 = *(const f32x4*)(g.bias + n), b1 = *(const f32x4*)(g.bias + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] += b0[e];
            v[4 + e] += b1[e];
        }
    }
    bf16x8 o;
    if constexpr (EPI == EPI_BF16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(rs * v[e]);
    } else if constexpr (EPI == EPI_ACT) {
        bf16x8 pre;
#pragma unroll
        for (int e = 0; e < 8; ++e) pre[e] = (bf16_t)v[e];
        bf16x8 dv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y, d;
            if (act == ACT_QGELU) quick_gelu_both((float)pre[e], y, d);
            else gelu_erf_both((float)pre[e], y, d);
            o[e] = (bf16_t)(rs * y);
            dv[e] = (bf16_t)d;
        }
        *(bf16x8*)((bf16_t*)g.out2 + (long long)m * g.ldo2 + n) = g.aux_grad ? dv : pre;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float p = (float)aux8[e];
            o[e] = (bf16_t)(rs * v[e] * (g.aux_grad ? p : act == ACT_QGELU ? quick_gelu_grad(p) : gelu_erf_grad(p)));
        }
    }
    *(bf16x8*)((bf16_t*)g.out + (long long)m * g.ldo + n) = o;
}

// ------------------------------------------------------------------------------------------------
// Epilogue of one wave's 128x64 accumulator tile (acc[8][4], 16x16 MFMA fragments), re-tiled through a
// private LDS scratch (8 rows x 272 B) so global accesses are whole 128-256 B row segments.
//
// EVERY global access here is a buffer instruction: lanes past N carry an out-of-range offset and rows
// past M fall off the end of the resource (loads return 0, stores are dropped), so there is no
// exec-masked branch around any VMEM instruction.  That matters because vmcnt is in-order and counts
// stores -- with a load under a divergent branch the compiler can only merge control flow with
// `s_waitcnt vmcnt(0)`, which makes each 8-row sub-pass wait for the previous sub-pass's store
// acknowledgement and for the next tile's staged operand loads (measured: ~11 us of a ~31 us tile).
// Per-row factors (af/at/bt) are computed once per tile into a wave-private LDS table, so the
// sub-passes themselves contain no load besides the prefetched resid / aux rows.
// Null optional inputs (bias, resid, vec, af, at, bt) get a zero-length resource and read as 0.
//
// Measured and rejected: re-tiling in registers (v_permlane32_swap + v_permlane16_swap give a lane 8 consecutive
// columns, no LDS round trip).  Its stores cover 16 rows x 64 B instead of 8 rows x 128 B (fp32: 4 rows x 256 B
// here), i.e. half cache lines: BF16 epilogue 3.9 -> 3.4 us but F32 12 -> 17 us, DACT 11 -> 13 us and a longer
// store drain in the next tile's K-loop; whole step 1012 -> 962 clips/s.  Full-line row segments win.
// ------------------------------------------------------------------------------------------------
constexpr int EPI_RS = 272;                      // scratch row stride: 64 f32 + 16 B pad (8-row fp32 sub-passes)
constexpr int EPI_RSH = 144;                     // scratch row stride: 64 bf16 + 16 B pad (16-row bf16 sub-passes)
constexpr int EPI_ROWFAC = 16 * EPI_RSH;         // offset of the per-row factor table: 128 rows x {rs, vs}  (>= 8 * EPI_RS)
constexpr int EPI_VEC = EPI_ROWFAC + 128 * 8;    // offset of the two candidate `vec` rows of the wave tile (RES16): 2 x 64 f32
constexpr int EPI_SCRATCH = EPI_VEC + 2 * 64 * 4;   // bytes per wave (fits beside the K-loop images)
static_assert(16 * EPI_RSH >= 8 * EPI_RS, "the fp32 sub-pass scratch must fit in front of the row-factor table");

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#ifndef AIM_STORE_POLICY
#define AIM_STORE_POLICY 2
#endif
#ifndef AIM_ACT8_MFMA_LAYOUT
#define AIM_ACT8_MFMA_LAYOUT 1       // 0: the fp8-output epilogue through the fp32 scratch (A/B builds)
#endif
#ifndef AIM_LOAD_POLICY
#define AIM_LOAD_POLICY 0
#endif

// resource over `rows` rows of `ld_bytes` starting at base + byte_off; null base or rows <= 0 -> empty
__device__ __forceinline__ __amdgpu_buffer_rsrc_t epi_rsrc(const void* base, long long byte_off, long long bytes) {
    const unsigned n = (!base || bytes <= 0) ? 0u : (bytes > 0x7fffffffLL ? 0x7fffffffu : (unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>((const char*)base + byte_off), 0, n, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load_f4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
// read-once streams (residual rows, saved pre-activations)
__device__ __forceinline__ f32x4 buf_stream_f4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, AIM_LOAD_POLICY));
}
__device__ __forceinline__ bf16x8 buf_load_h8(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, AIM_LOAD_POLICY));
}
__device__ __forceinline__ float buf_load_f1(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
template <typename V>
__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t r, unsigned voff, V v) {
#ifdef AIM_X_NOSTORE
    asm volatile("" ::"v"(__builtin_bit_cast(u32x4, v)), "v"(voff));
#else
    // nt (non-temporal, cache-policy bit 1): the output tile is not re-read by this kernel.  Measured: the store drain
    // that the next tile's first vmcnt wait is exposed to shrinks from ~3 us to ~0 (BF16 N=2304 K-loop 21.2 -> 18.2 us,
    // ACT 23.0 -> 19.5 us per tile; F32 / DACT unchanged); sc1 alone gives about half of that.
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, AIM_STORE_POLICY);
#endif
}
// a running byte offset the compiler may not pre-compute for all 16 sub-passes (that costs 16 VGPRs and spills)
__device__ __forceinline__ void epi_advance(unsigned& voff, unsigned step) {
    voff += step;
    asm volatile("" : "+v"(voff));
}

// WS: apply the per-output-channel dequantisation scale g.wscale (fp8 kernels only: compile-time, so the bf16 kernels
// carry no extra registers)
template <int EPI, bool WS = false>
__device__ __forceinline__ void wave_epilogue(const GemmArgs& g, f32x4 (&acc)[8][4], AIM_LDS char* scr, int m_base_in,
                                              int n_base_in, int lane) {
    const int m_base = __builtin_amdgcn_readfirstlane(m_base_in), n_base = __builtin_amdgcn_readfirstlane(n_base_in);
    // The lane id is laundered so that nothing derived from it here (scratch addresses, row / column offsets, masks) is
    // loop-invariant for the persistent tile loop: hoisted above the K-loop, which runs at the 256-VGPR limit, those values
    // are spilled, and a scratch reload waits on vmcnt behind the next tile's LDS-DMA.
    asm volatile("" : "+v"(lane));
    const int frow = lane & 15, fq = lane >> 4;
    // half of one 16-row MFMA tile -> scratch rows 0..7 (the lanes holding the other half sit out)
    auto dump8 = [&](int mt, int half) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // previous sub-pass's scratch reads are done (WAR)
        if ((frow >> 3) == half) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(AIM_LDS f32x4*)(scr + (frow & 7) * EPI_RS + (j * 16 + fq * 4) * 4) = acc[mt][j];
        }
#ifdef AIM_X_EPIWAIT
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        // No wait here: a wave's DS instructions execute in issue order, so the reads below see these writes.  The
        // statement only pins the COMPILER's order (it once hoisted the reads above the exec-masked writes).
        asm volatile("" ::: "memory");
#endif
    };
    const bool rowf = g.af || g.at || g.vec;
    const int rows_left = min(g.M - m_base, 128), cols_left = g.N - n_base;   // may be <= 0: every lane out of range
    const __amdgpu_buffer_rsrc_t rBias = epi_rsrc(g.bias, (long long)n_base * 4, 0x7fffffff);
    AIM_LDS float* rowfac = (AIM_LDS float*)(scr + EPI_ROWFAC);
    if (rowf) {
        // rows lane and lane + 64 of the wave tile: rs = af[frame] * at[tok], vs = bt[tok] (neutral when null)
        const __amdgpu_buffer_rsrc_t rAf = epi_rsrc(g.af, 0, 0x7fffffff), rAt = epi_rsrc(g.at, 0, 0x7fffffff),
                                     rBt = epi_rsrc(g.bt, 0, 0x7fffffff);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m_base + h * 64 + lane;
            const int mc = m < g.M ? m : g.M - 1;
            const int frame = mc / g.ntok, tok = mc - frame * g.ntok;
            const float a = buf_load_f1(rAf, (unsigned)frame * 4u), t = buf_load_f1(rAt, (unsigned)tok * 4u),
                        b = buf_load_f1(rBt, (unsigned)tok * 4u);
            f32x2 f;
            f[0] = (g.af ? a : 1.0f) * (g.at ? t : 1.0f);
            f[1] = g.vec ? (g.bt ? b : 1.0f) : 0.0f;
            *(AIM_LDS f32x2*)(rowfac + (h * 64 + lane) * 2) = f;
        }
    }

    if constexpr (EPI == EPI_BF16 || EPI == EPI_ACT || EPI == EPI_DACT || EPI == EPI_RES16) {
        // bf16 outputs without a second input (BF16, ACT): bias, row factor, activation and the cast to bf16 are applied in
        // the MFMA layout (lane = row frow, 4 consecutive columns per 16-column tile), and a whole 16-row tile of bf16 values
        // crosses the scratch at a time: every lane writes (no exec mask), half the LDS bytes of the fp32 sub-passes below and
        // half as many round trips (stamped with the stores compiled out: the fp32 form spent 3.5 us of a BF16 tile's 3.75 us
        // in these round trips, not in the stores).  Same arithmetic per element as before: bit-identical results.
        // A wave's DS instructions execute in order, so neither the reads after the writes nor the next tile's writes after
        // these reads need a wait; the empty asm statements only pin the compiler's order.
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
        const bool ncol = c8 < cols_left;
        const __amdgpu_buffer_rsrc_t rOut =
            epi_rsrc(g.out, ((long long)m_base * g.ldo + n_base) * 2, (long long)rows_left * g.ldo * 2);
        const __amdgpu_buffer_rsrc_t rOut2 = epi_rsrc(EPI == EPI_ACT ? g.out2 : nullptr,
                                                      ((long long)m_base * g.ldo2 + n_base) * 2, (long long)rows_left * g.ldo2 * 2);
        // second input in row segments: DACT's saved rows (aux) | RES16's bf16 residual rows (resid, ldr)
        constexpr bool ROWIN = EPI == EPI_DACT || EPI == EPI_RES16;
        const int ldin = EPI == EPI_RES16 ? g.ldr : g.ldaux;
        const __amdgpu_buffer_rsrc_t rAux = epi_rsrc(EPI == EPI_DACT ? g.aux : EPI == EPI_RES16 ? (const void*)g.resid : nullptr,
                                                     ((long long)m_base * ldin + n_base) * 2, (long long)rows_left * ldin * 2);
        unsigned voA = ncol ? (unsigned)(r8 * ldin + c8) * 2u : AIM_OOB;
        const unsigned stA = (unsigned)ldin * 16u;
        // RES16: the per-frame vector `vec` of the (at most two) frames this wave tile touches, 64 columns each, in the scratch
        AIM_LDS float* vecs = (AIM_LDS float*)(scr + EPI_VEC);
        int bnd = 1 << 30;            // first tile-local row that belongs to the second frame
        if constexpr (EPI == EPI_RES16) {
            if (g.vec) {
                const int mcl = m_base < g.M ? m_base : g.M - 1;
                const int frame0 = mcl / g.ntok;
                const int nframes = (g.M + g.ntok - 1) / g.ntok;
                bnd = (frame0 + 1) * g.ntok - m_base;
                const __amdgpu_buffer_rsrc_t rVec = epi_rsrc(g.vec, (long long)n_base * 4, 0x7fffffff);
                const bool in = lane < cols_left;
                vecs[lane] = buf_load_f1(rVec, in ? (unsigned)(frame0 * g.ldv + lane) * 4u : AIM_OOB);
                vecs[64 + lane] = buf_load_f1(rVec, (in && frame0 + 1 < nframes) ? (unsigned)((frame0 + 1) * g.ldv + lane) * 4u : AIM_OOB);
            } else {
                vecs[lane] = 0.f;
                vecs[64 + lane] = 0.f;
            }
            asm volatile("" ::: "memory");
        }
        // aux_frag: out2 (ACT) / aux (DACT) is a FRAGMENT-ordered buffer private to this kernel pair -- [row tile][column tile]
        // [wave][16x16 tile i * 4 + j][lane] x 4 bf16, the lane's own accumulator elements, so neither side re-tiles it through
        // LDS: the forward stores 8 bytes per lane (512 contiguous bytes per instruction), the dgrad loads them back.  Both
        // GEMMs have the same M, N and tiling; the buffer is padded to whole tiles (ops.frag_buffer).
        const bool afrag = (EPI == EPI_ACT || EPI == EPI_DACT) && g.aux_frag != 0;
        const long long ftile = ((long long)(m_base >> 8) * ((g.N + 255) >> 8) + (n_base >> 8)) * 8 + (((m_base >> 7) & 1) * 4 + ((n_base >> 6) & 3));
        const __amdgpu_buffer_rsrc_t rFrag = epi_rsrc(!afrag ? nullptr : EPI == EPI_ACT ? (const void*)g.out2 : (const void*)g.aux,
                                                      ftile * (32 * 64 * 8), 32 * 64 * 8);
        const unsigned voF = (unsigned)lane * 8u;

        // DACT: the saved pre-activations come in row segments (8 rows x 128 B per load) and cross the scratch the other
        // way, into the MFMA layout, one 16-row tile ahead of their use
        bf16x8 ax[3][2] = {};           // three tiles in flight (a tile is ~1 us of work, a load 2-3 us under load)
        // aux_frag: the same registers hold three 16-row tiles of fragments (ax[slot][h] = fragments j = 2h, 2h + 1)
        auto load_frag16 = [&](int slot, int i) {
            if constexpr (EPI == EPI_DACT) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(rFrag, voF + (unsigned)((i * 4 + 2 * h) * 512), 0, AIM_LOAD_POLICY);
                    const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(rFrag, voF + (unsigned)((i * 4 + 2 * h + 1) * 512), 0, AIM_LOAD_POLICY);
                    ax[slot][h] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
                }
            }
        };
        auto load_aux16 = [&](int slot) {
            if constexpr (ROWIN) {
                ax[slot][0] = buf_load_h8(rAux, voA);
                epi_advance(voA, stA);
                ax[slot][1] = buf_load_h8(rAux, voA);
                epi_advance(voA, stA);
            }
        };
        f32x4 bj[4], wsj[4];
        bool qg[4], rson[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cj = j * 16 + fq * 4;                       // the lane's columns of tile j (tile-local)
            const unsigned vb = cj < cols_left ? (unsigned)cj * 4u : AIM_OOB;
            bj[j] = buf_load_f4(rBias, vb);
            wsj[j] = f32x4{1.f, 1.f, 1.f, 1.f};
            if constexpr (WS) wsj[j] = buf_load_f4(epi_rsrc(g.wscale, (long long)n_base * 4, 0x7fffffff), vb);
            qg[j] = col_act(g, n_base + cj) == ACT_QGELU;
            rson[j] = EPI == EPI_BF16 || EPI == EPI_RES16 || g.n_split == 0 || n_base + cj >= g.n_split;       // (ACT / DACT: adapter columns only)
        }
        unsigned voO = ncol ? (unsigned)(r8 * g.ldo + c8) * 2u : AIM_OOB;
        unsigned voO2 = ncol ? (unsigned)(r8 * g.ldo2 + c8) * 2u : AIM_OOB;
        const unsigned stO = (unsigned)g.ldo * 16u, stO2 = (unsigned)g.ldo2 * 16u;        // 8 rows of bf16
        AIM_LDS char* wr = scr + frow * EPI_RSH + fq * 8;          // + j * 32: the lane's 4 columns of tile j
        const AIM_LDS char* rd = scr + r8 * EPI_RSH + c8 * 2;      // rows r8 and r8 + 8, 8 columns
        auto cross = [&](const bf16x4 (&t)[4], __amdgpu_buffer_rsrc_t r, unsigned& vo, unsigned st) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) *(AIM_LDS bf16x4*)(wr + j * 32) = t[j];
            asm volatile("" ::: "memory");
            const bf16x8 x0 = *(const AIM_LDS bf16x8*)rd;
            const bf16x8 x1 = *(const AIM_LDS bf16x8*)(rd + 8 * EPI_RSH);
            buf_store16(r, vo, x0);
            epi_advance(vo, st);
            buf_store16(r, vo, x1);
            epi_advance(vo, st);
        };
        // ALLQ: every column of this wave tile takes QuickGELU (all but the adapter's column tile of the fused c_fc GEMM):
        // decided once per tile with a ballot, so the sub-passes carry no per-lane activation branch
        auto run16 = [&](auto ROWF, auto ALLQ) {
            if (afrag) {
                load_frag16(0, 0);
                load_frag16(1, 1);
                load_frag16(2, 2);
            } else {
                load_aux16(0);
                load_aux16(1);
                load_aux16(2);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float rs = 1.0f;
                if constexpr (decltype(ROWF)::value) rs = rowfac[(i * 16 + frow) * 2];
                bf16x4 o[4], pre[4];
                if (EPI == EPI_DACT && afrag) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bf16x8 f8 = ax[i % 3][j >> 1];
                        pre[j] = (j & 1) ? bf16x4{f8[4], f8[5], f8[6], f8[7]} : bf16x4{f8[0], f8[1], f8[2], f8[3]};
                    }
                    if (i + 3 < 8) load_frag16(i % 3, i + 3);
                } else if constexpr (ROWIN) {       // this tile's second input (saved rows | residual rows): row segments -> MFMA layout (pre[j])
                    asm volatile("" ::: "memory");
                    *(AIM_LDS bf16x8*)(scr + r8 * EPI_RSH + c8 * 2) = ax[i % 3][0];
                    *(AIM_LDS bf16x8*)(scr + (r8 + 8) * EPI_RSH + c8 * 2) = ax[i % 3][1];
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int j = 0; j < 4; ++j) pre[j] = *(const AIM_LDS bf16x4*)(wr + j * 32);
                    if (i + 3 < 8) load_aux16(i % 3);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (WS) v[e] = acc[i][j][e] * wsj[j][e] + bj[j][e];
                        else v[e] = acc[i][j][e] + bj[j][e];
                    }
                    const float rsj = rson[j] ? rs : 1.0f;
                    if constexpr (EPI == EPI_BF16) {
                        o[j] = pack4(rsj * v[0], rsj * v[1], rsj * v[2], rsj * v[3]);
                    } else if constexpr (EPI == EPI_RES16) {
                        // out = resid + rs (acc + bias) + bt[tok] vec[frame]: the F32 epilogue's sum on a bf16 residual stream
                        const int rl = i * 16 + frow;
                        const float vs = decltype(ROWF)::value ? rowfac[rl * 2 + 1] : 0.0f;
                        const f32x4 vv = *(const AIM_LDS f32x4*)(vecs + (rl >= bnd ? 64 : 0) + j * 16 + fq * 4);
                        float y[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = ((float)pre[j][e] + rsj * v[e]) + vs * vv[e];
                        o[j] = pack4(y[0], y[1], y[2], y[3]);
                    } else if constexpr (EPI == EPI_DACT) {
                        if (g.aux_grad) {          // aux holds the activation's derivative already
                            const f32x2 y0 = (f32x2{v[0], v[1]} * rsj) * f32x2{(float)pre[j][0], (float)pre[j][1]};
                            const f32x2 y1 = (f32x2{v[2], v[3]} * rsj) * f32x2{(float)pre[j][2], (float)pre[j][3]};
                            o[j] = pack4(y0[0], y0[1], y1[0], y1[1]);
                        } else if (decltype(ALLQ)::value || qg[j]) {
                            const f32x2 d0 = quick_gelu_grad2(f32x2{(float)pre[j][0], (float)pre[j][1]});
                            const f32x2 d1 = quick_gelu_grad2(f32x2{(float)pre[j][2], (float)pre[j][3]});
                            const f32x2 y0 = (f32x2{v[0], v[1]} * rsj) * d0, y1 = (f32x2{v[2], v[3]} * rsj) * d1;
                            o[j] = pack4(y0[0], y0[1], y1[0], y1[1]);
                        } else {
                            o[j] = pack4(rsj * v[0] * gelu_erf_grad((float)pre[j][0]), rsj * v[1] * gelu_erf_grad((float)pre[j][1]),
                                         rsj * v[2] * gelu_erf_grad((float)pre[j][2]), rsj * v[3] * gelu_erf_grad((float)pre[j][3]));
                        }
                    } else {
                        pre[j] = pack4(v[0], v[1], v[2], v[3]);
                        if (decltype(ALLQ)::value || qg[j]) {
                            f32x2 y0, y1, d0, d1;
                            quick_gelu_both2(f32x2{(float)pre[j][0], (float)pre[j][1]}, y0, d0);
                            quick_gelu_both2(f32x2{(float)pre[j][2], (float)pre[j][3]}, y1, d1);
                            y0 *= rsj;
                            y1 *= rsj;
                            o[j] = pack4(y0[0], y0[1], y1[0], y1[1]);
                            if (g.aux_grad) pre[j] = pack4(d0[0], d0[1], d1[0], d1[1]);
                        } else {
                            float y[4], d[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) gelu_erf_both((float)pre[j][e], y[e], d[e]);
                            o[j] = pack4(rsj * y[0], rsj * y[1], rsj * y[2], rsj * y[3]);
                            if (g.aux_grad) pre[j] = pack4(d[0], d[1], d[2], d[3]);
                        }
                    }
                }
                cross(o, rOut, voO, stO);
                if constexpr (EPI == EPI_ACT) {
                    if (afrag) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pre[j]), rFrag, voF + (unsigned)((i * 4 + j) * 512), 0, AIM_STORE_POLICY);
                    } else {
                        cross(pre, rOut2, voO2, stO2);
                    }
                }
            }
        };
        const bool allq = EPI != EPI_BF16 && EPI != EPI_RES16 && __builtin_amdgcn_ballot_w64(qg[0] && qg[1] && qg[2] && qg[3]) == ~0ull;
        if (rowf) {
            if (allq) run16(std::true_type{}, std::true_type{}); else run16(std::true_type{}, std::false_type{});
        } else {
            if (allq) run16(std::false_type{}, std::true_type{}); else run16(std::false_type{}, std::false_type{});
        }
    } else if constexpr (EPI == EPI_ACT8 && AIM_ACT8_MFMA_LAYOUT) {
        // fp8 output (inference: the [c_fc | D_fc1] GEMM) in the MFMA layout, like the bf16 outputs above: scale, bias, activation,
        // row factor, the saturating cast and the packing of a lane's four columns into ONE dword happen on the accumulators,
        // and a 16-row x 64-byte tile of fp8 crosses the scratch per sub-pass -- every lane writes four dwords, reads sixteen
        // bytes of one row and stores them (four lanes per row, 16 rows per instruction).  The fp32-scratch form below moved
        // 4x the LDS bytes in twice as many round trips per row (measured on the ViT-L/14 c_fc GEMM, M = 98 688, N = 4 352,
        // K = 1 024: DESIGN.md, fp8 inference).  N and ldo in multiples of 16: sixteen-byte stores; in multiples of 8 only
        // (checked by the launcher): two eight-byte stores of eight rows each.
        const bool wide = ((g.N | g.ldo) & 15) == 0;
        const int rr = wide ? lane >> 2 : lane >> 3, cc = wide ? (lane & 3) * 16 : (lane & 7) * 8;     // the lane's row / first column
        const __amdgpu_buffer_rsrc_t rOut = epi_rsrc(g.out, (long long)m_base * g.ldo + n_base, (long long)rows_left * g.ldo);
        f32x4 bj[4], wsj[4];
        bool qg[4], rson[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cj = j * 16 + fq * 4;
            const unsigned vb = cj < cols_left ? (unsigned)cj * 4u : AIM_OOB;
            bj[j] = buf_load_f4(rBias, vb);
            wsj[j] = f32x4{1.f, 1.f, 1.f, 1.f};
            if constexpr (WS) wsj[j] = buf_load_f4(epi_rsrc(g.wscale, (long long)n_base * 4, 0x7fffffff), vb);
            qg[j] = col_act(g, n_base + cj) == ACT_QGELU;
            rson[j] = g.n_split == 0 || n_base + cj >= g.n_split;
        }
        constexpr int RS8 = 80;                                   // scratch row stride: 64 fp8 + 16 B pad
        unsigned voO = cc < cols_left ? (unsigned)(rr * g.ldo + cc) : AIM_OOB;
        const unsigned stO = (unsigned)g.ldo * (wide ? 16u : 8u);  // rows per store instruction
        AIM_LDS char* wr = scr + frow * RS8 + fq * 4;             // + j * 16: the lane's 4 columns of tile j
        const AIM_LDS char* rd = scr + rr * RS8 + cc;
        auto run8m = [&](auto ROWF, auto ALLQ) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float rs = 1.0f;
                if constexpr (decltype(ROWF)::value) rs = rowfac[(i * 16 + frow) * 2];
                unsigned w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float rsj = rson[j] ? rs : 1.0f;
                    float y[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v;
                        if constexpr (WS) v = acc[i][j][e] * wsj[j][e] + bj[j][e];
                        else v = acc[i][j][e] + bj[j][e];
                        y[e] = rsj * ((decltype(ALLQ)::value || qg[j]) ? quick_gelu(v) : gelu_erf(v));
                    }
                    w[j] = pack4_fp8(y[0], y[1], y[2], y[3]);
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) *(AIM_LDS unsigned*)(wr + j * 16) = w[j];
                asm volatile("" ::: "memory");
                if (wide) {
                    const u32x4 x = *(const AIM_LDS u32x4*)rd;
                    buf_store16(rOut, voO, x);
                    epi_advance(voO, stO);
                } else {
                    const u32x2 x0 = *(const AIM_LDS u32x2*)rd;
                    const u32x2 x1 = *(const AIM_LDS u32x2*)(rd + 8 * RS8);
                    __builtin_amdgcn_raw_buffer_store_b64(x0, rOut, voO, 0, AIM_STORE_POLICY);
                    epi_advance(voO, stO);
                    __builtin_amdgcn_raw_buffer_store_b64(x1, rOut, voO, 0, AIM_STORE_POLICY);
                    epi_advance(voO, stO);
                }
            }
        };
        const bool allq = __builtin_amdgcn_ballot_w64(qg[0] && qg[1] && qg[2] && qg[3]) == ~0ull;
        if (rowf) {
            if (allq) run8m(std::true_type{}, std::true_type{}); else run8m(std::true_type{}, std::false_type{});
        } else {
            if (allq) run8m(std::false_type{}, std::true_type{}); else run8m(std::false_type{}, std::false_type{});
        }
    } else if constexpr (EPI != EPI_F32) {
        // fp8 output (ACT8; the fp32-scratch form, which also still serves as the reference for the branch above): 8 lanes x 8 columns per row, 8 rows per wave-instruction, so every
        // global access is a 16-byte-per-lane, whole-128-B-row-segment instruction (the 8-byte form is
        // store-issue-bound).  N and the leading dimensions are multiples of 8 here (checked by the launcher).
        // DACT's saved pre-activations are prefetched one 32-row group ahead.
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
        const bool ncol = c8 < cols_left;
        const int n = n_base + c8;
        constexpr int OES = EPI == EPI_ACT8 ? 1 : 2;          // output element bytes (fp8 | bf16)
        const __amdgpu_buffer_rsrc_t rOut =
            epi_rsrc(g.out, ((long long)m_base * g.ldo + n_base) * OES, (long long)rows_left * g.ldo * OES);
        const __amdgpu_buffer_rsrc_t rOut2 = epi_rsrc(EPI == EPI_ACT ? g.out2 : nullptr,
                                                      ((long long)m_base * g.ldo2 + n_base) * 2, (long long)rows_left * g.ldo2 * 2);
        const __amdgpu_buffer_rsrc_t rAux = epi_rsrc(EPI == EPI_DACT ? g.aux : nullptr,
                                                     ((long long)m_base * g.ldaux + n_base) * 2, (long long)rows_left * g.ldaux * 2);
        const f32x4 b0 = buf_load_f4(rBias, ncol ? (unsigned)c8 * 4u : AIM_OOB);
        const f32x4 b1 = buf_load_f4(rBias, ncol ? (unsigned)c8 * 4u + 16u : AIM_OOB);
        // per-output-channel dequantisation scale of an fp8 weight (null -> 1: the branch is on a kernel argument)
        f32x4 ws0 = f32x4{1.f, 1.f, 1.f, 1.f}, ws1 = ws0;
        if constexpr (WS) {
            const __amdgpu_buffer_rsrc_t rWs = epi_rsrc(g.wscale, (long long)n_base * 4, 0x7fffffff);
            ws0 = buf_load_f4(rWs, ncol ? (unsigned)c8 * 4u : AIM_OOB);
            ws1 = buf_load_f4(rWs, ncol ? (unsigned)c8 * 4u + 16u : AIM_OOB);
        }
        const int act = col_act(g, n);
        const bool rs_on = EPI == EPI_BF16 || g.n_split == 0 || n >= g.n_split;
        unsigned voO = ncol ? (unsigned)(r8 * g.ldo + c8) * (unsigned)OES : AIM_OOB;
        unsigned voO2 = ncol ? (unsigned)(r8 * g.ldo2 + c8) * 2u : AIM_OOB;
        unsigned voA = ncol ? (unsigned)(r8 * g.ldaux + c8) * 2u : AIM_OOB;
        const unsigned stO = (unsigned)g.ldo * 8u * (unsigned)OES, stO2 = (unsigned)g.ldo2 * 16u, stA = (unsigned)g.ldaux * 16u;
        auto load_aux = [&](bf16x8 (&ax)[4]) {
            if constexpr (EPI == EPI_DACT) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    ax[t] = buf_load_h8(rAux, voA);
                    epi_advance(voA, stA);
                }
            }
        };
        auto finish8 = [&](int grp, const bf16x8 (&ax)[4], auto ROWF) {
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                dump8(grp * 2 + (sp >> 1), sp & 1);
                const f32x4 v0 = *(const AIM_LDS f32x4*)(scr + r8 * EPI_RS + c8 * 4);
                const f32x4 v1 = *(const AIM_LDS f32x4*)(scr + r8 * EPI_RS + c8 * 4 + 16);
                float rs = 1.0f;
                if constexpr (decltype(ROWF)::value) {
                    const float f = rowfac[(grp * 32 + sp * 8 + r8) * 2];
                    rs = rs_on ? f : 1.0f;
                }
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (WS) { v[e] = v0[e] * ws0[e] + b0[e]; v[4 + e] = v1[e] * ws1[e] + b1[e]; }
                    else { v[e] = v0[e] + b0[e]; v[4 + e] = v1[e] + b1[e]; }
                }
                bf16x8 o, pre;
                if constexpr (EPI == EPI_ACT8) {
                    // inference: activation, row factor, saturating cast to fp8 e4m3 (two per v_cvt_pk_fp8_f32)
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float t = rs * (act == ACT_QGELU ? quick_gelu(v[e]) : gelu_erf(v[e]));
                        y[e] = __builtin_fminf(__builtin_fmaxf(t, -448.f), 448.f);
                    }
                    int w0 = 0, w1 = 0;
                    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], w0, false);
                    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], w0, true);
                    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[4], y[5], w1, false);
                    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[6], y[7], w1, true);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)w0, (unsigned)w1}, rOut, voO, 0, AIM_STORE_POLICY);
                    epi_advance(voO, stO);
                } else if constexpr (EPI == EPI_BF16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(rs * v[e]);
                } else if constexpr (EPI == EPI_ACT) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) pre[e] = (bf16_t)v[e];
                    if (act == ACT_QGELU) {
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const f32x2 y = quick_gelu2(f32x2{(float)pre[e], (float)pre[e + 1]}) * rs;
                            o[e] = (bf16_t)y[0];
                            o[e + 1] = (bf16_t)y[1];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(rs * gelu_erf((float)pre[e]));
                    }
                } else {
                    if (act == ACT_QGELU) {
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const f32x2 d = quick_gelu_grad2(f32x2{(float)ax[sp][e], (float)ax[sp][e + 1]});
                            const f32x2 y = (f32x2{v[e], v[e + 1]} * rs) * d;
                            o[e] = (bf16_t)y[0];
                            o[e + 1] = (bf16_t)y[1];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            o[e] = (bf16_t)(rs * v[e] * gelu_erf_grad((float)ax[sp][e]));
                    }
                }
                if constexpr (EPI != EPI_ACT8) {
                    buf_store16(rOut, voO, o);
                    epi_advance(voO, stO);
                }
                if constexpr (EPI == EPI_ACT) {
                    buf_store16(rOut2, voO2, pre);
                    epi_advance(voO2, stO2);
                }
            }
        };
        bf16x8 xa[4], xb[4];
        auto run8 = [&](auto ROWF) {
            load_aux(xa);
            load_aux(xb); finish8(0, xa, ROWF);
            load_aux(xa); finish8(1, xb, ROWF);
            load_aux(xb); finish8(2, xa, ROWF);
            finish8(3, xb, ROWF);
        };
        if (rowf) run8(std::true_type{}); else run8(std::false_type{});
    } else {
        // fp32 output (residual stream): 16 lanes x 4 columns per row, 4 rows per wave-instruction
        const int rr = lane >> 4, cc = (lane & 15) * 4;
        const bool ncol = cc < cols_left;
        const __amdgpu_buffer_rsrc_t rOut =
            epi_rsrc(g.out, ((long long)m_base * g.ldo + n_base) * 4, (long long)rows_left * g.ldo * 4);
        const __amdgpu_buffer_rsrc_t rRes =
            epi_rsrc(g.resid, ((long long)m_base * g.ldr + n_base) * 4, (long long)rows_left * g.ldr * 4);
        const f32x4 bias4 = buf_load_f4(rBias, ncol ? (unsigned)cc * 4u : AIM_OOB);
        f32x4 ws4 = f32x4{1.f, 1.f, 1.f, 1.f};
        if constexpr (WS) ws4 = buf_load_f4(epi_rsrc(g.wscale, (long long)n_base * 4, 0x7fffffff), ncol ? (unsigned)cc * 4u : AIM_OOB);
        // `vec` rows are per frame, and the wave tile's 128 rows touch at most two frames (ntok >= 128, checked
        // by the launcher): both candidate rows are fetched once and selected per row
        f32x4 w0 = f32x4{0.f, 0.f, 0.f, 0.f}, w1 = w0;
        int bnd = 1 << 30;            // first tile-local row that belongs to the second frame
        if (g.vec) {
            const int mcl = m_base < g.M ? m_base : g.M - 1;
            const int frame0 = mcl / g.ntok;
            const int nframes = (g.M + g.ntok - 1) / g.ntok;
            bnd = (frame0 + 1) * g.ntok - m_base;
            const __amdgpu_buffer_rsrc_t rVec = epi_rsrc(g.vec, (long long)n_base * 4, 0x7fffffff);
            w0 = buf_load_f4(rVec, ncol ? (unsigned)(frame0 * g.ldv + cc) * 4u : AIM_OOB);
            w1 = buf_load_f4(rVec, (ncol && frame0 + 1 < nframes) ? (unsigned)((frame0 + 1) * g.ldv + cc) * 4u : AIM_OOB);
        }
        unsigned voO = ncol ? (unsigned)(rr * g.ldo + cc) * 4u : AIM_OOB;
        unsigned voR = ncol ? (unsigned)(rr * g.ldr + cc) * 4u : AIM_OOB;
        const unsigned stO = (unsigned)g.ldo * 16u, stR = (unsigned)g.ldr * 16u;     // 4 rows of fp32
        // group = 32 rows (two MFMA row-tiles): 8 rows per lane, rows t*4 + rr of the group
        auto load_rows = [&](f32x4 (&ri)[8]) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                ri[t] = buf_stream_f4(rRes, voR);
                epi_advance(voR, stR);
            }
        };
        auto finish = [&](int grp, const f32x4 (&ri)[8], auto ROWF) {
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {      // sub-pass: group rows 8*sp .. 8*sp+7
                dump8(grp * 2 + (sp >> 1), sp & 1);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = sp * 2 + tt;
                    f32x4 v = *(const AIM_LDS f32x4*)(scr + (tt * 4 + rr) * EPI_RS + cc * 4);
                    if constexpr (WS) v *= ws4;
                    if constexpr (decltype(ROWF)::value) {
                        const int r0 = grp * 32 + t * 4;           // uniform part of the tile-local row r0 + rr
                        const f32x2 f = *(const AIM_LDS f32x2*)(rowfac + rr * 2 + r0 * 2);
                        // rs_bias_only: v + rs*b, else (v + b)*rs -- one form, no branch
                        v = v * (g.rs_bias_only ? 1.0f : f[0]) + f[0] * bias4;
                        v += f[1] * (rr >= bnd - r0 ? w1 : w0);
                    } else {
                        v += bias4;
                    }
                    v += ri[t];
                    buf_store16(rOut, voO, v);
                    epi_advance(voO, stO);
                }
            }
        };
        // loads of group p+1 are issued before the stores of group p: a wave never waits on a store ack
        f32x4 ra[8], rb[8];
        auto run4 = [&](auto ROWF) {
            load_rows(ra);
            load_rows(rb); finish(0, ra, ROWF);
            load_rows(ra); finish(1, rb, ROWF);
            load_rows(rb); finish(2, ra, ROWF);
            finish(3, rb, ROWF);
        };
        if (rowf) run4(std::true_type{}); else run4(std::false_type{});
    }
}

// EXPSUM for the 256x256 kernel: each wave emits ONE partial (max, sum exp(x - max)) over the valid elements of its
// 128x64 sub-tile of scale * A W^T -> slot [wave] of the tile's 8 slots (aim_gemm_expsum_tiles counts them).  With an
// extra key (g.xrow) column g.N is reduced apart into slot [8 + wave].
__device__ __forceinline__ void wave_expsum(const GemmArgs& g, f32x4 (&acc)[8][4], int m_base, int n_base, int lane, float* slot) {
    const int frow = lane & 15, fq = lane >> 4;
    const bool cross = g.xrow != nullptr;
    float mx = -INFINITY, cmx = -INFINITY;
    float cv[8];                       // this lane's scores against the extra key (at most one column, 8 rows)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m_base + i * 16 + frow;
        cv[i] = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n_base + j * 16 + fq * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = acc[i][j][e] * g.scale;
                if (cross && n + e == g.N && m < g.M) cv[i] = x;
                const float v = (m < g.M && n + e < g.N) ? x : -INFINITY;
                acc[i][j][e] = v;
                mx = fmaxf(mx, v);
            }
        }
        cmx = fmaxf(cmx, cv[i]);
    }
    mx = wave_max(mx);
    float s = 0.f;
    if (mx > -INFINITY) {
        const float m2 = mx * 1.4426950408889634f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) s += __builtin_amdgcn_exp2f(acc[i][j][e] * 1.4426950408889634f - m2);
    }
    s = wave_sum(s);
    if (lane == 0) {
        slot[0] = mx;
        slot[1] = s;
    }
    if (cross) {
        cmx = wave_max(cmx);
        float cs = 0.f;
        if (cmx > -INFINITY) {
#pragma unroll
            for (int i = 0; i < 8; ++i) cs += __builtin_amdgcn_exp2f((cv[i] - cmx) * 1.4426950408889634f);
        }
        cs = wave_sum(cs);
        if (lane == 0) {
            slot[16] = cmx;
            slot[17] = cs;
        }
    }
}
