// Fused GEMM epilogues shared by the 128x128 and 256x256 kernels.  A lane owns, per 16x16 MFMA tile,
// row m and the 4 consecutive columns n..n+3 (the MFMA is issued with the weight fragment first).
#pragma once
#include "aim_common.h"
#include "aim_kernels_internal.h"

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
        r.frame = m / g.ntok;
        const int tok = m - r.frame * g.ntok;
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
    if (g.bias) {
        const f32x4 b = *(const f32x4*)(g.bias + n);
        if (EPI == EPI_F32 && g.rs_bias_only) v += rs * b; else v += b;
    }
    if constexpr (EPI == EPI_BF16) {
        v *= rs;
        *(bf16x4*)((bf16_t*)g.out + (long long)m * g.ldo + n) = pack4(v[0], v[1], v[2], v[3]);
    } else if constexpr (EPI == EPI_ACT) {
        const bf16x4 pre = pack4(v[0], v[1], v[2], v[3]);
        *(bf16x4*)((bf16_t*)g.out2 + (long long)m * g.ldo2 + n) = pre;
        const float p0 = (float)pre[0], p1 = (float)pre[1], p2 = (float)pre[2], p3 = (float)pre[3];
        const bf16x4 post = (act == ACT_QGELU)
                                ? pack4(rs * quick_gelu(p0), rs * quick_gelu(p1), rs * quick_gelu(p2), rs * quick_gelu(p3))
                                : pack4(rs * gelu_erf(p0), rs * gelu_erf(p1), rs * gelu_erf(p2), rs * gelu_erf(p3));
        *(bf16x4*)((bf16_t*)g.out + (long long)m * g.ldo + n) = post;
    } else if constexpr (EPI == EPI_DACT) {
        const bf16x4 pre = fin.aux;
        float d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            d[e] = (act == ACT_QGELU) ? quick_gelu_grad((float)pre[e]) : gelu_erf_grad((float)pre[e]);
        *(bf16x4*)((bf16_t*)g.out + (long long)m * g.ldo + n) =
            pack4(rs * v[0] * d[0], rs * v[1] * d[1], rs * v[2] * d[2], rs * v[3] * d[3]);
    } else if constexpr (EPI == EPI_F32) {
        if (!g.rs_bias_only) v *= rs;
        if (g.vec) {
            const f32x4 w = *(const f32x4*)(g.vec + (long long)rf.frame * g.ldv + n);
            v += rf.vs * w;
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
        *(bf16x8*)((bf16_t*)g.out2 + (long long)m * g.ldo2 + n) = pre;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float p = (float)pre[e];
            o[e] = (bf16_t)(rs * (act == ACT_QGELU ? quick_gelu(p) : gelu_erf(p)));
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float p = (float)aux8[e];
            o[e] = (bf16_t)(rs * v[e] * (act == ACT_QGELU ? quick_gelu_grad(p) : gelu_erf_grad(p)));
        }
    }
    *(bf16x8*)((bf16_t*)g.out + (long long)m * g.ldo + n) = o;
}

// ------------------------------------------------------------------------------------------------
// Epilogue of one wave's 128x64 accumulator tile (acc[8][4], 16x16 MFMA fragments), re-tiled through a
// private LDS scratch (16 rows x 272 B) so global accesses are whole 128-256 B row segments.
// Memory-side inputs are software-pipelined: the loads of row-block p+1 are issued BEFORE the stores
// of row-block p, so a wave never waits on a store acknowledgement (vmcnt is in-order and counts
// stores) and keeps two row-blocks of loads in flight.  Column-only inputs (bias) are loaded once.
// ------------------------------------------------------------------------------------------------
constexpr int EPI_RS = 272;                      // scratch row stride: 64 f32 + 16 B pad
constexpr int EPI_SCRATCH = 8 * EPI_RS;          // bytes per wave: 8 rows at a time (fits beside the K-loop images)

template <int EPI>
struct RowIn {       // memory-side inputs of one (row, lane) 4-column fragment
    f32x4 resid;
    bf16x4 aux;
};

template <int EPI>
__device__ __forceinline__ void wave_epilogue(const GemmArgs& g, f32x4 (&acc)[8][4], AIM_LDS char* scr, int m_base,
                                              int n_base, int lane) {
    const int frow = lane & 15, fq = lane >> 4;
    // half of one 16-row MFMA tile -> scratch rows 0..7 (the lanes holding the other half sit out)
    auto dump8 = [&](int mt, int half) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // previous sub-pass's scratch reads are done (WAR)
        if ((frow >> 3) == half) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(AIM_LDS f32x4*)(scr + (frow & 7) * EPI_RS + (j * 16 + fq * 4) * 4) = acc[mt][j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // writes landed before other lanes read them
    };
    const bool rowf = g.af || g.at || g.vec;
    if constexpr (EPI != EPI_F32) {
        // bf16 outputs (BF16 / ACT / DACT): 8 lanes x 8 columns per row, 8 rows per wave-instruction, so every
        // global access is a 16-byte-per-lane, whole-128-B-row-segment instruction (the 8-byte form is
        // store-issue-bound).  DACT's saved pre-activations are prefetched one 32-row group ahead.
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
        const int n = n_base + c8;
        const bool ncol = n < g.N;
        const bool wide = (g.ldo % 8) == 0 && n + 8 <= g.N && (EPI != EPI_ACT || (g.ldo2 % 8) == 0) &&
                          (EPI != EPI_DACT || (g.ldaux % 8) == 0);
        f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (g.bias && ncol) {
            b0 = *(const f32x4*)(g.bias + n);
            if (n + 4 < g.N) b1 = *(const f32x4*)(g.bias + n + 4);
        }
        const int act = col_act(g, n);
        const bool rs_on = EPI == EPI_BF16 || g.n_split == 0 || n >= g.n_split;
        auto load_aux = [&](int grp, bf16x8 (&ax)[4]) {
            if constexpr (EPI == EPI_DACT) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int m = m_base + grp * 32 + t * 8 + r8;
                    if (m < g.M && ncol) {
                        const bf16_t* ap = (const bf16_t*)g.aux + (long long)m * g.ldaux + n;
                        if (wide) {
                            ax[t] = *(const bf16x8*)ap;
                        } else {
                            const bf16x4 lo = *(const bf16x4*)ap;
                            bf16x4 hi = lo;
                            if (n + 4 < g.N) hi = *(const bf16x4*)(ap + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { ax[t][e] = lo[e]; ax[t][4 + e] = hi[e]; }
                        }
                    }
                }
            }
        };
        auto finish8 = [&](int grp, const bf16x8 (&ax)[4]) {
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                dump8(grp * 2 + (sp >> 1), sp & 1);
                const f32x4 v0 = *(const AIM_LDS f32x4*)(scr + r8 * EPI_RS + c8 * 4);
                const f32x4 v1 = *(const AIM_LDS f32x4*)(scr + r8 * EPI_RS + c8 * 4 + 16);
                const int m = m_base + grp * 32 + sp * 8 + r8;
                if (m >= g.M || !ncol) continue;
                float rs = 1.0f;
                if (rowf && rs_on) rs = row_factors(g, m).rs;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = v0[e] + b0[e]; v[4 + e] = v1[e] + b1[e]; }
                bf16x8 o, pre;
                if constexpr (EPI == EPI_BF16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(rs * v[e]);
                } else if constexpr (EPI == EPI_ACT) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        pre[e] = (bf16_t)v[e];
                        const float x = (float)pre[e];
                        o[e] = (bf16_t)(rs * (act == ACT_QGELU ? quick_gelu(x) : gelu_erf(x)));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = (float)ax[sp][e];
                        o[e] = (bf16_t)(rs * v[e] * (act == ACT_QGELU ? quick_gelu_grad(x) : gelu_erf_grad(x)));
                    }
                }
                bf16_t* op = (bf16_t*)g.out + (long long)m * g.ldo + n;
                if (wide) {
                    *(bf16x8*)op = o;
                    if constexpr (EPI == EPI_ACT) *(bf16x8*)((bf16_t*)g.out2 + (long long)m * g.ldo2 + n) = pre;
                } else {       // ragged N / odd strides: two 4-column halves
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        if (n + 4 * hh >= g.N) break;
                        *(bf16x4*)(op + 4 * hh) = bf16x4{o[4 * hh], o[4 * hh + 1], o[4 * hh + 2], o[4 * hh + 3]};
                        if constexpr (EPI == EPI_ACT)
                            *(bf16x4*)((bf16_t*)g.out2 + (long long)m * g.ldo2 + n + 4 * hh) =
                                bf16x4{pre[4 * hh], pre[4 * hh + 1], pre[4 * hh + 2], pre[4 * hh + 3]};
                    }
                }
            }
        };
        bf16x8 xa[4], xb[4];
        load_aux(0, xa);
        load_aux(1, xb); finish8(0, xa);
        load_aux(2, xa); finish8(1, xb);
        load_aux(3, xb); finish8(2, xa);
        finish8(3, xb);
    } else {
        const int rr = lane >> 4, cc = (lane & 15) * 4;
        const int n = n_base + cc;
        const bool ncol = n < g.N;
        f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g.bias && ncol) bias4 = *(const f32x4*)(g.bias + n);
        const int act = col_act(g, n);
        const bool rs_on = !(EPI == EPI_ACT || EPI == EPI_DACT) || g.n_split == 0 || n >= g.n_split;
        constexpr bool HAS_IN = (EPI == EPI_F32 || EPI == EPI_DACT);
        // group = 32 rows (two MFMA row-tiles): 8 rows per lane, rows t*4 + rr of the group
        auto load_rows = [&](int grp, RowIn<EPI> (&ri)[8]) {
            if constexpr (HAS_IN) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int m = m_base + grp * 32 + t * 4 + rr;
                    ri[t].resid = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (m < g.M && ncol) {
                        if constexpr (EPI == EPI_F32) {
                            if (g.resid) ri[t].resid = *(const f32x4*)(g.resid + (long long)m * g.ldr + n);
                        }
                        if constexpr (EPI == EPI_DACT)
                            ri[t].aux = *(const bf16x4*)((const bf16_t*)g.aux + (long long)m * g.ldaux + n);
                    }
                }
            }
        };
        auto finish = [&](int grp, const RowIn<EPI> (&ri)[8]) {
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {      // sub-pass: group rows 8*sp .. 8*sp+7
                dump8(grp * 2 + (sp >> 1), sp & 1);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = sp * 2 + tt;
                    f32x4 v = *(const AIM_LDS f32x4*)(scr + (tt * 4 + rr) * EPI_RS + cc * 4);
                    const int m = m_base + grp * 32 + t * 4 + rr;
                    if (m >= g.M || !ncol) continue;
                    RowFactors rf{1.0f, 0.0f, 0};
                    if (rowf && rs_on) rf = row_factors(g, m);
                    const float rs = rf.rs;
                    if (EPI == EPI_F32 && g.rs_bias_only) v += rs * bias4; else v += bias4;
                    if constexpr (EPI == EPI_ACT) {
                        const bf16x4 pre = pack4(v[0], v[1], v[2], v[3]);
                        *(bf16x4*)((bf16_t*)g.out2 + (long long)m * g.ldo2 + n) = pre;
                        float a[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = (float)pre[e];
                            a[e] = rs * (act == ACT_QGELU ? quick_gelu(x) : gelu_erf(x));
                        }
                        *(bf16x4*)((bf16_t*)g.out + (long long)m * g.ldo + n) = pack4(a[0], a[1], a[2], a[3]);
                    } else if constexpr (EPI == EPI_DACT) {
                        float a[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = (float)ri[t].aux[e];
                            a[e] = rs * v[e] * (act == ACT_QGELU ? quick_gelu_grad(x) : gelu_erf_grad(x));
                        }
                        *(bf16x4*)((bf16_t*)g.out + (long long)m * g.ldo + n) = pack4(a[0], a[1], a[2], a[3]);
                    } else {   // EPI_F32
                        if (!g.rs_bias_only) v *= rs;
                        if (g.vec) v += rf.vs * *(const f32x4*)(g.vec + (long long)rf.frame * g.ldv + n);
                        v += ri[t].resid;
                        *(f32x4*)((float*)g.out + (long long)m * g.ldo + n) = v;
                    }
                }
            }
        };
        // loads of group p+1 are issued before the stores of group p: a wave never waits on a store ack
        RowIn<EPI> ra[8], rb[8];
        load_rows(0, ra);
        load_rows(1, rb); finish(0, ra);
        load_rows(2, ra); finish(1, rb);
        load_rows(3, rb); finish(2, ra);
        finish(3, rb);
    }
}
