/*
 * aim_kernels.h -- C ABI of libaim_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * AIM ViT-CLIP + Adapter forward/backward hot path.
 *
 * The reference (bobochow/adapt-image-models) is 100 % Python over PyTorch eager ops and has no
 * FFI of its own; each entry point below replaces the eager-op sequence at the cited lines of
 * mmaction/models/backbones/vit_clip.py.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success; non-zero -> aim_last_error() (thread-local text);
 *   - all pointers are DEVICE pointers unless said otherwise; the caller owns all memory
 *     (no allocation, no implicit synchronisation and no state that outlives a call inside the
 *     library -- the only statics are idempotent one-time initialisations (a kernel's dynamic-LDS
 *     attribute, environment switches read once); every knob of a launch is an argument of that
 *     call, so two host threads may drive two streams through the library at once);
 *   - `stream` is a hipStream_t passed as void*; launches are stream-ordered;
 *   - matrices are row-major; "bf16" is IEEE bfloat16 stored as uint16_t; "ld*" are row strides
 *     in ELEMENTS;
 *   - token rows are frame-major: row = (b*T + t)*N + n for clip b, frame t, token n
 *     (the reference's [N, BT, D] tensor transposed; N = (res/patch)^2 + 1, D = width);
 *   - head_dim is 64 for every supported model (ViT-B/16: 12x64, ViT-L/14: 16x64).
 */
#ifndef AIM_KERNELS_H
#define AIM_KERNELS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t aim_bf16;

#define AIM_ABI_VERSION 6

int aim_version(void);                /* == AIM_ABI_VERSION */
const char* aim_last_error(void);     /* message of the last failing call on this thread */

/* ------------------------------------------------------------------------------------------
 * GEMM  C[m][n] = sum_k A[m][k] * W[n][k]  (bf16 operands, fp32 accumulate, MFMA 16x16x32)
 * replaces: q/k/v projections vit_clip.py:132-138,168-173; attn.out_proj :157,192;
 *           mlp.c_fc/QuickGELU/c_proj :93-97,286; Adapter.D_fc1/GELU/D_fc2 :62-64;
 *           residual combines :275,286; and (autograd) the dgrad of every frozen Linear
 *           (pass the transposed weight).
 * Requirements: K, lda, ldw multiples of 8; N, ldo (and ldo2/ldr/ldv/ldaux when used) multiples
 * of 4; pointers 16-byte aligned.
 * ------------------------------------------------------------------------------------------ */
enum {
    AIM_EPI_BF16 = 0,  /* out(bf16) = rs * (acc + bias)                                         */
    AIM_EPI_ACT = 1,   /* out2(bf16) = pre = acc + bias ; out(bf16) = rs * act(pre)             */
    AIM_EPI_DACT = 2,  /* out(bf16) = rs * (acc + bias) * act'(aux)        (aux = saved pre)    */
    AIM_EPI_F32 = 3,   /* out(f32) = resid + rs*(acc + bias) + bt[tok]*vec[frame][n]
                          (rs_bias_only != 0: out = resid + acc + rs*bias + ...)               */
    AIM_EPI_EXPSUM = 4,/* out(f32)[batch][tile][2] = (max, sum exp(scale*acc - max)) over the
                          valid part of each 128x128 tile (lambda statistics, :149-151)         */
    AIM_EPI_ACT8 = 5,  /* out(fp8 e4m3) = sat(rs * act(acc + bias)): inference only, nothing saved
                          for a backward (aim_gemm_fp8)                                         */
    AIM_EPI_RES16 = 6  /* out(bf16) = resid(bf16) + rs*(acc + bias) + bt[tok]*vec[frame][n]: AIM_EPI_F32's sum on
                          a bf16 residual stream (fp8 inference only, aim_gemm_fp8; ldr in bf16 elements)    */
};
enum { AIM_ACT_QGELU = 0, AIM_ACT_GELU = 1 };

typedef struct aim_gemm_args {
    const aim_bf16* A;      /* [batch][M, lda]                                                  */
    const aim_bf16* W;      /* [batch][N, ldw]                                                  */
    int32_t lda, ldw;
    int64_t strideA, strideW; /* batch strides in elements (0 for batch == 1)                   */
    int32_t M, N, K;
    const float* bias;      /* [N] or NULL                                                      */
    const float* resid;     /* [M, ldr] f32 or NULL (AIM_EPI_F32)                               */
    int32_t ldr;
    /* row factor rs = af[row / ntok] * at[row % ntok]  (either may be NULL -> 1)               */
    const float* af;        /* [M / ntok] per-frame factor, e.g. 1 - lamda                      */
    const float* at;        /* [ntok]     per-token factor, e.g. DropPath mask * adapter scale  */
    const float* vec;       /* [M / ntok, ldv] per-frame row vector added to every token        */
    const float* bt;        /* [ntok] factor on vec (NULL -> 1)                                 */
    int32_t ldv, ntok;
    const aim_bf16* aux;    /* [M, ldaux] saved pre-activation (AIM_EPI_DACT)                   */
    int32_t ldaux;
    void* out;  int32_t ldo;
    void* out2; int32_t ldo2;
    float scale;            /* logit scale (AIM_EPI_EXPSUM)                                     */
    int32_t act;            /* AIM_ACT_*                                                        */
    int32_t rs_bias_only;   /* AIM_EPI_F32: apply rs to the bias term only                      */
    /* column split for fused [frozen MLP | adapter] GEMMs (ACT / DACT): when n_split > 0, columns
       n < n_split use `act` with rs = 1, columns n >= n_split use `act2` with the row factor rs   */
    int32_t n_split, act2;
    /* AIM_EPI_EXPSUM, one 256x256 tile per batch item (128 < max(M, N), N < 256) only: an EXTRA key.  Row N of W for batch
       item z is xrow + z * ldx (K elements); the scores against it (column N of the tile) are reduced apart into slots
       8..15 of the item's 16 (max, sum) slots -- lamda's `cw` rides along with its `ow` (vit_clip.py:149-151,184-186). */
    const aim_bf16* xrow;
    int32_t ldx;
    /* Scheduling knob of THIS launch (no reference counterpart): the large-tile GEMM is persistent (one workgroup per
       CU); with reserve_cus > 0 its grid leaves that many CUs free, so that kernels queued on ANOTHER stream (the
       class-token chain beside the QKV projection) are not starved. */
    int32_t reserve_cus;
    /* Diagnostics of THIS launch (tools/probe_gemm.py only): when probe != NULL (device memory, probe_cap x 4 uint64)
       the large-tile kernel records {workgroup | K-loop cycles, tile start, K-loop end, epilogue end} (100 MHz ticks)
       per processed tile. */
    void* probe;
    int32_t probe_cap;
    /* per-output-channel dequantisation scale of W (aim_gemm_fp8): acc is multiplied by wscale[n] before bias / act /
       row factors.  NULL -> 1.  */
    const float* wscale;
    /* AIM_EPI_ACT / AIM_EPI_DACT: what `out2` / `aux` hold.  0: the pre-activation (the DACT epilogue evaluates the
       derivative from it).  1: the activation's DERIVATIVE at the bf16-rounded pre-activation, computed by the ACT epilogue
       from the sigmoid / normal CDF it needs anyway -- the DACT epilogue is then dgrad x aux x row factor, no
       transcendentals (a pre-activation is saved for the backward only: vit_clip.py:80-82,93-97 under autograd). */
    int32_t aux_grad;
    /* AIM_EPI_ACT / AIM_EPI_DACT on the large-tile kernel (M >= 1024, batch 1): `out2` / `aux` is a FRAGMENT-ordered buffer,
       private to the pair of launches with the same M and N: (ceil(M/256) x ceil(N/256) tiles) x [8 waves][32 fragments]
       [64 lanes] x 4 bf16 -- each lane's own accumulator elements, so neither epilogue re-tiles it (ldo2 / ldaux are
       ignored; the buffer holds ceil(M/256) * ceil(N/256) * 65536 elements). */
    int32_t aux_frag;
    /* Row index, in the caller's whole problem, of row 0 of THIS launch's A / out / resid pointers: the row factors are read
       as af[(row0 + m) / ntok], at[(row0 + m) % ntok] (same for vec / bt).  0 for a whole problem.  The library uses it
       itself when it peels the thin last tile round of a large launch into a second, low-latency launch (AIM_GEMM_PEEL);
       the small-tile kernels honour it, the persistent 256 x 256 kernel requires 0. */
    int32_t row0;
} aim_gemm_args;

int aim_gemm_bf16(const aim_gemm_args* args, int epilogue, int batch, void* stream);

/* fp8 inference GEMM (BASELINE configs[4]; no reference counterpart -- the reference runs apex-O1 fp16):
 *   C[m][n] = wscale[n] * sum_k A8[m][k] * W8[n][k]
 * A and W hold OCP e4m3 bytes (K-contiguous; lda / ldw / K in elements, multiples of 16), products run on the
 * block-scaled MFMA v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (2x the bf16 MFMA rate), fp32 accumulate.
 * Weights are quantised once per output channel (wscale[n] = amax_n / 448), activations are saturating casts made by
 * the producing kernel (aim_layernorm_fwd y_fp8, AIM_EPI_ACT8, aim_attn_fwd out_fp8).  Epilogues: AIM_EPI_BF16,
 * AIM_EPI_F32, AIM_EPI_ACT8, AIM_EPI_RES16.  Large-M problems only (M >= 1024, N >= 64, N % 8 == 0). */
int aim_gemm_fp8(const aim_gemm_args* args, int epilogue, void* stream);
/* number of (max,sum) pairs AIM_EPI_EXPSUM writes per batch entry */
int aim_gemm_expsum_tiles(int M, int N);

/* ------------------------------------------------------------------------------------------
 * Weight-gradient GEMM for the (trainable) adapters:
 *   dW[n][k] += sum_m G[m][n] * A[m][k]      db[n] += sum_m G[m][n]
 * With a workspace the split-M partial slabs are summed in a fixed order by a second kernel (no atomics, bitwise
 * reproducible: the form the training step uses); workspace == NULL falls back to fp32 atomics.
 * autograd counterpart of Adapter.D_fc1 / D_fc2 (vit_clip.py:57-58).  Nw, Kw multiples of 8.
 * ------------------------------------------------------------------------------------------ */
int aim_wgrad_bf16(const aim_bf16* G, int ldg, const aim_bf16* A, int lda, float* dW, int lddw,
                   float* db, int M, int Nw, int Kw,
                   float* workspace /* optional: aim_wgrad_workspace_bytes(); NULL -> fp32 atomics (not reproducible) */,
                   int64_t workspace_bytes, void* stream);
int64_t aim_wgrad_workspace_bytes(int M, int Nw, int Kw);
/* the same with the bias gradient's row factor: db[n] += sum_m at[m % ntok] * G[m][n] (at == NULL: 1), summed inside the
 * weight-gradient kernel from the G rows it already holds in LDS (no separate pass over G).  The DropPath-scaled bias of
 * the MLP_Adapter's D_fc2 (vit_clip.py:286) is the user. */
int aim_wgrad_bias_bf16(const aim_bf16* G, int ldg, const aim_bf16* A, int lda, float* dW, int lddw, float* db,
                        const float* at, int ntok, int M, int Nw, int Kw, float* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm (fp32 statistics, eps inside rsqrt) -- vit_clip.py:71-77 (ln_1, ln_2, ln_pre, ln_post)
 *   fwd: y = (x - mean) * rstd * gamma + beta ; x fp32 rows with stride ldx; y as bf16 and/or f32
 *   bwd: dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma   (frozen gamma)
 *        optional: dgamma/dbeta accumulation for the trainable ln_post (rows <= 4096: one ordered column pass,
 *        no atomics; more rows: per-element fp32 atomics).
 * ------------------------------------------------------------------------------------------ */
int aim_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta,
                      aim_bf16* y_bf16, float* y_f32, int64_t ldy, float* mean, float* rstd,
                      int rows, int D, float eps, void* stream);
/* inference form: y as fp8 e4m3 bytes (saturating cast, row stride ldy bytes); no statistics are saved */
int aim_layernorm_fwd_fp8(const float* x, int64_t ldx, const float* gamma, const float* beta, uint8_t* y_fp8,
                          int64_t ldy, int rows, int D, float eps, void* stream);
/* the same LayerNorm over bf16 rows (the fp8 inference path keeps its residual stream in bf16: AIM_EPI_RES16);
 * any subset of the three outputs */
int aim_layernorm_fwd_x16(const aim_bf16* x, int64_t ldx, const float* gamma, const float* beta, aim_bf16* y_bf16,
                          float* y_f32, uint8_t* y_fp8, int64_t ldy, int rows, int D, float eps, void* stream);
int aim_layernorm_bwd(const void* dy, int dy_is_bf16 /* dy is bf16 (1) or f32 (0) */, int64_t lddy,
                      const float* x, int64_t ldx, const float* gamma,
                      const float* mean, const float* rstd, const void* dres, int dres_is_bf16, float* dx,
                      aim_bf16* dx_bf16, int64_t lddx, float* dgamma, float* dbeta,
                      int rows, int D, void* stream);
/* The bf16 form of aim_layernorm_bwd (dy, dres, dx all bf16, rows = frames*ntok) that also emits per-frame weighted column
 * sums of the dx it stores: fsum_partial[frame][g][c] = sum over token group g (ceil(ntok/groups) tokens) of
 * w[n] * dx_bf16[frame*ntok + n][c]   (w NULL = 1).  aim_frame_sum(fsum_partial, f32, NULL, out, frames, groups, D) finishes
 * it: the backward's sum_n dms1[n] d(x1)[frame, n, :] (autograd of the broadcast S_Adapter term, vit_clip.py:272-275)
 * without a second pass over d(x1). */
int aim_layernorm_bwd_fsum(const aim_bf16* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                           const float* mean, const float* rstd, const aim_bf16* dres, aim_bf16* dx_bf16, int64_t lddx,
                           const float* w, float* fsum_partial, int groups, int frames, int ntok, int D, void* stream);

/* ------------------------------------------------------------------------------------------
 * Spatial multi-head self-attention over the N tokens of each frame (flash-style, scores never
 * leave the CU) -- vit_clip.py:139-156 (reshape, q k^T / sqrt(dh), softmax, @ v, merge heads).
 *   qkv [BT*N, 3*D] bf16 (q | k | v, D = H*64) ; out [BT*N, D] bf16 ; lse [BT, H, N] f32
 *   bwd: dqkv [BT*N, 3*D] bf16 from dout [BT*N, D] bf16 (recomputes the probabilities from lse).
 * ------------------------------------------------------------------------------------------ */
int aim_attn_fwd(const aim_bf16* qkv, aim_bf16* out, float* lse, int BT, int N, int H, void* stream);
/* inference form: out as fp8 e4m3 bytes [BT*N, D] (operand of the fp8 out_proj GEMM); lse may be NULL */
int aim_attn_fwd_fp8(const aim_bf16* qkv, uint8_t* out_fp8, float* lse, int BT, int N, int H, void* stream);
int aim_attn_bwd(const aim_bf16* qkv, const aim_bf16* out, const aim_bf16* dout, const float* lse,
                 float* delta /* scratch [BT, H, N] f32 */, aim_bf16* dqkv, int BT, int N, int H,
                 void* stream);

/* ------------------------------------------------------------------------------------------
 * Temporal attention over the T class tokens of each clip -- vit_clip.py:220-224 with
 * attention() :139-156 at seq = T, batch = B.  Reads the class rows (token 0) of qkv.
 *   out_cls [B*T, D] bf16 ; probs [B, H, T, T] f32 (saved for backward)
 *   bwd: compact == 0: ADDS dq/dk/dv of the class rows into dqkv [BT*N, 3*D] (bf16, rows n == 0);
 *        compact != 0: WRITES them to dqkv [B*T, 3*D] (the class rows' share alone, for a separate dgrad).
 * ------------------------------------------------------------------------------------------ */
int aim_cls_attn_fwd(const aim_bf16* qkv, aim_bf16* out_cls, float* probs, int B, int T, int N, int H,
                     void* stream);
int aim_cls_attn_bwd(const aim_bf16* qkv, const float* probs, const aim_bf16* dout_cls, aim_bf16* dqkv, int compact,
                     int B, int T, int N, int H, void* stream);

/* ------------------------------------------------------------------------------------------
 * Temporal attention over the T frames of EVERY token position -- the stock-AIM block
 * (mmaction/models/backbones/vitclip_aim.py:199-205: attention() :148-187 on 'n (b t) d -> t (b n) d', seq = T, batch = B*N).
 *   qkv [B*T*N, 3*D] bf16 frame-major ; out [B*T*N, D] bf16 ; probs [B*N, H, T, T] f32 (saved for backward)
 *   bwd WRITES dqkv [B*T*N, 3*D] bf16.
 * aim_add_bf16: out = a + b (bf16, row strides) ; aim_acc_bf16: x (f32, dense) += s (bf16): the residual adds of that
 * block's S_Adapter branch (skip_connect=True, :211) that no GEMM epilogue of this library covers.
 * ------------------------------------------------------------------------------------------ */
int aim_tattn_fwd(const aim_bf16* qkv, aim_bf16* out, float* probs, int B, int T, int N, int H, void* stream);
int aim_tattn_bwd(const aim_bf16* qkv, const float* probs, const aim_bf16* dout, aim_bf16* dqkv, int B, int T, int N, int H,
                  void* stream);
int aim_add_bf16(const aim_bf16* a, int64_t lda, const aim_bf16* b, int64_t ldb, aim_bf16* out, int64_t ldo, int R, int C,
                 void* stream);
int aim_acc_bf16(float* x, const aim_bf16* s, int64_t lds, int R, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * lamda = cw / (cw + ow) per frame -- vit_clip.py:149-151, 184-186, 272.
 *   ow[bt] = sum_{i,j} exp(q_i . k_j / 8)  arrives as AIM_EPI_EXPSUM partial pairs
 *            partials [BT, ntiles, 2];  cw[bt] = sum_i exp(q_i . kx[bt] / 8) is computed here.
 *   Both sums share one max shift (identical ratio, no overflow).  lam [BT] f32; one_minus [BT].
 * ------------------------------------------------------------------------------------------ */
/* ss [BT, N] f32 = scale * q_i . kx[bt] (full width): the one pass over q that lamda's cw needs; aim_lambda accepts it so
 * that this pass can run as soon as q exists.  ss == NULL in aim_lambda: computed inside (one-call form). */
/* lamda from the 16 slots per frame of an AIM_EPI_EXPSUM launch with `xrow` (8 ow partials, 8 cw partials). */
int aim_lambda_partials(const float* partials, float* lam, float* one_minus_lam, int BT, void* stream);
int aim_qk_cross(const aim_bf16* qkv, const aim_bf16* kx, int ldkx, float* ss, int BT, int N, int D, float scale, void* stream);
int aim_lambda(const aim_bf16* qkv, const aim_bf16* kx, int ldkx, const float* ss, const float* partials, int ntiles,
               float* lam, float* one_minus_lam, int BT, int N, int D, float scale, void* stream);
/* lamda statistics when a frame has ONE token more than the large-tile GEMM's 256 rows (ViT-L/14, N = 257): the scores of tokens
 * 0 .. N-2 against each other are one full 256 x 256 tile per frame of aim_gemm_bf16(AIM_EPI_EXPSUM, M = N = 256, 8 slots
 * with a per-item slot stride of ldo floats); this adds the border -- (max, sum exp) of scale q_i . k_{N-1} (i < N-1) into
 * partials[bt][slot0] and of scale q_{N-1} . k_j (j < N) into partials[bt][slot0 + 1] -- and the cross scores
 * ss[bt][i] = scale q_i . kx[bt] in the same pass over q.  aim_lambda(ss, partials, nslots) finishes.  vit_clip.py:149-151,184-186 */
int aim_qk_border(const aim_bf16* qkv, const aim_bf16* kx, int ldkx, float* ss, float* partials /* [BT, nslots, 2] */, int slot0,
                  int nslots, int BT, int N, int D, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Patch embedding glue -- vit_clip.py:434-447.
 *   patchify: imgs [B,3,T,H,W] (f32, or uint8 with GPUNormalize mean/std fused:
 *             mmaction/utils/module_hooks.py:73-85) -> A [B*T*G*G, Kp] bf16, Kp = 3*p*p padded
 *             to a multiple of 64 with zeros (conv1 as a GEMM against conv1.weight.view(D,-1)).
 *   embed_ln: tok [B*T*G*G, D] bf16 + class/positional/temporal embeddings -> ln_pre ->
 *             x [B*T, N, D] f32 ; saves mean/rstd [B*T*N].
 *   embed_bwd: d(temporal_embedding)[T, D] += sum_{b,n} ln_pre_bwd(dx)  (two-stage ordered reduction through the
 *             workspace; workspace == NULL: fp32 atomics).
 * ------------------------------------------------------------------------------------------ */
int aim_patchify(const void* imgs, int in_dtype /* 0 f32, 1 uint8, 2 bf16 */, const float* mean3, const float* std3, aim_bf16* A,
                 int B, int T, int H, int W, int p, int Kp, void* stream);
int aim_embed_ln(const aim_bf16* tok, const float* cls, const float* pos, const float* temporal,
                 const float* gamma, const float* beta, float* x, float* mean, float* rstd,
                 int B, int T, int N, int D, float eps, void* stream);
int aim_embed_bwd(const void* dx, int dx_is_bf16, const aim_bf16* tok, const float* cls, const float* pos,
                  const float* temporal, const float* gamma, const float* mean, const float* rstd,
                  float* dtemporal, int B, int T, int N, int D,
                  float* workspace /* optional scratch: two-stage, bitwise reproducible reduction */, int64_t workspace_bytes,
                  void* stream);
int64_t aim_embed_bwd_workspace_bytes(int B, int T, int N, int D);

/* ------------------------------------------------------------------------------------------
 * Small reductions / casts used by the block's backward and the optimizer boundary.
 *   frame_sum : out[frame][d] = sum_tok w[tok] * x[frame*ntok + tok][d]   (x f32, w may be NULL)
 *   colsum    : out[c] += sum_m rs(m) * X[m][c]   (X bf16; rs as in the GEMM; two-stage ordered reduction through
 *               the workspace, fp32 atomics only without one)
 *   cast      : f32 -> bf16 (optionally transposed [R,C] -> [C,R]) for weight staging
 *   scale_rows: y[r][c] = s[r] * x[r][c]  (f32 x -> bf16 y), used for lamda * crs_attn
 * ------------------------------------------------------------------------------------------ */
int aim_frame_sum(const void* x, int x_is_bf16, const float* w, float* out, int frames, int ntok, int D, void* stream);
int aim_colsum_bf16(const aim_bf16* X, int ldx, const float* af, const float* at, int ntok,
                    float* out, int M, int C, float* workspace /* optional: >= 1024*C floats -> two-stage, no atomics */,
                    int64_t workspace_bytes, void* stream);
int aim_cast_bf16(const float* src, aim_bf16* dst, int R, int C, int transpose, int ldd /* dst row stride, 0 = dense */,
                  void* stream);
int aim_scale_rows(const float* x, const float* s, aim_bf16* y, float* y_f32, int R, int C, void* stream);
/* dst[r * dst_row_stride + c] += src[r * C + c] (bf16 += fp32): folds the class rows' gradient share, computed apart on
 * the side stream, into a token-major tensor (dst_row_stride = N * D selects the class rows). */
int aim_add_rows_bf16(aim_bf16* dst, int64_t dst_row_stride, const float* src, int R, int C, void* stream);

/* A table of casts in ONE launch: dst = bf16(src) or bf16(src^T) with row stride ldd, for the ~150 small
 * adapter tensors that must be re-staged as bf16 GEMM operands after every optimizer step.  The table
 * lives in DEVICE memory (built once; the pointers are stable). */
typedef struct aim_cast_desc {
    const float* src;   /* [R, C] dense fp32 */
    void* dst;          /* bf16 [R, ldd] or, transposed, [C, ldd]; transpose == 2: fp32 [R, ldd] plain copy */
    int32_t R, C, ldd, transpose;
} aim_cast_desc;
int aim_cast_multi(const aim_cast_desc* table_dev, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer boundary: AdamW (torch.optim.AdamW semantics, decoupled decay) on ONE flat fp32 buffer
 * per parameter group -- the reference steps 149 small tensors through mmcv's optimizer hook
 * (mmaction/utils/optimizer.py:22-33, configs/recognition/vit/vitclip_base_k400.py:96-102).
 * `step` is the 1-based update count (bias correction).  Pointers 16-byte aligned.
 * `grad_scale` multiplies g on the fly: 1 / world_size turns the SUM all-reduce of the flat gradient buffer into
 * DDP's mean (mmaction/apis/train.py:106-110) without a separate pass over the buffer.
 * ------------------------------------------------------------------------------------------ */
int aim_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Classification tail -- the callers right after the backbone:
 *   head_fwd : I3DHead.forward (mmaction/models/heads/i3d_head.py:53-73): adaptive average pool over the T frames of
 *              feat [B, T, D] f32 (the backbone's [B, D, T, 1, 1] output stored frame-major), optional dropout as a
 *              caller-drawn factor table drop[B, D] (0 or 1/(1-p); NULL = eval), fc_cls:
 *              pooled[b][d] = drop[b][d] * mean_t feat[b][t][d]  (saved for backward) ; score = pooled W^T + bias.
 *   head_bwd : dW[C, D] += dscore^T pooled ; db[C] += colsum(dscore) ; dfeat[b][t][d] = drop[b][d]/T * (dscore W)[b][d].
 *   ce_topk  : CrossEntropyLoss hard-label path (mmaction/models/losses/cross_entropy_loss.py:78) + top-k accuracy
 *              (mmaction/core/evaluation/accuracy.py:90-109, numpy argsort tie order: a label is in the top k iff
 *              fewer than k classes score higher or tie with a larger index), all on the device:
 *              out3 = {mean over VALID b of -log softmax(score)[label], top1, top5} ; dscore[B, C] = (softmax - onehot) / n_valid.
 *              A label outside [0, C) is ignored exactly like F.cross_entropy's ignore_index (-100): no loss term, a zero
 *              dscore row, not counted in the CE mean (top1 / top5 keep the denominator B: such a sample is a miss).
 *              One workgroup per sample writes per-sample terms, the last pass sums them in sample order (no atomics).
 * ------------------------------------------------------------------------------------------ */
int aim_head_fwd(const float* feat, const float* drop, const float* W, const float* bias, float* pooled, float* score,
                 int B, int T, int D, int C, void* stream);
int aim_head_bwd(const float* dscore, const float* pooled, const float* drop, const float* W, float* dW, float* db,
                 float* dfeat, int B, int T, int D, int C, void* stream);
int aim_ce_topk(const float* score, const int64_t* label, float* dscore, float* per_sample /* [B, 4] scratch */,
                float* out3, int B, int C, int k2 /* second k of the accuracy pair, 5 */, void* stream);

/* ------------------------------------------------------------------------------------------
 * Reference-precision (fp32) path: `ViT_CLIP.set_precision('fp32')`, forward AND backward.  The product path
 * computes on bf16 MFMA operands; these entry points compute the SAME block from fp32 operands with fp32 accumulation
 * (v_mfma_f32_16x16x4_f32: exact f32 products, 1/16 of the bf16 rate), exact erf / exp, probabilities normalised before
 * P V -- the arithmetic of the reference's un-autocast run (vit_clip.py:433-458), so the GPU output is held DIRECTLY to
 * the reference's fixtures at 1e-5 (BASELINE north_star), with no bf16-emulating oracle in between.
 *   gemm_f32       : C = A W^T with A, W, bias, resid, out all f32 (aim_gemm_args with float operands); epilogues
 *                    AIM_EPI_BF16 (read: linear, out(f32) = rs * (acc + bias)), AIM_EPI_ACT (out(f32) = [rs *] act(acc + bias),
 *                    n_split / act2 as above; out2 != NULL receives the f32 pre-activation acc + bias, ldo2 floats per row),
 *                    AIM_EPI_DACT (out = [rs *] acc * act'(aux): aux = that f32 pre-activation, ldaux floats per row; exact
 *                    erf / exp derivative; aux_grad / aux_frag are ignored), AIM_EPI_F32; batch > 1 (linear only) writes
 *                    item z at out + z * M * ldo.  K, lda, ldw multiples of 4.  replaces vit_clip.py:93-97,132-138,157,286,436.
 *   attn_fwd_f32   : softmax(q k^T / 8) v per (frame, head) on the fused f32 qkv rows [BT*N, 3D]; out [BT*N, D].  :139-156
 *   cls_attn_fwd_f32 : the same over the T class tokens of each clip (sequence T, batch B); qkv rows of the class tokens
 *                    are `row_stride` floats apart (N * 3D in the fused buffer); out [B*T, D].  :220-229
 *   lambda_f32     : scores [BT][N][lds] = RAW full-width q_i . k_j (a batched gemm_f32 of q against k); kx [BT, ldkx] the
 *                    cross-attention's single key; lam = cw / (cw + ow) with ow = sum_ij exp(scale s_ij), cw = sum_i
 *                    exp(scale q_i . kx), one shared max shift.  :149-151,184-186,272
 *   patchify_f32 / embed_ln_f32 : aim_patchify / aim_embed_ln with an f32 patch matrix / f32 tokens (in_dtype 0 | 1).  :434-447
 *                    embed_ln_f32: pre / mean / rstd (all three or none) receive ln_pre's input rows [B*T*N, D] and statistics
 *                    -- what aim_layernorm_bwd needs for the gradient of the trainable temporal_embedding (:344,446).
 * Backward (the reference gets it from torch autograd; these follow autograd's formulas: softmax backward
 * dS = P o (dP - rowsum(P o dP)) then the 1/sqrt(dh) of :147, addmm backward for the Adapter's Linear layers):
 *   attn_bwd_f32   : dqkv [BT*N, 3D] (all three thirds WRITTEN) from qkv and d(out) [BT*N, D]; probabilities recomputed;
 *                    workspace of aim_attn_bwd_f32_workspace_bytes (row log-sum-exp and rowsum(P o dP)).  :139-156
 *   cls_attn_bwd_f32 : d(out_cls) [B*T, D] -> ADDED into the class rows of dqkv (row_stride floats apart), which already
 *                    hold the spatial attention's share (call after attn_bwd_f32).  :220-229
 *   wgrad_f32      : dW [Nw, Kw] += sum_m G[m][n] A[m][k]; db [Nw] += sum_m at[m % ntok] G[m][n] (at NULL -> 1; db NULL -> no
 *                    bias); G [M, ldg], A [M, lda] f32.  M is reduced in chunks whose partial results are summed in chunk
 *                    order (workspace of aim_wgrad_f32_workspace_bytes; no atomics).  Adapter.D_fc1 / D_fc2, :57-58,62-64
 * LayerNorm backward, the per-frame sums and the row scaling of the fp32 backward are the f32 forms of aim_layernorm_bwd,
 * aim_frame_sum and aim_scale_rows above.
 * ------------------------------------------------------------------------------------------ */
int aim_gemm_f32(const aim_gemm_args* args, int epilogue, int batch, void* stream);
int aim_attn_fwd_f32(const float* qkv, float* out, int BT, int N, int H, void* stream);
int aim_cls_attn_fwd_f32(const float* qkv, int64_t row_stride, float* out_cls, int B, int T, int H, void* stream);
int aim_lambda_f32(const float* scores, int lds, const float* qkv, const float* kx, int ldkx, float* lam,
                   float* one_minus_lam, int BT, int N, int D, float scale, void* stream);
int aim_patchify_f32(const void* imgs, int in_dtype, const float* mean3, const float* std3, float* A, int B, int T,
                     int H, int W, int p, int Kp, void* stream);
int aim_embed_ln_f32(const float* tok, const float* cls, const float* pos, const float* temporal, const float* gamma,
                     const float* beta, float* x, float* pre, float* mean, float* rstd, int B, int T, int N, int D, float eps,
                     void* stream);
int64_t aim_attn_bwd_f32_workspace_bytes(int BT, int N, int H);
int aim_attn_bwd_f32(const float* qkv, const float* dout, float* dqkv, int BT, int N, int H, float* workspace,
                     int64_t workspace_bytes, void* stream);
int aim_cls_attn_bwd_f32(const float* qkv, int64_t row_stride, const float* dout_cls, float* dqkv, int B, int T, int H,
                         void* stream);
/* stock-AIM block (vitclip_aim.py:199-204): attention over the T frames of EVERY token, on the frame-major fused f32 qkv rows
   [B*T*N, 3D]; out / dout [B*T*N, D]; the backward ADDS into dqkv (zero it first when nothing else wrote it) */
int aim_tattn_fwd_f32(const float* qkv, float* out, int B, int T, int N, int H, void* stream);
int aim_tattn_bwd_f32(const float* qkv, const float* dout, float* dqkv, int B, int T, int N, int H, void* stream);
int64_t aim_wgrad_f32_workspace_bytes(int M, int Nw, int Kw);
int aim_wgrad_f32(const float* G, int ldg, const float* A, int lda, float* dW, int M, int Nw, int Kw, float* db,
                  const float* at, int ntok, float* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AIM_KERNELS_H */
