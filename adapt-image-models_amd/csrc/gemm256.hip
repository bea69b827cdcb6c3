// bf16 MFMA GEMM, 256x256x64 block tile, 8 waves, software-pipelined LDS-DMA staging with counted
// vmcnt.  gfx950 only.  Same contract and epilogues as gemm.hip (C = A W^T, both K-contiguous); this is
// the kernel the large-M GEMMs of the AIM block run on (QKV / out_proj / c_fc / c_proj / adapter
// projections and their dgrads: reference vit_clip.py:93-97,132-138,157,286 and autograd).
//
// Geometry: 8 waves as 2(M) x 4(N); a wave owns a 128x64 output tile = 8x4 MFMA 16x16x32 tiles
// (128 accumulator VGPRs).  LDS = 2 K-tile buffers x {A_lo, A_hi, B_lo, B_hi} half-tiles of
// 128 rows x 64 k (16 KiB each, XOR-swizzled image of aim_common.h) = 128 KiB, one block per CU.
//
// Schedule: one loop iteration = 2 K-tiles (even -> buffer 0, odd -> buffer 1) = 4 phases.  Each phase
//   { fragment ds_reads ; stage TWO half-tiles (4 buffer_load...lds per wave) ; [s_waitcnt vmcnt(4)] ;
//     s_waitcnt lgkmcnt(0) ; s_barrier ; 32 MFMA ; s_barrier }.
// The two wave groups (wm = 0 | 1: one wave of each on every SIMD) run ONE BARRIER apart, so while one
// group issues its 32 MFMAs the other reads fragments and issues LDS-DMA: the matrix pipe, the LDS and
// the vector-memory issue work together instead of in turns.  (Measured per K-step at K=3072, shader
// cycles: 8 lock-step phases 4000, 8 skewed phases 3300, 4 skewed phases 2700; MFMA floor 2048.  An
// LDS-DMA piece costs the issuing wave ~100 cycles, so the read phase must not be longer than 32 MFMAs.)
// A wave reads all its B fragments and the first 64-row A sub-block in phase a (c), the second A
// sub-block in phase b (d); staging order and the RAW / WAR argument are spelled out at the loop.
// Tiles past K are staged with out-of-range offsets (zero fill, still counted by vmcnt) so the counts
// are uniform.  The skew is given back before each tile's epilogue so both groups' epilogues overlap.
//
// The kernel is PERSISTENT (one workgroup per CU walks the tiles): the look-ahead of the staging schedule runs
// across tile boundaries, so only the very first tile of a workgroup pays a prologue, and the epilogue
// (re-tiled through a wave-private 2 KiB LDS scratch beside the images) overlaps the next tile's loads.
#include "aim_common.h"
#include "aim_kernels_internal.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

namespace {

constexpr int HT = 128 * 64 * 2;        // half-tile bytes
constexpr int BUF = 4 * HT;             // one K-tile buffer: A_lo, A_hi, B_lo, B_hi
constexpr int OFF_A = 0, OFF_B = 2 * HT;
constexpr int LDS_BYTES = 2 * BUF + 8 * EPI_SCRATCH;   // K-loop images + 8 wave-private row-factor tables

// Per-tile staging state: buffer descriptors of the tile's A rows / W rows and the per-lane row offsets.
struct TileSrc {
    __amdgpu_buffer_rsrc_t rA, rW, rX;      // rX: the item's extra W row (EXPSUM `xrow`), else empty
    unsigned voA[2][2], voW[2][2];   // [half][piece]; AIM_OOB for rows past M / N
    int m0, n0, z;
};

// A batched problem (strideA / strideW between items) is walked as extra row tiles: virtual row tile tmv = z * tiles_m1 + tm.
// ES = operand element size in bytes (2: bf16, 1: fp8 e4m3); strides and K are in elements
template <int ES>
__device__ __forceinline__ TileSrc make_tile(const GemmArgs& g, int tile, int ntiles, int tiles_n, int tiles_m1, int wave,
                                             int srow, int schunk) {
    TileSrc t;
    const bool live = tile < ntiles;
    const int tn = live ? tile % tiles_n : 0, tmv = live ? tile / tiles_n : 0;
    const int z = tmv / tiles_m1, tm = tmv - z * tiles_m1;
    t.z = z;
    t.m0 = tm * 256;
    t.n0 = tn * 256;
    const int rowsA = live ? g.M - t.m0 : 0, rowsW = live ? g.N - t.n0 : 0;
    const char* Ab = (const char*)g.A + ((long long)z * g.strideA + (long long)t.m0 * g.lda) * ES;
    const char* Wb = (const char*)g.W + ((long long)z * g.strideW + (long long)t.n0 * g.ldw) * ES;
    t.rA = make_rsrc(Ab, live ? ((long long)(rowsA - 1) * g.lda + g.K) * ES : 0);
    t.rW = make_rsrc(Wb, live ? ((long long)(rowsW - 1) * g.ldw + g.K) * ES : 0);
    t.rX = make_rsrc(g.xrow ? (const bf16_t*)g.xrow + (long long)z * g.ldx : nullptr, (live && g.xrow) ? (long long)g.K * 2 : 0);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = h * 128 + (wave * 2 + j) * 8 + srow;
            t.voA[h][j] = r < rowsA ? (unsigned)(r * g.lda * ES + schunk * 16) : AIM_OOB;
            t.voW[h][j] = r < rowsW ? (unsigned)(r * g.ldw * ES + schunk * 16) : AIM_OOB;
        }
    return t;
}

// Persistent: gridDim.x workgroups (one per CU) walk the output tiles  tile = logical_id + i * gridDim.x.
// The staging pipeline never drains at a tile boundary: a tile occupies nkp = 2*ceil(nk/2) K-slots and
// the schedule's look-ahead (up to 3 K-tiles) simply runs into the NEXT tile's first K-tiles, so its
// prologue latency is hidden behind this tile's last MFMAs and its epilogue.
// F8: operands are fp8 e4m3 bytes; a K-tile is still one 128-byte LDS row per operand row (128 elements instead of 64),
// staged and read exactly like the bf16 image, and multiplied by ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 16x16 tile
// (unit block scales): the lane's two 16-byte chunks (ks = 0, 1) are its 32 k-values of that instruction -- any
// assignment of k to lanes works as long as both operands use the same one (tools/probe_mfma_fp8.hip).
template <int EPI, bool F8>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs g, int ntiles, int nbatch, int phase_skew,
                                                      unsigned long long* probe, int probe_cap) {
    constexpr int ES = F8 ? 1 : 2;                // operand element bytes
    constexpr int KT = 128 / ES;                  // elements per K-tile (one 128-byte LDS row)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    AIM_LDS char* smem = (AIM_LDS char*)smem_raw;
    AIM_LDS char* escr = smem + 2 * BUF;          // epilogue scratch lives beside the K-loop images

    const int tiles_n = (g.N + 255) >> 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 15, fq = lane >> 4;
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    const int nk = (g.K + KT - 1) / KT;
    // EXPSUM with an extra key: the wave whose W pieces hold tile row g.N issues one more LDS-DMA per W stage pair,
    // so its counted waits leave 5, not 4, operations in flight
    const bool xw = EPI == EPI_EXPSUM && g.xrow != nullptr && wave == ((g.N & 127) >> 4);
    const int nkp = (nk + 1) & ~1;               // K-slots per tile (even)

    // (the batched exp-sum GEMM reads what the QKV GEMM has just written: it walks its tiles -- frames -- from the last one
    //  down, like the other consumers of a just-written tensor: aim_common.h, AIM_REV_BLOCK; +0.2 % whole step)
    constexpr bool walk_down = EPI == EPI_EXPSUM;
    // ---- tile schedule ---------------------------------------------------------------------------
    // Blocks are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD; speed only, never correctness).  The
    // workgroups of an XCD take CONTIGUOUS ranks, so at any moment an XCD's CUs work on neighbouring tiles of the row-major
    // tile sequence (shared A rows / W columns in their L2):  tile = rank + i * gridDim.x.  (Column groups per XCD -- a
    // weight slice resident in each L2 -- were measured twice and removed: the re-fetched weights are Infinity-Cache hits.)
    const int rank = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3)
                                          : xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tiles_m1 = (g.M + 255) >> 8;
    const int wg_per_group = (int)gridDim.x;
    const int nseq = ntiles;
    auto seq_tile = [&](int seq) {
        if (seq >= ntiles) return ntiles;
        return walk_down ? ntiles - 1 - seq : seq;
    };
    int seq = rank;
    TileSrc cur = make_tile<ES>(g, seq_tile(seq), ntiles, tiles_n, tiles_m1, wave, srow, schunk);
    // The next tile is kept as an index only: its descriptors and per-lane row offsets (8 VGPRs) are built in this tile's last
    // K-iteration, when the staging state moves on to it -- held through the whole K-loop they pushed the kernel over 256 VGPRs.
    int nxt_tile = seq_tile(seq + wg_per_group);
    int nm0 = 0, nn0 = 0, nz = 0;
    // (fp8 keeps the first scheme -- the next tile's staging state precomputed, the tile of a slot decided per piece: with its
    //  8-register fragment tuples the just-in-time form spills 125 VGPRs and the inference GEMMs lose 7 %)
    TileSrc nxt8 = cur;
    if constexpr (F8) nxt8 = make_tile<ES>(g, nxt_tile, ntiles, tiles_n, tiles_m1, wave, srow, schunk);

    // which: 0 A_lo, 1 A_hi, 2 B_lo, 3 B_hi.  One LDS-DMA piece costs ONE v_add: the K offset of the slot is added to the piece's
    // row offset.  A row past M / N keeps an out-of-range offset (AIM_OOB + k0 < 2^32 stays past num_records); the K tail of a
    // row is only masked when K is not a multiple of the K-tile (ktail, fp8 only: otherwise the chunk past K would read the next row).
    // Stages that run into the NEXT tile (the look-ahead is at most two K-tiles: only in a tile's last K-iteration, after its
    // last own stage) find the next tile's descriptors and row offsets already moved into `cur` -- the first version decided
    // per piece which tile a slot belongs to (six VALU + two SALU instructions each, ~100 vector instructions per iteration
    // beside the MFMAs).
    constexpr bool ktail = false;          // the dispatcher sends K % 64 != 0 to the 128x128 kernel (fp8 masks per piece, below)
    auto stage_from = [&](const __amdgpu_buffer_rsrc_t& rA_, const __amdgpu_buffer_rsrc_t& rW_, const __amdgpu_buffer_rsrc_t& rX_,
                          const unsigned (&voA_)[2][2], const unsigned (&voW_)[2][2], int buf, int which, int kt) {
#ifdef AIM_X_NOSTAGE
        return;
#endif
        const unsigned k0b = (unsigned)(kt * KT * ES);
        AIM_LDS char* dst = smem + buf * BUF + which * HT + wave * 2048;
        const int h = which & 1;
        const bool kout = ktail && (kt * KT + schunk * (16 / ES)) >= g.K;      // (lane-dependent only when ktail)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (which < 2) {
                stage_piece(rA_, dst + j * 1024, kout ? AIM_OOB : voA_[h][j] + k0b);
            } else {
                stage_piece(rW_, dst + j * 1024, kout ? AIM_OOB : voW_[h][j] + k0b);
                if constexpr (EPI == EPI_EXPSUM) {
                    // the extra key: tile-local W row g.N comes from `xrow`.  Only the 8 lanes of that row take part
                    // (EXEC-masked LDS-DMA writes only its active lanes' 16-byte slots), over the zero the piece above
                    // left there; the wave that owns the row has ONE more vector-memory op per W stage (xw below).
                    if (g.xrow && h * 128 + (wave * 2 + j) * 8 + srow == g.N)
                        stage_piece(rX_, dst + j * 1024, kout ? AIM_OOB : (unsigned)(schunk * 16) + k0b);
                }
            }
        }
    };
    // fp8: slot counts K-tiles from the start of the CURRENT tile; slots >= nkp belong to the next one
    auto stage8 = [&](int buf, int which, int slot) {
#ifdef AIM_X_NOSTAGE
        return;
#endif
        const bool in_next = slot >= nkp;
        const int kt = in_next ? slot - nkp : slot;
        const int k0 = kt * KT;
        const bool kin = (k0 + schunk * (16 / ES)) < g.K;
        AIM_LDS char* dst = smem + buf * BUF + which * HT + wave * 2048;
        const int h = which & 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned vo = which < 2 ? (in_next ? nxt8.voA[h][j] : cur.voA[h][j]) : (in_next ? nxt8.voW[h][j] : cur.voW[h][j]);
            const unsigned v = (kin && vo != AIM_OOB) ? vo + (unsigned)(k0 * ES) : AIM_OOB;
            if (which < 2) stage_piece(in_next ? nxt8.rA : cur.rA, dst + j * 1024, v);
            else stage_piece(in_next ? nxt8.rW : cur.rW, dst + j * 1024, v);
        }
    };
    // slot `slot` of the current tile (fp8) = K-tile `kt` of the tile whose staging state `cur` holds (bf16)
    auto stage = [&](int buf, int which, int slot, int kt) {
        if constexpr (F8) stage8(buf, which, slot);
        else stage_from(cur.rA, cur.rW, cur.rX, cur.voA, cur.voW, buf, which, kt);
    };

    f32x4 acc[8][4];
    // fragments: bf16 -- two 16-byte k-substeps per row tile; fp8 -- ONE 32-byte operand of the K = 128 MFMA, held as an
    // 8-register tuple that the two ds_read_b128 fill in place (building it from two separate 4-register values costs
    // copies and, at this register budget, spills)
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    bf16x8 af[4][2] = {}, bfr[4][2] = {};
    i32x8 af8[4] = {}, bf8[4] = {};

    auto read_b = [&](int buf) {
#ifdef AIM_X_NOLDS
        return;
#endif
        const AIM_LDS char* sB = smem + buf * BUF + OFF_B + (wn >> 1) * HT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (F8) {
                bf8[j].lo = *(const AIM_LDS i32x4*)(sB + swz_off((wn & 1) * 64 + j * 16 + frow, fq));
                bf8[j].hi = *(const AIM_LDS i32x4*)(sB + swz_off((wn & 1) * 64 + j * 16 + frow, 4 + fq));
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bfr[j][ks] = lds_read8(sB + swz_off((wn & 1) * 64 + j * 16 + frow, ks * 4 + fq));
            }
        }
    };
    auto read_a = [&](int buf, int sub) {
#ifdef AIM_X_NOLDS
        return;
#endif
        const AIM_LDS char* sA = smem + buf * BUF + OFF_A + wm * HT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (F8) {
                af8[i].lo = *(const AIM_LDS i32x4*)(sA + swz_off(sub * 64 + i * 16 + frow, fq));
                af8[i].hi = *(const AIM_LDS i32x4*)(sA + swz_off(sub * 64 + i * 16 + frow, 4 + fq));
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af[i][ks] = lds_read8(sA + swz_off(sub * 64 + i * 16 + frow, ks * 4 + fq));
            }
        }
    };
    // 16 MFMA: A sub-block `sub` (4 m-tiles) x B tiles {jb, jb+1} x 2 k-substeps
    auto mma = [&](int sub, int jb) {
#ifdef AIM_X_NOMFMA
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(af[i][ks]));
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(bfr[jb + j][ks]));
        }
        return;
#endif
        __builtin_amdgcn_s_setprio(1);
        if constexpr (F8) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    // cbsz = blgp = 0: both operands fp8 e4m3; scale bytes 0x7F = 2^0
                    acc[sub * 4 + i][jb + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        bf8[jb + j], af8[i], acc[sub * 4 + i][jb + j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[sub * 4 + i][jb + j] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[jb + j][ks], af[i][ks], acc[sub * 4 + i][jb + j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
#define AIM_BAR() __builtin_amdgcn_s_barrier()
// s_waitcnt as builtins (simm16: vmcnt[3:0] | expcnt<<4 | lgkmcnt<<8 | vmcnt[5:4]<<14) so the compiler's own scoreboard
// sees them.  hipcc still flushes `vmcnt(0)` at the K-loop header while LDS-DMA is in flight (it is gone only when the
// stage calls are compiled out); stamped, that costs ~40 of an iteration's ~5500 cycles.
#define AIM_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0)
#define AIM_VM4() do { if (xw) __builtin_amdgcn_s_waitcnt(0x0F75); else __builtin_amdgcn_s_waitcnt(0x0F74); } while (0)

    // prologue (first tile only): K-tile 0 complete, three half-tiles of K-tile 1 in flight
    stage(0, 2, 0, 0); stage(0, 3, 0, 0); stage(0, 0, 0, 0); stage(0, 1, 0, 0);
    stage(1, 2, 1, 1); stage(1, 3, 1, 1);
    AIM_VM4();
    AIM_BAR();
    // The two wave groups (wm = 0 | 1; one wave of each per SIMD) run ONE BARRIER apart: while one group issues its
    // phase's MFMAs the other reads its fragments / issues its stages, so the matrix pipe and the LDS are busy
    // together instead of in turns.  RAW is safe because a buffer is read a full phase after the counted vmcnt +
    // barrier that retires it; WAR because B (read by both groups) is waited for before the reading phase's barrier
    // and A halves are private to one group.
    // The skew is taken at the top of every tile and given back before its epilogue, so both groups' epilogues
    // run together (skewed epilogues serialise: the leading group would wait at its next barrier for the other's).

#ifdef AIM_X_STAMPS      // diagnostic build only (tools/probe_gemm.py STAMPS=1): shader-clock stamps of one iteration
    unsigned long long stamps[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) stamps[i] = 0;
    int stamp_arm = 0;
#define AIM_STAMP(i) do { if (stamp_arm) { __builtin_amdgcn_sched_barrier(0); stamps[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define AIM_STAMP(i)
#endif
    int probe_i = 0;
    for (; seq < nseq; seq += wg_per_group) {
        unsigned long long tp0 = 0, tp1 = 0, tc0 = 0, tc1 = 0;
        if (probe) { tp0 = __builtin_amdgcn_s_memrealtime(); tc0 = __builtin_amdgcn_s_memtime(); }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (phase_skew && wm == 1) AIM_BAR();
        // One iteration = two K-tiles (even -> buffer 0, odd -> buffer 1) = four phases of
        //   [ds_read fragments + 2 stage calls (4 LDS-DMA pieces)] ; barrier ; [32 MFMA] ; barrier.
        // Stage order (each half goes out as soon as both wave groups are done reading it):
        //   Ra: A halves of the odd tile `to`      Rb: B halves of even tile te+2   + vmcnt(4): odd tile landed
        //   Rc: A halves of even tile te+2         Rd: B halves of odd tile to+2    + vmcnt(4): even tile te+2 landed
        // RAW: a buffer is read in the phase AFTER the counted vmcnt + barrier that retires it (one more barrier
        //      than the wave-group skew).  WAR: every read phase retires its ds_reads (lgkmcnt(0)) BEFORE its
        //      barrier, and a half is restaged no earlier than the second phase after its last read.
        // Slots >= nkp run into the NEXT tile's K-tiles (cross-tile prefetch).
        for (int it = 0; it < nkp / 2; ++it) {
            const int te = 2 * it, to = te + 1;
            // the K-tiles te+2 / to+2 staged in this iteration belong to the next tile in the last iteration (its tiles 0 / 1)
            const bool lastit = it == nkp / 2 - 1;
            const int kt_e = lastit ? 0 : te + 2, kt_o = lastit ? 1 : to + 2;
            // fp8: the launcher guarantees an even number of K-tiles, so no fragment read / MFMA is conditional (a
            // conditionally refilled 8-register fragment tuple costs copies and spills)
            const bool odd_live = F8 ? true : (to < nk);
#ifdef AIM_X_STAMPS
            stamp_arm = (probe != nullptr) && blockIdx.x == 0 && probe_i == 1 && it == 2;
#endif
            AIM_STAMP(0);
            read_b(0); read_a(0, 0);                               // ---- Ra
            stage(1, 0, to, to); stage(1, 1, to, to);
            if (!F8 && lastit) {    // this tile's last own stage is out: the staging state moves on to the next tile
                // (laundered lane: otherwise the tile-independent part of the eight row offsets is hoisted out of the tile loop,
                //  eight more live VGPRs, and the F32 kernel reloads spilled ones here behind an s_waitcnt vmcnt(0))
                int ll = lane;
                asm volatile("" : "+v"(ll));
                const TileSrc t = make_tile<ES>(g, nxt_tile, ntiles, tiles_n, tiles_m1, wave, ll >> 3, (ll & 7) ^ (ll >> 3));
                cur.rA = t.rA;
                cur.rW = t.rW;
                cur.rX = t.rX;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        cur.voA[h][j] = t.voA[h][j];
                        cur.voW[h][j] = t.voW[h][j];
                    }
                nm0 = t.m0;         // (the epilogue below still needs THIS tile's coordinates in cur.m0 / n0 / z)
                nn0 = t.n0;
                nz = t.z;
            }
            AIM_LGKM0();
            AIM_STAMP(1); AIM_BAR(); AIM_STAMP(2);
            mma(0, 0); mma(0, 2);
            AIM_STAMP(3); AIM_BAR(); AIM_STAMP(4);
            read_a(0, 1);                                          // ---- Rb
            stage(0, 2, te + 2, kt_e); stage(0, 3, te + 2, kt_e);
            AIM_VM4();
            AIM_LGKM0();
            AIM_STAMP(5); AIM_BAR(); AIM_STAMP(6);
            mma(1, 2); mma(1, 0);
            AIM_STAMP(7); AIM_BAR(); AIM_STAMP(8);
            if (odd_live) { read_b(1); read_a(1, 0); }             // ---- Rc
            stage(0, 0, te + 2, kt_e); stage(0, 1, te + 2, kt_e);
            AIM_LGKM0();
            AIM_STAMP(9); AIM_BAR(); AIM_STAMP(10);
            if (odd_live) { mma(0, 0); mma(0, 2); }
            AIM_STAMP(11); AIM_BAR(); AIM_STAMP(12);
            if (odd_live) read_a(1, 1);                            // ---- Rd
            stage(1, 2, to + 2, kt_o); stage(1, 3, to + 2, kt_o);
            AIM_VM4();
            AIM_LGKM0();
            AIM_STAMP(13); AIM_BAR(); AIM_STAMP(14);
            if (odd_live) { mma(1, 2); mma(1, 0); }
            AIM_STAMP(15); AIM_BAR(); AIM_STAMP(16);
        }
#ifdef AIM_X_STAMPS
        if (probe && blockIdx.x == 0 && probe_i == 1 && lane == 0 && (wave == 0 || wave == 4)) {
            unsigned long long* sp = probe + (long long)(probe_cap - 32) * 4 + (wave == 4 ? 64 : 0);
#pragma unroll
            for (int i = 0; i < 17; ++i) sp[i] = stamps[i];
        }
#endif
        // Epilogue of this tile; the next tile's first K-tiles are already in flight / landed in the K-loop
        // images, so the scratch is separate (wave-private, no barrier needed).
        if (phase_skew && wm == 0) AIM_BAR();
        if (probe) { tp1 = __builtin_amdgcn_s_memrealtime(); tc1 = __builtin_amdgcn_s_memtime(); }
        if constexpr (EPI == EPI_EXPSUM)
            // (max, sum) slots of a tile: 8 (16 with an extra key) pairs; g.ldo > 0 sets the float stride between tiles, so a
            //  caller can leave room for more slots behind them (aim_qk_border's)
            wave_expsum(g, acc, cur.m0 + wm * 128, cur.n0 + wn * 64, lane,
                        (float*)g.out + ((long long)(cur.z * tiles_m1 + (cur.m0 >> 8)) * tiles_n + (cur.n0 >> 8)) *
                                            (g.ldo > 0 ? g.ldo : (g.xrow ? 32 : 16)) + wave * 2);
        else
            wave_epilogue<EPI, F8>(g, acc, escr + wave * EPI_SCRATCH, cur.m0 + wm * 128, cur.n0 + wn * 64, lane);
        if (probe) {        // diagnostics (aim_gemm_args.probe): per-tile timestamps of wave 0, 100 MHz ticks
            const int slot = probe_i * (int)gridDim.x + (int)blockIdx.x;
            if (tid == 0 && slot < probe_cap) {
                probe[slot * 4 + 0] = (unsigned long long)blockIdx.x | ((tc1 - tc0) << 16);   // K-loop shader cycles
                probe[slot * 4 + 1] = tp0;
                probe[slot * 4 + 2] = tp1;
                probe[slot * 4 + 3] = __builtin_amdgcn_s_memrealtime();
            }
            ++probe_i;
        }
        if constexpr (F8) {
            cur = nxt8;
            nxt8 = make_tile<ES>(g, seq_tile(seq + 2 * wg_per_group), ntiles, tiles_n, tiles_m1, wave, srow, schunk);
        } else {
            cur.m0 = nm0;
            cur.n0 = nn0;
            cur.z = nz;
            nxt_tile = seq_tile(seq + 2 * wg_per_group);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the zero-fill stages of the tail
}

template <int EPI, bool F8 = false>
int launch256(const GemmArgs& g, int nbatch, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<EPI, F8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256) * nbatch;
    // a persistent grid that fills every CU starves whatever runs beside it on another stream: the caller can keep a few
    // CUs out of the grid while such work is in flight
    const int reserve = g.reserve_cus > 0 ? g.reserve_cus : 0;       // per-call knob (aim_gemm_args.reserve_cus)
    const int ncu = aim_device_cus();
    const int cus = ncu - reserve > 8 ? ncu - reserve : 8;
    int grid = tiles < cus ? tiles : cus;
    // balanced grid: the fewest workgroups that still finish in ceil(tiles / cus) rounds.  1 182 tiles take 5 rounds on 256
    // CUs and on 237: the 19 CUs that would idle through the last round are free for the other streams for the whole
    // launch instead.
    static const bool balanced = [] { const char* e = getenv("AIM_GEMM_BALANCED"); return !e || atoi(e) != 0; }();
    if (balanced && tiles > cus) {
        const int rounds = (tiles + cus - 1) / cus;
        const int need = (tiles + rounds - 1) / rounds;
        if (need < grid) grid = need;       // any size: without a multiple of 8 the kernel ranks its blocks by xcd_remap
    }
    static const int phase_skew = [] { const char* e = getenv("AIM_GEMM_SKEW"); return e ? atoi(e) : 1; }();
    hipLaunchKernelGGL((gemm256_kernel<EPI, F8>), dim3(grid), dim3(512), LDS_BYTES, st, g, tiles, nbatch, phase_skew,
                       (unsigned long long*)g.probe, g.probe ? g.probe_cap : 0);
    AIM_CHECK_LAUNCH("aim_gemm_bf16(256)");
    return 0;
}

}  // namespace

int aim_gemm256_fp8_launch(const GemmArgs& g, int epi, hipStream_t st) {
    AIM_CHECK_ARG((long long)256 * g.lda < 0x7fffffffLL && (long long)256 * g.ldw < 0x7fffffffLL, "gemm_fp8: leading dimension too large");
    switch (epi) {
        case EPI_BF16: return launch256<EPI_BF16, true>(g, 1, st);
        case EPI_F32: return launch256<EPI_F32, true>(g, 1, st);
        case EPI_ACT8: return launch256<EPI_ACT8, true>(g, 1, st);
        case EPI_RES16: return launch256<EPI_RES16, true>(g, 1, st);
    }
    aim_set_error("gemm_fp8: unsupported epilogue %d (BF16, F32, ACT8, RES16)", epi);
    return 1;
}

int aim_gemm256_launch(const GemmArgs& g, int epi, int nbatch, hipStream_t st) {
    AIM_CHECK_ARG((long long)256 * g.lda * 2 < 0x7fffffffLL && (long long)256 * g.ldw * 2 < 0x7fffffffLL, "gemm256: leading dimension too large");
    switch (epi) {
        case EPI_BF16: return launch256<EPI_BF16>(g, nbatch, st);
        case EPI_ACT: return launch256<EPI_ACT>(g, nbatch, st);
        case EPI_DACT: return launch256<EPI_DACT>(g, nbatch, st);
        case EPI_F32: return launch256<EPI_F32>(g, nbatch, st);
        case EPI_EXPSUM: return launch256<EPI_EXPSUM>(g, nbatch, st);
    }
    aim_set_error("gemm256: unsupported epilogue %d", epi);
    return 1;
}
