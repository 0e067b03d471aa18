// fxg_kernels.h -- the tile kernels of the engine.
//
// One workgroup (256 threads = 4 wave64) owns a tile of up to 256 consecutive reads.  Tiles are
// handed out by eight interleaved global tickets (so a workgroup only ever waits on tiles whose owners are already running) and each
// workgroup runs a two-stage software pipeline over its tiles:
//   stage A(tile i+1)  1. stream the quality rows once, 16 B per lane, into two LDS bitmaps
//                         (bit = "byte >= trim threshold", bit = "byte < filter threshold")      [HBM read L B/read]
//                      2. one thread per read: (clip DP over LDS-staged bases,) trim point = highest set
//                         bit, filter count = popcount of the trimmed prefix -> res[]            [HBM write 4 B/read]
//                      3. workgroup scan of (keep, new_len); publish the tile's totals (never waits)
//   stage B(tile i)    4. fetch the tile's global offsets: one dedicated workgroup (the scanner, fxg_device.h) turns the
//                         published totals into prefixes in tile order; it has had a whole stage A to do so
//                      5. order-preserving gather of the kept prefixes into the packed output, one 16 B
//                         aligned output chunk per lane                    [HBM read <= 2L, write 2*new_len / kept read]
// The -v report counters are tallied here as the result words go by (fxg_tile_tally: a ballot per drop reason, one LDS add per wave);
// fxg_kernel_finish_counters lays them out.  The clip instances write out two steps behind (three slots) where LDS allows.
#pragma once
#include "fxg_device.h"

#define FXG_INVALID_TUPLE 0xFFFFFFFFu

// LDS carve-up shared by host (size) and device (pointers); every region is 16-byte aligned (G17).
// k_off / k_src / k_idx / k_tab (the tile's kept reads, see fxg_tile_gather) are double buffered: stage B of tile i reads
// slot s while stage A of tile i+1 fills slot s^1 (the clip instances keep three slots: stage B runs two steps behind).
#define FXG_NTALLY 16                            // tally slots: [0] reads seen, [1] kept, [2] kept bases, [3] adapter-only, [4 + reason] dropped for that reason
struct FxgLds {
    u32 slot_bytes, so_ksrc, so_kidx, so_ktab;   // slot k lives at k * slot_bytes: k_off at +0, k_src at +so_ksrc, k_idx at +so_kidx, k_tab at +so_ktab
    u32 off_scratch, off_tally, off_bm_g, off_bm_l, off_bases, off_ptab, total;
    u32 has_tab;                                 // kernels that stage a tile of bases in LDS (the clipper) spend no LDS on k_tab
};
__host__ __device__ inline u32 fxg_r16(u32 x) { return (x + 15u) & ~15u; }
// nslots: tiles a workgroup keeps between decision and write-out (FxgTileDepth: 2, the clip instances 3)
// bitmaps: 0 none, 2 both, 1 = ONE shared by trimmer and filter (same threshold: "below" is the complement of "at least"; off_bm_l == off_bm_g),
//          4 = the base census's four, back to back from off_bm_g (fxg_census_bitmaps)
// ptab_bytes: the clip instances that take their pair values from an LDS table (fxg_clip_ptab_build, fxg_ptab_bytes)
__host__ __device__ inline FxgLds fxg_lds_layout(u32 T, u32 stride, u32 bitmaps, u32 stage_stride, u32 nslots = 2, bool no_tab = false, u32 ptab_bytes = 0u)   // stage_stride: row stride of the LDS copy of the tile's bases, 0 = none
{
    FxgLds l;
    l.so_ksrc = fxg_r16((T + 1) * 4);
    l.so_kidx = l.so_ksrc + fxg_r16(T * 4);
    l.so_ktab = l.so_kidx + fxg_r16(T * 2);
    l.has_tab = (stage_stride || no_tab) ? 0u : 1u;      // (no_tab: the clip instances whose DP reads the batch itself -- the table of a 1 000-base tile would be 32 KB per slot)
    l.slot_bytes = l.so_ktab + (l.has_tab ? fxg_r16(((T * stride + 15) / 16 + 1) * 2) : 0u);
    u32 o = nslots * l.slot_bytes;
    l.off_scratch = o; o += fxg_r16(48 * 4);
    l.off_tally = o;   o += FXG_NTALLY * 8;      // the workgroup's -v report tallies (u64), see fxg_tile_tally
    const u32 words = (T * stride + 31) / 32 + 2;
    l.off_bm_g = o;    o += bitmaps ? fxg_r16(words * 4) * (bitmaps == 4u ? 4u : 1u) : 0;
    l.off_bm_l = bitmaps == 1u ? l.off_bm_g : o;    o += bitmaps == 2u ? fxg_r16(words * 4) : 0;
    l.off_bases = o;   o += stage_stride ? fxg_r16(T * stage_stride + 16) : 0;
    l.off_ptab = o;    o += fxg_r16(ptab_bytes);
    l.total = o;
    return l;
}

// quality bitmaps a launch keeps in LDS (fxg_lds_layout): the clip instances share ONE between trimmer and filter when both use the
// same threshold (the usual pipe) -- their LDS holds a tile of bases as well, and 5 KB decide how many workgroups fit on a CU
FXG_HD u32 fxg_bitmap_count(const FxgKArgs &a, bool use_q, bool clip) { return !use_q ? 0u : (clip && a.tq == a.fq) ? 1u : 2u; }

// ------------------------------------------------------------------------------------------------
// No lane-divergent loops in the DP.  A loop whose trip count differs between the lanes of a wave is left when EXEC = 0, and ROCm 7.2's
// register allocator may put the spill stores / copies of the loop's live-out values into the exit block AHEAD of the instruction that
// brings the lanes back -- where they do nothing (how clip instances came out wrong on the GPU only, for some
// register budgets and not for others).  The DP's loops are the register-hungry ones, so they all run a WAVE-UNIFORM number of trips
// (the maximum over the wave's active lanes, fxg_wave_max) with the row predicated per lane: the loop branch is scalar, EXEC never
// becomes zero at a loop exit, and a lane idles exactly as long as it did in the divergent form.  The ISA check of the build
// (scripts/check_exec_zero.py) stays as the net under this.
// ------------------------------------------------------------------------------------------------
#ifdef FXG_HOST_EMULATION
FXG_HD int fxg_wave_max(int n) { return n; }
#else
// maximum of n >= 0 over the active lanes, as a scalar (ds_bpermute reads 0 from a lane that is not active)
__device__ __forceinline__ int fxg_wave_max(int n)
{
    int v = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return __builtin_amdgcn_readfirstlane(v);
}
#endif

// ------------------------------------------------------------------------------------------------
// fastx_clipper for one read: semi-global fp32 DP of read (query) x adapter (target) with the
// alignment path summary carried forward instead of a traceback matrix.
// Reference: sequence_alignment.cpp:340-428 (borders, cell rule, first-max), :496-604 (traceback
// counts), fastx_clipper.cpp:192-240 (accept rules), :282-319 (clip / discard cascade).
//   w0 = query_start<<16 | target_start<<8 | mismatches        w1 = path_len<<16 | matches
// ------------------------------------------------------------------------------------------------
// accept rules (fastx_clipper.cpp:192-240) + clip/discard cascade (:282-319) on the decoded alignment summary
FXG_HD void fxg_clip_finish(const FxgKArgs &a, int len, int qs, int ts, int mism, int sz, int matches, int bq, int first_n,
                            u32 *out_len, u32 *keep, u32 *reason, u32 *clipped, u32 *adapter_only)
{
    int i = -1;
    if (sz != 0 && !(a.clip_min_adapter_len > 0 && sz < a.clip_min_adapter_len)) {
        // floor(100 m / sz) >= k  <=>  100 m >= k sz  (k an integer, sz > 0): no division
        if (bq == len - 1 && mism == 0) i = qs;
        else if (sz > 5 && ts == 0 && matches * 100 >= 75 * sz) i = qs;
        else if (sz > 11 && matches * 100 >= 80 * sz) i = qs;
        else if (len >= 2 && bq >= len - 2 && sz <= 5 && matches >= 3) i = qs;
    }
    int cur = len;
    u32 k = 1, why = FXG_R_KEPT, cl = 0, ao = 0;
    if (i > 0) { i += a.clip_keep_delta; if (i < cur) cur = i; cl = 1; }
    if (i == 0) {
        ao = 1;
        if (!(a.clip_flags & FXG_CLIP_ADAPTER_ONLY)) { k = 0; why = FXG_R_CLIP_ADAPTER_ONLY; }
    } else if ((u32)cur < a.clip_min_len) { k = 0; why = FXG_R_CLIP_TOO_SHORT; }
    else if (i == -1 && (a.clip_flags & FXG_CLIP_DISCARD_NON_CLIPPED)) { k = 0; why = FXG_R_CLIP_NO_ADAPTER; }
    else if (i > 0 && (a.clip_flags & FXG_CLIP_DISCARD_CLIPPED)) { k = 0; why = FXG_R_CLIP_ADAPTER_FOUND; }
    else if (!(a.clip_flags & FXG_CLIP_KEEP_N) && first_n < cur) { k = 0; why = FXG_R_CLIP_N; }
    else if (a.clip_flags & FXG_CLIP_ADAPTER_ONLY) { k = 0; why = FXG_R_CLIP_K_MODE; }
    *out_len = (u32)cur; *keep = k; *reason = why; *clipped = cl; *adapter_only = ao;
}

// General form: w0 = query_start<<16 | target_start<<8 | mismatches, w1 = path_len<<16 | matches.
// Every select is written as a ternary on values (no control flow) so that the cell is ~30 straight VALU ops.
template <int AMAX>
FXG_HD void fxg_clip_read(const FxgKArgs &a, const uint8_t *rd, int len, int rows,
                          u32 *out_len, u32 *keep, u32 *reason, u32 *clipped, u32 *adapter_only, const bool UR = false)      // UR: `rows` is the same in every lane
{
    float S[AMAX];
    u32 W0[AMAX], W1[AMAX];
    const int A = a.alen;
#pragma unroll
    for (int t = 0; t < AMAX; ++t) {
        S[t] = (t <= 3) ? 0.0f : -5.0f * (float)(t - 3);   // target_border (:355-361)
        W0[t] = FXG_INVALID_TUPLE; W1[t] = 0u;
    }
    float best = -1000000.0f;
    u32 bw0 = FXG_INVALID_TUPLE, bw1 = 0u, bq = 0u;
    int first_n = len;
    const int rows_u = UR ? rows : fxg_wave_max(rows);
#pragma unroll 1
    for (int q = 0; q < rows_u; ++q) {                     // rows == len unless the stale tail of earlier reads is emulated
        if (q >= rows) continue;                           // (a lane whose read is shorter idles; the loop itself is wave-uniform)
        const u32 c = rd[q];
        const bool qn = (c == (u32)'N');
        first_n = (qn && first_n == len && q < len) ? q : first_n;
        float dS = 0.0f, uS = 0.0f;                        // S[q-1][-1] and S[q][-1]: query_border = 0 (N1 for q == 0)
        u32 dW0 = FXG_INVALID_TUPLE, dW1 = 0u, uW0 = FXG_INVALID_TUPLE, uW1 = 0u;
#pragma unroll
        for (int t = 0; t < AMAX; ++t) {
            // No break and no guard: the trip count stays a compile-time constant (t indexes registers) and the body is
            // straight-line.  Cells with t >= A compute harmless garbage (they only feed cells further right) and are
            // excluded from the best-cell update below.
            const u32 tc = (u32)(uint8_t)a.adapter[t];
            const bool tn = (tc == (u32)'N');
            const bool eq = (c == tc);
            const bool neutral = qn || tn;
            const float pair = neutral ? ((qn && tn) ? 0.0f : 0.1f) : (eq ? 1.0f : -1.0f);   // sequence_alignment.h:157-169
            const float ul = dS + pair;
            const float up = uS + -5.0f;
            float left = S[t] + -5.0f;
            if (t > 3) left = (t - 3 > q) ? -100000.0f : left;                  // :387-389
            const bool g1 = up > ul;                                            // ul always beats the -1e8 seed; diag > up > left on ties
            float sc = g1 ? up : ul;
            u32 w0 = g1 ? uW0 : dW0, w1 = g1 ? uW1 : dW1;
            const bool g2 = left > sc;
            sc = g2 ? left : sc;
            w0 = g2 ? W0[t] : w0; w1 = g2 ? W1[t] : w1;
            const bool diag = !(g1 || g2);
            const bool fresh = (w0 == FXG_INVALID_TUPLE);                       // the path enters the matrix in this cell
            w0 = fresh ? (((u32)q << 16) | ((u32)t << 8)) : w0;
            w1 = fresh ? 0u : w1;
            w1 += 0x10000u + ((diag && !neutral && eq) ? 1u : 0u);
            w0 += (diag && !neutral && !eq) ? 1u : 0u;
            dS = S[t]; dW0 = W0[t]; dW1 = W1[t];
            S[t] = sc; W0[t] = w0; W1[t] = w1;
            uS = sc; uW0 = w0; uW1 = w1;
            const bool gb = (sc > best) && (t < A);                             // first maximum in query-major order (:421-425)
            best = gb ? sc : best; bw0 = gb ? w0 : bw0; bw1 = gb ? w1 : bw1; bq = gb ? (u32)q : bq;
        }
    }
    fxg_clip_finish(a, len, (int)(bw0 >> 16), (int)((bw0 >> 8) & 0xFFu), (int)(bw0 & 0xFFu), (int)(bw1 >> 16), (int)(bw1 & 0xFFFFu),
                    (int)bq, first_n, out_len, keep, reason, clipped, adapter_only);
}

// Packed form for adapters up to 16 columns (every BASELINE config; the field widths would take 31): the whole path summary is ONE u32
//   w = query_start:8 | target_start:5 | diagonal:5 | path_len:9 | matches:5      (path_len <= L + A <= 286)
// `diagonal` counts the non-neutral diagonal steps (matches + mismatches): it grows by a constant of the row (0 when the read
// base is 'N'), and `matches` sits in the lowest bits so that "+1 where read base == adapter base" is the carry-in of that
// same addition (v_addc_co_u32): a diagonal step is ONE instruction.  mismatches = diagonal - matches at the end.
#define FXG_PK_MAT1  1u
#define FXG_PK_SZ1   (1u << 5)
#define FXG_PK_DIA1  (1u << 14)
// FIRST: row q == 0.  The "path enters the matrix here" test can only fire where a predecessor lies outside the matrix:
// anywhere in row 0, and in column 0 of the other rows -- so rows q >= 1 test it at t == 0 only.
// TN: the adapter may contain 'N'.  Without it the pair score of a cell is one select on "read base == adapter base" between
// two values fixed per row (an 'N' in the read makes both 0.1 / neutral, and can never equal an adapter base).
// W holds every cell's summary ALREADY extended by one gap step (w + SZ1): that is what both the cell below (up) and the cell to
// the right in the next row (left) need, so the step is added once per cell instead of once per use; the diagonal adds the
// difference.  "No predecessor" is the value 0, which no extended summary can be (its path_len is >= 2).
// Sm holds every cell's score minus the gap penalty for the same reason (one subtraction serves `up` and `left`).
// Cell rule (sequence_alignment.cpp:380-417): strict '>' from diag to up to left, i.e. the maximum with ties going to diag, then up:
//   score = max3(ul, up, left); diag iff score == ul; else up iff score == up; else left.
// TRACK = false: the row cannot hold the first maximum (the second pass of fxg_clip_two_pass knows its row) -- no best-cell update.
// vstart: what a path that enters the matrix in this row records as its query_start (8 bits): the row itself in the one-pass form, the
// row relative to the second pass' first row in fxg_clip_two_pass -- which is what lets that form take reads of any length.
template <int AMAX, bool EARLY, bool FIRST, bool TN, bool TRACK = true>
FXG_HD void fxg_clip_row_packed(const FxgKArgs &a, int A, u32 c, int q, u32 vstart, float (&S)[AMAX], float (&Sm)[AMAX], u32 (&W)[AMAX], float &best, u32 &bw, u32 &bq)
{
    const bool qn = (c == (u32)'N');
    const float pair_eq = qn ? 0.1f : 1.0f, pair_ne = qn ? 0.1f : -1.0f;                 // sequence_alignment.h:157-169 for a target base that is not N
    const u32 dxr = qn ? 0u : FXG_PK_DIA1;                                               // what a diagonal step adds besides a match
    const float best_in = best;
    // pass 1: everything a cell takes from the row above -- the diagonal candidates ul = S[q-1][t-1] + pair and their summaries.
    // Done for the whole row first so that the old S / W values are dead before pass 2 overwrites them in place (no register
    // rotation in the rolled row loop) and so that only the up/left chain is left on the dependent path.
    float ul[AMAX];
    u32 wd[AMAX], dWv[AMAX];
#pragma unroll
    for (int t = 0; t < AMAX; ++t) {
        const u32 tc = (u32)(uint8_t)a.adapter[t];    // straight-line body, see fxg_clip_read
        const bool eq = (c == tc);
        float pair = eq ? pair_eq : pair_ne;
        const float dS = t ? S[t - 1] : 0.0f;              // S[q-1][-1]: query_border = 0 (N1 for q == 0)
        const u32 dW = t ? W[t - 1] : 0u;                  // no predecessor left of column 0
        if (TN) {
            const bool tn = (tc == (u32)'N');
            pair = tn ? (qn ? 0.0f : 0.1f) : pair;
            wd[t] = dW + (tn ? 0u : dxr + (eq ? FXG_PK_MAT1 : 0u));
        } else {
            wd[t] = (dW + dxr) + (u32)eq;
        }
        ul[t] = dS + pair;
        dWv[t] = dW;
    }
    // pass 2: the chain along the row
    float uSm = -5.0f;                                     // S[q][-1] - 5
    u32 uW = 0u;
#pragma unroll
    for (int t = 0; t < AMAX; ++t) {
        const float up = uSm;
        float left = Sm[t];
        if (EARLY && t > 3) left = (t - 3 > q) ? -100000.0f : left;                          // :387-389, only rows q < A-4
        const float sc = fmaxf(fmaxf(ul[t], up), left);
        const bool isd = (sc == ul[t]), isu = (sc == up);
        u32 w;
        if (FIRST || t == 0) {                                                               // a predecessor may lie outside the matrix
            const u32 src = isd ? dWv[t] : (isu ? uW : W[t]);
            const u32 step = isd ? wd[t] - dWv[t] : 0u;
            w = (src == 0u) ? (((vstart << 24) | ((u32)t << 19)) + FXG_PK_SZ1 + step) : (src + step);   // the path enters the matrix here
        } else {
            w = isu ? uW : W[t];
            w = isd ? wd[t] : w;
        }
        const u32 wp = w + FXG_PK_SZ1;
        const float scm = sc + -5.0f;
        S[t] = sc; Sm[t] = scm; W[t] = wp;
        uSm = scm; uW = wp;
        // first maximum in query-major order.  Columns past the adapter (t >= A) do not count; the smallest adapter of this
        // bucket has AMIN bases, so only columns t >= AMIN need the test (none for the exact buckets 9..16) -- a mask that went
        // through a scalar AND costs a v_cndmask several times what a mask straight from a v_cmp does (scripts/ubench/valu_rate.hip).
        constexpr int AMIN = AMAX <= 4 ? 1 : (AMAX <= 8 ? 5 : (AMAX <= 16 ? AMAX : AMAX - 3));
        if (!TRACK) continue;
        if (t < AMIN) {
            const bool gb = sc > best;
            bw = gb ? w : bw;
            best = fmaxf(best, sc);
        } else {
            const bool gb = (sc > best) && (t < A);
            best = gb ? sc : best; bw = gb ? w : bw;
        }
    }
    if (TRACK) bq = (best > best_in) ? (u32)q : bq;        // the best cell moved into this row (best only ever grows)
}

template <int AMAX, bool TN>
FXG_HD void fxg_clip_rows_packed(const FxgKArgs &a, const uint8_t *rd, int len, int rows, float &best, u32 &bw, u32 &bq, int &first_n, const bool UR = false)
{
    float S[AMAX], Sm[AMAX];
    u32 W[AMAX];
    const int A = a.alen;
#pragma unroll
    for (int t = 0; t < AMAX; ++t) { S[t] = (t <= 3) ? 0.0f : -5.0f * (float)(t - 3); Sm[t] = S[t] + -5.0f; W[t] = 0u; }   // 0 = no predecessor (see fxg_clip_row_packed)
    const int early_rows = (A - 4 < rows) ? (A - 4 > 0 ? A - 4 : 0) : rows;  // rows where "t - 3 > q" can still hold for some t < A
    int q = 0;
    if (rows > 0) {                                                           // row 0: every cell may start a path
        const u32 c = rd[0];
        first_n = (c == (u32)'N' && 0 < len) ? 0 : first_n;
        if (early_rows > 0) fxg_clip_row_packed<AMAX, true, true, TN>(a, A, c, 0, 0u, S, Sm, W, best, bw, bq);
        else fxg_clip_row_packed<AMAX, false, true, TN>(a, A, c, 0, 0u, S, Sm, W, best, bw, bq);
        q = 1;
    }
    // The row's base comes out of LDS one row AHEAD of its use (rows >= 1 always exists in the staged tile: the bases of the next
    // read or the tile's slack follow), so the load's latency is never waited for at the top of a row.
    // wave-uniform loops (fxg_wave_max above): the early form tests the row number itself, so it serves every lane while ANY lane is early
    const int early_u = UR ? early_rows : fxg_wave_max(early_rows), rows_u = UR ? rows : fxg_wave_max(rows);
#pragma unroll 1
    for (; q < early_u; ++q) {
        if (q >= rows) continue;
        const u32 c = rd[q];
        first_n = (c == (u32)'N' && first_n == len && q < len) ? q : first_n;
        fxg_clip_row_packed<AMAX, true, false, TN>(a, A, c, q, (u32)q, S, Sm, W, best, bw, bq);
    }
#pragma unroll 1
    for (; q < rows_u; ++q) {
        if (q >= rows) continue;
        const u32 c = rd[q];
        first_n = (c == (u32)'N' && first_n == len && q < len) ? q : first_n;
        fxg_clip_row_packed<AMAX, false, false, TN>(a, A, c, q, (u32)q, S, Sm, W, best, bw, bq);
    }
}

// ------------------------------------------------------------------------------------------------
// Two passes for adapters up to 16 bases (every BASELINE config).  Only ONE cell's path summary is ever used -- the first maximum's --
// and that path is short: it enters the matrix with a diagonal step from a border value <= 0 (in row 0 the gap candidates are -5
// or -100000 against a diagonal >= -1; in column 0 the same), every diagonal step adds at most 1, every gap step -5, and the best
// score is at least -1 (cell (q, 0) >= its own pair score), so it has at most (A + 1) / 5 gap steps and covers at most
//   SPAN = A + (A + 1) / 5   rows   (15 for the 13-base adapter).
// Pass 1 therefore carries SCORES only (5 VALU instructions per cell instead of 15): it finds the row of the first maximum
// (first row whose maximum exceeds everything before it) and keeps checkpoints of the score row every C = SPAN / 2 rows -- three
// live ones, P[j % 3] = the row before chunk j.  When the best moves in chunk j, the checkpoint two chunks back (2 C >= SPAN - 1 rows
// before the chunk's first row) is remembered.  Pass 2 restarts the summary-carrying DP (fxg_clip_row_packed) from that checkpoint
// and runs at most 3 C rows up to the best row; the scores it recomputes are the same fp32 operations in the same order, the
// summaries of cells whose paths started before the checkpoint are garbage, and the best cell's path is not one of them.
// Row 0 needs no variant of its own here: the virtual cells above it carry the summary a path entering diagonally at (0, t) starts
// from ((t << 19) + one step, what FIRST computes on the spot), and gap moves out of the border never win (above).
// ------------------------------------------------------------------------------------------------
template <int AMAX> struct FxgClip2 {
    static constexpr int SPAN = AMAX + (AMAX + 1) / 5;
    static constexpr int C = SPAN / 2 > 0 ? SPAN / 2 : 1;      // ceil((SPAN - 1) / 2)
    static constexpr int WIN = 3 * C;
};

// one row of pass 1: scores only; returns the row's maximum over the adapter's columns.  One sweep (the cell above-left is saved
// as the sweep passes it): with four waves per SIMD a wave issues every 6-10 cycles, which covers the dependent max3 -> add chain,
// so nothing is gained by computing the diagonal candidates of the whole row first -- and that form copied the row every time.
template <int AMAX, bool EARLY>
FXG_HD float fxg_clip_row_score(const FxgKArgs &a, int A, u32 c, int q, float (&S)[AMAX], float (&Sm)[AMAX])
{
    const bool qn = (c == (u32)'N');
    const float pair_eq = qn ? 0.1f : 1.0f, pair_ne = qn ? 0.1f : -1.0f;                 // sequence_alignment.h:157-169 (adapter without N)
    float uSm = -5.0f, rowmax = -1000000.0f;                                             // S[q][-1] - 5
    constexpr int AMIN = AMAX <= 4 ? 1 : (AMAX <= 8 ? 5 : AMAX);      // smallest adapter of the bucket: columns below it always count
    // The match masks of the whole row first: a lane mask needs two wait states between the v_cmp that writes it and the v_cndmask
    // that reads it, and with the compare right in front of its select the compiler paid them in s_nops (11 per row).
    bool eq[AMAX];
#pragma unroll
    for (int t = 0; t < AMAX; ++t) eq[t] = (c == (u32)(uint8_t)a.adapter[t]);
    // the diagonal candidate of column t + 1 is taken from S[t] BEFORE the sweep overwrites S[t]: every array is updated in place
    float ul = 0.0f + (eq[0] ? pair_eq : pair_ne);                                       // S[q-1][-1] = query_border = 0
#pragma unroll
    for (int t = 0; t < AMAX; ++t) {
        float ul_next = 0.0f;
        if (t + 1 < AMAX) ul_next = S[t] + (eq[t + 1] ? pair_eq : pair_ne);
        float left = Sm[t];
        if (EARLY && t > 3) left = (t - 3 > q) ? -100000.0f : left;                      // sequence_alignment.cpp:387-389
        const float sc = fmaxf(fmaxf(ul, uSm), left);
        const float scm = sc + -5.0f;
        S[t] = sc; Sm[t] = scm; uSm = scm; ul = ul_next;
        if (t < AMIN) rowmax = fmaxf(rowmax, sc);
        else rowmax = (t < A) ? fmaxf(rowmax, sc) : rowmax;
    }
    return rowmax;
}

// ------------------------------------------------------------------------------------------------
// Pair values out of an LDS table (round 6).  A cell's pair score depends on (read base, adapter column) only, and a wave's lanes hold
// at most a handful of different read bases: instead of a compare and a select per CELL (2 of the 5 VALU instructions of pass 1) a lane
// fetches the whole row of pair values for its base -- 16 floats, four ds_read_b128 on the LDS pipe, which the DP leaves idle, one row
// ahead of its use -- and the diagonal candidate is the same v_add_f32 as before with the fetched value as its operand: bit-identical
// scores (+-1.0f, or 0.1f in every column where the read base is 'N': sequence_alignment.h:157-169, adapter without N).
// Pass 2 fetches a second row the same way: what a diagonal step adds to the path summary (FXG_PK_DIA1 + match, 0 for 'N'), so that
// its compare + add-with-carry per cell become one v_add_u32.
// Table (fxg_clip_ptab_build, one per workgroup): lut u16[256] = byte -> offset of its pair row; R pair rows of 16 floats, then R step
// rows of 16 u32, 64 bytes apart: row 0 = a byte the adapter does not contain, row 1 = 'N', rows 2.. = the adapter's distinct bytes in
// order of first appearance (R = 2 + distinct bytes, fxg_ptab_rows: 6 for an adapter over ACGT) -- A, C, G, T sit in rows of different
// 64-byte bank groups, so a wave's fetch has no bank conflict among them.  Any byte value is served (lower case, IUPAC codes: whatever
// the caller put into the batch compares as the reference's `==` does).
// Measured (profiles/r06/): cfg3 5.28 -> 4.49 ms, cfg5 7.66 -> 6.66 ms per 20 M reads with pass 1 alone on the table.  Rows of 16 HALVES and
// one v_fma_mix_f32 per cell (fma(half, 1.0f or 0.1f, S): exact too, half the fetch) came out slower, 4.80 / 7.19 ms: v_fma_mix_f32 issues
// at the rate of v_max3_f32 (2.6 cycles at four waves per SIMD, against 1.64 for v_add_f32; scripts/ubench/valu_rate.hip 77-80) and two
// slow instructions in a row do not overlap.
// ------------------------------------------------------------------------------------------------
#define FXG_PTAB_LUT_BYTES 512u
#define FXG_PTAB_ROW_BYTES 64u
#ifdef FXG_NO_PTAB      // A/B builds (scripts/clip_ab.py): the compare + select cell of rounds 3-5
__host__ __device__ constexpr bool fxg_clip_uses_ptab(int) { return false; }
#else
__host__ __device__ constexpr bool fxg_clip_uses_ptab(int amax) { return amax < 0; }                     // every packed instance: the register two-pass forms and the 17..99-column forms
#endif
// rows of the table for this adapter: "other", 'N', and one per distinct byte of the adapter
FXG_HD u32 fxg_ptab_rows(const char *adapter, int alen)
{
    u32 r = 2u;
    for (int t = 0; t < alen; ++t) {
        bool seen = adapter[t] == 'N';                     // ('N' has row 1 whatever the adapter holds)
        for (int u = 0; u < t; ++u) seen = seen || (adapter[u] == adapter[t]);
        r += seen ? 0u : 1u;
    }
    return r;
}
// Row stride for `cols` columns.  Up to 16 columns a row is 64 bytes and the rows of A, C, G, T land in different bank groups as they are; wider rows
// sweep all 64 banks, so the stride is made an ODD multiple of 16 bytes: the rows of any 16 different bytes then start in 16 different 16-byte bank groups
// and a wave whose lanes fetch the same block of different rows has no bank conflict.
FXG_HD u32 fxg_ptab_stride(u32 cols) { const u32 s = (cols * 4u + 15u) & ~15u; return cols <= 16u ? FXG_PTAB_ROW_BYTES : (((s >> 4) & 1u) ? s : s + 16u); }
FXG_HD u32 fxg_ptab_bytes(u32 rows, u32 stride) { return FXG_PTAB_LUT_BYTES + 2u * rows * stride; }
#define FXG_PTAB_MAX_ROWS_K 8u      // 17..99 columns: adapters of more than six distinct bytes (IUPAC-rich) take the general form instead (fxg_plan.h)

// which row a byte value has (host side, fxg_make_plan): 0 = a byte the adapter does not contain, 1 = 'N', 2.. = the adapter's distinct bytes in order of first
// appearance; bit 7 marks the ONE byte value whose thread writes the row (byte 0 for row 0: the adapter is a C string, 0 is never one of its bytes)
FXG_HD void fxg_ptab_row_map(const char *adapter, int alen, uint8_t (&row)[256])
{
    for (int b = 0; b < 256; ++b) row[b] = 0u;
    row[0] = 0x80u;
    u32 next = 2u;
    for (int t = 0; t < alen; ++t) {
        const u32 c = (u32)(uint8_t)adapter[t];
        if (c == (u32)'N' || row[c] != 0u) continue;
        row[c] = (uint8_t)(0x80u | next++);
    }
    row[(u32)'N'] = 0x80u | 1u;
}

FXG_HD void fxg_clip_ptab_build(const FxgKArgs &a, uint8_t *ptab, u32 tid, u32 nthreads)
{
    uint16_t *lut = reinterpret_cast<uint16_t *>(ptab);
    const int COLS = (int)a.clip_ptab_cols, A = a.alen < COLS ? a.alen : COLS;
    const u32 R = a.clip_ptab_rows, ST = a.clip_ptab_stride;
    for (u32 b = tid; b < 256u; b += nthreads) {
        const u32 e = a.clip_ptab_row[b], row = e & 0x7Fu;
        lut[b] = (uint16_t)(FXG_PTAB_LUT_BYTES + row * ST);
        if (e & 0x80u) {                                   // one writer per row
            float *pv = reinterpret_cast<float *>(ptab + FXG_PTAB_LUT_BYTES + row * ST);
            u32 *sv = reinterpret_cast<u32 *>(ptab + FXG_PTAB_LUT_BYTES + (R + row) * ST);
            for (int t = 0; t < COLS; ++t) {
                const u32 tc = t < A ? (u32)(uint8_t)a.adapter[t] : 0u;
                const bool eq = tc == b, qn = b == (u32)'N', tn = tc == (u32)'N';              // sequence_alignment.h:157-169: either base N -> neutral
                pv[t] = tn ? (qn ? 0.0f : 0.1f) : (qn ? 0.1f : (eq ? 1.0f : -1.0f));
                sv[t] = (tn || qn) ? 0u : a.clip_ptab_dia1 + (eq ? 1u : 0u);                   // DIA1 + MAT1 where the bases are equal; a neutral step counts neither
            }
        }
    }
}

// A scheduling fence on either side of the fetch of the next row's values: left alone the scheduler sinks the fetch to the end of the row and the next
// row begins by waiting for it (FXG_NO_PTAB_SCHED: A/B; pass 1 alone 2.33 -> 2.25 ms per 20 M reads of 100 bases, 1.89 without any fetch: profiles/r06/)
#if !defined(FXG_NO_PTAB_SCHED) && !defined(FXG_HOST_EMULATION)
#define FXG_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define FXG_SCHED_FENCE() ((void)0)
#endif

// the gap penalty as a register operand: as a 32-bit literal it makes every `S - 5` an 8-byte instruction (v_add_f32 with a literal issues at 1.83
// instead of 1.64 cycles at four waves per SIMD, scripts/ubench/valu_rate.hip 58)
FXG_HD float fxg_minus5()
{
    float v = -5.0f;
#if defined(FXG_M5_SGPR) && !defined(FXG_HOST_EMULATION)
    asm volatile("" : "+s"(v));
#elif defined(FXG_M5_VGPR) && !defined(FXG_HOST_EMULATION)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// the table rows at offset `off` into registers: pair values, and (STEPS) what a diagonal step adds to the path summary
template <int AMAX, bool STEPS>
FXG_HD void fxg_ptab_fetch(const uint8_t *ptab, u32 off, u32 step_off, u32 (&pr)[16], u32 (&st)[16])
{
    const u32x4 *p = reinterpret_cast<const u32x4 *>(ptab + off);
#pragma unroll
    for (int k = 0; k < (AMAX + 3) / 4; ++k) { const u32x4 v = p[k]; pr[4 * k] = v.x; pr[4 * k + 1] = v.y; pr[4 * k + 2] = v.z; pr[4 * k + 3] = v.w; }
    if constexpr (STEPS) {
        const u32x4 *g = reinterpret_cast<const u32x4 *>(ptab + off + step_off);
#pragma unroll
        for (int k = 0; k < (AMAX + 3) / 4; ++k) { const u32x4 v = g[k]; st[4 * k] = v.x; st[4 * k + 1] = v.y; st[4 * k + 2] = v.z; st[4 * k + 3] = v.w; }
    }
}

// One row of scores with the pair values in pr (this row's table row).  pr (and st, STEPS) are replaced by the rows at offset `o_next` as soon as
// the diagonal candidates are taken -- the up/left chain of the row covers the fetch.
template <int AMAX, bool EARLY, bool STEPS = false>
FXG_HD float fxg_clip_row_score_t(int A, int q, float (&S)[AMAX], float (&Sm)[AMAX], u32 (&pr)[16], u32 (&st)[16], const uint8_t *ptab, u32 o_next, u32 step_off)
{
    constexpr int AMIN = AMAX <= 4 ? 1 : (AMAX <= 8 ? 5 : AMAX);
    // everything a cell takes from the row above first: ul[t] = S[q-1][t-1] + pair(t); S[q-1][-1] = query_border = 0
    float ul[AMAX];
#pragma unroll
    for (int t = 0; t < AMAX; ++t) ul[t] = (t ? S[t - 1] : 0.0f) + __builtin_bit_cast(float, pr[t]);
    FXG_SCHED_FENCE();
#ifndef FXG_ABL_NOFETCH      // (timing experiment: the row without its fetch -- wrong results)
    fxg_ptab_fetch<AMAX, STEPS>(ptab, o_next, step_off, pr, st);      // the next row's values, into the registers this row no longer needs
#endif
    FXG_SCHED_FENCE();
    const float m5 = fxg_minus5();
    float uSm = -5.0f, rowmax = -1000000.0f;                                             // S[q][-1] - 5
#pragma unroll
    for (int t = 0; t < AMAX; ++t) {
        float left = Sm[t];
        if (EARLY && t > 3) left = (t - 3 > q) ? -100000.0f : left;                      // sequence_alignment.cpp:387-389
        const float sc = fmaxf(fmaxf(ul[t], uSm), left);
        const float scm = sc + m5;
        S[t] = sc; Sm[t] = scm; uSm = scm;
        if (t < AMIN) rowmax = fmaxf(rowmax, sc);
        else rowmax = (t < A) ? fmaxf(rowmax, sc) : rowmax;
    }
    return rowmax;
}

// One row of pass 2 (fxg_clip_row_packed<.., FIRST = false, TN = false>: the same cell, the same summary word) with the pair values and the
// diagonal's summary steps out of the table: ul = S[q-1][t-1] + pr[t], wd = W[q-1][t-1] + st[t].
template <int AMAX, bool EARLY, bool TRACK>
FXG_HD void fxg_clip_row_packed_t(int A, int q, u32 vstart, float (&S)[AMAX], float (&Sm)[AMAX], u32 (&W)[AMAX], float &best, u32 &bw, u32 &bq,
                                  u32 (&pr)[16], u32 (&st)[16], const uint8_t *ptab, u32 o_next, u32 step_off)
{
    const float best_in = best;
    float ul[AMAX];
    u32 wd[AMAX];
#pragma unroll
    for (int t = 0; t < AMAX; ++t) {
        ul[t] = (t ? S[t - 1] : 0.0f) + __builtin_bit_cast(float, pr[t]);                    // S[q-1][-1]: query_border = 0
        wd[t] = (t ? W[t - 1] : 0u) + st[t];                                                 // no predecessor left of column 0: the step alone
    }
    FXG_SCHED_FENCE();
    fxg_ptab_fetch<AMAX, true>(ptab, o_next, step_off, pr, st);
    FXG_SCHED_FENCE();
    const float m5 = fxg_minus5();
    float uSm = -5.0f;                                     // S[q][-1] - 5
    u32 uW = 0u;
#pragma unroll
    for (int t = 0; t < AMAX; ++t) {
        const float up = uSm;
        float left = Sm[t];
        if (EARLY && t > 3) left = (t - 3 > q) ? -100000.0f : left;                          // :387-389, only rows q < A-4
        const float sc = fmaxf(fmaxf(ul[t], up), left);
        const bool isd = (sc == ul[t]), isu = (sc == up);
        u32 w;
        if (t == 0) {                                                                        // diag and up come from outside the matrix: the path enters it here
            const u32 src = (isd || isu) ? 0u : W[0];
            const u32 step = isd ? wd[0] : 0u;
            w = (src == 0u) ? ((vstart << 24) + FXG_PK_SZ1 + step) : (src + step);
        } else {
            w = isu ? uW : W[t];
            w = isd ? wd[t] : w;
        }
        const u32 wp = w + FXG_PK_SZ1;
        const float scm = sc + m5;
        S[t] = sc; Sm[t] = scm; W[t] = wp;
        uSm = scm; uW = wp;
        constexpr int AMIN = AMAX <= 4 ? 1 : (AMAX <= 8 ? 5 : (AMAX <= 16 ? AMAX : AMAX - 3));
        if (!TRACK) continue;
        if (t < AMIN) {
            const bool gb = sc > best;
            bw = gb ? w : bw;
            best = fmaxf(best, sc);
        } else {
            const bool gb = (sc > best) && (t < A);
            best = gb ? sc : best; bw = gb ? w : bw;
        }
    }
    if (TRACK) bq = (best > best_in) ? (u32)q : bq;
}

// GL window: index of the array's last readable dword counted from a row that starts `off` bytes into it.  The count is 64-bit -- a batch
// of 8 GiB and more (30 M reads at a 300-byte stride) has more than 2^31 dwords behind its early rows -- and the window never looks
// further than a row's own length ahead, so it is clamped to what an int index can hold instead of being truncated.
FXG_HD int fxg_gl_last_dword(u64 total, u64 off)
{
    const u64 left = total > off ? (total - off) >> 2 : 0;
    return left == 0 ? 0 : (left - 1 > (u64)0x7FFFFFFF ? 0x7FFFFFFF : (int)(left - 1));
}

// dword i of a row that starts on a dword boundary, clamped to the array's last dword (the window of fxg_clip_two_pass<.., GL> runs two dwords ahead)
FXG_HD u32 fxg_ld32(const uint8_t *row, int i, int imax)
{
    const int k = i < imax ? i : imax;
#ifdef FXG_HOST_EMULATION
    u32 w; memcpy(&w, row + 4 * (size_t)k, 4); return w;
#else
    return reinterpret_cast<const u32 *>(row)[k];
#endif
}

// First 'N' of a read (the -n rule, fastx_clipper.cpp:306-311), for the two-pass forms, whose row loops do not look at it.  Rows that start on
// a dword boundary (stride a multiple of 4: the staged tile is 16-byte aligned) are scanned four bases at a time: a byte of v ^ "NNNN" is
// zero where the base is N, (x - 0x01010101) & ~x & 0x80808080 flags zero bytes -- a flag can be false only ABOVE a true one (the borrow), so the
// lowest flag of a dword is always a true N.  Top down, so the last hit kept is the first N; the trip count is the wave's (fxg_wave_max).
// (One byte per trip cost the default command line -- fastx_clipper without -n -- 9.5 % of the kernel: profiles/r04/ac_clip_n_rule.txt.)
FXG_HD int fxg_clip_first_n(const uint8_t *rd, int len, int len_u, u32 stride, int first_n)
{
    if ((stride & 3u) == 0u) {
#pragma unroll 1
        for (int k = (len_u + 3) / 4 - 1; k >= 0; --k) {
#ifdef FXG_HOST_EMULATION
            u32 w; memcpy(&w, rd + 4 * k, 4);                      // (the emulator's rows need not be aligned)
#else
            const u32 w = reinterpret_cast<const u32 *>(rd)[k];
#endif
            const u32 v = w ^ 0x4E4E4E4Eu;
            const u32 z = (v - 0x01010101u) & ~v & 0x80808080u;
            const int pos = 4 * k + (int)((u32)__builtin_ctz(z | 0x80000000u) >> 3);      // the lowest flag (z | top bit: ctz is defined, and a hit there is checked like any other)
            first_n = (z != 0u && pos < len) ? pos : first_n;
        }
    } else {
#pragma unroll 1
        for (int k = len_u - 1; k >= 0; --k) first_n = (k < len && rd[k] == (uint8_t)'N') ? k : first_n;
    }
    return first_n;
}

// returns the row the query_start field of bw counts from
// GL: `rd` points into the batch in global memory instead of a staged copy in LDS (fxg_plan.h: clip_global).  Pass 1, whose row number is a
// scalar, then takes the row's base out of a two-dword window that moves on every fourth row (one 4-byte load per lane and four rows: byte loads,
// each lane on a line of its own, would cost the CU's vector cache a lookup per lane and row); pass 2 and the N scan touch ~20 rows and read them directly.
// ptab (staged form): the workgroup's pair table (fxg_clip_ptab_build) -- pass 1 then takes its pair values from it (fxg_clip_row_score_t)
template <int AMAX, bool GL = false>
FXG_HD int fxg_clip_two_pass(const FxgKArgs &a, const uint8_t *rd, int len, int rows, float &best, u32 &bw, u32 &bq, int &first_n, const bool UR = false, const uint8_t *ptab = nullptr)
{
    constexpr int C = FxgClip2<AMAX>::C;
    float S[AMAX], Sm[AMAX], P0[AMAX], P1[AMAX], P2[AMAX], CB[AMAX];
    const int A = a.alen;
#pragma unroll
    for (int t = 0; t < AMAX; ++t) { S[t] = (t <= 3) ? 0.0f : -5.0f * (float)(t - 3); Sm[t] = S[t] + -5.0f; P0[t] = P1[t] = P2[t] = CB[t] = S[t]; }
    const int early_rows = (A - 4 < rows) ? (A - 4 > 0 ? A - 4 : 0) : rows;
    // every loop below runs a wave-uniform number of trips (fxg_wave_max): q and the chunk bounds are scalars, a lane past its own
    // last row skips the row, and a lane without rows at all (rows == 0) walks through with everything predicated off
    const int rows_u = UR ? rows : fxg_wave_max(rows > 0 ? rows : 0);
    if (rows_u <= 0) return 0;
    // ---- pass 1 ----
#if defined(FXG_ABL_ROWCLK) && !defined(FXG_HOST_EMULATION)      // timing experiment: shader cycles a wave spends in the row loops of either pass, and the rows it ran
    const u64 clk_a = __builtin_amdgcn_s_memtime();
#endif
    float b1 = -1000000.0f;
    int q = 0, r0 = 0, bq1 = 0;
    u32 cn = 0u, gw0 = 0u, gw1 = 0u;
    int gmax = 0;                                           // GL: last dword of the array that may be read, counted from this row's first
    // staged form: pr / mult = the pair values of row q, on = table offset of row q + 1, cn = the base of row q + 2 -- each fetched a row or more ahead of its use
    constexpr bool PT = fxg_clip_uses_ptab(-AMAX);
    u32 pr[16] = {}, st[16] = {}, on = 0u;
    const u32 step_off = PT ? a.clip_ptab_rows * a.clip_ptab_stride : 0u;      // from a byte's pair row to its step row
    if constexpr (GL) {
        gmax = fxg_gl_last_dword(a.clip_total, (u64)(rd - a.clip_src));
        gw0 = fxg_ld32(rd, 0, gmax); gw1 = fxg_ld32(rd, 1, gmax);
        if constexpr (PT) {
            const uint16_t *lut = reinterpret_cast<const uint16_t *>(ptab);
            on = lut[(gw0 >> 8) & 0xFFu];
            fxg_ptab_fetch<AMAX, false>(ptab, lut[gw0 & 0xFFu], step_off, pr, st);
        }
    } else if constexpr (PT) {
        const uint16_t *lut = reinterpret_cast<const uint16_t *>(ptab);
        const u32 o0 = lut[rd[0]];
        on = lut[rd[1]]; cn = rd[2];
        fxg_ptab_fetch<AMAX, false>(ptab, o0, step_off, pr, st);
    } else cn = rd[0];
#ifdef FXG_ABL_NOLUT         // (timing experiment: no base / lut reads in the row loop -- wrong results)
#define FXG_ABL_LUT_READS const u32 on2 = on;
#else
#define FXG_ABL_LUT_READS const u32 on2 = reinterpret_cast<const uint16_t *>(ptab)[cn]; cn = rd[q + 3];
#endif
    // chunk j = rows [j C, j C + C): saves the row before it in Psave = P[j % 3]; the restart row for a best found in it is
    // Pwin = P[(j + 1) % 3] = the row before chunk j - 2 (before chunk 0 for j < 2: the border, which all three start from)
#define FXG_CLIP_CHUNK(EARLY, Psave, Pwin)                                                                                   \
    {                                                                                                                        \
        _Pragma("unroll") for (int t = 0; t < AMAX; ++t) Psave[t] = S[t];                                                    \
        const int q0 = q, qend = q + C < rows_u ? q + C : rows_u;                                                            \
        bool upd = false;                                                                                                    \
        _Pragma("unroll 1") for (; q < qend; ++q) {                                                                          \
            float rm;                                                                                                        \
            if constexpr (GL && PT) {            /* the base of row q + 2 out of the window (q is a scalar), its table offset a row ahead of the fetch */ \
                const u32 ca = ((((q + 2) >> 2) == (q >> 2) ? gw0 : gw1) >> ((u32)((q + 2) & 3) << 3)) & 0xFFu;             \
                const u32 on2 = reinterpret_cast<const uint16_t *>(ptab)[ca];                                                \
                if ((q & 3) == 3) { gw0 = gw1; gw1 = fxg_ld32(rd, (q >> 2) + 2, gmax); }                                     \
                if (!UR && q >= rows) continue;                                                                              \
                rm = fxg_clip_row_score_t<AMAX, EARLY>(A, q, S, Sm, pr, st, ptab, on, step_off);                             \
                on = on2;                                                                                                    \
            } else if constexpr (GL) {                                                                                       \
                const u32 c = (gw0 >> ((u32)(q & 3) << 3)) & 0xFFu;                                                          \
                if ((q & 3) == 3) { gw0 = gw1; gw1 = fxg_ld32(rd, (q >> 2) + 2, gmax); }                                     \
                if (!UR && q >= rows) continue;                                                                              \
                rm = fxg_clip_row_score<AMAX, EARLY>(a, A, c, q, S, Sm);                                                     \
            } else if constexpr (!PT) {                                                                                      \
                const u32 c = cn;                                                                                            \
                cn = rd[q + 1];                                                                                              \
                if (!UR && q >= rows) continue;                                                                              \
                rm = fxg_clip_row_score<AMAX, EARLY>(a, A, c, q, S, Sm);                                                     \
            } else {                                                                                                         \
                if (!UR && q >= rows) continue;          /* (a lane past its rows never comes back: its fetch state may lapse) */ \
                FXG_ABL_LUT_READS                                                                                            \
                rm = fxg_clip_row_score_t<AMAX, EARLY>(A, q, S, Sm, pr, st, ptab, on, step_off);                             \
                on = on2;                                                                                                    \
            }                                                                                                                \
            const bool g = rm > b1;                                                                                          \
            b1 = g ? rm : b1; bq1 = g ? q : bq1; upd = upd || g;                                                             \
        }                                                                                                                    \
        if (upd) {                                                                                                           \
            _Pragma("unroll") for (int t = 0; t < AMAX; ++t) CB[t] = Pwin[t];                                                \
            r0 = q0 - 2 * C > 0 ? q0 - 2 * C : 0;                                                                            \
        }                                                                                                                    \
    }
    // the early rule (rows q < A - 4, sequence_alignment.cpp:387-389) can only fire in the first two chunks: 2 C >= SPAN - 1 >= A - 4
    static_assert(2 * C >= AMAX - 4, "early rows must end inside the first two chunks");
    FXG_CLIP_CHUNK(true, P0, P1)
    if (q < rows_u) FXG_CLIP_CHUNK(true, P1, P2)
    while (q < rows_u) {
        FXG_CLIP_CHUNK(false, P2, P0)
        if (q >= rows_u) break;
        FXG_CLIP_CHUNK(false, P0, P1)
        if (q >= rows_u) break;
        FXG_CLIP_CHUNK(false, P1, P2)
    }
#undef FXG_CLIP_CHUNK
#if defined(FXG_ABL_ROWCLK) && !defined(FXG_HOST_EMULATION)
    const u64 clk_b = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63u) == 0u) { atomicAdd(reinterpret_cast<u64 *>(a.errflag + 10) + 0, clk_b - clk_a); atomicAdd(reinterpret_cast<u64 *>(a.errflag + 10) + 1, (u64)rows_u); }
#endif
    if (rows <= 0) return 0;                                // (only now: the loops above are the wave's, not the lane's)
    if (FXG_DBG(a, 32u)) { best = b1; bq = (u32)bq1; bw = 0u; return 0; }      // ablation builds: pass 1 alone (wrong results, timing only)
    // ---- pass 2: rows r0 .. bq1 with the path summaries, from the checkpoint; only row bq1 can hold the first maximum ----
    u32 W[AMAX];
#pragma unroll
    for (int t = 0; t < AMAX; ++t) { S[t] = CB[t]; Sm[t] = CB[t] + -5.0f; W[t] = ((u32)(t + 1) << 19) + FXG_PK_SZ1; }
    q = r0;
    if constexpr (PT) {
        // The best path covers at most SPAN rows (above), i.e. starts in row rs = bq1 - SPAN + 1 or later: rows r0 .. rs - 1 re-run the SCORES only
        // (the cheap row of pass 1), rows rs .. bq1 carry the summaries, whose start field counts from rs.  Every lane is at rows of its own
        // here; the fetch pipeline is the one of pass 1 (values of row q in registers, table offset of row q + 1, base of row q + 2).
        constexpr int SPAN = FxgClip2<AMAX>::SPAN;
        const int rs = bq1 - SPAN + 1 > r0 ? bq1 - SPAN + 1 : r0;
        const uint16_t *lut = reinterpret_cast<const uint16_t *>(ptab);
        {
            const u32 o0 = lut[rd[q]];
            on = lut[rd[q + 1]]; cn = rd[q + 2];
            fxg_ptab_fetch<AMAX, true>(ptab, o0, step_off, pr, st);
        }
        const int n0 = rs - r0, n0u = fxg_wave_max(n0);
#pragma unroll 1
        for (int i = 0; i < n0u; ++i) {
            if (i >= n0) continue;
            const u32 on2 = lut[cn];
            cn = rd[q + 3];
            (void)fxg_clip_row_score_t<AMAX, true, true>(A, q, S, Sm, pr, st, ptab, on, step_off);      // (the early form tests the row number itself)
            on = on2; ++q;
        }
        // summary row i is read row rs + i >= i: rows past i = A - 4 are past the early rule, so the first A - 4 run in the early form (which tests the
        // row number itself) in EVERY lane and the two loops have the same trip counts across the wave -- a split by the lane's own rows would
        // make the wave issue both forms for as many rows as its slowest lane needs of each
        const int win = bq1 - rs, n1 = win < early_rows ? win : early_rows, n2 = win - n1;
        const int n1u = fxg_wave_max(n1), n2u = fxg_wave_max(n2);
#pragma unroll 1
        for (int i = 0; i < n1u; ++i) {
            if (i >= n1) continue;
            const u32 on2 = lut[cn];
            cn = rd[q + 3];
            fxg_clip_row_packed_t<AMAX, true, false>(A, q, (u32)(q - rs), S, Sm, W, best, bw, bq, pr, st, ptab, on, step_off);
            on = on2; ++q;
        }
#pragma unroll 1
        for (int i = 0; i < n2u; ++i) {
            if (i >= n2) continue;
            const u32 on2 = lut[cn];
            cn = rd[q + 3];
            fxg_clip_row_packed_t<AMAX, false, false>(A, q, (u32)(q - rs), S, Sm, W, best, bw, bq, pr, st, ptab, on, step_off);
            on = on2; ++q;
        }
        fxg_clip_row_packed_t<AMAX, true, true>(A, bq1, (u32)(bq1 - rs), S, Sm, W, best, bw, bq, pr, st, ptab, on, step_off);
#if defined(FXG_ABL_ROWCLK) && !defined(FXG_HOST_EMULATION)
        if ((threadIdx.x & 63u) == 0u) { atomicAdd(reinterpret_cast<u64 *>(a.errflag + 10) + 2, __builtin_amdgcn_s_memtime() - clk_b); atomicAdd(reinterpret_cast<u64 *>(a.errflag + 10) + 3, (u64)(n0u + n1u + n2u + 1)); atomicAdd(reinterpret_cast<u64 *>(a.errflag + 10) + 4, (u64)n0u); }
#endif
        if (!(a.clip_flags & FXG_CLIP_KEEP_N)) first_n = fxg_clip_first_n(rd, len, UR ? len : fxg_wave_max(len), a.clip_stride, first_n);
        return rs;
    }
    // window row i is read row r0 + i >= i: rows past i = A - 4 are past the early rule.  n1 rows in the early form, n2 in the other;
    // both loops run the wave's maximum, a lane beyond its own count skips the row
    const int win = bq1 - r0, n1 = win < early_rows ? win : early_rows, n2 = win - n1;
    const int n1u = fxg_wave_max(n1), n2u = fxg_wave_max(n2);
#pragma unroll 1
    for (int i = 0; i < n1u; ++i) {
        if (i >= n1) continue;
        fxg_clip_row_packed<AMAX, true, false, false, false>(a, A, (u32)rd[q], q, (u32)(q - r0), S, Sm, W, best, bw, bq);
        ++q;
    }
#pragma unroll 1
    for (int i = 0; i < n2u; ++i) {
        if (i >= n2) continue;
        fxg_clip_row_packed<AMAX, false, false, false, false>(a, A, (u32)rd[q], q, (u32)(q - r0), S, Sm, W, best, bw, bq);
        ++q;
    }
    fxg_clip_row_packed<AMAX, true, false, false, true>(a, A, (u32)rd[bq1], bq1, (u32)(bq1 - r0), S, Sm, W, best, bw, bq);
    // the -n rule needs the first N of the read itself (fastx_clipper.cpp:306-311); nothing else does
    if (!(a.clip_flags & FXG_CLIP_KEEP_N)) {
        first_n = fxg_clip_first_n(rd, len, UR ? len : fxg_wave_max(len), a.clip_stride, first_n);
    }
    return r0;
}

// ------------------------------------------------------------------------------------------------
// Adapters of 17..99 bases (MAX_ADAPTER_LEN, fastx_clipper.cpp:35): the summary still ONE u32 per cell; this is the one-pass driver (reads <= 255).
// A path can only enter the matrix in row 0 or in column 0, so its start is one number, not two:
//   k = start:9 | path_len:9 | diagonal:7 | matches:7        start = query_start (target_start 0), or 256 + target_start (query_start 0)
// (path_len <= L + A <= 355, diagonal and matches <= A <= 100).  The row is swept in place: the diagonal candidate of column t + 1 --
// score and summary -- is taken from column t before the sweep overwrites it, so a row needs no copy of itself and the live state
// is S (+ S - 5 where registers allow) and K: 2-3 registers per adapter column, which is what decides how many waves a SIMD holds
// (the earlier form kept three more arrays per row and fell to one wave per SIMD from 20 columns on: 2 200-2 700 GCUPS at 17..28
// bases, 1 040 at 32 and 200 from 33 on -- profiles/r03/u_clip_by_adapter_len_before.txt).
// The virtual cells above row 0 carry the summary a path entering diagonally at (0, t) starts from, as in fxg_clip_two_pass.
// ------------------------------------------------------------------------------------------------
#if defined(FXG_HOST_EMULATION) || defined(FXG_NO_KEEP_V)
#define FXG_KEEP_V(x) ((void)0)
#else
#define FXG_KEEP_V(x) asm volatile("" : "+v"(x))
#endif
#define FXG_K_MAT1 1u
#define FXG_K_DIA1 (1u << 7)
#define FXG_K_SZ1  (1u << 14)
#define FXG_K_START(v) ((u32)(v) << 23)
// smallest adapter of the bucket AMAX (fxg_plan.h): columns below it always count towards the best cell
__host__ __device__ constexpr int fxg_clip_k_amin(int amax, bool tn = false)
{
    return tn ? (amax <= 16 ? 1 : amax <= 24 ? 17 : amax <= 36 ? 25 : amax <= 48 ? 37 : amax <= 64 ? 49 : 65)        // buckets 16 24 36 48 64 100 (builds without the pair table: adapters with N)
              : (amax <= 16 ? 1 : amax <= 20 ? 17 : amax <= 64 ? amax - 3 : amax <= 88 ? amax - 7 : 89);             // buckets every 4 columns to 64, every 8 to 88, then 100 (fxg_plan.h)
}
template <int AMAX> struct FxgClipK { static constexpr bool SM = AMAX <= 24; static constexpr int NSM = SM ? AMAX : 1; };

// TN: the adapter may contain 'N' (sequence_alignment.h:157-169: a neutral pair scores 0.1, N against N 0.0, and counts neither as match
// nor as mismatch).  Which columns are N is the same for every lane, so it costs scalar selects of the masks and two more VALU
// instructions per cell (the pair value and the diagonal's step each take one more select); the instances without it are unchanged.
// The rows of the 17..99-column forms with the pair table (round 6): block b of a table row holds the pair values of columns 4 b .. 4 b + 3, the step row
// what a diagonal step into them adds to the summary.  The sweep is in place -- the diagonal candidate of column t + 1 is taken from column t before the sweep
// overwrites it -- so it needs value t + 1 while it is at column t: the current block and the next one are kept (one ds_read_b128 each for pairs and steps per
// four columns, requested a block ahead of their first use: 16 registers), and the cell is add, add, max3, two compares, two selects, add, add -- no compare
// against the adapter, no select of the pair, no add-with-carry.  An N in the adapter is a column of the table like any other.
template <int AMAX, bool EARLY, bool TRACK>
FXG_HD void fxg_clip_row_kt(const FxgKArgs &a, int A, u32 c, int q, u32 vstart, float (&S)[AMAX], float (&Sm)[FxgClipK<AMAX>::NSM], u32 (&W)[AMAX],
                            float &best, u32 &bw, u32 &bq, const uint8_t *ptab)
{
    constexpr bool SM = FxgClipK<AMAX>::SM;
    constexpr int AMIN = fxg_clip_k_amin(AMAX, false);
    const u32 po = reinterpret_cast<const uint16_t *>(ptab)[c & 0xFFu];
    const u32x4 *pp = reinterpret_cast<const u32x4 *>(ptab + po), *sp = reinterpret_cast<const u32x4 *>(ptab + po + a.clip_ptab_rows * a.clip_ptab_stride);
    const float best_in = best, m5 = fxg_minus5();
    float uSm = -5.0f;                                                                   // S[q][-1] - 5
    u32 uW = 0u;
    u32x4 cp = pp[0], cs = sp[0];
    float ul = 0.0f + __builtin_bit_cast(float, cp.x);                                   // S[q-1][-1] = query_border = 0
    u32 wd = cs.x;                                                                       // no predecessor left of column 0: the step alone
#pragma unroll
    for (int kb = 0; kb < AMAX; kb += 4) {
        u32x4 np = cp, ns = cs;
        if (kb + 4 < AMAX) { np = pp[kb / 4 + 1]; ns = sp[kb / 4 + 1]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = kb + j;
            if (t >= AMAX) break;
            float ul_next = 0.0f;
            u32 wd_next = 0u;
            if (t + 1 < AMAX) {
                const u32 pv = j == 0 ? cp.y : j == 1 ? cp.z : j == 2 ? cp.w : np.x;
                const u32 sv = j == 0 ? cs.y : j == 1 ? cs.z : j == 2 ? cs.w : ns.x;
                ul_next = S[t] + __builtin_bit_cast(float, pv);
                wd_next = W[t] + sv;
            }
            float left = SM ? Sm[SM ? t : 0] : S[t] + m5;
            if (EARLY && t > 3) left = (t - 3 > q) ? -100000.0f : left;                  // sequence_alignment.cpp:387-389, rows q < A - 4 only
            const float sc = fmaxf(fmaxf(ul, uSm), left);
            const bool isd = (sc == ul), isu = (sc == uSm);                              // diag > up > left on ties (:380-417)
            u32 w;
            if (t == 0) {                                                                // diag and up come from outside the matrix: the path starts here
                const u32 fresh = FXG_K_START(vstart) + FXG_K_SZ1 + (isd ? wd : 0u);
                w = (isd || isu) ? fresh : W[0];
            } else {
                w = isu ? uW : W[t];
                w = isd ? wd : w;
            }
            const u32 wp = w + FXG_K_SZ1;                                                // stored already extended by one gap step, see fxg_clip_row_packed
            const float scm = sc + m5;
            S[t] = sc; W[t] = wp;
            if (SM) Sm[SM ? t : 0] = scm;
            uSm = scm; uW = wp; ul = ul_next; wd = wd_next;
            if (!TRACK) continue;
            if (t < AMIN) {
                const bool gb = sc > best;
                bw = gb ? w : bw;
                best = fmaxf(best, sc);
            } else {
                const bool gb = (sc > best) && (t < A);
                best = gb ? sc : best; bw = gb ? w : bw;
            }
        }
        cp = np; cs = ns;
    }
    if (TRACK) bq = (best > best_in) ? (u32)q : bq;
}

// scores only (pass 1 of fxg_clip_two_pass_k and its re-run up to the first summary row)
template <int AMAX, bool EARLY>
FXG_HD float fxg_clip_row_score_kt(const FxgKArgs &a, int A, u32 c, int q, float (&S)[AMAX], float (&Sm)[AMAX], const uint8_t *ptab)
{
    constexpr bool SM = AMAX <= 48;
    constexpr int AMIN = fxg_clip_k_amin(AMAX, false);
    const u32 po = reinterpret_cast<const uint16_t *>(ptab)[c & 0xFFu];
    const u32x4 *pp = reinterpret_cast<const u32x4 *>(ptab + po);
    const float m5 = fxg_minus5();
    float uSm = -5.0f, rowmax = -1000000.0f;
    u32x4 cp = pp[0];
    float ul = 0.0f + __builtin_bit_cast(float, cp.x);
#pragma unroll
    for (int kb = 0; kb < AMAX; kb += 4) {
        u32x4 np = cp;
        if (kb + 4 < AMAX) np = pp[kb / 4 + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = kb + j;
            if (t >= AMAX) break;
            float ul_next = 0.0f;
            if (t + 1 < AMAX) ul_next = S[t] + __builtin_bit_cast(float, j == 0 ? cp.y : j == 1 ? cp.z : j == 2 ? cp.w : np.x);
            float left = SM ? Sm[SM ? t : 0] : S[t] + m5;
            if (EARLY && t > 3) left = (t - 3 > q) ? -100000.0f : left;
            const float sc = fmaxf(fmaxf(ul, uSm), left);
            const float scm = sc + m5;
            S[t] = sc;
            if (SM) Sm[SM ? t : 0] = scm;
            uSm = scm; ul = ul_next;
            if (t < AMIN) rowmax = fmaxf(rowmax, sc);
            else rowmax = (t < A) ? fmaxf(rowmax, sc) : rowmax;
        }
        cp = np;
    }
    return rowmax;
}

template <int AMAX, bool EARLY, bool TRACK, bool TN = false>
FXG_HD void fxg_clip_row_k(const FxgKArgs &a, int A, u32 c, int q, u32 vstart, float (&S)[AMAX], float (&Sm)[FxgClipK<AMAX>::NSM], u32 (&W)[AMAX],
                           float &best, u32 &bw, u32 &bq, const uint8_t *ptab = nullptr)
{
    if constexpr (fxg_clip_uses_ptab(-AMAX)) { fxg_clip_row_kt<AMAX, EARLY, TRACK>(a, A, c, q, vstart, S, Sm, W, best, bw, bq, ptab); return; }
    constexpr bool SM = FxgClipK<AMAX>::SM;
    constexpr int AMIN = fxg_clip_k_amin(AMAX, TN);
    const bool qn = (c == (u32)'N');
    const float pair_eq = qn ? 0.1f : 1.0f, pair_ne = qn ? 0.1f : -1.0f;                 // sequence_alignment.h:157-169 for an adapter base that is not N
    const float pair_tn = qn ? 0.0f : 0.1f;                                              // ... and for one that is
    const u32 dxr = qn ? 0u : FXG_K_DIA1;
    const float best_in = best;
    float uSm = -5.0f;                                                                   // S[q][-1] - 5
    u32 uW = 0u;
    const bool eq0 = (c == (u32)(uint8_t)a.adapter[0]);
    const bool tn0 = TN && ((u32)(uint8_t)a.adapter[0] == (u32)'N');
    float ul = 0.0f + (tn0 ? pair_tn : (eq0 ? pair_eq : pair_ne));                       // S[q-1][-1] = query_border = 0
    u32 wd = tn0 ? 0u : dxr + (u32)eq0;                                                  // no predecessor left of column 0: the step alone
#pragma unroll
    for (int b0 = 0; b0 < AMAX; b0 += 16) {
        // the match masks of the next 16 diagonals first (a lane mask needs two wait states between its v_cmp and the v_cndmask that reads it)
        bool eq[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) eq[k] = (b0 + k + 1 < AMAX) ? (c == (u32)(uint8_t)a.adapter[b0 + k + 1]) : false;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int t = b0 + k;
            if (t >= AMAX) break;
            float ul_next = 0.0f;
            u32 wd_next = 0u;
            if (t + 1 < AMAX) {
                const bool tn1 = TN && ((u32)(uint8_t)a.adapter[t + 1] == (u32)'N');      // uniform: a scalar condition
                const float pv = eq[k] ? pair_eq : pair_ne;
                ul_next = S[t] + (tn1 ? pair_tn : pv);
                u32 wdx = W[t] + (tn1 ? 0u : dxr);
                FXG_KEEP_V(wdx);                                                         // keeps "+ match" the carry-in of one v_addc_co_u32 (else: select 0/1, or, add)
                wd_next = wdx + (u32)(eq[k] && !tn1);
            }
            float left = SM ? Sm[SM ? t : 0] : S[t] + -5.0f;
            if (EARLY && t > 3) left = (t - 3 > q) ? -100000.0f : left;                  // sequence_alignment.cpp:387-389, rows q < A - 4 only
            const float sc = fmaxf(fmaxf(ul, uSm), left);
            const bool isd = (sc == ul), isu = (sc == uSm);                              // diag > up > left on ties (:380-417)
            u32 w;
            if (t == 0) {                                                                // diag and up come from outside the matrix: the path starts here
                const u32 fresh = FXG_K_START(vstart) + FXG_K_SZ1 + (isd ? wd : 0u);
                w = (isd || isu) ? fresh : W[0];
            } else {
                w = isu ? uW : W[t];
                w = isd ? wd : w;
            }
            const u32 wp = w + FXG_K_SZ1;                                                // stored already extended by one gap step, see fxg_clip_row_packed
            const float scm = sc + -5.0f;
            S[t] = sc; W[t] = wp;
            if (SM) Sm[SM ? t : 0] = scm;
            uSm = scm; uW = wp; ul = ul_next; wd = wd_next;
            if (!TRACK) continue;
            if (t < AMIN) {
                const bool gb = sc > best;
                bw = gb ? w : bw;
                best = fmaxf(best, sc);
            } else {
                const bool gb = (sc > best) && (t < A);
                best = gb ? sc : best; bw = gb ? w : bw;
            }
        }
    }
    if (TRACK) bq = (best > best_in) ? (u32)q : bq;
}

template <int AMAX, bool TN>
FXG_HD void fxg_clip_rows_k(const FxgKArgs &a, const uint8_t *rd, int len, int rows, float &best, u32 &bw, u32 &bq, int &first_n, const bool UR = false, const uint8_t *ptab = nullptr)
{
    float S[AMAX], Sm[FxgClipK<AMAX>::NSM];
    u32 W[AMAX];
    const int A = a.alen;
#pragma unroll
    for (int t = 0; t < AMAX; ++t) {
        S[t] = (t <= 3) ? 0.0f : -5.0f * (float)(t - 3);                                 // target_border (:355-361)
        if (FxgClipK<AMAX>::SM) Sm[FxgClipK<AMAX>::SM ? t : 0] = S[t] + -5.0f;
        W[t] = FXG_K_START(256 + t + 1) + FXG_K_SZ1;                                     // what a path entering diagonally at (0, t + 1) starts from
    }
    const int early_rows = (A - 4 < rows) ? (A - 4 > 0 ? A - 4 : 0) : rows;
    // wave-uniform loops (fxg_wave_max): the early form tests the row number itself and serves every lane while ANY lane is early
    const int early_u = UR ? early_rows : fxg_wave_max(early_rows), rows_u = UR ? rows : fxg_wave_max(rows > 0 ? rows : 0);
    if (rows_u <= 0) return;
    int q = 0;
    u32 cn = rd[0];
#pragma unroll 1
    for (; q < early_u; ++q) {
        const u32 c = cn;
        cn = rd[q + 1];
        if (!UR && q >= rows) continue;
        first_n = (c == (u32)'N' && first_n == len && q < len) ? q : first_n;
        fxg_clip_row_k<AMAX, true, true, TN>(a, A, c, q, (u32)q, S, Sm, W, best, bw, bq, ptab);
    }
#pragma unroll 1
    for (; q < rows_u; ++q) {
        const u32 c = cn;
        cn = rd[q + 1];
        if (!UR && q >= rows) continue;
        first_n = (c == (u32)'N' && first_n == len && q < len) ? q : first_n;
        fxg_clip_row_k<AMAX, false, true, TN>(a, A, c, q, (u32)q, S, Sm, W, best, bw, bq, ptab);
    }
}

// ------------------------------------------------------------------------------------------------
// The same two passes as fxg_clip_two_pass for 17..99 adapter columns, where the checkpoints no longer fit in registers: pass 1
// (scores only, 6-7 VALU instructions per cell) leaves its score row in global scratch every a.clip_ck_rows rows -- slot j - 1 holds
// the row before row j * clip_ck_rows, [slot][column][thread] so that a wave's store is one line per column -- and finds the row bq1
// of the first maximum.  The best path covers at most SPAN = A + (A + 1) / 5 rows (see fxg_clip_two_pass), i.e. starts in row
// r0 = bq1 - SPAN + 1 or later: the scores are re-run from the last checkpoint at or before r0 up to r0, then rows r0..bq1 carry the
// summaries (fxg_clip_row_k), and only row bq1 looks for the best cell.  A path that starts in pass 2 starts in column 0 (or in row 0
// when r0 = 0), so its `start` is the row RELATIVE to r0: reads of any length fit the 9 bits.
// ------------------------------------------------------------------------------------------------
template <int AMAX, bool EARLY, bool TN>
FXG_HD float fxg_clip_row_score_k(const FxgKArgs &a, int A, u32 c, int q, float (&S)[AMAX], float (&Sm)[AMAX], const uint8_t *ptab = nullptr)
{
    if constexpr (fxg_clip_uses_ptab(-AMAX)) return fxg_clip_row_score_kt<AMAX, EARLY>(a, A, c, q, S, Sm, ptab);
    constexpr bool SM = AMAX <= 48;                 // the score rows keep S - 5 as well where that still leaves room: nothing else is live while they run
    constexpr int AMIN = fxg_clip_k_amin(AMAX, TN);
    const bool qn = (c == (u32)'N');
    const float pair_eq = qn ? 0.1f : 1.0f, pair_ne = qn ? 0.1f : -1.0f;
    const float pair_tn = qn ? 0.0f : 0.1f;
    float uSm = -5.0f, rowmax = -1000000.0f;
    const bool tn0 = TN && ((u32)(uint8_t)a.adapter[0] == (u32)'N');
    float ul = 0.0f + (tn0 ? pair_tn : ((c == (u32)(uint8_t)a.adapter[0]) ? pair_eq : pair_ne));
#pragma unroll
    for (int b0 = 0; b0 < AMAX; b0 += 16) {
        bool eq[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) eq[k] = (b0 + k + 1 < AMAX) ? (c == (u32)(uint8_t)a.adapter[b0 + k + 1]) : false;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int t = b0 + k;
            if (t >= AMAX) break;
            float ul_next = 0.0f;
            if (t + 1 < AMAX) {
                const bool tn1 = TN && ((u32)(uint8_t)a.adapter[t + 1] == (u32)'N');
                const float pv = eq[k] ? pair_eq : pair_ne;
                ul_next = S[t] + (tn1 ? pair_tn : pv);
            }
            float left = SM ? Sm[SM ? t : 0] : S[t] + -5.0f;
            if (EARLY && t > 3) left = (t - 3 > q) ? -100000.0f : left;
            const float sc = fmaxf(fmaxf(ul, uSm), left);
            const float scm = sc + -5.0f;
            S[t] = sc;
            if (SM) Sm[SM ? t : 0] = scm;
            uSm = scm; ul = ul_next;
            if (t < AMIN) rowmax = fmaxf(rowmax, sc);
            else rowmax = (t < A) ? fmaxf(rowmax, sc) : rowmax;
        }
    }
    return rowmax;
}

#ifdef FXG_CLIP_DEBUG
#define FXG_CLIP_DBG_WORDS 512u   // per read: [0,16) scalars and hashes, [16,272) level 3: (hash S, hash W) after each summary row, [272,464): S at r0, S and W before the last row
#define FXG_CLIP_DBG(i, v) do { if (dbg) dbg[i] = (u32)(v); } while (0)
FXG_HD u32 fxg_fbits(float f) { union { float f; u32 u; } x; x.f = f; return x.u; }
template <int N> FXG_HD u32 fxg_dbg_hash(const float (&S)[N]) { u32 h = 0; for (int t = 0; t < N; ++t) h = h * 31u + fxg_fbits(S[t]); return h; }
template <int N> FXG_HD u32 fxg_dbg_hash(const u32 (&W)[N]) { u32 h = 0; for (int t = 0; t < N; ++t) h = h * 31u + W[t]; return h; }
#else
FXG_HD u32 fxg_fbits(float) { return 0u; }
#define FXG_CLIP_DBG(i, v) do { } while (0)
#endif
// returns r0 (the row the `start` field of bw counts from)
template <int AMAX, bool TN, bool GL = false>      // GL: as in fxg_clip_two_pass
FXG_HD int fxg_clip_two_pass_k(const FxgKArgs &a, const uint8_t *rd, int len, int rows, float *ck, u32 cks, float &best, u32 &bw, u32 &bq, int &first_n, u32 *dbg = nullptr, const bool UR = false,
                               const uint8_t *ptab = nullptr)
{
    (void)dbg;
    float S[AMAX], Sm[AMAX];
    const int A = a.alen, K = (int)a.clip_ck_rows;
    const int early_rows = (A - 4 < rows) ? (A - 4 > 0 ? A - 4 : 0) : rows;
    // every loop runs a wave-uniform number of trips (fxg_wave_max): in pass 1 the row number q and the checkpoint schedule are scalars, a
    // lane past its own last row skips the row (and the checkpoint, which it will never read); the early form tests the row number
    // itself and serves every lane while ANY lane is early
    const int early_u = UR ? early_rows : fxg_wave_max(early_rows), rows_u = UR ? rows : fxg_wave_max(rows > 0 ? rows : 0);
    if (rows_u <= 0) return 0;
#pragma unroll
    for (int t = 0; t < AMAX; ++t) { S[t] = (t <= 3) ? 0.0f : -5.0f * (float)(t - 3); Sm[t] = S[t] + -5.0f; }
    // ---- pass 1 ----
    float b1 = -1000000.0f;
    int bq1 = 0, next_ck = K, q = 0;
    float *slot = ck;
    u32 cn = 0u, gw0 = 0u, gw1 = 0u;
    int gmax = 0;
    if constexpr (GL) {
        gmax = fxg_gl_last_dword(a.clip_total, (u64)(rd - a.clip_src));
        gw0 = fxg_ld32(rd, 0, gmax); gw1 = fxg_ld32(rd, 1, gmax);
    } else cn = rd[0];
#define FXG_CK_ROW(EARLY)                                                                                                    \
    {                                                                                                                        \
        const bool mine = UR || q < rows;                                                                                    \
        if (q == next_ck) {                                                                                                  \
            if (mine) { _Pragma("unroll") for (int t = 0; t < AMAX; ++t) slot[(size_t)t * cks] = S[t]; }                     \
            slot += (size_t)AMAX * cks; next_ck += K;                                                                        \
        }                                                                                                                    \
        u32 c;                                                                                                               \
        if constexpr (GL) {                                                                                                  \
            c = (gw0 >> ((u32)(q & 3) << 3)) & 0xFFu;                                                                        \
            if ((q & 3) == 3) { gw0 = gw1; gw1 = fxg_ld32(rd, (q >> 2) + 2, gmax); }                                         \
        } else { c = cn; cn = rd[q + 1]; }                                                                                   \
        if (mine) {                                                                                                          \
            const float rm = fxg_clip_row_score_k<AMAX, EARLY, TN>(a, A, c, q, S, Sm, ptab);                                 \
            const bool g = rm > b1;                                                                                          \
            b1 = g ? rm : b1; bq1 = g ? q : bq1;                                                                             \
        }                                                                                                                    \
    }
#pragma unroll 1
    for (; q < early_u; ++q) FXG_CK_ROW(true)
#pragma unroll 1
    for (; q < rows_u; ++q) FXG_CK_ROW(false)
#undef FXG_CK_ROW
    if (rows <= 0) return 0;                                // (only now: the loops above are the wave's, not the lane's)
    // ---- pass 2 ----
    const int span = A + (A + 1) / 5;
    const int r0 = bq1 - span + 1 > 0 ? bq1 - span + 1 : 0;
    const int j0 = r0 / K;
    q = j0 * K;
    FXG_CLIP_DBG(0, bq1); FXG_CLIP_DBG(1, fxg_fbits(b1)); FXG_CLIP_DBG(2, r0); FXG_CLIP_DBG(3, j0);
    {
        const float *from = ck + (size_t)(j0 > 0 ? j0 - 1 : 0) * AMAX * cks;
#pragma unroll
        for (int t = 0; t < AMAX; ++t) {
            const float border = (t <= 3) ? 0.0f : -5.0f * (float)(t - 3);
            S[t] = j0 > 0 ? from[(size_t)t * cks] : border;
            Sm[t] = S[t] + -5.0f;
        }
    }
    // Every lane has its own rows here, so the loops keep ONE body each: the early form (which tests the row number itself) wherever a
    // row below A - 4 can occur -- the < clip_ck_rows rows of the score re-run and the first A - 4 rows of the summary window.
#if defined(FXG_CLIP_DEBUG) && FXG_CLIP_DEBUG >= 2
    FXG_CLIP_DBG(4, fxg_dbg_hash(S));
#endif
    {
        const int n0 = r0 - q, n0u = fxg_wave_max(n0);      // < clip_ck_rows rows of scores up to the first summary row
#pragma unroll 1
        for (int i = 0; i < n0u; ++i) {
            if (i >= n0) continue;
            (void)fxg_clip_row_score_k<AMAX, true, TN>(a, A, (u32)rd[q], q, S, Sm, ptab);
            ++q;
        }
    }
#if defined(FXG_CLIP_DEBUG) && FXG_CLIP_DEBUG >= 2
    FXG_CLIP_DBG(5, fxg_dbg_hash(S));
#endif
    u32 W[AMAX];
#pragma unroll
    for (int t = 0; t < AMAX; ++t) W[t] = FXG_K_START(256 + t + 1) + FXG_K_SZ1;         // the cells above row 0 (r0 > 0: never on the best path)
    float (&Sk)[FxgClipK<AMAX>::NSM] = reinterpret_cast<float (&)[FxgClipK<AMAX>::NSM]>(Sm);    // the summary rows keep S - 5 only where registers allow
#ifdef FXG_CLIP_DEBUG
    constexpr int DT0 = AMAX > 64 ? AMAX - 64 : 0;          // the LAST 64 columns of the wide buckets
    (void)DT0;
#endif
#if defined(FXG_CLIP_DEBUG) && FXG_CLIP_DEBUG >= 3
    if (dbg) for (int t = DT0; t < AMAX; ++t) dbg[272 + t - DT0] = fxg_fbits(S[t]);
#define FXG_CLIP_DBG_ROW() do { const int wr_ = q - r0; if (dbg && wr_ < 128) { dbg[16 + 2 * wr_] = fxg_dbg_hash(S); dbg[17 + 2 * wr_] = fxg_dbg_hash(W); } } while (0)
#else
#define FXG_CLIP_DBG_ROW() do { } while (0)
#endif
    const int win = bq1 - r0, n1 = win < early_rows ? win : early_rows, n2 = win - n1;      // summary rows in the early form / in the other
    const int n1u = fxg_wave_max(n1), n2u = fxg_wave_max(n2);
#pragma unroll 1
    for (int i = 0; i < n1u; ++i) {
        if (i >= n1) continue;
        fxg_clip_row_k<AMAX, true, false, TN>(a, A, (u32)rd[q], q, (u32)(q - r0), S, Sk, W, best, bw, bq, ptab); FXG_CLIP_DBG_ROW();
        ++q;
    }
#pragma unroll 1
    for (int i = 0; i < n2u; ++i) {
        if (i >= n2) continue;
        fxg_clip_row_k<AMAX, false, false, TN>(a, A, (u32)rd[q], q, (u32)(q - r0), S, Sk, W, best, bw, bq, ptab); FXG_CLIP_DBG_ROW();
        ++q;
    }
#if defined(FXG_CLIP_DEBUG) && FXG_CLIP_DEBUG >= 3
    if (dbg) for (int t = DT0; t < AMAX; ++t) { dbg[336 + t - DT0] = fxg_fbits(S[t]); dbg[400 + t - DT0] = W[t]; }
#endif
#ifdef FXG_CLIP_DEBUG_W      // lighter probes (the level-3 build no longer failed): only W / only a window of W before the last row
    if (dbg) for (int t = (FXG_CLIP_DEBUG_W); t < AMAX && t < 64; ++t) dbg[400 + t] = W[t];
#endif
#ifdef FXG_CLIP_DEBUG_S
    if (dbg) for (int t = (FXG_CLIP_DEBUG_S); t < AMAX && t < 64; ++t) dbg[336 + t] = fxg_fbits(S[t]);
#endif
#if defined(FXG_CLIP_DEBUG) && FXG_CLIP_DEBUG >= 2
    FXG_CLIP_DBG(10, fxg_dbg_hash(S)); FXG_CLIP_DBG(11, fxg_dbg_hash(W));
#endif
    fxg_clip_row_k<AMAX, true, true, TN>(a, A, (u32)rd[bq1], bq1, (u32)(bq1 - r0), S, Sk, W, best, bw, bq, ptab);
    FXG_CLIP_DBG(6, fxg_fbits(best)); FXG_CLIP_DBG(7, bw); FXG_CLIP_DBG(8, bq);
#if defined(FXG_CLIP_DEBUG) && FXG_CLIP_DEBUG >= 2
    FXG_CLIP_DBG(12, fxg_dbg_hash(S)); FXG_CLIP_DBG(13, fxg_dbg_hash(W));
#endif
    if (!(a.clip_flags & FXG_CLIP_KEEP_N)) {                                             // the -n rule needs the first N of the read itself
        first_n = fxg_clip_first_n(rd, len, UR ? len : fxg_wave_max(len), a.clip_stride, first_n);
    }
    return r0;
}

// KFORM: the one-word summary of fxg_clip_row_k (17..99 columns; also 16 columns for reads beyond 255 bases, which the register
// form of fxg_clip_two_pass cannot describe: its start field is absolute)
template <int AMAX, bool KFORM, bool TN = false, bool GL = false>
FXG_HD void fxg_clip_read_packed(const FxgKArgs &a, const uint8_t *rd, int len, int rows,
                                 u32 *out_len, u32 *keep, u32 *reason, u32 *clipped, u32 *adapter_only, float *ck = nullptr, u32 cks = 0u, u32 *dbg = nullptr, const bool UR = false,
                                 const uint8_t *ptab = nullptr)
{
    (void)dbg; (void)ptab;
    float best = -1000000.0f;
    u32 bw = FXG_INVALID_TUPLE, bq = 0u;
    int first_n = len, qbase = 0;
#ifndef FXG_CLIP_ONE_PASS
    if constexpr (!KFORM) qbase = fxg_clip_two_pass<AMAX, GL>(a, rd, len, rows, best, bw, bq, first_n, UR, ptab);
    else
#endif
    if constexpr (!KFORM) fxg_clip_rows_packed<AMAX, false>(a, rd, len, rows, best, bw, bq, first_n, UR);
    if constexpr (KFORM) {
        int r0 = 0;
        if (ck) r0 = fxg_clip_two_pass_k<AMAX, TN, GL>(a, rd, len, rows, ck, cks, best, bw, bq, first_n, dbg, UR, ptab);
        else fxg_clip_rows_k<AMAX, TN>(a, rd, len, rows, best, bw, bq, first_n, UR, ptab);
        const int v = (int)(bw >> 23), matches = (int)(bw & 127u), diag = (int)((bw >> 7) & 127u);
        fxg_clip_finish(a, len, v < 256 ? r0 + v : 0, v < 256 ? 0 : v - 256, diag - matches, (int)((bw >> 14) & 511u), matches,
                        (int)bq, first_n, out_len, keep, reason, clipped, adapter_only);
        return;
    }
    const int matches = (int)(bw & 31u), diag = (int)((bw >> 14) & 31u);
    fxg_clip_finish(a, len, qbase + (int)(bw >> 24), (int)((bw >> 19) & 31u), diag - matches, (int)((bw >> 5) & 511u), matches,
                    (int)bq, first_n, out_len, keep, reason, clipped, adapter_only);
}

// The -v report counters are a pure function of res[]: one accumulator set per thread of the
// counting pass (or per emulated run).
struct FxgCounts {
    u64 in, kept, bases, too_short, adapter_only, no_adapter, adapter_found, has_n, qtrim, qfilter, ftrim, k_mode, artifact;
};
#define FXG_RES_ADAPTER_ONLY_BIT 22   /* clipper found the adapter at index 0 (counted even when -k keeps the read) */

FXG_HD void fxg_count_res(u32 w, FxgCounts &c)
{
    const u32 keep = (w >> 16) & 1u, why = (w >> 17) & 0xFu;
    c.in++;
    c.kept += keep;
    c.bases += keep ? (w & 0xFFFFu) : 0u;
    c.adapter_only += (w >> FXG_RES_ADAPTER_ONLY_BIT) & 1u;
    c.too_short += (why == FXG_R_CLIP_TOO_SHORT);
    c.no_adapter += (why == FXG_R_CLIP_NO_ADAPTER);
    c.adapter_found += (why == FXG_R_CLIP_ADAPTER_FOUND);
    c.has_n += (why == FXG_R_CLIP_N);
    c.qtrim += (why == FXG_R_QTRIM);
    c.qfilter += (why == FXG_R_QFILTER);
    c.ftrim += (why == FXG_R_FTRIM);
    c.k_mode += (why == FXG_R_CLIP_K_MODE);
    c.artifact += (why == FXG_R_ARTIFACT);
}

// counters[] from the accumulators; clip_out / qtrim_out are what survives each stage of the chain
FXG_HD void fxg_counts_to_slots(const FxgCounts &c, u32 stages, u64 *slot)
{
    for (int i = 0; i < FXG_NCOUNTERS; ++i) slot[i] = 0;
    slot[FXG_C_INPUT] = c.in; slot[FXG_C_KEPT] = c.kept; slot[FXG_C_KEPT_BASES] = c.bases;
    slot[FXG_C_CLIP_TOO_SHORT] = c.too_short; slot[FXG_C_CLIP_ADAPTER_ONLY] = c.adapter_only;
    slot[FXG_C_CLIP_NO_ADAPTER] = c.no_adapter; slot[FXG_C_CLIP_ADAPTER_FOUND] = c.adapter_found; slot[FXG_C_CLIP_N] = c.has_n;
    slot[FXG_C_QTRIM_DROPPED] = c.qtrim; slot[FXG_C_QFILTER_DROPPED] = c.qfilter; slot[FXG_C_FTRIM_DROPPED] = c.ftrim;
    const u64 clip_out = c.kept + c.qtrim + c.qfilter;
    slot[FXG_C_CLIP_OUT] = (stages & FXG_STAGE_CLIP) ? clip_out : 0;
    slot[FXG_C_QTRIM_OUT] = (stages & FXG_STAGE_QTRIM) ? c.kept + c.qfilter : 0;
    slot[FXG_C_ARTIFACT_DROPPED] = c.artifact;
}

// ---- per-thread phase bodies (host+device so tests/emu can run them serially) ----

#ifndef FXG_BITMAP_U
#define FXG_BITMAP_U 5
#endif
// phase 1: quality rows of the tile -> two bitmaps.  Full in-range tiles take the batched path: five
// independent 16-byte loads per lane are issued before any is consumed.
FXG_HD void fxg_phase_bitmaps(const FxgKArgs &a, u64 tb, u32 tbytes, u32 *bm_g, u32 *bm_l, u32 tid, u32 nthreads)
{
    const u32 Kg = (128u - a.tq) * 0x01010101u, Kf = (128u - a.fq) * 0x01010101u;
    const bool same = a.tq == a.fq;      // trimmer and filter at the same threshold (the usual pipe): "below" is the complement of "at least"
    const u32 nchunks = (tbytes + 15u) >> 4;
    uint16_t *g16 = reinterpret_cast<uint16_t *>(bm_g), *l16 = reinterpret_cast<uint16_t *>(bm_l);
    const bool one = bm_l == bm_g;       // one shared bitmap (fxg_bitmap_count): the filter counts its complement
    const uint8_t *src = a.qual + tb;
    if ((tbytes & 15u) == 0u && tb + tbytes <= a.total_bytes) {
        constexpr u32 U = FXG_BITMAP_U;
        for (u32 c0 = tid; c0 < nchunks; c0 += nthreads * U) {
            u32x4 v[U];
#pragma unroll
            for (u32 u = 0; u < U; ++u) {
                const u32 c = c0 + u * nthreads;
                if (c < nchunks) v[u] = fxg_ld16(src + ((u64)c << 4));
            }
#pragma unroll
            for (u32 u = 0; u < U; ++u) {
                const u32 c = c0 + u * nthreads;
                if (c < nchunks) { const u32 g = fxg_mask16(v[u], Kg); g16[c] = (uint16_t)g; if (!one) l16[c] = (uint16_t)(~(same ? g : fxg_mask16(v[u], Kf))); }
            }
        }
        return;
    }
    for (u32 c = tid; c < nchunks; c += nthreads) {
        const u32 o = c << 4;
        const u32x4 v = fxg_window(a.qual, (long long)(tb + o), a.total_bytes, 0, (int)(tbytes - o < 16u ? tbytes - o : 16u));
        g16[c] = (uint16_t)fxg_mask16(v, Kg);
        if (!one) l16[c] = (uint16_t)(~fxg_mask16(v, Kf));
    }
}

// phase 1 (clipper, census): rows of the tile -> LDS, so that one thread can walk one read.  tb/tbytes/total are in bytes of `src`
FXG_HD void fxg_phase_stage_bases(const uint8_t *src, u64 total, u64 tb, u32 tbytes, uint8_t *sb, u32 tid, u32 nthreads)
{
    const u32 nchunks = (tbytes + 15u) >> 4;
    if ((tb & 15u) == 0u && (tbytes & 15u) == 0u && tb + tbytes <= total) {
        // whole aligned chunks inside the array (every tile but a batch's last): four independent loads per lane in flight before any is stored --
        // one after the other the 6-10 trips of a tile each paid a memory round trip
        constexpr u32 U = 4;
        for (u32 c0 = tid; c0 < nchunks; c0 += nthreads * U) {
            u32x4 v[U];
#pragma unroll
            for (u32 u = 0; u < U; ++u) { const u32 c = c0 + u * nthreads; if (c < nchunks) v[u] = fxg_ld16(src + tb + ((u64)c << 4)); }
#pragma unroll
            for (u32 u = 0; u < U; ++u) { const u32 c = c0 + u * nthreads; if (c < nchunks) *reinterpret_cast<u32x4 *>(sb + (c << 4)) = v[u]; }
        }
        return;
    }
    for (u32 c = tid; c < nchunks; c += nthreads) {
        const u32 o = c << 4;
        const u64 at = tb + o;
        u32x4 v;
        if ((at & 15u) == 0u) v = fxg_window(src, (long long)at, total, 0, (int)(tbytes - o < 16u ? tbytes - o : 16u));
        else {                                             // rows of an extended-query array need not start 16-byte aligned
            u64 lo = 0, hi = 0;
            const int nb = (int)(tbytes - o < 16u ? tbytes - o : 16u);
            for (int i = 0; i < nb; ++i) { const u64 b = src[at + i]; if (i < 8) lo |= b << (8 * i); else hi |= b << (8 * (i - 8)); }
            v = (u32x4){(u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32)};
        }
        *reinterpret_cast<u32x4 *>(sb + o) = v;
    }
}

// phase 2, group A: thread tid decides read r0 + tid
// AMAX > 0: general clipper (fallback, FXG_NO_PACKED_CLIP); AMAX < 0: packed clipper with bucket -AMAX columns (-216: 16 columns in the form of the 17..99 buckets, ablation build only)
//   -(300 + columns): the same form for adapters that contain 'N' (buckets 16 24 36 48 64 100)
__host__ __device__ constexpr int fxg_clip_cols(int amax) { return amax <= -300 ? -amax - 300 : amax <= -200 ? -amax - 200 : -amax; }
__host__ __device__ constexpr bool fxg_clip_kform(int amax) { return amax < -16; }
__host__ __device__ constexpr bool fxg_clip_tn(int amax) { return amax <= -300; }
// GL (two-pass forms only): the DP reads the batch in global memory, nothing was staged (sb unused)
template <int AMAX, bool GL = false>
FXG_HD u32 fxg_decide_a(const FxgKArgs &a, const u32 *bm_g, const u32 *bm_l, const uint8_t *sb, u32 r0, u32 tid,
                        u32 *keep_out, u32 *len_out, float *ck = nullptr, u32 cks = 0u, const uint8_t *ptab = nullptr)      // ck: this thread's checkpoint scratch (fxg_clip_two_pass_k), cks its stride; ptab: the workgroup's pair table (fxg_clip_ptab_build)
{
    const u32 stride = a.stride;
    const u32 rl = a.len ? (u32)a.len[r0 + tid] : a.fixed_len;
    u32 reason = FXG_R_KEPT, clipped = 0, keep = 1, curlen = rl, ao = 0;
    const int rows = a.wlen ? (int)a.wlen[r0 + tid] : (int)rl;      // clip history: the DP also runs over the stale tail (fxg_history.h)
    if constexpr (AMAX > 0) fxg_clip_read<AMAX>(a, sb + tid * a.clip_stride, (int)rl, rows, &curlen, &keep, &reason, &clipped, &ao, !a.len && !a.wlen);
    if constexpr (AMAX < 0) {
        // fixed-length batch without clip history: the row count is a scalar, so every loop of the DP is a scalar loop
        constexpr int COLS = fxg_clip_cols(AMAX);
        constexpr bool KF = fxg_clip_kform(AMAX), TN = fxg_clip_tn(AMAX);
#ifdef FXG_CLIP_DEBUG
        u32 *dbg = a.clip_dbg ? a.clip_dbg + (size_t)(r0 + tid) * FXG_CLIP_DBG_WORDS : nullptr;
#else
        u32 *dbg = nullptr;
#endif
        const uint8_t *rd = GL ? a.clip_src + (u64)(r0 + tid) * a.clip_stride : sb + tid * a.clip_stride;
        if (!a.len && !a.wlen) fxg_clip_read_packed<COLS, KF, TN, GL>(a, rd, (int)a.fixed_len, (int)a.fixed_len, &curlen, &keep, &reason, &clipped, &ao, ck, cks, dbg, true, ptab);
        else fxg_clip_read_packed<COLS, KF, TN, GL>(a, rd, (int)rl, rows, &curlen, &keep, &reason, &clipped, &ao, ck, cks, dbg, false, ptab);
    }
    if (keep && (a.stages & FXG_STAGE_QTRIM)) {             // fastq_quality_trimmer.c:94-101
        const u32 k = fxg_bits_last(bm_g, tid * stride, curlen);
        curlen = k;
        if (!(k > 0 && (int)k >= a.qt_min_len)) { keep = 0; reason = FXG_R_QTRIM; }
    }
    if (keep && (a.stages & FXG_STAGE_QFILTER)) {           // fastq_quality_filter.c:110-129,155 in closed form
        const u32 low = bm_l == bm_g ? curlen - fxg_bits_count(bm_g, tid * stride, curlen) : fxg_bits_count(bm_l, tid * stride, curlen);
        int n0 = (int)curlen * a.qf_keep_pct / 100;
        if (n0 < 0) n0 = 0;
        if (a.qf_drop_all || (int)low > n0) { keep = 0; reason = FXG_R_QFILTER; }
    }
    const u32 w = (curlen & 0xFFFFu) | (keep << 16) | (reason << 17) | (clipped << 21) | (ao << FXG_RES_ADAPTER_ONLY_BIT);
    a.res[r0 + tid] = w;
    *keep_out = keep; *len_out = curlen;
    return w;
}

// phase 2, group B: fixed trimming is arithmetic on the length; reverse-complement only moves the anchor
template <bool REV>
FXG_HD u32 fxg_decide_b(const FxgKArgs &a, u32 r0, u32 tid, u32 *keep_out, u32 *len_out, u32 *anchor_out)
{
    const u32 rl = a.len ? (u32)a.len[r0 + tid] : a.fixed_len;
    u32 reason = FXG_R_KEPT, start = 0, keep = 1, curlen = rl;
    if (a.stages & FXG_STAGE_FTRIM) {                       // fastx_trimmer.c:122-134
        if (a.ft_last != 0 && (u32)a.ft_last < curlen) curlen = (u32)a.ft_last;
        if (a.ft_first != 1) {
            if (curlen < (u32)a.ft_first) keep = 0;
            else { start = (u32)a.ft_first - 1u; curlen -= start; }
        }
    }
    if (keep && (a.stages & FXG_STAGE_FTRIM_END)) {         // fastx_trimmer.c:136-144
        if (curlen <= a.ft_trim_end) keep = 0;
        else if (curlen - a.ft_trim_end < a.ft_min_len) keep = 0;
        else curlen -= a.ft_trim_end;
    }
    if (!keep) reason = FXG_R_FTRIM;
    const u32 w = (curlen & 0xFFFFu) | (keep << 16) | (reason << 17);
    a.res[r0 + tid] = w;
    // output byte k of this read comes from source byte anchor + k (forward) or anchor - k (reverse-complement)
    *anchor_out = REV ? tid * a.stride + (rl - 1u - start) : tid * a.stride + start;
    *keep_out = keep; *len_out = curlen;
    return w;
}

// fastq_masker (fastq_masker.c:92-108): every read is kept at full length; n_low = bases below the threshold
FXG_HD u32 fxg_decide_mask(const FxgKArgs &a, const u32 *bm_l, u32 r0, u32 tid, u32 *keep_out, u32 *len_out, u32 *n_low)
{
    const u32 rl = a.len ? (u32)a.len[r0 + tid] : a.fixed_len;
    *n_low = fxg_bits_count(bm_l, tid * a.stride, rl);
    a.res[r0 + tid] = (rl & 0xFFFFu) | (1u << 16);
    *keep_out = 1u; *len_out = rl;
    return (rl & 0xFFFFu) | (1u << 16);
}

// base census of one read, shared by
//   fastx_artifacts_filter (fastx_artifacts_filter.c:56-112): drop when one of A/C/G/T fills all but <= 3 positions
//   fastq_to_fasta N-discard (fastq_to_fasta.c:79-82)        : drop when the read contains an N (unless -n)
// Round 6, second form: the tile's bases are not staged in LDS and walked one read per thread any more (150 ds_read_u8 and ~1 650 VALU instructions per read; four
// bases per step through aligned dwords and two v_perm tables took 6.90 -> 6.28 ms of 50 M x 150, profiles/r06/census_swar_vs_bytes.txt).  Phase 1 streams the rows
// once, 16 bytes per lane and five loads in flight like the quality bitmaps, and leaves FOUR bitmaps of the tile in LDS, one bit per base: bit 1 and bit 2 of the
// letter (A 00, C 01, T 10, G and N 11), "is N", and "is none of A C G T N" (upper case only, as the reference's tools have it here: fastx_artifacts_filter.c:70-95,
// anything else is "invalid nucleotide value").  A read's census is then five population counts over the 5-6 words of its bit range.  The bitmaps take half the
// bytes of the staged tile, so the kernel is a streaming instance like the masker now: tiles of 20 KB and the granule table of the write-out (fxg_lds_layout).
#define FXG_CENSUS_EXP_LO 0x43FF41FFu      // the byte a letter has to be, by its low three bits: -- A -- C | T -- N G   (FF: no byte with those bits is valid)
#define FXG_CENSUS_EXP_HI 0x474EFF54u
struct FxgCensusBm { u32 *b0, *b1, *bn, *bi; };
__host__ __device__ inline u32 fxg_census_words(u32 T, u32 stride) { return fxg_r16(((T * stride + 31u) / 32u + 2u) * 4u) / 4u; }
FXG_HD FxgCensusBm fxg_census_bitmaps(u32 *p, u32 T, u32 stride)      // p: off_bm_g of a layout with four bitmaps
{
    const u32 w = fxg_census_words(T, stride);
    return FxgCensusBm{p, p + w, p + 2u * w, p + 3u * w};
}
FXG_HD void fxg_census_chunk(const u32x4 &v, u32 c, const FxgCensusBm &bm)
{
    const u32 w[4] = {v.x, v.y, v.z, v.w};
    u32 f0[4], f1[4], fn[4], fi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f0[i] = (w[i] << 6) & 0x80808080u;                                                 // bit 1 of the letter
        f1[i] = (w[i] << 5) & 0x80808080u;                                                 // bit 2
        fn[i] = ~fxg_nonzero_flags(w[i] ^ 0x4E4E4E4Eu) & 0x80808080u;                      // 'N'
        fi[i] = fxg_nonzero_flags(fxg_perm(FXG_CENSUS_EXP_HI, FXG_CENSUS_EXP_LO, w[i] & 0x07070707u) ^ w[i]);
    }
    reinterpret_cast<uint16_t *>(bm.b0)[c] = (uint16_t)fxg_flags16(f0[0], f0[1], f0[2], f0[3]);
    reinterpret_cast<uint16_t *>(bm.b1)[c] = (uint16_t)fxg_flags16(f1[0], f1[1], f1[2], f1[3]);
    reinterpret_cast<uint16_t *>(bm.bn)[c] = (uint16_t)fxg_flags16(fn[0], fn[1], fn[2], fn[3]);
    reinterpret_cast<uint16_t *>(bm.bi)[c] = (uint16_t)fxg_flags16(fi[0], fi[1], fi[2], fi[3]);
}
// phase 1 (census): base rows of the tile -> the four bitmaps (the loop of fxg_phase_bitmaps)
FXG_HD void fxg_phase_census(const FxgKArgs &a, u64 tb, u32 tbytes, const FxgCensusBm &bm, u32 tid, u32 nthreads)
{
    const u32 nchunks = (tbytes + 15u) >> 4;
    const uint8_t *src = a.bases + tb;
    if ((tbytes & 15u) == 0u && tb + tbytes <= a.total_bytes) {
        constexpr u32 U = FXG_BITMAP_U;
        for (u32 c0 = tid; c0 < nchunks; c0 += nthreads * U) {
            u32x4 v[U];
#pragma unroll
            for (u32 u = 0; u < U; ++u) { const u32 c = c0 + u * nthreads; if (c < nchunks) v[u] = fxg_ld16(src + ((u64)c << 4)); }
#pragma unroll
            for (u32 u = 0; u < U; ++u) { const u32 c = c0 + u * nthreads; if (c < nchunks) fxg_census_chunk(v[u], c, bm); }
        }
        return;
    }
    for (u32 c = tid; c < nchunks; c += nthreads) {
        const u32 o = c << 4;
        fxg_census_chunk(fxg_window(a.bases, (long long)(tb + o), a.total_bytes, 0, (int)(tbytes - o < 16u ? tbytes - o : 16u)), c, bm);
    }
}
// s0: the read's first bit in the tile's bitmaps
FXG_HD u32 fxg_decide_census(const FxgKArgs &a, const FxgCensusBm &bm, u32 s0, u32 r0, u32 tid, u32 *keep_out, u32 *len_out, u32 *bad)
{
    const u32 rl = a.len ? (u32)a.len[r0 + tid] : a.fixed_len;
    u32 ca = 0, cc = 0, cg = 0, ct = 0, cn = 0, invalid = 0;
    if (rl) {
        const u32 e1 = s0 + rl - 1u, w0 = s0 >> 5, w1 = e1 >> 5;
        for (u32 w = w0; w <= w1; ++w) {
            u32 m = 0xFFFFFFFFu;
            if (w == w0) m &= 0xFFFFFFFFu << (s0 & 31u);
            if (w == w1) m &= 0xFFFFFFFFu >> (31u - (e1 & 31u));
            const u32 x0 = bm.b0[w], x1 = bm.b1[w], xn = bm.bn[w];
            invalid |= bm.bi[w] & m;
            ca += (u32)__builtin_popcount(~x0 & ~x1 & m); cc += (u32)__builtin_popcount(x0 & ~x1 & m); ct += (u32)__builtin_popcount(~x0 & x1 & m);
            cg += (u32)__builtin_popcount(x0 & x1 & ~xn & m); cn += (u32)__builtin_popcount(xn & m);
        }
    }
    if (invalid) *bad = 1u;                               // "invalid nucleotide value" in the reference
    u32 keep = 1u, why = FXG_R_KEPT;
    if (a.stages & FXG_STAGE_ARTIFACTS) {
        const int lim = (int)rl - 3;
        if (((int)ca >= lim) | ((int)cc >= lim) | ((int)cg >= lim) | ((int)ct >= lim)) { keep = 0u; why = FXG_R_ARTIFACT; }
    }
    if ((a.stages & FXG_STAGE_NFILTER) && !a.nf_keep_n && cn != 0u) { keep = 0u; why = FXG_R_HAS_N; }
    a.res[r0 + tid] = (rl & 0xFFFFu) | (keep << 16) | (why << 17);
    *keep_out = keep; *len_out = rl;
    return (rl & 0xFFFFu) | (keep << 16) | (why << 17);
}

// per-kept-read side outputs
FXG_HD void fxg_write_kept_meta(const FxgKArgs &a, u64 rank, u32 olen, u32 read_index, u64 byte_off)
{
    if (a.out_len) a.out_len[rank] = (uint16_t)olen;
    if (a.kept_index) a.kept_index[rank] = read_index;
    if (a.out_off) a.out_off[rank] = byte_off;
}

#ifndef FXG_HOST_EMULATION   // everything below is device code proper (wave intrinsics, __global__)

// wave-level tally of one tile's res words into the workgroup's LDS slots (FXG_NTALLY): only the reasons the instance can produce
template <int AMAX, int MODE>
__device__ __forceinline__ void fxg_tile_tally(u32 w, bool valid, u64 *tally)
{
    const u32 why = valid ? (w >> 17) & 0xFu : 0u;
    const bool lead = fxg_lane() == 0u;
    auto count = [&](u32 reason) {
        const u64 b = __ballot(why == reason);
        if (lead && b) atomicAdd(&tally[4 + reason], (u64)__builtin_popcountll(b));
    };
    if constexpr (MODE == 0) {
        if constexpr (AMAX != 0) {
            count(FXG_R_CLIP_TOO_SHORT); count(FXG_R_CLIP_ADAPTER_ONLY); count(FXG_R_CLIP_NO_ADAPTER); count(FXG_R_CLIP_ADAPTER_FOUND);
            count(FXG_R_CLIP_N); count(FXG_R_CLIP_K_MODE);
            const u64 ao = __ballot(valid && ((w >> FXG_RES_ADAPTER_ONLY_BIT) & 1u));
            if (lead && ao) atomicAdd(&tally[3], (u64)__builtin_popcountll(ao));
        }
        count(FXG_R_QTRIM); count(FXG_R_QFILTER);
    } else if constexpr (MODE == 1 || MODE == 2 || MODE == 5) {
        count(FXG_R_FTRIM);
    } else if constexpr (MODE == 4) {
        count(FXG_R_ARTIFACT); count(FXG_R_HAS_N);
    }
}

// MODE 0: [clip][qtrim][qfilter] (AMAX = adapter bucket, 0 = no clip); MODE 1: fixed trim; MODE 2: reverse-complement [+ fixed trim];
// MODE 3: fastq_masker; MODE 4: base census (fastx_artifacts_filter, fastq_to_fasta N-discard)
#ifndef FXG_MIN_WAVES
#define FXG_MIN_WAVES 5   // __launch_bounds__ 2nd argument (waves per SIMD) for the streaming instances
#endif
#ifndef FXG_CLIP_WAVES
#define FXG_CLIP_WAVES 4  // the same for the packed clip instances up to 16 adapter columns (S, S-5 and the path summary of every column live in registers)
#endif
#define FXG_NO_TILE 0xFFFFFFFFu
#ifndef FXG_PRIO_WRITEOUT
#define FXG_PRIO_WRITEOUT 3   // s_setprio of the clip instances' write-out and staging phases (0 = none)
#endif
#ifndef FXG_PRIO_STAGING
#define FXG_PRIO_STAGING 2
#endif
// Threads per workgroup of an instance (FXG_CLIP_TBLOCK for the two-pass clip instances).  One wave per workgroup (64-read tiles, no
// workgroup barrier anywhere) was built and measured for them: cfg3 14.2 against 13.9 ms, cfg5 60 against 50 ms (four times the tiles
// to scan, publish and ticket, and 13 single-wave workgroups do not spread evenly over four SIMDs) -- profiles/r03/i_ablate.txt.
// Steps between a tile's decision and its write-out (FxgKArgs.depth, chosen in fxg_plan.h).  The write-out needs the tile's place in the output, i.e. every EARLIER tile decided:
// one step of slack (all the streaming instances need) is not enough where a step is a 50-microsecond DP whose speed depends on
// what the other three waves of the SIMD are doing -- the clip instances waited for the prefix 17 % of their time
// (profiles/r03/c_ablate_clip.txt); two steps behind, the slowest of the ~1000 tiles in flight has a whole extra DP to catch up.
// waves per SIMD the clip instances are compiled for: the packed forms keep 2-3 registers per adapter column (two-pass: 6 of <= 16).
// 64 columns at three waves came out WRONG on the GPU in round 3 (19 of 223 reads): ROCm 7.2's register allocator had put the spill
// stores of the summary loop's live-out values into the loop's exit block ahead of the EXEC restore, where they run for no lane
// (DESIGN.md section 3).  Round 4 took the lane-divergent loops out of the DP (fxg_wave_max), which is what that placement needs.  The budget here is a performance choice; what keeps that miscompile out of the product is the ISA check every
// built library goes through (scripts/check_exec_zero.py, build.py) and the launch-bounds matrix (tests/test_gpu_clip_matrix.py).
#ifndef FXG_CLIP_WAVES_WIDE
#define FXG_CLIP_WAVES_WIDE 2   // 65..99 columns; at three the instance spills its DP row: 45 -> 115 ms (profiles/r04/q_wide_bucket_waves.txt)
#endif
#ifdef FXG_CLIP_WAVES_ALL    // the instance x launch-bounds parity matrix (tests/test_gpu_clip_matrix.py): every packed clip instance at this many waves per SIMD
__host__ __device__ constexpr int fxg_clip_waves(int amax) { return amax > 0 ? 1 : FXG_CLIP_WAVES_ALL; }
#else
// (round 4, every budget being correct now -- profiles/r04/p_clip_waves_by_adapter_len.txt, 10 M reads: 36 columns 11.1 -> 9.7 ms at four waves,
//  64 columns 28.4 -> 24.2 ms (100-base reads) / 41.6 -> 35.2 ms (150) at three waves instead of two)
__host__ __device__ constexpr int fxg_clip_waves(int amax) { return amax > 0 ? 1 : fxg_clip_cols(amax) <= 36 ? FXG_CLIP_WAVES : fxg_clip_cols(amax) <= 64 ? 3 : FXG_CLIP_WAVES_WIDE; }
#endif
template <int AMAX, int MODE> struct FxgTileBlock { static constexpr int threads = (MODE == 0 && AMAX < 0 && AMAX >= -16) ? FXG_CLIP_TBLOCK : FXG_TBLOCK; };
// GL (clip instances, a.clip_global): the DP reads the batch itself (fxg_clip_two_pass<.., GL>) -- a kernel of its own, so that the staged form's code and
// register allocation are exactly what they are without it (both forms in one kernel cost cfg3 1.6 % and cfg5 2.1 %, profiles/r04/ai_ab_pre_gl_vs_head.txt)
template <int AMAX, int MODE, bool GL = false>
__global__ __launch_bounds__((FxgTileBlock<AMAX, MODE>::threads), (AMAX == 0 ? FXG_MIN_WAVES : fxg_clip_waves(AMAX))) void fxg_kernel_tiles(const FxgKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool REV = (MODE == 2 || MODE == 5);      // 5: the reverse complement with dword-aligned window loads (fxg_ld16_dw; picked by the plan, FxgKArgs::rev_dw)
    constexpr u32 TB = (u32)FxgTileBlock<AMAX, MODE>::threads, TW = TB / 64u;
    const u32 T = a.tile_reads, stride = a.stride, tid = threadIdx.x;
    const bool use_q = (MODE == 0 && (a.stages & (FXG_STAGE_QTRIM | FXG_STAGE_QFILTER)) != 0) || MODE == 3;
    const u32 NSLOT = (MODE == 0 && AMAX != 0) ? a.depth : 2u;        // tiles between decision and write-out (fxg_plan.h)
    constexpr bool PTAB = MODE == 0 && fxg_clip_uses_ptab(AMAX);            // the DP takes its pair values from an LDS table (staged form and over-the-batch form alike)
    const FxgLds L = fxg_lds_layout(T, stride, MODE == 4 ? 4u : fxg_bitmap_count(a, use_q, MODE == 0 && AMAX != 0), (MODE == 0 && AMAX != 0) ? (GL ? 0u : a.clip_stride) : 0u, NSLOT, MODE == 0 && AMAX != 0, PTAB ? fxg_ptab_bytes(a.clip_ptab_rows, a.clip_ptab_stride) : 0u);
    u32 m_reads = 0, m_nt = 0, art_bad = 0;      // MODE 3 report counters / MODE 4 alphabet check, folded once at the end
    u32 *bm_g = reinterpret_cast<u32 *>(smem + L.off_bm_g);
    u32 *bm_l = reinterpret_cast<u32 *>(smem + L.off_bm_l);
    uint8_t *sb = smem + L.off_bases;
    u32 *scratch = reinterpret_cast<u32 *>(smem + L.off_scratch);   // [0,2W) scan, then 2 words of tile totals per slot (<= 4 slots), 2 tickets, the broadcast pair
    u32 *s_tot = scratch + 2 * TW, *s_ticket = s_tot + 8;          // two words of tile totals per slot (<= 4 slots), two tickets
    u64 *bc = reinterpret_cast<u64 *>(s_tot + 10);                   // [0,2) broadcast of the resolved bases
    u64 *tally = reinterpret_cast<u64 *>(smem + L.off_tally);       // this workgroup's share of the -v report counters
    if (tid < FXG_NTALLY) tally[tid] = 0ull;
    if constexpr (PTAB) fxg_clip_ptab_build(a, smem + L.off_ptab, tid, TB);      // (visible to every thread after the barrier that follows the first tile's staging)
#ifdef FXG_ABLATION
    u64 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = __builtin_amdgcn_s_memrealtime();   // 100 MHz clocks per phase (wave 0 of the workgroup), summed over its tiles
    const u64 clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#define FXG_TPHASE(i) do { const u64 now_ = __builtin_amdgcn_s_memrealtime(); ph[i] += now_ - pt; pt = now_; } while (0)
#else
#define FXG_TPHASE(i) do { } while (0)
#endif

    // One workgroup turns the tiles' totals into prefixes (fxg_scanner); the others process tiles.
    if (a.compact) {
        if (tid == 0) s_ticket[0] = atomicAdd(a.role, 1u);
        __syncthreads();
        const bool scanner = (s_ticket[0] == 0u);
        __syncthreads();
        if (scanner) { if (tid < 64) fxg_scanner(a); return; }
    }
    // Sharded dispenser: workgroup b draws from counter g = b % groups, which hands out tiles g, g+groups, ...
    // The smallest unfinished tile is always either owned by a running workgroup or the next ticket of its
    // counter (whose earlier tiles are all finished, so a workgroup of that group is about to draw it):
    // progress never depends on residency, dispatch order or placement.
    const u32 G = a.ticket_groups, grp = blockIdx.x % G;
    u32 *my_ticket = a.ticket + grp * FXG_TICKET_STRIDE;
    if (tid == 0) s_ticket[0] = atomicAdd(my_ticket, 1u);
    __syncthreads();
    u32 cur = s_ticket[0] * G + grp;
    u32 pend = FXG_NO_TILE, mid = FXG_NO_TILE, mid2 = FXG_NO_TILE;      // pend: written out in this step; mid / mid2 (three / four slots): decided one / two steps ago, written out later
    u32 slot = 0, tk = 0;
    for (;;) {
        u64 peek = 0;                                    // first look at the prefix of `pend`, in flight during stage A
        if (tid < 64 && pend != FXG_NO_TILE && a.compact) peek = fxg_peek_prefix(a, pend);
        // ------------------------------ stage A: tile `cur` into slot `slot` ------------------------------
        if (cur < a.ntiles) {
            // The next ticket is in flight while this tile is decided.  (Holding a ticket is only harmless because nothing in
            // this step waits for a tile later than `pend`: a one-slot variant that gathered `cur` itself serialised the chip.)
            // Clip instances draw it only AFTER this tile's decision (below): a ticket held through a 50-microsecond DP means its tile is decided a
            // whole step later than the tickets other workgroups drew right after it, and every later tile's place in the output waits for
            // it (cfg3 14.4 -> 13.8 ms, cfg5 -1.5 %, profiles/r04/l_ticket_timing.txt; the streaming instances lose 1 % that way and keep the early draw)
            constexpr bool TICKET_AFTER_DECISION = (MODE == 0 && AMAX != 0);
            if (!TICKET_AFTER_DECISION && tid == 0) s_ticket[tk ^ 1u] = atomicAdd(my_ticket, 1u);
            const u32 r0 = cur * T;
            const u64 left = a.n - (u64)r0;
            const u32 nreads = left < (u64)T ? (u32)left : T;
            const u64 tb = (u64)r0 * stride;
            const u32 tbytes = nreads * stride;
            unsigned char *sl = smem + slot * L.slot_bytes;
            if constexpr (MODE == 0 || MODE == 3 || MODE == 4) {
                if constexpr (MODE == 0 && AMAX != 0) __builtin_amdgcn_s_setprio(FXG_PRIO_STAGING);      // the same for the staging loads of the next tile
                if (use_q && !FXG_DBG(a, 8u)) fxg_phase_bitmaps(a, tb, tbytes, bm_g, bm_l, tid, TB);
                if constexpr (MODE == 0 && AMAX != 0) { if (!GL && (!FXG_DBG(a, 16u) || pend == FXG_NO_TILE)) fxg_phase_stage_bases(a.clip_src, a.clip_total, (u64)r0 * a.clip_stride, nreads * a.clip_stride, sb, tid, TB); }   // 16: the DP on the workgroup's first tile over and over (the DP's own rate)
                if constexpr (MODE == 4) fxg_phase_census(a, tb, tbytes, fxg_census_bitmaps(bm_g, T, stride), tid, TB);
                __syncthreads();
                if constexpr (MODE == 0 && AMAX != 0) __builtin_amdgcn_s_setprio(0);
            }
            FXG_TPHASE(0);
            u32 keep = 0, olen = 0, anchor = tid * stride, word = 0;
            if (tid < nreads) {
                if constexpr (MODE == 0 && AMAX < -16) {
                    float *ck = a.clip_ck ? a.clip_ck + (size_t)blockIdx.x * ((size_t)FXG_CK_SLOTS * (u32)fxg_clip_cols(AMAX) * TB) + tid : nullptr;
                    word = fxg_decide_a<AMAX, GL>(a, bm_g, bm_l, sb, r0, tid, &keep, &olen, ck, TB, PTAB ? smem + L.off_ptab : nullptr);      // (GL only with checkpoint scratch: fxg_plan.h)
                } else if constexpr (MODE == 0 && AMAX < 0) {     // register two-pass instances: the DP over the staged tile, or straight over the batch (fxg_plan.h: clip_global)
                    word = fxg_decide_a<AMAX, GL>(a, bm_g, bm_l, sb, r0, tid, &keep, &olen, nullptr, 0u, PTAB ? smem + L.off_ptab : nullptr);
                } else if constexpr (MODE == 0) word = fxg_decide_a<AMAX>(a, bm_g, bm_l, sb, r0, tid, &keep, &olen);
                else if constexpr (MODE == 3) { u32 nl; word = fxg_decide_mask(a, bm_l, r0, tid, &keep, &olen, &nl); m_nt += nl; m_reads += (nl != 0u); }
                else if constexpr (MODE == 4) word = fxg_decide_census(a, fxg_census_bitmaps(bm_g, T, stride), tid * stride, r0, tid, &keep, &olen, &art_bad);
                else word = fxg_decide_b<REV>(a, r0, tid, &keep, &olen, &anchor);
            }
            // the -v report counters (a12) are functions of res[]: tally the drop reasons this instance can produce, per wave, as the
            // words go by (popcount of a ballot; one LDS add per wave and reason that occurred) instead of a second pass over res[]
#ifdef FXG_ABLATION
            {   // how long the waves of a workgroup wait for its slowest one after the decision (all waves, summed): an extra barrier, timed
                const u64 b0 = __builtin_amdgcn_s_memrealtime();
                __syncthreads();
                if ((tid & 63u) == 0u) atomicAdd(reinterpret_cast<u64 *>(a.errflag + 10) + 5, __builtin_amdgcn_s_memrealtime() - b0);
            }
#endif
            FXG_TPHASE(1);
            if (TICKET_AFTER_DECISION && tid == 0) s_ticket[tk ^ 1u] = atomicAdd(my_ticket, 1u);
            fxg_tile_tally<AMAX, MODE>(word, tid < nreads, tally);
            u32 exc, exb, totc, totb;
            fxg_block_scan2<(int)TW>(keep, keep ? olen : 0u, scratch, &exc, &exb, &totc, &totb);   // one __syncthreads inside
            if (tid == 0) { tally[0] += nreads; tally[1] += totc; tally[2] += totb; }
            if (a.compact) {
                if (tid == 0) fxg_publish_total(a, cur, totc, totb);     // as early as possible: the scanner and every later tile wait for it
                u32 *k_off = reinterpret_cast<u32 *>(sl);
                if (tid < nreads && keep) {                          // kept reads only, indexed by their rank inside the tile
                    k_off[exc] = exb;
                    reinterpret_cast<u32 *>(sl + L.so_ksrc)[exc] = anchor;
                    reinterpret_cast<uint16_t *>(sl + L.so_kidx)[exc] = (uint16_t)tid;
                    if (L.has_tab) fxg_tab_fill(reinterpret_cast<uint16_t *>(sl + L.so_ktab), exc, exb, olen);
                }
                if (tid == 0) { k_off[totc] = totb; s_tot[2 * slot] = totc; s_tot[2 * slot + 1] = totb; }
            }
            FXG_TPHASE(2);
        }
        // ------------------------------ stage B: tile `pend`, decided NSLOT - 1 steps ago ------------------------------
        // (behind stage A: the scanner has had at least a whole stage A to deliver the prefix, so the wait is off the critical path)
        if (pend != FXG_NO_TILE && a.compact) {
            // clip instances: the write-out's few, latency-bound instructions go ahead of the other workgroups' DP rows (a wave that waits here is a wave
            // missing from its SIMD's interleave of four); with the staging below cfg5 -1.5 %, cfg3 -0.8 % (profiles/r06/w_clip_ab_setprio.txt)
            if constexpr (MODE == 0 && AMAX != 0) __builtin_amdgcn_s_setprio(FXG_PRIO_WRITEOUT);
            const u32 ps = slot + 1u == NSLOT ? 0u : slot + 1u;
            const u32 r0 = pend * T;
            const u64 left = a.n - (u64)r0;
            const u32 nreads = left < (u64)T ? (u32)left : T;
            const unsigned char *sl = smem + ps * L.slot_bytes;
            const u32 *k_off = reinterpret_cast<const u32 *>(sl);
            const u32 *k_src = reinterpret_cast<const u32 *>(sl + L.so_ksrc);
            const uint16_t *k_idx = reinterpret_cast<const uint16_t *>(sl + L.so_kidx);
            const uint16_t *k_tab = L.has_tab ? reinterpret_cast<const uint16_t *>(sl + L.so_ktab) : nullptr;
            if (tid < 64 && !FXG_DBG(a, 2u)) fxg_wait_prefix(a, pend, peek, bc);
            __syncthreads();
            FXG_TPHASE(3);
            const u64 base_c = bc[0], base_b = bc[1];
            const u32 nk = s_tot[2 * ps], totb = s_tot[2 * ps + 1];
            const bool placed = base_c != ~0ull;        // false: the wait for the prefix expired (error flag set) -- nothing of this tile is written
            if (placed && tid < nk) fxg_write_kept_meta(a, base_c + tid, k_off[tid + 1] - k_off[tid], r0 + k_idx[tid], base_b + k_off[tid]);
            if (placed && !FXG_DBG(a, 1u)) {
                // the clip instances run four waves per SIMD: their gather keeps four chunks per lane in flight (FXG_GATHER_K)
                constexpr int GK = (MODE == 0 && AMAX != 0) ? FXG_CLIP_GATHER_K : FXG_GATHER_K;
                const u32 bad = fxg_tile_gather<REV, MODE == 3, GK, MODE == 5>(a, k_off, k_src, k_tab, nk, (u64)r0 * stride, nreads * stride, base_b, totb, tid, TB);
                if (REV && bad) atomicOr(a.errflag, FXG_DEV_ERR_BAD_BASE);
            }
            FXG_TPHASE(4);
            if constexpr (MODE == 0 && AMAX != 0) __builtin_amdgcn_s_setprio(0);
        }
        if (cur >= a.ntiles && (NSLOT == 2u || (mid == FXG_NO_TILE && mid2 == FXG_NO_TILE))) break;
        __syncthreads();            // ticket written by thread 0 in stage A; also fences slot reuse NSLOT iterations apart
        const u32 done = cur < a.ntiles ? cur : FXG_NO_TILE;
        if (NSLOT == 4u) { pend = mid; mid = mid2; mid2 = done; } else if (NSLOT == 3u) { pend = mid; mid = done; } else pend = done;
        if (cur < a.ntiles) { tk ^= 1u; cur = s_ticket[tk] * G + grp; }
        slot = slot + 1u == NSLOT ? 0u : slot + 1u;
    }
    __syncthreads();
#ifdef FXG_ABLATION
    if (tid == 0) for (int i = 0; i < 5; ++i) atomicAdd(reinterpret_cast<u64 *>(a.errflag + 10) + i, ph[i]);
    // the shader clock this kernel ran at: s_memtime ticks (shader cycles) against the constant 100 MHz counter, one workgroup's whole life
    if (tid == 0 && blockIdx.x == 5u) {
        reinterpret_cast<u64 *>(a.errflag + 10)[6] = __builtin_amdgcn_s_memtime() - clk_t0;
        reinterpret_cast<u64 *>(a.errflag + 10)[7] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
#endif
    if (tid < FXG_NTALLY && tally[tid]) atomicAdd(&a.tally[tid], tally[tid]);     // one global add per workgroup and non-zero slot
    if constexpr (MODE == 3) {                       // masked reads / nucleotides: wave sums, one atomic pair per wave, once
        for (int d = 32; d >= 1; d >>= 1) { m_reads += __shfl_xor(m_reads, d, 64); m_nt += __shfl_xor(m_nt, d, 64); }
        if (fxg_lane() == 0 && (m_reads | m_nt)) { atomicAdd(&a.extra[0], (u64)m_reads); atomicAdd(&a.extra[1], (u64)m_nt); }
    }
    if constexpr (MODE == 4) { if (art_bad) atomicOr(a.errflag, FXG_DEV_ERR_BAD_BASE); }
}

#if !defined(FXG_CLIP_TU)      // (the translation units of the clip instances have no kernels of their own besides fxg_kernel_tiles: fxg_host.h)
// tally slots of all workgroups -> counters[FXG_NCOUNTERS]; also folds the device error word and the masker sums in
__global__ void fxg_kernel_finish_counters(const u64 *tally, u32 stages, const u32 *errflag, const u64 *extra, u64 *counters)
{
    if (threadIdx.x != 0) return;
    FxgCounts c = {};
    c.in = tally[0]; c.kept = tally[1]; c.bases = tally[2]; c.adapter_only = tally[3];
    c.too_short = tally[4 + FXG_R_CLIP_TOO_SHORT]; c.no_adapter = tally[4 + FXG_R_CLIP_NO_ADAPTER]; c.adapter_found = tally[4 + FXG_R_CLIP_ADAPTER_FOUND];
    c.has_n = tally[4 + FXG_R_CLIP_N]; c.qtrim = tally[4 + FXG_R_QTRIM]; c.qfilter = tally[4 + FXG_R_QFILTER]; c.ftrim = tally[4 + FXG_R_FTRIM];
    c.k_mode = tally[4 + FXG_R_CLIP_K_MODE]; c.artifact = tally[4 + FXG_R_ARTIFACT];
    u64 slot[FXG_NCOUNTERS];
    fxg_counts_to_slots(c, stages, slot);
    slot[FXG_C_ERRORS] = (u64)*errflag;
    slot[FXG_C_MASKED_READS] = extra[0];
    slot[FXG_C_MASKED_NT] = extra[1];
    for (int i = 0; i < FXG_NCOUNTERS; ++i) counters[i] = slot[i];
}

// ------------------------------------------------------------------------------------------------
// deterministic synthetic reads; rows are built in LDS and written with
// coalesced 16-byte stores.  One workgroup = 64 consecutive reads.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 fxg_splitmix(u64 x)
{
    x += 0x9E3779B97F4A7C15ull;
    u64 z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

#define FXG_SYNTH_TILE 64
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_synth(u64 seed, u64 first, u64 n, u32 L, int with_adapter,
                                                              uint8_t *bases, uint8_t *qual, u32 stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 tbytes_full = FXG_SYNTH_TILE * stride;
    uint8_t *sb = smem, *sq = smem + fxg_r16(tbytes_full);
    const u64 r0 = (u64)blockIdx.x * FXG_SYNTH_TILE;
    const u64 left = n - r0;
    const u32 nreads = left < FXG_SYNTH_TILE ? (u32)left : FXG_SYNTH_TILE;
    const u32 tbytes = nreads * stride;
    const char acgt[4] = {'A', 'C', 'G', 'T'};
    for (u32 i = threadIdx.x; i < tbytes; i += FXG_BLOCK) {
        const u32 r = i / stride, p = i - r * stride;
        uint8_t b = 0, q = 0;
        if (p < L) {
            const u64 key = fxg_splitmix(seed ^ ((first + r0 + r) * 0x9E3779B97F4A7C15ull));
            const u64 u = fxg_splitmix(key + p);
            b = (uint8_t)acgt[u & 3];
            if ((u >> 8) % 200 == 0) b = 'N';
            if (qual) {
                const u64 d = fxg_splitmix(key + (1ull << 32)) % (L + L / 3);
                const bool noisy = (fxg_splitmix(key + (1ull << 32) + 1) % 4) == 0;
                const u64 v = fxg_splitmix(key + (2ull << 32) + p);
                const u32 lo = 2 + (u32)(v % 18), hi = 25 + (u32)(v % 16);
                const bool dip = ((v >> 16) % (noisy ? 4 : 16)) == 0;
                q = (uint8_t)(33 + ((p < d) ? (dip ? lo : hi) : lo));
            }
        }
        sb[i] = b;
        if (qual) sq[i] = q;
    }
    __syncthreads();
    if (with_adapter && threadIdx.x < nreads) {
        const u32 r = threadIdx.x;
        const u64 key = fxg_splitmix(seed ^ ((first + r0 + r) * 0x9E3779B97F4A7C15ull));
        const u64 av = fxg_splitmix(key + (3ull << 32));
        if (av % 2 == 0) {
            const char ad[13] = {'A', 'G', 'A', 'T', 'C', 'G', 'G', 'A', 'A', 'G', 'A', 'G', 'C'};
            const u32 pos = (u32)((av >> 8) % (L + 1));
            const bool sub = ((av >> 40) % 8) == 0;
            const u32 sidx = (u32)((av >> 44) % 13);
            const char sch = acgt[(av >> 48) & 3];
            for (u32 k = 0; k < 13 && pos + k < L; ++k) sb[r * stride + pos + k] = (uint8_t)((sub && k == sidx) ? sch : ad[k]);
        }
    }
    __syncthreads();
    const u64 gb = r0 * stride;
    for (u32 i = threadIdx.x * 16; i < tbytes; i += FXG_BLOCK * 16) {
        if (i + 16 <= tbytes) {
            *reinterpret_cast<u32x4 *>(bases + gb + i) = *reinterpret_cast<const u32x4 *>(sb + i);
            if (qual) *reinterpret_cast<u32x4 *>(qual + gb + i) = *reinterpret_cast<const u32x4 *>(sq + i);
        } else {
            for (u32 k = i; k < tbytes; ++k) { bases[gb + k] = sb[k]; if (qual) qual[gb + k] = sq[k]; }
        }
    }
}
#endif      // !FXG_CLIP_TU
#endif  // FXG_HOST_EMULATION
