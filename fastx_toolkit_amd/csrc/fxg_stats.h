// fxg_stats.h -- fastx_quality_stats as a device reduction.
//
// Reference: src/fastx_quality_stats/fastx_quality_stats.c:166-216 (read_file): for every base of every read, per column
// ("cycle") and per nucleotide class, a count, min/max/sum of the quality value and a counting-sort histogram of it.  All
// of that is a function of one histogram  hist[column][class A,C,G,T,N][quality byte]  -- count, sum, min, max and the
// order statistics of print_statistics (:218-340) follow from it on the host -- so the device only builds that.
//
// Work decomposition: every row is read from HBM once.  A workgroup owns a contiguous slice of the reads and the block
// histogram of up to 160 columns -- 10 strips of 16 columns x 6 rows (A C G T N + a spare row for padding) x a 64-value
// window of the quality byte, 16-bit counters, the two columns of a pair in one word: 120 KB of LDS, updated with ds_add.
// A work item is one 16-byte piece (strip) of one row; consecutive lanes take consecutive pieces, so a wave reads 1 KB of
// contiguous rows per load.  The workgroup is 960 threads wide -- a multiple of the 10 strips -- so a lane keeps ITS strip
// for the whole kernel: the strip's LDS base, its swizzle and (fixed-length batches) the mask of its bytes past the end
// of the read are loop invariants, and the read index advances by a constant.
// Per-base work on the fast path (16 valid bases, qualities inside the window) is SWAR on dwords: one v_perm_b32 looks the
// class of four bases up in an 8-entry table indexed by the low three bits of the letter (1 3 7 4 6 for A C G T N), a second
// one yields the letter those bits should belong to (the validity test), qualities are rebased, scaled to byte offsets and
// XOR-swizzled four at a time, and a third perm pairs class and offset into the 16-bit LDS address of each base
// (row = class * 256 B), leaving one address add and one ds_add per base.
// The swizzle (offset ^= 16 * class ^ 12 * strip, within the 256-byte row) makes the LDS bank depend on class and strip
// as well as on the quality: reads of one wave share most of their quality values.
// Before a counter could wrap (65 535 reads) the workgroup adds the block to ITS OWN u32 partial in HBM (no atomics) and
// clears it.  Quality bytes outside the window (Phred+33 codes 33..96 are inside) go straight to the result with a global
// atomic.  A second kernel folds the partials into the caller's u64 histogram.  Reads longer than 160 take one pass per
// column block.  HBM-bound by design (2 bytes per base in, nothing out).
// Dense fixed-length batches of even length 16 .. 160 (stride == length: what the text path packs and what BASELINE's batches are) run in the PIECE form
// instead (further down): the rows are one byte stream, a lane takes an aligned 16-byte piece of it, nothing is masked and no padding byte is added.
#pragma once
#include "fxg_device.h"

#define FXG_QS_STRIP 16u
#define FXG_QS_WAVES 10u                                   // strips per column block (the name is historic: not tied to waves)
#define FXG_QS_TBLOCK 960u                                 // one workgroup per CU (LDS); a multiple of the strips per block AND of the wave size
#define FXG_QS_BLOCK_COLS (FXG_QS_WAVES * FXG_QS_STRIP)    // 160 columns per pass
#define FXG_QS_WBASE 33u                                   // first quality byte of the LDS window
#define FXG_QS_WBINS 64u
#define FXG_QS_PART_WORDS (FXG_QS_BLOCK_COLS * FXG_QS_CLASSES * FXG_QS_WBINS)   // one workgroup's partial (u32) = 51 200 counters
#define FXG_QS_LROWS 6u                                    // LDS rows per column pair: A C G T N + the spare row (bytes past the end of a read; never flushed)
#define FXG_QS_LROW_WORDS FXG_QS_WBINS                     // 256 bytes: class k of a pair lives at byte k << 8 of the pair's block
#define FXG_QS_LDS_WORDS ((FXG_QS_BLOCK_COLS / 2u) * FXG_QS_LROWS * FXG_QS_LROW_WORDS)   // word = two u16 counters: even column low, odd column high
#ifndef FXG_QS_DEPTH
#define FXG_QS_DEPTH 3u                                    // trips a lane's row loads run ahead of its LDS adds (round-robin loop)
#endif
#ifndef FXG_QS_UNROLL
#define FXG_QS_UNROLL 1u                                   // rows per lane and trip of the tested loop
#endif

struct FxgStatsArgs {
    const uint8_t  *bases, *qual;     // qual may be null (FASTA): bin 0 counts
    const uint16_t *len;
    u64  n, total_bytes;
    u32  fixed_len, stride;
    u32  strip0;                      // first strip of this pass (column block)
    u32  nwg;                         // workgroups = slices of the reads
    u32 *partial;                     // [nwg][FXG_QS_PART_WORDS]
    u32  round_robin;                 // fixed-length batches with qualities: 1 = trips dealt round robin, piece form where the batch allows it (the default); measurement knobs:
                                      // 0 static slices through the tested loop, 2 contiguous runs of trips through the dealt loop, 3 round robin in the row-strip form everywhere
    u64 *hist;                        // [hist_cols][FXG_QS_CLASSES][FXG_QS_BINS]
    u32  hist_cols;
};

#ifdef FXG_HOST_EMULATION
#define FXG_LDS_ADD(p, v) ((void)(*(p) += (v)))
#define FXG_GLOBAL_INC64(p) ((void)(++*(p)))
#define FXG_GLOBAL_ADD32(p, v) ((void)(*(p) += (v)))
#else
#define FXG_LDS_ADD(p, v) ((void)atomicAdd((p), (v)))
#define FXG_GLOBAL_INC64(p) ((void)atomicAdd((p), 1ull))
#define FXG_GLOBAL_ADD32(p, v) ((void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))   // result unused: global_atomic_add_u32 without return
#endif

// class of a base: A C G T N -> 0..4 (either case, fastx_quality_stats.c:142-155), anything else -> 5 (not counted)
FXG_HD u32 fxg_stats_class(u32 c)
{
    const u32 u = c & 0xDFu;
    const u32 i = (u >> 1) & 3u;                            // A 0, C 1, T 2, G 3 (N collides with G)
    const u32 acgt = (i == 2u) ? 3u : (i == 3u ? 2u : i);
    const bool ok = (u == 0x41u) | (u == 0x43u) | (u == 0x47u) | (u == 0x54u);
    return ok ? acgt : (u == 0x4Eu ? 4u : 5u);
}

// One read, one strip.  Split in two so that the kernel can have the rows of several reads in flight before it touches one.
struct FxgStripRow { u32x4 vb, vq; u32 nb; };

FXG_HD void fxg_stats_load(const FxgStatsArgs &a, u64 r, u32 strip, FxgStripRow &o)
{
    const u32 L = a.len ? (u32)a.len[r] : a.fixed_len;
    const u32 c0 = strip * FXG_QS_STRIP;
    o.nb = 0u; o.vq = (u32x4){0u, 0u, 0u, 0u}; o.vb = o.vq;
    if (c0 >= L) return;
    o.nb = L - c0 < FXG_QS_STRIP ? L - c0 : FXG_QS_STRIP;
    const u64 at = r * a.stride + c0;
    if (at + 16u <= a.total_bytes) { o.vb = fxg_ld16(a.bases + at); if (a.qual) o.vq = fxg_ld16(a.qual + at); }
    else { o.vb = fxg_window(a.bases, (long long)at, a.total_bytes, 0, (int)o.nb); if (a.qual) o.vq = fxg_window(a.qual, (long long)at, a.total_bytes, 0, (int)o.nb); }
}

// LDS layout.  Byte offset of the counter word of (strip s of the block, column j of the strip, class k, window bin w):
//   block of the column pair  (s * 8 + j / 2) * 6 * 256
//   row of the class          k * 256
//   swizzled bin              (4 * w) ^ fxg_stats_swz(s, k)
// the counter is the word's low half for even j, its high half for odd j.
FXG_HD u32 fxg_stats_swz(u32 s, u32 k) { return ((k << 4) ^ (s * 12u)) & 0xFCu; }
FXG_HD u32 fxg_stats_pair_base(u32 s, u32 j) { return (s * (FXG_QS_STRIP / 2u) + (j >> 1)) * (FXG_QS_LROWS * FXG_QS_LROW_WORDS * 4u); }
FXG_HD u32 fxg_stats_byte(u32 s, u32 j, u32 k, u32 w) { return fxg_stats_pair_base(s, j) + (k << 8) + ((w << 2) ^ fxg_stats_swz(s, k)); }

// masks of the bytes of a 16-byte strip piece that lie inside a read with nb bytes in this strip
FXG_HD void fxg_stats_masks(u32 nb, u32 (&m)[4])
{
#pragma unroll
    for (u32 d = 0; d < 4u; ++d) {
        const int keep = (int)nb - (int)(4u * d);
        m[d] = fxg_lowbytes32(keep < 0 ? 0 : (keep > 4 ? 4 : keep));
    }
}

// table of the fast path, indexed by the low three bits of a letter: '@' 0, A 1, C 3, T 4, N 6, G 7 (2 and 5 belong to no base)
//   expected upper-case letter:  40 41 00 43 54 00 4E 47        class row:  5 0 - 1 3 - 4 2
#define FXG_QS_EXP_LO 0x43004140u
#define FXG_QS_EXP_HI 0x474E0054u
#define FXG_QS_ROW_LO 0x01070005u
#define FXG_QS_ROW_HI 0x02040703u

// sl: strip of the block (LDS position); m: fxg_stats_masks(o.nb)
FXG_HD void fxg_stats_accumulate(const FxgStatsArgs &a, const FxgStripRow &o, u32 sl, u32 col0, const u32 (&m)[4], u32 *lds)
{
#ifdef FXG_QS_NOACC        // timing experiment (wrong counts): the kernel's loads alone -- one LDS add per row keeps them alive
    FXG_LDS_ADD(lds + (threadIdx.x & 63u), o.vb.x ^ o.vb.y ^ o.vb.z ^ o.vb.w ^ o.vq.x ^ o.vq.y ^ o.vq.z ^ o.vq.w);
    return;
#endif
    if (o.nb == 0u) return;
    u32 wb[4] = {o.vb.x, o.vb.y, o.vb.z, o.vb.w}, wq[4] = {o.vq.x, o.vq.y, o.vq.z, o.vq.w};
    // Fast path: 16 bases A C G T N (either case) whose quality bytes all lie in the window -- no per-base test, no branch.
    // Bytes past the end of the read (the last strip of a row, ragged reads) are first replaced by '@' (counted in the spare
    // row) with quality WBASE, so that short strips take the fast path as well: every wave holds some, and a wave that has
    // ONE lane on the slow path executes the slow path.
    u32 bad = 0u, k4[4], o4[4];
    const u32 sw = (sl * 12u) & 0xFCu;
#pragma unroll
    for (u32 d = 0; d < 4u; ++d) {
        const u32 b = (wb[d] & m[d]) | (0x40404040u & ~m[d]);
        const u32 q = (wq[d] & m[d]) | ((FXG_QS_WBASE * 0x01010101u) & ~m[d]);
        const u32 sel = b & 0x07070707u;
        bad |= fxg_perm(FXG_QS_EXP_HI, FXG_QS_EXP_LO, sel) ^ (b & 0xDFDFDFDFu);          // not one of @ A C G T N a c g t n
        k4[d] = fxg_perm(FXG_QS_ROW_HI, FXG_QS_ROW_LO, sel);
        const u32 w4 = q - FXG_QS_WBASE * 0x01010101u;                                   // a byte below the window borrows: its own result is then >= 0xDF
        bad |= w4 & 0xC0C0C0C0u;                                                         // below / above the window
        o4[d] = (w4 << 2) ^ (k4[d] << 4) ^ (sw * 0x01010101u);                           // fxg_stats_byte's swizzled bin offsets (k <= 5, so k << 4 stays inside its byte)
    }
    if (!bad) {
        unsigned char *base = reinterpret_cast<unsigned char *>(lds) + fxg_stats_pair_base(sl, 0u);
#pragma unroll
        for (u32 j = 0; j < FXG_QS_STRIP; j += 2u) {
            // halves of h: class << 8 | offset of base j (low) and of base j + 1 (high) = byte offset inside the pair's block
            const u32 h = fxg_perm(k4[j >> 2], o4[j >> 2], (j & 2u) ? 0x07030602u : 0x05010400u);
            unsigned char *pb = base + (j >> 1) * (FXG_QS_LROWS * FXG_QS_LROW_WORDS * 4u);
#ifdef FXG_QS_FAKE_BANKS      // timing experiment (wrong counts): every lane of a wave on its own bank -- what the kernel would take without LDS bank conflicts
            FXG_LDS_ADD(reinterpret_cast<u32 *>(pb + ((h & 0xFF00u) | ((threadIdx.x & 63u) << 2))), 1u);
            FXG_LDS_ADD(reinterpret_cast<u32 *>(pb + (((h >> 16) & 0xFF00u) | ((threadIdx.x & 63u) << 2))), 0x10000u);
#else
            FXG_LDS_ADD(reinterpret_cast<u32 *>(pb + (h & 0xFFFFu)), 1u);
            FXG_LDS_ADD(reinterpret_cast<u32 *>(pb + (h >> 16)), 0x10000u);
#endif
        }
        return;
    }
#pragma unroll 1
    for (u32 j = 0; j < o.nb; ++j) {                                                     // odd bytes, rare qualities: one base at a time
        const u32 b = (wb[j >> 2] >> (8u * (j & 3u))) & 0xFFu, q = (wq[j >> 2] >> (8u * (j & 3u))) & 0xFFu;
        const u32 k = fxg_stats_class(b);
        if (k >= FXG_QS_CLASSES || q >= FXG_QS_BINS) continue;
        const u32 w = q - FXG_QS_WBASE;
        if (w < FXG_QS_WBINS) FXG_LDS_ADD(reinterpret_cast<u32 *>(reinterpret_cast<unsigned char *>(lds) + fxg_stats_byte(sl, j, k, w)), (j & 1u) ? 0x10000u : 1u);
        else if (col0 + j < a.hist_cols) FXG_GLOBAL_INC64(&a.hist[((u64)(col0 + j) * FXG_QS_CLASSES + k) * FXG_QS_BINS + q]);
    }
}

// ---- the piece form (round 6): dense fixed-length batches of even length 16 .. 160 ----
// Rows that lie back to back (stride == length) are one byte stream, and a run of R whole reads whose bytes are a multiple of 16 is a run of ALIGNED 16-byte
// pieces with no byte to mask: lane p of the workgroup takes piece p of every trip.  Because a trip is whole reads, the column of the lane's first byte,
// o = 16 p mod L, is the same in every trip; the length is even, so o is even, a piece's byte pairs are the histogram's column pairs, and the one place where
// a piece runs from the end of a read into the next one (column L - 2 -> 0) lies between two pairs.  The LDS position of the lane's eight pairs and their
// swizzles are loop invariants (8 + 4 registers).  Against the row-strip form on 150-byte rows: no 16-byte load straddles a 16-byte boundary, no lane adds
// the ten padding bytes of a row's last strip (160 adds per 150 bases), 900 of 960 lanes carry 16 bases each -- rows of 160 bytes, where the row-strip form
// has all of that by itself, ran 8 % faster than rows of 150 (profiles/r06/stats_by_row_length.txt).
// R = m * r0 reads = m * p0 pieces per trip, r0 = 16 / gcd(L, 16) the shortest run of whole reads that is whole pieces, m as large as the workgroup allows.
FXG_HD bool fxg_stats_piece_plan(const FxgStatsArgs &a, u32 *reads, u32 *pieces)
{
    const u32 L = a.fixed_len;
    if (a.len || !a.qual || L != a.stride || L < FXG_QS_STRIP || L > FXG_QS_BLOCK_COLS || a.strip0 != 0u) return false;
    const u32 low = L & (0u - L), g = low < 16u ? low : 16u;             // gcd(L, 16)
    const u32 r0 = 16u / g, p0 = r0 * L / 16u;                           // p0 <= 159 (odd lengths: 16 reads are L pieces)
    // a trip whose bytes are whole 128-byte lines keeps every trip on the line grid by itself: taken when it costs at most 32 lanes against the widest trip
    // that leaves room for the seven pieces a trip can grow by when its cuts are moved onto the grid (fxg_stats_piece_cut)
    const u32 m_sh = (FXG_QS_TBLOCK - 7u) / p0;
    u32 m_nat = FXG_QS_TBLOCK / p0;
    while (m_nat && (m_nat * p0) % 8u) --m_nat;
    const u32 m = (m_nat * p0 + 32u >= m_sh * p0) ? m_nat : m_sh;
    *reads = m * r0; *pieces = m * p0;
    return true;
}
// The cuts between trips sit on the 128-byte line grid: trip T is the bytes [cut(T), cut(T + 1)), cut(T) = T * RL moved DOWN to a line (the first cut is 0, the
// last one the end of the last whole trip), so that every wave's load of 64 pieces is eight whole lines.  On 150-byte rows (RL = 14 400 = 112.5 lines) the trips
// are 896 and 904 pieces by turns.  Measured before this was built, the piece form with cuts at T * RL: 2.55 ms where RL is whole lines (rows of 160), 2.65 at half
// a line (150), 2.74-2.77 at a quarter or an eighth (144, 126, 100, 76, 36) for the same 15 GB (profiles/r06/stats_piece_vs_rows_by_length_first_form.txt).
// A workgroup's trips are G apart; when G * RL is whole lines (G a multiple of 8: every launch that fills the chip) its distance to the grid, and with it the column
// of every lane's first byte, is the same in all its trips.  Otherwise (`aligned` false) the cuts stay at T * RL.
FXG_HD u64 fxg_stats_piece_cut(u64 T, u64 ntrip, u64 RL, bool aligned) { const u64 s = T * RL; return (aligned && T < ntrip) ? (s & ~127ull) : s; }
// ODD (odd lengths: 51, 75, 101, 151 ...): the column of a lane's first byte can be odd and its parity turns over where a piece runs into the next read, so the
// LDS block and the counter half (low for even columns, high for odd ones) are kept per BYTE instead of per byte pair: 16 + 16 registers instead of 8.
template <bool ODD> struct FxgPieceLane { u32 o; u32 pb[ODD ? 16 : 8]; u32 val[ODD ? 16 : 1]; u32 sw4[4]; };   // column of the piece's first byte; byte offset of each pair's (byte's) LDS block; the counter halves; the strip swizzles, one per byte
template <bool ODD>
FXG_HD void fxg_stats_piece_lane(u32 L, u32 o, FxgPieceLane<ODD> &c)     // o: column of the piece's first byte (< L; even unless ODD)
{
    c.o = o;
#pragma unroll
    for (u32 d = 0; d < 4u; ++d) c.sw4[d] = 0u;
    if constexpr (ODD) {
#pragma unroll
        for (u32 i = 0; i < 16u; ++i) {
            u32 col = c.o + i;
            if (col >= L) col -= L;
            c.pb[i] = fxg_stats_pair_base(col >> 4, col & 15u);
            c.val[i] = (col & 1u) ? 0x10000u : 1u;
            c.sw4[i >> 2] |= (((col >> 4) * 12u) & 0xFCu) << (8u * (i & 3u));
        }
    } else {
        c.val[0] = 0u;
#pragma unroll
        for (u32 jj = 0; jj < 8u; ++jj) {
            u32 col = c.o + 2u * jj;
            if (col >= L) col -= L;
            c.pb[jj] = fxg_stats_pair_base(col >> 4, col & 15u);
            c.sw4[jj >> 1] |= ((((col >> 4) * 12u) & 0xFCu) * 0x0101u) << (16u * (jj & 1u));
        }
    }
}
template <bool ODD>
FXG_HD void fxg_stats_accumulate_piece(const FxgStatsArgs &a, const FxgStripRow &r, const FxgPieceLane<ODD> &c, u32 *lds)
{
#ifdef FXG_QS_NOACC
    FXG_LDS_ADD(lds + (threadIdx.x & 63u), r.vb.x ^ r.vb.y ^ r.vb.z ^ r.vb.w ^ r.vq.x ^ r.vq.y ^ r.vq.z ^ r.vq.w);
    return;
#endif
    const u32 wb[4] = {r.vb.x, r.vb.y, r.vb.z, r.vb.w}, wq[4] = {r.vq.x, r.vq.y, r.vq.z, r.vq.w};
    u32 bad = 0u, k4[4], o4[4];
#pragma unroll
    for (u32 d = 0; d < 4u; ++d) {                                                       // as fxg_stats_accumulate's fast path; nothing to mask
        const u32 sel = wb[d] & 0x07070707u;
        bad |= fxg_perm(FXG_QS_EXP_HI, FXG_QS_EXP_LO, sel) ^ (wb[d] & 0xDFDFDFDFu);
        k4[d] = fxg_perm(FXG_QS_ROW_HI, FXG_QS_ROW_LO, sel);
        const u32 w4 = wq[d] - FXG_QS_WBASE * 0x01010101u;
        bad |= w4 & 0xC0C0C0C0u;
        o4[d] = (w4 << 2) ^ (k4[d] << 4) ^ c.sw4[d];
    }
    unsigned char *base = reinterpret_cast<unsigned char *>(lds);
    if (!bad) {
#pragma unroll
        for (u32 jj = 0; jj < 8u; ++jj) {
            const u32 h = fxg_perm(k4[jj >> 1], o4[jj >> 1], (jj & 1u) ? 0x07030602u : 0x05010400u);
            if constexpr (ODD) {
                FXG_LDS_ADD(reinterpret_cast<u32 *>(base + c.pb[2u * jj] + (h & 0xFFFFu)), c.val[2u * jj]);
                FXG_LDS_ADD(reinterpret_cast<u32 *>(base + c.pb[2u * jj + 1u] + (h >> 16)), c.val[2u * jj + 1u]);
            } else {
                unsigned char *pb = base + c.pb[jj];
                FXG_LDS_ADD(reinterpret_cast<u32 *>(pb + (h & 0xFFFFu)), 1u);
                FXG_LDS_ADD(reinterpret_cast<u32 *>(pb + (h >> 16)), 0x10000u);
            }
        }
        return;
    }
    const u32 L = a.fixed_len;
#pragma unroll 1
    for (u32 j = 0; j < FXG_QS_STRIP; ++j) {                                             // odd bytes, rare qualities: one base at a time
        const u32 b = (wb[j >> 2] >> (8u * (j & 3u))) & 0xFFu, q = (wq[j >> 2] >> (8u * (j & 3u))) & 0xFFu;
        const u32 col = c.o + j >= L ? c.o + j - L : c.o + j;
        const u32 k = fxg_stats_class(b);
        if (k >= FXG_QS_CLASSES || q >= FXG_QS_BINS) continue;
        const u32 w = q - FXG_QS_WBASE;
        if (w < FXG_QS_WBINS) FXG_LDS_ADD(reinterpret_cast<u32 *>(base + fxg_stats_byte(col >> 4, col & 15u, k, w)), (col & 1u) ? 0x10000u : 1u);
        else if (col < a.hist_cols) FXG_GLOBAL_INC64(&a.hist[((u64)col * FXG_QS_CLASSES + k) * FXG_QS_BINS + q]);
    }
}

// slice of the reads that workgroup g owns
FXG_HD void fxg_stats_slice(const FxgStatsArgs &a, u32 g, u64 *lo, u64 *hi)
{
    const u64 per = (a.n + a.nwg - 1) / a.nwg;
    *lo = (u64)g * per < a.n ? (u64)g * per : a.n;
    *hi = *lo + per < a.n ? *lo + per : a.n;
}

// thread `t` of `nt`: move the LDS block into the workgroup's partial ([column-in-block][class][window bin], u32) and clear it.
// A word of the block always belongs to the same thread (i = t mod nt), and so do the two counters of the partial it feeds, so the
// workgroup's FIRST flush writes them with plain stores, zeros included (the partial needs no clearing pass and no load), and every later
// one adds the non-zero ones with fire-and-forget atomics: no thread ever waits for a value to come back from HBM.  (Round 6: the
// flush used to be `p[0] += v` -- a dependent load, add and store per counter, 32 trips of it per thread behind a workgroup barrier,
// three or four times per kernel, with every workgroup of the chip at the barrier at the same time.)
FXG_HD void fxg_stats_flush(u32 *lds, u32 *part, u32 t, u32 nt, bool first)
{
#ifdef FXG_QS_NOFLUSH     // timing experiment (wrong counts): what the kernel takes without the flushes' traffic
    if (!first) return;
#endif
    for (u32 i = t; i < FXG_QS_LDS_WORDS; i += nt) {
        const u32 x = i % FXG_QS_LROW_WORDS, pk = i / FXG_QS_LROW_WORDS, k = pk % FXG_QS_LROWS, pair = pk / FXG_QS_LROWS;
        const u32 v = lds[i];
        if (v != 0u) lds[i] = 0u;
        if (k >= FXG_QS_CLASSES) continue;                                   // the spare row
        if (v == 0u && !first) continue;
        const u32 w = x ^ (fxg_stats_swz(pair / (FXG_QS_STRIP / 2u), k) >> 2);   // undo the swizzle
        u32 *p = part + ((2u * pair) * FXG_QS_CLASSES + k) * FXG_QS_WBINS + w;
        if (first) { p[0] = v & 0xFFFFu; p[FXG_QS_CLASSES * FXG_QS_WBINS] = v >> 16; }
        else {
            if (v & 0xFFFFu) FXG_GLOBAL_ADD32(p, v & 0xFFFFu);
            if (v >> 16) FXG_GLOBAL_ADD32(p + FXG_QS_CLASSES * FXG_QS_WBINS, v >> 16);
        }
    }
}

// element e of the fold: counter (column-in-block, class, window bin) summed over the workgroups' partials
FXG_HD void fxg_stats_fold_put(const FxgStatsArgs &a, u32 e, u64 sum)
{
    const u32 w = e % FXG_QS_WBINS, ck = e / FXG_QS_WBINS;
    const u32 col = a.strip0 * FXG_QS_STRIP + ck / FXG_QS_CLASSES, k = ck % FXG_QS_CLASSES;
    if (col >= a.hist_cols) return;
    if (sum) a.hist[((u64)col * FXG_QS_CLASSES + k) * FXG_QS_BINS + FXG_QS_WBASE + w] += sum;
}
FXG_HD void fxg_stats_fold(const FxgStatsArgs &a, u32 e)
{
    u64 sum = 0;
    for (u32 g = 0; g < a.nwg; ++g) sum += a.partial[(u64)g * FXG_QS_PART_WORDS + e];
    fxg_stats_fold_put(a, e, sum);
}

// work item g of a slice that starts at read lo: read lo + g / 10, strip g % 10 of the block
FXG_HD void fxg_stats_item(u64 lo, u64 g, u64 *r, u32 *sl) { *r = lo + g / FXG_QS_WAVES; *sl = (u32)(g % FXG_QS_WAVES); }

#ifndef FXG_HOST_EMULATION
// Cache policy of the loop's row loads.  ROW-STRIP form: the DEFAULT policy.  With the non-temporal policy it is 1.0-1.7 % faster (2.667 against 2.713 ms, mean of
// eight alternating runs, profiles/r06/stats_nt_loads.txt; 2.724 against 2.751, stats_variants_one_call.txt) but fetches 7 % more: the 16-byte pieces of 150-byte
// rows share their first and last 128-byte lines with the neighbouring wave's, and a line marked non-temporal is gone before the neighbour asks (FETCH_SIZE
// 1.05 -> 1.12 x the rows, profiles/r06_pmc_stats).  Bytes over the fabric are the scarcer thing; -DFXG_QS_NTL builds the other arm.
// PIECE form: NON-TEMPORAL.  Its waves load whole lines that no other wave touches, so the policy costs no byte (FETCH_SIZE 1.024 x the rows either way) and the
// kernel takes 2.42-2.44 instead of 2.52-2.54 ms, four alternating runs (profiles/r06/stats_piece_nt_depth.txt, stats_piece_nt_traffic.txt; loads 2 or 4 trips
// ahead instead of 3: no difference, same file); -DFXG_QS_PIECE_DEFAULT_LOADS builds the other arm.
struct FxgLdDefault {};
struct FxgLdStream {};
__device__ __forceinline__ u32x4 fxg_qs_ld(const uint8_t *p, FxgLdStream) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4_unaligned *>(p)); }
#ifdef FXG_QS_NTL
__device__ __forceinline__ u32x4 fxg_qs_ld(const uint8_t *p, FxgLdDefault) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4_unaligned *>(p)); }
#else
__device__ __forceinline__ u32x4 fxg_qs_ld(const uint8_t *p, FxgLdDefault) { return fxg_ld16(p); }
#endif
#ifdef FXG_QS_PIECE_DEFAULT_LOADS
typedef FxgLdDefault FxgLdPiece;
#else
typedef FxgLdStream FxgLdPiece;
#endif
// Fixed-length batches with qualities: trip T of the launch is reads [96 T, 96 T + 96) and belongs to workgroup T mod G, so that at any moment the whole chip
// reads ONE narrow window of each array (256 x 14 KB) -- the order the memory system serves best: a read-only stream of 15 GB takes 2.33 ms that way
// (2.20 with the non-temporal policy) against 2.45-2.50 ms from one far-apart slice per workgroup (scripts/ubench/read_stream.hip, profiles/r06/
// x_read_stream_by_access_form.txt; this kernel's loads alone 2.49 -> 2.33 ms).  Tickets for chunks of 2 .. 32 trips were built and measured too
// (profiles/r06/stats_order.txt): all workgroups then end within 2 % of one another (static slices: 2.50 .. 2.70 ms, profiles/r06/stats_wg_clocks_*.txt)
// and the kernel takes the same time or longer -- the memory system is the limit, not the slowest workgroup -- so the dispenser went again.
// The rows of a lane's trip k + D are requested before trip k goes into the histogram (FXG_QS_DEPTH; 1 -> 4: -1.5 %).  Only trips whose every read may
// be loaded 16 bytes at a time from any column are dealt (all but the batch's last 1..96 reads); workgroup 0 counts the rest through the tested loop.
// ODDK: the kernel for dense batches of ODD length (the piece form with block and counter half per byte: 24 registers more, an allocation of its own)
template <bool ODDK>
__device__ __forceinline__ void fxg_quality_stats_body(const FxgStatsArgs &a)
{
    extern __shared__ __attribute__((aligned(16))) u32 qs_h[];
    const u32 tid = threadIdx.x;
#ifdef FXG_QS_CLOCKS      // measurement build: when each workgroup of the launch started, left its loop and ended (100 MHz ticks), left in unused bins of the result
    const u64 qs_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    u32 *part = a.partial + (u64)blockIdx.x * FXG_QS_PART_WORDS;
    for (u32 i = tid; i < FXG_QS_LDS_WORDS; i += FXG_QS_TBLOCK) qs_h[i] = 0u;
    const bool dealt = !a.len && a.qual && a.round_robin != 0u;
    __syncthreads();
    // FXG_QS_TBLOCK is a multiple of the strips per block: item g0 + u * TBLOCK + tid is strip tid % 10 of read lo + g / 10
    const u32 sl = tid % FXG_QS_WAVES, rl = tid / FXG_QS_WAVES;
    const u32 reads_per_step = FXG_QS_TBLOCK / FXG_QS_WAVES;                                           // 96
    const u32 trip_reads = reads_per_step * FXG_QS_UNROLL;                                             // reads a trip touches
    const u32 c0 = (a.strip0 + sl) * FXG_QS_STRIP;
    const u32 nb = a.fixed_len > c0 ? (a.fixed_len - c0 < FXG_QS_STRIP ? a.fixed_len - c0 : FXG_QS_STRIP) : 0u;
    u32 mfix[4];                                                                                        // fixed-length batches: the lane's tail mask never changes
    fxg_stats_masks(nb, mfix);
    u32 since = 0, nflush = 0;                                // reads added to the LDS block since it was last cleared (an upper bound); flushes so far
    // the tested loop: ragged reads, FASTA, small batches' slices, the batch's last reads
    auto tested = [&](u64 lo, u64 hi) {
        const u64 nitems = (hi - lo) * FXG_QS_WAVES;
        u64 r0 = lo + rl;
        for (u64 g0 = 0; g0 < nitems; g0 += (u64)FXG_QS_TBLOCK * FXG_QS_UNROLL, r0 += trip_reads) {
            if (since + trip_reads > 65535u) {                    // a 16-bit counter could wrap: move the block out (uniform branch)
                __syncthreads();
                fxg_stats_flush(qs_h, part, tid, FXG_QS_TBLOCK, nflush++ == 0u);
                __syncthreads();
                since = 0;
            }
            FxgStripRow row[FXG_QS_UNROLL];
#pragma unroll
            for (u32 u = 0; u < FXG_QS_UNROLL; ++u) {
                const u64 r = r0 + (u64)u * reads_per_step;
                row[u].nb = 0u;
                if (r < hi) fxg_stats_load(a, r, a.strip0 + sl, row[u]);
            }
#pragma unroll
            for (u32 u = 0; u < FXG_QS_UNROLL; ++u) {
                if (a.len) { u32 m[4]; fxg_stats_masks(row[u].nb, m); fxg_stats_accumulate(a, row[u], sl, c0, m, qs_h); }
                else fxg_stats_accumulate(a, row[u], sl, c0, mfix, qs_h);
            }
            since += trip_reads;
        }
    };
    if (!dealt) {
        u64 lo, hi;
        fxg_stats_slice(a, blockIdx.x, &lo, &hi);
        tested(lo, hi);
    } else {
        const u32 G = gridDim.x;
        // round_robin 1: this workgroup's trips are blockIdx.x + k G -- in the piece form where the batch allows it; 3: the row-strip form everywhere;
        // 2 (measurement knob): a contiguous run of trips per workgroup through the same loop, row-strip form
        u32 R = reads_per_step, P = 0u;
        const bool piece = a.round_robin == 1u && fxg_stats_piece_plan(a, &R, &P) && ((a.fixed_len & 1u) != 0u) == ODDK;      // (the host launches the kernel that matches the length)
        const u64 ntrip = piece ? a.n / R : (a.n ? (a.n - 1) / R : 0u);            // row-strip form: trips whose every read lies below n - 1 (16-byte loads from any column)
        if (blockIdx.x == 0u) tested(ntrip * R, a.n);
        const bool rr = a.round_robin != 2u;
        const u64 per = (ntrip + G - 1) / G, first = rr ? (u64)blockIdx.x : (u64)blockIdx.x * per;
        if (first < ntrip) {
            constexpr u32 D = FXG_QS_DEPTH;                                        // trips a lane's loads run ahead of its adds
            const u64 cnt = rr ? (ntrip - first + G - 1) / G : (first + per <= ntrip ? per : ntrip - first);
            const u64 RL = (u64)R * a.stride, tb = (u64)(rr ? G : 1u) * RL;
            // piece form: this workgroup's distance to the line grid before and behind each of its trips (fxg_stats_piece_cut), pieces of a trip, pieces of the
            // batch's last trip (which ends where the whole trips end, not on the grid)
            const bool grid = piece && tb % 128u == 0u;
            const u32 d0 = grid ? (u32)((blockIdx.x * RL) & 127u) : 0u, d1 = grid ? (u32)(((blockIdx.x + 1ull) * RL) & 127u) : 0u;
            const u32 Pn = (u32)((RL + d0 - d1) >> 4), Pl = (u32)((RL + d0) >> 4);
            const bool owns_last = piece && (ntrip - 1u) % G == blockIdx.x;
            const u32 Rup = piece ? R + 8u : R;                                   // reads a trip can add to one column (a trip moved onto the grid holds up to 112 bytes more)
            // the lane's first 16 bytes (a row-strip lane whose strip lies past the reads' end loads its row's first bytes, a piece lane past the trip's last piece
            // loads the trip's first one: neither adds anything)
            u64 at = piece ? first * RL - d0 + (tid < Pl ? 16u * tid : 0u) : (first * R + rl) * a.stride + (nb ? c0 : 0u);
            FxgStripRow buf[D + 1];
#pragma unroll
            for (u32 u = 0; u <= D; ++u) { buf[u].vb = (u32x4){0u, 0u, 0u, 0u}; buf[u].vq = buf[u].vb; buf[u].nb = nb; }
            auto run = [&](const u64 trips, auto &&acc, auto pol) {
                auto step = [&](const FxgStripRow &row) {
                    if (since + Rup > 65535u) {
                        __syncthreads();
                        fxg_stats_flush(qs_h, part, tid, FXG_QS_TBLOCK, nflush++ == 0u);
                        __syncthreads();
                        since = 0;
                    }
                    acc(row);
                    since += Rup;
                };
                u64 k0 = 0;
                if (trips >= 2u * D + 1u) {
#pragma unroll
                    for (u32 u = 0; u < D; ++u) { buf[u].vb = fxg_qs_ld(a.bases + at + u * tb, pol); buf[u].vq = fxg_qs_ld(a.qual + at + u * tb, pol); }
                    for (; k0 + 2u * D + 1u <= trips; k0 += D + 1u) {                  // a group of D + 1 trips whose D successors exist: no load is tested
#pragma unroll
                        for (u32 u = 0; u <= D; ++u) {
                            FxgStripRow &in = buf[(u + D) % (D + 1u)];
                            in.vb = fxg_qs_ld(a.bases + at + (u64)D * tb, pol); in.vq = fxg_qs_ld(a.qual + at + (u64)D * tb, pol);
                            step(buf[u]);
                            at += tb;
                        }
                    }
#pragma unroll
                    for (u32 u = 0; u < D; ++u) { step(buf[u]); at += tb; }          // the D trips whose rows are already on their way
                    k0 += D;
                }
                for (; k0 < trips; ++k0, at += tb) {                                   // at most D + 1 more, one at a time
                    buf[0].vb = fxg_qs_ld(a.bases + at, pol); buf[0].vq = fxg_qs_ld(a.qual + at, pol);
                    step(buf[0]);
                }
            };
            if (piece) {
                const u32 o = tid < Pl ? (16u * tid + 8u * a.fixed_len - d0) % a.fixed_len : 0u;      // column of the lane's first byte (8 L >= 128 > d0)
                const bool mine = tid < Pn, mine_last = tid < Pl;
                auto pieces = [&](auto &pc) {
                    fxg_stats_piece_lane(a.fixed_len, o, pc);
                    run(cnt - (owns_last ? 1u : 0u), [&](const FxgStripRow &row) { if (mine) fxg_stats_accumulate_piece(a, row, pc, qs_h); }, FxgLdPiece{});
                    if (owns_last) run(1u, [&](const FxgStripRow &row) { if (mine_last) fxg_stats_accumulate_piece(a, row, pc, qs_h); }, FxgLdPiece{});
                };
                FxgPieceLane<ODDK> pc;
                pieces(pc);
            } else if constexpr (!ODDK) run(cnt, [&](const FxgStripRow &row) { fxg_stats_accumulate(a, row, sl, c0, mfix, qs_h); }, FxgLdDefault{});
        }
    }
    __syncthreads();
#ifdef FXG_QS_CLOCKS
    const u64 qs_t1 = __builtin_amdgcn_s_memrealtime();
#endif
    fxg_stats_flush(qs_h, part, tid, FXG_QS_TBLOCK, nflush++ == 0u);
#ifdef FXG_QS_CLOCKS
    if (tid == 0) {      // into bins 0..2 of (column g / 5, class g % 5): no quality byte of the measurement's input lands there
        u64 *o = a.hist + ((u64)(blockIdx.x / 5u) * FXG_QS_CLASSES + blockIdx.x % 5u) * FXG_QS_BINS;
        o[0] = qs_t0; o[1] = qs_t1; o[2] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

__global__ __launch_bounds__(FXG_QS_TBLOCK) void fxg_kernel_quality_stats(const FxgStatsArgs a) { fxg_quality_stats_body<false>(a); }
__global__ __launch_bounds__(FXG_QS_TBLOCK) void fxg_kernel_quality_stats_odd(const FxgStatsArgs a) { fxg_quality_stats_body<true>(a); }

// 64 counters per workgroup; four groups of lanes sum a quarter of the workgroups' partials each, eight loads in flight per lane (one thread per counter walking
// all 256 partials took 95 us per pass, 3 % on top of the kernel itself: profiles/r06_stats_kernel_stats.md)
#define FXG_QS_FOLD_E 64u
__global__ __launch_bounds__(256) void fxg_kernel_quality_stats_fold(const FxgStatsArgs a)
{
    __shared__ u64 s_sum[256];
    const u32 tid = threadIdx.x, e = blockIdx.x * FXG_QS_FOLD_E + (tid & 63u), q = tid >> 6;
    u64 sum = 0;
    if (e < FXG_QS_PART_WORDS) {
        const u32 *p = a.partial + e;
        u32 g = q;
        for (; g + 28u < a.nwg; g += 32u) {
            u32 v[8];
#pragma unroll
            for (u32 i = 0; i < 8u; ++i) v[i] = p[(u64)(g + 4u * i) * FXG_QS_PART_WORDS];
#pragma unroll
            for (u32 i = 0; i < 8u; ++i) sum += v[i];
        }
        for (; g < a.nwg; g += 4u) sum += p[(u64)g * FXG_QS_PART_WORDS];
    }
    s_sum[tid] = sum;
    __syncthreads();
    if (q == 0u && e < FXG_QS_PART_WORDS) fxg_stats_fold_put(a, e, s_sum[tid] + s_sum[tid + 64u] + s_sum[tid + 128u] + s_sum[tid + 192u]);
}
#endif
