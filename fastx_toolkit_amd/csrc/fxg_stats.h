// fxg_stats.h -- fastx_quality_stats as a device reduction.
//
// Reference: src/fastx_quality_stats/fastx_quality_stats.c:166-216 (read_file): for every base of every read, per column
// ("cycle") and per nucleotide class, a count, min/max/sum of the quality value and a counting-sort histogram of it.  All
// of that is a function of one histogram  hist[column][class A,C,G,T,N][quality byte]  -- count, sum, min, max and the
// order statistics of print_statistics (:218-340) follow from it on the host -- so the device only builds that.
//
// Work decomposition: every row is read from HBM once.  A workgroup owns a contiguous slice of the reads and the block
// histogram of up to 160 columns -- 10 strips of 16 columns x 6 rows (A C G T N, other) x a 64-value window of the quality
// byte, 16-bit counters, the two columns of a pair in one word, rows padded to 65 words so that the LDS bank depends on
// strip, column and class as well as on the quality: 122 KB of LDS, updated with ds_add.  A work item is one 16-byte
// piece (strip) of one row; consecutive lanes take consecutive pieces, so a wave reads 1 KB of contiguous rows per load
// and its lanes spread over all ten strips (few same-bank, fewer same-address updates).  Before a counter could wrap
// (65 535 reads) the workgroup adds the block to ITS OWN u32 partial in HBM (no atomics) and clears it.  Quality bytes
// outside the window (Phred+33 codes 33..96 are inside) go straight to the result with a global atomic.  A second kernel
// folds the partials into the caller's u64 histogram.  Reads longer than 160 take one pass per column block.
// HBM-bound by design (2 bytes per base in, nothing out); measured limiter: VALU + LDS update issue, see DESIGN.md.
#pragma once
#include "fxg_device.h"

#define FXG_QS_STRIP 16u
#define FXG_QS_WAVES 10u                                   // strips per column block (the name is historic: not tied to waves any more)
#define FXG_QS_TBLOCK 1024u                                // one workgroup per CU (LDS), so make it as wide as a workgroup gets
#define FXG_QS_BLOCK_COLS (FXG_QS_WAVES * FXG_QS_STRIP)    // 160 columns per pass
#define FXG_QS_WBASE 33u                                   // first quality byte of the LDS window
#define FXG_QS_WBINS 64u
#define FXG_QS_PART_WORDS (FXG_QS_BLOCK_COLS * FXG_QS_CLASSES * FXG_QS_WBINS)   // one workgroup's partial (u32) = 51 200 counters
#define FXG_QS_LROWS 6u                                    // LDS rows per column pair: A C G T N + one for bytes that are none of them (never flushed)
#define FXG_QS_LROW_WORDS (FXG_QS_WBINS + 1u)              // +1: bank = f(strip, column pair, row, quality), not of the quality alone
#define FXG_QS_LDS_WORDS ((FXG_QS_BLOCK_COLS / 2u) * FXG_QS_LROWS * FXG_QS_LROW_WORDS)   // word = two u16 counters: even column low, odd column high
#ifndef FXG_QS_UNROLL
#define FXG_QS_UNROLL 2u                                   // reads per lane and trip
#endif

struct FxgStatsArgs {
    const uint8_t  *bases, *qual;     // qual may be null (FASTA): bin 0 counts
    const uint16_t *len;
    u64  n, total_bytes;
    u32  fixed_len, stride;
    u32  strip0;                      // first strip of this pass (column block)
    u32  nwg;                         // workgroups = slices of the reads
    u32 *partial;                     // [nwg][FXG_QS_PART_WORDS]
    u64 *hist;                        // [hist_cols][FXG_QS_CLASSES][FXG_QS_BINS]
    u32  hist_cols;
};

#ifdef FXG_HOST_EMULATION
#define FXG_LDS_ADD(p, v) ((void)(*(p) += (v)))
#define FXG_GLOBAL_INC64(p) ((void)(++*(p)))
#else
#define FXG_LDS_ADD(p, v) ((void)atomicAdd((p), (v)))
#define FXG_GLOBAL_INC64(p) ((void)atomicAdd((p), 1ull))
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

// LDS word of (column j of strip `wave`, row k, window bin w); the counter is its low half for even j, its high half for odd j
FXG_HD u32 fxg_stats_word(u32 wave, u32 j, u32 k, u32 w) { return (((wave * (FXG_QS_STRIP / 2u) + (j >> 1)) * FXG_QS_LROWS + k) * FXG_QS_LROW_WORDS) + w; }

// row of a base that is known to be a letter (bit 6 set, bit 7 clear): A C G T N -> 0..4 in either case, any other letter -> 5.
// The low five bits of the five letters are 1, 3, 7, 20, 14; their low three bits (1, 3, 7, 4, 6) index a 3-bit table.
FXG_HD u32 fxg_stats_row_of_letter(u32 b)
{
    const u32 x = b & 31u;
    const u32 valid = (((1u << 1) | (1u << 3) | (1u << 7) | (1u << 20) | (1u << 14)) >> x) & 1u;
    const u32 k = (0x503200u >> (3u * (x & 7u))) & 7u;
    return valid ? k : 5u;
}

FXG_HD void fxg_stats_accumulate(const FxgStatsArgs &a, const FxgStripRow &o, u32 wave, u32 col0, u32 *lds)
{
    if (o.nb == 0u) return;
    u32 wb[4] = {o.vb.x, o.vb.y, o.vb.z, o.vb.w}, wq[4] = {o.vq.x, o.vq.y, o.vq.z, o.vq.w};
    // Fast path: 16 letters whose quality bytes all lie in the window -- no per-base test, no branch.  Bytes past the end of
    // the read (the last strip of a row, ragged reads) are first replaced by '@' (a letter that is no base: counted in the
    // spare row) with quality WBASE, so that short strips take the fast path as well: in this item order every wave holds
    // some, and a wave that has ONE lane on the slow path executes the slow path.
    const u32 K_lo = (128u - FXG_QS_WBASE) * 0x01010101u, K_hi = (128u - (FXG_QS_WBASE + FXG_QS_WBINS)) * 0x01010101u;
    u32 bad = 0u;
#pragma unroll
    for (u32 d = 0; d < 4u; ++d) {
        const int keep = (int)o.nb - (int)(4u * d);
        const u32 m = fxg_lowbytes32(keep < 0 ? 0 : (keep > 4 ? 4 : keep));
        wb[d] = (wb[d] & m) | (0x40404040u & ~m);
        wq[d] = (wq[d] & m) | ((FXG_QS_WBASE * 0x01010101u) & ~m);
        bad |= (wb[d] & 0xC0C0C0C0u) ^ 0x40404040u;                                     // not a letter
        bad |= (fxg_ge_flags(wq[d], K_lo) ^ 0x80808080u) | fxg_ge_flags(wq[d], K_hi);   // below / above the window (also catches bytes >= 128)
    }
    if (!bad) {
#pragma unroll
        for (u32 j = 0; j < FXG_QS_STRIP; ++j) {
            const u32 sh = 8u * (j & 3u);
            const u32 k = fxg_stats_row_of_letter((wb[j >> 2] >> sh) & 0xFFu);
            const u32 w = ((wq[j >> 2] >> sh) & 0xFFu) - FXG_QS_WBASE;
            FXG_LDS_ADD(&lds[fxg_stats_word(wave, j, k, w)], (j & 1u) ? 0x10000u : 1u);
        }
        return;
    }
#pragma unroll 1
    for (u32 j = 0; j < o.nb; ++j) {                                                     // ragged ends, odd bytes, rare qualities: one base at a time
        const u32 b = (wb[j >> 2] >> (8u * (j & 3u))) & 0xFFu, q = (wq[j >> 2] >> (8u * (j & 3u))) & 0xFFu;
        const u32 k = fxg_stats_class(b);
        if (k >= FXG_QS_CLASSES || q >= FXG_QS_BINS) continue;
        const u32 w = q - FXG_QS_WBASE;
        if (w < FXG_QS_WBINS) FXG_LDS_ADD(&lds[fxg_stats_word(wave, j, k, w)], (j & 1u) ? 0x10000u : 1u);
        else if (col0 + j < a.hist_cols) FXG_GLOBAL_INC64(&a.hist[((u64)(col0 + j) * FXG_QS_CLASSES + k) * FXG_QS_BINS + q]);
    }
}

// slice of the reads that workgroup g owns
FXG_HD void fxg_stats_slice(const FxgStatsArgs &a, u32 g, u64 *lo, u64 *hi)
{
    const u64 per = (a.n + a.nwg - 1) / a.nwg;
    *lo = (u64)g * per < a.n ? (u64)g * per : a.n;
    *hi = *lo + per < a.n ? *lo + per : a.n;
}

// thread `t` of `nt`: add the LDS block to the workgroup's partial ([column-in-block][class][window bin], u32) and clear it
FXG_HD void fxg_stats_flush(u32 *lds, u32 *part, u32 t, u32 nt)
{
    for (u32 i = t; i < (FXG_QS_BLOCK_COLS / 2u) * FXG_QS_LROWS * FXG_QS_WBINS; i += nt) {
        const u32 w = i % FXG_QS_WBINS, pk = i / FXG_QS_WBINS, k = pk % FXG_QS_LROWS, pair = pk / FXG_QS_LROWS;
        u32 *l = lds + pk * FXG_QS_LROW_WORDS + w;
        const u32 v = *l;
        if (v == 0u) continue;
        *l = 0u;
        if (k >= FXG_QS_CLASSES) continue;                                   // the row of bytes that are not A C G T N
        u32 *p = part + ((2u * pair) * FXG_QS_CLASSES + k) * FXG_QS_WBINS + w;
        p[0] += v & 0xFFFFu;
        p[FXG_QS_CLASSES * FXG_QS_WBINS] += v >> 16;
    }
}

// element e of the fold: counter (column-in-block, class, window bin) summed over the workgroups' partials
FXG_HD void fxg_stats_fold(const FxgStatsArgs &a, u32 e)
{
    const u32 w = e % FXG_QS_WBINS, ck = e / FXG_QS_WBINS;
    const u32 col = a.strip0 * FXG_QS_STRIP + ck / FXG_QS_CLASSES, k = ck % FXG_QS_CLASSES;
    if (col >= a.hist_cols) return;
    u64 sum = 0;
    for (u32 g = 0; g < a.nwg; ++g) sum += a.partial[(u64)g * FXG_QS_PART_WORDS + e];
    if (sum) a.hist[((u64)col * FXG_QS_CLASSES + k) * FXG_QS_BINS + FXG_QS_WBASE + w] += sum;
}

// work item g of a slice that starts at read lo: read lo + g / 10, strip g % 10 of the block
FXG_HD void fxg_stats_item(u64 lo, u64 g, u64 *r, u32 *sl) { *r = lo + g / FXG_QS_WAVES; *sl = (u32)(g % FXG_QS_WAVES); }

#ifndef FXG_HOST_EMULATION
__global__ __launch_bounds__(FXG_QS_TBLOCK) void fxg_kernel_quality_stats(const FxgStatsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32 qs_h[];
    const u32 tid = threadIdx.x;
    u32 *part = a.partial + (u64)blockIdx.x * FXG_QS_PART_WORDS;
    for (u32 i = tid; i < FXG_QS_LDS_WORDS; i += FXG_QS_TBLOCK) qs_h[i] = 0u;
    for (u32 i = tid; i < FXG_QS_PART_WORDS; i += FXG_QS_TBLOCK) part[i] = 0u;
    __syncthreads();
    u64 lo, hi;
    fxg_stats_slice(a, blockIdx.x, &lo, &hi);
    const u64 nitems = (hi - lo) * FXG_QS_WAVES;
    const u32 trip_reads = (FXG_QS_TBLOCK * FXG_QS_UNROLL + FXG_QS_WAVES - 1u) / FXG_QS_WAVES + 1u;   // reads a trip can touch
    u32 since = 0;                                            // reads added to the LDS block since it was last cleared
    for (u64 g0 = 0; g0 < nitems; g0 += (u64)FXG_QS_TBLOCK * FXG_QS_UNROLL) {
        if (since + trip_reads > 65535u) {                    // a 16-bit counter could wrap: move the block out (uniform branch)
            __syncthreads();
            fxg_stats_flush(qs_h, part, tid, FXG_QS_TBLOCK);
            __syncthreads();
            since = 0;
        }
        FxgStripRow row[FXG_QS_UNROLL];
        u32 sl[FXG_QS_UNROLL];
#pragma unroll
        for (u32 u = 0; u < FXG_QS_UNROLL; ++u) {
            const u64 g = g0 + (u64)u * FXG_QS_TBLOCK + tid;
            u64 r = 0;
            row[u].nb = 0u; sl[u] = 0u;
            if (g < nitems) { fxg_stats_item(lo, g, &r, &sl[u]); fxg_stats_load(a, r, a.strip0 + sl[u], row[u]); }
        }
#pragma unroll
        for (u32 u = 0; u < FXG_QS_UNROLL; ++u) fxg_stats_accumulate(a, row[u], sl[u], (a.strip0 + sl[u]) * FXG_QS_STRIP, qs_h);
        since += trip_reads;
    }
    __syncthreads();
    fxg_stats_flush(qs_h, part, tid, FXG_QS_TBLOCK);
}

__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_quality_stats_fold(const FxgStatsArgs a)
{
    const u32 e = blockIdx.x * FXG_BLOCK + threadIdx.x;
    if (e < FXG_QS_PART_WORDS) fxg_stats_fold(a, e);
}
#endif
