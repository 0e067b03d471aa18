// fxg_stats.h -- fastx_quality_stats as a device reduction.
//
// Reference: src/fastx_quality_stats/fastx_quality_stats.c:166-216 (read_file): for every base of every read, per column
// ("cycle") and per nucleotide class, a count, min/max/sum of the quality value and a counting-sort histogram of it.  All
// of that is a function of one histogram  hist[column][class A,C,G,T,N][quality byte]  -- count, sum, min, max and the
// order statistics of print_statistics (:218-340) follow from it on the host -- so the device only builds that.
//
// Work decomposition: a workgroup owns one 16-column strip of a chunk of reads and keeps the strip's histogram
// (16 x 5 x 128 u32 = 40 KB) in LDS; one lane takes one read and adds its 16 bases with LDS atomics, starting at a
// lane-dependent column so that the lanes of a wave spread over the strip instead of hammering one column's few
// popular bins.  The strips of a chunk run on the same XCD (see the kernel), so the 128-byte lines of a row are fetched
// from HBM once and served to the other strips out of that XCD's L2.  The strip is flushed with u64 global atomics.
// HBM-bound: 2 bytes per base in, nothing out.
#pragma once
#include "fxg_device.h"

#define FXG_QS_STRIP 16u
#define FXG_QS_ROW (FXG_QS_BINS + 1u)     // LDS row of one (column, class): +1 word so that the bank depends on column and class, not only on the quality
#define FXG_QS_LDS_WORDS (FXG_QS_STRIP * FXG_QS_CLASSES * FXG_QS_ROW)

struct FxgStatsArgs {
    const uint8_t  *bases, *qual;     // qual may be null (FASTA): bin 0 counts
    const uint16_t *len;
    u64  n, total_bytes;
    u32  fixed_len, stride;
    u32  nstrips, reads_per_chunk;
    u64 *hist;                        // [hist_cols][FXG_QS_CLASSES][FXG_QS_BINS]
    u32  hist_cols;
};

#ifdef FXG_HOST_EMULATION
#define FXG_LDS_INC(p) ((void)(++*(p)))
#else
#define FXG_LDS_INC(p) ((void)atomicAdd((p), 1u))
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

// one read, one strip: h[(j * CLASSES + class) * ROW + quality byte]++ for the strip's columns j that the read has.
// Split in two so that the kernel can have the rows of several reads in flight before it touches the first.
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

FXG_HD void fxg_stats_accumulate(const FxgStripRow &o, u32 rot, u32 *h)
{
    if (o.nb == 0u) return;
    const u32 wb[4] = {o.vb.x, o.vb.y, o.vb.z, o.vb.w}, wq[4] = {o.vq.x, o.vq.y, o.vq.z, o.vq.w};
#pragma unroll
    for (u32 s = 0; s < FXG_QS_STRIP; ++s) {
        const u32 j = (s + rot) & (FXG_QS_STRIP - 1u);
        const u32 b = (wb[j >> 2] >> (8u * (j & 3u))) & 0xFFu, q = (wq[j >> 2] >> (8u * (j & 3u))) & 0x7Fu;
        const u32 k = fxg_stats_class(b);
        if (j < o.nb && k < FXG_QS_CLASSES) FXG_LDS_INC(&h[(j * FXG_QS_CLASSES + k) * FXG_QS_ROW + q]);
    }
}

FXG_HD void fxg_stats_read_strip(const FxgStatsArgs &a, u64 r, u32 strip, u32 rot, u32 *h)
{
    FxgStripRow o;
    fxg_stats_load(a, r, strip, o);
    fxg_stats_accumulate(o, rot, h);
}

#ifndef FXG_QS_UNROLL
#define FXG_QS_UNROLL 1      // reads per lane whose rows are requested before the first is consumed: measured 1/2/4/8 -> 11.8/13.7/13.7/20.0 ms
#endif

#ifndef FXG_HOST_EMULATION
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_quality_stats(const FxgStatsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32 qs_h[];
    // Workgroups are dealt to the 8 XCDs round-robin and each XCD has its own L2: the strips of one chunk must run on ONE
    // XCD to share the chunk's rows, so consecutive workgroups of an XCD (b = 8 v + xcd) take consecutive strips of a chunk.
    const u32 xcd = blockIdx.x & 7u, v = blockIdx.x >> 3;
    const u32 strip = v % a.nstrips, chunk = (v / a.nstrips) * 8u + xcd;
    const u64 lo = (u64)chunk * a.reads_per_chunk;
    if (lo >= a.n) return;
    for (u32 i = threadIdx.x; i < FXG_QS_LDS_WORDS; i += FXG_BLOCK) qs_h[i] = 0u;
    __syncthreads();
    const u64 hi = lo + a.reads_per_chunk < a.n ? lo + a.reads_per_chunk : a.n;
    for (u64 r = lo + threadIdx.x; r < hi; r += (u64)FXG_BLOCK * FXG_QS_UNROLL) {
        FxgStripRow row[FXG_QS_UNROLL];
#pragma unroll
        for (u32 u = 0; u < FXG_QS_UNROLL; ++u) {
            row[u].nb = 0u;
            if (r + (u64)u * FXG_BLOCK < hi) fxg_stats_load(a, r + (u64)u * FXG_BLOCK, strip, row[u]);
        }
#pragma unroll
        for (u32 u = 0; u < FXG_QS_UNROLL; ++u) fxg_stats_accumulate(row[u], threadIdx.x + u, qs_h);
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < FXG_QS_STRIP * FXG_QS_CLASSES * FXG_QS_BINS; i += FXG_BLOCK) {
        const u32 row = i / FXG_QS_BINS, bin = i % FXG_QS_BINS;             // row = column-in-strip * CLASSES + class
        const u32 v = qs_h[row * FXG_QS_ROW + bin];
        const u32 col = strip * FXG_QS_STRIP + row / FXG_QS_CLASSES;
        if (v != 0u && col < a.hist_cols) atomicAdd(&a.hist[((u64)col * FXG_QS_CLASSES + row % FXG_QS_CLASSES) * FXG_QS_BINS + bin], (u64)v);
    }
}
#endif
