// fxg_device.h -- device-side building blocks of the gfx950 FASTQ engine (wave64, LDS, HBM streaming).
//
// Nothing here is GEMM shaped: every kernel is a byte scan / gather bounded by HBM bandwidth, except
// the adapter aligner which is a per-thread fp32 dynamic program bounded by VALU issue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fxg.h"

typedef unsigned int       u32;
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// pure per-thread helpers are __host__ __device__ so that tests/emu can run them on the CPU
#define FXG_HD __host__ __device__ __forceinline__

// timing-ablation switches exist only in -DFXG_ABLATION builds (scripts/ablate.py); the product build folds them to 0
#ifdef FXG_ABLATION
#define FXG_DBG(a, bit) ((a).debug & (bit))
#else
#define FXG_DBG(a, bit) 0u
#endif

#define FXG_BLOCK 256           // threads per workgroup = 4 wave64
#define FXG_WAVES (FXG_BLOCK / 64)
#define FXG_MAX_TILE 256        // reads per tile (one thread decides one read)
#define FXG_TICKET_GROUPS 8      // a single device-scope counter saturates near 88 tickets/us; shard it (one per XCD)
#define FXG_TICKET_STRIDE 32     // u32 words between dispensers (128 B: one cache line each)

// ------------------------------------------------------------------------------------------------
// launch arguments (passed by value; adapter bytes therefore live in the kernarg segment / SGPRs)
// ------------------------------------------------------------------------------------------------
struct FxgKArgs {
    const uint8_t  *bases;
    const uint8_t  *qual;
    const uint16_t *len;
    u64  n;
    u64  total_bytes;       // n * stride
    u32  fixed_len;
    u32  stride;
    u32  tile_reads;        // power of two, <= FXG_MAX_TILE
    u32  ntiles;
    // outputs
    u32      *res;
    uint8_t  *out_bases;
    uint8_t  *out_qual;
    uint16_t *out_len;
    u32      *kept_index;
    u64      *out_off;
    // engine state
    u64 *status_cnt;        // [ntiles] decoupled look-back granules (kept reads)
    u64 *status_bytes;      // [ntiles] decoupled look-back granules (kept bytes)
    u64 *partial;           // [count grid][FXG_NCOUNTERS]
    u32 *ticket;            // dynamic tile dispensers, FXG_TICKET_STRIDE words apart (zeroed before every launch)
    u32  ticket_groups;     // number of dispensers (<= 8): dispenser g hands out tiles g, g+groups, g+2*groups, ...
    u32 *errflag;
    u32  compact;           // 1 = stream-compact kept reads into out_bases/out_qual
    u32  debug;             // FXG_DEBUG ablation bits (timing experiments only; results are wrong when set)
    // folded tool parameters
    u32  stages;
    u32  tq;                // quality trimmer: byte >= tq  <=>  q >= -t      (0..128)
    u32  fq;                // quality filter : byte <  fq  <=>  q <  -q      (0..128)
    int  qt_min_len;
    int  qf_keep_pct;       // 100 - p
    u32  qf_drop_all;       // quirk F2: -p omitted and -q > 93
    int  alen;
    u32  clip_min_len;
    int  clip_keep_delta;
    int  clip_min_adapter_len;
    u32  clip_flags;
    int  ft_first, ft_last;
    u32  ft_trim_end, ft_min_len;
    u32  mask_char;         // fastq_masker -r; the mask threshold shares `fq` (byte < fq is masked)
    u32  nf_keep_n;         // fastq_to_fasta -n
    u64 *extra;             // [0] masked reads, [1] masked nucleotides (fastq_masker report)
    char adapter[100];
};

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 fxg_lane() { return threadIdx.x & 63u; }

// byte >= thr for every byte of w (bytes assumed < 128, thr in 0..128): bit 7 of each byte.
// K = (128 - thr) * 0x01010101
FXG_HD u32 fxg_ge_flags(u32 w, u32 K) { return (((w & 0x7f7f7f7fu) + K) | w) & 0x80808080u; }

// gather the four bit-7 flags of a dword into bits 0..3
FXG_HD u32 fxg_pack4(u32 f)
{
    u32 t = f >> 7;          // bits 0, 8, 16, 24
    t |= t >> 7;             // bit 1 <- 8, bit 17 <- 24
    t |= t >> 14;            // bit 2 <- 16, bit 3 <- 17
    return t & 0xFu;
}

FXG_HD u32 fxg_mask16(u32x4 v, u32 K)
{
    return fxg_pack4(fxg_ge_flags(v.x, K)) | (fxg_pack4(fxg_ge_flags(v.y, K)) << 4) |
           (fxg_pack4(fxg_ge_flags(v.z, K)) << 8) | (fxg_pack4(fxg_ge_flags(v.w, K)) << 12);
}

// 16-byte load from an arbitrarily aligned address (one global_load_dwordx4 on gfx950)
FXG_HD u32x4 fxg_ld16(const uint8_t *p)
{
    u32x4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// bytes [vlo, vhi) of a 16-byte window starting at absolute offset off; the rest is zero.
// The fast path needs the whole window inside [0, total); the slow path touches only needed bytes.
FXG_HD u32x4 fxg_window(const uint8_t *arr, long long off, u64 total, int vlo, int vhi)
{
    if (off >= 0 && (u64)off + 16 <= total) return fxg_ld16(arr + off);
    u64 lo = 0, hi = 0;
    for (int i = vlo; i < vhi; ++i) {
        const u64 b = arr[off + i];
        if (i < 8) lo |= b << (8 * i); else hi |= b << (8 * (i - 8));
    }
    u32x4 v = {(u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32)};
    return v;
}

FXG_HD u64 fxg_lowbytes64(int x)   // x in [0, 8]: low x bytes set
{
    return x >= 8 ? ~0ull : ((1ull << (8 * x)) - 1ull);
}

// keep bytes [lo, hi) of v (0 <= lo <= hi <= 16)
FXG_HD u32x4 fxg_keep_bytes(u32x4 v, int lo, int hi)
{
    const int lo0 = lo < 8 ? lo : 8, hi0 = hi < 8 ? hi : 8;
    const int lo1 = lo > 8 ? lo - 8 : 0, hi1 = hi > 8 ? hi - 8 : 0;
    const u64 m0 = fxg_lowbytes64(hi0) & ~fxg_lowbytes64(lo0);
    const u64 m1 = fxg_lowbytes64(hi1) & ~fxg_lowbytes64(lo1);
    v.x &= (u32)m0; v.y &= (u32)(m0 >> 32); v.z &= (u32)m1; v.w &= (u32)(m1 >> 32);
    return v;
}

FXG_HD u32x4 fxg_reverse16(u32x4 v)
{
    u32x4 r = {__builtin_bswap32(v.w), __builtin_bswap32(v.z), __builtin_bswap32(v.y), __builtin_bswap32(v.x)};
    return r;
}

// A<->T, C<->G, N->N (case preserved) on four packed ASCII bases:  A^T = 0x15, C^G = 0x04.
//   bit1 clear          -> A/T
//   bit1 set, bit3 clear -> C/G
FXG_HD u32 fxg_complement4(u32 w)
{
    const u32 b1 = (w >> 1) & 0x01010101u;
    const u32 at = b1 ^ 0x01010101u;
    const u32 cg = b1 & ~(w >> 3) & 0x01010101u;
    return w ^ (at * 0x15u) ^ (cg << 2);
}

// nonzero if any byte of w selected by m (0x00/0xFF per byte) is not one of ACGTN/acgtn
FXG_HD u32 fxg_invalid_bases4(u32 w, u32 m)
{
    const u32 x = (w & 0xDFDFDFDFu) ^ 0x40404040u;         // A=01 C=03 G=07 N=0E T=14
    const u32 LUT = (1u << 0x01) | (1u << 0x03) | (1u << 0x07) | (1u << 0x0E) | (1u << 0x14);
    const u32 nb = (~(LUT >> (x & 31u)) & m) | (~(LUT >> ((x >> 8) & 31u)) & (m >> 8)) |
                   (~(LUT >> ((x >> 16) & 31u)) & (m >> 16)) | (~(LUT >> ((x >> 24) & 31u)) & (m >> 24));
    return (x & 0xE0E0E0E0u & m) | (nb & 1u);
}

// ------------------------------------------------------------------------------------------------
// bit-range queries on an LDS bitmap (bit i = byte i of the tile)
// ------------------------------------------------------------------------------------------------
// 1 + index (relative to s0) of the highest set bit in [s0, s0+n), 0 if none
FXG_HD u32 fxg_bits_last(const u32 *bm, u32 s0, u32 n)
{
    if (n == 0) return 0;
    const u32 e1 = s0 + n - 1;
    int w = (int)(e1 >> 5);
    const int w0 = (int)(s0 >> 5);
    u32 x = bm[w] & (0xFFFFFFFFu >> (31u - (e1 & 31u)));
    for (;;) {
        if (w == w0) x &= 0xFFFFFFFFu << (s0 & 31u);
        if (x) return ((u32)w << 5) + 32u - (u32)__builtin_clz(x) - s0;
        if (w == w0) return 0;
        --w;
        x = bm[w];
    }
}

// number of set bits in [s0, s0+n)
FXG_HD u32 fxg_bits_count(const u32 *bm, u32 s0, u32 n)
{
    if (n == 0) return 0;
    const u32 e1 = s0 + n - 1;
    const u32 w0 = s0 >> 5, w1 = e1 >> 5;
    u32 c = 0;
    for (u32 w = w0; w <= w1; ++w) {
        u32 x = bm[w];
        if (w == w0) x &= 0xFFFFFFFFu << (s0 & 31u);
        if (w == w1) x &= 0xFFFFFFFFu >> (31u - (e1 & 31u));
        c += (u32)__builtin_popcount(x);
    }
    return c;
}

// ------------------------------------------------------------------------------------------------
// inter-workgroup granules (decoupled look-back).  One naturally aligned u64 = {status:2, value:62},
// written by ONE relaxed agent-scope store (sc1) and polled with relaxed agent-scope loads: the data
// is its own flag, so no fence is needed and nothing depends on workgroup placement.
// ------------------------------------------------------------------------------------------------
#define FXG_ST_INVALID 0ull
#define FXG_ST_AGG     1ull
#define FXG_ST_PREFIX  2ull
#define FXG_ST_VALUE(x) ((x) & 0x3FFFFFFFFFFFFFFFull)

__device__ __forceinline__ void fxg_granule_store(u64 *g, u64 status, u64 value)
{
    __hip_atomic_store(g, (status << 62) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 fxg_granule_load(u64 *g)
{
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum over the 32 lanes of each half-wave, result in every lane of that half
__device__ __forceinline__ u64 fxg_half_sum(u64 v)
{
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) {
        u32 lo = __shfl_xor((u32)v, d, 64), hi = __shfl_xor((u32)(v >> 32), d, 64);
        v += ((u64)hi << 32) | lo;
    }
    return v;
}

// Both helpers are executed by wave 0 only.  Lanes 0..31 handle the kept-read count, lanes 32..63 the
// kept-byte count.
//
// fxg_publish_aggregate: as soon as a tile knows its own totals it publishes them (tile 0 publishes its
// inclusive prefix right away).  It never waits, so every running workgroup always makes progress.
__device__ __forceinline__ void fxg_publish_aggregate(const FxgKArgs &a, u32 tile, u64 agg_cnt, u64 agg_bytes)
{
    const u32 lane = fxg_lane();
    if ((lane & 31u) == 0u) {
        u64 *st = (lane >> 5) ? a.status_bytes : a.status_cnt;
        fxg_granule_store(st + tile, tile == 0 ? FXG_ST_PREFIX : FXG_ST_AGG, (lane >> 5) ? agg_bytes : agg_cnt);
    }
}

// fxg_resolve_prefix: walk back over predecessor tiles, 32 per step, summing aggregates until a tile with
// a full prefix is met; then publish this tile's inclusive prefix.  Tiles are handed out by a global
// ticket, so every predecessor is owned by a workgroup that is already running: the wait terminates
// whatever the residency or placement.  Returns the exclusive prefix in every lane.
__device__ __forceinline__ void fxg_resolve_prefix(const FxgKArgs &a, u32 tile, u64 agg_cnt, u64 agg_bytes,
                                                   u64 *base_cnt, u64 *base_bytes)
{
    const u32 lane = fxg_lane();
    const u32 half = lane >> 5, hl = lane & 31u;
    u64 *st = half ? a.status_bytes : a.status_cnt;
    const u64 agg = half ? agg_bytes : agg_cnt;
    u64 running = 0;
    if (tile != 0) {
        long long pos = (long long)tile - 1 - (long long)hl;
        bool done = false;
        const u64 t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        u32 spins = 0;
        for (;;) {
            u64 v = (FXG_ST_PREFIX << 62);                  // virtual tile -1: prefix 0
            if (!done && pos >= 0) v = fxg_granule_load(st + pos);
            const u32 s = (u32)(v >> 62);
            const u64 b_inv = __ballot(s == FXG_ST_INVALID);
            const u64 b_pfx = __ballot(s == FXG_ST_PREFIX);
            const u32 inv = (u32)(b_inv >> (32 * half)), pfx = (u32)(b_pfx >> (32 * half));
            const u32 fp = pfx ? (u32)__builtin_ctz(pfx) : 32u;          // nearest predecessor with a full prefix
            const u32 need = (fp >= 31u) ? 0xFFFFFFFFu : ((2u << fp) - 1u);
            const bool ready = !done && ((inv & need) == 0u);
            // every lane takes part in the shuffles; only ready halves consume the sum
            const u64 contrib = (ready && hl <= fp) ? FXG_ST_VALUE(v) : 0ull;
            const u64 sum = fxg_half_sum(contrib);
            if (ready) {
                running += sum;
                if (fp < 32u) done = true; else pos -= 32;
            }
            if (__ballot(!done) == 0ull) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0u) {   // never hang the GPU: 2 s without progress, or another workgroup already gave up
                const bool late = __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull;
                const u32 flagged = __hip_atomic_load(a.errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & FXG_DEV_ERR_SCAN_TIMEOUT;
                if (late || flagged) {
                    if (lane == 0) atomicOr(a.errflag, FXG_DEV_ERR_SCAN_TIMEOUT);
                    break;
                }
            }
        }
        if (hl == 0) fxg_granule_store(st + tile, FXG_ST_PREFIX, running + agg);
    }
    *base_cnt = ((u64)__shfl((u32)(running >> 32), 0, 64) << 32) | __shfl((u32)running, 0, 64);
    *base_bytes = ((u64)__shfl((u32)(running >> 32), 32, 64) << 32) | __shfl((u32)running, 32, 64);
}

// ------------------------------------------------------------------------------------------------
// workgroup exclusive scan of (keep, out_len) over FXG_BLOCK threads.  scratch: u32[2*FXG_WAVES + 2]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fxg_block_scan2(u32 c, u32 b, u32 *scratch, u32 *ex_c, u32 *ex_b, u32 *tot_c, u32 *tot_b)
{
    const u32 lane = fxg_lane(), wave = threadIdx.x >> 6;
    u32 ic = c, ib = b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 tc = __shfl_up(ic, d, 64), tb = __shfl_up(ib, d, 64);
        if ((int)lane >= d) { ic += tc; ib += tb; }
    }
    if (lane == 63) { scratch[wave] = ic; scratch[FXG_WAVES + wave] = ib; }
    __syncthreads();
    u32 oc = 0, ob = 0, sc = 0, sb = 0;
#pragma unroll
    for (int w = 0; w < FXG_WAVES; ++w) {
        const u32 wc = scratch[w], wb = scratch[FXG_WAVES + w];
        if (w < (int)wave) { oc += wc; ob += wb; }
        sc += wc; sb += wb;
    }
    *ex_c = oc + ic - c; *ex_b = ob + ib - b; *tot_c = sc; *tot_b = sb;
}

// ------------------------------------------------------------------------------------------------
// order-preserving gather of one tile's kept reads into the packed output.
//   v_off[0..nreads] : exclusive prefix of kept lengths inside the tile (LDS)
//   v_src[r]         : tile-relative source byte of output byte 0 of read r (LDS)
//   REV              : output byte k comes from source byte v_src[r] - k, complemented (bases)
// Work item = one 16-byte aligned chunk of the GLOBAL output; consecutive lanes write consecutive
// chunks (1 KiB per wave store).  A chunk that straddles reads is assembled from one window per read.
// ------------------------------------------------------------------------------------------------
// One source segment of an output chunk: chunk bytes [blo, bhi) come from the 16-byte window that starts
// `src` bytes after the tile's first input byte (read backwards when REV).  Everything per lane is 32-bit and
// tile-relative; the 64-bit bases are wave-uniform and stay in SGPRs.
struct FxgSeg { int src; int blo, bhi; };

template <bool REV>
FXG_HD FxgSeg fxg_make_seg(const u32 *v_off, const u32 *v_src, u32 r, u32 o, u32 seg_end, int cs)
{
    FxgSeg g;
    g.blo = (int)o - cs;
    g.bhi = (int)seg_end - cs;
    const int j0 = (int)(o - v_off[r]);
    g.src = REV ? (int)v_src[r] - j0 + g.blo - 15 : (int)v_src[r] + j0 - g.blo;
    return g;
}

// The 16-byte window of one array for one segment, reversed/complemented/validated when REV, masked to
// [blo, bhi).  tile_ptr = array + tile_in_base (uniform); [lo_ok, hi_ok) = tile-relative range of the array.
template <bool REV, bool BASES, bool MASK = false>
FXG_HD u32x4 fxg_seg_bytes(const uint8_t *tile_ptr, int lo_ok, int hi_ok, const FxgSeg &g, u32 *bad, const uint8_t *tile_q = nullptr,
                           u32 Kmask = 0u, u32 mask4 = 0u)
{
    const int vlo = REV ? 16 - g.bhi : g.blo, vhi = REV ? 16 - g.blo : g.bhi;
    const bool inside = g.src >= lo_ok && g.src + 16 <= hi_ok;
    u32x4 w, wq = {0u, 0u, 0u, 0u};
    if (inside) { w = fxg_ld16(tile_ptr + g.src); if (MASK) wq = fxg_ld16(tile_q + g.src); }
    else {                                         // window pokes out of the array: touch only the needed bytes
        u64 lo = 0, hi = 0, qlo = 0, qhi = 0;
        for (int i = vlo; i < vhi; ++i) {
            const u64 b = tile_ptr[g.src + i];
            if (i < 8) lo |= b << (8 * i); else hi |= b << (8 * (i - 8));
            if (MASK) { const u64 q = tile_q[g.src + i]; if (i < 8) qlo |= q << (8 * i); else qhi |= q << (8 * (i - 8)); }
        }
        w = (u32x4){(u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32)};
        if (MASK) wq = (u32x4){(u32)qlo, (u32)(qlo >> 32), (u32)qhi, (u32)(qhi >> 32)};
    }
    if (MASK) {                                    // fastq_masker.c:94-99: base := mask character where quality < threshold
        const u32 ws[4] = {w.x, w.y, w.z, w.w}, qs[4] = {wq.x, wq.y, wq.z, wq.w};
        u32 o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 low = (fxg_ge_flags(qs[i], Kmask) ^ 0x80808080u) >> 7;     // 0x01 per byte with quality below the threshold
            const u32 m = low * 0xFFu;
            o[i] = (ws[i] & ~m) | (mask4 & m);
        }
        w = (u32x4){o[0], o[1], o[2], o[3]};
    }
    if (REV) {
        w = fxg_reverse16(w);
        if (BASES) {
            const u32x4 m = fxg_keep_bytes((u32x4){~0u, ~0u, ~0u, ~0u}, g.blo, g.bhi);
            *bad |= fxg_invalid_bases4(w.x, m.x) | fxg_invalid_bases4(w.y, m.y) | fxg_invalid_bases4(w.z, m.z) | fxg_invalid_bases4(w.w, m.w);
            w.x = fxg_complement4(w.x); w.y = fxg_complement4(w.y); w.z = fxg_complement4(w.z); w.w = fxg_complement4(w.w);
        }
    }
    return fxg_keep_bytes(w, g.blo, g.bhi);
}

// one array of one chunk: both candidate windows in flight, then the (rare) tail of further segments
template <bool REV, bool BASES, bool MASK = false>
FXG_HD void fxg_gather_array(const uint8_t *tile_ptr, int lo_ok, int hi_ok, uint8_t *out_chunk, const u32 *v_off, const u32 *v_src,
                             const FxgSeg &s1, const FxgSeg &s2, bool two, bool more, u32 r_next, u32 o_next, u32 o_end, int cs,
                             int lo_c, int hi_c, u32 *bad, const uint8_t *tile_q = nullptr, u32 Kmask = 0u, u32 mask4 = 0u)
{
    u32x4 acc = fxg_seg_bytes<REV, BASES, MASK>(tile_ptr, lo_ok, hi_ok, s1, bad, tile_q, Kmask, mask4);
    if (two) acc |= fxg_seg_bytes<REV, BASES, MASK>(tile_ptr, lo_ok, hi_ok, s2, bad, tile_q, Kmask, mask4);
    if (more) {
        u32 r = r_next, o = o_next;
        while (o < o_end) {
            const u32 e = v_off[r + 1];
            if (e > o) {
                const u32 se = e < o_end ? e : o_end;
                const FxgSeg g = fxg_make_seg<REV>(v_off, v_src, r, o, se, cs);
                acc |= fxg_seg_bytes<REV, BASES, MASK>(tile_ptr, lo_ok, hi_ok, g, bad, tile_q, Kmask, mask4);
                o = se;
            }
            ++r;
        }
    }
    if (lo_c == 0 && hi_c == 16) { *reinterpret_cast<u32x4 *>(out_chunk) = acc; return; }
    // first / last chunk of the tile: the neighbouring tile owns the other bytes
    const u64 b0 = ((u64)acc.y << 32) | acc.x, b1 = ((u64)acc.w << 32) | acc.z;
    for (int i = lo_c; i < hi_c; ++i) out_chunk[i] = (uint8_t)((i < 8 ? b0 : b1) >> (8 * (i & 7)));
}

template <bool REV, bool MASK = false>
FXG_HD u32 fxg_tile_gather(const FxgKArgs &a, const u32 *v_off, const u32 *v_src, u32 nreads,
                           u64 tile_in_base, u64 B, u32 S, u32 tid, u32 nthreads)
{
    if (S == 0) return 0u;
    const u32 Kmask = (128u - a.fq) * 0x01010101u, mask4 = (a.mask_char & 0xFFu) * 0x01010101u;
    const bool has_q = a.qual != nullptr && a.out_qual != nullptr && !FXG_DBG(a, 4u);
    // wave-uniform 64-bit quantities
    const u64 c_first = B >> 4;
    const u32 nchunks = (u32)(((B + S - 1) >> 4) - c_first) + 1u;
    const int head = (int)(B & 15u);                                   // chunk 0 starts `head` bytes before the tile's output
    const uint8_t *src_b = a.bases + tile_in_base, *src_q = has_q ? a.qual + tile_in_base : nullptr;
    uint8_t *out_b = a.out_bases + (c_first << 4), *out_q = has_q ? a.out_qual + (c_first << 4) : nullptr;
    const int lo_ok = tile_in_base > 0x3FFFFFFFull ? -0x3FFFFFFF : -(int)tile_in_base;
    const u64 after = a.total_bytes - tile_in_base;
    const int hi_ok = after > 0x3FFFFFFFull ? 0x3FFFFFFF : (int)after;
    u32 bad = 0;
    for (u32 ci = tid; ci < nchunks; ci += nthreads) {
        const int cs = (int)(ci << 4) - head;                          // tile-relative output offset of chunk byte 0
        const int lo_c = cs < 0 ? -cs : 0;
        const int rem = (int)S - cs;
        const int hi_c = rem >= 16 ? 16 : rem;
        const u32 o = (u32)(cs + lo_c), o_end = (u32)(cs + hi_c);
        // largest r with v_off[r] <= o   (v_off[0] = 0 <= o < S = v_off[nreads]); that read is never empty
        u32 lo = 0, hi = nreads;
        while (hi - lo > 1) {
            const u32 mid = (lo + hi) >> 1;
            if (v_off[mid] <= o) lo = mid; else hi = mid;
        }
        const u32 e1 = v_off[lo + 1];
        const u32 end1 = e1 < o_end ? e1 : o_end;
        const FxgSeg s1 = fxg_make_seg<REV>(v_off, v_src, lo, o, end1, cs);
        FxgSeg s2 = s1;
        bool two = false, more = false;
        u32 r_next = 0, o_next = 0;
        if (end1 < o_end) {                                            // the chunk continues in the next non-empty read
            u32 r = lo + 1;
            while (v_off[r + 1] == end1) ++r;
            const u32 e2 = v_off[r + 1];
            const u32 end2 = e2 < o_end ? e2 : o_end;
            s2 = fxg_make_seg<REV>(v_off, v_src, r, end1, end2, cs);
            two = true;
            more = end2 < o_end;
            r_next = r + 1; o_next = end2;
        }
        fxg_gather_array<REV, true, MASK>(src_b, lo_ok, hi_ok, out_b + (ci << 4), v_off, v_src, s1, s2, two, more, r_next, o_next, o_end, cs, lo_c, hi_c, &bad,
                                          a.qual + tile_in_base, Kmask, mask4);
        if (has_q) {
            u32 dummy = 0;
            fxg_gather_array<REV, false>(src_q, lo_ok, hi_ok, out_q + (ci << 4), v_off, v_src, s1, s2, two, more, r_next, o_next, o_end, cs, lo_c, hi_c, &dummy);
        }
    }
    return bad;   // nonzero: a byte outside ACGTN/acgtn reached the complement (REV only)
}
