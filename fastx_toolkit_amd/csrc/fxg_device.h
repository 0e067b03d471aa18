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
#if defined(FXG_ABLATION) || defined(FXG_DBG_BITS)      // (FXG_DBG_BITS: the switches alone, without the phase clocks and their barrier -- the product's own timing with a phase taken out)
#define FXG_DBG(a, bit) ((a).debug & (bit))
#else
#define FXG_DBG(a, bit) 0u
#endif

#define FXG_BLOCK 256           // threads per workgroup = 4 wave64
#define FXG_WAVES (FXG_BLOCK / 64)
#ifndef FXG_TBLOCK
#define FXG_TBLOCK 256          // threads per workgroup of the tile kernels
#endif
#define FXG_TWAVES (FXG_TBLOCK / 64)
#ifndef FXG_STORE_GRID
#define FXG_STORE_GRID 1            // fxg_rows_flush deals its 16-byte units to lanes by the unit's place in its 128-byte line (0: by the unit's place in the tile's output; the A/B arm).
                                    // The tile kernels' gather was measured the same way and lost 1.6-3 % (cfg4, cfg3, cfg5: profiles/r06/store_grid_other.txt): it keeps chunk ci on lane ci.
#endif
#ifndef FXG_CLIP_GATHER_K
#define FXG_CLIP_GATHER_K 4     // chunks per lane in flight in the clip instances' gather (FXG_GATHER_K for the streaming instances)
#endif
#ifndef FXG_CLIP_DEPTH
#define FXG_CLIP_DEPTH 4u       // slots of the clip instances at most (fxg_plan.h takes the deepest pipeline that does not cost a workgroup per CU: cfg3 four, cfg5 three)
#endif
#define FXG_CK_SLOTS 7u         // checkpoints a read can leave (fxg_plan.h picks the interval so that they suffice)
#ifndef FXG_CLIP_TBLOCK
#define FXG_CLIP_TBLOCK 256     // threads per workgroup of the two-pass clip instances (FxgTileBlock in fxg_kernels.h); 64 = one wave per workgroup, measured 6-25 % slower
#endif
#define FXG_MAX_TILE FXG_TBLOCK  // reads per tile (one thread decides one read)
#define FXG_TICKET_GROUPS 8      // a single device-scope counter saturates near 88 tickets/us; shard it (one per XCD)
#define FXG_TICKET_STRIDE 32     // u32 words between dispensers (128 B: one cache line each)
#define FXG_CTRL_WORDS 64         // u32 words of the control block in front of the dispensers: [0] error bits, [2..5] masker sums, [8] scanner role, [32..63] tallies

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
    u64 *agg;               // [ntiles]   tile totals   {tag:8 | kept reads:16 << 32 | kept bytes:32}, published by the tile's workgroup
    u64 *pfx;               // [2*ntiles] exclusive prefixes {tag:8 | value:56}: [2t] kept reads before tile t, [2t+1] kept bytes before it (scanner)
    u64 *bbase;             // [2*nbatch] fxg_scanner_multi: (kept reads, kept bytes) of every batch of tiles, tagged like agg
    u32  nscan;             // scanner waves of fxg_kernel_rows (1 = the single scanner of fxg_kernel_tiles)
    u32 *role;              // the workgroups that draw the first numbers here become the scanners
    u32  tag;               // launch epoch 1..255: granules of earlier launches are invalid without a memset
    u64 *tally;             // [FXG_NTALLY] the launch's -v report tallies (zeroed with the control block)
    u32 *ticket;            // dynamic tile dispensers, FXG_TICKET_STRIDE words apart (zeroed before every launch)
    u32  ticket_groups;     // number of dispensers (<= 8): dispenser g hands out tiles g, g+groups, g+2*groups, ...
    u32 *errflag;
    u32  compact;           // 1 = stream-compact kept reads into out_bases/out_qual
    u32  debug;             // FXG_DEBUG ablation bits (timing experiments only; results are wrong when set)
    u32  depth;             // clip instances: slots = tiles a workgroup keeps between decision and write-out (2 or 3)
    u32  clip_global;       // register two-pass clip instances: the DP reads the batch in global memory, no tile of bases is staged in LDS (fxg_plan.h)
    // folded tool parameters
    u32  stages;
    u32  tq;                // quality trimmer: byte >= tq  <=>  q >= -t      (0..128)
    u32  fq;                // quality filter : byte <  fq  <=>  q <  -q      (0..128)
    int  qt_min_len;
    int  qf_keep_pct;       // 100 - p
    u32  qf_drop_all;       // quirk F2: -p omitted and -q > 93
    int  alen;
    u32  adapter_has_n;     // the adapter contains 'N' (the clip kernels then keep the per-column neutral-match selects)
    u32  clip_min_len;
    int  clip_keep_delta;
    int  clip_min_adapter_len;
    u32  clip_flags;
    int  ft_first, ft_last;
    u32  ft_trim_end, ft_min_len;
    u32  mask_char;         // fastq_masker -r; the mask threshold shares `fq` (byte < fq is masked)
    u32  nf_keep_n;         // fastq_to_fasta -n
    u32  rev_dw;            // reverse complement: no source window of the launch starts on a dword boundary -> the instance with dword-aligned window loads (fxg_plan.h, fxg_ld16_dw)
    u64 *extra;             // [0] masked reads, [1] masked nucleotides (fastq_masker report)
    // what the clipper's DP reads: the batch itself, or (clip history, fxg_history.h) the queries extended by the stale tail
    const uint8_t  *clip_src;
    u64  clip_total;        // n * clip_stride
    u32  clip_stride;
    const uint16_t *wlen;   // DP rows per read (null: the read's own length)
    float *clip_ck;         // two-pass clipper for 17..99 adapter columns (fxg_clip_two_pass_k): score-row checkpoints, FXG_CK_SLOTS x bucket x threads floats per workgroup (null: one pass)
    u32  clip_ck_rows;      // a checkpoint every this many rows
    u32  clip_ptab_rows;    // rows of the workgroup's pair table (fxg_kernels.h: fxg_ptab_rows), 0 = the instance has none
    u32  clip_ptab_stride;  // bytes between two of its rows (an odd multiple of 16 where rows are wider than a bank sweep: fxg_ptab_stride)
    u32  clip_ptab_cols;    // columns per row (the instance's bucket rounded up to a multiple of four)
    u32  clip_ptab_dia1;    // what a non-neutral diagonal step adds to the path summary of the instance's form (FXG_PK_DIA1 / FXG_K_DIA1)
#ifdef FXG_CLIP_DEBUG
    u32 *clip_dbg;          // debug builds only (scripts/debug/clip64_bisect.py): 16 words per read of fxg_clip_two_pass_k's intermediate state
#endif
    char adapter[100];
    uint8_t clip_ptab_row[256];   // pair table: row of every byte value (fxg_plan.h fills it: 0 = not in the adapter, 1 = 'N', 2.. = the adapter's distinct bytes in order of first appearance); bit 7 = this byte's thread writes the row
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
#ifndef FXG_HOST_EMULATION
    // the flags are bytes of 0x80: a dot product with the byte weights (1,2,4,8) / (16,32,64,128) gathers eight of them at a time
    // (v_dot4_u32_u8 accumulates), 128 x the mask
    const u32 lo = __builtin_amdgcn_udot4(fxg_ge_flags(v.x, K), 0x08040201u, __builtin_amdgcn_udot4(fxg_ge_flags(v.y, K), 0x80402010u, 0u, false), false);
    const u32 hi = __builtin_amdgcn_udot4(fxg_ge_flags(v.z, K), 0x08040201u, __builtin_amdgcn_udot4(fxg_ge_flags(v.w, K), 0x80402010u, 0u, false), false);
    return (lo >> 7) | (hi << 1);
#endif
    return fxg_pack4(fxg_ge_flags(v.x, K)) | (fxg_pack4(fxg_ge_flags(v.y, K)) << 4) |
           (fxg_pack4(fxg_ge_flags(v.z, K)) << 8) | (fxg_pack4(fxg_ge_flags(v.w, K)) << 12);
}

// the bit-7 flags of sixteen bytes (four dwords whose bytes are 0x80 or 0) as a 16-bit mask, bit i = byte i
FXG_HD u32 fxg_flags16(u32 f0, u32 f1, u32 f2, u32 f3)
{
#ifndef FXG_HOST_EMULATION
    const u32 lo = __builtin_amdgcn_udot4(f0, 0x08040201u, __builtin_amdgcn_udot4(f1, 0x80402010u, 0u, false), false);      // (as fxg_mask16: 128 x the mask)
    const u32 hi = __builtin_amdgcn_udot4(f2, 0x08040201u, __builtin_amdgcn_udot4(f3, 0x80402010u, 0u, false), false);
    return (lo >> 7) | (hi << 1);
#endif
    return fxg_pack4(f0) | (fxg_pack4(f1) << 4) | (fxg_pack4(f2) << 8) | (fxg_pack4(f3) << 12);
}
// bit 7 of every byte of x that is not zero
FXG_HD u32 fxg_nonzero_flags(u32 x) { return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; }

// 16-byte load from an arbitrarily aligned address (one global_load_dwordx4 on gfx950)
FXG_HD u32x4 fxg_ld16(const uint8_t *p)
{
    u32x4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
// Streaming forms for data touched exactly once by the gather.  Measured on cfg2 / cfg4: `nt` on the packed-output stores
// is worth 4 % / -2 %; `nt` on the source windows costs 4 % / 7 % (neighbouring lanes' unaligned windows share lines), so
// the loads stay plain unless FXG_V_NTL is defined.
typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));
FXG_HD u32x4 fxg_ld16_stream(const uint8_t *p)
{
#if defined(FXG_V_NTL) && !defined(FXG_HOST_EMULATION)
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4_unaligned *>(p));
#else
    return fxg_ld16(p);
#endif
}
FXG_HD void fxg_st16_stream(uint8_t *p, u32x4 v)
{
#if !defined(FXG_V_NO_NTS) && !defined(FXG_HOST_EMULATION)
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
#else
    *reinterpret_cast<u32x4 *>(p) = v;
#endif
}

// The 16 bytes at p through DWORD-ALIGNED loads: four words and the one behind them, realigned with a funnel shift.  For the reverse complement's source
// windows: a 16-byte load that starts off a dword boundary is served at a lower rate, which shows once EVERY window of a launch does -- the reversed windows of
// 150-byte rows all start at 2 mod 4: 6.2 ms against 5.0-5.4 on rows where at least every other read's windows are dword aligned; the forward kernels never come
// to more than three windows in four (profiles/r06/gather_alignment*.txt).  It costs a load and four funnel shifts per window: -8 % where no window is aligned,
// +3..8 % everywhere else, and nothing gained as a run-time switch inside one kernel (rev_dword_loads_*.txt) -- hence an instance of its own,
// fxg_kernel_tiles<0, 5>, that the plan picks for exactly those launches (FxgKArgs::rev_dw).  Reads up to 3 bytes before and 4 bytes behind the window.
FXG_HD u32x4 fxg_ld16_dw(const uint8_t *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
    const u32 sh = (u32)(reinterpret_cast<uintptr_t>(p) & 3u);
    const uint8_t *q = p - sh;
    const u32x4 v = *reinterpret_cast<const u32x4_a4 *>(q);
    const u32 t = *reinterpret_cast<const u32 *>(q + 16);
    return (u32x4){__builtin_amdgcn_alignbyte(v.y, v.x, sh), __builtin_amdgcn_alignbyte(v.z, v.y, sh), __builtin_amdgcn_alignbyte(v.w, v.z, sh), __builtin_amdgcn_alignbyte(t, v.w, sh)};
#else
    return fxg_ld16(p);
#endif
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
// inter-workgroup granules.  One naturally aligned u64 = {tag:8, value:56}, written by ONE relaxed agent-scope store
// (sc1) and polled with relaxed agent-scope loads: the data is its own flag, so no fence is needed and nothing depends on
// workgroup placement.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fxg_granule_store_raw(u64 *g, u64 v)
{
    __hip_atomic_store(g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 fxg_granule_load(u64 *g)
{
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------
// Central scanner.  The tiles' (kept reads, kept bytes) totals must be turned into exclusive prefixes in
// tile order.  Instead of every tile walking back over its predecessors (1 000+ tiles are in flight, so a walk is dozens of
// dependent memory round trips), ONE wave does nothing else: it polls the totals in tile order, 64 * FXG_SCAN_K granules per
// round trip, prefix-sums whatever contiguous run has been published and writes the prefixes back; a tile then needs ONE load
// of its own prefix.  (Round 1 used decoupled look-back: 4.31 ms against 4.13 ms for cfg2, and 4.94 against 4.15 with 128-read
// tiles, profiles/r02/variants_a.txt.)  The scanner is whichever workgroup draws 0 from a role counter at kernel start, so it is running by
// construction; it only ever waits for totals of tiles whose tickets were drawn (their owners are running and publish before
// they wait for anything), and a tile only waits for the scanner: progress is independent of residency, dispatch order and
// placement.  Granules carry the launch's epoch tag, are written by ONE relaxed agent-scope store and polled with relaxed
// agent-scope loads (the data is its own flag: no fences).
// ------------------------------------------------------------------------------------------------
#ifndef FXG_SCAN_K
#define FXG_SCAN_K 8      // 512 tiles per round trip; the batch lives in registers of every instance of the tile kernel (16 cost the streaming kernels a wave per SIMD)
#endif
#define FXG_TAG_SHIFT 56
#define FXG_TAG_VALUE(x) ((x) & ((1ull << FXG_TAG_SHIFT) - 1ull))

__device__ __forceinline__ void fxg_publish_total(const FxgKArgs &a, u32 tile, u32 cnt, u32 bytes)
{
    fxg_granule_store_raw(a.agg + tile, ((u64)a.tag << FXG_TAG_SHIFT) | ((u64)cnt << 32) | (u64)bytes);
}

// bounded spin shared by the scanner and the tiles: 2 s without progress, or somebody else already gave up
__device__ __forceinline__ bool fxg_spin_expired(const FxgKArgs &a, u64 t0)
{
    const bool late = __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull;   // 100 MHz
    const u32 flagged = __hip_atomic_load(a.errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & FXG_DEV_ERR_SCAN_TIMEOUT;
    if (late || flagged) { atomicOr(a.errflag, FXG_DEV_ERR_SCAN_TIMEOUT); return true; }
    return false;
}

// inclusive prefix sum over the wave's 64 lanes without LDS or address registers: four row shifts, two row broadcasts (DPP)
__device__ __forceinline__ u32 fxg_wave_scan_dpp(u32 x)
{
    u32 v = x;
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8  -> inclusive within rows of 16
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// executed by ONE wave; K = granules per lane and round trip
template <int K>
__device__ __forceinline__ void fxg_scanner_k(const FxgKArgs &a)
{
    const u32 lane = fxg_lane();
    const u64 tagw = (u64)a.tag << FXG_TAG_SHIFT;
    __builtin_amdgcn_s_setprio(3);                          // every tile waits for this wave: do not queue it behind its CU's streaming waves
    u64 run_c = 0, run_b = 0;                               // exclusive prefix of tile t0
    u32 t0 = 0, spins = 0;
    u64 tlast = __builtin_amdgcn_s_memrealtime();
#ifdef FXG_ABLATION
    u64 sc_rounds = 0, sc_load = 0, sc_comp = 0, sc_t = __builtin_amdgcn_s_memrealtime();
#endif
    while (t0 < a.ntiles) {
        u64 v[K];
#pragma unroll
        for (u32 k = 0; k < (u32)K; ++k) {                  // all loads of the round are in flight together
            const u64 idx = (u64)t0 + k * 64u + lane;
            v[k] = idx < a.ntiles ? fxg_granule_load(a.agg + idx) : 0ull;
        }
#ifdef FXG_ABLATION
        { u64 sink = 0; for (u32 k = 0; k < (u32)K; ++k) sink |= v[k]; asm volatile("" :: "v"(sink)); const u64 n_ = __builtin_amdgcn_s_memrealtime(); sc_load += n_ - sc_t; sc_t = n_; ++sc_rounds; }
#endif
        u32 adv = 0;
        bool open = true;
#pragma unroll
        for (u32 k = 0; k < (u32)K; ++k) {
            if (!open) continue;                            // wave-uniform
            const u64 idx = (u64)t0 + k * 64u + lane;
            const bool valid = idx < a.ntiles && (u32)(v[k] >> FXG_TAG_SHIFT) == a.tag;
            const u64 bal = __ballot(valid);
            const u32 m = bal == ~0ull ? 64u : (u32)__builtin_ctzll(~bal);      // leading run of published totals
            const u32 c = lane < m ? (u32)(v[k] >> 32) & 0xFFFFu : 0u, b = lane < m ? (u32)v[k] : 0u;
            // 64 tiles: at most 64 * 256 reads, 64 * 2^24 bytes.  DPP scans: the shuffle form (12 dependent ds_bpermute per 64 tiles,
            // ~13 000 cycles per 1024 tiles) capped the whole kernel near 140 tiles/us (profiles/r02/k_ablate.txt)
            const u32 ic = fxg_wave_scan_dpp(c), ib = fxg_wave_scan_dpp(b);
            if (lane < m) {
                fxg_granule_store_raw(a.pfx + 2 * idx, tagw | (run_c + (ic - c)));
                fxg_granule_store_raw(a.pfx + 2 * idx + 1, tagw | (run_b + (ib - b)));
            }
            run_c += (u32)__builtin_amdgcn_readlane((int)ic, 63); run_b += (u32)__builtin_amdgcn_readlane((int)ib, 63);
            adv += m;
            open = (m == 64u);
        }
        t0 += adv;
#ifdef FXG_ABLATION
        { const u64 n_ = __builtin_amdgcn_s_memrealtime(); sc_comp += n_ - sc_t; sc_t = n_; }
#endif
        if (adv) { spins = 0; tlast = __builtin_amdgcn_s_memrealtime(); continue; }
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 255u) == 0u && fxg_spin_expired(a, tlast)) return;   // never hang the GPU
    }
#ifdef FXG_ABLATION
    if (lane == 0) { u64 *d = reinterpret_cast<u64 *>(a.errflag + 10) + 8; d[0] = sc_rounds; d[1] = sc_load; d[2] = sc_comp; }
#endif
}

// Several scanner waves (fxg_kernel_rows: 64-read tiles arrive four times as fast, and ONE wave's round trip -- 16 loads of 64
// granules each behind its CU's streaming traffic, 6-7 us, then 2.3 us of scans -- capped the kernel near 110 tiles/us,
// profiles/r02/v_ablate.txt).  Wave j of S takes the batches j, j + S, ... of 64 K tiles: it waits until the whole batch has
// published, scans it locally and publishes the batch's TOTAL at once (bbase[2b], bbase[2b+1]).  The prefix at the start of its
// batch b is then its own running prefix at the end of batch b - S plus the totals of the S - 1 batches in between, which the other
// waves publish as soon as THEIR batches are complete: no wave waits for another wave's prefix, so there is no serial chain from
// batch to batch -- only "every earlier tile has published", which a prefix needs anyway.  Every scanner wave is running by
// construction (roles are drawn at kernel start).  Waiting for a WHOLE batch assumes enough resident workers to decide it (two tiles
// each); when the batch stalls the wave falls back to publishing run by run (below), so progress does not depend on residency.
template <int K>
__device__ __forceinline__ void fxg_scanner_multi(const FxgKArgs &a, u32 j)
{
    const u32 lane = fxg_lane();
    const u64 tagw = (u64)a.tag << FXG_TAG_SHIFT;
    constexpr u32 BATCH = (u32)K * 64u;
    const u32 nb = (a.ntiles + BATCH - 1u) / BATCH, S = a.nscan;
    __builtin_amdgcn_s_setprio(3);
    u64 end_c = 0, end_b = 0;                                   // this wave's prefix at the END of its previous batch (b - S)
    for (u32 b = j; b < nb; b += S) {
        const u32 t0 = b * BATCH;
        u64 v[K];
        u32 spins = 0;
        const u64 tstart = __builtin_amdgcn_s_memrealtime();
        for (;;) {                                              // until every tile of the batch has published its totals
            u64 bad = 0;
#pragma unroll
            for (u32 k = 0; k < (u32)K; ++k) {
                const u32 idx = t0 + k * 64u + lane;
                v[k] = idx < a.ntiles ? fxg_granule_load(a.agg + idx) : tagw;          // past the last tile: published, nothing kept
            }
#pragma unroll
            for (u32 k = 0; k < (u32)K; ++k) bad |= __ballot((u32)(v[k] >> FXG_TAG_SHIFT) != a.tag);
            if (!bad) break;
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            // The batch does not fill up.  A worker holds at most two decided tiles and draws its next ticket only after the earlier
            // one is written out, so with few workers RESIDENT (a shared GPU, a CU mask, FXG_BLOCKS_PER_CU=1) the tiles that would
            // complete this batch may never be drawn while everybody waits for its prefixes.  Every ~64 polls the wave therefore
            // publishes the prefixes of the batch's leading published run (what fxg_scanner_k does every round): the workers waiting
            // for those move on and draw the missing tiles.  The values are the final ones, published again when the batch is whole.
            if ((spins & 63u) == 0u) {
                const u32 first = b >= S ? b - S + 1u : 0u, cnt = b - first;
                u64 wc = tagw, wb = tagw;
                if (lane < cnt) { wc = fxg_granule_load(a.bbase + 2 * (u64)(first + lane)); wb = fxg_granule_load(a.bbase + 2 * (u64)(first + lane) + 1); }
                if (__ballot((u32)(wc >> FXG_TAG_SHIFT) != a.tag || (u32)(wb >> FXG_TAG_SHIFT) != a.tag) == 0ull) {      // every earlier batch is whole
                    const u32 sc = fxg_wave_scan_dpp(lane < cnt ? (u32)FXG_TAG_VALUE(wc) : 0u), sb = fxg_wave_scan_dpp(lane < cnt ? (u32)FXG_TAG_VALUE(wb) : 0u);
                    u64 run_c = end_c + (u32)__builtin_amdgcn_readlane((int)sc, 63), run_b = end_b + (u32)__builtin_amdgcn_readlane((int)sb, 63);
                    bool open = true;
#pragma unroll
                    for (u32 k = 0; k < (u32)K; ++k) {
                        if (!open) continue;                    // wave-uniform
                        const u64 idx = (u64)t0 + k * 64u + lane;
                        const u64 bal = __ballot((u32)(v[k] >> FXG_TAG_SHIFT) == a.tag);
                        const u32 m = bal == ~0ull ? 64u : (u32)__builtin_ctzll(~bal);
                        const u32 c = lane < m ? (u32)(v[k] >> 32) & 0xFFFFu : 0u, bb = lane < m ? (u32)v[k] : 0u;
                        const u32 ic = fxg_wave_scan_dpp(c), ib = fxg_wave_scan_dpp(bb);
                        if (lane < m && idx < a.ntiles) {
                            fxg_granule_store_raw(a.pfx + 2 * idx, tagw | (run_c + (ic - c)));
                            fxg_granule_store_raw(a.pfx + 2 * idx + 1, tagw | (run_b + (ib - bb)));
                        }
                        run_c += (u32)__builtin_amdgcn_readlane((int)ic, 63); run_b += (u32)__builtin_amdgcn_readlane((int)ib, 63);
                        open = (m == 64u);
                    }
                }
            }
            if ((spins & 255u) == 0u && fxg_spin_expired(a, tstart)) return;
        }
        u32 exc[K], exb[K], rc = 0, rb = 0;                     // exclusive prefixes inside the batch; 64 K tiles of < 2^20 bytes each
#pragma unroll
        for (u32 k = 0; k < (u32)K; ++k) {
            const u32 c = (u32)(v[k] >> 32) & 0xFFFFu, bb = (u32)v[k];
            const u32 ic = fxg_wave_scan_dpp(c), ib = fxg_wave_scan_dpp(bb);
            exc[k] = rc + ic - c; exb[k] = rb + ib - bb;
            rc += (u32)__builtin_amdgcn_readlane((int)ic, 63); rb += (u32)__builtin_amdgcn_readlane((int)ib, 63);
        }
        if (lane < 2u) fxg_granule_store_raw(a.bbase + 2 * (u64)b + lane, tagw | (u64)(lane == 0u ? rc : rb));   // the batch's total: the other waves need it
        // the totals of the batches between this wave's previous batch and this one (the first S batches: of all earlier ones)
        const u32 first = b >= S ? b - S + 1u : 0u, cnt = b - first;
        u64 wc = tagw, wb = tagw;
        spins = 0;
        const u64 t1 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            if (lane < cnt) { wc = fxg_granule_load(a.bbase + 2 * (u64)(first + lane)); wb = fxg_granule_load(a.bbase + 2 * (u64)(first + lane) + 1); }
            if (__ballot((u32)(wc >> FXG_TAG_SHIFT) != a.tag || (u32)(wb >> FXG_TAG_SHIFT) != a.tag) == 0ull) break;
            if ((++spins & 255u) == 0u && fxg_spin_expired(a, t1)) return;
        }
        const u32 sc = fxg_wave_scan_dpp(lane < cnt ? (u32)FXG_TAG_VALUE(wc) : 0u), sb = fxg_wave_scan_dpp(lane < cnt ? (u32)FXG_TAG_VALUE(wb) : 0u);
        const u64 base_c = end_c + (u32)__builtin_amdgcn_readlane((int)sc, 63), base_b = end_b + (u32)__builtin_amdgcn_readlane((int)sb, 63);
#pragma unroll
        for (u32 k = 0; k < (u32)K; ++k) {
            const u64 idx = (u64)t0 + k * 64u + lane;
            if (idx < a.ntiles) {
                fxg_granule_store_raw(a.pfx + 2 * idx, tagw | (base_c + exc[k]));
                fxg_granule_store_raw(a.pfx + 2 * idx + 1, tagw | (base_b + exb[k]));
            }
        }
        end_c = base_c + rc; end_b = base_b + rb;
    }
}
__device__ __forceinline__ void fxg_scanner(const FxgKArgs &a) { fxg_scanner_k<FXG_SCAN_K>(a); }

// executed by wave 0 of a tile's workgroup: lanes 0 / 1 fetch the tile's (reads, bytes) prefix; `peek` is an earlier load of it
__device__ __forceinline__ u64 fxg_peek_prefix(const FxgKArgs &a, u32 tile)
{
    const u32 lane = fxg_lane();
    return lane < 2u ? fxg_granule_load(a.pfx + 2 * (u64)tile + lane) : ((u64)a.tag << FXG_TAG_SHIFT);
}
__device__ __forceinline__ void fxg_wait_prefix(const FxgKArgs &a, u32 tile, u64 peek, u64 *bc)
{
    const u32 lane = fxg_lane();
    u64 v = peek;
    u32 spins = 0;
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    while (__ballot((u32)(v >> FXG_TAG_SHIFT) != a.tag) != 0ull) {
        __builtin_amdgcn_s_sleep(1);
        if (lane < 2u) v = fxg_granule_load(a.pfx + 2 * (u64)tile + lane);
        if ((++spins & 255u) == 0u && fxg_spin_expired(a, t0)) break;
    }
    // expired: the granule belongs to an EARLIER launch (the arrays are not cleared between launches) -- its value must not be used
    // as an offset.  ~0 tells the caller to write nothing for this tile; the host sees FXG_DEV_ERR_SCAN_TIMEOUT.
    const bool ok = __ballot((u32)(v >> FXG_TAG_SHIFT) != a.tag) == 0ull;
    if (lane < 2u) bc[lane] = ok ? FXG_TAG_VALUE(v) : ~0ull;
}

// the same for a workgroup that IS one wave (fxg_rows.h): the prefix comes back in registers
#ifndef FXG_POLL_SLEEP
#define FXG_POLL_SLEEP 1
#endif
__device__ __forceinline__ void fxg_wait_prefix_wave(const FxgKArgs &a, u32 tile, u64 *bc)
{
    const u32 lane = fxg_lane();
    u64 v = fxg_peek_prefix(a, tile);
    u32 spins = 0;
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    while (__ballot((u32)(v >> FXG_TAG_SHIFT) != a.tag) != 0ull) {
        __builtin_amdgcn_s_sleep(FXG_POLL_SLEEP);
        if (lane < 2u) v = fxg_granule_load(a.pfx + 2 * (u64)tile + lane);
        if ((++spins & 255u) == 0u && fxg_spin_expired(a, t0)) break;
    }
    const bool ok = __ballot((u32)(v >> FXG_TAG_SHIFT) != a.tag) == 0ull;      // expired: a stale granule of an earlier launch, see fxg_wait_prefix
    v = ok ? FXG_TAG_VALUE(v) : ~0ull;                      // scalars from here on
    bc[0] = ((u64)(u32)__builtin_amdgcn_readlane((int)(v >> 32), 0) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)v, 0);
    bc[1] = ((u64)(u32)__builtin_amdgcn_readlane((int)(v >> 32), 1) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)v, 1);
}

// ------------------------------------------------------------------------------------------------
// workgroup exclusive scan of (keep, out_len) over TW waves.  scratch: u32[2*TW]
// ------------------------------------------------------------------------------------------------
template <int TW = FXG_TWAVES>
__device__ __forceinline__ void fxg_block_scan2(u32 c, u32 b, u32 *scratch, u32 *ex_c, u32 *ex_b, u32 *tot_c, u32 *tot_b)
{
    const u32 lane = fxg_lane(), wave = threadIdx.x >> 6;
    const u32 ic = fxg_wave_scan_dpp(c), ib = fxg_wave_scan_dpp(b);
    if constexpr (TW == 1) {                                 // the workgroup is one wave: totals by readlane, no LDS, no barrier
        *ex_c = ic - c; *ex_b = ib - b;
        *tot_c = (u32)__builtin_amdgcn_readlane((int)ic, 63); *tot_b = (u32)__builtin_amdgcn_readlane((int)ib, 63);
        return;
    }
    if (lane == 63) { scratch[wave] = ic; scratch[TW + wave] = ib; }
    __syncthreads();
    u32 oc = 0, ob = 0, sc = 0, sb = 0;
#pragma unroll
    for (int w = 0; w < TW; ++w) {
        const u32 wc = scratch[w], wb = scratch[TW + w];
        if (w < (int)wave) { oc += wc; ob += wb; }
        sc += wc; sb += wb;
    }
    *ex_c = oc + ic - c; *ex_b = ob + ib - b; *tot_c = sc; *tot_b = sb;
}

// ------------------------------------------------------------------------------------------------
// order-preserving gather of one tile's kept reads into the packed output.
// Stage A leaves the tile's KEPT reads, in input order, in LDS (index = rank of the read among the tile's kept reads):
//   k_off[0..nk] : exclusive prefix of kept lengths (tile-relative output offset of each kept read; k_off[nk] = S)
//   k_src[k]     : tile-relative source byte of output byte 0 of kept read k
//   k_tab[g]     : rank of the read that owns tile-relative output byte 16*g (g < ceil(S/16))
//   REV          : output byte j of a read comes from source byte k_src[k] - j, complemented (bases)
// Work item = one 16-byte aligned chunk of the GLOBAL output; consecutive lanes write consecutive chunks (1 KiB per
// wave store).  A chunk takes one unaligned 16-byte source window per array, two when it straddles two reads; the
// (rare) bytes of a third and later read inside one chunk, the partial chunks at either end of the tile (shared with
// the neighbouring tiles) and the first/last tile of the batch (where a window could leave the arrays) go byte by byte.
// Everything per lane is 32-bit and tile-relative; the 64-bit bases are wave-uniform and stay in SGPRs.
// ------------------------------------------------------------------------------------------------
FXG_HD void fxg_tab_fill(uint16_t *k_tab, u32 rank, u32 off, u32 len)
{
    for (u32 g = (off + 15u) >> 4; (g << 4) < off + len; ++g) k_tab[g] = (uint16_t)rank;
}

// largest k with k_off[k] <= o   (k_off[0] = 0 <= o < S = k_off[nk]); that read is never empty
FXG_HD u32 fxg_rank_of(const u32 *k_off, const uint16_t *k_tab, u32 nk, u32 S, u32 o)
{
    u32 lo = 0, hi = nk;
    if (k_tab) {                                            // o lies between the owners of its granule's first byte and of the next granule's
        const u32 g = o >> 4;
        lo = k_tab[g];
        hi = ((g + 1u) << 4) < S ? (u32)k_tab[g + 1u] + 1u : nk;
    }
#ifndef FXG_NO_RANK_GUESS
    else if (nk > 8u) {
        // No granule table (the clip instances: their LDS holds a tile of bases): start from where the byte would be if all kept reads had the tile's mean
        // length and bracket it with steps of 1, 2, 4, ... -- kept lengths are close to one another, so the bracket is a few reads wide instead of the whole
        // tile and the bisection below takes 1-3 dependent LDS reads instead of 8 (round 6).
        u32 g = (u32)((float)o * ((float)nk / (float)S));
        g = g < nk ? g : nk - 1u;
        if (k_off[g] <= o) {
            lo = g;
            u32 step = 1u;
            hi = g + 1u;
            while (hi < nk && k_off[hi] <= o) { lo = hi; step <<= 1; hi = hi + step < nk ? hi + step : nk; }
        } else {
            hi = g;
            u32 step = 1u;
            lo = g - 1u;                                    // (k_off[0] = 0 <= o, so g >= 1 here)
            while (k_off[lo] > o) { hi = lo; step <<= 1; lo = lo > step ? lo - step : 0u; }
        }
    }
#endif
    while (hi - lo > 1u) {
        const u32 mid = (lo + hi) >> 1;
        if (k_off[mid] <= o) lo = mid; else hi = mid;
    }
    return lo;
}

// v_perm_b32: byte i of the result is byte sel[i] of the 8-byte value {hi:lo} (selectors 0..3 -> lo, 4..7 -> hi)
FXG_HD u32 fxg_perm(u32 hi, u32 lo, u32 sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const u64 v = ((u64)hi << 32) | lo;
    u32 r = 0;
    for (int i = 0; i < 4; ++i) r |= (u32)((v >> (8u * ((sel >> (8 * i)) & 7u))) & 0xFFu) << (8 * i);
    return r;
#endif
}

// v_alignbyte_b32: the 4 bytes of {hi:lo} that start at byte sh (0..3)
FXG_HD u32 fxg_alignbyte(u32 hi, u32 lo, u32 sh)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
#else
    return (u32)(((((u64)hi) << 32) | lo) >> (8u * (sh & 3u)));
#endif
}
// x in [0, 4]: low x bytes set
FXG_HD u32 fxg_lowbytes32(int x) { return x >= 4 ? 0xFFFFFFFFu : ((1u << (8 * x)) - 1u); }
// e in [0, 16]: low e bytes of a 16-byte value set
FXG_HD u32x4 fxg_lowmask16(int e)
{
    const int e0 = e < 4 ? e : 4, e1 = e < 4 ? 0 : (e < 8 ? e - 4 : 4), e2 = e < 8 ? 0 : (e < 12 ? e - 8 : 4), e3 = e < 12 ? 0 : e - 12;
    return (u32x4){fxg_lowbytes32(e0), fxg_lowbytes32(e1), fxg_lowbytes32(e2), fxg_lowbytes32(e3)};
}
FXG_HD u32x4 fxg_select16(u32x4 m, u32x4 x, u32x4 y) { return (x & m) | (y & ~m); }

// one output byte: tile-relative output offset o of kept read `anchor` at position j
template <bool REV, bool MASK>
FXG_HD void fxg_gather_byte(const FxgKArgs &a, const uint8_t *src_b, const uint8_t *src_q, uint8_t *dst_b, uint8_t *dst_q,
                            u32 anchor, u32 j, u32 o, u32 *bad)
{
    const u32 s = REV ? anchor - j : anchor + j;
    u32 b = src_b[s];
    if (MASK) { if ((u32)src_q[s] < a.fq) b = a.mask_char & 0xFFu; }      // fastq_masker.c:94-99
    if (REV) { *bad |= fxg_invalid_bases4(b, 0xFFu); b = fxg_complement4(b) & 0xFFu; }
    dst_b[o] = (uint8_t)b;
    if (dst_q) dst_q[o] = src_q[s];
}

// One 16-byte output chunk in flight: bytes [0, e) come from kept read k (window wb/wq, first forward byte at chunk byte 0),
// bytes [e, e2) from read k + 1 (window vb/vq), bytes [e2, 16) from later reads.  e = 0 marks an unused slot.
#ifndef FXG_GATHER_K
#define FXG_GATHER_K 1   // measured: 1 and 2 tie on cfg2 at 5 workgroups/CU (2..4 only win when fewer are resident); cfg4 prefers 1 (7 workgroups/CU)
#endif
struct FxgChunk { u32 o, k; int e, e2; u32x4 wb, wq, vb, vq; };

template <bool REV, bool MASK, bool DW = false>      // DW (REV): windows through fxg_ld16_dw
FXG_HD void fxg_chunk_load(FxgChunk &c, const uint8_t *src_b, const uint8_t *src_q, bool want_q, const u32 *k_off, const u32 *k_src,
                           const uint16_t *k_tab, u32 nk, u32 S, u32 o)
{
    const u32 k = fxg_rank_of(k_off, k_tab, nk, S, o);
    const u32 e1 = k_off[k + 1u];
    const int j0 = (int)(o - k_off[k]);
    const int e = e1 - o < 16u ? (int)(e1 - o) : 16;
    const int p1 = REV ? (int)k_src[k] - j0 - 15 : (int)k_src[k] + j0;
    c.o = o; c.k = k; c.e = e; c.e2 = 16;
    c.wb = DW ? fxg_ld16_dw(src_b + p1) : fxg_ld16_stream(src_b + p1);
    c.wq = (u32x4){0u, 0u, 0u, 0u};
    if (want_q) c.wq = DW ? fxg_ld16_dw(src_q + p1) : fxg_ld16_stream(src_q + p1);
    c.vb = c.vq = (u32x4){0u, 0u, 0u, 0u};
    if (e < 16) {                                                         // the chunk continues in the next kept read
        const u32 n2 = k_off[k + 2u] - e1;
        c.e2 = n2 < (u32)(16 - e) ? e + (int)n2 : 16;
        const int p2 = REV ? (int)k_src[k + 1u] + e - 15 : (int)k_src[k + 1u] - e;
        c.vb = DW ? fxg_ld16_dw(src_b + p2) : fxg_ld16_stream(src_b + p2);
        if (want_q) c.vq = DW ? fxg_ld16_dw(src_q + p2) : fxg_ld16_stream(src_q + p2);
    }
}

template <bool REV, bool MASK = false, int GK = FXG_GATHER_K, bool DW = false>
FXG_HD u32 fxg_tile_gather(const FxgKArgs &a, const u32 *k_off, const u32 *k_src, const uint16_t *k_tab, u32 nk,
                           u64 tile_in_base, u32 tile_bytes, u64 B, u32 S, u32 tid, u32 nthreads)
{
    if (S == 0) return 0u;
    const bool has_q = a.qual != nullptr && a.out_qual != nullptr && !FXG_DBG(a, 4u);
    // wave-uniform 64-bit quantities; tile-relative output byte o lives at dst[o], tile-relative source byte s at src[s]
    const uint8_t *src_b = a.bases + tile_in_base, *src_q = (has_q || MASK) ? a.qual + tile_in_base : nullptr;
    uint8_t *dst_b = a.out_bases + B, *dst_q = has_q ? a.out_qual + B : nullptr;
    // a window covers up to 15 bytes either side of the rows it serves: whole windows only where that stays inside the arrays
    const bool interior = tile_in_base >= (DW ? 18u : 15u) && tile_in_base + tile_bytes + (DW ? 19u : 15u) <= a.total_bytes;      // (fxg_ld16_dw reads 3 bytes before and 4 behind its window)
    u32 o_lo = 0, nfull = 0;
    if (interior) {
        const u32 head = (16u - (u32)(B & 15u)) & 15u;                   // output bytes before the first aligned chunk
        o_lo = head < S ? head : S;
        nfull = (S - o_lo) >> 4;
    }
    const u32 o_hi = o_lo + (nfull << 4);
    const u32 Kmask = (128u - a.fq) * 0x01010101u, mask4 = (a.mask_char & 0xFFu) * 0x01010101u;
    u32 bad = 0;

    // GK chunks per lane and trip: the source windows of all of them are requested before any is consumed
    for (u32 c0 = tid; c0 < nfull; c0 += nthreads * GK) {
        FxgChunk ch[GK];
#pragma unroll
        for (int u = 0; u < GK; ++u) {
            const u32 ci = c0 + (u32)u * nthreads;
            ch[u].e = 0;
            if (ci < nfull) fxg_chunk_load<REV, MASK, DW>(ch[u], src_b, src_q, has_q || MASK, k_off, k_src, k_tab, nk, S, o_lo + (ci << 4));
        }
#pragma unroll
        for (int u = 0; u < GK; ++u) {
            if (ch[u].e == 0) continue;
            FxgChunk &c = ch[u];
            if (REV) { c.wb = fxg_reverse16(c.wb); c.wq = fxg_reverse16(c.wq); }
            if (c.e < 16) {
                if (REV) { c.vb = fxg_reverse16(c.vb); c.vq = fxg_reverse16(c.vq); }
                const u32x4 m = fxg_lowmask16(c.e);
                c.wb = fxg_select16(m, c.wb, c.vb);
                c.wq = fxg_select16(m, c.wq, c.vq);
            }
            if (MASK) {                                                   // fastq_masker.c:94-99: base := mask character where quality < threshold
                u32 *pb = reinterpret_cast<u32 *>(&c.wb);
                const u32 *pq = reinterpret_cast<const u32 *>(&c.wq);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32 sel = ((fxg_ge_flags(pq[i], Kmask) ^ 0x80808080u) >> 7) * 0xFFu;   // 0xFF per byte with quality below the threshold
                    pb[i] = (pb[i] & ~sel) | (mask4 & sel);
                }
            }
            if (REV) {
                const u32x4 mv = fxg_lowmask16(c.e2);
                bad |= fxg_invalid_bases4(c.wb.x, mv.x) | fxg_invalid_bases4(c.wb.y, mv.y) | fxg_invalid_bases4(c.wb.z, mv.z) | fxg_invalid_bases4(c.wb.w, mv.w);
                c.wb.x = fxg_complement4(c.wb.x); c.wb.y = fxg_complement4(c.wb.y); c.wb.z = fxg_complement4(c.wb.z); c.wb.w = fxg_complement4(c.wb.w);
            }
            fxg_st16_stream(dst_b + c.o, c.wb);
            if (has_q) fxg_st16_stream(dst_q + c.o, c.wq);
            if (c.e2 < 16) {                                              // third and later reads of this chunk: overwrite byte by byte (same lane, program order)
                u32 r = c.k + 2u;
                for (u32 x = c.o + (u32)c.e2; x < c.o + 16u; ++x) {
                    while (k_off[r + 1u] <= x) ++r;
                    fxg_gather_byte<REV, MASK>(a, src_b, src_q, dst_b, dst_q, k_src[r], x - k_off[r], x, &bad);
                }
            }
        }
    }
    // bytes outside the full chunks: [0, o_lo) and [o_hi, S)
    const u32 nb = o_lo + (S - o_hi);
    for (u32 x = tid; x < nb; x += nthreads) {
        const u32 o = x < o_lo ? x : o_hi + (x - o_lo);
        const u32 k = fxg_rank_of(k_off, k_tab, nk, S, o);
        fxg_gather_byte<REV, MASK>(a, src_b, src_q, dst_b, dst_q, k_src[k], o - k_off[k], o, &bad);
    }
    return bad;   // nonzero: a byte outside ACGTN/acgtn reached the complement (REV only)
}
