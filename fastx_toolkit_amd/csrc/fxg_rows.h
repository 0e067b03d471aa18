// fxg_rows.h -- quality-trim / quality-filter with compaction for rows of 80 to 152 bytes: one WAVE per tile of 64 reads, one LANE
// per read, HBM traffic = input + output.
//
// Why a second kernel for the same stages (fxg_kernel_tiles<0,0> stays the general form: any read length, every other stage):
// the tile kernel decides from the quality rows, then waits for the tile's place in the output (a prefix over all earlier tiles)
// and gathers the kept prefixes from HBM -- it fetches the quality rows twice and the bases at cache-line granularity around every
// kept read: 27.4 GB through the L2s per cfg2 launch against 22.3 GB of input + output.  Here every byte crosses the fabric once
// (rocprofv3 PMC: 15.2 GB read + 7.7 GB written, profiles/pmc_traffic.json, r02_rows_pmc/):
//   stage A, tile `cur`:  the quality rows come in by LDS-DMA (buffer_load ... lds, 1 KB per wave instruction) and are transposed
//      through a 9.6 KB staging buffer: lane r holds read r (38 dwords for 150 bases, v_alignbyte_b32).  The lane builds its read's
//      threshold bitmap in registers (v_dot4_u32_u8 gathers the compare flags), decides it, the wave scans (keep, length) with DPP
//      and publishes the tile's totals to the scanners (fxg_scanner_multi, fxg_device.h).
//   stage B, tile `pend` (decided one step earlier; its quality rows waited in registers): the tile's place in the output has
//      arrived; every lane writes its kept prefix into the staging buffer at its offset in the tile's packed output, and the wave
//      stores the packed bytes as whole, aligned 16-byte units of the global array (fxg_rows_flush).  The base rows take the same
//      road: LDS-DMA, transpose, pack, flush.
// A workgroup IS one wave: no workgroup barrier anywhere, no gather tables, no second pass.  Twelve waves per CU (138 VGPRs).
// Measured against fxg_kernel_tiles<0,0> on the same box it is 3-4 % faster (4.03 against 4.16-4.19 ms for cfg2), not the 18 % the
// bytes suggest: both sit at what the memory system gives this read : write mix -- a plain streaming kernel that reads 15 GB and writes 7.3 GB, nothing
// else, takes 4.25-4.5 ms (scripts/ubench/mix_rw.hip, profiles/r02/z_mix_rw.txt); read alone 2.34 ms, write alone 1.18 ms.
#pragma once
#include "fxg_kernels.h"

#define FXG_ROWS_T 64u                  // reads per tile = lanes per wave
#ifndef FXG_ROWS_LB
#define FXG_ROWS_LB 3                   // waves per SIMD (__launch_bounds__): two tiles' quality rows live in registers; the kernel uses 138 VGPRs of the 168 this allows
#endif
#ifndef FXG_ROWS_SCAN_K
#define FXG_ROWS_SCAN_K 8               // tiles per scanner batch / 64 (fxg_scanner_multi)
#endif
#ifndef FXG_ROWS_LD_AUX
#define FXG_ROWS_LD_AUX 2               // cache policy bits of the row loads: 2 = nt, every row is read once (round 2 saw no gain; re-measured at HEAD: cfg2 4.13 -> 4.06 ms in
                                        // four alternating runs, profiles/r06/y_cfg2_rows_nt.txt; a read-only LDS-DMA stream of 15 GB: 6.36 -> 6.77 TB/s, x_read_stream_by_access_form.txt)
#endif
#ifndef FXG_ROWS_NSCAN
#define FXG_ROWS_NSCAN 8                // scanner waves
#endif

// staging buffer: the tile's rows + slack (every lane reads and packs a full register row, up to 156 bytes, whatever the stride)
__host__ __device__ inline u32 fxg_rows_lds(u32 stride, u32 lanes_per_read = 1u, u32 reads_per_lane = 1u) { return fxg_r16(FXG_ROWS_T * reads_per_lane / lanes_per_read * stride) + 176u; }

// The lane's row -> its own threshold bitmap, bit i = (byte i >= thr), K = (128 - thr) * 0x01010101 (fxg_ge_flags).  Two dwords at a
// time: the flags are bytes of 0x80 and a dot product with the byte weights (1,2,4,8) / (16,32,64,128) gathers eight of them
// (v_dot4_u32_u8 accumulates), 128 x the byte of the bitmap.
template <int NW>
FXG_HD void fxg_rows_bits(const u32 (&row)[NW], u32 K, u32 (&M)[(NW * 4 + 31) / 32])
{
#pragma unroll
    for (int p = 0; p < (NW + 1) / 2; ++p) {
        const u32 f0 = fxg_ge_flags(row[2 * p], K), f1 = 2 * p + 1 < NW ? fxg_ge_flags(row[2 * p + 1], K) : 0u;
#ifndef FXG_HOST_EMULATION
        const u32 b = __builtin_amdgcn_udot4(f1, 0x80402010u, __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false), false);
#else       // the same sum of products, spelled out (tests/emu)
        u32 b = 0;
        for (int i = 0; i < 4; ++i) b += ((f0 >> (8 * i)) & 0xFFu) * (1u << i) + ((f1 >> (8 * i)) & 0xFFu) * (16u << i);
#endif
        if ((p & 3) == 0) M[p >> 2] = b >> 7;
        else M[p >> 2] |= b << (8 * (p & 3) - 7);
    }
}
// 1 + the highest set bit below `len`, 0 when there is none (fxg_bits_last on a bitmap that starts at bit 0 and lives in registers)
template <int NM>
FXG_HD u32 fxg_rows_last(const u32 (&M)[NM], u32 len)
{
    u32 r = 0;
#pragma unroll
    for (int w = 0; w < NM; ++w) {
        const int nb = (int)len - 32 * w;
        const u32 x = nb >= 32 ? M[w] : nb > 0 ? M[w] & ((1u << nb) - 1u) : 0u;
        r = x ? 32u * (u32)w + 32u - (u32)__builtin_clz(x | 1u) : r;
    }
    return r;
}
// set bits below `len` (fxg_bits_count)
template <int NM>
FXG_HD u32 fxg_rows_count(const u32 (&M)[NM], u32 len, bool invert)
{
    u32 c = 0;
#pragma unroll
    for (int w = 0; w < NM; ++w) {
        const int nb = (int)len - 32 * w;
        const u32 m = invert ? ~M[w] : M[w];
        c += (u32)__builtin_popcount(nb >= 32 ? m : nb > 0 ? m & ((1u << nb) - 1u) : 0u);
    }
    return c;
}

// fxg_decide_a<0> for lanes that hold quality rows in registers, in pieces so that ONE lane (rows up to 152 bytes) or TWO lanes (rows
// up to 304 bytes: lane 2r holds bytes [0, 4 NW) of read r, lane 2r + 1 the rest) can decide a read:
//   fxg_rows_piece_last : this piece's threshold bitmap G and 1 + its highest set bit below `plen` (quality trimmer, fastq_quality_trimmer.c:94-101)
//   fxg_rows_piece_low  : bases below the filter threshold among the piece's first `plen` (fastq_quality_filter.c:110-129 in closed form)
//   fxg_rows_verdict    : the read's result word from its length, trim point and low count
template <int NW>
FXG_HD u32 fxg_rows_piece_last(const FxgKArgs &a, const u32 (&q)[NW], u32 plen, u32 (&G)[(NW * 4 + 31) / 32])
{
    constexpr int NM = (NW * 4 + 31) / 32;
    const bool trim = (a.stages & FXG_STAGE_QTRIM) != 0u, same = a.tq == a.fq;
    if (trim || same) fxg_rows_bits<NW>(q, (128u - a.tq) * 0x01010101u, G);
    return trim ? fxg_rows_last<NM>(G, plen) : 0u;
}
template <int NW>
FXG_HD u32 fxg_rows_piece_low(const FxgKArgs &a, const u32 (&q)[NW], const u32 (&G)[(NW * 4 + 31) / 32], u32 plen)
{
    constexpr int NM = (NW * 4 + 31) / 32;
    if (!(a.stages & FXG_STAGE_QFILTER)) return 0u;
    if (a.tq == a.fq) return fxg_rows_count<NM>(G, plen, true);      // trimmer and filter at the same threshold share one bitmap
    u32 F[NM];
    fxg_rows_bits<NW>(q, (128u - a.fq) * 0x01010101u, F);
    return fxg_rows_count<NM>(F, plen, true);
}
FXG_HD u32 fxg_rows_verdict(const FxgKArgs &a, u32 rl, u32 k, u32 low, u32 *keep_out, u32 *len_out)
{
    u32 reason = FXG_R_KEPT, keep = 1, curlen = rl;
    if (a.stages & FXG_STAGE_QTRIM) {                         // fastq_quality_trimmer.c:94-101
        curlen = k;
        if (!(k > 0 && (int)k >= a.qt_min_len)) { keep = 0; reason = FXG_R_QTRIM; }
    }
    if (a.stages & FXG_STAGE_QFILTER) {                       // fastq_quality_filter.c:110-129,155 in closed form
        int n0 = (int)curlen * a.qf_keep_pct / 100;
        if (n0 < 0) n0 = 0;
        if (keep && (a.qf_drop_all || (int)low > n0)) { keep = 0; reason = FXG_R_QFILTER; }
    }
    *keep_out = keep; *len_out = curlen;
    return (curlen & 0xFFFFu) | (keep << 16) | (reason << 17);
}

// one lane, one read
template <int NW>
FXG_HD u32 fxg_rows_decide(const FxgKArgs &a, const u32 (&q)[NW], u32 read, u32 *keep_out, u32 *len_out)
{
    constexpr int NM = (NW * 4 + 31) / 32;
    const u32 rl = a.len ? (u32)a.len[read] : a.fixed_len;
    u32 G[NM];
    const u32 k = fxg_rows_piece_last<NW>(a, q, rl, G);
    const u32 curlen = (a.stages & FXG_STAGE_QTRIM) ? k : rl;
    const u32 low = fxg_rows_piece_low<NW>(a, q, G, curlen);
    const u32 w = fxg_rows_verdict(a, rl, k, low, keep_out, len_out);
    a.res[read] = w;
    return w;
}
// the piece lengths of a read of `len` bytes split at HB = 4 NW
FXG_HD u32 fxg_rows_piece_len(u32 len, u32 HB, u32 h) { return h == 0u ? (len < HB ? len : HB) : (len > HB ? len - HB : 0u); }

#ifndef FXG_HOST_EMULATION
// A workgroup is ONE wave: its LDS accesses execute in order, so "every lane's reads / writes before this point are done" needs no
// barrier, only the wave's own LDS counter at zero and the compiler kept from moving accesses across the point.
typedef __attribute__((address_space(3))) unsigned char fxg_lds_u8;     // the staging buffer as an explicit LDS pointer
#define FXG_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// Tile rows src[tb, tb + tbytes) -> LDS staging buffer, 16-byte chunk c at byte 16 c, by LDS-DMA: 1 KB per wave instruction (lane i
// lands at base + 16 i), no registers in between, and NOT waited for here (fxg_rows_landed() does that).  The loads go through a
// buffer descriptor over the tile's bytes: one scalar base for the wave, lanes past the tile's end deliver zeros.
// wave load K = bytes [1024 K, 1024 K + 1024): 4096 (K / 4) in the lane's offset, 1024 (K % 4) in the instruction's (a compile-time
// field) -- both are range-checked against the tile's bytes (a scalar offset would not be) and the instruction offset also moves the
// LDS side
template <int K, int NC>
__device__ __forceinline__ void fxg_rows_fetch_from(__amdgpu_buffer_rsrc_t rs, fxg_lds_u8 *sbuf, u32 lane, u32 nck, u32 room)
{
    typedef __attribute__((address_space(3))) void lptr_t;
    if constexpr (K < NC) {
        // the last wave load of a tile may reach past the buffer's rows (64 rows are seldom whole KBs): those lanes stay out
        if ((u32)K < nck && ((u32)(K + 1) * 1024u <= room || (u32)K * 1024u + (lane << 4) < room)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t *)(sbuf + (K >> 2) * 4096), 16, (int)(lane << 4) + (K >> 2) * 4096, 0, (K & 3) * 1024, FXG_ROWS_LD_AUX);
        fxg_rows_fetch_from<K + 1, NC>(rs, sbuf, lane, nck, room);
    }
}
template <int NW>
__device__ __forceinline__ void fxg_rows_fetch(const uint8_t *src, u64 tb, u32 tbytes, fxg_lds_u8 *sbuf, u32 lane)
{
    constexpr int NC = (NW * 4 + 15) / 16;            // chunks per lane: 64 lanes x NC x 16 bytes cover 64 rows of up to 4 NW bytes
    const u32 nck = (tbytes + 1023u) >> 10;           // wave loads this tile needs (uniform)
    // whole dwords (the range check is per dword); only a batch's last tile can end inside a dword (64 rows are whole dwords): its
    // last 1-3 bytes come in by byte loads, so nothing past the arrays is ever read
    const u32 whole = tbytes & ~3u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(src + tb), 0, (int)whole, 0x00020000);
    fxg_rows_fetch_from<0, NC>(rs, sbuf, lane, nck, fxg_r16(tbytes));
    if (whole != tbytes && lane < tbytes - whole) sbuf[whole + lane] = src[tb + whole + lane];
}
__device__ __forceinline__ void fxg_rows_landed()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FXG_WAVE_SYNC();
}

// staging buffer -> row[] of the lane's own piece, NW dwords from byte `base` on: dwords from the 4-byte aligned address below its first
// byte, shifted down by the odd bytes
template <int NW>
__device__ __forceinline__ void fxg_rows_read(const unsigned char *sbuf, u32 base, u32 (&row)[NW])
{
    const u32 sh = base & 3u;
    typedef const u32 __attribute__((address_space(3))) lds_u32;                // dword reads: the address is only 4-byte aligned
    lds_u32 *w = (lds_u32 *)(sbuf + (base & ~3u));
    // batches of eight dwords: all 39 reads in flight at once would need 39 more registers next to the rows already held
    u32 carry = w[0];
#pragma unroll
    for (int k0 = 0; k0 < NW; k0 += 8) {
        u32 d[9];
        d[0] = carry;
#pragma unroll
        for (int j = 1; j <= 8; ++j) d[j] = k0 + j <= NW ? w[k0 + j] : 0u;       // past the row's end: bytes nobody looks at
#pragma unroll
        for (int j = 0; j < 8; ++j) if (k0 + j < NW) row[k0 + j] = __builtin_amdgcn_alignbyte(d[j + 1], d[j], sh);
        carry = d[8];
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The lane's first `len` bytes of row[] into the staging buffer at byte D (the read's place in the tile's packed output):
// up to three head bytes to reach a dword boundary, dwords of the row shifted to match, up to three tail bytes.
// `fast` (wave-uniform: every kept read of the tile has at least 4 bytes): every lane writes ALL its dwords, from the top down and
// without predicates.  What a lane writes past its own bytes lands in dwords that start inside a later read, and that read's lane
// writes them in a LATER step (its index for the same address is smaller), so the right bytes win; the dword a read shares with
// its predecessor only gets the read's own head bytes, last of all.
template <int NW>
__device__ __forceinline__ void fxg_rows_pack(unsigned char *obuf, u32 D, const u32 (&row)[NW], u32 len, bool fast)
{
    if (fast) {
        const u32 al = D & 3u;
        const u32 sel = 0x07060504u - al * 0x01010101u;          // v_perm_b32: dword j of the output = row bytes 4 j - al ... 4 j - al + 3
        volatile u32 *w = reinterpret_cast<volatile u32 *>(obuf + (D & ~3u));   // volatile: this order, one dword per instruction
#pragma unroll
        for (int j = NW; j >= 1; --j) w[j] = __builtin_amdgcn_perm(j < NW ? row[j] : 0u, row[j - 1], sel);
        volatile unsigned char *p = obuf + D;
        if (al == 0u) w[0] = row[0];
        else {
            p[0] = (unsigned char)row[0];
            if (al <= 2u) p[1] = (unsigned char)(row[0] >> 8);
            if (al == 1u) p[2] = (unsigned char)(row[0] >> 16);
        }
        return;
    }
    u32 h = (0u - D) & 3u;
    h = h < len ? h : len;
    const u32 nb = (len - h) >> 2, t = (len - h) & 3u;
    unsigned char *p = obuf + D;
    if (h >= 1u) p[0] = (unsigned char)row[0];
    if (h >= 2u) p[1] = (unsigned char)(row[0] >> 8);
    if (h >= 3u) p[2] = (unsigned char)(row[0] >> 16);
    u32 *w = reinterpret_cast<u32 *>(p + h);
    u32 tw = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const u32 v = __builtin_amdgcn_alignbyte(k + 1 < NW ? row[k + 1] : 0u, row[k], h);
        if ((u32)k < nb) w[k] = v;
        tw = (u32)k == nb ? v : tw;
    }
    unsigned char *e = p + h + (nb << 2);
    if (t >= 1u) e[0] = (unsigned char)tw;
    if (t >= 2u) e[1] = (unsigned char)(tw >> 8);
    if (t >= 3u) e[2] = (unsigned char)(tw >> 16);
}

// The tile's packed output obuf[0, totb) -> out[base, base + totb): whole 16-byte units of the GLOBAL array, one per lane and
// coalesced (the unit at 16 i holds packed bytes 16 i - (base & 15) ...: two aligned LDS reads and a funnel shift that is the
// same for the whole tile); the bytes in the units shared with the neighbouring tiles go one by one.
__device__ __forceinline__ void fxg_rows_flush(uint8_t *out, u64 base, u32 totb, const unsigned char *obuf, u32 lane)
{
    const u32 sh = (u32)base & 15u;
    uint8_t *g0 = out + (base - sh);
    const u32 first = sh ? 1u : 0u, end = (sh + totb) >> 4;      // whole units: first <= i < end
    const u32 s2 = (16u - sh) & 15u, qd = s2 >> 2, r = s2 & 3u;  // unit i starts at packed byte 16 i - sh = 16 (i - first) + s2
#if FXG_STORE_GRID
    // lane l takes the units whose place in their 128-byte line is l mod 8: every store instruction of the wave then covers eight whole lines of the output
    // (the first one starts at most seven lanes short) instead of straddling nine: cfg2 4.08 -> 4.03 ms, mean of four alternating runs (profiles/r06/store_grid_cfg2.txt)
    const int a8 = (int)(((base - sh) >> 4) & 7u);
    for (int ii = (int)lane - a8; ii < (int)end; ii += 64) {
        if (ii < (int)first) continue;
        const u32 i = (u32)ii;
#else
    for (u32 i = first + lane; i < end; i += 64u) {
#endif
        const u32x4 *src = reinterpret_cast<const u32x4 *>(obuf + ((i - first) << 4));
        const u32x4 v0 = src[0], v1 = src[1];
        u32 w0, w1, w2, w3, w4;
        if (qd == 0u)      { w0 = v0.x; w1 = v0.y; w2 = v0.z; w3 = v0.w; w4 = v1.x; }
        else if (qd == 1u) { w0 = v0.y; w1 = v0.z; w2 = v0.w; w3 = v1.x; w4 = v1.y; }
        else if (qd == 2u) { w0 = v0.z; w1 = v0.w; w2 = v1.x; w3 = v1.y; w4 = v1.z; }
        else               { w0 = v0.w; w1 = v1.x; w2 = v1.y; w3 = v1.z; w4 = v1.w; }
        const u32x4 o = {__builtin_amdgcn_alignbyte(w1, w0, r), __builtin_amdgcn_alignbyte(w2, w1, r), __builtin_amdgcn_alignbyte(w3, w2, r), __builtin_amdgcn_alignbyte(w4, w3, r)};
        fxg_st16_stream(g0 + ((u64)i << 4), o);
    }
    const bool whole = end > first;
    const u32 hb = whole ? (sh ? 16u - sh : 0u) : totb;          // packed bytes [0, hb) and [ts, totb) lie in shared units
    const u32 ts = whole ? (end << 4) - sh : totb;
    if (lane < hb) out[base + lane] = obuf[lane];
    if (ts + lane < totb) out[base + ts + lane] = obuf[ts + lane];
}

#ifdef FXG_ABLATION
#define FXG_PHASE(i) do { const u64 now_ = __builtin_amdgcn_s_memrealtime(); ph[i] += now_ - pt; pt = now_; } while (0)
#else
#define FXG_PHASE(i) do { } while (0)
#endif

// H = lanes per read: 1 for rows up to 4 NW bytes; 2 for rows up to 8 NW bytes -- lane 2r holds the first 4 NW bytes of read r, lane 2r + 1
// the rest, a tile is 32 reads.  The two lanes exchange their pieces' trim point and low-quality count (one DPP swap each) and then
// behave like two reads that happen to be adjacent in the output: every later step (scan, pack, flush) works on pieces.
template <int NW, int H = 1>
__global__ __launch_bounds__(64, FXG_ROWS_LB) void fxg_kernel_rows(const FxgKArgs a)
{
    constexpr u32 TR = FXG_ROWS_T / (u32)H, HB = 4u * (u32)NW;          // reads per tile; bytes per piece
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    fxg_lds_u8 *lsm = (fxg_lds_u8 *)smem;
    u32 lane = threadIdx.x;
    const u32 stride = a.stride;
    // the first a.nscan workgroups (= waves) to get here turn the tiles' totals into prefixes; the others process tiles
    u32 role = 0;
    if (lane == 0) role = atomicAdd(a.role, 1u);
    role = (u32)__builtin_amdgcn_readfirstlane((int)role);
    // one scanner: the run-by-run form, whose progress does not depend on how many workers are resident (fxg_device.h)
    if (role < a.nscan) { if (a.nscan == 1u) fxg_scanner_k<FXG_ROWS_SCAN_K>(a); else fxg_scanner_multi<FXG_ROWS_SCAN_K>(a, role); return; }
    const u32 G = a.ticket_groups, grp = blockIdx.x % G;
    u32 *my_ticket = a.ticket + grp * FXG_TICKET_STRIDE;
    u32 tk = 0;
    if (lane == 0) tk = atomicAdd(my_ticket, 1u);
    u32 cur = (u32)__builtin_amdgcn_readfirstlane((int)tk) * G + grp;          // lane 0 drew it; a scalar from here on (tile offsets, descriptors, loop bounds)
    u64 t_in = 0, t_kept = 0, t_bases = 0, t_qtrim = 0, t_qfilter = 0;        // wave-uniform tallies (a12), added to the launch's slots once
#ifdef FXG_ABLATION
    u64 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = __builtin_amdgcn_s_memrealtime();   // 100 MHz clocks per phase, summed over the wave's tiles
#endif
    // Two tiles per wave in flight, one step apart (as in fxg_kernel_tiles): stage A decides tile `cur` and publishes its totals;
    // stage B writes out tile `pend`, decided one step earlier -- its prefix has had a whole stage B and a whole stage A to arrive.
    // What stage B needs of `pend` waits in registers: the quality rows (qp), and per lane (keep, length, offsets inside the tile).
    u32 qp[NW];
    u32 p_info = 0, p_exc = 0, p_totb = 0, pend = FXG_NO_TILE;          // p_info = keep << 31 | kept length << 16 | byte offset in the tile's packed output
    for (;;) {
        asm volatile("" : "+v"(lane));     // opaque per iteration: otherwise every per-lane address of the body is hoisted out of the loop and spilled
        const bool havec = cur < a.ntiles, havep = pend != FXG_NO_TILE;
        u32 q[NW];
        u32 c_info = 0, c_exc = 0, c_totb = 0;
        // ------------------------------ stage A: tile `cur` ------------------------------
        if (havec) {
            const u32 r0 = cur * TR;
            const u64 left = a.n - (u64)r0;
            const u32 nreads = left < (u64)TR ? (u32)left : TR;
            const u64 tb = (u64)r0 * stride;
            const u32 tbytes = nreads * stride;
            const u32 rd = lane / (u32)H, hf = lane % (u32)H;               // this lane's read inside the tile and which piece of it
            // quality rows: HBM -> staging buffer -> this lane's piece in registers; the lane(s) decide the read from them
            fxg_rows_fetch<NW>(a.qual, tb, tbytes, lsm, lane);
            fxg_rows_landed();
            FXG_PHASE(0);
            fxg_rows_read<NW>(smem, rd * stride + hf * HB, q);
            FXG_WAVE_SYNC();                                                  // every lane has its row: the buffer is free for stage B
            u32 keep = 0, olen = 0, word = 0;
            if constexpr (H == 1) {
                if (lane < nreads) word = fxg_rows_decide<NW>(a, q, r0 + lane, &keep, &olen);
            } else {
                constexpr int NM = (NW * 4 + 31) / 32;
                const bool live = rd < nreads;
                const u32 rl = live ? (a.len ? (u32)a.len[r0 + rd] : a.fixed_len) : 0u;
                u32 G[NM];
                const u32 kp = fxg_rows_piece_last<NW>(a, q, fxg_rows_piece_len(rl, HB, hf), G);
                const u32 ko = (u32)__builtin_amdgcn_mov_dpp((int)kp, 0xB1, 0xf, 0xf, true);          // quad_perm [1,0,3,2]: the partner lane's value
                const u32 k_lo = hf ? ko : kp, k_hi = hf ? kp : ko;
                const u32 k = k_hi ? HB + k_hi : k_lo;                       // 1 + the highest position at or above the threshold
                const u32 cl = (a.stages & FXG_STAGE_QTRIM) ? k : rl;
                const u32 lp = fxg_rows_piece_low<NW>(a, q, G, fxg_rows_piece_len(cl, HB, hf));
                const u32 lo_ = (u32)__builtin_amdgcn_mov_dpp((int)lp, 0xB1, 0xf, 0xf, true);
                u32 rlen = 0;
                word = fxg_rows_verdict(a, rl, k, lp + lo_, &keep, &rlen);
                if (!live) { keep = 0; word = 0; }
                if (live && hf == 0u) a.res[r0 + rd] = word;
                olen = keep ? fxg_rows_piece_len(rlen, HB, hf) : 0u;         // this piece's share of the kept prefix
                if (hf) word = 0;                                            // tallies and the kept count: once per read
            }
            {
                const u32 why = (H == 1 ? lane < nreads : rd < nreads) ? (word >> 17) & 0xFu : 0u;
                t_qtrim += (u64)__builtin_popcountll(__ballot(why == FXG_R_QTRIM));
                t_qfilter += (u64)__builtin_popcountll(__ballot(why == FXG_R_QFILTER));
            }
            // wave scan of (kept reads, kept bytes) in ONE word: at most 64 reads and 64 * 152 bytes per tile; a read counts once
            const u32 mine = keep ? ((hf == 0u ? 1u << 16 : 0u) | olen) : 0u;
            const u32 inc = fxg_wave_scan_dpp(mine);
            const u32 tot = (u32)__builtin_amdgcn_readlane((int)inc, 63);
            const u32 totc = tot >> 16;
            c_totb = tot & 0xFFFFu;
            c_exc = (inc - mine) >> 16;
            // "keep" of a piece: it has bytes to write -- and the FIRST piece of a kept read always carries it, bytes or not: it writes the read's
            // metadata (a read of length 0 is kept by the filter alone; an empty piece packs nothing and sends the tile down the predicated path)
            c_info = ((keep && (H == 1 || hf == 0u || olen)) ? 1u << 31 : 0u) | (olen << 16) | ((inc - mine) & 0xFFFFu);
            if (lane == 0) fxg_publish_total(a, cur, totc, c_totb);
            t_in += nreads; t_kept += totc; t_bases += c_totb;
            FXG_PHASE(1);
        }
        // ------------------------------ stage B: tile `pend` ------------------------------
        u64 bc[2] = {0, 0};
        if (havep) {
            if (!FXG_DBG(a, 2u)) fxg_wait_prefix_wave(a, pend, bc); else { bc[0] = (u64)pend * TR; bc[1] = bc[0] * stride; }
            FXG_PHASE(2);
        }
        u32 nxt = 0;
        if (havec && lane == 0) nxt = atomicAdd(my_ticket, 1u);               // next ticket: in flight during the write-out (never held across a wait)
        if (havep) {
            const u32 r0 = pend * TR;
            const u64 left = a.n - (u64)r0;
            const u32 nreads = left < (u64)TR ? (u32)left : TR;
            const u64 tb = (u64)r0 * stride;
            const u32 tbytes = nreads * stride;
            const u32 rd = lane / (u32)H, hf = lane % (u32)H;
            const bool placed = bc[0] != ~0ull;         // false: the wait for the prefix expired (error flag set) -- nothing of this tile is written
            const u32 keep = placed ? p_info >> 31 : 0u, olen = (p_info >> 16) & 0x7FFFu, exb = p_info & 0xFFFFu;
            // the kept prefixes of the quality rows, packed in read order, into the staging buffer; whole 16-byte units from there
            const bool fast = __ballot(keep && olen < 4u) == 0ull;
            if (keep && !FXG_DBG(a, 64u)) fxg_rows_pack<NW>(smem, exb, qp, olen, fast);
            FXG_WAVE_SYNC();
            FXG_PHASE(6);
            if (placed && !FXG_DBG(a, 1u)) fxg_rows_flush(a.out_qual, bc[1], p_totb, smem, lane);
            if constexpr (H == 1) { if (keep) fxg_write_kept_meta(a, bc[0] + p_exc, olen, r0 + lane, bc[1] + exb); }
            else {                                                            // the first piece speaks for the read (it is never empty when the read is kept)
                const u32 oo = (u32)__builtin_amdgcn_mov_dpp((int)olen, 0xB1, 0xf, 0xf, true);
                if (keep && hf == 0u) fxg_write_kept_meta(a, bc[0] + p_exc, olen + oo, r0 + rd, bc[1] + exb);
            }
            FXG_WAVE_SYNC();                                                  // the buffer is free again
            FXG_PHASE(3);
            // the base rows take the same road: HBM -> staging buffer -> registers -> packed -> out
            if (!FXG_DBG(a, 4u)) {
                u32 b[NW];
                fxg_rows_fetch<NW>(a.bases, tb, tbytes, lsm, lane);
                fxg_rows_landed();
                FXG_PHASE(4);
                fxg_rows_read<NW>(smem, rd * stride + hf * HB, b);
                FXG_WAVE_SYNC();
                if (keep && !FXG_DBG(a, 64u)) fxg_rows_pack<NW>(smem, exb, b, olen, fast);
                FXG_WAVE_SYNC();
                FXG_PHASE(7);
                if (placed && !FXG_DBG(a, 1u)) fxg_rows_flush(a.out_bases, bc[1], p_totb, smem, lane);
                FXG_WAVE_SYNC();
            }
            FXG_PHASE(5);
        }
        if (!havec) break;
#pragma unroll
        for (int k = 0; k < NW; ++k) qp[k] = q[k];
        p_info = c_info; p_exc = c_exc; p_totb = c_totb; pend = cur;
        cur = (u32)__builtin_amdgcn_readfirstlane((int)nxt) * G + grp;
    }
#ifdef FXG_ABLATION
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(reinterpret_cast<u64 *>(a.errflag + 10) + i, ph[i]);
#endif
    if (lane == 0) {
        if (t_in) atomicAdd(&a.tally[0], t_in);
        if (t_kept) atomicAdd(&a.tally[1], t_kept);
        if (t_bases) atomicAdd(&a.tally[2], t_bases);
        if (t_qtrim) atomicAdd(&a.tally[4 + FXG_R_QTRIM], t_qtrim);
        if (t_qfilter) atomicAdd(&a.tally[4 + FXG_R_QFILTER], t_qfilter);
    }
}
// ------------------------------------------------------------------------------------------------
// Short rows (28..79 bytes: 36-, 50- and 76-base reads), round 6: R reads per lane.  A 64-read tile of 36-byte rows is 2.3 KB -- the per-tile work (ticket,
// scan, publish, the wait for the tile's place, two flushes) then costs more than the bytes, and the tile kernel was the faster one below 80 bytes at 0.44-0.59 of
// the HBM peak (profiles/r02/af_rows_vs_tiles_by_length.txt).  Here a tile is R sub-tiles of 64 consecutive reads (R x NW ~ 40 dwords per lane, ~10 KB per
// tile: what a 64-read tile of 150-byte rows is), fetched by ONE LDS-DMA burst, published as ONE total, placed by ONE prefix and flushed once per array; lane l
// holds reads l, l + 64, ... of the tile, so that sub-tile j is a contiguous run of the packed output and the sub-tiles are packed in order: what the
// unpredicated pack of one sub-tile writes past its own bytes lands in bytes of a LATER sub-tile (or the buffer's slack) and is overwritten by it, exactly as
// between the lanes of one sub-tile (fxg_rows_pack).  Everything else is fxg_kernel_rows<NW, 1>: same decisions (fxg_rows_decide), same scanners, same flush.
// ------------------------------------------------------------------------------------------------
template <int NW, int R>
__global__ __launch_bounds__(64, FXG_ROWS_LB) void fxg_kernel_rows_multi(const FxgKArgs a)
{
    constexpr u32 TR = FXG_ROWS_T * (u32)R;                              // reads per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    fxg_lds_u8 *lsm = (fxg_lds_u8 *)smem;
    u32 lane = threadIdx.x;
    const u32 stride = a.stride;
    u32 role = 0;
    if (lane == 0) role = atomicAdd(a.role, 1u);
    role = (u32)__builtin_amdgcn_readfirstlane((int)role);
    if (role < a.nscan) { if (a.nscan == 1u) fxg_scanner_k<FXG_ROWS_SCAN_K>(a); else fxg_scanner_multi<FXG_ROWS_SCAN_K>(a, role); return; }
    const u32 G = a.ticket_groups, grp = blockIdx.x % G;
    u32 *my_ticket = a.ticket + grp * FXG_TICKET_STRIDE;
    u32 tk = 0;
    if (lane == 0) tk = atomicAdd(my_ticket, 1u);
    u32 cur = (u32)__builtin_amdgcn_readfirstlane((int)tk) * G + grp;
    u64 t_in = 0, t_kept = 0, t_bases = 0, t_qtrim = 0, t_qfilter = 0;
    u32 qp[R][NW];
    u32 p_info[R], p_exc[R], p_totb = 0, pend = FXG_NO_TILE;            // p_info[j] = keep << 31 | kept length << 16 | byte offset in the tile's packed output
#pragma unroll
    for (int j = 0; j < R; ++j) { p_info[j] = 0u; p_exc[j] = 0u; }
    for (;;) {
        asm volatile("" : "+v"(lane));
        const bool havec = cur < a.ntiles, havep = pend != FXG_NO_TILE;
        u32 q[R][NW];
        u32 c_info[R], c_exc[R], c_totb = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) { c_info[j] = 0u; c_exc[j] = 0u; }
        // ------------------------------ stage A: tile `cur` ------------------------------
        if (havec) {
            const u32 r0 = cur * TR;
            const u64 left = a.n - (u64)r0;
            const u32 nreads = left < (u64)TR ? (u32)left : TR;
            const u64 tb = (u64)r0 * stride;
            const u32 tbytes = nreads * stride;
            fxg_rows_fetch<NW * R>(a.qual, tb, tbytes, lsm, lane);
            fxg_rows_landed();
#pragma unroll
            for (int j = 0; j < R; ++j) fxg_rows_read<NW>(smem, ((u32)j * FXG_ROWS_T + lane) * stride, q[j]);
            FXG_WAVE_SYNC();                                                  // every lane has its rows: the buffer is free for stage B
            u32 base = 0;                                                     // (kept reads << 16 | kept bytes) of the sub-tiles so far
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const u32 rd = (u32)j * FXG_ROWS_T + lane;
                u32 keep = 0, olen = 0, word = 0;
                if (rd < nreads) word = fxg_rows_decide<NW>(a, q[j], r0 + rd, &keep, &olen);
                const u32 why = rd < nreads ? (word >> 17) & 0xFu : 0u;
                t_qtrim += (u64)__builtin_popcountll(__ballot(why == FXG_R_QTRIM));
                t_qfilter += (u64)__builtin_popcountll(__ballot(why == FXG_R_QFILTER));
                const u32 mine = keep ? ((1u << 16) | olen) : 0u;
                const u32 inc = fxg_wave_scan_dpp(mine);
                const u32 ex = inc - mine + base;
                base += (u32)__builtin_amdgcn_readlane((int)inc, 63);
                c_exc[j] = ex >> 16;
                c_info[j] = (keep ? 1u << 31 : 0u) | (olen << 16) | (ex & 0xFFFFu);
            }
            const u32 totc = base >> 16;
            c_totb = base & 0xFFFFu;
            if (lane == 0) fxg_publish_total(a, cur, totc, c_totb);
            t_in += nreads; t_kept += totc; t_bases += c_totb;
        }
        // ------------------------------ stage B: tile `pend` ------------------------------
        u64 bc[2] = {0, 0};
        if (havep) fxg_wait_prefix_wave(a, pend, bc);
        u32 nxt = 0;
        if (havec && lane == 0) nxt = atomicAdd(my_ticket, 1u);
        if (havep) {
            const u32 r0 = pend * TR;
            const u64 left = a.n - (u64)r0;
            const u32 nreads = left < (u64)TR ? (u32)left : TR;
            const u64 tb = (u64)r0 * stride;
            const u32 tbytes = nreads * stride;
            const bool placed = bc[0] != ~0ull;
            u64 shortm = 0;
#pragma unroll
            for (int j = 0; j < R; ++j) shortm |= __ballot(placed && (p_info[j] >> 31) && ((p_info[j] >> 16) & 0x7FFFu) < 4u);
            const bool fast = shortm == 0ull;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const u32 keep = placed ? p_info[j] >> 31 : 0u, olen = (p_info[j] >> 16) & 0x7FFFu, exb = p_info[j] & 0xFFFFu;
                if (keep) fxg_rows_pack<NW>(smem, exb, qp[j], olen, fast);
            }
            FXG_WAVE_SYNC();
            if (placed) fxg_rows_flush(a.out_qual, bc[1], p_totb, smem, lane);
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const u32 keep = placed ? p_info[j] >> 31 : 0u, olen = (p_info[j] >> 16) & 0x7FFFu, exb = p_info[j] & 0xFFFFu;
                if (keep) fxg_write_kept_meta(a, bc[0] + p_exc[j], olen, r0 + (u32)j * FXG_ROWS_T + lane, bc[1] + exb);
            }
            FXG_WAVE_SYNC();                                                  // the buffer is free again
            u32 b[R][NW];
            fxg_rows_fetch<NW * R>(a.bases, tb, tbytes, lsm, lane);
            fxg_rows_landed();
#pragma unroll
            for (int j = 0; j < R; ++j) fxg_rows_read<NW>(smem, ((u32)j * FXG_ROWS_T + lane) * stride, b[j]);
            FXG_WAVE_SYNC();
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const u32 keep = placed ? p_info[j] >> 31 : 0u, olen = (p_info[j] >> 16) & 0x7FFFu, exb = p_info[j] & 0xFFFFu;
                if (keep) fxg_rows_pack<NW>(smem, exb, b[j], olen, fast);
            }
            FXG_WAVE_SYNC();
            if (placed) fxg_rows_flush(a.out_bases, bc[1], p_totb, smem, lane);
            FXG_WAVE_SYNC();
        }
        if (!havec) break;
#pragma unroll
        for (int j = 0; j < R; ++j) {
#pragma unroll
            for (int k = 0; k < NW; ++k) qp[j][k] = q[j][k];
            p_info[j] = c_info[j]; p_exc[j] = c_exc[j];
        }
        p_totb = c_totb; pend = cur;
        cur = (u32)__builtin_amdgcn_readfirstlane((int)nxt) * G + grp;
    }
    if (lane == 0) {
        if (t_in) atomicAdd(&a.tally[0], t_in);
        if (t_kept) atomicAdd(&a.tally[1], t_kept);
        if (t_bases) atomicAdd(&a.tally[2], t_bases);
        if (t_qtrim) atomicAdd(&a.tally[4 + FXG_R_QTRIM], t_qtrim);
        if (t_qfilter) atomicAdd(&a.tally[4 + FXG_R_QFILTER], t_qfilter);
    }
}
#endif  // FXG_HOST_EMULATION
