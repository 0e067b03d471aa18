// fxg_kernels.h -- the persistent tile kernels of the engine.
//
// One workgroup (256 threads = 4 wave64) owns a tile of up to 256 consecutive reads and walks
// tiles tile0, tile0+grid, ... (persistent grid, every workgroup co-resident).  Per tile:
//   1. stream the quality rows once, 16 B per lane, and reduce them to two LDS bitmaps
//      (bit = "byte >= trim threshold", bit = "byte < filter threshold")           [HBM read: L bytes/read]
//   2. one thread per read: (clip DP over LDS-staged bases,) trim point = highest set bit,
//      filter count = popcount over the trimmed prefix -> keep / new length -> res[]  [HBM write: 4 B/read]
//   3. workgroup scan of (keep, new_len) + decoupled look-back across tiles -> output offsets
//   4. order-preserving gather of the kept prefixes into the packed output, one 16 B aligned output
//      chunk per lane                                            [HBM read <= 2L, write 2*new_len per kept read]
#pragma once
#include "fxg_device.h"

#define FXG_INVALID_TUPLE 0xFFFFFFFFu

// LDS carve-up shared by host (size) and device (pointers); every region is 16-byte aligned (G17)
struct FxgLds {
    u32 off_voff, off_vsrc, off_scratch, off_cacc, off_bm_g, off_bm_l, off_bases, total;
};
__host__ __device__ inline u32 fxg_r16(u32 x) { return (x + 15u) & ~15u; }
__host__ __device__ inline FxgLds fxg_lds_layout(u32 T, u32 stride, bool bitmaps, bool stage_bases)
{
    FxgLds l;
    u32 o = 0;
    l.off_voff = o;    o += fxg_r16((T + 1) * 4);
    l.off_vsrc = o;    o += fxg_r16(T * 4);
    l.off_scratch = o; o += fxg_r16(32 * 4);
    l.off_cacc = o;    o += fxg_r16(FXG_NCOUNTERS * 8);
    const u32 words = (T * stride + 31) / 32 + 2;
    l.off_bm_g = o;    o += bitmaps ? fxg_r16(words * 4) : 0;
    l.off_bm_l = o;    o += bitmaps ? fxg_r16(words * 4) : 0;
    l.off_bases = o;   o += stage_bases ? fxg_r16(T * stride + 16) : 0;
    l.total = o;
    return l;
}

// ------------------------------------------------------------------------------------------------
// fastx_clipper for one read: semi-global fp32 DP of read (query) x adapter (target) with the
// alignment path summary carried forward instead of a traceback matrix.
// Reference: sequence_alignment.cpp:340-428 (borders, cell rule, first-max), :496-604 (traceback
// counts), fastx_clipper.cpp:192-240 (accept rules), :282-319 (clip / discard cascade).
//   w0 = query_start<<16 | target_start<<8 | mismatches        w1 = path_len<<16 | matches
// ------------------------------------------------------------------------------------------------
template <int AMAX>
FXG_HD void fxg_clip_read(const FxgKArgs &a, const uint8_t *rd, int len,
                                              u32 *out_len, u32 *keep, u32 *reason, u32 *clipped, u32 *adapter_only)
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
    u32 bw0 = FXG_INVALID_TUPLE, bw1 = 0u;
    int bq = 0, first_n = len;
    for (int q = 0; q < len; ++q) {
        const u32 c = rd[q];
        const bool qn = (c == (u32)'N');
        if (qn && first_n == len) first_n = q;
        float dS = 0.0f, uS = 0.0f;                        // S[q-1][-1] and S[q][-1]: query_border = 0 (N1 for q == 0)
        u32 dW0 = FXG_INVALID_TUPLE, dW1 = 0u, uW0 = FXG_INVALID_TUPLE, uW1 = 0u;
#pragma unroll
        for (int t = 0; t < AMAX; ++t) {
            if (t >= A) break;
            const u32 tc = (u32)(uint8_t)a.adapter[t];
            const bool tn = (tc == (u32)'N');
            const bool eq = (c == tc);
            const bool neutral = qn || tn;
            const float pair = neutral ? ((qn && tn) ? 0.0f : 0.1f) : (eq ? 1.0f : -1.0f);   // sequence_alignment.h:157-169
            const float ul = dS + pair;
            const float up = uS + -5.0f;
            float left = S[t] + -5.0f;
            if (t > 3 && t - 3 > q) left = -100000.0f;      // :387-389
            float sc = ul; u32 w0 = dW0, w1 = dW1; bool diag = true;   // ul always beats the -1e8 seed
            if (up > sc)   { sc = up;   w0 = uW0;   w1 = uW1;   diag = false; }
            if (left > sc) { sc = left; w0 = W0[t]; w1 = W1[t]; diag = false; }
            if (w0 == FXG_INVALID_TUPLE) { w0 = ((u32)q << 16) | ((u32)t << 8); w1 = 0u; }   // path enters the matrix here
            w1 += 0x10000u + ((diag && !neutral && eq) ? 1u : 0u);
            w0 += (diag && !neutral && !eq) ? 1u : 0u;
            dS = S[t]; dW0 = W0[t]; dW1 = W1[t];
            S[t] = sc; W0[t] = w0; W1[t] = w1;
            uS = sc; uW0 = w0; uW1 = w1;
            if (sc > best) { best = sc; bw0 = w0; bw1 = w1; bq = q; }   // first maximum in query-major order (:421-425)
        }
    }
    const int qs = (int)(bw0 >> 16), ts = (int)((bw0 >> 8) & 0xFFu), mism = (int)(bw0 & 0xFFu);
    const int sz = (int)(bw1 >> 16), matches = (int)(bw1 & 0xFFFFu);
    int i = -1;
    if (sz != 0 && !(a.clip_min_adapter_len > 0 && sz < a.clip_min_adapter_len)) {
        if (bq == len - 1 && mism == 0) i = qs;
        else if (sz > 5 && ts == 0 && (matches * 100 / sz) >= 75) i = qs;
        else if (sz > 11 && (matches * 100 / sz) >= 80) i = qs;
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

// per-thread event counters, folded into the workgroup's partial[] row once at the end
struct FxgCounts {
    u32 in, kept, too_short, adapter_only, no_adapter, adapter_found, has_n, qtrim, qfilter, ftrim, clip_out, qtrim_out;
    u64 bases;
};

#ifndef FXG_HOST_EMULATION
__device__ __forceinline__ void fxg_flush_counts(const FxgKArgs &a, const FxgCounts &k, u64 *cacc)
{
    // cacc was zeroed at kernel start; LDS atomics, once per workgroup lifetime
    const u64 v[13] = {k.in, k.kept, k.bases, k.too_short, k.adapter_only, k.no_adapter, k.adapter_found,
                       k.has_n, k.qtrim, k.qfilter, k.ftrim, k.clip_out, k.qtrim_out};
#pragma unroll
    for (int i = 0; i < 13; ++i)
        if (v[i]) atomicAdd(&cacc[i], v[i]);
    __syncthreads();
    if (threadIdx.x < FXG_NCOUNTERS) a.partial[(u64)blockIdx.x * FXG_NCOUNTERS + threadIdx.x] = cacc[threadIdx.x];
}

#endif  // FXG_HOST_EMULATION

// ---- per-thread phase bodies (host+device so tests/emu can run them serially) ----

// phase 1: quality rows of the tile -> two bitmaps, 16 B per step
FXG_HD void fxg_phase_bitmaps(const FxgKArgs &a, u64 tb, u32 tbytes, u32 *bm_g, u32 *bm_l, u32 tid, u32 nthreads)
{
    const u32 Kg = (128u - a.tq) * 0x01010101u, Kf = (128u - a.fq) * 0x01010101u;
    const u32 nchunks = (tbytes + 15u) >> 4;
    uint16_t *g16 = reinterpret_cast<uint16_t *>(bm_g), *l16 = reinterpret_cast<uint16_t *>(bm_l);
    for (u32 c = tid; c < nchunks; c += nthreads) {
        const u32 o = c << 4;
        const u32x4 v = fxg_window(a.qual, (long long)(tb + o), a.total_bytes, 0, (int)(tbytes - o < 16u ? tbytes - o : 16u));
        g16[c] = (uint16_t)fxg_mask16(v, Kg);
        l16[c] = (uint16_t)(~fxg_mask16(v, Kf));
    }
}

// phase 1 (clipper only): bases rows of the tile -> LDS, so that one thread can walk one read
FXG_HD void fxg_phase_stage_bases(const FxgKArgs &a, u64 tb, u32 tbytes, uint8_t *sb, u32 tid, u32 nthreads)
{
    const u32 nchunks = (tbytes + 15u) >> 4;
    for (u32 c = tid; c < nchunks; c += nthreads) {
        const u32 o = c << 4;
        const u32x4 v = fxg_window(a.bases, (long long)(tb + o), a.total_bytes, 0, (int)(tbytes - o < 16u ? tbytes - o : 16u));
        *reinterpret_cast<u32x4 *>(sb + o) = v;
    }
}

// phase 2, group A: thread tid decides read r0 + tid
template <int AMAX>
FXG_HD void fxg_decide_a(const FxgKArgs &a, const u32 *bm_g, const u32 *bm_l, const uint8_t *sb, u32 r0, u32 tid,
                         FxgCounts &cnt, u32 *keep_out, u32 *len_out)
{
    const u32 stride = a.stride;
    const u32 rl = a.len ? (u32)a.len[r0 + tid] : a.fixed_len;
    u32 reason = FXG_R_KEPT, clipped = 0, keep = 1, curlen = rl;
    cnt.in++;
    if constexpr (AMAX > 0) {
        u32 ao;
        fxg_clip_read<AMAX>(a, sb + tid * stride, (int)rl, &curlen, &keep, &reason, &clipped, &ao);
        cnt.adapter_only += ao;
        cnt.too_short += (reason == FXG_R_CLIP_TOO_SHORT);
        cnt.no_adapter += (reason == FXG_R_CLIP_NO_ADAPTER);
        cnt.adapter_found += (reason == FXG_R_CLIP_ADAPTER_FOUND);
        cnt.has_n += (reason == FXG_R_CLIP_N);
        cnt.clip_out += keep;
    }
    if (keep && (a.stages & FXG_STAGE_QTRIM)) {             // fastq_quality_trimmer.c:94-101
        const u32 k = fxg_bits_last(bm_g, tid * stride, curlen);
        curlen = k;
        if (!(k > 0 && (int)k >= a.qt_min_len)) { keep = 0; reason = FXG_R_QTRIM; cnt.qtrim++; }
        else cnt.qtrim_out++;
    }
    if (keep && (a.stages & FXG_STAGE_QFILTER)) {           // fastq_quality_filter.c:110-129,155 in closed form
        const u32 low = fxg_bits_count(bm_l, tid * stride, curlen);
        int n0 = (int)curlen * a.qf_keep_pct / 100;
        if (n0 < 0) n0 = 0;
        if (a.qf_drop_all || (int)low > n0) { keep = 0; reason = FXG_R_QFILTER; cnt.qfilter++; }
    }
    a.res[r0 + tid] = (curlen & 0xFFFFu) | (keep << 16) | (reason << 17) | (clipped << 21);
    cnt.kept += keep;
    cnt.bases += keep ? curlen : 0u;
    *keep_out = keep; *len_out = curlen;
}

// phase 2, group B: fixed trimming is arithmetic on the length; reverse-complement only moves the anchor
template <bool REV>
FXG_HD void fxg_decide_b(const FxgKArgs &a, u32 r0, u32 tid, FxgCounts &cnt, u32 *keep_out, u32 *len_out, u32 *anchor_out)
{
    const u32 rl = a.len ? (u32)a.len[r0 + tid] : a.fixed_len;
    u32 reason = FXG_R_KEPT, start = 0, keep = 1, curlen = rl;
    cnt.in++;
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
    if (!keep) { reason = FXG_R_FTRIM; cnt.ftrim++; }
    a.res[r0 + tid] = (curlen & 0xFFFFu) | (keep << 16) | (reason << 17);
    cnt.kept += keep;
    cnt.bases += keep ? curlen : 0u;
    // output byte k of this read comes from source byte anchor + k (forward) or anchor - k (reverse-complement)
    *anchor_out = REV ? tid * a.stride + (rl - 1u - start) : tid * a.stride + start;
    *keep_out = keep; *len_out = curlen;
}

// per-kept-read side outputs
FXG_HD void fxg_write_kept_meta(const FxgKArgs &a, u64 rank, u32 olen, u32 read_index, u64 byte_off)
{
    if (a.out_len) a.out_len[rank] = (uint16_t)olen;
    if (a.kept_index) a.kept_index[rank] = read_index;
    if (a.out_off) a.out_off[rank] = byte_off;
}

#ifndef FXG_HOST_EMULATION   // everything below is device code proper (wave intrinsics, __global__)
// steps 3+4 shared by both kernels.  Called by every thread of the workgroup.
template <bool REV>
__device__ __forceinline__ void fxg_tile_emit(const FxgKArgs &a, unsigned char *smem, const FxgLds &L, u32 tile, u32 r0,
                                              u32 nreads, u64 tile_in_base, u32 keep, u32 olen, u32 src_anchor)
{
    u32 *v_off = reinterpret_cast<u32 *>(smem + L.off_voff);
    u32 *v_src = reinterpret_cast<u32 *>(smem + L.off_vsrc);
    u32 *scratch = reinterpret_cast<u32 *>(smem + L.off_scratch);
    const u32 tid = threadIdx.x;
    u32 exc, exb, totc, totb;
    fxg_block_scan2(keep, keep ? olen : 0u, scratch, &exc, &exb, &totc, &totb);
    if (!a.compact) return;
    if (tid < nreads) { v_off[tid] = exb; v_src[tid] = src_anchor; }
    if (tid == 0) v_off[nreads] = totb;
    u64 *bc = reinterpret_cast<u64 *>(scratch + 16);
    if (tid < 64) {
        u64 base_c, base_b;
        fxg_lookback(a, tile, totc, totb, &base_c, &base_b);
        if (tid == 0) { bc[0] = base_c; bc[1] = base_b; }
    }
    __syncthreads();
    const u64 base_c = bc[0], base_b = bc[1];
    if (keep) fxg_write_kept_meta(a, base_c + exc, olen, r0 + tid, base_b + exb);
    const u32 bad = fxg_tile_gather<REV>(a, v_off, v_src, nreads, tile_in_base, base_b, totb, tid, FXG_BLOCK);
    if (REV && bad) atomicOr(a.errflag, FXG_DEV_ERR_BAD_BASE);
}

// ------------------------------------------------------------------------------------------------
// group A: [fastx_clipper] -> [fastq_quality_trimmer] -> [fastq_quality_filter]
// ------------------------------------------------------------------------------------------------
template <int AMAX>
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_clip_qtrim_qfilter(const FxgKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 T = a.tile_reads, stride = a.stride, tid = threadIdx.x;
    const bool use_q = (a.stages & (FXG_STAGE_QTRIM | FXG_STAGE_QFILTER)) != 0;
    const FxgLds L = fxg_lds_layout(T, stride, use_q, AMAX > 0);
    u32 *bm_g = reinterpret_cast<u32 *>(smem + L.off_bm_g);
    u32 *bm_l = reinterpret_cast<u32 *>(smem + L.off_bm_l);
    uint8_t *sb = smem + L.off_bases;
    u64 *cacc = reinterpret_cast<u64 *>(smem + L.off_cacc);
    if (tid < FXG_NCOUNTERS) cacc[tid] = 0ull;
    FxgCounts cnt = {};

    for (u32 tile = fxg_first_tile(); tile < a.ntiles; tile += gridDim.x) {
        const u32 r0 = tile * T;
        const u64 left = a.n - (u64)r0;
        const u32 nreads = left < (u64)T ? (u32)left : T;
        const u64 tb = (u64)r0 * stride;
        const u32 tbytes = nreads * stride;

        if (use_q) fxg_phase_bitmaps(a, tb, tbytes, bm_g, bm_l, tid, FXG_BLOCK);
        if constexpr (AMAX > 0) fxg_phase_stage_bases(a, tb, tbytes, sb, tid, FXG_BLOCK);
        __syncthreads();

        u32 keep = 0, curlen = 0;
        if (tid < nreads) fxg_decide_a<AMAX>(a, bm_g, bm_l, sb, r0, tid, cnt, &keep, &curlen);

        fxg_tile_emit<false>(a, smem, L, tile, r0, nreads, tb, keep, curlen, tid * stride);
        __syncthreads();
    }
    fxg_flush_counts(a, cnt, cacc);
}

// ------------------------------------------------------------------------------------------------
// group B: [fastx_reverse_complement] -> [fastx_trimmer]   (pure index remap + complement)
// ------------------------------------------------------------------------------------------------
template <bool REV>
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_revcomp_ftrim(const FxgKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 T = a.tile_reads, stride = a.stride, tid = threadIdx.x;
    const FxgLds L = fxg_lds_layout(T, stride, false, false);
    u64 *cacc = reinterpret_cast<u64 *>(smem + L.off_cacc);
    if (tid < FXG_NCOUNTERS) cacc[tid] = 0ull;
    FxgCounts cnt = {};

    for (u32 tile = fxg_first_tile(); tile < a.ntiles; tile += gridDim.x) {
        const u32 r0 = tile * T;
        const u64 left = a.n - (u64)r0;
        const u32 nreads = left < (u64)T ? (u32)left : T;
        const u64 tb = (u64)r0 * stride;
        u32 keep = 0, curlen = 0, anchor = 0;
        if (tid < nreads) fxg_decide_b<REV>(a, r0, tid, cnt, &keep, &curlen, &anchor);
        fxg_tile_emit<REV>(a, smem, L, tile, r0, nreads, tb, keep, curlen, anchor);
        __syncthreads();
    }
    fxg_flush_counts(a, cnt, cacc);
}

// partial[grid][16] -> counters[16]; also folds the device error word into counters[FXG_C_ERRORS]
__global__ void fxg_kernel_reduce_counters(const u64 *partial, u32 rows, const u32 *errflag, u64 *counters)
{
    __shared__ u64 acc[FXG_NCOUNTERS];
    if (threadIdx.x < FXG_NCOUNTERS) acc[threadIdx.x] = 0ull;
    __syncthreads();
    const u32 col = threadIdx.x % FXG_NCOUNTERS;
    u64 s = 0;
    for (u32 r = threadIdx.x / FXG_NCOUNTERS; r < rows; r += blockDim.x / FXG_NCOUNTERS) s += partial[(u64)r * FXG_NCOUNTERS + col];
    if (s) atomicAdd(&acc[col], s);
    __syncthreads();
    if (threadIdx.x < FXG_NCOUNTERS) counters[threadIdx.x] = (threadIdx.x == FXG_C_ERRORS) ? (u64)*errflag : acc[threadIdx.x];
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
#endif  // FXG_HOST_EMULATION
