// fxg_fallback.h -- the compaction once more, without anything that waits (the output of a4..a11 is "the kept, transformed reads in input order").
//
// The tile kernels compact in one pass: a tile publishes its totals, a scanner workgroup turns totals into prefixes, the tile waits for its prefix
// (fxg_device.h).  Every wait is bounded -- two seconds without progress raise FXG_DEV_ERR_SCAN_TIMEOUT and the launch writes nothing more -- and by
// construction a wait only ends that way when the workgroups stop being scheduled: two tool processes of a pipe sharing one GPU, a debugger, a hung
// neighbour.  Until round 5 that cost the user the run.  Now the launch is done again in a form that cannot wait on another workgroup:
//   1. the same request planned as a decision-only pass (no packed outputs: no scanner, no prefix, every tile independent) -- res[] and the -v tallies;
//   2. per block of FXG_FB_BLOCK reads the kept reads and kept bytes (fxg_kernel_fb_sums), one workgroup turns the block sums into exclusive
//      prefixes (fxg_kernel_fb_scan);
//   3. every read copies itself to its place (fxg_kernel_fb_gather): the byte mapping is the tile kernels' own (fxg_gather_byte: forward slice,
//      reverse-complement from the anchor, masked bases), one thread per read.  Slow (uncoalesced) and rare.
// Three launches with nothing between them but stream order.  The host side is fxg_read_counters (fxg_engine.hip), where the time-out is first seen.
#pragma once

#define FXG_FB_BLOCK 1024u

struct FxgFbArgs {
    FxgKArgs ka;
    u64 *blk;               // [2 * nblk + 2]: kept reads, kept bytes of every block; exclusive prefixes after fxg_kernel_fb_scan; [2 nblk], [2 nblk + 1] the totals
    u32 nblk;
};

// (keep, output length, anchor) of read r from res[] and the folded parameters: fxg_decide_a / fxg_decide_b / fxg_decide_mask / fxg_decide_census
FXG_HD void fxg_fb_read(const FxgKArgs &a, u64 r, u32 *keep, u32 *olen, u32 *anchor_fwd, u32 *anchor_rev)
{
    const u32 w = a.res[r];
    const u32 rl = a.len ? (u32)a.len[r] : a.fixed_len;
    u32 start = 0u;
    if ((a.stages & FXG_STAGE_FTRIM) && a.ft_first != 1) start = (u32)a.ft_first - 1u;      // fastx_trimmer.c:126-134 (a dropped read has no output)
    *keep = (w >> 16) & 1u; *olen = w & 0xFFFFu;
    *anchor_fwd = start; *anchor_rev = rl - 1u - start;
}

template <typename T>
__device__ __forceinline__ T fxg_fb_block_scan(T v, T *sh, u32 tid)      // inclusive scan over the workgroup's FXG_FB_BLOCK threads
{
    sh[tid] = v;
    __syncthreads();
    for (u32 d = 1u; d < FXG_FB_BLOCK; d <<= 1) {
        const T x = tid >= d ? sh[tid - d] : (T)0;
        __syncthreads();
        sh[tid] += x;
        __syncthreads();
    }
    return sh[tid];
}

__global__ void __launch_bounds__(FXG_FB_BLOCK) fxg_kernel_fb_sums(FxgFbArgs f)
{
    __shared__ u64 sh[FXG_FB_BLOCK];
    const u32 tid = threadIdx.x;
    const u64 r = (u64)blockIdx.x * FXG_FB_BLOCK + tid;
    u32 keep = 0u, olen = 0u, af, ar;
    if (r < f.ka.n) fxg_fb_read(f.ka, r, &keep, &olen, &af, &ar);
    const u64 cnt = fxg_fb_block_scan<u64>((u64)keep, sh, tid);
    __syncthreads();
    const u64 byt = fxg_fb_block_scan<u64>(keep ? (u64)olen : 0ull, sh, tid);
    if (tid == FXG_FB_BLOCK - 1u) { f.blk[2 * (u64)blockIdx.x] = cnt; f.blk[2 * (u64)blockIdx.x + 1] = byt; }
}

__global__ void __launch_bounds__(FXG_FB_BLOCK) fxg_kernel_fb_scan(FxgFbArgs f)
{
    __shared__ u64 sh[FXG_FB_BLOCK];
    const u32 tid = threadIdx.x;
    const u64 per = ((u64)f.nblk + FXG_FB_BLOCK - 1u) / FXG_FB_BLOCK, b0 = per * tid, b1 = b0 + per < f.nblk ? b0 + per : f.nblk;
    for (int which = 0; which < 2; ++which) {
        u64 s = 0;
        for (u64 b = b0; b < b1; ++b) s += f.blk[2 * b + which];
        const u64 inc = fxg_fb_block_scan<u64>(s, sh, tid);
        u64 run = inc - s;
        for (u64 b = b0; b < b1; ++b) { const u64 v = f.blk[2 * b + which]; f.blk[2 * b + which] = run; run += v; }
        if (tid == FXG_FB_BLOCK - 1u) f.blk[2 * (u64)f.nblk + which] = inc;
        __syncthreads();
    }
}

template <bool REV, bool MASK>
__global__ void __launch_bounds__(FXG_FB_BLOCK) fxg_kernel_fb_gather(FxgFbArgs f)
{
    __shared__ u64 sh[FXG_FB_BLOCK];
    const FxgKArgs &a = f.ka;
    const u32 tid = threadIdx.x;
    const u64 r = (u64)blockIdx.x * FXG_FB_BLOCK + tid;
    u32 keep = 0u, olen = 0u, af = 0u, ar = 0u;
    if (r < a.n) fxg_fb_read(a, r, &keep, &olen, &af, &ar);
    const u64 cnt = fxg_fb_block_scan<u64>((u64)keep, sh, tid);
    __syncthreads();
    const u64 byt = fxg_fb_block_scan<u64>(keep ? (u64)olen : 0ull, sh, tid);
    if (!keep) return;
    const u64 rank = f.blk[2 * (u64)blockIdx.x] + cnt - 1u, off = f.blk[2 * (u64)blockIdx.x + 1] + byt - olen;
    const uint8_t *sb = a.bases + r * a.stride, *sq = a.qual ? a.qual + r * a.stride : nullptr;
    u32 bad = 0u;
    for (u32 j = 0; j < olen; ++j)
        fxg_gather_byte<REV, MASK>(a, sb, MASK || a.out_qual ? sq : nullptr, a.out_bases + off, (a.out_qual && sq) ? a.out_qual + off : nullptr, REV ? ar : af, j, j, &bad);
    if (REV && bad) atomicOr(a.errflag, FXG_DEV_ERR_BAD_BASE);
    fxg_write_kept_meta(a, rank, olen, (u32)r, off);
}

// error bits the gather raised (after the decision pass laid out its counters): OR-ed into the counter block
__global__ void fxg_kernel_fb_errors(const u32 *errflag, u64 *counters) { counters[FXG_C_ERRORS] |= (u64)errflag[0]; }
