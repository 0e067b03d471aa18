// fxg_history.h -- the reference clipper's stale query tail, reproduced as a pre-pass over the batch.
//
// Reference behaviour (src/libfastx/sequence_alignment.cpp:135-136 resize_matrix never shrinks, :157/:375 every loop runs to
// matrix_width() = longest read so far, set_sequences assigns into the same std::string): read i is aligned as the string
//     Q'_i[x] = read_i[x]           x <  len_i
//               '\0'                x == len_i
//               B_{i-1}[x]          len_i < x < W_i         W_i = max(len_0 .. len_i)
// where B is the aligner's query buffer: every read j writes B[0 .. len_j] = read_j + '\0' and leaves the rest alone, so
// B_{i-1}[x] belongs to the LAST read j < i with len_j >= x.  "Last writer" is a prefix maximum over read indices, column by
// column, hence a scan:
//   fxg_kernel_hist_tiles  : M[tile][x] = 1 + max{ j in tile : len_j >= x }  (0 = none);  M[tile][stride+1] = max len in the tile
//   fxg_kernel_hist_blocks : exclusive prefix maximum of M inside blocks of FXG_HIST_BLOCK tiles, block totals to BT
//   fxg_kernel_hist_top    : exclusive prefix maximum of BT
//   fxg_kernel_hist_extend : walks the reads of a tile in order, column by column, and materialises Q'_i (ext, row stride
//                            estride) and W_i (wlen); the last tile also leaves the buffer B and W for the next batch
// The clip kernels then stage `ext` instead of `bases` and run the DP over W_i rows; everything after the DP uses the real
// length.  Per-thread bodies are host+device so tests/emu can run them serially.
#pragma once
#include "fxg_device.h"

#define FXG_HIST_BLOCK 256u          // tiles per scan block
#define FXG_HIST_CAP   (FXG_MAX_READ_LEN + 17u)   // bytes of one history buffer

struct FxgHist {
    const uint8_t  *bases;
    const uint16_t *len;             // null => every read has fixed_len
    u32  fixed_len, stride;
    u64  n;
    u32  tile_reads, ntiles;
    u32 *M;                          // [ntiles][stride + 2]
    u32 *BT;                         // [ceil(ntiles / FXG_HIST_BLOCK)][stride + 2]
    const uint8_t *hist_in;          // query buffer left by the previous batch (FXG_HIST_CAP bytes) and its width
    const u32     *w_in;
    uint8_t *hist_out;
    u32     *w_out;
    uint8_t *ext;                    // [n][estride]
    u32      estride;                // >= stride and >= every width that can occur
    uint16_t *wlen;                  // [n]
};

FXG_HD u32 fxg_hist_len(const FxgHist &h, u64 r) { return h.len ? (u32)h.len[r] : h.fixed_len; }
FXG_HD u32 fxg_umax(u32 a, u32 b) { return a > b ? a : b; }

// column x of tile `tile`: x <= stride: last writer (index + 1); x == stride + 1: longest read
FXG_HD void fxg_hist_tile_column(const FxgHist &h, u32 tile, u32 x)
{
    const u64 r0 = (u64)tile * h.tile_reads;
    const u64 left = h.n - r0;
    const u32 nreads = left < (u64)h.tile_reads ? (u32)left : h.tile_reads;
    u32 m = 0;
    for (u32 i = 0; i < nreads; ++i) {
        const u32 L = fxg_hist_len(h, r0 + i);
        if (x <= h.stride) { if (L >= x) m = (u32)(r0 + i) + 1u; }
        else m = fxg_umax(m, L);
    }
    h.M[(u64)tile * (h.stride + 2u) + x] = m;
}

FXG_HD void fxg_hist_block_column(const FxgHist &h, u32 blk, u32 x)
{
    const u32 S2 = h.stride + 2u;
    const u32 t0 = blk * FXG_HIST_BLOCK;
    const u32 t1 = t0 + FXG_HIST_BLOCK < h.ntiles ? t0 + FXG_HIST_BLOCK : h.ntiles;
    u32 run = 0;
    for (u32 t = t0; t < t1; ++t) {
        u32 *p = h.M + (u64)t * S2 + x;
        const u32 v = *p;
        *p = run;
        run = fxg_umax(run, v);
    }
    h.BT[(u64)blk * S2 + x] = run;
}

FXG_HD void fxg_hist_top_column(const FxgHist &h, u32 nblk, u32 x)
{
    const u32 S2 = h.stride + 2u;
    u32 run = 0;
    for (u32 b = 0; b < nblk; ++b) {
        u32 *p = h.BT + (u64)b * S2 + x;
        const u32 v = *p;
        *p = run;
        run = fxg_umax(run, v);
    }
}

// the byte the query buffer holds at column x when its last writer is `cur` (index + 1, 0 = nobody in this batch)
FXG_HD u32 fxg_hist_byte(const FxgHist &h, u32 cur, u32 x)
{
    if (cur == 0u) return h.hist_in[x];
    const u64 j = (u64)cur - 1u;
    return x < fxg_hist_len(h, j) ? (u32)h.bases[j * h.stride + x] : 0u;
}

// column x (0 <= x < max(estride, stride + 1)) of tile `tile`: ext / wlen rows of its reads; the last tile hands the buffer on
FXG_HD void fxg_hist_extend_column(const FxgHist &h, u32 tile, u32 x)
{
    const u32 S2 = h.stride + 2u, blk = tile / FXG_HIST_BLOCK;
    const u64 r0 = (u64)tile * h.tile_reads;
    const u64 left = h.n - r0;
    const u32 nreads = left < (u64)h.tile_reads ? (u32)left : h.tile_reads;
    const u32 *Mt = h.M + (u64)tile * S2, *Bt = h.BT + (u64)blk * S2;
    u32 cur = x <= h.stride ? fxg_umax(Mt[x], Bt[x]) : 0u;
    u32 W = fxg_umax(*h.w_in, fxg_umax(Mt[h.stride + 1u], Bt[h.stride + 1u]));
    for (u32 i = 0; i < nreads; ++i) {
        const u64 r = r0 + i;
        const u32 L = fxg_hist_len(h, r);
        W = fxg_umax(W, L);
        u32 ch;
        if (x <= L) { cur = (u32)r + 1u; ch = x < L ? (u32)h.bases[r * h.stride + x] : 0u; }
        else ch = fxg_hist_byte(h, cur, x);
        if (x < W && x < h.estride) h.ext[r * h.estride + x] = (uint8_t)ch;
        if (x == 0u) h.wlen[r] = (uint16_t)W;
    }
    if (tile == h.ntiles - 1u) {
        h.hist_out[x] = (uint8_t)fxg_hist_byte(h, cur, x);
        if (x == 0u) *h.w_out = W;
    }
}

FXG_HD u32 fxg_hist_columns(const FxgHist &h) { return h.estride > h.stride + 1u ? h.estride : h.stride + 1u; }

#ifndef FXG_HOST_EMULATION
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_hist_tiles(const FxgHist h)
{
    for (u32 x = threadIdx.x; x < h.stride + 2u; x += FXG_BLOCK) fxg_hist_tile_column(h, blockIdx.x, x);
}
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_hist_blocks(const FxgHist h)
{
    for (u32 x = threadIdx.x; x < h.stride + 2u; x += FXG_BLOCK) fxg_hist_block_column(h, blockIdx.x, x);
}
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_hist_top(const FxgHist h, u32 nblk)
{
    for (u32 x = blockIdx.x * FXG_BLOCK + threadIdx.x; x < h.stride + 2u; x += gridDim.x * FXG_BLOCK) fxg_hist_top_column(h, nblk, x);
}
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_hist_extend(const FxgHist h)
{
    const u32 ncol = fxg_hist_columns(h);
    for (u32 x = threadIdx.x; x < ncol; x += FXG_BLOCK) fxg_hist_extend_column(h, blockIdx.x, x);
}
// fixed-length batch that cannot see a stale tail (width so far <= L): only the buffer moves on
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_hist_fixed(const uint8_t *bases, u64 n, u32 L, u32 stride, const uint8_t *hist_in, const u32 *w_in,
                                                                   uint8_t *hist_out, u32 *w_out)
{
    const u32 x = blockIdx.x * FXG_BLOCK + threadIdx.x;
    if (x >= FXG_HIST_CAP) return;
    hist_out[x] = x < L ? bases[(n - 1u) * stride + x] : (x == L ? (uint8_t)0 : hist_in[x]);
    if (x == 0u) *w_out = *w_in > L ? *w_in : L;
}
#endif
