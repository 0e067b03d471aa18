// fxg_engine_clip.hip -- the clip instances of fxg_kernel_tiles, in three groups so that they can be compiled side by side (fxg_host.h).
// -DFXG_CLIP_TU=1 .. 7 compiles one group; without it (included by fxg_engine.hip in a single-unit build) all of them.
#include "fxg_host.h"

#define FXG_TILES_A(N) (fxg_kernel_tiles<N, 0>)
#define FXG_TILES_C(N) (pl.ka.clip_global ? fxg_kernel_tiles<N, 0, true> : fxg_kernel_tiles<N, 0, false>)      // packed clip instances: the DP over the staged tile, or over the batch (fxg_plan.h)

#if !defined(FXG_CLIP_TU) || FXG_CLIP_TU == 1
// adapters of up to 16 bases without N: two passes in registers; and the general two-word form (positive codes: FXG_NO_PACKED_CLIP, one-pass corner cases)
int fxg_launch_clip_reg(fxg_ctx *c, FxgPlan &pl, u64 *ctr)
{
    switch (pl.amax) {
        case -4: return fxg_launch_tiles(c, FXG_TILES_C(-4), "fxg_kernel_tiles<-4,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, pl.block);
        case -8: return fxg_launch_tiles(c, FXG_TILES_C(-8), "fxg_kernel_tiles<-8,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, pl.block);
        case -9: return fxg_launch_tiles(c, FXG_TILES_C(-9), "fxg_kernel_tiles<-9,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, pl.block);
        case -10: return fxg_launch_tiles(c, FXG_TILES_C(-10), "fxg_kernel_tiles<-10,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, pl.block);
        case -11: return fxg_launch_tiles(c, FXG_TILES_C(-11), "fxg_kernel_tiles<-11,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, pl.block);
        case -12: return fxg_launch_tiles(c, FXG_TILES_C(-12), "fxg_kernel_tiles<-12,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, pl.block);
        case -13: return fxg_launch_tiles(c, FXG_TILES_C(-13), "fxg_kernel_tiles<-13,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, pl.block);
        case -14: return fxg_launch_tiles(c, FXG_TILES_C(-14), "fxg_kernel_tiles<-14,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, pl.block);
        case -15: return fxg_launch_tiles(c, FXG_TILES_C(-15), "fxg_kernel_tiles<-15,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, pl.block);
        case -16: return fxg_launch_tiles(c, FXG_TILES_C(-16), "fxg_kernel_tiles<-16,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, pl.block);
#ifdef FXG_CLIP_ONE_PASS     // (ablation build only: reads beyond 255 bases with a short adapter; the regular build's register form takes them)
        case -216: return fxg_launch_tiles(c, FXG_TILES_C(-216), "fxg_kernel_tiles<-216,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
#endif
        case 16: return fxg_launch_tiles(c, FXG_TILES_A(16), "fxg_kernel_tiles<16,0> clip[+qtrim+qfilter]", pl.ka, pl.lds, ctr);
        case 32: return fxg_launch_tiles(c, FXG_TILES_A(32), "fxg_kernel_tiles<32,0> clip[+qtrim+qfilter]", pl.ka, pl.lds, ctr);
        case 64: return fxg_launch_tiles(c, FXG_TILES_A(64), "fxg_kernel_tiles<64,0> clip[+qtrim+qfilter]", pl.ka, pl.lds, ctr);
        default: return fxg_launch_tiles(c, FXG_TILES_A(100), "fxg_kernel_tiles<100,0> clip[+qtrim+qfilter]", pl.ka, pl.lds, ctr);
    }
}
#endif

#if !defined(FXG_CLIP_TU) || FXG_CLIP_TU == 2
// 17..100 columns: the in-place row with one start field, checkpoints in global scratch (fxg_clip_two_pass_k)
int fxg_launch_clip_k(fxg_ctx *c, FxgPlan &pl, u64 *ctr)
{
    switch (pl.amax) {
        case -20: return fxg_launch_tiles(c, FXG_TILES_C(-20), "fxg_kernel_tiles<-20,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
        case -24: return fxg_launch_tiles(c, FXG_TILES_C(-24), "fxg_kernel_tiles<-24,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
        case -28: return fxg_launch_tiles(c, FXG_TILES_C(-28), "fxg_kernel_tiles<-28,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
        case -32: return fxg_launch_tiles(c, FXG_TILES_C(-32), "fxg_kernel_tiles<-32,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
        case -36: return fxg_launch_tiles(c, FXG_TILES_C(-36), "fxg_kernel_tiles<-36,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
    default: return fxg_launch_clip_k_wide(c, pl, ctr);
    }
}
#endif

#if !defined(FXG_CLIP_TU) || FXG_CLIP_TU == 4
// ... its instances of more than 36 columns (three and two waves per SIMD: the longest compiles)
int fxg_launch_clip_k_wide(fxg_ctx *c, FxgPlan &pl, u64 *ctr)
{
    switch (pl.amax) {
        case -40: return fxg_launch_tiles(c, FXG_TILES_C(-40), "fxg_kernel_tiles<-40,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
        case -48: return fxg_launch_tiles(c, FXG_TILES_C(-48), "fxg_kernel_tiles<-48,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
        case -56: return fxg_launch_tiles(c, FXG_TILES_C(-56), "fxg_kernel_tiles<-56,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
    default: return fxg_launch_clip_k_wide_wide(c, pl, ctr);
    }
}
#endif

#if !defined(FXG_CLIP_TU) || FXG_CLIP_TU == 6
// ... 64, 80 and 100 columns
int fxg_launch_clip_k_wide_wide(fxg_ctx *c, FxgPlan &pl, u64 *ctr)
{
    switch (pl.amax) {
        case -64: return fxg_launch_tiles(c, FXG_TILES_C(-64), "fxg_kernel_tiles<-64,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
        case -80: return fxg_launch_tiles(c, FXG_TILES_C(-80), "fxg_kernel_tiles<-80,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
        case -100: return fxg_launch_tiles(c, FXG_TILES_C(-100), "fxg_kernel_tiles<-100,0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
    default: return fxg_launch_clip_n(c, pl, ctr);      // (the buckets of round 6 live in the translation units the N instances left: below)
    }
}
#endif

// adapters that contain N: instances of their own only in builds without the pair table (-DFXG_NO_PTAB: A/B measurements); with it an N is one more column
// pattern of the table and the plan never asks for them (fxg_plan.h)
#ifndef FXG_NO_PTAB
#define FXG_K(N) case -N: return fxg_launch_tiles(c, FXG_TILES_C(-N), "fxg_kernel_tiles<-" #N ",0> clip(packed)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg)
#if !defined(FXG_CLIP_TU) || FXG_CLIP_TU == 3
int fxg_launch_clip_n(fxg_ctx *c, FxgPlan &pl, u64 *ctr) { switch (pl.amax) { FXG_K(44); FXG_K(52); default: return fxg_launch_clip_n_wide(c, pl, ctr); } }
#endif
#if !defined(FXG_CLIP_TU) || FXG_CLIP_TU == 5
int fxg_launch_clip_n_wide(fxg_ctx *c, FxgPlan &pl, u64 *ctr) { switch (pl.amax) { FXG_K(60); FXG_K(72); default: return fxg_launch_clip_n_wide_wide(c, pl, ctr); } }
#endif
#if !defined(FXG_CLIP_TU) || FXG_CLIP_TU == 7
int fxg_launch_clip_n_wide_wide(fxg_ctx *c, FxgPlan &pl, u64 *ctr) { switch (pl.amax) { FXG_K(88); default: return fxg_fail(c, FXG_E_INVALID, "no clip instance %d", pl.amax); } }
#endif
#undef FXG_K
#else
#if !defined(FXG_CLIP_TU) || FXG_CLIP_TU == 3
// adapters that contain N
int fxg_launch_clip_n(fxg_ctx *c, FxgPlan &pl, u64 *ctr)
{
    switch (pl.amax) {
        case -316: return fxg_launch_tiles(c, FXG_TILES_C(-316), "fxg_kernel_tiles<-316,0> clip(packed, N in the adapter)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
        case -324: return fxg_launch_tiles(c, FXG_TILES_C(-324), "fxg_kernel_tiles<-324,0> clip(packed, N in the adapter)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
        case -336: return fxg_launch_tiles(c, FXG_TILES_C(-336), "fxg_kernel_tiles<-336,0> clip(packed, N in the adapter)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
    default: return fxg_launch_clip_n_wide(c, pl, ctr);
    }
}
#endif

#if !defined(FXG_CLIP_TU) || FXG_CLIP_TU == 5
// ... of more than 36 columns
int fxg_launch_clip_n_wide(fxg_ctx *c, FxgPlan &pl, u64 *ctr)
{
    switch (pl.amax) {
        case -348: return fxg_launch_tiles(c, FXG_TILES_C(-348), "fxg_kernel_tiles<-348,0> clip(packed, N in the adapter)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
        case -356: return fxg_launch_tiles(c, FXG_TILES_C(-356), "fxg_kernel_tiles<-356,0> clip(packed, N in the adapter)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
    default: return fxg_launch_clip_n_wide_wide(c, pl, ctr);
    }
}
#endif

#if !defined(FXG_CLIP_TU) || FXG_CLIP_TU == 7
// ... 64, 80 and 100 columns
int fxg_launch_clip_n_wide_wide(fxg_ctx *c, FxgPlan &pl, u64 *ctr)
{
    switch (pl.amax) {
        case -364: return fxg_launch_tiles(c, FXG_TILES_C(-364), "fxg_kernel_tiles<-364,0> clip(packed, N in the adapter)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
        case -380: return fxg_launch_tiles(c, FXG_TILES_C(-380), "fxg_kernel_tiles<-380,0> clip(packed, N in the adapter)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
        case -400: return fxg_launch_tiles(c, FXG_TILES_C(-400), "fxg_kernel_tiles<-400,0> clip(packed, N in the adapter)[+qtrim+qfilter]", pl.ka, pl.lds, ctr, FXG_TBLOCK, false, pl.ck_per_wg);
    default: return fxg_fail(c, FXG_E_INVALID, "no clip instance %d", pl.amax);
    }
}
#endif
#endif      // FXG_NO_PTAB
#undef FXG_TILES_A
#undef FXG_TILES_C
