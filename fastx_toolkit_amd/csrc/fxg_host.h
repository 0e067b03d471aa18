// fxg_host.h -- what the translation units of the engine share on the host side: the context, the error macros, the launch of one instance of a tile
// kernel.  The engine is built from EIGHT translation units compiled side by side (fastx_toolkit_amd/build.py): fxg_engine.hip (every entry point of the
// C-ABI and every kernel but the clipper's) and fxg_engine_clip.hip seven times (-DFXG_CLIP_TU=1: the register forms of up to 16 columns and the general
// form; 2 / 4 / 6: 17..36 / 40..56 / 64..100 columns; 3 / 5 / 7: the same for adapters that contain N) -- the 64 clip instances are most of what hipcc spends its time on, and one translation unit of
// 100 kernels took three and a half minutes.  Compiled alone (no -DFXG_SPLIT: the variant / matrix / ablation builds of scripts/) fxg_engine.hip includes
// the clip file and is the one translation unit it used to be.
#pragma once
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>

#include "fxg_plan.h"
#include "fxg_rows.h"

struct FxgTextState;

struct fxg_ctx {
    int device;
    int cus;
    hipStream_t own_stream;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
#define FXG_KEV_RING 64
    hipEvent_t kev0[FXG_KEV_RING], kev1[FXG_KEV_RING];  // around the dominant kernel of the last FXG_KEV_RING launches when profiling
    int profiling;
    u64 kev_count;          // profiled launches since fxg_set_profiling(1)
    int kev_ready;          // every event of both rings exists
    u64 *status;            // FXG_STATUS_WORDS(status_cap) granules: tile totals [cap], prefixes [2 * cap], batch bases
    size_t status_cap;      // in tiles
    u32 epoch;              // tag of the granules of the current launch (1..255); 0 = never valid
    void *attr_kernel[8];   // kernels whose launch attributes were set, with the LDS size and the occupancy answer
    u32 attr_lds[8];
    int attr_per_cu[8];
    int attr_next;          // next slot to evict
    void *lds_kernel[32];   // largest MaxDynamicSharedMemorySize set per kernel (the attribute is only ever raised)
    u32 lds_max[32];
    int env_blocks_per_cu, env_ticket_groups;   // tuning knobs, read once
    int env_workers, env_nscan;                 // test knobs: cap on the worker workgroups of a launch / forced number of scanner waves (fxg_kernel_rows)
    u32 *errflag;           // [0] device error bits; tile dispensers start at word FXG_TICKET_STRIDE
    u64 *counters_scratch;  // used when the caller passes no counter block
    u64 *text_ws;           // newline census / scan levels / format items
    size_t text_ws_cap;     // in u64 words
    FxgTextState *text_state;
    // clip history (fxg_set_clip_history): the reference aligner's query buffer, carried from batch to batch
    int hist_on;
    u32 hist_wcap;          // host-side upper bound of the buffer width (the exact width lives on the device)
    int hist_cur;           // which of hist_buf[2] / hist_w[2] is current
    uint8_t *hist_buf[2];
    u32 *hist_w;            // [2]
    uint8_t *hist_ws;       // M, BT, ext, wlen
    size_t hist_ws_cap;
    float *clip_ck;         // fxg_clip_two_pass_k: checkpoint scratch, FxgPlan.ck_per_wg floats per workgroup
    size_t clip_ck_cap;     // in floats
    u32 *stats_ws;          // fxg_run_quality_stats: one u32 partial histogram per workgroup
    size_t stats_ws_cap;
    char err[512];
    char last_kernel[96];
    u32 last_grid, last_block, last_lds, last_tile;
    // the last compacting launch, kept so that it can be done again without the scanner when its waits timed out (fxg_fallback.h)
    struct { FxgKArgs ka; u64 *counters; int valid; } fb;                                      // the launch (fxg_launch_tiles): its arguments as the kernel got them
    struct { fxg_batch in; fxg_params p; fxg_out out; bool hist; u32 estride; int valid; } fb_req;      // the request (fxg_run_pipeline)
    u64 *fb_blk; size_t fb_blk_cap;     // block sums / prefixes of the fallback
    int recoveries;                     // launches redone that way since the context was made (fxg_scan_recoveries)
    int test_force_timeout;             // FXG_TEST_SCAN_TIMEOUT=1: every compacting launch starts with the time-out flag up (GPU tier)
};

static int fxg_fail(fxg_ctx *ctx, int code, const char *fmt, ...) __attribute__((unused));
static int fxg_fail(fxg_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

#define FXG_HIP(ctx, call)                                                                              \
    do {                                                                                                \
        hipError_t e__ = (call);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fxg_fail(ctx, FXG_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

// the -v report counters of a launch: the tile kernel tallied them, one tiny kernel lays them out (defined with that kernel, in fxg_engine.hip)
#define FXG_INTERNAL __attribute__((visibility("hidden")))      /* between the engine's translation units; not part of the C-ABI */
FXG_INTERNAL int fxg_enqueue_finish_counters(fxg_ctx *c, const FxgKArgs &ka, u64 *counters);
// the clip instances, by translation unit (fxg_engine_clip.hip)
FXG_INTERNAL int fxg_launch_clip_reg(fxg_ctx *c, FxgPlan &pl, u64 *ctr);
FXG_INTERNAL int fxg_launch_clip_k(fxg_ctx *c, FxgPlan &pl, u64 *ctr);
FXG_INTERNAL int fxg_launch_clip_k_wide(fxg_ctx *c, FxgPlan &pl, u64 *ctr);
FXG_INTERNAL int fxg_launch_clip_k_wide_wide(fxg_ctx *c, FxgPlan &pl, u64 *ctr);
FXG_INTERNAL int fxg_launch_clip_n(fxg_ctx *c, FxgPlan &pl, u64 *ctr);
FXG_INTERNAL int fxg_launch_clip_n_wide(fxg_ctx *c, FxgPlan &pl, u64 *ctr);
FXG_INTERNAL int fxg_launch_clip_n_wide_wide(fxg_ctx *c, FxgPlan &pl, u64 *ctr);

// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------

// dynamic-LDS attribute and occupancy of a kernel: asked once per (kernel, LDS size), not per launch
template <typename K>
static int fxg_kernel_fit(fxg_ctx *c, K kernel, const char *kname, u32 lds, int *per_cu, u32 block = FXG_TBLOCK)
{
    for (int i = 0; i < 8; ++i)
        if (c->attr_kernel[i] == (void *)kernel && c->attr_lds[i] == lds) { *per_cu = c->attr_per_cu[i]; return FXG_OK; }
    // the attribute is a property of the kernel, not of the cache entry: one kernel alternating between LDS sizes must never be
    // launched with more than was last set, so it is only ever raised
    int k = -1;
    for (int i = 0; i < 32; ++i) { if (c->lds_kernel[i] == (void *)kernel) { k = i; break; } if (!c->lds_kernel[i] && k < 0) k = i; }
    if (k < 0 || c->lds_kernel[k] != (void *)kernel || c->lds_max[k] < lds) {
        FXG_HIP(c, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (k >= 0) { c->lds_kernel[k] = (void *)kernel; c->lds_max[k] = lds; }
    }
    FXG_HIP(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kernel, (int)block, lds));
    if (*per_cu < 1) return fxg_fail(c, FXG_E_INVALID, "%s does not fit on a CU (lds=%u)", kname, lds);
    int slot = -1;
    for (int i = 0; i < 8; ++i) if (!c->attr_kernel[i]) { slot = i; break; }
    if (slot < 0) { slot = c->attr_next; c->attr_next = (c->attr_next + 1) % 8; }
    c->attr_kernel[slot] = (void *)kernel; c->attr_lds[slot] = lds; c->attr_per_cu[slot] = *per_cu;
    return FXG_OK;
}

// u64 words of the inter-workgroup state for `cap` tiles: totals, two prefixes per tile, two per scanner batch (batches of >= 256 tiles)
#define FXG_STATUS_WORDS(cap) (3 * (size_t)(cap) + 2 * ((size_t)(cap) / 256 + 2))

template <typename K>
static int fxg_launch_tiles(fxg_ctx *c, K kernel, const char *kname, FxgKArgs &ka, u32 lds, u64 *counters, u32 block = FXG_TBLOCK, bool rows_kernel = false, u64 ck_per_wg = 0)
{
    FXG_HIP(c, hipSetDevice(c->device));
    int per_cu = 0;
    const int frc = fxg_kernel_fit(c, kernel, kname, lds, &per_cu, block);
    if (frc != FXG_OK) return frc;
    // Tiles are dispensed by ticket, so nothing depends on every workgroup being resident: fill the chip.
    const int most = block == 64u ? 16 : 8;      // single-wave workgroups (fxg_rows.h, the two-pass clip instances): the LDS allows sixteen per CU
    int use = per_cu > most ? most : per_cu;
    if (c->env_blocks_per_cu > 0) use = c->env_blocks_per_cu;
    u64 workers = (u64)c->cus * (u64)use;
    if (c->env_workers > 0 && workers > (u64)c->env_workers) workers = (u64)c->env_workers;
    if (workers > ka.ntiles) workers = ka.ntiles;
    if (workers < 1) workers = 1;
    // more workgroups: the scanner(s) (fxg_device.h); fxg_kernel_rows runs several once there is work for them
    ka.nscan = !ka.compact ? 0u : (rows_kernel && workers >= 64u * FXG_ROWS_NSCAN ? (u32)FXG_ROWS_NSCAN : 1u);
    if (ka.compact && rows_kernel && c->env_nscan > 0) ka.nscan = (u32)c->env_nscan;
    const u64 grid = workers + ka.nscan;

    if (ka.compact) {
        bool fresh = false;
        if (c->status_cap < ka.ntiles) {
            (void)hipFree(c->status);
            c->status = nullptr; c->status_cap = 0;
            size_t cap = (size_t)ka.ntiles + (size_t)ka.ntiles / 4 + 1024;
            FXG_HIP(c, hipMalloc((void **)&c->status, FXG_STATUS_WORDS(cap) * sizeof(u64)));
            c->status_cap = cap;
            fresh = true;
        }
        // granules carry the launch's epoch, so the arrays are only cleared when they are new or the 8-bit epoch wraps
        c->epoch = c->epoch >= 255u ? 1u : c->epoch + 1u;
        if (fresh || c->epoch == 1u) FXG_HIP(c, hipMemsetAsync(c->status, 0, FXG_STATUS_WORDS(c->status_cap) * sizeof(u64), c->stream));
        ka.agg = c->status;
        ka.pfx = c->status + c->status_cap;
        ka.bbase = c->status + 3 * c->status_cap;
        ka.tag = c->epoch;
    }
    ka.clip_ck = nullptr;
    if (ck_per_wg) {                             // checkpoint rows of the two-pass clipper (fxg_clip_two_pass_k): written and read by the same thread
        const size_t need = (size_t)grid * (size_t)ck_per_wg;
        if (c->clip_ck_cap < need) {
            (void)hipFree(c->clip_ck);
            c->clip_ck = nullptr; c->clip_ck_cap = 0;
            FXG_HIP(c, hipMalloc((void **)&c->clip_ck, need * sizeof(float)));
            c->clip_ck_cap = need;
        }
        ka.clip_ck = c->clip_ck;
    }
#if defined(FXG_ABLATION) || defined(FXG_DBG_BITS)
    { const char *dbg = getenv("FXG_DEBUG"); ka.debug = dbg ? (u32)atoi(dbg) : 0u; }
#endif
#ifdef FXG_CLIP_DEBUG   // debug builds only (scripts/debug/clip64_bisect.py): FXG_CLIP_DBG_WORDS words per read from fxg_clip_two_pass_k, appended to $FXG_CLIP_DEBUG_OUT
    u32 *clip_dbg = nullptr;
    ka.clip_dbg = nullptr;
    if (ck_per_wg && getenv("FXG_CLIP_DEBUG_OUT")) {
        FXG_HIP(c, hipMalloc((void **)&clip_dbg, (size_t)ka.n * FXG_CLIP_DBG_WORDS * 4));
        FXG_HIP(c, hipMemsetAsync(clip_dbg, 0xEE, (size_t)ka.n * FXG_CLIP_DBG_WORDS * 4, c->stream));
        ka.clip_dbg = clip_dbg;
    }
#endif
    ka.errflag = c->errflag;                     // control block (zeroed before every launch), layout at FXG_CTRL_WORDS
    ka.ticket = c->errflag + FXG_CTRL_WORDS;
    ka.extra = (u64 *)(c->errflag + 2);
    ka.role = c->errflag + 8;
    ka.tally = (u64 *)(c->errflag + 32);
    // Dispenser g serves the workgroups with blockIdx % groups == g, so every group needs a worker even if the scanner role
    // falls to its members: eight groups only when each has more workgroups than there are scanners.
    { u32 g = c->env_ticket_groups > 0 ? (u32)c->env_ticket_groups : FXG_TICKET_GROUPS; ka.ticket_groups = grid >= (u64)(ka.nscan + 1u) * g ? g : 1u; }
    FXG_HIP(c, hipMemsetAsync(c->errflag, 0, (FXG_CTRL_WORDS + FXG_TICKET_GROUPS * FXG_TICKET_STRIDE) * sizeof(u32), c->stream));
    c->fb.valid = 0;
    if (ka.compact) {                            // what fxg_read_counters needs to do this launch again should its waits time out
        c->fb.ka = ka; c->fb.counters = counters; c->fb.valid = 1;
        if (c->test_force_timeout) {             // "somebody already gave up": every wait of this launch that lasts ends without a result
            static const u32 up = FXG_DEV_ERR_SCAN_TIMEOUT;
            FXG_HIP(c, hipMemcpyAsync(c->errflag, &up, sizeof up, hipMemcpyHostToDevice, c->stream));
        }
    }

    if (c->profiling) FXG_HIP(c, hipEventRecord(c->kev0[c->kev_count % FXG_KEV_RING], c->stream));
    hipLaunchKernelGGL(kernel, dim3((u32)grid), dim3(block), lds, c->stream, ka);
    FXG_HIP(c, hipGetLastError());
    if (c->profiling) { FXG_HIP(c, hipEventRecord(c->kev1[c->kev_count % FXG_KEV_RING], c->stream)); c->kev_count++; }
    { const int frc2 = fxg_enqueue_finish_counters(c, ka, counters); if (frc2 != FXG_OK) return frc2; }      // -v report counters: the tile kernel tallied them; one tiny kernel lays them out
    snprintf(c->last_kernel, sizeof c->last_kernel, "%s", kname);
    c->last_grid = (u32)grid; c->last_block = block; c->last_lds = lds; c->last_tile = ka.tile_reads;
#ifdef FXG_CLIP_DEBUG
    if (clip_dbg) {
        u32 *h = (u32 *)malloc((size_t)ka.n * FXG_CLIP_DBG_WORDS * 4);
        FXG_HIP(c, hipStreamSynchronize(c->stream));
        FXG_HIP(c, hipMemcpy(h, clip_dbg, (size_t)ka.n * FXG_CLIP_DBG_WORDS * 4, hipMemcpyDeviceToHost));
        FILE *f = fopen(getenv("FXG_CLIP_DEBUG_OUT"), "ab");
        if (f) { fwrite(h, FXG_CLIP_DBG_WORDS * 4, (size_t)ka.n, f); fclose(f); }
        free(h);
        (void)hipFree(clip_dbg);
    }
#endif
    return FXG_OK;
}
