// fxg_engine.hip -- host side of the C-ABI declared in include/fxg.h (gfx950 only).
#include "fxg_host.h"
#include "fxg_text.h"
#include "fxg_history.h"
#include "fxg_stats.h"
#include "fxg_fallback.h"

extern "C" int fxg_abi_version(void) { return FXG_ABI_VERSION; }

extern "C" int fxg_ctx_create(int device_id, fxg_ctx **out)
{
    if (!out) return FXG_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return FXG_E_HIP;
    fxg_ctx *c = (fxg_ctx *)calloc(1, sizeof(fxg_ctx));
    if (!c) return FXG_E_NOMEM;
    c->device = device_id;
    hipDeviceProp_t prop;
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&prop, device_id) != hipSuccess) { free(c); return FXG_E_HIP; }
    c->cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { free(c); return FXG_E_HIP; }
    c->stream = c->own_stream;
    { const char *e = getenv("FXG_BLOCKS_PER_CU"); c->env_blocks_per_cu = (e && atoi(e) > 0 && atoi(e) <= 16) ? atoi(e) : 0; }
    { const char *e = getenv("FXG_WORKERS"); c->env_workers = (e && atoi(e) > 0) ? atoi(e) : 0; }
    { const char *e = getenv("FXG_NSCAN"); c->env_nscan = (e && atoi(e) > 0 && atoi(e) <= 64) ? atoi(e) : 0; }
    { const char *e = getenv("FXG_TEST_SCAN_TIMEOUT"); c->test_force_timeout = (e && atoi(e) > 0) ? 1 : 0; }
    { const char *e = getenv("FXG_TICKET_GROUPS"); c->env_ticket_groups = (e && atoi(e) > 0 && atoi(e) <= FXG_TICKET_GROUPS) ? atoi(e) : 0; }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipMalloc((void **)&c->errflag, (FXG_CTRL_WORDS + FXG_TICKET_GROUPS * FXG_TICKET_STRIDE) * sizeof(u32)) != hipSuccess ||
        hipMalloc((void **)&c->counters_scratch, FXG_NCOUNTERS * sizeof(u64)) != hipSuccess) {
        free(c);
        return FXG_E_HIP;
    }
    *out = c;
    return FXG_OK;
}

extern "C" void fxg_ctx_destroy(fxg_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->status); (void)hipFree(c->errflag); (void)hipFree(c->counters_scratch);
    (void)hipFree(c->text_ws); (void)hipFree(c->text_state); (void)hipFree(c->fb_blk);
    (void)hipFree(c->hist_buf[0]); (void)hipFree(c->hist_buf[1]); (void)hipFree(c->hist_w); (void)hipFree(c->hist_ws); (void)hipFree(c->stats_ws); (void)hipFree(c->clip_ck);
    (void)hipEventDestroy(c->ev0); (void)hipEventDestroy(c->ev1);
    for (int i = 0; i < FXG_KEV_RING; ++i) { if (c->kev0[i]) (void)hipEventDestroy(c->kev0[i]); if (c->kev1[i]) (void)hipEventDestroy(c->kev1[i]); }
    (void)hipStreamDestroy(c->own_stream);
    free(c);
}

extern "C" const char *fxg_last_error(const fxg_ctx *c) { return c ? c->err : "null context"; }

extern "C" int fxg_set_stream(fxg_ctx *c, void *s)
{
    if (!c) return FXG_E_INVALID;
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return FXG_OK;
}

extern "C" int fxg_sync(fxg_ctx *c)
{
    if (!c) return FXG_E_INVALID;
    FXG_HIP(c, hipStreamSynchronize(c->stream));
    return FXG_OK;
}

extern "C" int fxg_device_info(fxg_ctx *c, int *cus, size_t *total_mem, char *name, size_t cap)
{
    if (!c) return FXG_E_INVALID;
    hipDeviceProp_t prop;
    FXG_HIP(c, hipGetDeviceProperties(&prop, c->device));
    if (cus) *cus = prop.multiProcessorCount;
    if (total_mem) *total_mem = prop.totalGlobalMem;
    if (name && cap) snprintf(name, cap, "%s (%s)", prop.name, prop.gcnArchName);
    return FXG_OK;
}

extern "C" int fxg_malloc_device(fxg_ctx *c, size_t bytes, void **p)
{
    if (!c || !p) return FXG_E_INVALID;
    FXG_HIP(c, hipSetDevice(c->device));
    FXG_HIP(c, hipMalloc(p, bytes ? bytes : 16));
    return FXG_OK;
}
extern "C" int fxg_free_device(fxg_ctx *c, void *p) { if (!c) return FXG_E_INVALID; FXG_HIP(c, hipFree(p)); return FXG_OK; }
extern "C" int fxg_malloc_host(fxg_ctx *c, size_t bytes, void **p)
{
    if (!c || !p) return FXG_E_INVALID;
    FXG_HIP(c, hipHostMalloc(p, bytes ? bytes : 16, hipHostMallocPortable));   // usable by every GPU of a multi-GPU run
    return FXG_OK;
}
extern "C" int fxg_free_host(fxg_ctx *c, void *p) { if (!c) return FXG_E_INVALID; FXG_HIP(c, hipHostFree(p)); return FXG_OK; }
extern "C" int fxg_memcpy_h2d(fxg_ctx *c, void *d, const void *s, size_t n)
{
    if (!c) return FXG_E_INVALID;
    FXG_HIP(c, hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, c->stream));
    return FXG_OK;
}
extern "C" int fxg_memcpy_d2h(fxg_ctx *c, void *d, const void *s, size_t n)
{
    if (!c) return FXG_E_INVALID;
    FXG_HIP(c, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, c->stream));
    return FXG_OK;
}
extern "C" int fxg_memset_device(fxg_ctx *c, void *d, int v, size_t n)
{
    if (!c) return FXG_E_INVALID;
    FXG_HIP(c, hipMemsetAsync(d, v, n, c->stream));
    return FXG_OK;
}

extern "C" int fxg_timer_start(fxg_ctx *c)
{
    if (!c) return FXG_E_INVALID;
    FXG_HIP(c, hipEventRecord(c->ev0, c->stream));
    return FXG_OK;
}
extern "C" int fxg_timer_stop(fxg_ctx *c, float *ms)
{
    if (!c || !ms) return FXG_E_INVALID;
    FXG_HIP(c, hipEventRecord(c->ev1, c->stream));
    FXG_HIP(c, hipEventSynchronize(c->ev1));
    FXG_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return FXG_OK;
}


int fxg_enqueue_finish_counters(fxg_ctx *c, const FxgKArgs &ka, u64 *counters)
{
    hipLaunchKernelGGL(fxg_kernel_finish_counters, dim3(1), dim3(64), 0, c->stream, (const u64 *)ka.tally, ka.stages, (const u32 *)c->errflag,
                       (const u64 *)(c->errflag + 2), counters ? counters : c->counters_scratch);
    FXG_HIP(c, hipGetLastError());
    return FXG_OK;
}


// ------------------------------------------------------------------------------------------------
// clip history: the stale tail of the reference aligner's query buffer (fxg_history.h)
// ------------------------------------------------------------------------------------------------
extern "C" int fxg_set_clip_history(fxg_ctx *c, int on)
{
    if (!c) return FXG_E_INVALID;
    FXG_HIP(c, hipSetDevice(c->device));
    if (on && !c->hist_buf[0]) {
        FXG_HIP(c, hipMalloc((void **)&c->hist_buf[0], FXG_HIST_CAP));
        FXG_HIP(c, hipMalloc((void **)&c->hist_buf[1], FXG_HIST_CAP));
        FXG_HIP(c, hipMalloc((void **)&c->hist_w, 2 * sizeof(u32)));
    }
    if (on) {                                               // a fresh aligner: empty buffer, width 0
        FXG_HIP(c, hipMemsetAsync(c->hist_buf[0], 0, FXG_HIST_CAP, c->stream));
        FXG_HIP(c, hipMemsetAsync(c->hist_buf[1], 0, FXG_HIST_CAP, c->stream));
        FXG_HIP(c, hipMemsetAsync(c->hist_w, 0, 2 * sizeof(u32), c->stream));
    }
    c->hist_on = on ? 1 : 0; c->hist_wcap = 0; c->hist_cur = 0;
    return FXG_OK;
}

// Builds the extended queries of one batch and moves the buffer on.  Returns FXG_OK with *use = 0 when the batch cannot see a
// stale tail (fixed length, nothing longer before it): the clip kernel then reads the batch itself.
static int fxg_hist_prepass(fxg_ctx *c, const fxg_batch *in, u32 T, u32 estride, FxgKArgs *ka, int *use)
{
    const u32 lmax = in->len ? in->stride : in->fixed_len;
    const int cur = c->hist_cur;
    *use = 0;
    if (!in->len && c->hist_wcap <= in->fixed_len) {
        hipLaunchKernelGGL(fxg_kernel_hist_fixed, dim3((FXG_HIST_CAP + FXG_BLOCK - 1) / FXG_BLOCK), dim3(FXG_BLOCK), 0, c->stream,
                           (const uint8_t *)in->bases, (u64)in->n, in->fixed_len, in->stride, (const uint8_t *)c->hist_buf[cur], (const u32 *)(c->hist_w + cur),
                           c->hist_buf[cur ^ 1], c->hist_w + (cur ^ 1));
        FXG_HIP(c, hipGetLastError());
    } else {
        const u32 S2 = in->stride + 2u;
        const u32 ntiles = (u32)((in->n + T - 1) / T), nblk = (ntiles + FXG_HIST_BLOCK - 1) / FXG_HIST_BLOCK;
        const size_t bM = (((size_t)ntiles * S2 * 4) + 255) & ~(size_t)255, bBT = (((size_t)nblk * S2 * 4) + 255) & ~(size_t)255;
        const size_t bExt = (((size_t)in->n * estride + 16) + 255) & ~(size_t)255, bW = (((size_t)in->n * 2) + 255) & ~(size_t)255;
        const size_t need = bM + bBT + bExt + bW;
        if (c->hist_ws_cap < need) {
            FXG_HIP(c, hipStreamSynchronize(c->stream));
            (void)hipFree(c->hist_ws);
            c->hist_ws = nullptr; c->hist_ws_cap = 0;
            FXG_HIP(c, hipMalloc((void **)&c->hist_ws, need + need / 8));
            c->hist_ws_cap = need + need / 8;
        }
        FxgHist h;
        h.bases = in->bases; h.len = in->len; h.fixed_len = in->fixed_len; h.stride = in->stride; h.n = in->n;
        h.tile_reads = T; h.ntiles = ntiles;
        h.M = (u32 *)c->hist_ws; h.BT = (u32 *)(c->hist_ws + bM);
        h.ext = c->hist_ws + bM + bBT; h.estride = estride; h.wlen = (uint16_t *)(c->hist_ws + bM + bBT + bExt);
        h.hist_in = c->hist_buf[cur]; h.w_in = c->hist_w + cur; h.hist_out = c->hist_buf[cur ^ 1]; h.w_out = c->hist_w + (cur ^ 1);
        hipLaunchKernelGGL(fxg_kernel_hist_tiles, dim3(ntiles), dim3(FXG_BLOCK), 0, c->stream, h);
        hipLaunchKernelGGL(fxg_kernel_hist_blocks, dim3(nblk), dim3(FXG_BLOCK), 0, c->stream, h);
        hipLaunchKernelGGL(fxg_kernel_hist_top, dim3((S2 + FXG_BLOCK - 1) / FXG_BLOCK), dim3(FXG_BLOCK), 0, c->stream, h, nblk);
        hipLaunchKernelGGL(fxg_kernel_hist_extend, dim3(ntiles), dim3(FXG_BLOCK), 0, c->stream, h);
        FXG_HIP(c, hipGetLastError());
        ka->clip_src = h.ext; ka->clip_stride = estride; ka->clip_total = (u64)in->n * estride; ka->wlen = h.wlen;
        *use = 1;
    }
    c->hist_cur = cur ^ 1;
    if (lmax > c->hist_wcap) c->hist_wcap = lmax;
    return FXG_OK;
}

static int fxg_launch_plan(fxg_ctx *c, FxgPlan &pl, u64 *ctr);

extern "C" int fxg_run_pipeline(fxg_ctx *c, const fxg_batch *in, const fxg_params *p, const fxg_out *out)
{
    if (!c || !in || !p || !out) return FXG_E_INVALID;
    FxgPlan pl;
    // clip history: the DP may have to run over rows as wide as anything seen so far
    const bool hist = c->hist_on && (p->stages & FXG_STAGE_CLIP) && in->n != 0;
    const u32 estride = hist && c->hist_wcap > in->stride ? c->hist_wcap : in->stride;
    const int rc = fxg_make_plan(in, p, out, &pl, c->err, sizeof c->err, hist ? estride : 0u);
    if (rc != FXG_OK) return rc;
    if (in->n == 0) {
        if (out->counters) FXG_HIP(c, hipMemsetAsync(out->counters, 0, FXG_NCOUNTERS * sizeof(u64), c->stream));
        return FXG_OK;
    }
    if (hist) {
        int use = 0;
        FXG_HIP(c, hipSetDevice(c->device));
        const int hrc = fxg_hist_prepass(c, in, pl.ka.tile_reads, estride, &pl.ka, &use);
        if (hrc != FXG_OK) return hrc;
        if (!use) { pl.ka.clip_src = in->bases; pl.ka.clip_stride = in->stride; pl.ka.clip_total = in->n * (u64)in->stride; pl.ka.wlen = nullptr; pl.lds = fxg_plan_lds(&pl); }
    }
    // what fxg_read_counters needs to do a compacting pass again, should its waits time out (fxg_fallback.h)
    c->fb_req.valid = 0;
    if (pl.ka.compact) { c->fb_req.in = *in; c->fb_req.p = *p; c->fb_req.out = *out; c->fb_req.hist = hist; c->fb_req.estride = estride; c->fb_req.valid = 1; }
    return fxg_launch_plan(c, pl, (u64 *)out->counters);
}

// the launch of a planned pass: which instance of which kernel family
static int fxg_launch_plan(fxg_ctx *c, FxgPlan &pl, u64 *ctr)
{
#define FXG_TILES_A(N) (fxg_kernel_tiles<N, 0>)
#define FXG_TILES_C(N) (pl.ka.clip_global ? fxg_kernel_tiles<N, 0, true> : fxg_kernel_tiles<N, 0, false>)      // packed clip instances: the DP over the staged tile, or over the batch (fxg_plan.h)
    if (pl.rows_nw) {       // rows of 80..152 bytes through the quality stages: one lane per read, rows in registers (fxg_rows.h)
        if (pl.rows_h == 2) {     // rows of 153..304 bytes: two lanes per read
            if (pl.rows_nw == 26) return fxg_launch_tiles(c, fxg_kernel_rows<26, 2>, "fxg_kernel_rows<26,2> qtrim+qfilter", pl.ka, pl.lds, ctr, 64u, true);
            return fxg_launch_tiles(c, fxg_kernel_rows<38, 2>, "fxg_kernel_rows<38,2> qtrim+qfilter", pl.ka, pl.lds, ctr, 64u, true);
        }
        switch (pl.rows_nw) {
        case 10: return fxg_launch_tiles(c, fxg_kernel_rows_multi<10, 4>, "fxg_kernel_rows_multi<10,4> qtrim+qfilter", pl.ka, pl.lds, ctr, 64u, true);      // rows of 28..40 bytes, four reads per lane
        case 14: return fxg_launch_tiles(c, fxg_kernel_rows_multi<14, 3>, "fxg_kernel_rows_multi<14,3> qtrim+qfilter", pl.ka, pl.lds, ctr, 64u, true);      // 41..56 bytes, three
        case 20: return fxg_launch_tiles(c, fxg_kernel_rows_multi<20, 2>, "fxg_kernel_rows_multi<20,2> qtrim+qfilter", pl.ka, pl.lds, ctr, 64u, true);      // 57..79 bytes, two
        case 26: return fxg_launch_tiles(c, fxg_kernel_rows<26>, "fxg_kernel_rows<26> qtrim+qfilter", pl.ka, pl.lds, ctr, 64u, true);
        default: return fxg_launch_tiles(c, fxg_kernel_rows<38>, "fxg_kernel_rows<38> qtrim+qfilter", pl.ka, pl.lds, ctr, 64u, true);
        }
    }
    if (pl.group_a) {
        if (pl.amax == 0) return fxg_launch_tiles(c, FXG_TILES_A(0), "fxg_kernel_tiles<0,0> qtrim+qfilter", pl.ka, pl.lds, ctr);
        // the clipper's instances live in translation units of their own (fxg_engine_clip.hip)
        return pl.amax <= -300 ? fxg_launch_clip_n(c, pl, ctr) : (pl.amax < -16 && pl.amax > -200) ? fxg_launch_clip_k(c, pl, ctr) : fxg_launch_clip_reg(c, pl, ctr);
    }
    if (pl.mask) return fxg_launch_tiles(c, fxg_kernel_tiles<0, 3>, "fxg_kernel_tiles<0,3> mask", pl.ka, pl.lds, ctr);
    if (pl.artifacts) return fxg_launch_tiles(c, fxg_kernel_tiles<0, 4>, "fxg_kernel_tiles<0,4> base census", pl.ka, pl.lds, ctr);
    if (pl.rev && pl.ka.rev_dw) return fxg_launch_tiles(c, fxg_kernel_tiles<0, 5>, "fxg_kernel_tiles<0,5> revcomp[+ftrim], dword-aligned windows", pl.ka, pl.lds, ctr);
    if (pl.rev) return fxg_launch_tiles(c, fxg_kernel_tiles<0, 2>, "fxg_kernel_tiles<0,2> revcomp[+ftrim]", pl.ka, pl.lds, ctr);
    return fxg_launch_tiles(c, fxg_kernel_tiles<0, 1>, "fxg_kernel_tiles<0,1> ftrim", pl.ka, pl.lds, ctr);
#undef FXG_TILES_A
}

extern "C" int fxg_run_quality_stats(fxg_ctx *c, const fxg_batch *in, uint64_t *d_hist, uint32_t hist_cols)
{
    if (!c || !in || !d_hist) return FXG_E_INVALID;
    if (!in->bases || in->stride == 0 || in->stride > FXG_MAX_READ_LEN || (!in->len && (in->fixed_len == 0 || in->fixed_len > in->stride)))
        return fxg_fail(c, FXG_E_INVALID, "quality_stats: bad batch (stride %u, fixed_len %u)", in->stride, in->fixed_len);
    if (hist_cols < in->stride) return fxg_fail(c, FXG_E_INVALID, "quality_stats: histogram has %u columns, batch stride is %u", hist_cols, in->stride);
    if (in->n == 0) return FXG_OK;
    FXG_HIP(c, hipSetDevice(c->device));
    FxgStatsArgs a;
    a.bases = in->bases; a.qual = in->qual; a.len = in->len; a.n = in->n; a.total_bytes = in->n * (u64)in->stride;
    a.fixed_len = in->fixed_len; a.stride = in->stride; a.hist = (u64 *)d_hist; a.hist_cols = hist_cols;
    // one workgroup per CU (its LDS holds the 100 KB block histogram); fewer when the batch is small
    u64 nwg = (in->n + 255) / 256;
    if (nwg > (u64)c->cus) nwg = (u64)c->cus;
    a.nwg = (u32)nwg;
    const size_t need = (size_t)a.nwg * FXG_QS_PART_WORDS * sizeof(u32);
    if (c->stats_ws_cap < need) {
        FXG_HIP(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->stats_ws);
        c->stats_ws = nullptr; c->stats_ws_cap = 0;
        FXG_HIP(c, hipMalloc((void **)&c->stats_ws, need));
        c->stats_ws_cap = need;
    }
    a.partial = c->stats_ws;
    a.round_robin = 1u;
    if (const char *e = getenv("FXG_QS_ROUND_ROBIN")) a.round_robin = (u32)strtoul(e, nullptr, 0);      // measurement knob (csrc/fxg_stats.h: FxgStatsArgs::round_robin): 0 = one static slice per workgroup, tested loads; 3 = the row-strip form where the piece form would run
    const u32 lds = FXG_QS_LDS_WORDS * sizeof(u32);
    FXG_HIP(c, hipFuncSetAttribute((const void *)fxg_kernel_quality_stats, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FXG_HIP(c, hipFuncSetAttribute((const void *)fxg_kernel_quality_stats_odd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const u32 nstrips = (in->stride + FXG_QS_STRIP - 1) / FXG_QS_STRIP;
    for (u32 s0 = 0; s0 < nstrips; s0 += FXG_QS_WAVES) {       // one pass per block of 160 columns (one pass for reads up to 160)
        a.strip0 = s0;
        const bool timed = c->profiling && s0 == 0;
        if (timed) FXG_HIP(c, hipEventRecord(c->kev0[c->kev_count % FXG_KEV_RING], c->stream));
        u32 pr = 0, pp = 0;
        // dense batches of odd length 17 .. 159: the piece form that keeps LDS block and counter half per byte, a kernel (and a register allocation) of its own
        if (a.round_robin == 1u && (a.fixed_len & 1u) && fxg_stats_piece_plan(a, &pr, &pp)) hipLaunchKernelGGL(fxg_kernel_quality_stats_odd, dim3(a.nwg), dim3(FXG_QS_TBLOCK), lds, c->stream, a);
        else hipLaunchKernelGGL(fxg_kernel_quality_stats, dim3(a.nwg), dim3(FXG_QS_TBLOCK), lds, c->stream, a);
        FXG_HIP(c, hipGetLastError());
        if (timed) { FXG_HIP(c, hipEventRecord(c->kev1[c->kev_count % FXG_KEV_RING], c->stream)); c->kev_count++; }
        hipLaunchKernelGGL(fxg_kernel_quality_stats_fold, dim3((FXG_QS_PART_WORDS + FXG_QS_FOLD_E - 1) / FXG_QS_FOLD_E), dim3(256), 0, c->stream, a);
        FXG_HIP(c, hipGetLastError());
    }
    snprintf(c->last_kernel, sizeof c->last_kernel, "fxg_kernel_quality_stats");
    c->last_grid = a.nwg; c->last_block = FXG_QS_TBLOCK; c->last_lds = lds; c->last_tile = 64u * FXG_QS_UNROLL;
    return FXG_OK;
}

static void fxg_params_default(fxg_params *p)
{
    memset(p, 0, sizeof *p);
    p->qoffset = 33;
    strcpy(p->adapter, "CCTTAAGG");   // fastx_clipper.cpp:68
    p->clip_min_len = 5;              // :69
    p->ft_first = 1;
    p->mask_min_quality = 10;         // fastq_masker.c:47-48
    p->mask_char = 'N';
}

extern "C" int fxg_run_qtrim_qfilter(fxg_ctx *c, const fxg_batch *in, int qoffset, int use_trim, int trim_threshold,
                                     int trim_min_len, int use_filter, int filter_min_quality, int filter_min_percent,
                                     const fxg_out *out)
{
    fxg_params p;
    fxg_params_default(&p);
    p.qoffset = qoffset;
    if (use_trim) { p.stages |= FXG_STAGE_QTRIM; p.qt_threshold = trim_threshold; p.qt_min_len = trim_min_len; }
    if (use_filter) { p.stages |= FXG_STAGE_QFILTER; p.qf_min_quality = filter_min_quality; p.qf_min_percent = filter_min_percent; }
    return fxg_run_pipeline(c, in, &p, out);
}

extern "C" int fxg_run_clip(fxg_ctx *c, const fxg_batch *in, const char *adapter, uint32_t min_len, int keep_delta,
                            int min_adapter_len, uint32_t clip_flags, const fxg_out *out)
{
    fxg_params p;
    fxg_params_default(&p);
    p.stages = FXG_STAGE_CLIP;
    if (adapter) { strncpy(p.adapter, adapter, sizeof p.adapter - 1); p.adapter[sizeof p.adapter - 1] = 0; }
    p.clip_min_len = min_len; p.clip_keep_delta = keep_delta; p.clip_min_adapter_len = min_adapter_len; p.clip_flags = clip_flags;
    return fxg_run_pipeline(c, in, &p, out);
}

extern "C" int fxg_run_revcomp_trim(fxg_ctx *c, const fxg_batch *in, int reverse_complement, int first_base, int last_base,
                                    const fxg_out *out)
{
    fxg_params p;
    fxg_params_default(&p);
    if (reverse_complement) p.stages |= FXG_STAGE_REVCOMP;
    if (first_base != 1 || last_base != 0 || !reverse_complement) { p.stages |= FXG_STAGE_FTRIM; p.ft_first = first_base; p.ft_last = last_base; }
    return fxg_run_pipeline(c, in, &p, out);
}

// The last compacting pass once more, in the form that cannot wait (fxg_fallback.h).  Everything goes behind the failed launch on the same stream.
static int fxg_redo_without_scanner(fxg_ctx *c)
{
    const FxgKArgs failed = c->fb.ka;
    u64 *const counters = c->fb.counters;
    // 1. the decisions: the same request planned without the packed outputs -- the kernel instance the plan picks for a decision-only pass has no scanner
    //    and no wait (the row-per-lane kernel of fxg_rows.h only exists in its compacting form).  A run with clip history keeps the extended queries the
    //    failed pass built: the aligner's buffer has moved on once for this batch and must not move again.
    fxg_out o2 = c->fb_req.out;
    o2.out_bases = o2.out_qual = nullptr; o2.out_len = nullptr; o2.kept_index = nullptr; o2.out_off = nullptr;
    FxgPlan pl;
    const int prc = fxg_make_plan(&c->fb_req.in, &c->fb_req.p, &o2, &pl, c->err, sizeof c->err, c->fb_req.hist ? c->fb_req.estride : 0u);
    if (prc != FXG_OK) return prc;
    if (c->fb_req.hist) { pl.ka.clip_src = failed.clip_src; pl.ka.clip_stride = failed.clip_stride; pl.ka.clip_total = failed.clip_total; pl.ka.wlen = failed.wlen; pl.lds = fxg_plan_lds(&pl); }
    FXG_HIP(c, hipSetDevice(c->device));
    const int lrc = fxg_launch_plan(c, pl, counters);
    if (lrc != FXG_OK) return lrc;
    // 2. + 3. block sums -> prefixes -> every kept read to its place
    FxgFbArgs f;
    f.ka = failed;                               // the arrays and folded parameters of the pass as it was asked for
    f.ka.errflag = c->errflag;
    f.nblk = (u32)((f.ka.n + FXG_FB_BLOCK - 1u) / FXG_FB_BLOCK);
    const size_t need = 2 * (size_t)f.nblk + 2;
    if (c->fb_blk_cap < need) {
        (void)hipFree(c->fb_blk);
        c->fb_blk = nullptr; c->fb_blk_cap = 0;
        FXG_HIP(c, hipMalloc((void **)&c->fb_blk, need * sizeof(u64)));
        c->fb_blk_cap = need;
    }
    f.blk = c->fb_blk;
    hipLaunchKernelGGL(fxg_kernel_fb_sums, dim3(f.nblk), dim3(FXG_FB_BLOCK), 0, c->stream, f);
    FXG_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(fxg_kernel_fb_scan, dim3(1), dim3(FXG_FB_BLOCK), 0, c->stream, f);
    FXG_HIP(c, hipGetLastError());
    if (f.ka.stages & FXG_STAGE_REVCOMP) hipLaunchKernelGGL((fxg_kernel_fb_gather<true, false>), dim3(f.nblk), dim3(FXG_FB_BLOCK), 0, c->stream, f);
    else if (f.ka.stages & FXG_STAGE_MASK) hipLaunchKernelGGL((fxg_kernel_fb_gather<false, true>), dim3(f.nblk), dim3(FXG_FB_BLOCK), 0, c->stream, f);
    else hipLaunchKernelGGL((fxg_kernel_fb_gather<false, false>), dim3(f.nblk), dim3(FXG_FB_BLOCK), 0, c->stream, f);
    FXG_HIP(c, hipGetLastError());
    if (f.ka.stages & FXG_STAGE_REVCOMP) {       // a base that has no complement: found by the gather, reported like the tile kernels do
        hipLaunchKernelGGL(fxg_kernel_fb_errors, dim3(1), dim3(1), 0, c->stream, (const u32 *)c->errflag, counters ? counters : c->counters_scratch);
        FXG_HIP(c, hipGetLastError());
    }
    c->recoveries++;
    return FXG_OK;
}

extern "C" int fxg_scan_recoveries(const fxg_ctx *c) { return c ? c->recoveries : 0; }

extern "C" int fxg_read_counters(fxg_ctx *c, const uint64_t *d_counters, uint64_t host[FXG_NCOUNTERS])
{
    if (!c || !host) return FXG_E_INVALID;
    const u64 *src = d_counters ? (const u64 *)d_counters : c->counters_scratch;
    FXG_HIP(c, hipMemcpyAsync(host, src, FXG_NCOUNTERS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    FXG_HIP(c, hipStreamSynchronize(c->stream));
    if ((host[FXG_C_ERRORS] & FXG_DEV_ERR_SCAN_TIMEOUT) && c->fb.valid && c->fb_req.valid && src == (c->fb.counters ? c->fb.counters : c->counters_scratch) && !getenv("FXG_NO_SCAN_FALLBACK")) {
        // the launch these counters belong to gave up waiting (its workgroups were not being scheduled): the same work again without anything that waits
        c->fb.valid = 0; c->fb_req.valid = 0;
        const int rc = fxg_redo_without_scanner(c);
        if (rc != FXG_OK) return rc;
        FXG_HIP(c, hipMemcpyAsync(host, src, FXG_NCOUNTERS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        FXG_HIP(c, hipStreamSynchronize(c->stream));
    }
    if (host[FXG_C_ERRORS] & FXG_DEV_ERR_SCAN_TIMEOUT) return fxg_fail(c, FXG_E_DEVICE, "device: look-back scan timed out");
    if (host[FXG_C_ERRORS] & FXG_DEV_ERR_BAD_BASE)
        return fxg_fail(c, FXG_E_DEVICE, "Invalid nucleotide value in reverse_complement_base()");   // fastx_reverse_complement.c:67-68
    return FXG_OK;
}

extern "C" int fxg_synth_generate(fxg_ctx *c, uint64_t seed, uint64_t first, uint64_t n, uint32_t L, int with_adapter,
                                  uint8_t *bases, uint8_t *qual, uint32_t stride)
{
    if (!c || !bases || L == 0 || stride < L) return FXG_E_INVALID;
    if ((((uintptr_t)bases | (uintptr_t)qual) & 15u) != 0) return fxg_fail(c, FXG_E_INVALID, "synth: buffers must be 16-byte aligned");
    if (n == 0) return FXG_OK;
    const u32 lds = 2 * fxg_r16(FXG_SYNTH_TILE * stride);
    if (lds > 160 * 1024) return fxg_fail(c, FXG_E_INVALID, "synth: stride %u too large", stride);
    FXG_HIP(c, hipSetDevice(c->device));
    FXG_HIP(c, hipFuncSetAttribute((const void *)fxg_kernel_synth, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const u64 blocks = (n + FXG_SYNTH_TILE - 1) / FXG_SYNTH_TILE;
    if (blocks > 0x7FFFFFFFull) return fxg_fail(c, FXG_E_INVALID, "synth: too many reads for one launch");
    hipLaunchKernelGGL(fxg_kernel_synth, dim3((u32)blocks), dim3(FXG_BLOCK), lds, c->stream, (u64)seed, (u64)first, (u64)n, L,
                       with_adapter, bases, qual, stride);
    FXG_HIP(c, hipGetLastError());
    return FXG_OK;
}

// ------------------------------------------------------------------------------------------------
// FASTQ text on the device
// ------------------------------------------------------------------------------------------------
static int fxg_text_reserve(fxg_ctx *c, size_t words)
{
    if (!c->text_state) FXG_HIP(c, hipMalloc((void **)&c->text_state, sizeof(FxgTextState)));
    if (c->text_ws_cap >= words) return FXG_OK;
    (void)hipFree(c->text_ws);
    c->text_ws = nullptr; c->text_ws_cap = 0;
    const size_t cap = words + words / 4 + 4096;
    FXG_HIP(c, hipMalloc((void **)&c->text_ws, cap * sizeof(u64)));
    c->text_ws_cap = cap;
    return FXG_OK;
}

// in-place exclusive scan of data[0..n); tmp must hold the block-sum levels (n/1024 + n/1024^2 + ... + 8 words)
static int fxg_scan_u64(fxg_ctx *c, u64 *data, u64 n, u64 *tmp)
{
    if (n == 0) return FXG_OK;
    const u64 nb = (n + FXG_SCAN_PER_BLOCK - 1) / FXG_SCAN_PER_BLOCK;
    hipLaunchKernelGGL(fxg_kernel_scan_blocks, dim3((u32)nb), dim3(FXG_BLOCK), 0, c->stream, data, n, tmp);
    FXG_HIP(c, hipGetLastError());
    if (nb > 1) {
        const int rc = fxg_scan_u64(c, tmp, nb, tmp + nb);
        if (rc != FXG_OK) return rc;
        hipLaunchKernelGGL(fxg_kernel_scan_add, dim3((u32)nb), dim3(FXG_BLOCK), 0, c->stream, data, n, (const u64 *)tmp);
        FXG_HIP(c, hipGetLastError());
    }
    return FXG_OK;
}

extern "C" int fxg_fastq_index(fxg_ctx *c, const uint8_t *d_text, uint64_t text_len, int at_eof, int lines_per_record, uint32_t *d_line,
                               uint64_t cap_lines, uint16_t *d_len, uint8_t *d_flags, fxg_text_info *info)
{
    if (!c || !d_text || !d_line || !d_len || !d_flags || !info || (lines_per_record != 4 && lines_per_record != 2)) return FXG_E_INVALID;
    memset(info, 0, sizeof *info);
    info->first_bad = 0xFFFFFFFFu;
    if (text_len == 0) return FXG_OK;
    if (text_len > 0xFFFFFFF0ull) return fxg_fail(c, FXG_E_INVALID, "text block too large (%llu bytes)", (unsigned long long)text_len);
    FXG_HIP(c, hipSetDevice(c->device));
    const u64 lpr = (u64)lines_per_record;
    const u64 nseg = (text_len + FXG_TEXT_SEG - 1) / FXG_TEXT_SEG;
    int rc = fxg_text_reserve(c, (size_t)(nseg + nseg / 512 + 4096));
    if (rc != FXG_OK) return rc;
    u32 *d_ls = d_line, *d_le = d_line + cap_lines;
    FxgTextState init;
    memset(&init, 0, sizeof init);
    init.min_len = 0xFFFFFFFFu; init.first_bad = 0xFFFFFFFFu;
    FXG_HIP(c, hipMemcpyAsync(c->text_state, &init, sizeof init, hipMemcpyHostToDevice, c->stream));
    u64 *seg = c->text_ws;
    hipLaunchKernelGGL(fxg_kernel_nl_count, dim3((u32)nseg), dim3(FXG_BLOCK), 0, c->stream, d_text, (u64)text_len, seg, c->text_state);
    FXG_HIP(c, hipGetLastError());
    // total newlines = last exclusive prefix + last count: keep the last count before the scan overwrites it
    u64 last_count = 0, last_off = 0;
    FXG_HIP(c, hipMemcpyAsync(&last_count, seg + (nseg - 1), sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    rc = fxg_scan_u64(c, seg, nseg, seg + nseg);
    if (rc != FXG_OK) return rc;
    FXG_HIP(c, hipMemcpyAsync(&last_off, seg + (nseg - 1), sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    hipLaunchKernelGGL(fxg_kernel_nl_scatter, dim3((u32)nseg), dim3(FXG_BLOCK), 0, c->stream, d_text, (u64)text_len, (const u64 *)seg, d_ls, d_le, (u64)cap_lines);
    FXG_HIP(c, hipGetLastError());
    FXG_HIP(c, hipStreamSynchronize(c->stream));
    const u64 lines = last_off + last_count;
    info->lines = lines;
    u64 n = lines / lpr;
    if (lpr * n + 1 > cap_lines) n = (cap_lines - 1) / lpr;
    info->records = n;
    if (n == 0) { if (at_eof && lines % lpr != 0) info->irregular |= FXG_TEXT_IRR_TAIL; return FXG_OK; }
    const u32 grid = (u32)((n + FXG_BLOCK - 1) / FXG_BLOCK);
    if (lines_per_record == 4) hipLaunchKernelGGL(fxg_kernel_text_records<4>, dim3(grid), dim3(FXG_BLOCK), 0, c->stream, d_text, (const u32 *)d_ls, d_le, n, d_len, d_flags, c->text_state);
    else hipLaunchKernelGGL(fxg_kernel_text_records<2>, dim3(grid), dim3(FXG_BLOCK), 0, c->stream, d_text, (const u32 *)d_ls, d_le, n, d_len, d_flags, c->text_state);
    FXG_HIP(c, hipGetLastError());
    FxgTextState st;
    u32 consumed = 0;
    FXG_HIP(c, hipMemcpyAsync(&st, c->text_state, sizeof st, hipMemcpyDeviceToHost, c->stream));
    FXG_HIP(c, hipMemcpyAsync(&consumed, d_ls + lpr * n, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    FXG_HIP(c, hipStreamSynchronize(c->stream));
    info->consumed = consumed;
    info->max_len = st.max_len; info->min_len = st.min_len;
    info->irregular = st.irregular;
    info->first_bad = st.first_bad;
    info->numeric_records = st.n_numeric;
    info->has_cr = st.has_cr;
    if (at_eof && (lines % lpr != 0 || consumed != text_len)) info->irregular |= FXG_TEXT_IRR_TAIL;
    return FXG_OK;
}

extern "C" int fxg_fastq_pack(fxg_ctx *c, const uint8_t *d_text, uint64_t text_len, int lines_per_record, const uint32_t *d_line, uint64_t cap_lines,
                              const uint8_t *d_flags, uint64_t n, uint32_t stride, int qoffset, uint8_t *d_bases, uint8_t *d_qual, uint32_t *irregular)
{
    if (!c || !d_text || !d_line || !d_flags || !d_bases || !irregular || stride == 0 || (lines_per_record != 4 && lines_per_record != 2)) return FXG_E_INVALID;
    *irregular = 0;
    if (n == 0) return FXG_OK;
    if ((((uintptr_t)d_bases | (uintptr_t)d_qual) & 15u) != 0) return fxg_fail(c, FXG_E_INVALID, "row arrays must be 16-byte aligned");
    if (lines_per_record == 2 && d_qual) return fxg_fail(c, FXG_E_INVALID, "FASTA records have no qualities");
    FXG_HIP(c, hipSetDevice(c->device));
    const u32 *d_ls = d_line, *d_le = d_line + cap_lines;
    const u64 nchunks = (n * (u64)stride + 15) >> 4;
    u64 grid = (nchunks + FXG_BLOCK - 1) / FXG_BLOCK;
    if (grid > 65536) grid = 65536;
    FXG_HIP(c, hipMemsetAsync(&c->text_state->irregular, 0, sizeof(u32), c->stream));
    if (lines_per_record == 4)
        hipLaunchKernelGGL((fxg_kernel_text_pack<false, 4>), dim3((u32)grid), dim3(FXG_BLOCK), 0, c->stream, d_text, (u64)text_len, d_ls, d_le, d_flags, (u64)n, stride, qoffset, d_bases, c->text_state);
    else
        hipLaunchKernelGGL((fxg_kernel_text_pack<false, 2>), dim3((u32)grid), dim3(FXG_BLOCK), 0, c->stream, d_text, (u64)text_len, d_ls, d_le, d_flags, (u64)n, stride, qoffset, d_bases, c->text_state);
    FXG_HIP(c, hipGetLastError());
    if (d_qual) {
        hipLaunchKernelGGL((fxg_kernel_text_pack<true, 4>), dim3((u32)grid), dim3(FXG_BLOCK), 0, c->stream, d_text, (u64)text_len, d_ls, d_le, d_flags, (u64)n, stride, qoffset, d_qual, c->text_state);
        FXG_HIP(c, hipGetLastError());
        // records with numeric quality lines (rare: one pass over the flags, the parse itself only where a flag is set)
        hipLaunchKernelGGL(fxg_kernel_text_numeric, dim3((u32)((n + FXG_BLOCK - 1) / FXG_BLOCK)), dim3(FXG_BLOCK), 0, c->stream, d_text, d_ls, d_le, d_flags, (u64)n, stride, d_qual);
        FXG_HIP(c, hipGetLastError());
    }
    FXG_HIP(c, hipMemcpyAsync(irregular, &c->text_state->irregular, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    FXG_HIP(c, hipStreamSynchronize(c->stream));
    return FXG_OK;
}

extern "C" int fxg_fastq_format(fxg_ctx *c, const uint8_t *d_text, int lines_per_record, const uint32_t *d_line, uint64_t cap_lines, const uint8_t *d_flags,
                                uint64_t n, const uint32_t *d_res, uint32_t fwd_start, int reverse, const uint8_t *d_pk_bases, const uint8_t *d_pk_qual,
                                const uint64_t *d_pk_off, const uint8_t *d_rows_qual, uint32_t stride, int qoffset, int out_fasta, uint8_t *d_out,
                                uint64_t *out_bytes)
{
    if (!c || !d_text || !d_line || !d_flags || !d_res || !d_out || !out_bytes || (lines_per_record != 4 && lines_per_record != 2)) return FXG_E_INVALID;
    *out_bytes = 0;
    if (n == 0) return FXG_OK;
    const bool fastq_out = lines_per_record == 4 && !out_fasta;
    if (d_pk_bases && (!d_pk_off || (fastq_out && !d_pk_qual))) return fxg_fail(c, FXG_E_INVALID, "packed output needs bases, out_off and (FASTQ) qual");
    if (fastq_out && !d_rows_qual) return fxg_fail(c, FXG_E_INVALID, "FASTQ output needs the batch's quality rows (numeric records are printed from them)");
    FXG_HIP(c, hipSetDevice(c->device));
    int rc = fxg_text_reserve(c, (size_t)(n + n / 512 + 4096));
    if (rc != FXG_OK) return rc;
    u64 *item = c->text_ws;
    FxgFormatArgs a;
    a.text = d_text; a.ls = d_line; a.le = d_line + cap_lines; a.res = d_res; a.flags = d_flags; a.item_scan = item; a.n = n;
    a.fwd_start = fwd_start; a.rev = reverse ? 1u : 0u; a.pk_bases = d_pk_bases; a.pk_qual = d_pk_qual; a.pk_off = (const u64 *)d_pk_off;
    a.rows_qual = d_rows_qual; a.stride = stride; a.qoffset = qoffset; a.out_fasta = out_fasta ? 1u : 0u; a.out = d_out;
    const u32 nb = (u32)((n + FXG_BLOCK - 1) / FXG_BLOCK);
    if (lines_per_record == 4) hipLaunchKernelGGL(fxg_kernel_text_sizes<4>, dim3(nb), dim3(FXG_BLOCK), 0, c->stream, a, item);
    else hipLaunchKernelGGL(fxg_kernel_text_sizes<2>, dim3(nb), dim3(FXG_BLOCK), 0, c->stream, a, item);
    FXG_HIP(c, hipGetLastError());
    u64 last_item = 0, last_scan = 0;
    FXG_HIP(c, hipMemcpyAsync(&last_item, item + (n - 1), sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    rc = fxg_scan_u64(c, item, n, item + n);
    if (rc != FXG_OK) return rc;
    FXG_HIP(c, hipMemcpyAsync(&last_scan, item + (n - 1), sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    const u32 fgrid = (u32)((n * 16 + FXG_BLOCK - 1) / FXG_BLOCK);
    if (lines_per_record == 4) hipLaunchKernelGGL(fxg_kernel_text_format<4>, dim3(fgrid), dim3(FXG_BLOCK), 0, c->stream, a);
    else hipLaunchKernelGGL(fxg_kernel_text_format<2>, dim3(fgrid), dim3(FXG_BLOCK), 0, c->stream, a);
    FXG_HIP(c, hipGetLastError());
    FXG_HIP(c, hipStreamSynchronize(c->stream));
    *out_bytes = ((last_scan + last_item) & ((1ull << 40) - 1ull));
    return FXG_OK;
}

extern "C" int fxg_fasta_weights(fxg_ctx *c, const uint8_t *d_text, const uint32_t *d_line, uint64_t cap_lines, uint64_t n, const uint32_t *d_res,
                                 uint64_t weighted[8])
{
    if (!c || !d_text || !d_line || !d_res || !weighted) return FXG_E_INVALID;
    memset(weighted, 0, 8 * sizeof(uint64_t));
    if (n == 0) return FXG_OK;
    FXG_HIP(c, hipSetDevice(c->device));
    if (!c->text_state) return fxg_fail(c, FXG_E_INVALID, "fxg_fasta_weights: index the block first");
    FXG_HIP(c, hipMemsetAsync(c->text_state->weighted, 0, sizeof c->text_state->weighted, c->stream));
    hipLaunchKernelGGL(fxg_kernel_text_weights, dim3((u32)((n + FXG_BLOCK - 1) / FXG_BLOCK)), dim3(FXG_BLOCK), 0, c->stream, d_text, d_line, d_line + cap_lines, d_res, (u64)n, c->text_state);
    FXG_HIP(c, hipGetLastError());
    FXG_HIP(c, hipMemcpyAsync(weighted, c->text_state->weighted, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    FXG_HIP(c, hipStreamSynchronize(c->stream));
    return FXG_OK;
}

extern "C" int fxg_host_register(fxg_ctx *c, void *ptr, size_t bytes)
{
    if (!c || !ptr) return FXG_E_INVALID;
    FXG_HIP(c, hipHostRegister(ptr, bytes, hipHostRegisterPortable));
    return FXG_OK;
}
extern "C" int fxg_host_unregister(fxg_ctx *c, void *ptr)
{
    if (!c || !ptr) return FXG_E_INVALID;
    FXG_HIP(c, hipHostUnregister(ptr));
    return FXG_OK;
}

// ------------------------------------------------------------------------------------------------
// several GPUs: shard ranges, the epilogue arithmetic and the concatenation (host only)
// ------------------------------------------------------------------------------------------------
extern "C" int fxg_device_numa_node(int device)
{
    char id[64] = {0}, path[128];
    if (hipDeviceGetPCIBusId(id, (int)sizeof id - 1, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *q = id; *q; ++q) if (*q >= 'A' && *q <= 'F') *q = (char)(*q - 'A' + 'a');       // sysfs names are lower case
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", id);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

extern "C" int fxg_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

extern "C" int fxg_concat_peer(fxg_ctx *dst, void *d_dst, uint64_t byte_off, fxg_ctx *src, const void *d_src, uint64_t bytes)
{
    if (!dst || !src || (!d_dst && bytes) || (!d_src && bytes)) return FXG_E_INVALID;
    if (!bytes) return FXG_OK;
    FXG_HIP(src, hipSetDevice(src->device));
    if (dst->device == src->device) FXG_HIP(src, hipMemcpyAsync((char *)d_dst + byte_off, d_src, bytes, hipMemcpyDeviceToDevice, src->stream));
    else FXG_HIP(src, hipMemcpyPeerAsync((char *)d_dst + byte_off, dst->device, d_src, src->device, bytes, src->stream));
    return FXG_OK;
}

// the multi-GPU host code (shard ranges, epilogue, concatenation, RCCL transport) is fxg_comm.h: it reaches device memory through
// these hooks only, so that the CPU tier compiles the very same code over host memory (tests/emu/fxg_stub.cpp)
#define FXG_COMM_FAIL(c, code, ...) fxg_fail(c, code, __VA_ARGS__)
#define FXG_COMM_SET_DEVICE(c) (hipSetDevice((c)->device) == hipSuccess)
#define FXG_COMM_MALLOC(pp, bytes) (hipMalloc((void **)(pp), (bytes)) == hipSuccess)
#define FXG_COMM_FREE(p) ((void)hipFree(p))
#define FXG_COMM_STREAM(c) ((void *)(c)->stream)
#define FXG_COMM_SCRATCH(c) ((const uint64_t *)(c)->counters_scratch)
static const char *fxg_comm_d2h_sync(fxg_ctx *c, void *dst, const void *src, size_t bytes)
{
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}
#include "fxg_comm.h"

extern "C" int fxg_set_profiling(fxg_ctx *c, int enabled)
{
    if (!c) return FXG_E_INVALID;
    if (enabled && !c->kev_ready) {                          // the event ring is made on first use: all 2 x FXG_KEV_RING events, or none
        FXG_HIP(c, hipSetDevice(c->device));
        hipError_t e = hipSuccess;
        for (int i = 0; i < FXG_KEV_RING && e == hipSuccess; ++i) { e = hipEventCreate(&c->kev0[i]); if (e == hipSuccess) e = hipEventCreate(&c->kev1[i]); }
        if (e != hipSuccess) {                               // a ring with holes would be recorded into and waited on through null events
            for (int i = 0; i < FXG_KEV_RING; ++i) {
                if (c->kev0[i]) (void)hipEventDestroy(c->kev0[i]);
                if (c->kev1[i]) (void)hipEventDestroy(c->kev1[i]);
                c->kev0[i] = c->kev1[i] = nullptr;
            }
            c->profiling = 0;
            return fxg_fail(c, FXG_E_HIP, "hipEventCreate failed: %s (profiling stays off)", hipGetErrorString(e));
        }
        c->kev_ready = 1;
    }
    c->profiling = enabled ? 1 : 0;
    c->kev_count = 0;
    return FXG_OK;
}

#if defined(FXG_ABLATION) || defined(FXG_ABL_ROWCLK)
// timing experiments only (scripts/ablate.py): the phase clocks fxg_kernel_rows adds up in the control block, words 10..
extern "C" int fxg_debug_phase_clocks(fxg_ctx *c, uint64_t out[11])
{
    if (!c || !out) return FXG_E_INVALID;
    FXG_HIP(c, hipStreamSynchronize(c->stream));
    FXG_HIP(c, hipMemcpy(out, c->errflag + 10, 11 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return FXG_OK;
}
#endif

extern "C" int fxg_last_kernel_ms(fxg_ctx *c, float *ms)
{
    if (!c || !ms) return FXG_E_INVALID;
    if (!c->kev_count) return fxg_fail(c, FXG_E_INVALID, "no profiled launch recorded");
    const int k = (int)((c->kev_count - 1) % FXG_KEV_RING);
    FXG_HIP(c, hipEventSynchronize(c->kev1[k]));
    FXG_HIP(c, hipEventElapsedTime(ms, c->kev0[k], c->kev1[k]));
    return FXG_OK;
}

extern "C" int fxg_profiled_kernel_ms(fxg_ctx *c, float *ms, uint32_t cap, uint32_t *n)
{
    if (!c || !ms || !n) return FXG_E_INVALID;
    const u64 have = c->kev_count < FXG_KEV_RING ? c->kev_count : FXG_KEV_RING;
    const u32 k = (u32)(have < cap ? have : cap);
    *n = 0;
    for (u32 i = 0; i < k; ++i) {                            // oldest first
        const int slot = (int)((c->kev_count - k + i) % FXG_KEV_RING);
        FXG_HIP(c, hipEventSynchronize(c->kev1[slot]));
        FXG_HIP(c, hipEventElapsedTime(&ms[i], c->kev0[slot], c->kev1[slot]));
    }
    *n = k;
    return FXG_OK;
}

extern "C" int fxg_last_launch_info(const fxg_ctx *c, char *name, size_t cap, uint32_t *grid, uint32_t *block, uint32_t *lds,
                                    uint32_t *tile)
{
    if (!c) return FXG_E_INVALID;
    if (name && cap) snprintf(name, cap, "%s", c->last_kernel);
    if (grid) *grid = c->last_grid;
    if (block) *block = c->last_block;
    if (lds) *lds = c->last_lds;
    if (tile) *tile = c->last_tile;
    return FXG_OK;
}

#ifndef FXG_SPLIT      // a single-unit build: the clip instances here as well
#include "fxg_engine_clip.hip"
#endif
