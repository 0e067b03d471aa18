/* fxh_lanes.c -- the device text path of the tools (fxh_priv.h). */
/* ---------------------------------------------------------------------------------------------- */
/* device text path: a block of FASTQ text is indexed, checked, packed, run through    */
/* the pipeline and formatted on a GPU.  Blocks are cut on the host at record boundaries (a record   */
/* is four lines counted from the start of the input, exactly as the reference reads them,           */
/* fastx.c:314-404) and dealt round-robin to LANES: one thread + one engine context (own stream,     */
/* own device buffers) each, FXH_LANES per GPU over the GPUs of FXG_DEVICES.  Lanes overlap one      */
/* another's upload, kernels and download; the main thread collects the blocks in input order, so    */
/* the output is the concatenation a single GPU would have produced.  A block that is irregular in   */
/* any way is only DETECTED on the device: it then goes through the host parser (fxh_host_block),    */
/* which owns the reference's messages and corner cases, at its turn in the output order.            */
/* ---------------------------------------------------------------------------------------------- */
#include "fxh_priv.h"
int g_parts_abort;
pthread_mutex_t g_first_ctx_mu = PTHREAD_MUTEX_INITIALIZER;
static int g_first_ctx_done;
int g_hip_touched;

/* A launch whose waits ran out is done again by the engine in a form that cannot wait (include/fxg.h: fxg_scan_recoveries): the run goes on, the user is
 * told once -- it means the GPU was not scheduling this process's work for seconds at a time */
void fxh_note_recoveries(fxh_state *st)
{
    static int told;
    const int now = fxg_scan_recoveries(st->ctx);
    if (now > st->recoveries_seen) {
        st->recoveries_seen = now;
        if (!__atomic_exchange_n(&told, 1, __ATOMIC_RELAXED))
            warnx("the GPU stopped running this process's work for a while (shared with another process?); a block was done again without the fast path's waits -- the output is not affected");
    }
}

void fxh_lane_run(fxh_lane *ln)
{
    fxh_state *st = &ln->st;
    const size_t len = ln->len;
    const int revcomp = ln->revcomp;
    ln->handled = 0; ln->out_len = 0; ln->fixed_len = 0;
    if (st->d_text_cap < len + 32) {
        if (st->d_text) { fxg_free_device(st->ctx, st->d_text); fxg_free_device(st->ctx, st->d_out_text); }
        st->d_text_cap = len + len / 8 + 4096;
        FXG_CHECK(st, fxg_malloc_device(st->ctx, st->d_text_cap, (void **)&st->d_text));
        /* the output can be longer than the input: an empty third line still gets its '+' (fastx.c:460), one byte per record */
        FXG_CHECK(st, fxg_malloc_device(st->ctx, st->d_text_cap + st->d_text_cap / 7 + 64, (void **)&st->d_out_text));
    }
    const int lpr = ln->lpr;
    const size_t cap_lines = len * 4 / 7 + 16;              /* the shortest records, "@\nA\n\nI\n" and ">\nA\n", have 1.75 / 2 bytes per line */
    if (st->d_ls_cap < cap_lines) {
        if (st->d_ls) { fxg_free_device(st->ctx, st->d_ls); fxg_free_device(st->ctx, st->d_len16); fxg_free_device(st->ctx, st->d_flags); }
        st->d_ls_cap = cap_lines + cap_lines / 8;
        FXG_CHECK(st, fxg_malloc_device(st->ctx, 2 * st->d_ls_cap * sizeof(uint32_t), (void **)&st->d_ls));
        FXG_CHECK(st, fxg_malloc_device(st->ctx, (st->d_ls_cap / 2 + 4) * sizeof(uint16_t), (void **)&st->d_len16));
        FXG_CHECK(st, fxg_malloc_device(st->ctx, st->d_ls_cap / 2 + 4, (void **)&st->d_flags));
    }
    if (ln->pinned && ln->text_base) {     /* page-lock the input buffer on first use, so that the upload is real DMA */
        struct fxh_pinned *pn = ln->pinned;
        int known = 0;
        pthread_mutex_lock(&pn->mu);
        for (int i = 0; i < pn->n; ++i) if (pn->ptr[i] == ln->text_base) known = 1;
        if (!known && pn->n < (int)(sizeof pn->ptr / sizeof pn->ptr[0])) { pn->ptr[pn->n++] = ln->text_base; pthread_mutex_unlock(&pn->mu); (void)fxg_host_register(st->ctx, ln->text_base, ln->text_cap); }
        else pthread_mutex_unlock(&pn->mu);
    }
    double tc = fxh_now(), tn;
#define FXH_TCALL(k) do { tn = fxh_now(); ln->t_call[k] += tn - tc; tc = tn; } while (0)
    FXG_CHECK(st, fxg_memcpy_h2d(st->ctx, st->d_text, ln->text, len));
    FXH_TCALL(0);
    fxg_text_info info;
    FXG_CHECK(st, fxg_fastq_index(st->ctx, st->d_text, len, 1, lpr, st->d_ls, st->d_ls_cap, st->d_len16, st->d_flags, &info));
    FXH_TCALL(1);
    if (info.irregular || info.records == 0 || info.consumed != len) return;
    if (ln->records == FXH_RECORDS_UNKNOWN) ln->records = info.records;      /* a chunk of the one-file sharded run: its cut is proven by being whole records (fxh_strands.c) */
    else if (info.records != ln->records) return;
    const uint64_t n = info.records;
    uint32_t stride = info.max_len;
    /* long reads of one length through the clipper: rows on dword boundaries, so that the clip kernel can read them where they are instead of staging
     * tiles of them in LDS (csrc/fxg_plan.h: clip_global -- 300 bases 34 % faster, 1 000 bases 2.2x) */
    if ((ln->p->stages & FXG_STAGE_CLIP) && info.min_len == info.max_len && stride > 160u) stride = (stride + 3u) & ~3u;
    ln->fixed_len = info.min_len == info.max_len ? info.max_len : 0u;
    if (ln->clip_guard && !ln->fixed_len) return;           /* ragged block of a clipper run: the one-aligner mode takes over at this block (fxh_clip_go_serial) */
    if ((uint64_t)n * stride > (uint64_t)8 * len + (1u << 20)) return;   /* ragged beyond reason: the host path handles it */
    fxh_grow_device(st, n, (size_t)n * stride + 16, revcomp);
    uint32_t irr = 0;
    FXG_CHECK(st, fxg_fastq_pack(st->ctx, st->d_text, len, lpr, st->d_ls, st->d_ls_cap, st->d_flags, n, stride, ln->qoffset, st->d_bases,
                                 ln->has_q ? st->d_qual : NULL, &irr));
    FXH_TCALL(2);
    if (irr) return;
    if (revcomp && st->d_off_cap < n) {
        if (st->d_out_off) fxg_free_device(st->ctx, st->d_out_off);
        st->d_off_cap = n + n / 8 + 1024;
        FXG_CHECK(st, fxg_malloc_device(st->ctx, st->d_off_cap * sizeof(uint64_t), (void **)&st->d_out_off));
    }
    const int fixed = info.min_len == info.max_len;
    fxg_batch in = {st->d_bases, ln->has_q ? st->d_qual : NULL, fixed ? NULL : st->d_len16, info.max_len, stride, n};
    fxg_out out = {st->d_res, revcomp ? st->d_out_bases : NULL, (revcomp && ln->has_q) ? st->d_out_qual : NULL, NULL, NULL, revcomp ? st->d_out_off : NULL, st->d_counters};
    fxg_params pp = *ln->p;
    pp.qoffset = 33;
    FXG_CHECK(st, fxg_run_pipeline(st->ctx, &in, &pp, &out));
    FXH_TCALL(3);
    {
        int rc = fxg_read_counters(st->ctx, st->d_counters, ln->ctr);
        if (rc == FXG_E_DEVICE && (ln->ctr[FXG_C_ERRORS] & FXG_DEV_ERR_BAD_BASE)) return;   /* the host parser prints the reference's message at its turn */
        if (rc != 0) errx(1, "GPU engine error %d: %s", rc, fxg_last_error(st->ctx));
    }
    FXH_TCALL(4);
    fxh_note_recoveries(st);
    if (lpr == 2) FXG_CHECK(st, fxg_fasta_weights(st->ctx, st->d_text, st->d_ls, st->d_ls_cap, n, st->d_res, ln->weighted));
    uint64_t out_bytes = 0;
    FXG_CHECK(st, fxg_fastq_format(st->ctx, st->d_text, lpr, st->d_ls, st->d_ls_cap, st->d_flags, n, st->d_res, ln->fwd_start, ln->reverse,
                                   revcomp ? st->d_out_bases : NULL, (revcomp && ln->has_q) ? st->d_out_qual : NULL, revcomp ? st->d_out_off : NULL,
                                   ln->has_q ? st->d_qual : NULL, stride, ln->qoffset, ln->out_fasta, st->d_out_text, &out_bytes));
    FXH_TCALL(5);
    if (ln->on_size) ln->on_size(ln, out_bytes);           /* the block's place in ONE output file depends only on the sizes before it: known here, before the download */
    if (ln->on_place) {                                     /* rank mode of the one-file run: the text stays on the device for now */
        if (ln->on_place(ln, out_bytes) < 0) return;
        FXH_TCALL(6);
        ln->t_call[7] += 1.0;
        ln->out_len = (size_t)out_bytes;
        ln->handled = 1;
        return;
    }
    const int s = ln->slot;
    if (ln->out_cap[s] < out_bytes + 16) {
        if (ln->out[s]) fxg_free_host(st->ctx, ln->out[s]);
        ln->out_cap[s] = (size_t)out_bytes + (size_t)out_bytes / 8 + 4096;
        FXG_CHECK(st, fxg_malloc_host(st->ctx, ln->out_cap[s], (void **)&ln->out[s]));
    }
    FXG_CHECK(st, fxg_memcpy_d2h(st->ctx, ln->out[s], st->d_out_text, out_bytes));
    FXG_CHECK(st, fxg_sync(st->ctx));
    FXH_TCALL(6);
    ln->t_call[7] += 1.0;
#undef FXH_TCALL
    ln->out_len = (size_t)out_bytes;
    ln->handled = 1;
}

/* the lane's context, created by the lane's own thread: the process-wide first context comes alone (two threads inside the runtime's
 * first-use initialisation take twice as long as one after the other) */
void fxh_lane_open_ctx(fxh_lane *ln)
{
    double t0 = fxh_now();
    int rc;
    pthread_mutex_lock(&g_first_ctx_mu);    /* the parts of a sharded run each have a lane 0: the process-wide first context still comes alone */
    g_hip_touched = 1;
    if (!g_first_ctx_done) { rc = fxg_ctx_create(ln->device, &ln->st.ctx); g_first_ctx_done = 1; pthread_mutex_unlock(&g_first_ctx_mu); }
    else { pthread_mutex_unlock(&g_first_ctx_mu); rc = fxg_ctx_create(ln->device, &ln->st.ctx); }
    if (rc != 0) errx(1, "no usable MI355X/HIP device %d (fxg_ctx_create = %d); this build has no CPU path", ln->device, rc);
    FXG_CHECK(&ln->st, fxg_malloc_device(ln->st.ctx, FXG_NCOUNTERS * sizeof(uint64_t), (void **)&ln->st.d_counters));
    /* one fastx_clipper process = one aligner whose query buffer survives from read to read (sequence_alignment.cpp:135-136,
     * SURVEY N3): every block of the run, host-parsed ones included, goes through this one context in input order */
    if (ln->clip_history) FXG_CHECK(&ln->st, fxg_set_clip_history(ln->st.ctx, 1));
    ln->t_init = fxh_now() - t0;
}

/* everything the lane holds on the device and page-locked on the host, then its context: by the lane's own thread, so that the lanes of a run let go
 * side by side instead of the kernel doing it for all of them, one after the other, when the process exits */
void fxh_lane_release(fxh_lane *ln)
{
    fxh_state *st = &ln->st;
    if (!st->ctx) return;
    (void)fxg_sync(st->ctx);
    void *dev[] = {st->d_text, st->d_out_text, st->d_ls, st->d_len16, st->d_flags, st->d_bases, st->d_qual, st->d_len, st->d_res, st->d_out_bases, st->d_out_qual, st->d_out_off, st->d_counters};
    for (size_t i = 0; i < sizeof dev / sizeof dev[0]; ++i) if (dev[i]) (void)fxg_free_device(st->ctx, dev[i]);
    for (int i = 0; i < FXH_LANE_OUT_SLOTS; ++i) if (ln->out[i]) {
        (void)fxg_free_host(st->ctx, ln->out[i]);
        ln->out[i] = NULL; ln->out_cap[i] = 0;
    }
    fxg_ctx_destroy(st->ctx);
    memset(st, 0, sizeof *st);
}

static void *fxh_lane_main(void *arg)
{
    fxh_lane *ln = (fxh_lane *)arg;
    if (ln->first && ln->first != ln) {
        pthread_mutex_lock(&ln->first->mu);
        while (!ln->first->ready) pthread_cond_wait(&ln->first->cv, &ln->first->mu);
        pthread_mutex_unlock(&ln->first->mu);
    }
    double t0;
    fxh_lane_open_ctx(ln);
    pthread_mutex_lock(&ln->mu);
    ln->ready = 1;
    pthread_cond_broadcast(&ln->cv);
    for (;;) {
        while (ln->state != 1 && ln->state != 3) pthread_cond_wait(&ln->cv, &ln->mu);
        if (ln->state == 3) break;
        pthread_mutex_unlock(&ln->mu);
        t0 = fxh_now();
        fxh_lane_run(ln);
        ln->t_busy += fxh_now() - t0;
        pthread_mutex_lock(&ln->mu);
        ln->state = 2;
        pthread_cond_broadcast(&ln->cv);
    }
    pthread_mutex_unlock(&ln->mu);
    return NULL;
}

static void fxh_lane_post(fxh_lane *ln, char *base, size_t cap, const char *text, size_t len, uint64_t records, int slot)
{
    pthread_mutex_lock(&ln->mu);
    ln->text_base = base; ln->text_cap = cap;
    ln->text = text; ln->len = len; ln->records = records; ln->slot = slot; ln->state = 1;
    pthread_cond_broadcast(&ln->cv);
    pthread_mutex_unlock(&ln->mu);
}

static void fxh_lane_wait(fxh_lane *ln)
{
    pthread_mutex_lock(&ln->mu);
    while (ln->state == 1) pthread_cond_wait(&ln->cv, &ln->mu);
    ln->state = 0;
    pthread_mutex_unlock(&ln->mu);
}

/* A run that uses ONE GPU moves to the CPUs of that GPU's NUMA node before it creates its helper threads and touches its buffers
 * (they are page-locked where first touched): uploads from the other socket cross the socket link -- 61.9 against 68.7 Mreads/s on the
 * sharded run of 64 M reads (profiles/r03/z_e2e_numa.txt, bench.py e2e).  The calling thread only; threads it creates inherit it.
 * FXH_NO_NUMA=1 leaves the placement to the caller (taskset / numactl / a job scheduler that already did it). */
/* returns 1 and the previous CPU set in *before when the calling thread was moved (the caller puts it back when the run is over) */
/* The NUMA node of GPU `device` without starting the HIP runtime (tens of milliseconds that the first context pays anyway, but off this thread):
 * the driver lists its nodes under /sys/class/kfd/kfd/topology/nodes/<i>/properties; the ones with SIMDs are the GPUs, in the runtime's device
 * order as long as no *_VISIBLE_DEVICES variable reorders or hides any; domain + location_id name the PCI function, whose numa_node sysfs has.
 * -1 = not found this way (no such driver node, a visibility variable, an emulated device): the caller asks the runtime. */
static int fxh_numa_node_sysfs(int device)
{
    if (getenv("HIP_VISIBLE_DEVICES") || getenv("ROCR_VISIBLE_DEVICES") || getenv("CUDA_VISIBLE_DEVICES") || getenv("GPU_DEVICE_ORDINAL")) return -1;
    int seen = 0;
    for (int i = 0; i < 256; ++i) {
        char path[160], key[64];
        snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d", i);
        if (access(path, F_OK) != 0) return -1;  /* nodes are numbered without gaps: the device is not there */
        snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/properties", i);
        FILE *f = fopen(path, "r");
        if (!f) continue;                        /* a node this process may not see (a container with some of the box's GPUs): the runtime does not count it either */
        unsigned long long v, simd = 0, loc = 0, dom = 0;
        while (fscanf(f, "%63s %llu", key, &v) == 2) {
            if (strcmp(key, "simd_count") == 0) simd = v;
            else if (strcmp(key, "location_id") == 0) loc = v;
            else if (strcmp(key, "domain") == 0) dom = v;
        }
        fclose(f);
        if (!simd) continue;                     /* a CPU node */
        if (seen++ != device) continue;
        snprintf(path, sizeof path, "/sys/bus/pci/devices/%04llx:%02llx:%02llx.%llx/numa_node", dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7);
        int node = -1;
        f = fopen(path, "r");
        if (f) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
        if (node >= 0) return node;
        /* a guest or a container whose PCI tree does not show the function under that name: the driver's own link table does -- the GPU's PCIe
         * link (type 2) ends at the CPU node it hangs off, and the driver makes its CPU nodes one per NUMA node, in NUMA order, ahead of the GPUs */
        for (int l = 0; l < 32; ++l) {
            unsigned long long type = 0, to = ~0ull;
            snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/io_links/%d/properties", i, l);
            f = fopen(path, "r");
            if (!f) break;
            while (fscanf(f, "%63s %llu", key, &v) == 2) { if (strcmp(key, "type") == 0) type = v; else if (strcmp(key, "node_to") == 0) to = v; }
            fclose(f);
            if (type == 2 && to < (unsigned long long)i) {
                snprintf(path, sizeof path, "/sys/devices/system/node/node%llu/cpulist", to);
                return access(path, R_OK) == 0 ? (int)to : -1;
            }
        }
        return -1;
    }
    return -1;
}

int fxh_bind_near_device(int device, cpu_set_t *before)
{
    if (getenv("FXH_NO_NUMA")) return 0;
    const double t0 = fxh_now();
    int node = fxh_numa_node_sysfs(device);
    const int from_sysfs = node >= 0;
    if (node < 0) {
        pthread_mutex_lock(&g_first_ctx_mu);     /* the query is a first use of the HIP runtime: one thread at a time, like the first context */
        node = fxg_device_numa_node(device);
        g_hip_touched = 1;                       /* (no fork() over an initialised runtime from here on, fxh_run_parts) */
        pthread_mutex_unlock(&g_first_ctx_mu);
    }
    if (getenv("FXH_TIMING")) fprintf(stderr, "fxh timing placement: GPU %d is on NUMA node %d (%s, %.1f ms)\n", device, node, from_sysfs ? "from sysfs" : "asked the runtime", 1e3 * (fxh_now() - t0));
    if (node < 0) return 0;
    char path[96], line[4096];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    const int got = fgets(line, sizeof line, f) != NULL;
    fclose(f);
    if (!got) return 0;
    cpu_set_t now, want;
    if (sched_getaffinity(0, sizeof now, &now) != 0) return 0;
    CPU_ZERO(&want);
    int any = 0;
    for (const char *q = line; *q && *q != '\n';) {                  /* "0-63,128-191" */
        char *end;
        long a = strtol(q, &end, 10), b = a;
        if (end == q) break;
        if (*end == '-') { q = end + 1; b = strtol(q, &end, 10); if (end == q) break; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (c >= 0 && CPU_ISSET((int)c, &now)) { CPU_SET((int)c, &want); any = 1; }
        q = (*end == ',') ? end + 1 : end;
        if (*end != ',') break;
    }
    if (!any || sched_setaffinity(0, sizeof want, &want) != 0) return 0;      /* (never widens what the caller allowed) */
    *before = now;
    return 1;
}

/* FXG_DEVICES = "0,1,3" | "all" | unset (then FXG_DEVICE, default 0) */
int fxh_device_list(int *dev, int cap)
{
    const char *e = getenv("FXG_DEVICES");
    int n = 0;
    if (e && strcmp(e, "all") == 0) {
        int nd = fxg_device_count();
        for (int i = 0; i < nd && n < cap; ++i) dev[n++] = i;
    } else if (e && *e) {
        const char *q = e;
        while (*q && n < cap) {
            char *end;
            long v = strtol(q, &end, 10);
            if (end == q) break;
            dev[n++] = (int)v;
            q = (*end == ',') ? end + 1 : end;
            if (*end != ',') break;
        }
    }
    if (n == 0) { const char *d = getenv("FXG_DEVICE"); dev[n++] = d ? atoi(d) : 0; }
    return n;
}

/* ---- the lanes loop (device text path) in four pieces: start the lanes, emit a finished block, cut the next block, stop ---- */
static fxh_lane *fxh_lanes_start(fxh_run *R, int nlanes, const int *lane_dev)
{
    FASTX *fx = R->fx;
    struct fxh_pinned *pinned = &R->pinned;
    pthread_mutex_init(&pinned->mu, NULL);
    fxh_lane *lanes = (fxh_lane *)calloc((size_t)nlanes, sizeof(fxh_lane));
    if (!lanes) err(1, "out of memory");
    for (int i = 0; i < nlanes; ++i) {
        fxh_lane *ln = &lanes[i];
        ln->id = i; ln->device = lane_dev[i]; ln->p = R->p; ln->revcomp = R->job.revcomp; ln->fwd_start = R->job.fwd_start;
        ln->qoffset = fx->fastq_ascii_quality_offset;
        ln->reverse = (R->p->stages & FXG_STAGE_REVCOMP) != 0; ln->lpr = R->job.lpr; ln->has_q = R->job.has_q; ln->out_fasta = !fx->write_fastq;
        ln->pinned = pinned; ln->first = &lanes[0];
        if ((R->p->stages & FXG_STAGE_CLIP) && !R->clip_auto) { ln->clip_history = 1; R->st_shared = 1; }      /* (one lane: fxh_run_impl saw to that) */
        ln->clip_guard = R->clip_auto;
        pthread_mutex_init(&ln->mu, NULL); pthread_cond_init(&ln->cv, NULL);
        if (pthread_create(&ln->th, NULL, fxh_lane_main, ln) != 0) err(1, "pthread_create");
    }
    return lanes;
}

/* Block `b` is next in the output order: wait for its lane and hand the formatted text to the writer, or -- a block the device
 * flagged, or one that never went to a lane -- run it through the host parser at its turn.  Returns 0 when the run must stop
 * (a part of a sharded run met such a block: R->aborted), 2 when a clipper run in its parallel phase meets its first block that is
 * not "reads of the one length seen so far" (nothing of the block has been written; see fxh_clip_go_serial). */
static int fxh_lanes_emit(fxh_run *R, fxh_lane *lanes, fxh_block *b)
{
    FASTX *fx = R->fx;
    struct fxh_reader *rd = fx->reader;
    int handled = 0;
    if (b->lane >= 0) {
        fxh_lane *ln = &lanes[b->lane];
        double tw = fxh_now();
        fxh_lane_wait(ln);
        R->t_wait_lane += fxh_now() - tw;
        if (R->clip_auto && !(ln->handled && ln->fixed_len && (R->clip_len == 0u || ln->fixed_len == R->clip_len))) {
            if (R->nparts > 1) { R->aborted = 1; FXH_ABORT_SET(); return 0; }      /* a part cannot know what came before it: the whole run starts over as one stream */
            return 2;                      /* the caller switches to the one-aligner mode and brings this block back */
        }
        if (ln->handled) {
            handled = 1;
            if (R->clip_auto) {            /* remember the block's last record: what the aligner would hold if the next block is the first different one */
                R->clip_len = ln->fixed_len;
                const char *t = b->buf + b->beg, *e = b->buf + b->end, *q = e;
                for (int k = 0; k < ln->lpr && q > t; ++k) { const char *r = (const char *)memrchr(t, '\n', (size_t)(q - 1 - t)); q = r ? r + 1 : t; }
                const size_t need = (size_t)(e - q);
                if (R->clip_seed_cap < need) { free(R->clip_seed); R->clip_seed_cap = need + 256; R->clip_seed = (char *)malloc(R->clip_seed_cap); if (!R->clip_seed) err(1, "out of memory"); }
                memcpy(R->clip_seed, q, need); R->clip_seed_len = need;
            }
            tw = fxh_now();
            fxh_awriter_submit_ext(&R->aw, fx->writer, ln->out[ln->slot], ln->out_len);
            R->t_wait_writer += fxh_now() - tw;
            if (!R->overlap) fxh_awriter_wait(&R->aw);
            fxh_add_counters(R->tot, ln->ctr, b->records, ln->lpr == 2 ? ln->weighted : NULL);
        }
    }
    if (!handled && R->clip_auto && R->nparts <= 1) return 2;      /* (a block that never went to a lane: the host parser needs the one aligner too) */
    if (!handled && R->nparts > 1) {   /* a part of a sharded run only takes what the device path takes: the whole run starts over unsharded */
        R->aborted = 1; FXH_ABORT_SET();
        return 0;
    }
    if (!handled) {                    /* this block goes through the host parser, at its place in the output order */
        R->n_fallback++;
        if (R->st_shared && !R->st.ctx) {          /* serial clipper run: the host parser works through lane 0's context */
            fxh_lane *l0 = &lanes[0];
            pthread_mutex_lock(&l0->mu);
            while (!l0->ready) pthread_cond_wait(&l0->cv, &l0->mu);
            pthread_mutex_unlock(&l0->mu);
            R->st.ctx = l0->st.ctx; R->st.d_counters = l0->st.d_counters;
        }
        struct fxh_reader save = *rd;
        const unsigned long long save_line = fx->input_line_number;
        rd->buf = b->buf; rd->beg = b->beg; rd->end = b->end; rd->eof = b->eof;
        fx->input_line_number = b->line0;
        while (rd->beg < rd->end && !R->have_err) {
            const size_t before = rd->beg;
            fxh_host_block(R);
            if (rd->beg == before) break;
        }
        if (!R->have_err && rd->beg < rd->end) errx(1, "internal error: host parser left %zu bytes of a block", rd->end - rd->beg);
        *rd = save;
        fx->input_line_number = save_line;
        R->at_eof = 0;
    }
    fx->num_input_sequences = R->tot->input_sequences; fx->num_input_reads = R->tot->input_reads;
    fx->num_output_sequences = R->tot->output_sequences; fx->num_output_reads = R->tot->output_reads;
    return 1;
}

/* Cut the unread text of the reader's buffer at a record boundary: records are groups of lpr lines counted from the start of the
 * input, so the cut only needs the number of complete lines.  fresh_nl = newlines the reader threads counted in the freshly read
 * part ((size_t)-1: unknown), carry_lines = complete lines of the unread tail in front of it (when *have_carry).  Out: `end` (the
 * text's end incl. a '\n' appended at end of input), `lines` up to there, `cut` (end of the last whole record). */
static void fxh_cut_records(fxh_run *R, size_t fresh_nl, int have_carry, unsigned long long carry_lines, size_t *end_out, unsigned long long *lines_out, size_t *cut_out)
{
    struct fxh_reader *rd = R->fx->reader;
    size_t end = rd->end;
    if (rd->eof && rd->buf[end - 1] != '\n') { rd->buf[end] = '\n'; end += 1; }       /* the buffer has one spare byte */
    fxh_job *job = &R->job;
    const int T = job->nworkers;
    for (int i = 0; i < T; ++i) {
        job->w[i].a0 = rd->beg + (size_t)((unsigned long long)(end - rd->beg) * (unsigned)i / (unsigned)T);
        job->w[i].a1 = rd->beg + (size_t)((unsigned long long)(end - rd->beg) * (unsigned)(i + 1) / (unsigned)T);
    }
    unsigned long long lines = 0;
    if (fresh_nl != (size_t)-1 && have_carry) lines = carry_lines + fresh_nl + (end > rd->end ? 1u : 0u);   /* tail of the last block + fresh data (+ the appended '\n') */
    else {
        fxh_parallel(job, fxh_phase_census);
        for (int i = 0; i < T; ++i) lines += job->w[i].nl_count;
    }
    const unsigned lpr = (unsigned)job->lpr;
    size_t cut = end;
    if (!rd->eof || lines % lpr != 0) {  /* drop the incomplete last line and the lines of the incomplete record in front of it */
        unsigned drop = (unsigned)(lines % lpr);
        const char *q = (const char *)memrchr(rd->buf + rd->beg, '\n', end - rd->beg);
        cut = q ? (size_t)(q - rd->buf) + 1 : rd->beg;
        while (drop-- && cut > rd->beg) {
            q = (const char *)memrchr(rd->buf + rd->beg, '\n', cut - 1 - rd->beg);
            cut = q ? (size_t)(q - rd->buf) + 1 : rd->beg;
        }
    }
    *end_out = end; *lines_out = lines; *cut_out = cut;
}

void fxh_lanes_stop(fxh_run *R, fxh_lane *lanes, int nlanes, double *t_lane_init)
{
    /* (when an error is pending, blocks after the bad record are abandoned, like everything after an errx() in the reference) */
    for (int i = 0; i < nlanes; ++i) {
        fxh_lane *ln = &lanes[i];
        pthread_mutex_lock(&ln->mu);
        while (ln->state == 1) pthread_cond_wait(&ln->cv, &ln->mu);
        ln->state = 3;
        pthread_cond_broadcast(&ln->cv);
        pthread_mutex_unlock(&ln->mu);
        pthread_join(ln->th, NULL);
        *t_lane_init += ln->t_init;
        R->t_gpu += ln->t_busy;
        if (getenv("FXH_TIMING") && ln->t_call[7] > 0)
            fprintf(stderr, "fxh timing lane %d: %.0f blocks, ms per block: h2d %.3f index %.3f pack %.3f pipeline %.3f counters %.3f format %.3f d2h+sync %.3f\n", i, ln->t_call[7],
                    1e3 * ln->t_call[0] / ln->t_call[7], 1e3 * ln->t_call[1] / ln->t_call[7], 1e3 * ln->t_call[2] / ln->t_call[7], 1e3 * ln->t_call[3] / ln->t_call[7],
                    1e3 * ln->t_call[4] / ln->t_call[7], 1e3 * ln->t_call[5] / ln->t_call[7], 1e3 * ln->t_call[6] / ln->t_call[7]);
    }
    { const double tw = fxh_now(); fxh_awriter_wait(&R->aw); R->t_drain += fxh_now() - tw; }   /* the last lane buffer must be on its way out before the contexts go */
    /* The process is about to exit: device buffers, streams and page-locked memory go with it, there is nothing to gain from
     * tearing each context down first (FXH_TEARDOWN=1 does it anyway, for leak checkers). */
    if (R->st_shared) { if (R->st.ctx) fxg_sync(R->st.ctx); R->st.ctx = NULL; }
    if (getenv("FXH_TEARDOWN") || (R->nparts > 1 && (R->aborted || R->have_err || FXH_ABORTED())))     /* an abandoned sharded attempt ends with the device idle and no context left */
        for (int i = 0; i < nlanes; ++i) fxg_ctx_destroy(lanes[i].st.ctx);
    free(lanes);
}

/* A clipper run leaves its parallel phase at block blk[first]: every lane comes to rest (what the lanes made of this and the later
 * blocks is dropped -- none of it has been written), lane 0 becomes the reference's one aligner (history on, as in a serial run) and
 * is brought to the state that aligner has after reads of one length -- its buffer holds the LAST of them -- by running the last record
 * before the block through it; then the blocks already cut go through it again, in order.  From here on the run is the serial run. */
static void fxh_clip_go_serial(fxh_run *R, fxh_lane *lanes, int nlanes, fxh_block *blk, int NB, size_t first, size_t nblocks, size_t *lane_uses)
{
    for (int i = 0; i < nlanes; ++i) { fxh_lane_wait(&lanes[i]); lanes[i].clip_guard = 0; }
    fxh_awriter_wait(&R->aw);               /* no output buffer of a lane is with the writer while lane 0 runs the seed */
    fxh_lane *l0 = &lanes[0];
    pthread_mutex_lock(&l0->mu);
    while (!l0->ready) pthread_cond_wait(&l0->cv, &l0->mu);
    pthread_mutex_unlock(&l0->mu);
    FXG_CHECK(&l0->st, fxg_set_clip_history(l0->st.ctx, 1));
    l0->clip_history = 1;
    R->clip_auto = 0; R->st_shared = 1;
    if (R->clip_seed_len) {                 /* (no record before the block: the aligner is fresh, as at the start of a serial run) */
        fxh_lane_post(l0, NULL, 0, R->clip_seed, R->clip_seed_len, 1, (int)(lane_uses[0]++ & 1u));
        fxh_lane_wait(l0);
        if (!l0->handled) errx(1, "internal error: the record before the first ragged block did not pass the device path a second time");
    }
    for (size_t j = first; j < nblocks; ++j) {          /* the blocks already cut: lane 0 takes them one by one as they are emitted */
        fxh_block *b = &blk[j % (size_t)NB];
        if (b->lane >= 0) { b->lane = 0; b->posted = 0; }
    }
    if (getenv("FXH_TIMING")) fprintf(stderr, "fxh timing clipper: reads of one length (%u) up to block %zu; one aligner with history from there on\n", R->clip_len, first);
}

/* The lanes loop.  Returns when the input is exhausted or an error is pending in R. */
void fxh_run_lanes(fxh_run *R, fxh_prefetch *pf, int nlanes, const int *lane_dev, double *t_read, double *t_lane_init)
{
    FASTX *fx = R->fx;
    struct fxh_reader *rd = fx->reader;
    fxh_lane *lanes = fxh_lanes_start(R, nlanes, lane_dev);
    const int NB = nlanes + 2;             /* input buffers: nlanes blocks in flight + the one being cut + the one being read */
    char **inbuf = (char **)calloc((size_t)NB, sizeof(char *));
    fxh_block *blk = (fxh_block *)calloc((size_t)NB, sizeof(fxh_block));
    if (!inbuf || !blk) err(1, "out of memory");
    inbuf[0] = rd->buf;
    size_t nblocks = 0, next_emit = 0;
    size_t lane_uses[FXH_MAX_LANES] = {0};
    int input_done = 0, have_carry = 0;
    unsigned long long carry_lines = 0;
    int nl = nlanes;                        /* lanes that take blocks: all of them, or lane 0 alone once a clipper run has gone serial */

    while (!R->have_err && !R->aborted && !(R->nparts > 1 && FXH_ABORTED())) {
        /* ---- collect finished blocks in input order until a lane and an input buffer are free ---- */
        while (next_emit < nblocks && (nblocks - next_emit >= (size_t)nl || input_done)) {
            fxh_block *eb = &blk[next_emit % (size_t)NB];
            if (eb->lane >= 0 && !eb->posted) {          /* a block fxh_clip_go_serial took back: lane 0 runs it now, with history */
                fxh_lane_post(&lanes[0], eb->buf, rd->cap + 1, eb->buf + eb->beg, eb->end - eb->beg, eb->records, (int)(lane_uses[0]++ & 1u));
                eb->posted = 1;
            }
            const int erc = fxh_lanes_emit(R, lanes, eb);
            if (erc == 2) { fxh_clip_go_serial(R, lanes, nlanes, blk, NB, next_emit, nblocks, lane_uses); nl = 1; continue; }
            if (!erc) break;
            next_emit++;
            if (R->have_err) break;
        }
        if (R->aborted) break;
        if (R->have_err || input_done) { if (next_emit >= nblocks) break; else continue; }

        /* ---- next block of text: [unread tail of the previous block | prefetched data] ---- */
        double t0 = fxh_now();
        size_t fresh_nl = (size_t)-1;      /* newlines in the freshly read part, when the reader threads counted them */
        {
            const size_t nxt = (nblocks + 1) % (size_t)NB;        /* where the read-ahead for the block after this one goes */
            if (!inbuf[nxt]) { inbuf[nxt] = (char *)malloc(rd->cap + 1); if (!inbuf[nxt]) err(1, "out of memory"); }
            fxh_next_block_ring(pf, rd, inbuf[nxt], &fresh_nl);
        }
        *t_read += fxh_now() - t0;
        if (rd->beg == rd->end && rd->eof) { input_done = 1; continue; }

        /* ---- cut it at a record boundary and give it to the next lane ---- */
        t0 = fxh_now();
        size_t end, cut;
        unsigned long long lines;
        fxh_cut_records(R, fresh_nl, have_carry, carry_lines, &end, &lines, &cut);
        const unsigned lpr = (unsigned)R->job.lpr;
        const uint64_t records = lines / lpr;
        R->t_index += fxh_now() - t0;
        fxh_block *b = &blk[nblocks % (size_t)NB];
        b->buf = rd->buf; b->beg = rd->beg; b->line0 = fx->input_line_number; b->records = records; b->lane = -1; b->posted = 0;
        if (rd->eof && (lines % lpr != 0 || records == 0)) {
            /* ragged end of input: the host parser owns the message; hand it everything that is left */
            b->end = end; b->eof = 1;
            rd->beg = rd->end; input_done = 1;
        } else if (records == 0) {
            errx(1, "input record does not fit in the %zu MB read buffer", rd->cap >> 20);
        } else {
            b->end = cut; b->eof = (rd->eof && cut == end);
            const int li = (int)(nblocks % (size_t)nl);
            b->lane = li; b->posted = 1;
            fxh_lane_post(&lanes[li], rd->buf, rd->cap + 1, rd->buf + rd->beg, cut - rd->beg, records, (int)(lane_uses[li]++ & 1u));
            rd->beg = cut < rd->end ? cut : rd->end;
            carry_lines = lines - (unsigned long long)lpr * records - (end > rd->end ? 1u : 0u); have_carry = 1;   /* complete lines left in the unread tail */
            fx->input_line_number += (unsigned long long)lpr * records;
            if (rd->eof && cut == end) input_done = 1;
        }
        nblocks++;
    }
    fxh_lanes_stop(R, lanes, nlanes, t_lane_init);
    for (int k = 1; k < NB; ++k) if (inbuf[k] && inbuf[k] != rd->buf) free(inbuf[k]);
    free(inbuf); free(blk);
}

