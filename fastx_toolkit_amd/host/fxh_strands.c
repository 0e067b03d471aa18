/* fxh_strands.c -- `tool -i in.fq -o out.fq`, ONE output file, at the speed of the sharded run (fxh_priv.h). */
/* ---------------------------------------------------------------------------------------------- */
/* The reference writes one output stream (fastx.c:251-271: one fdopen; fastx.c:440-473).  The sharded run of fxh_parts.c is fast because */
/* it has k of everything -- k byte ranges read side by side, k x lanes on the device, k output FILES -- and that last k is what no        */
/* reference command line has.  Here the same input goes to ONE file, written at exact offsets by many threads:                             */
/*   * the input file is cut into CHUNKS (16 MB) at record boundaries found by pattern and proven by induction, exactly like the parts       */
/*     (chunk c ends where chunk c + 1 begins; a chunk that is not a whole number of regular records stops the attempt);                   */
/*   * STRANDS (one thread + one engine context each, FXH_STRANDS per GPU) draw chunk numbers from one counter, so the chunk order IS the   */
/*     output order; a strand reads its chunk with parallel pread(), uploads, indexes, decides and formats it on the device;                */
/*   * the formatted SIZE of a chunk is known before its text comes down.  Sizes are published in an array; the file offset of chunk c is   */
/*     the sum of the sizes before it (a running scan advanced by whoever publishes), so it is known as soon as every earlier chunk has     */
/*     been DECIDED -- not written, not even downloaded.  The smallest unfinished chunk never waits for anybody, every strand owns its      */
/*     buffers, so there is no deadlock by construction.  No second pass over the input, no text held back in HBM: the two-phase form       */
/*     (sizes first, text later) would serialise upload and download on the PCIe link, which is the resource the run is bound by;           */
/*   * the SINK.  One inode of a tmpfs (or any page cache) takes fresh pages from ONE thread at a time: concurrent pwrite() serialise on    */
/*     the inode lock (4 GB/s with 8 threads against 9 with one), concurrent faults on a shared mapping on the mapping's locks (5 GB/s),     */
/*     k files take 30-90 GB/s (profiles/r05/a_one_file_write.txt).  What does scale is copying into pages that EXIST: fallocate()          */
/*     allocates without zeroing or copying at 18 GB/s as long as nobody faults on the file meanwhile, and 16 threads then copy into the     */
/*     mapping at 30 GB/s.  So allocation and copies take turns behind a gate (fxh_sf_alloc_main), the allocator runs ahead in 128 MB        */
/*     windows -- from the first milliseconds of the process, while the device is still starting and nothing is there to be written --       */
/*     and the copies drop their page-table entries themselves (MADV_DONTNEED, shared mmap_lock), so the process does not spend 0.2 s        */
/*     unmapping at exit.  Measured with the tool's own access pattern: 10 GB into one tmpfs file in 0.65 s after a 0.25 s head start,       */
/*     against 1.4 s through one pwrite() stream and 1.6 s ungated (profiles/r05/d_one_file_gate.txt).  Other file systems get pwrite().    */
/* Anything irregular (a malformed record, a cut that was not a boundary, a clipper input whose reads are not all of one length) abandons   */
/* the attempt exactly like the sharded run: it lives in a forked child, which empties the file and leaves with FXH_EXIT_ABANDON, and the   */
/* parent -- which has not touched the GPU -- runs the input as one stream, so messages, exit codes and partial output are the reference's. */
/* ---------------------------------------------------------------------------------------------- */
#include "fxh_priv.h"
#include <sys/mman.h>
#include <sys/vfs.h>
#ifndef TMPFS_MAGIC
#define TMPFS_MAGIC 0x01021994
#endif

#define FXH_TICKET_DONE ((uint64_t)-1)
#define FXH_MAX_STRANDS FXH_MAX_LANES

/* ---- a small pool of worker threads: tasks never wait for other tasks ---- */
typedef struct { void (*fn)(void *); void *arg; } fxh_task;
typedef struct fxh_pool {
    pthread_mutex_t mu;
    pthread_cond_t cv_work, cv_space;
    fxh_task *q;
    unsigned cap, head, count;
    int quit, nth;
    pthread_t th[64];
} fxh_pool;

static void *fxh_pool_main(void *arg)
{
    fxh_pool *P = (fxh_pool *)arg;
    pthread_mutex_lock(&P->mu);
    for (;;) {
        while (P->count == 0 && !P->quit) pthread_cond_wait(&P->cv_work, &P->mu);
        if (P->count == 0) break;
        const fxh_task t = P->q[P->head];
        P->head = (P->head + 1) % P->cap; P->count--;
        pthread_cond_signal(&P->cv_space);
        pthread_mutex_unlock(&P->mu);
        t.fn(t.arg);
        pthread_mutex_lock(&P->mu);
    }
    pthread_mutex_unlock(&P->mu);
    return NULL;
}

static void fxh_pool_start(fxh_pool *P, int nth, unsigned cap)
{
    memset(P, 0, sizeof *P);
    pthread_mutex_init(&P->mu, NULL); pthread_cond_init(&P->cv_work, NULL); pthread_cond_init(&P->cv_space, NULL);
    P->cap = cap; P->q = (fxh_task *)calloc(cap, sizeof(fxh_task));
    if (!P->q) err(1, "out of memory");
    if (nth > 64) nth = 64;
    if (nth < 1) nth = 1;
    P->nth = nth;
    for (int i = 0; i < nth; ++i) if (pthread_create(&P->th[i], NULL, fxh_pool_main, P) != 0) err(1, "pthread_create");
}

static void fxh_pool_submit(fxh_pool *P, void (*fn)(void *), void *arg)
{
    pthread_mutex_lock(&P->mu);
    while (P->count == P->cap) pthread_cond_wait(&P->cv_space, &P->mu);
    P->q[(P->head + P->count) % P->cap].fn = fn;
    P->q[(P->head + P->count) % P->cap].arg = arg;
    P->count++;
    pthread_cond_signal(&P->cv_work);
    pthread_mutex_unlock(&P->mu);
}

static void fxh_pool_stop(fxh_pool *P)       /* queued tasks are still run */
{
    pthread_mutex_lock(&P->mu);
    P->quit = 1;
    pthread_cond_broadcast(&P->cv_work);
    pthread_mutex_unlock(&P->mu);
    for (int i = 0; i < P->nth; ++i) pthread_join(P->th[i], NULL);
    free(P->q);
}

/* ---- the run ---- */
typedef struct fxh_sf fxh_sf;
typedef struct fxh_strand fxh_strand;
typedef struct { fxh_strand *s; int fd; char *dst; size_t n; off_t off; } fxh_rjob;
typedef struct { fxh_strand *s; int slot; } fxh_wjob;

struct fxh_strand {
    int id;
    fxh_sf *S;
    fxh_lane ln;                           /* the engine context, its device buffers and the two page-locked output buffers (fxh_lanes.c) */
    pthread_t th_gpu, th_rd;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    /* input slots: filled by the strand's reader in ticket order, emptied by its device thread */
    char *in[2];
    size_t in_len[2];
    uint64_t in_ticket[2];
    int in_full[2];
    int rd_pending;                        /* pread slices of the chunk being read that have not finished */
    fxh_rjob rjob[16];
    /* output slots: ln.out[0 .. nout), from the device thread to the copy tasks */
    size_t out_len[FXH_LANE_OUT_SLOTS];
    uint64_t out_off[FXH_LANE_OUT_SLOTS];
    int out_full[FXH_LANE_OUT_SLOTS];
    fxh_wjob wjob[FXH_LANE_OUT_SLOTS];
    uint64_t cur_ticket;
    fxh_totals tot;
    uint64_t chunks;
    double t_read, t_wait_in, t_gpu, t_wait_out, t_wait_off, t_release;
};

struct fxh_sf {
    FASTX *fx;
    const fxg_params *p;
    int in_fd, lpr, clip_auto;
    off_t *cut;                            /* chunk c is the input bytes [cut[c], cut[c + 1]) */
    uint64_t nchunks, in_total;
    size_t in_cap;
    uint64_t next_ticket;                  /* atomic: the next chunk nobody has taken */
    int nstrands, nread_slices, nout, release;
    fxh_strand *st;
    fxh_pool rpool, wpool;
    /* sizes -> offsets, the clipper's one read length, the allocator's state: all under mu / cv */
    pthread_mutex_t mu;
    pthread_cond_t cv;
    uint64_t *size, *offset;
    uint8_t *have;
    uint64_t scanned, scan_off, published, in_done, out_done;
    uint32_t clip_len;
    /* sink */
    /* rank mode (FXH_RANK / FXH_WORLD): one process per GPU, each over its byte range of the input.  Where a rank's text goes in the ONE file is only
     * known once every rank has decided its range, so the formatted chunks stay on the device (the arena: HBM holds any realistic range) until the
     * ranks have exchanged their counter blocks -- one RCCL all-gather (fxg_epilogue_rccl) -- and then go down and out at base + local offset */
    int rank, world;
    fxg_ctx *main_ctx;
    uint8_t *arena;
    uint64_t arena_cap;
    int out_fd, mapped;
    int prealloc;                          /* rank mode, rank 0, a tmpfs: the allocator makes the JOB's pages while the ranks compute (nobody copies meanwhile) */
    char *map;
    uint64_t map_len, alloc_end, need, window;
    uint64_t alloc_in_total;               /* the input the allocator's estimate is about: this process's (strands) or the job's (rank 0 of a rank job) */
    uint64_t ratio_after;                  /* input bytes that must have been decided before the measured output/input ratio counts (64 MB) */
    int keep_surplus;                      /* FXH_ONE_FILE_KEEP_SURPLUS=1: measurements of the run without it */
    uint64_t given_back;                   /* pages the head start made and the output turned out not to need, returned during the run */
    uint64_t alloc_final; int alloc_final_set;      /* the exact size, once known: strands -- every chunk published; rank job -- the exchange */
    uint64_t map_base;                     /* rank mode: file offset of map[0] (the page this rank's slice starts in) */
    int alloc_errno, alloc_stop, alloc_capped;      /* alloc_capped: the head start met ENOSPC -- from here on only what a copy asks for */
    int drain_errno;                       /* rank mode: errno of the first piece of this rank's text that did not get into the file */
    pthread_t th_alloc;
    pthread_rwlock_t gate;                 /* fallocate() exclusive, copies shared; writer-preferring */
    double t_alloc, t_copy, t_copy_wait;   /* t_copy*: summed over the copy tasks, under mu */
    uint64_t alloc_calls;
};

static void fxh_sf_abort(fxh_sf *S)
{
    FXH_ABORT_SET();
    for (int i = 0; i < S->nstrands; ++i) { pthread_mutex_lock(&S->st[i].mu); pthread_cond_broadcast(&S->st[i].cv); pthread_mutex_unlock(&S->st[i].mu); }
    pthread_mutex_lock(&S->mu); pthread_cond_broadcast(&S->cv); pthread_mutex_unlock(&S->mu);
}

/* how far the allocation should reach now (mu held): everything, once every size is known; before that the output expected from the
 * chunks decided so far -- and, while there are none, a quarter of the input (at most 8 GB): the head start the device's start-up gives */
static uint64_t fxh_sf_alloc_goal(const fxh_sf *S)
{
    uint64_t goal;
    if (S->alloc_final_set) goal = S->alloc_final;
    else if (S->in_done >= S->ratio_after && S->in_done) {
        const long double r = (long double)S->out_done / (long double)S->in_done;
        goal = (uint64_t)(r * 1.02L * (long double)S->alloc_in_total) + S->window / 4;
    } else {
        goal = S->alloc_in_total / 4;
        if (goal > ((uint64_t)8 << 30)) goal = (uint64_t)8 << 30;
    }
    if (S->alloc_capped && !S->alloc_final_set) goal = S->need;      /* no room to run ahead in: exactly what the copies need */
    if (!S->alloc_final_set && goal < S->need) goal = S->need;
    if (goal > S->map_len) goal = S->map_len;
    return goal;
}

/* The allocator.  Allocation and copies exclude each other (S->gate: fallocate() exclusive, every copied megabyte shared, the allocator preferred),
 * and the allocator is EAGER: it runs ahead of the copies towards the expected size of the output whenever it gets the file.  What it allocates before
 * the first chunk comes back from the device costs nothing, and the sooner it is through, the longer the copies have the file to themselves at full
 * parallelism.  (Measured alternative: copies first while their pages exist, the allocator only in the sink's idle moments and with priority when a
 * copy waits for pages -- 45 against 53-56 Mreads/s: the sink then spends the run switching, each switch waiting for the running copies to drain;
 * profiles/r05/g_e2e_one_file_copies_first.txt.) */
static void *fxh_sf_alloc_main(void *arg)
{
    fxh_sf *S = (fxh_sf *)arg;
    pthread_mutex_lock(&S->mu);
    while (!S->alloc_stop && !FXH_ABORTED() && !S->alloc_errno) {
        const uint64_t goal = fxh_sf_alloc_goal(S);
        if (S->alloc_end >= goal) {
            if (S->alloc_final_set) break;                   /* the whole output has its pages */
            /* A tool that keeps little: the head start (a quarter of the input, made before anything was known) is several times what the output will take.
             * Those pages go back NOW, in the shadow of the run, instead of in the ftruncate() at its end, which the caller waits for.  What stays -- the
             * estimate plus a window -- lies above every byte a copy can be holding: off + len <= bytes decided so far <= the estimate. */
            if (S->mapped && !S->keep_surplus && S->in_done >= S->ratio_after && S->in_done && S->alloc_end > 2 * goal + S->window) {
                const uint64_t keep = (goal + S->window + 4095u) & ~(uint64_t)4095u, end = S->alloc_end;
                S->alloc_end = keep;                         /* first: nobody starts a copy beyond it from here on */
                pthread_mutex_unlock(&S->mu);
                pthread_rwlock_wrlock(&S->gate);
                const double t0 = fxh_now();
                (void)fallocate(S->out_fd, FALLOC_FL_PUNCH_HOLE | FALLOC_FL_KEEP_SIZE, (off_t)keep, (off_t)(end - keep));      /* (if it fails the pages stay until the end, as before) */
                const double dt = fxh_now() - t0;
                pthread_rwlock_unlock(&S->gate);
                pthread_mutex_lock(&S->mu);
                S->t_alloc += dt; S->given_back += end - keep;
                continue;
            }
            pthread_cond_wait(&S->cv, &S->mu);
            continue;
        }
        const uint64_t a = S->alloc_end;
        uint64_t step = goal - a < S->window ? goal - a : S->window;
        step = (step + 4095u) & ~(uint64_t)4095u;
        if (a + step > S->map_len) step = S->map_len - a;
        pthread_mutex_unlock(&S->mu);
        pthread_rwlock_wrlock(&S->gate);                     /* no copy faults on the file while its pages are being made */
        const double t0 = fxh_now();
        int rc;
        do rc = fallocate(S->out_fd, 0, (off_t)a, (off_t)step); while (rc != 0 && errno == EINTR);
        const int e = rc != 0 ? errno : 0;
        const double dt = fxh_now() - t0;
        pthread_rwlock_unlock(&S->gate);
        pthread_mutex_lock(&S->mu);
        S->t_alloc += dt; S->alloc_calls++;
        /* ENOSPC on pages nobody has asked for yet (the head start is a guess made before any output size is known; a filter that keeps 1 % needs a fraction
         * of it) is not the run's problem: stop running ahead and fail only when a copy really cannot get its pages (advisor, round 5) */
        if (e == ENOSPC && a >= S->need && !S->alloc_final_set && !S->alloc_capped) S->alloc_capped = 1;
        else if (e) S->alloc_errno = e; else S->alloc_end = a + step;
        pthread_cond_broadcast(&S->cv);
    }
    pthread_mutex_unlock(&S->mu);
    return NULL;
}

/* the lane's hook: the chunk's formatted size is known (fxh_lane_run, after the format kernels, before the download) */
static void fxh_sf_publish(fxh_lane *ln, uint64_t bytes)
{
    fxh_strand *s = (fxh_strand *)ln->owner;
    fxh_sf *S = s->S;
    const uint64_t t = s->cur_ticket;
    pthread_mutex_lock(&S->mu);
    S->size[t] = bytes; S->have[t] = 1; S->published++;
    S->in_done += (uint64_t)(S->cut[t + 1] - S->cut[t]); S->out_done += bytes;
    while (S->scanned < S->nchunks && S->have[S->scanned]) { S->offset[S->scanned] = S->scan_off; S->scan_off += S->size[S->scanned]; S->scanned++; }
    if (S->published == S->nchunks && !S->prealloc) { S->alloc_final = S->scan_off; S->alloc_final_set = 1; }      /* (a rank job's size is the exchange's to say) */
    pthread_cond_broadcast(&S->cv);
    pthread_mutex_unlock(&S->mu);
}

/* rank mode: the chunk's text stays on the device -- copied behind the format kernels into the arena at the sum of the sizes before it (the local
 * offset: known once every earlier chunk of THIS rank has been decided).  1 = placed, -1 = the run is being abandoned. */
static int fxh_sf_place(fxh_lane *ln, uint64_t bytes)
{
    fxh_strand *s = (fxh_strand *)ln->owner;
    fxh_sf *S = s->S;
    const uint64_t t = s->cur_ticket;
    const double t0 = fxh_now();
    pthread_mutex_lock(&S->mu);
    while (S->scanned <= t && !FXH_ABORTED()) pthread_cond_wait(&S->cv, &S->mu);
    const uint64_t off = S->offset[t];
    pthread_mutex_unlock(&S->mu);
    s->t_wait_off += fxh_now() - t0;
    if (FXH_ABORTED()) return -1;
    if (off + bytes > S->arena_cap) { fxh_sf_abort(S); return -1; }      /* (cannot happen: 8/7 of the range bounds its output) */
    FXG_CHECK(&ln->st, fxg_concat_peer(S->main_ctx, S->arena, off, ln->st.ctx, ln->st.d_out_text, bytes));
    FXG_CHECK(&ln->st, fxg_sync(ln->st.ctx));
    return 1;
}

static void fxh_sf_read_task(void *arg)
{
    fxh_rjob *j = (fxh_rjob *)arg;
    size_t got = 0;
    while (got < j->n) {
        const ssize_t k = pread(j->fd, j->dst + got, j->n - got, j->off + (off_t)got);
        if (k < 0) { if (errno == EINTR) continue; err(1, "read failed"); }
        if (k == 0) break;                                   /* the file shrank under the run */
        got += (size_t)k;
    }
    fxh_strand *s = j->s;
    if (got < j->n) fxh_sf_abort(s->S);
    pthread_mutex_lock(&s->mu);
    s->rd_pending--;
    pthread_cond_broadcast(&s->cv);
    pthread_mutex_unlock(&s->mu);
}

static void *fxh_strand_reader(void *arg)
{
    fxh_strand *s = (fxh_strand *)arg;
    fxh_sf *S = s->S;
    for (int k = 0;; k ^= 1) {
        pthread_mutex_lock(&s->mu);
        while (s->in_full[k] && !FXH_ABORTED()) pthread_cond_wait(&s->cv, &s->mu);
        pthread_mutex_unlock(&s->mu);
        if (FXH_ABORTED()) break;
        const uint64_t t = __atomic_fetch_add(&S->next_ticket, 1, __ATOMIC_RELAXED);      /* drawn with a free buffer in hand: the smallest open chunk always has one */
        if (t >= S->nchunks) {
            pthread_mutex_lock(&s->mu);
            s->in_ticket[k] = FXH_TICKET_DONE; s->in_full[k] = 1;
            pthread_cond_broadcast(&s->cv);
            pthread_mutex_unlock(&s->mu);
            break;
        }
        const double t0 = fxh_now();
        const off_t off = S->cut[t];
        size_t n = (size_t)(S->cut[t + 1] - off);
        int ns = S->nread_slices;
        if ((size_t)ns > n / ((size_t)1 << 20)) ns = (int)(n / ((size_t)1 << 20));
        if (ns < 1) ns = 1;
        const size_t per = (n + (size_t)ns - 1) / (size_t)ns;
        pthread_mutex_lock(&s->mu); s->rd_pending = ns; pthread_mutex_unlock(&s->mu);
        for (int i = 0; i < ns; ++i) {
            const size_t o = (size_t)i * per;
            fxh_rjob *j = &s->rjob[i];
            j->s = s; j->fd = S->in_fd; j->dst = s->in[k] + o; j->off = off + (off_t)o; j->n = o >= n ? 0 : (n - o < per ? n - o : per);
            if (i + 1 < ns) fxh_pool_submit(&S->rpool, fxh_sf_read_task, j);
        }
        fxh_sf_read_task(&s->rjob[ns - 1]);                 /* the last slice on this thread */
        pthread_mutex_lock(&s->mu);
        while (s->rd_pending > 0) pthread_cond_wait(&s->cv, &s->mu);
        pthread_mutex_unlock(&s->mu);
        if (FXH_ABORTED()) break;
        if (t + 1 == S->nchunks && s->in[k][n - 1] != '\n') s->in[k][n++] = '\n';       /* the reference takes a last line without its newline (chomp.c:36-41) */
        s->t_read += fxh_now() - t0;
        pthread_mutex_lock(&s->mu);
        s->in_len[k] = n; s->in_ticket[k] = t; s->in_full[k] = 1;
        pthread_cond_broadcast(&s->cv);
        pthread_mutex_unlock(&s->mu);
    }
    return NULL;
}

static void fxh_sf_write_task(void *arg)
{
    fxh_wjob *j = (fxh_wjob *)arg;
    fxh_strand *s = j->s;
    fxh_sf *S = s->S;
    const char *src = s->ln.out[j->slot];
    const size_t len = s->out_len[j->slot];
    const uint64_t off = s->out_off[j->slot];
    double t0 = fxh_now(), t_wait = 0;
    if (len && S->mapped && off + len > S->map_len) fxh_sf_abort(S);      /* (cannot happen: 8/7 of the input bounds the output; one stream if it ever does) */
    else if (len && S->mapped) {
        pthread_mutex_lock(&S->mu);
        while (S->alloc_end < off + len && !S->alloc_errno && !FXH_ABORTED()) {
            if (S->need < off + len) { S->need = off + len; pthread_cond_broadcast(&S->cv); }
            pthread_cond_wait(&S->cv, &S->mu);
        }
        const int e = S->alloc_errno;
        pthread_mutex_unlock(&S->mu);
        t_wait = fxh_now() - t0;
        if (e) fxh_sf_abort(S);                      /* no room for the pages (ENOSPC): the one-stream run meets the same wall and reports it the way it always did, with the output it got that far */
        if (!FXH_ABORTED()) {
            /* the whole buffer under one hold of the gate.  The allocator is preferred, so the sink strictly alternates: a window of pages, then
             * EVERY copy that has piled up meanwhile side by side (copies are only fast many at a time), then the next window.  (A megabyte per
             * hold let the allocator in sooner and the copies trickle: 48.7 against 54.6 Mreads/s, profiles/r05/h_e2e_one_file_piecewise.txt.) */
            const double tw = fxh_now();
            pthread_rwlock_rdlock(&S->gate);
            t_wait += fxh_now() - tw;
            memcpy(S->map + off, src, len);
            /* the pages stay in the file; only this process's view of them goes, now and by this thread, instead of at exit and by one */
            const uint64_t a = (off + 4095u) & ~(uint64_t)4095u, b = (off + len) & ~(uint64_t)4095u;
            if (b > a) (void)madvise(S->map + a, (size_t)(b - a), MADV_DONTNEED);
            pthread_rwlock_unlock(&S->gate);
        }
    } else if (len) {
        size_t done = 0;
        while (done < len && !FXH_ABORTED()) {
            const ssize_t k = pwrite(S->out_fd, src + done, len - done, (off_t)(off + done));
            if (k < 0) { if (errno == EINTR) continue; err(1, "writing output failed"); }
            done += (size_t)k;
        }
    }
    const double dt = fxh_now() - t0 - t_wait;
    pthread_mutex_lock(&S->mu); S->t_copy += dt; S->t_copy_wait += t_wait; pthread_mutex_unlock(&S->mu);
    pthread_mutex_lock(&s->mu);
    s->out_full[j->slot] = 0;
    pthread_cond_broadcast(&s->cv);
    pthread_mutex_unlock(&s->mu);
}

static void *fxh_strand_gpu(void *arg)
{
    fxh_strand *s = (fxh_strand *)arg;
    fxh_sf *S = s->S;
    fxh_lane *ln = &s->ln;
    fxh_lane_open_ctx(ln);
    for (int k = 0; k < 2; ++k) (void)fxg_host_register(ln->st.ctx, s->in[k], S->in_cap);      /* page-locked: the upload is real DMA */
    int j = 0;
    for (int k = 0;; k ^= 1) {
        double t0 = fxh_now();
        pthread_mutex_lock(&s->mu);
        while (!s->in_full[k] && !FXH_ABORTED()) pthread_cond_wait(&s->cv, &s->mu);
        const uint64_t t = s->in_ticket[k];
        const size_t len = s->in_len[k];
        pthread_mutex_unlock(&s->mu);
        s->t_wait_in += fxh_now() - t0;
        if (FXH_ABORTED() || t == FXH_TICKET_DONE) break;
        t0 = fxh_now();
        pthread_mutex_lock(&s->mu);
        while (!S->arena && s->out_full[j] && !FXH_ABORTED()) pthread_cond_wait(&s->cv, &s->mu);
        pthread_mutex_unlock(&s->mu);
        s->t_wait_out += fxh_now() - t0;
        if (FXH_ABORTED()) break;
        t0 = fxh_now();
        ln->text_base = NULL; ln->text_cap = 0;
        ln->text = s->in[k]; ln->len = len; ln->records = FXH_RECORDS_UNKNOWN; ln->slot = j;
        s->cur_ticket = t;
        fxh_lane_run(ln);
        s->t_gpu += fxh_now() - t0;
        int ok = ln->handled;
        if (ok && S->clip_auto) {                            /* the clipper's lanes are exact while ALL reads have one length: every chunk the same one */
            pthread_mutex_lock(&S->mu);
            if (!ln->fixed_len) ok = 0;
            else if (!S->clip_len) S->clip_len = ln->fixed_len;
            else if (S->clip_len != ln->fixed_len) ok = 0;
            pthread_mutex_unlock(&S->mu);
        }
        if (!ok) { fxh_sf_abort(S); break; }                 /* whatever it is, the one-stream run owns the reference's behaviour for it */
        fxh_add_counters(&s->tot, ln->ctr, ln->records, ln->lpr == 2 ? ln->weighted : NULL);
        s->chunks++;
        pthread_mutex_lock(&s->mu);                          /* the text is on the device: the reader may fill the buffer again */
        s->in_full[k] = 0;
        pthread_cond_broadcast(&s->cv);
        pthread_mutex_unlock(&s->mu);
        if (S->arena) continue;                              /* rank mode: the text is in the arena already (fxh_sf_place) */
        t0 = fxh_now();
        pthread_mutex_lock(&S->mu);
        while (S->scanned <= t && !FXH_ABORTED()) pthread_cond_wait(&S->cv, &S->mu);
        const uint64_t off = S->offset[t];
        pthread_mutex_unlock(&S->mu);
        s->t_wait_off += fxh_now() - t0;
        if (FXH_ABORTED()) break;
        pthread_mutex_lock(&s->mu);
        s->out_len[j] = ln->out_len; s->out_off[j] = off; s->out_full[j] = 1;
        pthread_cond_broadcast(&s->cv);
        pthread_mutex_unlock(&s->mu);
        s->wjob[j].s = s; s->wjob[j].slot = j;
        fxh_pool_submit(&S->wpool, fxh_sf_write_task, &s->wjob[j]);
        j = (j + 1) % S->nout;
    }
    pthread_mutex_lock(&s->mu);                              /* the strand's text is in the file (or the run is over) before its buffers may go */
    for (;;) {
        int busy = 0;
        for (int q = 0; q < S->nout; ++q) busy |= s->out_full[q];
        if (!busy || FXH_ABORTED()) break;
        pthread_cond_wait(&s->cv, &s->mu);
    }
    pthread_mutex_unlock(&s->mu);
    if (S->release && !FXH_ABORTED()) {
        const double t0 = fxh_now();
        for (int k = 0; k < 2; ++k) (void)fxg_host_unregister(ln->st.ctx, s->in[k]);
        fxh_lane_release(ln);
        s->t_release = fxh_now() - t0;
    }
    return NULL;
}

/* the cuts, found side by side before anything runs */
typedef struct { int fd, lpr; off_t start, size; size_t chunk; off_t *cut; uint64_t c0, c1; int bad; } fxh_cutjob;
static void *fxh_cut_main(void *arg)
{
    fxh_cutjob *j = (fxh_cutjob *)arg;
    for (uint64_t c = j->c0; c < j->c1 && !j->bad; ++c) {
        const off_t from = j->start + (off_t)(c * (uint64_t)j->chunk);
        off_t f = fxh_find_cut(j->fd, from, j->size, j->lpr, (size_t)64 << 10);
        if (f < 0) f = fxh_find_cut(j->fd, from, j->size, j->lpr, 0);
        if (f < 0) f = j->size;                              /* no record starts behind it: the chunk in front runs to the end (and is then checked like any other) */
        j->cut[c] = f;
    }
    return NULL;
}

/* The chunk boundaries of one byte range [lo, hi) of the input: cut[0] = lo, cut[n] = hi, every cut in between the start of a record (found by pattern, several
 * threads).  Returns the cuts (caller frees) with *nchunks_io / *longest set, or NULL when the range cannot be cut into chunks of about `chunk` bytes (records
 * longer than a chunk, no record pattern in reach): such an input runs as one stream. */
static off_t *fxh_range_cuts(int fd, int lpr, off_t lo, off_t hi, off_t file_size, size_t chunk, uint64_t *nchunks_io, size_t *longest_out)
{
    uint64_t nchunks = *nchunks_io;
    off_t *cut = (off_t *)calloc(nchunks + 1, sizeof(off_t));
    if (!cut) err(1, "out of memory");
    cut[0] = lo; cut[nchunks] = hi;
    fxh_cutjob cj[8];
    pthread_t th[8];
    const int nt = nchunks > 64 ? 8 : 1;
    for (int i = 0; i < nt; ++i) {
        cj[i].fd = fd; cj[i].lpr = lpr; cj[i].start = lo; cj[i].size = file_size; cj[i].chunk = chunk; cj[i].cut = cut; cj[i].bad = 0;
        cj[i].c0 = 1 + (nchunks - 1) * (uint64_t)i / (uint64_t)nt; cj[i].c1 = 1 + (nchunks - 1) * (uint64_t)(i + 1) / (uint64_t)nt;
    }
    for (int i = 1; i < nt; ++i) if (pthread_create(&th[i], NULL, fxh_cut_main, &cj[i]) != 0) err(1, "pthread_create");
    fxh_cut_main(&cj[0]);
    for (int i = 1; i < nt; ++i) pthread_join(th[i], NULL);
    int bad = 0;
    for (int i = 0; i < nt; ++i) bad |= cj[i].bad;
    while (!bad && nchunks > 1 && cut[nchunks - 1] >= hi) nchunks--;      /* a last nominal cut whose record start is the range's end: no chunk there */
    cut[nchunks] = hi;
    size_t longest = 0;
    for (uint64_t c = 0; c < nchunks && !bad; ++c) {
        if (cut[c + 1] <= cut[c]) bad = 1;               /* records longer than a chunk, or no record pattern in reach: one stream */
        else if ((size_t)(cut[c + 1] - cut[c]) > longest) longest = (size_t)(cut[c + 1] - cut[c]);
    }
    if (bad || longest > chunk + chunk / 2) { free(cut); return NULL; }
    *nchunks_io = nchunks; *longest_out = longest;
    return cut;
}

static long fxh_env_long(const char *name, long dflt, long lo, long hi)
{
    const char *e = getenv(name);
    long v = e && *e ? atol(e) : dflt;
    if (v < lo) v = lo;
    if (v > hi) v = hi;
    return v;
}

/* the counter block a rank contributes to the job's one all-gather (u64[FXG_NCOUNTERS], text level): records and reads in and out, the BYTES of its
 * formatted output where the batch ABI has kept bases -- so that fxg_epilogue's exclusive scan is the rank's offset in the file -- and the -v tallies */
enum { FXH_B_IN_SEQ = FXG_C_INPUT, FXH_B_OUT_SEQ = FXG_C_KEPT, FXH_B_OUT_BYTES = FXG_C_KEPT_BASES, FXH_B_BAD = FXG_C_ERRORS,
       FXH_B_IN_READS = 17, FXH_B_OUT_READS = 18, FXH_B_CLIP_IN = 19, FXH_B_CLIP_LEN = 20 };
#define FXH_BAD_IRREGULAR ((uint64_t)1 << 40)      /* (above the device's own error bits) */

static void fxh_totals_add(fxh_totals *tot, const fxh_totals *t)
{
    tot->input_sequences += t->input_sequences; tot->input_reads += t->input_reads; tot->output_sequences += t->output_sequences; tot->output_reads += t->output_reads;
    tot->clip_input += t->clip_input; tot->clip_too_short += t->clip_too_short; tot->clip_adapter_only += t->clip_adapter_only;
    tot->clip_no_adapter += t->clip_no_adapter; tot->clip_adapter_found += t->clip_adapter_found; tot->clip_n += t->clip_n;
    tot->masked_reads += t->masked_reads; tot->masked_nucleotides += t->masked_nucleotides; tot->qtrim_dropped += t->qtrim_dropped;
}

typedef struct { fxh_sf *S; const char *src; size_t len; uint64_t off; int *busy; } fxh_djob;
static void fxh_sf_drain_task(void *arg)
{
    fxh_djob *j = (fxh_djob *)arg;
    int e = 0;
    if (j->S->map) {                             /* pages that exist (rank 0 made them): a copy, and this thread drops its own page-table entries */
        char *dst = j->S->map + (j->off - j->S->map_base);
        memcpy(dst, j->src, j->len);
        const uintptr_t a = ((uintptr_t)dst + 4095u) & ~(uintptr_t)4095u, b = ((uintptr_t)dst + j->len) & ~(uintptr_t)4095u;
        if (b > a) (void)madvise((void *)a, (size_t)(b - a), MADV_DONTNEED);
    } else {
        errno = 0;
        e = fxg_concat_pwrite(j->S->out_fd, j->src, j->len, j->off) != 0 ? (errno ? errno : EIO) : 0;
    }
    pthread_mutex_lock(&j->S->mu);
    if (e && !j->S->drain_errno) j->S->drain_errno = e;       /* reported to the job (the second exchange), not died of: the other ranks are waiting */
    *j->busy = 0;
    pthread_cond_broadcast(&j->S->cv);
    pthread_mutex_unlock(&j->S->mu);
}

/* rank mode: nobody waits for a dead rank for ever.  An all-gather that a rank never joins does not return (RCCL has no time-out of its own), so every
 * exchange runs under a watch: FXH_RANK_TIMEOUT seconds (default 900) without the other ranks' answer end this rank with a message and exit code 1. */
typedef struct { pthread_mutex_t mu; pthread_cond_t cv; pthread_t th; int done, secs, rank, world; const char *what; } fxh_watch;
static void *fxh_watch_main(void *arg)
{
    fxh_watch *w = (fxh_watch *)arg;
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_sec += w->secs;
    pthread_mutex_lock(&w->mu);
    int rc = 0;
    while (!w->done && rc != ETIMEDOUT) rc = pthread_cond_timedwait(&w->cv, &w->mu, &ts);
    const int late = !w->done;
    pthread_mutex_unlock(&w->mu);
    if (late) {
        warnx("rank %d of %d: no answer from the other ranks within %d s (%s): a rank has died or is stuck (FXH_RANK_TIMEOUT)", w->rank, w->world, w->secs, w->what);
        fflush(NULL);
        _exit(1);
    }
    return NULL;
}
static void fxh_watch_start(fxh_watch *w, int rank, int world, const char *what)
{
    memset(w, 0, sizeof *w);
    pthread_mutex_init(&w->mu, NULL); pthread_cond_init(&w->cv, NULL);
    w->secs = (int)fxh_env_long("FXH_RANK_TIMEOUT", 900, 1, 7 * 86400); w->rank = rank; w->world = world; w->what = what;
    if (pthread_create(&w->th, NULL, fxh_watch_main, w) != 0) err(1, "pthread_create");
}
static void fxh_watch_stop(fxh_watch *w)
{
    pthread_mutex_lock(&w->mu);
    w->done = 1;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
    pthread_join(w->th, NULL);
}

/* one exchange of the job: this rank's block up, ncclAllGather (fxg_epilogue_rccl), every rank's block and the totals back */
static void fxh_rank_exchange(fxh_sf *S, fxh_lane *rl, fxg_comm *comm, uint64_t *d_block, const uint64_t *blk, uint64_t *totals, uint64_t *byte_off, uint64_t *gathered,
                              const char *what)
{
    fxh_watch w;
    uint64_t read_off = 0;
    fxh_watch_start(&w, S->rank, S->world, what);
    FXG_CHECK(&rl->st, fxg_memcpy_h2d(S->main_ctx, d_block, blk, FXG_NCOUNTERS * sizeof(uint64_t)));
    const int erc = fxg_epilogue_rccl(S->main_ctx, comm, d_block, totals, &read_off, byte_off, gathered);
    fxh_watch_stop(&w);
    if (erc != 0) errx(1, "rank %d of %d: %s failed (%d): %s", S->rank, S->world, what, erc, fxg_last_error(S->main_ctx));
}

static int fxh_one_file_attempt(FASTX *fx, const fxg_params *p, fxh_totals *tot, int rank, int world);

/* 0 = done (in the child of the fork below: the caller goes on to print its reports); -1 = run as one stream (not eligible, or the attempt was abandoned).
 * FXH_WORLD = n > 1 with FXH_RANK = 0 .. n-1: n processes, one per GPU (FXG_DEVICE, default rank mod #GPUs), started by any launcher -- or by hand --
 * with the SAME command line; they meet through FXH_RENDEZVOUS (default: <output>.rdv).  Rank 0 prints the -v report of the whole job. */
int fxh_run_one_file(FASTX *fx, const fxg_params *p, fxh_totals *tot)
{
    const int world = (int)fxh_env_long("FXH_WORLD", 1, 1, 4096), rank = (int)fxh_env_long("FXH_RANK", 0, 0, world - 1);
    if (world > 1 && rank == 0) {
        /* A rendezvous record that a killed run of the same command left under the same name (no FXG_COMM_JOB: the job token is 0 both times) must be gone before
         * any other rank of THIS job can look at it: rank 0 removes the name here, first thing -- fxg_comm_create does it again, but only after the runtime and
         * RCCL have started, seconds during which a rank that is already polling could read the dead job's id twice unchanged and take it (advisor, round 5). */
        char rdv[PATH_MAX + 16];
        const char *re = getenv("FXH_RENDEZVOUS");
        if (re && *re) snprintf(rdv, sizeof rdv, "%s", re); else snprintf(rdv, sizeof rdv, "%s.rdv", fx->output_file_name);
        (void)unlink(rdv);
    }
    const int rc = fxh_one_file_attempt(fx, p, tot, rank, world);
    if (world > 1 && rc != 0 && rank > 0) {      /* not a job for ranks (a pipe, a tiny input) or abandoned: rank 0 runs it as one stream, the reference's way */
        if (fx->writer && fx->writer->fd >= 0 && fx->writer->fd != STDOUT_FILENO) close(fx->writer->fd);
        fx->writer->fd = -1;
        fflush(NULL);
        _exit(0);
    }
    return rc;
}

static int fxh_one_file_attempt(FASTX *fx, const fxg_params *p, fxh_totals *tot, const int rank, const int world)
{
    struct fxh_reader *rd = fx->reader;
    const int ranked = world > 1 || getenv("FXH_RANK_MODE") != NULL;      /* (FXH_RANK_MODE=1: the rank path with a world of one -- the GPU tier's way to the real RCCL) */
    struct fxh_writer *w0 = fx->writer;
    struct stat sb, ob;
    const char *sw = getenv("FXH_ONE_FILE");
    if (sw && atoi(sw) == 0) return -1;
    if (fstat(rd->fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return -1;
    if (strcmp(fx->output_file_name, "-") == 0 || fx->compress_output || g_rename_ids || getenv("FXH_HOST_PARSE")) return -1;
    if (!w0 || w0->fd < 0 || !w0->positional || w0->len != 0 || fstat(w0->fd, &ob) != 0 || !S_ISREG(ob.st_mode)) return -1;
    if (w0->off != 0) return -1;                 /* (the strands and the rank drain place their bytes counted from the file's first byte: a writer that does not start there runs as one stream) */
    if (ob.st_dev == sb.st_dev && ob.st_ino == sb.st_ino) return -1;
    const int clip = (p->stages & FXG_STAGE_CLIP) != 0;
    if (clip && getenv("FXH_CLIP_SERIAL") != NULL && getenv("FXH_CLIP_PARALLEL") == NULL) return -1;      /* one aligner asked for */
    if (g_hip_touched) return -1;                /* no fork over a live runtime (a host that calls in twice) */
    const off_t pos = lseek(rd->fd, 0, SEEK_CUR);
    if (pos < 0) return -1;
    const off_t start = pos - (off_t)(rd->end - rd->beg), size = sb.st_size;      /* where the unread input begins in the file */
    if (start < 0 || start >= size) return -1;
    /* from about half a gigabyte on the strands pay for their contexts: 0.64 GB 0.193 against 0.197 s, 1.3 GB 0.246 / 0.197, 2.6 GB 0.361 / 0.239,
     * 5.1 GB 0.660 / 0.364, 20.5 GB 1.79 / 1.13 (one stream / this run, profiles/r05/l_e2e_one_file_by_size.txt) */
    const long min_mb = fxh_env_long("FXH_ONE_FILE_MIN_MB", 512, 0, 1 << 30);
    if (!ranked && (long long)(size - start) < ((long long)min_mb << 20)) return -1;
    const int lpr = fx->read_fastq ? 4 : 2;
    /* rank mode: this process takes byte range `rank` of `world` (cut at record starts found by pattern, as the parts of fxh_parts.c; every rank computes
     * the same cuts); from here on `start` .. `size` is that range */
    const off_t file_size = size;
    off_t my_start = start, my_end = size;
    size_t chunk = (size_t)fxh_env_long("FXH_STRAND_KB", 0, 0, 1 << 22) << 10;     /* (tests: chunks of a few KB) */
    if (!chunk) chunk = (size_t)fxh_env_long("FXH_STRAND_MB", 16, 1, 1024) << 20;      /* 16 MB: 54.6 against 51.7 Mreads/s with 8 (profiles/r05/f_e2e_one_file_timeline.txt) */
    off_t *cut = NULL;
    uint64_t nchunks = 0;
    size_t S_longest_ = 0;
    if (world > 1) {
        /* Whether the job runs by ranks is decided HERE, before any rank meets another, and it must be the same decision in every rank: a rank that left
         * on a check of its own range alone would leave the others waiting in the rendezvous (the communicator has no watch of its own).  So every rank
         * looks at EVERY rank's range -- the same cuts, the same verdict (a few thousand small reads for a file of 100 GB) -- and keeps the cuts of its own. */
        off_t lo = start;
        for (int g = 1; g <= world; ++g) {
            const off_t hi = g == world ? file_size : fxh_find_cut(rd->fd, start + (off_t)((unsigned long long)(file_size - start) * (unsigned)g / (unsigned)world), file_size, lpr, 0);
            if (hi < 0 || hi <= lo) { free(cut); return -1; }      /* an input too small for that many ranks */
            uint64_t n = ((uint64_t)(hi - lo) + chunk - 1) / chunk;
            size_t longest = 0;
            off_t *c = fxh_range_cuts(rd->fd, lpr, lo, hi, file_size, chunk, &n, &longest);
            if (!c) { free(cut); return -1; }              /* some rank's range cannot be cut: no rank starts, rank 0 runs the input as one stream */
            if (g - 1 == rank) { my_start = lo; my_end = hi; cut = c; nchunks = n; S_longest_ = longest; } else free(c);
            lo = hi;
        }
        chunk = S_longest_;                                  /* (now: the buffer a chunk of THIS rank needs) */
    } else {
        nchunks = ((uint64_t)(my_end - my_start) + chunk - 1) / chunk;
        if (nchunks < 2 && !ranked) return -1;
        size_t longest = 0;
        cut = fxh_range_cuts(rd->fd, lpr, my_start, my_end, file_size, chunk, &nchunks, &longest);
        if (!cut) return -1;
        chunk = longest;                                     /* (now: the buffer a chunk needs) */
    }
    /* The attempt runs in a CHILD process, like the sharded run's (fxh_parts.c): anything irregular abandons it, the child empties the
     * file and exits with FXH_EXIT_ABANDON, and this process -- which has not touched the GPU -- runs the same input as one stream. */
    fflush(NULL);
    const double t_fork = fxh_now();
    const pid_t child = fork();
    if (child < 0) {
        /* no process to run the attempt in.  Alone, that means one stream.  In a job the other ranks are on their way to the rendezvous: this rank cannot
         * take part (no fork over a live runtime: see above), so it says so and ends the job -- the others' watch (fxh_watch around the rendezvous) ends them. */
        free(cut);
        if (world > 1) err(1, "rank %d of %d: fork", rank, world);
        return -1;
    }
    if (child > 0) {
        int st = 0;
        free(cut);
        while (waitpid(child, &st, 0) < 0) { if (errno != EINTR) err(1, "waitpid"); }
        if (getenv("FXH_TIMING")) fprintf(stderr, "fxh timing one file: the child was gone %.3f s after the fork, at %.3f (CLOCK_MONOTONIC)\n", fxh_now() - t_fork, fxh_now());
        if (WIFEXITED(st) && WEXITSTATUS(st) == FXH_EXIT_ABANDON) {
            if (rank == 0 && ftruncate(w0->fd, w0->off) != 0) warn("%s", fx->output_file_name);      /* (no rank has written: the file is rank 0's again) */
            return -1;
        }
        if (WIFSIGNALED(st)) { signal(WTERMSIG(st), SIG_DFL); raise(WTERMSIG(st)); _exit(128 + WTERMSIG(st)); }
        _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 1);          /* the child printed the reports and closed the file */
    }
    (void)prctl(PR_SET_PDEATHSIG, SIGTERM);      /* the child: a tool process that was killed takes its attempt along */
    const int timing = getenv("FXH_TIMING") != NULL;
    const double t_run0 = fxh_now();
    static fxh_sf S_;                            /* (static: zeroed, and alive for the whole child) */
    fxh_sf *S = &S_;
    S->fx = fx; S->p = p; S->in_fd = rd->fd; S->lpr = lpr; S->cut = cut; S->nchunks = nchunks; S->in_total = (uint64_t)(my_end - my_start);
    S->in_cap = (chunk + 4096 + 4095) & ~(size_t)4095;
    S->clip_auto = clip && getenv("FXH_CLIP_PARALLEL") == NULL;
    S->out_fd = w0->fd;
    pthread_mutex_init(&S->mu, NULL); pthread_cond_init(&S->cv, NULL);
    S->size = (uint64_t *)calloc(nchunks, sizeof(uint64_t)); S->offset = (uint64_t *)calloc(nchunks, sizeof(uint64_t)); S->have = (uint8_t *)calloc(nchunks, 1);
    if (!S->size || !S->offset || !S->have) err(1, "out of memory");
    __atomic_store_n(&g_parts_abort, 0, __ATOMIC_RELAXED);
    g_parts_mode = 1;

    int dev[FXH_MAX_LANES];
    int ndev = fxh_device_list(dev, FXH_MAX_LANES);
    S->rank = rank; S->world = world;
    if (ranked) {                                /* one process per GPU: FXG_DEVICE if the launcher set it, else the rank's turn among the GPUs of the box */
        if (!getenv("FXG_DEVICE") && !getenv("FXG_DEVICES")) { const int nd = fxg_device_count(); g_hip_touched = 1; dev[0] = nd > 0 ? rank % nd : 0; }
        ndev = 1;
    }
    /* the sink: a tmpfs file written from offset 0 gets the gated mapping, everything else positional writes */
    {
        struct statfs fs;
        const char *sk = getenv("FXH_ONE_FILE_SINK");        /* "map" | "pwrite": force one (tests) */
        int want_map = sk ? strcmp(sk, "map") == 0 : (fstatfs(w0->fd, &fs) == 0 && (unsigned long)fs.f_type == (unsigned long)TMPFS_MAGIC);
        if (w0->off != 0) want_map = 0;
        S->alloc_in_total = S->in_total;
        S->window = (uint64_t)fxh_env_long("FXH_ONE_FILE_WINDOW_MB", 128, 1, 1 << 16) << 20;
        S->ratio_after = (uint64_t)fxh_env_long("FXH_ONE_FILE_RATIO_MB", 64, 0, 1 << 20) << 20;
        S->keep_surplus = getenv("FXH_ONE_FILE_KEEP_SURPLUS") != NULL;
        if (want_map && ranked) {
            /* A rank job: the ranks' text stays in HBM until the exchange, and then every rank wants the file at once -- through the one inode lock if the pages
             * still have to be made.  So rank 0 makes the JOB's pages meanwhile: the same eager allocator, alone on the file (nobody copies before the exchange),
             * towards the output expected from the WHOLE input at rank 0's own measured ratio, to the exact size once the exchange has said it.  The ranks then
             * copy into pages that exist, each through a mapping of its own slice (separate processes: separate page tables). */
            want_map = 0;
            if (rank == 0) {
                const uint64_t whole = (uint64_t)(file_size - start);
                S->alloc_in_total = whole;
                S->map_len = (whole + whole / 7 + (1u << 20) + 4095u) & ~(uint64_t)4095u;
                if (fallocate(w0->fd, 0, 0, 4096) == 0) {
                    S->prealloc = 1; S->alloc_end = 4096;
                    pthread_rwlock_init(&S->gate, NULL);
                    if (pthread_create(&S->th_alloc, NULL, fxh_sf_alloc_main, S) != 0) err(1, "pthread_create");
                }
            }
        }
        if (want_map) {
            S->map_len = (S->in_total + S->in_total / 7 + (1u << 20) + 4095u) & ~(uint64_t)4095u;      /* an empty third line still gets its '+': at most 8/7 of the input */
            void *m = MAP_FAILED;
            if (ftruncate(w0->fd, (off_t)S->map_len) == 0) m = mmap(NULL, (size_t)S->map_len, PROT_READ | PROT_WRITE, MAP_SHARED, w0->fd, 0);
            if (m != MAP_FAILED && fallocate(w0->fd, 0, 0, 4096) == 0) {
                S->map = (char *)m; S->mapped = 1; S->alloc_end = 4096;
                pthread_rwlockattr_t ra;
                pthread_rwlockattr_init(&ra);
                pthread_rwlockattr_setkind_np(&ra, PTHREAD_RWLOCK_PREFER_WRITER_NONRECURSIVE_NP);      /* the allocator is one against many: it goes first */
                pthread_rwlock_init(&S->gate, &ra);
                if (pthread_create(&S->th_alloc, NULL, fxh_sf_alloc_main, S) != 0) err(1, "pthread_create");      /* from the first millisecond on */
            } else {                                          /* no mapping or no fallocate() here: positional writes */
                if (m != MAP_FAILED) munmap(m, (size_t)S->map_len);
                if (ftruncate(w0->fd, 0) != 0) warn("%s", fx->output_file_name);
            }
        }
    }

    /* Placement AFTER the allocator has started: finding the GPU's NUMA node can take 50 ms (the runtime has to be asked where visibility variables hide the
     * topology), which is a gigabyte of pages at the allocator's rate.  The allocator thread then follows the calling thread onto the GPU's node. */
    cpu_set_t cpus_before;
    const double t_dev = fxh_now();
    if (ndev == 1) (void)fxh_bind_near_device(dev[0], &cpus_before);      /* buffers and the output's pages are touched (and page-locked) on the GPU's node; every thread below inherits it */
    if (S->mapped || S->prealloc) { cpu_set_t now_set; if (sched_getaffinity(0, sizeof now_set, &now_set) == 0) (void)pthread_setaffinity_np(S->th_alloc, sizeof now_set, &now_set); }
    const double t_bound = fxh_now();


    /* Four strands per GPU, four preads in flight each (16 reading threads: what one tmpfs file gives, 31 GB/s).  Four feed the sink as well as six or eight -- it is the
     * sink that bounds a run whose output is large -- and cost less to start and to take down: 64 M reads 60.8 / 59.3 / 54.9 Mreads/s with 4 / 6 / 8 strands, the child gone
     * 1.04 / 1.06 / 1.15 s after the fork (profiles/r05/r_e2e_one_file_strands.txt); 2.6 GB of input 0.239 against 0.268 s (l_). */
    int per = (int)fxh_env_long("FXH_STRANDS", 4, 1, FXH_MAX_STRANDS);
    int ns = per * ndev;
    if (ns > FXH_MAX_STRANDS) ns = FXH_MAX_STRANDS;
    if ((uint64_t)ns > nchunks) ns = (int)nchunks;
    S->nstrands = ns;
    S->nread_slices = (int)fxh_env_long("FXH_STRAND_READERS", 4, 1, 16);
    S->st = (fxh_strand *)calloc((size_t)ns, sizeof(fxh_strand));
    if (!S->st) err(1, "out of memory");
    fxh_pool_start(&S->rpool, (int)fxh_env_long("FXH_IO_THREADS", ns * (S->nread_slices - 1) > 0 ? ns * (S->nread_slices - 1) : 1, 1, 64), (unsigned)(ns * 16));
    /* copies into the mapping: one thread per output buffer; positional writes: ONE stream (more of them only queue at the inode lock, 27 against 40 Mreads/s) */
    S->release = (int)fxh_env_long("FXH_STRAND_RELEASE", 0, 0, 1);
    S->nout = (int)fxh_env_long("FXH_STRAND_OUT_SLOTS", 4, 2, FXH_LANE_OUT_SLOTS);      /* output buffers per strand: what the strands can put aside while the allocator has the file */
    fxh_pool_start(&S->wpool, S->mapped ? (int)fxh_env_long("FXH_COPY_THREADS", 16, 1, 64) : 1, (unsigned)(S->nout * ns));
    const int revcomp = (p->stages & (FXG_STAGE_REVCOMP | FXG_STAGE_MASK)) != 0;
    fxg_comm *comm = NULL;
    fxh_lane rank_lane;                          /* rank mode: the context that owns the arena and the communicator */
    memset(&rank_lane, 0, sizeof rank_lane);
    uint64_t *d_block = NULL;
    if (ranked) {
        rank_lane.device = dev[0];
        fxh_lane_open_ctx(&rank_lane);
        S->main_ctx = rank_lane.st.ctx;
        S->arena_cap = S->in_total + S->in_total / 7 + (1u << 20);
        if (fxg_malloc_device(S->main_ctx, (size_t)S->arena_cap, (void **)&S->arena) != 0 || !S->arena)
            errx(1, "rank %d of %d: %.1f GB of device memory for this rank's share of the output are not to be had (%s); start more ranks", rank, world,
                 1e-9 * (double)S->arena_cap, fxg_last_error(S->main_ctx));
        FXG_CHECK(&rank_lane.st, fxg_malloc_device(S->main_ctx, FXG_NCOUNTERS * sizeof(uint64_t), (void **)&d_block));
        /* stdout is the tool's data and report channel (the -v report goes there when -o names a file): a collective library told to talk
         * (NCCL_DEBUG=VERSION / INFO in the job's environment) talks to stderr, unless the user has sent it somewhere already */
        if (getenv("NCCL_DEBUG") && !getenv("NCCL_DEBUG_FILE")) (void)setenv("NCCL_DEBUG_FILE", "/dev/stderr", 0);
        char rdv[PATH_MAX + 16];
        const char *re = getenv("FXH_RENDEZVOUS");
        if (re && *re) snprintf(rdv, sizeof rdv, "%s", re); else snprintf(rdv, sizeof rdv, "%s.rdv", fx->output_file_name);
        /* (and whatever it prints unasked -- RCCL 2.26 greets with its version, the runtime's and the host name on stdout -- goes to stderr as well:
         * descriptor 1 is descriptor 2 while the communicator is made) */
        fflush(stdout);
        const int saved_out = dup(STDOUT_FILENO);
        if (saved_out >= 0) (void)dup2(STDERR_FILENO, STDOUT_FILENO);
        /* (under a watch like the exchanges: the rendezvous time-out covers the wait for the record, not a communicator that a missing rank never completes) */
        fxh_watch cw;
        fxh_watch_start(&cw, rank, world, "the rendezvous (making the communicator)");
        const int crc = fxg_comm_create(S->main_ctx, rdv, (uint32_t)rank, (uint32_t)world, (int)fxh_env_long("FXH_RENDEZVOUS_TIMEOUT", 120, 1, 86400), &comm);
        fxh_watch_stop(&cw);
        fflush(stdout);
        if (saved_out >= 0) { (void)dup2(saved_out, STDOUT_FILENO); close(saved_out); }
        if (crc != 0) errx(1, "rank %d of %d: no communicator (%d): %s", rank, world, crc, fxg_last_error(S->main_ctx));
    }
    for (int i = 0; i < ns; ++i) {
        fxh_strand *s = &S->st[i];
        s->id = i; s->S = S;
        pthread_mutex_init(&s->mu, NULL); pthread_cond_init(&s->cv, NULL);
        for (int k = 0; k < 2; ++k) if (posix_memalign((void **)&s->in[k], 4096, S->in_cap) != 0) err(1, "out of memory");
        fxh_lane *ln = &s->ln;
        ln->id = i; ln->device = dev[i % ndev]; ln->p = p; ln->revcomp = revcomp;
        ln->fwd_start = (p->stages & FXG_STAGE_FTRIM) && p->ft_first > 1 ? (uint32_t)p->ft_first - 1u : 0u;
        ln->qoffset = fx->fastq_ascii_quality_offset;
        ln->reverse = (p->stages & FXG_STAGE_REVCOMP) != 0; ln->lpr = lpr; ln->has_q = fx->read_fastq; ln->out_fasta = !fx->write_fastq;
        ln->clip_guard = S->clip_auto;
        ln->on_size = fxh_sf_publish; ln->owner = s;
        if (ranked) ln->on_place = fxh_sf_place;
        if (pthread_create(&s->th_rd, NULL, fxh_strand_reader, s) != 0) err(1, "pthread_create");      /* reading starts while the device does */
    }
    for (int i = 0; i < ns; ++i) if (pthread_create(&S->st[i].th_gpu, NULL, fxh_strand_gpu, &S->st[i]) != 0) err(1, "pthread_create");
    for (int i = 0; i < ns; ++i) { pthread_join(S->st[i].th_gpu, NULL); }
    int bad = FXH_ABORTED();
    if (bad) fxh_sf_abort(S);                    /* (readers that were between two checks) */
    for (int i = 0; i < ns; ++i) pthread_join(S->st[i].th_rd, NULL);
    fxh_pool_stop(&S->rpool);
    fxh_pool_stop(&S->wpool);
    pthread_mutex_lock(&S->mu);
    if (!bad && S->scanned != nchunks) bad = 1;
    if (!S->prealloc) S->alloc_stop = 1;        /* (a rank job's allocator goes on until the exchange has said the size) */
    pthread_cond_broadcast(&S->cv);
    pthread_mutex_unlock(&S->mu);
    if (S->mapped) pthread_join(S->th_alloc, NULL);
    uint64_t base = 0, job_total = S->scan_off;
    double t_drain = 0, t_drain0 = 0;
    int drained_by_copy = 0;
    fxh_totals mine;
    memset(&mine, 0, sizeof mine);
    for (int i = 0; i < ns; ++i) fxh_totals_add(&mine, &S->st[i].tot);
    if (ranked) {
        /* The exchange of the job: every rank's counter block, one ncclAllGather behind nothing (the strands have synchronised).  A rank that met
         * something irregular still takes part -- with its flag up -- so that ALL ranks leave together and rank 0 alone runs the input as one stream. */
        uint64_t blk[FXG_NCOUNTERS] = {0}, totals[FXG_NCOUNTERS], byte_off = 0;
        uint64_t *gathered = (uint64_t *)calloc((size_t)world * FXG_NCOUNTERS, sizeof(uint64_t));
        if (!gathered) err(1, "out of memory");
        blk[FXH_B_IN_SEQ] = mine.input_sequences; blk[FXH_B_OUT_SEQ] = mine.output_sequences; blk[FXH_B_OUT_BYTES] = S->scan_off;
        blk[FXH_B_IN_READS] = mine.input_reads; blk[FXH_B_OUT_READS] = mine.output_reads; blk[FXH_B_CLIP_IN] = mine.clip_input;
        blk[FXG_C_CLIP_TOO_SHORT] = mine.clip_too_short; blk[FXG_C_CLIP_ADAPTER_ONLY] = mine.clip_adapter_only; blk[FXG_C_CLIP_NO_ADAPTER] = mine.clip_no_adapter;
        blk[FXG_C_CLIP_ADAPTER_FOUND] = mine.clip_adapter_found; blk[FXG_C_CLIP_N] = mine.clip_n; blk[FXG_C_QTRIM_DROPPED] = mine.qtrim_dropped;
        blk[FXG_C_MASKED_READS] = mine.masked_reads; blk[FXG_C_MASKED_NT] = mine.masked_nucleotides;
        blk[FXH_B_CLIP_LEN] = S->clip_len; blk[FXH_B_BAD] = bad ? FXH_BAD_IRREGULAR : 0;
        fxh_rank_exchange(S, &rank_lane, comm, d_block, blk, totals, &byte_off, gathered, "the exchange of the counter blocks");
        if (totals[FXH_B_BAD]) bad = 1;
        if (S->clip_auto) {                      /* the clipper is exact across ranks while ALL reads of the job have one length */
            uint64_t len0 = 0;
            for (int g = 0; g < world; ++g) { const uint64_t l = gathered[(size_t)g * FXG_NCOUNTERS + FXH_B_CLIP_LEN]; if (!l) continue; if (!len0) len0 = l; else if (l != len0) bad = 1; }
        }
        base = byte_off; job_total = totals[FXH_B_OUT_BYTES];
        memset(&mine, 0, sizeof mine);           /* rank 0 reports the JOB */
        mine.input_sequences = totals[FXH_B_IN_SEQ]; mine.output_sequences = totals[FXH_B_OUT_SEQ]; mine.input_reads = totals[FXH_B_IN_READS]; mine.output_reads = totals[FXH_B_OUT_READS];
        mine.clip_input = (unsigned)totals[FXH_B_CLIP_IN]; mine.clip_too_short = (unsigned)totals[FXG_C_CLIP_TOO_SHORT]; mine.clip_adapter_only = (unsigned)totals[FXG_C_CLIP_ADAPTER_ONLY];
        mine.clip_no_adapter = (unsigned)totals[FXG_C_CLIP_NO_ADAPTER]; mine.clip_adapter_found = (unsigned)totals[FXG_C_CLIP_ADAPTER_FOUND]; mine.clip_n = (unsigned)totals[FXG_C_CLIP_N];
        mine.qtrim_dropped = totals[FXG_C_QTRIM_DROPPED]; mine.masked_reads = totals[FXG_C_MASKED_READS]; mine.masked_nucleotides = totals[FXG_C_MASKED_NT];
        /* rank 0: the job's pages, to the byte (or, abandoned: the allocator stops where it is; the file is emptied below) */
        int pages = 0, alloc_e = 0;
        if (S->prealloc) {
            pthread_mutex_lock(&S->mu);
            if (bad) S->alloc_stop = 1; else { S->alloc_final = job_total; S->alloc_final_set = 1; }
            pthread_cond_broadcast(&S->cv);
            pthread_mutex_unlock(&S->mu);
            pthread_join(S->th_alloc, NULL);
            alloc_e = S->alloc_errno;
            if (!bad && !alloc_e && ftruncate(w0->fd, (off_t)job_total) != 0) alloc_e = errno;      /* what the estimate overshot goes back */
            pages = !bad && !alloc_e;
        }
        if (!bad) {
            /* "the pages are there" (or not: another file system -- positional writes then): rank 0 says, everybody hears; also the barrier between the last
             * fallocate() and the first copy */
            uint64_t blkp[FXG_NCOUNTERS] = {0}, totalsp[FXG_NCOUNTERS], offp = 0;
            blkp[FXH_B_IN_SEQ] = (uint64_t)pages; blkp[FXH_B_BAD] = (uint64_t)alloc_e;
            fxh_rank_exchange(S, &rank_lane, comm, d_block, blkp, totalsp, &offp, gathered, "waiting for rank 0 to have made the output file's pages");
            if (totalsp[FXH_B_BAD]) {
                if (rank == 0) warnx("writing output failed: %s", strerror(alloc_e));
                fflush(NULL);
                _exit(1);
            }
            pages = gathered[FXH_B_IN_SEQ] != 0;         /* (rank 0's block) */
            /* this rank's text: down from the arena in pieces, each put where it belongs -- base (the bytes of the ranks before) + its place in the arena.
             * A piece that does not get into the file (no space, a file size limit) is not died of here: the other ranks are waiting for this one. */
            t_drain0 = fxh_now();
            if (pages && S->scan_off) {
                S->map_base = base & ~(uint64_t)4095u;
                void *m = mmap(NULL, (size_t)(base + S->scan_off - S->map_base), PROT_READ | PROT_WRITE, MAP_SHARED, w0->fd, (off_t)S->map_base);
                if (m != MAP_FAILED) S->map = (char *)m;      /* (no mapping -- a descriptor without read access --: positional writes into the same pages) */
            }
            if (!pages && ftruncate(w0->fd, (off_t)job_total) != 0) S->drain_errno = errno;      /* (every rank says the same size; no rank's bytes lie beyond it) */
            enum { NBMAX = 9 };
            const size_t piece = (size_t)fxh_env_long("FXH_DRAIN_MB", 32, 1, 1024) << 20;
            const int nth = (int)fxh_env_long("FXH_DRAIN_THREADS", S->map ? 4 : 2, 1, NBMAX - 1), NB = nth + 1;      /* copies want company, writers only queue at the inode */
            char *hb[NBMAX]; int busy[NBMAX] = {0}; fxh_djob dj[NBMAX];
            fxh_pool dpool;
            fxh_pool_start(&dpool, nth, (unsigned)NB);
            for (int k = 0; k < NB; ++k) FXG_CHECK(&rank_lane.st, fxg_malloc_host(S->main_ctx, piece, (void **)&hb[k]));
            int k = 0;
            for (uint64_t o = 0; o < S->scan_off; o += piece, k = (k + 1) % NB) {
                const size_t n = S->scan_off - o < piece ? (size_t)(S->scan_off - o) : piece;
                pthread_mutex_lock(&S->mu);
                while (busy[k]) pthread_cond_wait(&S->cv, &S->mu);
                const int failed = S->drain_errno;
                if (!failed) busy[k] = 1;
                pthread_mutex_unlock(&S->mu);
                if (failed) break;
                FXG_CHECK(&rank_lane.st, fxg_memcpy_d2h(S->main_ctx, hb[k], S->arena + o, n));
                FXG_CHECK(&rank_lane.st, fxg_sync(S->main_ctx));
                dj[k].S = S; dj[k].src = hb[k]; dj[k].len = n; dj[k].off = base + o; dj[k].busy = &busy[k];
                fxh_pool_submit(&dpool, fxh_sf_drain_task, &dj[k]);
            }
            fxh_pool_stop(&dpool);
            drained_by_copy = S->map != NULL;
            if (S->map) { munmap(S->map, (size_t)(base + S->scan_off - S->map_base)); S->map = NULL; }
            t_drain = fxh_now() - t_drain0;
            /* The job is done when EVERY rank's text is in the file, and rank 0's exit code says so: a second exchange, each rank's errno (0: written).  A rank
             * that died on the way never joins it -- the watch (or the transport) ends the wait -- so rank 0 never reports a file that has a hole as done. */
            uint64_t blk2[FXG_NCOUNTERS] = {0}, totals2[FXG_NCOUNTERS], off2 = 0;
            blk2[FXH_B_BAD] = (uint64_t)S->drain_errno;
            fxh_rank_exchange(S, &rank_lane, comm, d_block, blk2, totals2, &off2, gathered, "waiting for every rank to have written its part");
            if (totals2[FXH_B_BAD]) {
                if (S->drain_errno) warnx("rank %d of %d: writing output failed: %s", rank, world, strerror(S->drain_errno));
                if (rank == 0)
                    for (int g = 1; g < world; ++g) {
                        const uint64_t e = gathered[(size_t)g * FXG_NCOUNTERS + FXH_B_BAD];
                        if (e) warnx("rank %d of %d could not write its part of the output (%s): %s is incomplete", g, world, strerror((int)e), fx->output_file_name);
                    }
                fflush(NULL);
                _exit(1);
            }
        }
        free(gathered);
        fxg_comm_destroy(comm);
    }
    if (bad) {
        /* Abandoned.  Every thread has been joined, the contexts go, the file is emptied through its own descriptor -- which the parent
         * shares -- and the process leaves with _exit: no exit handler of this half-finished attempt gets to run. */
        for (int i = 0; i < ns; ++i) if (S->st[i].ln.st.ctx) fxg_ctx_destroy(S->st[i].ln.st.ctx);
        if (S->mapped) munmap(S->map, (size_t)S->map_len);
        if (S->main_ctx) fxg_ctx_destroy(S->main_ctx);
        w0->len = 0;
        if (rank == 0 && (ftruncate(w0->fd, w0->off) != 0 || lseek(w0->fd, w0->off, SEEK_SET) < 0)) warn("%s", fx->output_file_name);
        if (timing) fprintf(stderr, "fxh timing one file: abandoned, contexts destroyed, output emptied\n");
        fflush(NULL);
        _exit(FXH_EXIT_ABANDON);
    }
    const uint64_t total = job_total;
    if (S->mapped) {
        munmap(S->map, (size_t)S->map_len);
        if (ftruncate(w0->fd, (off_t)total) != 0) err(1, "writing output failed");
    }
    w0->off += (off_t)total;                     /* the writer closes with the descriptor where a write() stream would have left it */
    *tot = mine;
    fx->num_input_sequences = tot->input_sequences; fx->num_input_reads = tot->input_reads;
    fx->num_output_sequences = tot->output_sequences; fx->num_output_reads = tot->output_reads;
    if (timing) {
        double rd_s = 0, gpu_s = 0, win = 0, wout = 0, woff = 0, init = 0, rel = 0;
        for (int i = 0; i < ns; ++i) {
            const fxh_strand *s = &S->st[i];
            rd_s += s->t_read; gpu_s += s->t_gpu; win += s->t_wait_in; wout += s->t_wait_out; woff += s->t_wait_off; init += s->ln.t_init; rel += s->t_release;
            if (s->ln.t_call[7] > 0)
                fprintf(stderr, "fxh timing strand %d: %.0f chunks, ms per chunk: h2d %.3f index %.3f pack %.3f pipeline %.3f counters %.3f format %.3f d2h+sync %.3f\n", i, s->ln.t_call[7],
                        1e3 * s->ln.t_call[0] / s->ln.t_call[7], 1e3 * s->ln.t_call[1] / s->ln.t_call[7], 1e3 * s->ln.t_call[2] / s->ln.t_call[7], 1e3 * s->ln.t_call[3] / s->ln.t_call[7],
                        1e3 * s->ln.t_call[4] / s->ln.t_call[7], 1e3 * s->ln.t_call[5] / s->ln.t_call[7], 1e3 * s->ln.t_call[6] / s->ln.t_call[7]);
        }
        fprintf(stderr, "fxh timing one file (%d strands on %d GPU(s), %llu chunks, sink %s): run %.3f s (set-up %.3f, placement %.3f); summed over strands: context %.3f read %.3f wait-input %.3f device %.3f wait-outbuf %.3f wait-offset %.3f release %.3f; "
                        "sink: %llu fallocate calls %.3f s (to %.2f GB for %.2f GB of output, %.1f MB given back on the way), copies %.3f s + %.3f s at the gate (summed over %d threads)\n",
                ns, ndev, (unsigned long long)nchunks, S->mapped ? "gated mapping" : "pwrite", fxh_now() - t_run0, t_dev - t_run0, t_bound - t_dev, init, rd_s, win, gpu_s, wout, woff, rel,
                (unsigned long long)S->alloc_calls, S->t_alloc, 1e-9 * (double)S->alloc_end, 1e-9 * (double)total, 1e-6 * (double)S->given_back, S->t_copy, S->t_copy_wait, S->wpool.nth);
        if (ranked) fprintf(stderr, "fxh timing rank %d of %d: input bytes [%lld, %lld), %.3f GB of text held on the device, written at offset %llu of %llu in %.3f s (%s)\n", rank, world,
                               (long long)my_start, (long long)my_end, 1e-9 * (double)S->scan_off, (unsigned long long)base, (unsigned long long)job_total, t_drain,
                               drained_by_copy ? "copies into pages rank 0 made" : "positional writes");
        if (S->prealloc) fprintf(stderr, "fxh timing rank 0: the job's pages: %llu fallocate calls, %.3f s\n", (unsigned long long)S->alloc_calls, S->t_alloc);
    }
    if (rank > 0) {                              /* the job's report is rank 0's */
        if (w0->fd != STDOUT_FILENO) close(w0->fd);
        w0->fd = -1;
        fflush(NULL);
        _exit(0);
    }
    return 0;
}
