/* fxh_io.c -- I/O overlap of the batch path: one thread reads the next block while the current one is processed, another one writes the previous
 * output while the next is being formatted (fxh_priv.h). */
#include "fxh_priv.h"
int g_parts_mode;               /* a sharded run is under way (fxh_run_parts): smaller read-ahead per part */

/* Regular files are read with several pread() in flight (page-cache copies scale with threads; one read() stream is ~3 GB/s);
 * pipes and terminals keep the single read() loop. */
static size_t fxh_count_newlines(const char *p, size_t n)
{
    size_t c = 0;
    const char *e = p + n;
    while (p < e) {
        const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
        if (!q) break;
        c++;
        p = q + 1;
    }
    return c;
}

typedef struct { int fd; char *dst; size_t n; off_t off; size_t got, newlines; } fxh_pread_job;
static void *fxh_pread_main(void *arg)
{
    fxh_pread_job *j = (fxh_pread_job *)arg;
    j->got = 0;
    while (j->got < j->n) {
        ssize_t k = pread(j->fd, j->dst + j->got, j->n - j->got, j->off + (off_t)j->got);
        if (k < 0) { if (errno == EINTR) continue; err(1, "read failed"); }
        if (k == 0) break;
        j->got += (size_t)k;
    }
    j->newlines = fxh_count_newlines(j->dst, j->got);       /* the census the record cutter needs, while the slice is cache-warm */
    return NULL;
}

static int fxh_io_threads(void)
{
    const char *e = getenv("FXH_IO_THREADS");
    long n = e ? atol(e) : (g_parts_mode ? 4 : 8), ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    if (n < 1) n = 1;
    if (n > 16) n = 16;
    if (ncpu > 0 && n > ncpu) n = ncpu;
    return (int)n;
}

/* ---- a pipe as input ----
 * read() copies out of a pipe under the pipe's lock: one thread, 4.7 GB/s, and that was what `trimmer | filter` ran at.  splice() from a pipe into another pipe
 * moves page references and copies nothing, so the reading thread only deals the incoming pages out -- a pipe-capacity at a time, round robin -- to a few
 * private pipes, and one thread per private pipe does the copying, side by side, each piece straight to its place in the block.  (The pages may stay in the
 * private pipes a little longer than they would have stayed in the one they came through; a writer that hands over pages it will write to again is wrong with
 * any splicing reader -- pv is one -- and the tools here hand over pages of their own, fastx_io.c: fxh_pipe_write_all.) */
#define FXH_FAN_MAX 8
typedef struct { char *dst; size_t n; } fxh_fan_piece;
typedef struct {
    int rfd, wfd;
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    fxh_fan_piece q[64];
    unsigned head, count;
    int quit;
    size_t pending, newlines;          /* pieces dealt but not yet in the block; '\n' bytes among the finished ones */
} fxh_fan_lane;
struct fxh_fan { int n, next; size_t piece; fxh_fan_lane lane[FXH_FAN_MAX]; };

static void *fxh_fan_main(void *arg)
{
    fxh_fan_lane *L = (fxh_fan_lane *)arg;
    pthread_mutex_lock(&L->mu);
    for (;;) {
        while (L->count == 0 && !L->quit) pthread_cond_wait(&L->cv, &L->mu);
        if (L->count == 0) break;
        const fxh_fan_piece pc = L->q[L->head];
        L->head = (L->head + 1) % 64u; L->count--;
        pthread_cond_broadcast(&L->cv);
        pthread_mutex_unlock(&L->mu);
        size_t got = 0;
        while (got < pc.n) {
            const ssize_t k = read(L->rfd, pc.dst + got, pc.n - got);
            if (k < 0) { if (errno == EINTR) continue; err(1, "read failed"); }
            if (k == 0) errx(1, "read failed");      /* (cannot happen: the bytes were in the pipe when the piece was queued) */
            got += (size_t)k;
        }
        const size_t nl = fxh_count_newlines(pc.dst, pc.n);
        pthread_mutex_lock(&L->mu);
        L->newlines += nl; L->pending--;
        pthread_cond_broadcast(&L->cv);
    }
    pthread_mutex_unlock(&L->mu);
    return NULL;
}

static struct fxh_fan *fxh_fan_open(void)
{
    const char *e = getenv("FXH_PIPE_READERS");
    long n = e ? atol(e) : 3;
    if (n < 2 || getenv("FXH_NO_PIPE_FANOUT")) return NULL;
    if (n > FXH_FAN_MAX) n = FXH_FAN_MAX;
    struct fxh_fan *F = (struct fxh_fan *)calloc(1, sizeof *F);
    if (!F) return NULL;
    F->piece = (size_t)1 << 20;
    for (int i = 0; i < (int)n; ++i) {
        int pfd[2];
        if (pipe2(pfd, O_CLOEXEC) != 0) break;
        fxh_fan_lane *L = &F->lane[i];
        L->rfd = pfd[0]; L->wfd = pfd[1];
        const size_t cap = fxh_tune_pipe(L->wfd);
        if (cap && cap < F->piece) F->piece = cap;
        pthread_mutex_init(&L->mu, NULL); pthread_cond_init(&L->cv, NULL);
        if (pthread_create(&L->th, NULL, fxh_fan_main, L) != 0) { close(pfd[0]); close(pfd[1]); break; }
        F->n = i + 1;
    }
    if (F->n < 2) {                    /* (no descriptors, no threads: the plain loop) */
        for (int i = 0; i < F->n; ++i) { fxh_fan_lane *L = &F->lane[i]; pthread_mutex_lock(&L->mu); L->quit = 1; pthread_cond_broadcast(&L->cv); pthread_mutex_unlock(&L->mu); pthread_join(L->th, NULL); close(L->rfd); close(L->wfd); }
        free(F);
        return NULL;
    }
    return F;
}

static void fxh_fan_close(struct fxh_fan *F)
{
    if (!F) return;
    for (int i = 0; i < F->n; ++i) {
        fxh_fan_lane *L = &F->lane[i];
        pthread_mutex_lock(&L->mu); L->quit = 1; pthread_cond_broadcast(&L->cv); pthread_mutex_unlock(&L->mu);
        pthread_join(L->th, NULL);
        close(L->rfd); close(L->wfd);
    }
    free(F);
}

/* fills dst[0, cap) from the pipe `fd` (less only at its end).  -1: the descriptor cannot be spliced from (nothing has been taken): use read() */
static long long fxh_fan_fill(struct fxh_fan *F, int fd, char *dst, size_t cap, int *eof, size_t *newlines)
{
    size_t got = 0;
    while (got < cap) {
        fxh_fan_lane *L = &F->lane[F->next];
        const size_t want = cap - got < F->piece ? cap - got : F->piece;
        const ssize_t k = splice(fd, NULL, L->wfd, NULL, want, SPLICE_F_MOVE);
        if (k < 0) {
            if (errno == EINTR) continue;
            if (got == 0 && (errno == EINVAL || errno == ENOSYS || errno == EBADF)) return -1;
            err(1, "read failed");
        }
        if (k == 0) { *eof = 1; break; }
        pthread_mutex_lock(&L->mu);
        while (L->count == 64u) pthread_cond_wait(&L->cv, &L->mu);
        L->q[(L->head + L->count) % 64u].dst = dst + got; L->q[(L->head + L->count) % 64u].n = (size_t)k;
        L->count++; L->pending++;
        pthread_cond_broadcast(&L->cv);
        pthread_mutex_unlock(&L->mu);
        got += (size_t)k;
        F->next = (F->next + 1) % F->n;
    }
    size_t nl = 0;
    for (int i = 0; i < F->n; ++i) {   /* the block is whole when every piece is in place */
        fxh_fan_lane *L = &F->lane[i];
        pthread_mutex_lock(&L->mu);
        while (L->pending) pthread_cond_wait(&L->cv, &L->mu);
        nl += L->newlines; L->newlines = 0;
        pthread_mutex_unlock(&L->mu);
    }
    *newlines = nl;
    return (long long)got;
}

static void *fxh_prefetch_main(void *arg)
{
    fxh_prefetch *pf = (fxh_prefetch *)arg;
    pthread_mutex_lock(&pf->mu);
    for (;;) {
        while (pf->state != 1 && pf->state != 3) pthread_cond_wait(&pf->cv, &pf->mu);
        if (pf->state == 3) break;
        char *buf = pf->buf; const size_t cap = pf->cap;
        pthread_mutex_unlock(&pf->mu);
        size_t got = 0, newlines = (size_t)-1; int eof = 0;
        const size_t gap = pf->gap;
        if (pf->regular) {
            size_t want = cap - gap;
            if (pf->limit > 0) {                                       /* a part of a sharded run: the input ends at `limit` */
                if (pf->offset >= pf->limit) want = 0;
                else if ((off_t)want > pf->limit - pf->offset) want = (size_t)(pf->limit - pf->offset);
            }
            if (want == 0) { eof = 1; newlines = 0; }
            else {
                int nt = pf->io_threads;
                if ((size_t)nt > want / pf->io_slice) nt = (int)(want / pf->io_slice);
                if (nt < 1) nt = 1;
                pthread_t th[16];
                fxh_pread_job job[16];
                const size_t per = (want + (size_t)nt - 1) / (size_t)nt;
                for (int i = 0; i < nt; ++i) {
                    const size_t o = (size_t)i * per;
                    job[i].fd = pf->fd; job[i].dst = buf + gap + o; job[i].off = pf->offset + (off_t)o;
                    job[i].n = o >= want ? 0 : (want - o < per ? want - o : per);
                }
                for (int i = 1; i < nt; ++i) if (pthread_create(&th[i], NULL, fxh_pread_main, &job[i]) != 0) err(1, "pthread_create");
                fxh_pread_main(&job[0]);
                for (int i = 1; i < nt; ++i) pthread_join(th[i], NULL);
                size_t nl = 0;
                for (int i = 0; i < nt; ++i) { got += job[i].got; nl += job[i].newlines; if (job[i].got < job[i].n) { eof = 1; break; } }   /* a short slice is the end of the file */
                pf->offset += (off_t)got;
                if (pf->limit > 0 && pf->offset >= pf->limit) eof = 1;
                newlines = nl;
            }
        } else {
            if (pf->fifo && !pf->fan) { pf->fan = fxh_fan_open(); if (!pf->fan) pf->fifo = 0; }
            if (pf->fan) {
                const long long k = fxh_fan_fill(pf->fan, pf->fd, buf + gap, cap - gap, &eof, &newlines);
                if (k < 0) { fxh_fan_close(pf->fan); pf->fan = NULL; pf->fifo = 0; newlines = (size_t)-1; }
                else got = (size_t)k;
            }
            while (!pf->fan && gap + got < cap) {
                ssize_t k = read(pf->fd, buf + gap + got, cap - gap - got);
                if (k < 0) { if (errno == EINTR) continue; err(1, "read failed"); }
                if (k == 0) { eof = 1; break; }
                got += (size_t)k;
            }
        }
        if (newlines == (size_t)-1) newlines = fxh_count_newlines(buf + gap, got);
        pthread_mutex_lock(&pf->mu);
        pf->filled = got; pf->eof = eof; pf->newlines = newlines; pf->state = 2;
        pthread_cond_broadcast(&pf->cv);
    }
    pthread_mutex_unlock(&pf->mu);
    fxh_fan_close(pf->fan);
    pf->fan = NULL;
    return NULL;
}

/* call once, before the thread starts: is the input a regular file whose position we can take over? */
static void fxh_prefetch_probe(fxh_prefetch *pf, int fd)
{
    struct stat sb;
    const off_t pos = lseek(fd, 0, SEEK_CUR);
    pf->regular = (pos >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) ? 1 : 0;
    pf->fifo = !pf->regular && fstat(fd, &sb) == 0 && S_ISFIFO(sb.st_mode);
    pf->fan = NULL;
    pf->offset = pos;
    pf->io_threads = fxh_io_threads();
    { const char *e = getenv("FXH_IO_SLICE_MB"); const long v = e ? atol(e) : 0; pf->io_slice = (size_t)(v >= 1 && v <= 1024 ? v : (g_parts_mode ? 2 : 4)) << 20; }
}

static void fxh_prefetch_request(fxh_prefetch *pf, char *buf, size_t cap)
{
    pthread_mutex_lock(&pf->mu);
    pf->buf = buf; pf->cap = cap; pf->state = 1;
    pthread_cond_broadcast(&pf->cv);
    pthread_mutex_unlock(&pf->mu);
}

/* Make the next block current: [unread tail of the old block | prefetched data]; hand the old buffer back to the thread. */
void fxh_next_block(fxh_prefetch *pf, struct fxh_reader *rd, char **spare)
{
    if (!pf->started) {                /* first block: synchronous, then start reading ahead */
        fxh_reader_fill(rd);
        if (!rd->eof) {
            pthread_mutex_init(&pf->mu, NULL); pthread_cond_init(&pf->cv, NULL);
            pf->fd = rd->fd; pf->state = 0; pf->started = 1;
            pf->gap = rd->cap / 4 < FXH_GAP_MAX ? rd->cap / 4 : FXH_GAP_MAX;
            fxh_prefetch_probe(pf, rd->fd);
            pf->limit = rd->limit;
            if (pthread_create(&pf->th, NULL, fxh_prefetch_main, pf) != 0) err(1, "pthread_create");
            *spare = (char *)malloc(rd->cap + 1);
            if (!*spare) err(1, "out of memory");
            fxh_prefetch_request(pf, *spare, rd->cap);
        }
        return;
    }
    if (rd->eof) return;               /* everything has been read already; only the tail remains in rd */
    pthread_mutex_lock(&pf->mu);
    while (pf->state != 2) pthread_cond_wait(&pf->cv, &pf->mu);
    pf->state = 0;
    char *nb = pf->buf; const size_t filled = pf->filled; const int eof = pf->eof;
    pthread_mutex_unlock(&pf->mu);
    const size_t tail = rd->end - rd->beg;
    if (tail > pf->gap) errx(1, "input record longer than %zu bytes", pf->gap);
    memcpy(nb + pf->gap - tail, rd->buf + rd->beg, tail);
    char *old = rd->buf;
    rd->buf = nb; rd->beg = pf->gap - tail; rd->end = pf->gap + filled; rd->eof = eof;
    *spare = old;
    if (!eof) fxh_prefetch_request(pf, old, rd->cap);
}

/* The same for the lanes loop, where the previous buffers may still be in use: the read-ahead for the FOLLOWING block goes
 * to `target` (a buffer no block in flight refers to). */
void fxh_next_block_ring(fxh_prefetch *pf, struct fxh_reader *rd, char *target, size_t *fresh_newlines)
{
    *fresh_newlines = (size_t)-1;      /* unknown: the caller counts */
    if (!pf->started) {                /* first block: synchronous, then start reading ahead */
        fxh_reader_fill(rd);
        if (!rd->eof) {
            pthread_mutex_init(&pf->mu, NULL); pthread_cond_init(&pf->cv, NULL);
            pf->fd = rd->fd; pf->state = 0; pf->started = 1;
            pf->gap = rd->cap / 4 < FXH_GAP_MAX ? rd->cap / 4 : FXH_GAP_MAX;
            fxh_prefetch_probe(pf, rd->fd);
            pf->limit = rd->limit;
            if (pthread_create(&pf->th, NULL, fxh_prefetch_main, pf) != 0) err(1, "pthread_create");
            fxh_prefetch_request(pf, target, rd->cap);
        }
        return;
    }
    if (rd->eof) return;               /* everything has been read already; only the tail remains in rd */
    pthread_mutex_lock(&pf->mu);
    while (pf->state != 2) pthread_cond_wait(&pf->cv, &pf->mu);
    pf->state = 0;
    char *nb = pf->buf; const size_t filled = pf->filled; const int eof = pf->eof;
    *fresh_newlines = pf->newlines;
    pthread_mutex_unlock(&pf->mu);
    const size_t tail = rd->end - rd->beg;
    if (tail > pf->gap) errx(1, "input record longer than %zu bytes", pf->gap);
    memcpy(nb + pf->gap - tail, rd->buf + rd->beg, tail);
    rd->buf = nb; rd->beg = pf->gap - tail; rd->end = pf->gap + filled; rd->eof = eof;
    if (!eof) fxh_prefetch_request(pf, target, rd->cap);
}

void fxh_prefetch_stop(fxh_prefetch *pf)
{
    if (!pf->started) return;
    pthread_mutex_lock(&pf->mu);
    while (pf->state == 1) pthread_cond_wait(&pf->cv, &pf->mu);
    pf->state = 3;
    pthread_cond_broadcast(&pf->cv);
    pthread_mutex_unlock(&pf->mu);
    pthread_join(pf->th, NULL);
}

static void *fxh_awriter_main(void *arg)
{
    fxh_awriter *aw = (fxh_awriter *)arg;
    pthread_mutex_lock(&aw->mu);
    for (;;) {
        while (aw->state != 1 && aw->state != 3) pthread_cond_wait(&aw->cv, &aw->mu);
        if (aw->state == 3) break;
        const char *b = aw->buf; size_t n = aw->len;
        pthread_mutex_unlock(&aw->mu);
        fxh_writer_emit(aw->w, b, n);       /* raw write, or parallel gzip members with -z */
        pthread_mutex_lock(&aw->mu);
        aw->state = 0;
        pthread_cond_broadcast(&aw->cv);
    }
    pthread_mutex_unlock(&aw->mu);
    return NULL;
}

void fxh_awriter_wait(fxh_awriter *aw)
{
    if (!aw->started) return;
    pthread_mutex_lock(&aw->mu);
    while (aw->state == 1) pthread_cond_wait(&aw->cv, &aw->mu);
    pthread_mutex_unlock(&aw->mu);
}

/* hand the writer's filled buffer to the thread and continue formatting into the other one */
void fxh_awriter_submit(fxh_awriter *aw, struct fxh_writer *w, char **spare, size_t *spare_cap)
{
    if (!aw->started) {
        pthread_mutex_init(&aw->mu, NULL); pthread_cond_init(&aw->cv, NULL);
        aw->w = w; aw->state = 0; aw->started = 1;
        if (pthread_create(&aw->th, NULL, fxh_awriter_main, aw) != 0) err(1, "pthread_create");
    }
    fxh_awriter_wait(aw);              /* the other buffer is free again */
    if (!*spare) { *spare_cap = w->cap; *spare = (char *)malloc(*spare_cap); if (!*spare) err(1, "out of memory"); }
    pthread_mutex_lock(&aw->mu);
    aw->buf = w->buf; aw->len = w->len; aw->state = 1;
    pthread_cond_broadcast(&aw->cv);
    pthread_mutex_unlock(&aw->mu);
    char *t = w->buf; size_t tc = w->cap;
    w->buf = *spare; w->cap = *spare_cap; w->len = 0;
    *spare = t; *spare_cap = tc;
}

void fxh_awriter_stop(fxh_awriter *aw)
{
    if (!aw->started) return;
    fxh_awriter_wait(aw);
    pthread_mutex_lock(&aw->mu);
    aw->state = 3;
    pthread_cond_broadcast(&aw->cv);
    pthread_mutex_unlock(&aw->mu);
    pthread_join(aw->th, NULL);
}

/* hand a buffer owned by somebody else (a lane's output block) to the writer thread; it must stay untouched until a
 * LATER submit / wait has returned */
void fxh_awriter_submit_ext(fxh_awriter *aw, struct fxh_writer *w, const char *buf, size_t len)
{
    if (!aw->started) {
        pthread_mutex_init(&aw->mu, NULL); pthread_cond_init(&aw->cv, NULL);
        aw->w = w; aw->state = 0; aw->started = 1;
        if (pthread_create(&aw->th, NULL, fxh_awriter_main, aw) != 0) err(1, "pthread_create");
    }
    fxh_awriter_wait(aw);
    pthread_mutex_lock(&aw->mu);
    aw->buf = buf; aw->len = len; aw->state = 1;
    pthread_cond_broadcast(&aw->cv);
    pthread_mutex_unlock(&aw->mu);
}

