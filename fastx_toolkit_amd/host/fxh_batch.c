/* fxh_batch.c -- see fxh_batch.h. */
#define _GNU_SOURCE
#include "fxh_batch.h"

#include <err.h>
#include <fcntl.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <sys/prctl.h>
#include <sched.h>
#include <signal.h>
#include <unistd.h>

#include "fxh_internal.h"

static double fxh_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    const char *name, *seq, *name2, *qual;
    uint32_t name_len, seq_len, name2_len, qual_len;
    uint32_t reads_count;
    uint8_t is_ascii;
} fxh_rec;

typedef struct {
    fxg_ctx *ctx;
    /* host side (pinned) */
    uint8_t *h_bases, *h_qual;
    uint16_t *h_len;
    uint32_t *h_res;
    uint8_t *h_out_bases, *h_out_qual;
    size_t h_cap_bytes, h_cap_reads;
    /* device side */
    uint8_t *d_bases, *d_qual, *d_out_bases, *d_out_qual;
    uint16_t *d_len;
    uint32_t *d_res;
    uint64_t *d_counters;
    size_t d_cap_bytes, d_cap_reads;
    /* device text path (8f-1) */
    uint8_t *d_text, *d_out_text, *d_flags;
    uint32_t *d_ls;                        /* line starts [d_ls_cap] then line ends [d_ls_cap] */
    uint16_t *d_len16;
    uint64_t *d_out_off;
    size_t d_text_cap, d_ls_cap, d_off_cap;
    const void *registered[4];
} fxh_state;

#define FXG_CHECK(st, call)                                                                     \
    do {                                                                                        \
        int rc__ = (call);                                                                      \
        if (rc__ != 0) errx(1, "GPU engine error %d: %s", rc__, fxg_last_error((st)->ctx));   \
    } while (0)

void fxh_default_params(fxg_params *p, int qoffset)
{
    memset(p, 0, sizeof *p);
    p->qoffset = qoffset;
    strcpy(p->adapter, "CCTTAAGG");   /* fastx_clipper.cpp:68 */
    p->clip_min_len = 5;              /* fastx_clipper.cpp:69 */
    p->ft_first = 1;
    p->mask_min_quality = 10;         /* fastq_masker.c:47 */
    p->mask_char = 'N';               /* fastq_masker.c:48 */
}

static void fxh_grow_device(fxh_state *st, size_t reads, size_t bytes, int revcomp)
{
    if (reads > st->d_cap_reads || bytes > st->d_cap_bytes) {
        if (st->d_bases) {
            fxg_free_device(st->ctx, st->d_bases); fxg_free_device(st->ctx, st->d_qual); fxg_free_device(st->ctx, st->d_len);
            fxg_free_device(st->ctx, st->d_res);
            if (st->d_out_bases) { fxg_free_device(st->ctx, st->d_out_bases); fxg_free_device(st->ctx, st->d_out_qual); }
        }
        st->d_cap_reads = reads + reads / 4 + 1024;
        st->d_cap_bytes = bytes + bytes / 4 + 4096;
        FXG_CHECK(st, fxg_malloc_device(st->ctx, st->d_cap_bytes, (void **)&st->d_bases));
        FXG_CHECK(st, fxg_malloc_device(st->ctx, st->d_cap_bytes, (void **)&st->d_qual));
        FXG_CHECK(st, fxg_malloc_device(st->ctx, st->d_cap_reads * sizeof(uint16_t), (void **)&st->d_len));
        FXG_CHECK(st, fxg_malloc_device(st->ctx, st->d_cap_reads * sizeof(uint32_t), (void **)&st->d_res));
        st->d_out_bases = st->d_out_qual = NULL;
        if (revcomp) {
            FXG_CHECK(st, fxg_malloc_device(st->ctx, st->d_cap_bytes + 16, (void **)&st->d_out_bases));
            FXG_CHECK(st, fxg_malloc_device(st->ctx, st->d_cap_bytes + 16, (void **)&st->d_out_qual));
        }
    }
}

/* host-parser path: pinned staging rows as well as the device rows */
static void fxh_grow(fxh_state *st, size_t reads, size_t bytes, int revcomp)
{
    if (reads > st->h_cap_reads || bytes > st->h_cap_bytes) {
        if (st->h_bases) {
            fxg_free_host(st->ctx, st->h_bases); fxg_free_host(st->ctx, st->h_qual); fxg_free_host(st->ctx, st->h_len);
            fxg_free_host(st->ctx, st->h_res);
            if (st->h_out_bases) { fxg_free_host(st->ctx, st->h_out_bases); fxg_free_host(st->ctx, st->h_out_qual); }
        }
        st->h_cap_reads = reads + reads / 4 + 1024;
        st->h_cap_bytes = bytes + bytes / 4 + 4096;
        FXG_CHECK(st, fxg_malloc_host(st->ctx, st->h_cap_bytes, (void **)&st->h_bases));
        FXG_CHECK(st, fxg_malloc_host(st->ctx, st->h_cap_bytes, (void **)&st->h_qual));
        FXG_CHECK(st, fxg_malloc_host(st->ctx, st->h_cap_reads * sizeof(uint16_t), (void **)&st->h_len));
        FXG_CHECK(st, fxg_malloc_host(st->ctx, st->h_cap_reads * sizeof(uint32_t), (void **)&st->h_res));
        st->h_out_bases = st->h_out_qual = NULL;
        if (revcomp) {
            FXG_CHECK(st, fxg_malloc_host(st->ctx, st->h_cap_bytes + 16, (void **)&st->h_out_bases));
            FXG_CHECK(st, fxg_malloc_host(st->ctx, st->h_cap_bytes + 16, (void **)&st->h_out_qual));
        }
    }
    fxh_grow_device(st, reads, bytes, revcomp);
}

/* Size (dst == NULL) or write one kept record; seq/qual point at `len` output bytes; qual bytes are either raw input
 * characters (raw_qual) or Phred+33 codes (engine output / numeric input).  Returns the number of bytes. */
static int g_rename_ids = 0;
void fxh_set_rename_ids(int on) { g_rename_ids = on; }

static size_t fxh_emit(const FASTX *fx, const fxh_rec *r, const uint8_t *seq, const uint8_t *qual, size_t len, int raw_qual, char *d, size_t out_index)
{
    size_t k = 0;
#define PUTC(ch) do { if (d) d[k] = (char)(ch); k++; } while (0)
#define PUTS(ptr, n_) do { if (d) memcpy(d + k, (ptr), (n_)); k += (n_); } while (0)
    PUTC(fx->output_sequence_id_prefix);
    if (g_rename_ids) {                      /* the record's 1-based position in the output replaces its name */
        char num[24];
        int nd = 0;
        size_t v = out_index;
        do { num[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (nd) { --nd; PUTC(num[nd]); }      /* no side effects inside PUTC's argument: it is not evaluated when sizing */
    } else PUTS(r->name, r->name_len);
    PUTC('\n');
    PUTS(seq, len); PUTC('\n');
    if (fx->write_fastq) {
        const int ascii = fx->copy_input_fastq_format_to_output ? r->is_ascii : fx->write_fastq_ascii;   /* R6 */
        PUTC('+');
        PUTS(r->name2, r->name2_len); PUTC('\n');
        if (ascii) {
            if (raw_qual) PUTS(qual, len);                                              /* R8: q + Q is the input byte */
            else { const int sh = fx->fastq_ascii_quality_offset - 33; for (size_t i = 0; i < len; ++i) PUTC(qual[i] + sh); }
        } else {
            for (size_t i = 0; i < len; ++i) {
                int v = raw_qual ? (int)(signed char)qual[i] - fx->fastq_ascii_quality_offset : (int)qual[i] - 33;
                if (i) PUTC(' ');
                if (v < 0) { PUTC('-'); v = -v; }
                if (v >= 10) PUTC('0' + v / 10);
                PUTC('0' + v % 10);
            }
        }
        PUTC('\n');
    }
#undef PUTC
#undef PUTS
    return k;
}

/* ---------------------------------------------------------------------------------------------- */
/* worker threads: every host phase (index+validate, pack, format) is split by record range         */
/* ---------------------------------------------------------------------------------------------- */
#include <pthread.h>

typedef struct fxh_job fxh_job;
typedef struct fxh_worker {
    int id;
    fxh_job *job;
    FASTX *shadow;                 /* private parser state; reads the shared buffer through `view` */
    struct fxh_reader view;
    struct fxh_rawrec raw;
    size_t a0, a1, nl_count, first_nl;     /* newline census of the raw byte range [a0, a1) */
    size_t start;                          /* first record boundary at or after a0 */
    unsigned long long start_line;         /* lines before `start` (absolute input line numbering) */
    fxh_rec *rec;
    size_t nrec, rec_cap, maxlen, minlen;
    int rc_end;                            /* why indexing stopped: 0 end of input, -1 range/buffer end, -2 error */
    size_t end_pos;
    unsigned long long end_line;
    char errmsg[768];
    long bad_q;                            /* local index of the first record with an invalid quality line, or -1 */
    size_t rec0, use;                      /* global index of rec[0]; how many of this worker's records are in the batch */
    size_t out_bytes, out_off, kept_bytes, kept_off;
    size_t kept_count, kept_base;          /* kept records in this range; output index of its first kept record (1-based) */
    fxh_totals tot;
} fxh_worker;

struct fxh_job {
    FASTX *fx;
    fxh_state *st;
    const fxg_params *p;
    int nworkers, has_q, revcomp, lpr;
    uint32_t stride, fwd_start;
    char *out_dst;
    void (*phase)(fxh_worker *);
    fxh_worker *w;
};

static void *fxh_thread_main(void *arg) { fxh_worker *w = (fxh_worker *)arg; w->job->phase(w); return NULL; }

static void fxh_parallel(fxh_job *job, void (*phase)(fxh_worker *))
{
    pthread_t th[64];
    job->phase = phase;
    for (int i = 1; i < job->nworkers; ++i)
        if (pthread_create(&th[i], NULL, fxh_thread_main, &job->w[i]) != 0) err(1, "pthread_create");
    phase(&job->w[0]);
    for (int i = 1; i < job->nworkers; ++i) pthread_join(th[i], NULL);
}

static void fxh_phase_census(fxh_worker *w)
{
    const struct fxh_reader *rd = w->job->fx->reader;
    size_t n = 0, first = (size_t)-1, i = w->a0;
    while (i < w->a1) {
        const char *q = (const char *)memchr(rd->buf + i, '\n', w->a1 - i);
        if (!q) break;
        if (first == (size_t)-1) first = (size_t)(q - rd->buf);
        n++;
        i = (size_t)(q - rd->buf) + 1;
    }
    w->nl_count = n; w->first_nl = first;
}

static void fxh_phase_index(fxh_worker *w)
{
    FASTX *sh = w->shadow;
    w->nrec = 0; w->maxlen = 0; w->minlen = (size_t)-1; w->rc_end = -1;
    sh->reader = &w->view;
    sh->input_line_number = w->start_line;
    if (w->view.beg >= w->view.end && !w->view.eof) { w->end_pos = w->view.beg; w->end_line = sh->input_line_number; return; }
    for (;;) {
        int rc = fxh_next_raw(sh, &w->raw, 0);
        if (rc != 1) {
            w->rc_end = rc;
            if (rc == -2) memcpy(w->errmsg, w->raw.errmsg, sizeof w->errmsg);
            break;
        }
        if (w->nrec == w->rec_cap) {
            w->rec_cap = w->rec_cap ? w->rec_cap * 2 : (1u << 14);
            w->rec = (fxh_rec *)realloc(w->rec, w->rec_cap * sizeof(fxh_rec));
            if (!w->rec) err(1, "out of memory");
        }
        fxh_rec *r = &w->rec[w->nrec++];
        r->name = w->raw.name; r->seq = w->raw.seq; r->name2 = w->raw.name2; r->qual = w->raw.qual;
        r->name_len = (uint32_t)w->raw.name_len; r->seq_len = (uint32_t)w->raw.seq_len;
        r->name2_len = (uint32_t)w->raw.name2_len; r->qual_len = (uint32_t)w->raw.qual_len;
        r->is_ascii = (uint8_t)w->raw.is_ascii;
        r->reads_count = (uint32_t)fxh_reads_count(sh, w->raw.name, w->raw.name_len);
        if (w->raw.seq_len > w->maxlen) w->maxlen = w->raw.seq_len;
        if (w->raw.seq_len < w->minlen) w->minlen = w->raw.seq_len;
    }
    w->end_pos = w->view.beg;
    w->end_line = sh->input_line_number;
}

static void fxh_phase_pack(fxh_worker *w)
{
    fxh_job *job = w->job;
    fxh_state *st = job->st;
    FASTX *sh = w->shadow;
    const uint32_t stride = job->stride;
    const int lpr = job->lpr;
    w->bad_q = -1;
    for (size_t k = 0; k < w->use; ++k) {
        const fxh_rec *r = &w->rec[k];
        const size_t i = w->rec0 + k;
        memcpy(st->h_bases + i * stride, r->seq, r->seq_len);
        st->h_len[i] = (uint16_t)r->seq_len;
        if (!job->has_q) continue;
        uint8_t *q = st->h_qual + i * stride;
        if (r->is_ascii && sh->fastq_ascii_quality_offset == 33) {
            uint8_t bad = 0;
            for (uint32_t j = 0; j < r->qual_len; ++j) { const uint8_t c = (uint8_t)r->qual[j]; bad |= (uint8_t)((c < 18) | (c > 126)); }
            if (!bad) { memcpy(q, r->qual, r->qual_len); continue; }
        }
        w->raw.seq_len = r->seq_len; w->raw.qual = r->qual; w->raw.qual_len = r->qual_len; w->raw.is_ascii = r->is_ascii;
        sh->input_line_number = w->start_line + (unsigned long long)lpr * (k + 1);   /* the quality line of this record */
        if (fxh_decode_quality(sh, &w->raw, NULL, q) != 0) {   /* first bad record of this range */
            w->bad_q = (long)k;
            memcpy(w->errmsg, w->raw.errmsg, sizeof w->errmsg);
            return;
        }
    }
}

static void fxh_phase_count(fxh_worker *w)
{
    const fxh_state *st = w->job->st;
    size_t kept = 0;
    for (size_t k = 0; k < w->use; ++k) kept += FXG_RES_KEEP(st->h_res[w->rec0 + k]);
    w->kept_count = kept;
}

/* bytes this worker will write and the report totals of its records */
static void fxh_phase_size(fxh_worker *w)
{
    fxh_job *job = w->job;
    fxh_state *st = job->st;
    const FASTX *fx = job->fx;
    memset(&w->tot, 0, sizeof w->tot);
    size_t bytes = 0, kept_bytes = 0, oidx = w->kept_base;
    for (size_t k = 0; k < w->use; ++k) {
        const fxh_rec *r = &w->rec[k];
        const size_t i = w->rec0 + k;
        const uint32_t x = st->h_res[i], len = FXG_RES_LEN(x), rc_ = r->reads_count;
        fxh_totals *t = &w->tot;
        t->input_sequences++; t->input_reads += rc_;
        t->clip_input += rc_;
        if (FXG_RES_ADAPTER_ONLY(x)) t->clip_adapter_only += rc_;
        switch (FXG_RES_REASON(x)) {
        case FXG_R_CLIP_TOO_SHORT: t->clip_too_short += rc_; break;
        case FXG_R_CLIP_NO_ADAPTER: t->clip_no_adapter += rc_; break;
        case FXG_R_CLIP_ADAPTER_FOUND: t->clip_adapter_found += rc_; break;
        case FXG_R_CLIP_N: t->clip_n += rc_; break;
        case FXG_R_QTRIM: t->qtrim_dropped += rc_; break;
        default: break;
        }
        if (!FXG_RES_KEEP(x)) continue;
        t->output_sequences++; t->output_reads += rc_;
        kept_bytes += len;
        /* only numeric-quality output depends on the values (digit counts); reversal does not change their multiset */
        const uint8_t *qv = st->h_qual ? st->h_qual + i * job->stride + (job->revcomp ? r->seq_len - job->fwd_start - len : job->fwd_start) : NULL;
        bytes += fxh_emit(fx, r, (const uint8_t *)r->seq, r->is_ascii ? (const uint8_t *)r->seq : qv, len, 0, NULL, oidx++);
    }
    w->out_bytes = bytes; w->kept_bytes = kept_bytes;
}

static void fxh_phase_format(fxh_worker *w)
{
    fxh_job *job = w->job;
    fxh_state *st = job->st;
    const FASTX *fx = job->fx;
    char *d = job->out_dst + w->out_off;
    size_t k2 = 0, opos = w->kept_off, oidx = w->kept_base;
    for (size_t k = 0; k < w->use; ++k) {
        const fxh_rec *r = &w->rec[k];
        const size_t i = w->rec0 + k;
        const uint32_t x = st->h_res[i], len = FXG_RES_LEN(x);
        if (!FXG_RES_KEEP(x)) continue;
        if (job->revcomp) {
            k2 += fxh_emit(fx, r, st->h_out_bases + opos, st->h_out_qual ? st->h_out_qual + opos : NULL, len, 0, d + k2, oidx++);
            opos += len;
        } else if (r->is_ascii) {
            k2 += fxh_emit(fx, r, (const uint8_t *)r->seq + job->fwd_start, (const uint8_t *)r->qual + job->fwd_start, len, 1, d + k2, oidx++);
        } else {
            k2 += fxh_emit(fx, r, (const uint8_t *)r->seq + job->fwd_start, st->h_qual + i * job->stride + job->fwd_start, len, 0, d + k2, oidx++);
        }
    }
    w->out_bytes = k2;
}

/* ---------------------------------------------------------------------------------------------- */
/* I/O overlap: one thread reads the next block while the current one is processed, another one     */
/* writes the previous output while the next is being formatted                                     */
/* ---------------------------------------------------------------------------------------------- */
#include <errno.h>
#define FXH_GAP_MAX ((size_t)1 << 20)  /* room in front of a prefetched block for the previous block's unread tail */

typedef struct {
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int fd, started;
    /* request / response, protected by mu */
    size_t gap;                        /* min(1 MB, cap / 4) */
    char *buf; size_t cap;             /* buffer to fill: data goes to buf[gap, cap) */
    size_t filled; int eof;
    size_t newlines;                   /* '\n' bytes among the `filled` bytes */
    int state;                         /* 0 idle, 1 requested, 2 done, 3 quit */
    int regular, io_threads;           /* regular file: parallel pread() from `offset` on */
    size_t io_slice;                   /* smallest piece worth a thread of its own */
    off_t offset, limit;               /* limit > 0: the input ends at this file offset (a part of a sharded run) */
} fxh_prefetch;

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

static int g_parts_mode;               /* a sharded run is under way (fxh_run_parts): smaller blocks and fewer helper threads per part */

static int fxh_io_threads(void)
{
    const char *e = getenv("FXH_IO_THREADS");
    long n = e ? atol(e) : (g_parts_mode ? 4 : 8), ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    if (n < 1) n = 1;
    if (n > 16) n = 16;
    if (ncpu > 0 && n > ncpu) n = ncpu;
    return (int)n;
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
            while (gap + got < cap) {
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
    return NULL;
}

/* call once, before the thread starts: is the input a regular file whose position we can take over? */
static void fxh_prefetch_probe(fxh_prefetch *pf, int fd)
{
    struct stat sb;
    const off_t pos = lseek(fd, 0, SEEK_CUR);
    pf->regular = (pos >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) ? 1 : 0;
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
static void fxh_next_block(fxh_prefetch *pf, struct fxh_reader *rd, char **spare)
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
static void fxh_next_block_ring(fxh_prefetch *pf, struct fxh_reader *rd, char *target, size_t *fresh_newlines)
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

static void fxh_prefetch_stop(fxh_prefetch *pf)
{
    if (!pf->started) return;
    pthread_mutex_lock(&pf->mu);
    while (pf->state == 1) pthread_cond_wait(&pf->cv, &pf->mu);
    pf->state = 3;
    pthread_cond_broadcast(&pf->cv);
    pthread_mutex_unlock(&pf->mu);
    pthread_join(pf->th, NULL);
}

typedef struct {
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    struct fxh_writer *w;
    int started;
    const char *buf; size_t len;
    int state;                         /* 0 idle, 1 pending, 3 quit */
} fxh_awriter;

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

static void fxh_awriter_wait(fxh_awriter *aw)
{
    if (!aw->started) return;
    pthread_mutex_lock(&aw->mu);
    while (aw->state == 1) pthread_cond_wait(&aw->cv, &aw->mu);
    pthread_mutex_unlock(&aw->mu);
}

/* hand the writer's filled buffer to the thread and continue formatting into the other one */
static void fxh_awriter_submit(fxh_awriter *aw, struct fxh_writer *w, char **spare, size_t *spare_cap)
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

static void fxh_awriter_stop(fxh_awriter *aw)
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
static void fxh_awriter_submit_ext(fxh_awriter *aw, struct fxh_writer *w, const char *buf, size_t len)
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
#ifndef FXH_MAX_LANES
#define FXH_MAX_LANES 32
#endif
static int g_parts_abort;                  /* sharded run: some part met input it does not handle (fxh_run_parts); relaxed atomics, it is only a "stop soon" */
#define FXH_ABORT_SET() __atomic_store_n(&g_parts_abort, 1, __ATOMIC_RELAXED)
#define FXH_ABORTED()   __atomic_load_n(&g_parts_abort, __ATOMIC_RELAXED)
static pthread_mutex_t g_first_ctx_mu = PTHREAD_MUTEX_INITIALIZER;   /* the HIP runtime's first-use initialisation: one thread at a time */
static int g_first_ctx_done;
static int g_hip_touched;                   /* this process has initialised the HIP runtime (a context, or the device query of fxh_bind_near_device): never fork() after that */
struct fxh_pinned { pthread_mutex_t mu; const void *ptr[FXH_MAX_LANES + 4]; int n; };

#define FXH_MAX_LANES 32
typedef struct fxh_lane {
    int id, device;
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int state;                             /* 0 idle, 1 job posted, 2 done, 3 quit */
    int ready;                             /* the context exists (created by the lane's own thread, off the main thread's path) */
    struct fxh_lane *first;                /* lane 0: the others create their contexts after it (two threads inside the runtime's first-use initialisation take twice as long as one after the other) */
    fxh_state st;
    struct fxh_pinned *pinned;             /* input buffers already page-locked (shared by the lanes) */
    char *text_base; size_t text_cap;      /* the input buffer the job's text lives in */
    const fxg_params *p;                   /* configuration, read-only */
    int revcomp, qoffset;                  /* revcomp: the output comes from the engine's packed arrays (reverse-complement, masker) */
    int reverse, lpr, has_q, out_fasta;    /* the packed output is reversed; lines per record; qualities present; write FASTA */
    uint32_t fwd_start;
    const char *text; size_t len;          /* job: whole records, every line '\n'-terminated */
    uint64_t records;
    int clip_history;                      /* this lane is the one aligner of a fastx_clipper run */
    int clip_guard;                        /* clipper run in its parallel phase (fxh_run.clip_auto): a block whose reads are not all of one length is handed back untouched */
    uint32_t fixed_len;                    /* result: the one length of the block's reads, 0 = they differ (or the block was not indexed) */
    int slot;                              /* which of out[] receives the text (the other may still be with the writer) */
    int handled;                           /* result: 0 = irregular block, parse it on the host */
    char *out[2]; size_t out_cap[2]; size_t out_len;
    uint64_t ctr[FXG_NCOUNTERS];
    uint64_t weighted[8];                  /* FASTA: tallies weighted by the records' read counts (fxg_fasta_weights) */
    double t_busy, t_init;
    double t_call[8];                        /* FXH_TIMING: seconds inside h2d, index, pack, pipeline, counters, format, d2h+sync, blocks */
} fxh_lane;

static void fxh_lane_run(fxh_lane *ln)
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
    if (info.irregular || info.records == 0 || info.records != ln->records || info.consumed != len) return;
    const uint64_t n = info.records;
    const uint32_t stride = info.max_len;
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
    fxg_batch in = {st->d_bases, ln->has_q ? st->d_qual : NULL, fixed ? NULL : st->d_len16, stride, stride, n};
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
    if (lpr == 2) FXG_CHECK(st, fxg_fasta_weights(st->ctx, st->d_text, st->d_ls, st->d_ls_cap, n, st->d_res, ln->weighted));
    uint64_t out_bytes = 0;
    FXG_CHECK(st, fxg_fastq_format(st->ctx, st->d_text, lpr, st->d_ls, st->d_ls_cap, st->d_flags, n, st->d_res, ln->fwd_start, ln->reverse,
                                   revcomp ? st->d_out_bases : NULL, (revcomp && ln->has_q) ? st->d_out_qual : NULL, revcomp ? st->d_out_off : NULL,
                                   ln->has_q ? st->d_qual : NULL, stride, ln->qoffset, ln->out_fasta, st->d_out_text, &out_bytes));
    FXH_TCALL(5);
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

static void *fxh_lane_main(void *arg)
{
    fxh_lane *ln = (fxh_lane *)arg;
    if (ln->first && ln->first != ln) {
        pthread_mutex_lock(&ln->first->mu);
        while (!ln->first->ready) pthread_cond_wait(&ln->first->cv, &ln->first->mu);
        pthread_mutex_unlock(&ln->first->mu);
    }
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
static int fxh_bind_near_device(int device, cpu_set_t *before)
{
    if (getenv("FXH_NO_NUMA")) return 0;
    pthread_mutex_lock(&g_first_ctx_mu);         /* the query is a first use of the HIP runtime: one thread at a time, like the first context */
    const int node = fxg_device_numa_node(device);
    g_hip_touched = 1;                           /* (no fork() over an initialised runtime from here on, fxh_run_parts) */
    pthread_mutex_unlock(&g_first_ctx_mu);
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
static int fxh_device_list(int *dev, int cap)
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

/* fastx_quality_stats mode of the run loop: batches feed fxg_run_quality_stats instead of the pipeline, nothing is written */
typedef struct fxh_stats_run {
    uint64_t *d_hist;
    uint32_t cols;
} fxh_stats_run;

static void fxh_stats_reserve(fxh_state *st, fxh_stats_run *sr, uint32_t need_cols)
{
    if (need_cols <= sr->cols) return;
    uint32_t ncols = sr->cols ? sr->cols * 2 : 256;
    while (ncols < need_cols) ncols *= 2;
    const size_t per_col = (size_t)FXG_QS_CLASSES * FXG_QS_BINS * sizeof(uint64_t);
    uint64_t *nh = NULL;
    FXG_CHECK(st, fxg_malloc_device(st->ctx, (size_t)ncols * per_col, (void **)&nh));
    FXG_CHECK(st, fxg_memset_device(st->ctx, nh, 0, (size_t)ncols * per_col));
    if (sr->d_hist) {                       /* a later block has longer reads: carry the columns over (rare, via the host) */
        void *tmp = malloc((size_t)sr->cols * per_col);
        if (!tmp) err(1, "out of memory");
        FXG_CHECK(st, fxg_memcpy_d2h(st->ctx, tmp, sr->d_hist, (size_t)sr->cols * per_col));
        FXG_CHECK(st, fxg_sync(st->ctx));
        FXG_CHECK(st, fxg_memcpy_h2d(st->ctx, nh, tmp, (size_t)sr->cols * per_col));
        FXG_CHECK(st, fxg_sync(st->ctx));
        free(tmp);
        fxg_free_device(st->ctx, sr->d_hist);
    }
    sr->d_hist = nh; sr->cols = ncols;
}

/* everything one run of a tool shares between its blocks */
typedef struct fxh_run {
    FASTX *fx;
    const fxg_params *p;
    fxh_totals *tot;
    fxh_stats_run *stats;
    fxh_state st;                          /* the host-parser path's own context and buffers (created on first use) */
    int st_device, st_shared;              /* st_shared: the context belongs to lane 0 (serial clipper run) */
    /* fastx_clipper without being told anything: the reference's aligner carries its query buffer from read to read, but the
     * stale tail only exists once a read SHORTER than the longest so far turns up (sequence_alignment.cpp:135-136).  While every block so
     * far consists of reads of ONE length (clip_len, the first block's), blocks are independent: lanes and parts run in parallel without
     * history (clip_auto).  The first block that is different -- ragged, another length, or anything the device path hands back -- switches
     * the run to the reference's mode at that block: one lane, history on, seeded with the last record before it (clip_seed), which is
     * exactly the aligner's state after reads of one length (fxh_clip_go_serial). */
    int clip_auto;
    uint32_t clip_len;
    char *clip_seed; size_t clip_seed_len, clip_seed_cap;
    fxh_job job;
    fxh_awriter aw;
    char *wr_spare; size_t wr_spare_cap;
    int overlap;
    char errmsg[768];
    int have_err, at_eof;
    int part, nparts;                      /* sharded run (FXH_PARTS): this run is part `part` of `nparts`; irregular input aborts it (fxh_run_parts) */
    int aborted;
    struct fxh_pinned pinned;              /* input buffers the lanes have page-locked */
    unsigned long n_fallback;
    double t_index, t_pack, t_gpu, t_fmt, t_init;
    double t_wait_lane, t_wait_writer, t_drain;      /* lanes loop: main thread blocked on a lane / on the writer / final drain */
} fxh_run;

static void fxh_run_ctx(fxh_run *R)
{
    if (R->st.ctx) return;
    const double t0 = fxh_now();
    pthread_mutex_lock(&g_first_ctx_mu);
    g_hip_touched = 1;
    pthread_mutex_unlock(&g_first_ctx_mu);
    int rc = fxg_ctx_create(R->st_device, &R->st.ctx);
    if (rc != 0) errx(1, "no usable MI355X/HIP device (fxg_ctx_create = %d); this build has no CPU path", rc);
    FXG_CHECK(&R->st, fxg_malloc_device(R->st.ctx, FXG_NCOUNTERS * sizeof(uint64_t), (void **)&R->st.d_counters));
    /* one fastx_clipper process = one aligner whose query buffer survives from read to read (sequence_alignment.cpp:135-136,
     * SURVEY N3): the engine reproduces that across the batches of this run */
    if (R->p->stages & FXG_STAGE_CLIP) FXG_CHECK(&R->st, fxg_set_clip_history(R->st.ctx, 1));
    R->t_init += fxh_now() - t0;
}

/* what the four steps of the host-parser path hand to one another */
typedef struct { size_t beg, end, n, maxlen, minlen; int stop; uint64_t ctr[FXG_NCOUNTERS]; } fxh_hb;

/* 1. split [beg, end) into worker ranges at record boundaries, index and validate the records in parallel, merge in input order
 * (the first error / end condition wins) and, where one long read would blow the rows up, cut the batch at a record boundary.
 * Returns 0 when there is nothing to process (R->have_err / R->at_eof say why). */
static int fxh_host_index(fxh_run *R, fxh_hb *hb)
{
    FASTX *fx = R->fx;
    fxh_job *job = &R->job;
    struct fxh_reader *rd = fx->reader;
    const size_t beg = hb->beg, end = hb->end;
    const int T = job->nworkers;
    for (int i = 0; i < T; ++i) {
        job->w[i].a0 = beg + (size_t)((unsigned long long)(end - beg) * (unsigned)i / (unsigned)T);
        job->w[i].a1 = beg + (size_t)((unsigned long long)(end - beg) * (unsigned)(i + 1) / (unsigned)T);
    }
    fxh_parallel(job, fxh_phase_census);
    {
        /* worker i starts at the first record boundary at or after a0: lines are counted from the block start,
         * which is itself a record boundary (the previous block stopped at one) */
        unsigned long long lines_before = 0;          /* complete lines in [beg, a0) */
        for (int i = 0; i < T; ++i) {
            fxh_worker *w = &job->w[i];
            size_t s;
            unsigned long long ls;                    /* complete lines in [beg, s) */
            if (i == 0) { s = beg; ls = 0; }
            else if (w->first_nl == (size_t)-1) { s = (size_t)-1; ls = 0; }       /* no line starts here: same boundary as the next range */
            else {
                s = w->first_nl + 1; ls = lines_before + 1;
                while (ls % (unsigned)job->lpr != 0) {         /* walk to the next record boundary */
                    const char *q = s < end ? (const char *)memchr(rd->buf + s, '\n', end - s) : NULL;
                    if (!q) { s = end; break; }
                    s = (size_t)(q - rd->buf) + 1; ls++;
                }
            }
            if (s != (size_t)-1 && s > end) s = end;
            w->start = s;
            w->start_line = fx->input_line_number + ls;
            lines_before += w->nl_count;
        }
        for (int i = T - 1; i >= 0; --i)                      /* ranges without a line start are empty */
            if (job->w[i].start == (size_t)-1) job->w[i].start = (i + 1 < T) ? job->w[i + 1].start : end;
        for (int i = 0; i < T; ++i) {
            fxh_worker *w = &job->w[i];
            w->view.buf = rd->buf; w->view.cap = rd->cap; w->view.fd = -1;
            w->view.beg = w->start;
            w->view.end = (i + 1 < T) ? job->w[i + 1].start : end;
            w->view.eof = 0;
        }
        /* the last non-empty range owns the end-of-input / incomplete-tail semantics */
        for (int i = T - 1; i >= 0; --i)
            if (job->w[i].view.beg < end || i == 0) { job->w[i].view.end = end; job->w[i].view.eof = rd->eof; break; }
    }
    fxh_parallel(job, fxh_phase_index);
    /* merge in input order; the first error / end condition wins */
    size_t n = 0, maxlen = 0, minlen = (size_t)-1;
    int stop = -1;                                   /* worker at which the batch ends */
    for (int i = 0; i < T; ++i) {
        fxh_worker *w = &job->w[i];
        w->rec0 = n; w->use = w->nrec;
        n += w->nrec;
        if (w->nrec) { if (w->maxlen > maxlen) maxlen = w->maxlen; if (w->minlen < minlen) minlen = w->minlen; }
        const int is_last = (w->view.end == end);
        if (w->rc_end == -2) { R->have_err = 1; memcpy(R->errmsg, w->errmsg, sizeof R->errmsg); stop = i; }
        else if (w->rc_end == 0) { R->at_eof = 1; stop = i; }
        else if (is_last) stop = i;
        if (stop >= 0) { rd->beg = w->end_pos; fx->input_line_number = w->end_line; break; }
    }
    for (int i = stop + 1; i < T; ++i) { job->w[i].use = 0; job->w[i].rec0 = n; }
    if (n == 0) {
        if (!R->have_err && !R->at_eof) errx(1, "input record does not fit in the %zu MB read buffer", rd->cap >> 20);
        return 0;
    }
    /* One long read among short ones must not blow the rows up (rows are n * longest): cut the batch at a record boundary when
     * the SoA would exceed ~16x the text of the block; the rest of the block is the next call's business. */
    {
        const size_t budget = (size_t)16 * (end - beg) + ((size_t)64 << 20);
        if (n * maxlen > budget && n > 1) {
            size_t keep_n = 0, run_max = 0;
            int cut_w = -1; size_t cut_k = 0;
            for (int i = 0; i <= stop && cut_w < 0; ++i) {
                fxh_worker *w = &job->w[i];
                for (size_t k = 0; k < w->use; ++k) {
                    const size_t L = w->rec[k].seq_len;
                    const size_t m = L > run_max ? L : run_max;
                    if ((keep_n + 1) * m > budget && keep_n > 0) { cut_w = i; cut_k = k; break; }
                    run_max = m; keep_n++;
                }
            }
            if (cut_w >= 0) {
                fxh_worker *w = &job->w[cut_w];
                rd->beg = (size_t)((w->rec[cut_k].name - 1) - rd->buf);           /* the record's first byte ('@' / '>') */
                fx->input_line_number = w->start_line + (unsigned long long)job->lpr * cut_k;
                w->use = cut_k;
                for (int j = cut_w + 1; j < T; ++j) job->w[j].use = 0;
                n = keep_n; maxlen = run_max;
                minlen = (size_t)-1;
                for (int i = 0; i <= cut_w; ++i) for (size_t k = 0; k < job->w[i].use; ++k) if (job->w[i].rec[k].seq_len < minlen) minlen = job->w[i].rec[k].seq_len;
                stop = cut_w;
                R->have_err = 0; R->at_eof = 0;          /* whatever ended the block lies beyond the cut */
            }
        }
    }
    hb->n = n; hb->maxlen = maxlen; hb->minlen = minlen; hb->stop = stop;
    return 1;
}

/* 2. pack the SoA rows (qualities normalised to Phred+33 codes); a bad base or quality ends the batch in front of its record */
static void fxh_host_pack(fxh_run *R, fxh_hb *hb)
{
    fxh_job *job = &R->job;
    const int T = job->nworkers, stop = hb->stop;
    size_t n = hb->n;
    job->stride = (uint32_t)hb->maxlen;
    fxh_grow(&R->st, n, n * (size_t)job->stride, job->revcomp);
    fxh_parallel(job, fxh_phase_pack);
    for (int i = 0; i <= stop; ++i) {
        fxh_worker *w = &job->w[i];
        if (w->bad_q >= 0) {                       /* first bad record wins: drop it and everything after it */
            R->have_err = 1; R->at_eof = 0;
            memcpy(R->errmsg, w->errmsg, sizeof R->errmsg);
            n = w->rec0 + (size_t)w->bad_q;
            w->use = (size_t)w->bad_q;
            for (int j = i + 1; j < T; ++j) job->w[j].use = 0;
            break;
        }
    }
    hb->n = n;
}

/* 3. upload, one engine call, download res[] (and the packed arrays of reverse-complemented / masked output).  Returns 0 when the
 * batch fed fastx_quality_stats (nothing to write). */
static int fxh_host_engine(fxh_run *R, fxh_hb *hb)
{
    FASTX *fx = R->fx;
    fxh_job *job = &R->job;
    fxh_state *st = &R->st;
    fxh_totals *tot = R->tot;
    const fxg_params *p = R->p;
    const size_t n = hb->n, maxlen = hb->maxlen, minlen = hb->minlen;
    uint64_t *ctr = hb->ctr;
    const uint32_t stride = job->stride;
    const size_t bytes = n * (size_t)stride;
    const int fixed = (minlen == maxlen);
    FXG_CHECK(st, fxg_memcpy_h2d(st->ctx, st->d_bases, st->h_bases, bytes));
    if (job->has_q) FXG_CHECK(st, fxg_memcpy_h2d(st->ctx, st->d_qual, st->h_qual, bytes));
    if (!fixed) FXG_CHECK(st, fxg_memcpy_h2d(st->ctx, st->d_len, st->h_len, n * sizeof(uint16_t)));
    fxg_batch in = {st->d_bases, job->has_q ? st->d_qual : NULL, fixed ? NULL : st->d_len, (uint32_t)maxlen, stride, n};
    if (R->stats) {                            /* fastx_quality_stats: reduce, nothing to write */
        fxh_stats_reserve(st, R->stats, stride);
        FXG_CHECK(st, fxg_run_quality_stats(st->ctx, &in, R->stats->d_hist, R->stats->cols));
        FXG_CHECK(st, fxg_sync(st->ctx));
        tot->input_sequences += n; tot->input_reads += n;
        fx->num_input_sequences = tot->input_sequences; fx->num_input_reads = tot->input_reads;
        return 0;
    }
    fxg_out out = {st->d_res, job->revcomp ? st->d_out_bases : NULL, (job->revcomp && job->has_q) ? st->d_out_qual : NULL, NULL, NULL, NULL, st->d_counters};
    fxg_params pp = *p;
    pp.qoffset = 33;                        /* rows hold Phred+33 codes whatever -Q was */
    FXG_CHECK(st, fxg_run_pipeline(st->ctx, &in, &pp, &out));
    FXG_CHECK(st, fxg_memcpy_d2h(st->ctx, st->h_res, st->d_res, n * sizeof(uint32_t)));
    {
        int rc = fxg_read_counters(st->ctx, st->d_counters, ctr);   /* synchronises */
        if (rc == FXG_E_DEVICE && (ctr[FXG_C_ERRORS] & FXG_DEV_ERR_BAD_BASE)) errx(1, "%s", fxg_last_error(st->ctx));
        if (rc != 0) errx(1, "GPU engine error %d: %s", rc, fxg_last_error(st->ctx));
    }
    tot->masked_reads += ctr[FXG_C_MASKED_READS]; tot->masked_nucleotides += ctr[FXG_C_MASKED_NT];
    if (job->revcomp) {
        FXG_CHECK(st, fxg_memcpy_d2h(st->ctx, st->h_out_bases, st->d_out_bases, ctr[FXG_C_KEPT_BASES]));
        if (job->has_q) FXG_CHECK(st, fxg_memcpy_d2h(st->ctx, st->h_out_qual, st->d_out_qual, ctr[FXG_C_KEPT_BASES]));
        FXG_CHECK(st, fxg_sync(st->ctx));
    }
    return 1;
}

/* 4. format the kept records in input order (each worker its own slice), tally the report counters, hand the text to the writer */
static void fxh_host_format(fxh_run *R)
{
    FASTX *fx = R->fx;
    fxh_job *job = &R->job;
    fxh_totals *tot = R->tot;
    const int T = job->nworkers;
    fxh_parallel(job, fxh_phase_count);
    {
        size_t base = tot->output_sequences + 1;
        for (int i = 0; i < T; ++i) { job->w[i].kept_base = base; base += job->w[i].kept_count; }
    }
    fxh_parallel(job, fxh_phase_size);
    size_t total = 0, kept_total = 0;
    for (int i = 0; i < T; ++i) {
        fxh_worker *w = &job->w[i];
        w->out_off = total; w->kept_off = kept_total;
        total += w->out_bytes; kept_total += w->kept_bytes;
        tot->input_sequences += w->tot.input_sequences; tot->input_reads += w->tot.input_reads;
        tot->output_sequences += w->tot.output_sequences; tot->output_reads += w->tot.output_reads;
        tot->clip_input += w->tot.clip_input; tot->clip_too_short += w->tot.clip_too_short;
        tot->clip_adapter_only += w->tot.clip_adapter_only; tot->clip_no_adapter += w->tot.clip_no_adapter;
        tot->clip_adapter_found += w->tot.clip_adapter_found; tot->clip_n += w->tot.clip_n;
        tot->qtrim_dropped += w->tot.qtrim_dropped;
    }
    struct fxh_writer *wr = fx->writer;
    job->out_dst = fxh_writer_reserve(wr, total + 16);
    fxh_parallel(job, fxh_phase_format);
    wr->len += total;
    if (R->overlap) fxh_awriter_submit(&R->aw, wr, &R->wr_spare, &R->wr_spare_cap); else fxh_writer_flush(wr);
    fx->num_input_sequences = tot->input_sequences; fx->num_input_reads = tot->input_reads;
    fx->num_output_sequences = tot->output_sequences; fx->num_output_reads = tot->output_reads;
}


/* Host-parser path for the unread part [rd->beg, rd->end) of the current block: index + validate in parallel, pack the SoA rows,
 * run the engine, format the kept records.  Consumes whole records (rd->beg moves on), sets have_err / at_eof. */
static void fxh_host_block(fxh_run *R)
{
    struct fxh_reader *rd = R->fx->reader;
    fxh_hb hb;
    memset(&hb, 0, sizeof hb);
    double t0 = fxh_now();
    fxh_run_ctx(R);
    hb.beg = rd->beg; hb.end = rd->end;
    const int have = fxh_host_index(R, &hb);
    R->t_index += fxh_now() - t0; t0 = fxh_now();
    if (!have) return;
    fxh_host_pack(R, &hb);
    R->t_pack += fxh_now() - t0; t0 = fxh_now();
    if (hb.n == 0) return;
    const int to_write = fxh_host_engine(R, &hb);
    R->t_gpu += fxh_now() - t0; t0 = fxh_now();
    if (!to_write) return;
    fxh_host_format(R);
    R->t_fmt += fxh_now() - t0;
}

static void fxh_add_counters(fxh_totals *tot, const uint64_t *ctr, uint64_t n, const uint64_t *weighted)
{
    tot->input_sequences += n;
    tot->output_sequences += ctr[FXG_C_KEPT];
    if (weighted) {                        /* FASTA: a record ">id-count" stands for `count` reads (fastx.c:475-495) */
        tot->input_reads += weighted[0]; tot->output_reads += weighted[1];
        tot->clip_input += (unsigned)weighted[0];
        tot->clip_too_short += (unsigned)weighted[2]; tot->clip_adapter_only += (unsigned)weighted[3];
        tot->clip_no_adapter += (unsigned)weighted[4]; tot->clip_adapter_found += (unsigned)weighted[5];
        tot->clip_n += (unsigned)weighted[6];
    } else {                               /* FASTQ ids are never collapsed: every record counts as one read (fastx.c:480-481) */
        tot->input_reads += n; tot->output_reads += ctr[FXG_C_KEPT];
        tot->clip_input += (unsigned)n;
        tot->clip_too_short += (unsigned)ctr[FXG_C_CLIP_TOO_SHORT]; tot->clip_adapter_only += (unsigned)ctr[FXG_C_CLIP_ADAPTER_ONLY];
        tot->clip_no_adapter += (unsigned)ctr[FXG_C_CLIP_NO_ADAPTER]; tot->clip_adapter_found += (unsigned)ctr[FXG_C_CLIP_ADAPTER_FOUND];
        tot->clip_n += (unsigned)ctr[FXG_C_CLIP_N];
    }
    tot->masked_reads += ctr[FXG_C_MASKED_READS]; tot->masked_nucleotides += ctr[FXG_C_MASKED_NT];
    tot->qtrim_dropped += ctr[FXG_C_QTRIM_DROPPED];
}

/* the blocks in flight: where their text lives and what the host parser needs if a lane hands one back */
typedef struct fxh_block {
    char *buf; size_t beg, end;            /* whole records; buf[end - 1] == '\n' */
    int eof;                               /* the input ends with this block */
    unsigned long long line0;              /* lines read before it */
    uint64_t records;
    int lane;                              /* -1: not given to a lane (ragged end of input, oversized record): host parser */
    int posted;                            /* its lane has the job (0 only between fxh_clip_go_serial and the block's turn) */
} fxh_block;

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

static void fxh_lanes_stop(fxh_run *R, fxh_lane *lanes, int nlanes, double *t_lane_init)
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
static void fxh_run_lanes(fxh_run *R, fxh_prefetch *pf, int nlanes, const int *lane_dev, double *t_read, double *t_lane_init)
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

static uint32_t g_part_clip_len[FXH_MAX_LANES];      /* clipper parts: the one read length each part saw (0: not a clipper run / no reads) */
static int fxh_run_impl(FASTX *fx, const fxg_params *p, fxh_totals *tot, fxh_stats_run *stats, uint64_t **hist_out, uint32_t *cols_out, int part, int nparts)
{
    fxh_run R;
    memset(&R, 0, sizeof R);
    memset(tot, 0, sizeof *tot);
    R.fx = fx; R.p = p; R.tot = tot; R.stats = stats; R.part = part; R.nparts = nparts;
    const int timing = getenv("FXH_TIMING") != NULL;
    const double t_run0 = fxh_now();
    double t_read = 0, t_lane_init = 0, t0;
    int dev[FXH_MAX_LANES];
    int ndev = fxh_device_list(dev, FXH_MAX_LANES);
    if (nparts > 1 && ndev > 1) { dev[0] = dev[part % ndev]; ndev = 1; }      /* a part of a sharded run stays on one GPU */
    R.st_device = dev[0];
    cpu_set_t cpus_before;
    const int moved = ndev == 1 ? fxh_bind_near_device(dev[0], &cpus_before) : 0;
    struct fxh_reader *rd = fx->reader;
    /* one engine call per 64 MB of text; the parts of a sharded run take 8 MB blocks (four parts x two lanes keep the link busy with
     * less to allocate, page-lock and touch first: 52 -> 62 Mreads/s on the 64 M read sample, profiles/r03/p_e2e_parts_block_size.txt) */
    if (!getenv("FXH_READ_BUFFER_MB")) fxh_reader_reserve(rd, (size_t)(nparts > 1 ? 8 : 64) << 20);
    fxh_job *job = &R.job;
    job->fx = fx; job->st = &R.st; job->p = p;
    job->revcomp = (p->stages & (FXG_STAGE_REVCOMP | FXG_STAGE_MASK)) != 0;   /* stages whose output is not a slice of the input text */
    job->has_q = fx->read_fastq;
    job->lpr = fx->read_fastq ? 4 : 2;
    job->fwd_start = (p->stages & FXG_STAGE_FTRIM) && p->ft_first > 1 ? (uint32_t)p->ft_first - 1u : 0u;
    {
        const char *te = getenv("FXH_THREADS");
        long nt = te ? atol(te) : (nparts > 1 ? 4 : 16), ncpu = sysconf(_SC_NPROCESSORS_ONLN);
        if (nt < 1) nt = 1;
        if (nt > 64) nt = 64;
        if (ncpu > 0 && nt > ncpu) nt = ncpu;
        job->nworkers = (int)nt;
    }
    job->w = (fxh_worker *)calloc((size_t)job->nworkers, sizeof(fxh_worker));
    if (!job->w) err(1, "out of memory");
    for (int i = 0; i < job->nworkers; ++i) {
        fxh_worker *w = &job->w[i];
        w->id = i; w->job = job;
        w->shadow = (FASTX *)malloc(sizeof(FASTX));
        if (!w->shadow) err(1, "out of memory");
        memcpy(w->shadow, fx, sizeof(FASTX));
        w->raw.defer_errors = 1;
    }
    fxh_prefetch pf;
    memset(&pf, 0, sizeof pf);
    char *rd_spare = NULL;
    R.overlap = getenv("FXH_NO_OVERLAP") == NULL;
    /* device-side parse/format (FASTQ or FASTA in; the same, or FASTA, out); FXH_HOST_PARSE=1 forces the host parser */
    const int gpu_text = !stats && !g_rename_ids && getenv("FXH_HOST_PARSE") == NULL;
    int nlanes = 0;
    int lane_dev[FXH_MAX_LANES];
    if (gpu_text) {
        /* The clipper's aligner carries state from read to read.  By default the run is parallel for as long as that state
         * cannot matter and goes serial at the first block where it can (clip_auto); FXH_CLIP_PARALLEL=1 is the caller's word that the
         * input has one fixed length (no checks), FXH_CLIP_SERIAL=1 asks for the one aligner from the start. */
        const int clip = (p->stages & FXG_STAGE_CLIP) != 0;
        const int serial = clip && getenv("FXH_CLIP_SERIAL") != NULL && getenv("FXH_CLIP_PARALLEL") == NULL;      /* FXH_CLIP_SERIAL=1: the one aligner from the first read on */
        R.clip_auto = clip && !serial && getenv("FXH_CLIP_PARALLEL") == NULL;      /* the default: parallel while it is exact, see fxh_run.clip_auto */
        const char *le = getenv("FXH_LANES");
        int per = le ? atoi(le) : 2;
        if (per < 1) per = 1;
        if (serial) per = 1;
        for (int d = 0; d < (serial ? 1 : ndev); ++d)
            for (int k = 0; k < per && nlanes < FXH_MAX_LANES; ++k) lane_dev[nlanes++] = dev[d];
        /* interleave the devices: consecutive blocks go to different GPUs */
        if (!serial && ndev > 1) { nlanes = 0; for (int k = 0; k < per; ++k) for (int d = 0; d < ndev && nlanes < FXH_MAX_LANES; ++d) lane_dev[nlanes++] = dev[d]; }
    }

    if (nlanes > 0) {
        fxh_run_lanes(&R, &pf, nlanes, lane_dev, &t_read, &t_lane_init);
    } else {
        while (!R.at_eof && !R.have_err) {
            t0 = fxh_now();
            if (R.overlap) fxh_next_block(&pf, rd, &rd_spare); else fxh_reader_fill(rd);
            t_read += fxh_now() - t0;
            if (rd->beg == rd->end && rd->eof) break;
            fxh_host_block(&R);
        }
    }
    fxh_awriter_stop(&R.aw);
    fxh_prefetch_stop(&pf);
    free(R.clip_seed);
    if (nparts > 1) g_part_clip_len[part] = R.clip_auto ? R.clip_len : 0u;
    if (nparts > 1 && (R.aborted || R.have_err || FXH_ABORTED())) {      /* fxh_run_parts starts the whole job over, unsharded */
        FXH_ABORT_SET();
        for (int i = 0; i < job->nworkers; ++i) { free(job->w[i].rec); free(job->w[i].shadow); }
        free(job->w);
        if (moved) (void)sched_setaffinity(0, sizeof cpus_before, &cpus_before);
        return 2;
    }
    if (R.have_err) {
        if (!stats) fxh_writer_flush(fx->writer);   /* every record before the bad one has been written, like the reference */
        errx(1, "%s", R.errmsg);
    }
    if (stats) {
        const size_t per_col = (size_t)FXG_QS_CLASSES * FXG_QS_BINS * sizeof(uint64_t);
        *cols_out = stats->cols;
        *hist_out = (uint64_t *)calloc(stats->cols ? stats->cols : 1, per_col);
        if (!*hist_out) err(1, "out of memory");
        if (stats->d_hist) {
            FXG_CHECK(&R.st, fxg_memcpy_d2h(R.st.ctx, *hist_out, stats->d_hist, (size_t)stats->cols * per_col));
            FXG_CHECK(&R.st, fxg_sync(R.st.ctx));
        }
    }
    if (timing)
        fprintf(stderr, "fxh timing part %d/%d (%d threads, %s parse, %d lanes on %d GPU(s), %lu host-parsed blocks): run %.3f = init %.3f read %.3f index %.3f pack %.3f format+write %.3f wait-lane %.3f wait-writer %.3f drain %.3f; gpu(h2d+kernel+d2h, summed over lanes) %.3f s\n",
                part, nparts, job->nworkers, gpu_text ? "device" : "host", nlanes, nlanes ? ndev : 1, R.n_fallback, fxh_now() - t_run0, R.t_init + t_lane_init, t_read, R.t_index, R.t_pack, R.t_fmt,
                R.t_wait_lane, R.t_wait_writer, R.t_drain, R.t_gpu);
    if (R.st.ctx) fxg_ctx_destroy(R.st.ctx);
    for (int i = 0; i < job->nworkers; ++i) { free(job->w[i].rec); free(job->w[i].shadow); }
    free(job->w);
    if (moved) (void)sched_setaffinity(0, sizeof cpus_before, &cpus_before);      /* a host that calls in again finds its own CPU set */
    return 0;
}

/* ---------------------------------------------------------------------------------------------- */
/* Sharded run (FXH_PARTS=k): the text-level analogue of fxg_shard_range / fxg_epilogue / fxg_concat_pwrite.  The input file is   */
/* cut into k contiguous byte ranges at record boundaries; k runs (each the lanes loop above: own reader threads, own lanes, own    */
/* writer thread) work through them at the same time, part r on GPU r mod #GPUs, and write k output parts whose concatenation in    */
/* part order is the output of the unsharded run: one writer stream is what caps a single run (a tmpfs or page-cache write is one   */
/* thread under the inode lock), k parts are k streams.  `-o NAME` names part 0 NAME and part r NAME.r; `-o out.%r.fq` substitutes. */
/* An index NAME.parts lists (part, file, input bytes, records in, records out, output bytes).                                       */
/*   A record is four (two) lines counted from the start of the input, so a cut is only KNOWN to be a record boundary when the line */
/* count before it is: the cut points are found by pattern (an '@' line, a '+' line two below it, equal lengths, the same again)    */
/* and then PROVEN -- part r ends exactly at part r+1's cut, so if its lines are a whole number of records and part r started at a  */
/* boundary, so does part r+1 (induction from offset 0).  A part that meets anything the device path does not take (a ragged end =   */
/* a wrong cut, a malformed record, CR-less oddities the host parser owns) stops all parts; the attempt ran in a child process, which */
/* empties the parts and exits, and the parent runs the input as one stream: messages, exit codes and partial output are the         */
/* reference's in every case.                                                                                                         */
/* ---------------------------------------------------------------------------------------------- */
static off_t fxh_find_cut(int fd, off_t from, off_t size, int lpr)
{
    const size_t W = (size_t)4 << 20;
    char *w = (char *)malloc(W);
    if (!w) err(1, "out of memory");
    ssize_t got = pread(fd, w, W, from);
    off_t found = -1;
    if (got > 0) {
        size_t n = (size_t)got, ls[12];
        const char *nl = (const char *)memchr(w, '\n', n);
        size_t pos = nl ? (size_t)(nl - w) + 1 : n;                         /* first line start after `from` */
        while (pos < n && found < 0) {
            int k = 0;                                                       /* starts of this line and the next 2 lpr */
            size_t q = pos;
            while (k < 2 * lpr + 1 && q < n) { ls[k++] = q; const char *e = (const char *)memchr(w + q, '\n', n - q); if (!e) { q = n; break; } q = (size_t)(e - w) + 1; }
            if (k < 2 * lpr + 1) break;                                     /* not enough text in the window */
            int ok;
            if (lpr == 2) ok = w[ls[0]] == '>' && w[ls[2]] == '>';
            else ok = w[ls[0]] == '@' && w[ls[2]] == '+' && (ls[2] - ls[1]) == (ls[4] - ls[3]) &&
                      w[ls[4]] == '@' && w[ls[6]] == '+' && (ls[6] - ls[5]) == (ls[8] - ls[7]);
            if (ok) found = from + (off_t)ls[0];
            else pos = ls[1];
        }
    }
    free(w);
    return (found > 0 && found < size) ? found : -1;
}

typedef struct { FASTX *fx; const fxg_params *p; fxh_totals tot; int part, nparts, rc; pthread_t th; off_t start, limit; char name[PATH_MAX + 16]; } fxh_part;
static void *fxh_part_main(void *arg)
{
    fxh_part *pt = (fxh_part *)arg;
    pt->rc = fxh_run_impl(pt->fx, pt->p, &pt->tot, NULL, NULL, NULL, pt->part, pt->nparts);
    return NULL;
}

static void fxh_part_name(const FASTX *fx, int r, char *dst, size_t cap)
{
    const char *name = fx->output_file_name, *pr = strstr(name, "%r");
    if (pr) snprintf(dst, cap, "%.*s%d%s", (int)(pr - name), name, r, pr + 2);
    else if (r == 0) snprintf(dst, cap, "%s", name);
    else snprintf(dst, cap, "%s.%d", name, r);
}

#define FXH_EXIT_ABANDON 99
/* 0 = done (in the child of the fork below: the caller goes on to print its reports); -1 = run unsharded (not eligible, or the sharded attempt was abandoned) */
static int fxh_run_parts(FASTX *fx, const fxg_params *p, fxh_totals *tot, int k)
{
    struct fxh_reader *rd = fx->reader;
    struct stat sb;
    if (k > FXH_MAX_LANES) k = FXH_MAX_LANES;
    if (rd->fd == STDIN_FILENO || fstat(rd->fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return -1;
    if (strcmp(fx->output_file_name, "-") == 0 || fx->compress_output || g_rename_ids || getenv("FXH_HOST_PARSE")) return -1;
    if ((p->stages & FXG_STAGE_CLIP) && getenv("FXH_CLIP_SERIAL") != NULL && getenv("FXH_CLIP_PARALLEL") == NULL) return -1;      /* one aligner asked for */
    const off_t size = sb.st_size, here = lseek(rd->fd, 0, SEEK_CUR);
    const int lpr = fx->read_fastq ? 4 : 2;
    off_t cut[FXH_MAX_LANES + 1];
    cut[0] = 0; cut[k] = size;
    for (int r = 1; r < k; ++r) {
        cut[r] = fxh_find_cut(rd->fd, (off_t)((unsigned long long)size * (unsigned)r / (unsigned)k), size, lpr);
        if (cut[r] < 0 || cut[r] <= cut[r - 1] || (r == 1 && cut[r] < here)) return -1;       /* small or odd input: one run */
    }
    /* The sharded attempt runs in a CHILD process.  Irregular input anywhere (or a cut that was no record boundary) abandons it: the
     * reference's behaviour -- message, exit code, what has been written before the bad record -- is defined for ONE stream, so the
     * child empties the parts and exits with FXH_EXIT_ABANDON, and this process -- which has not touched the GPU yet -- runs the same
     * input unsharded (part 0 then receives everything).  Nothing is ever exec'd or killed with device work in flight: the child ends
     * like any tool run, after its threads have been joined and its contexts destroyed. */
    if (g_hip_touched) return -1;                /* this process has used the HIP runtime already (a host that calls in twice): no fork over a live runtime */
    /* Every part is opened HERE, before anything has run: an output that cannot take parts -- /dev/null, a FIFO, a directory where the
     * sibling names cannot be created -- means one stream (part 0 alone, as named by the caller), never a failure halfway. */
    int part_fd[FXH_MAX_LANES];
    {
        struct stat ob;
        struct fxh_writer *w0 = fx->writer;
        if (!w0 || w0->fd < 0 || fstat(w0->fd, &ob) != 0 || !S_ISREG(ob.st_mode)) return -1;
        for (int r = 1; r < k; ++r) {
            char name[PATH_MAX + 16];
            fxh_part_name(fx, r, name, sizeof name);
            part_fd[r] = open(name, O_CREAT | O_WRONLY | O_TRUNC, 0666);
            if (part_fd[r] < 0 || fstat(part_fd[r], &ob) != 0 || !S_ISREG(ob.st_mode)) {
                warn("%s: cannot be an output part, running as one stream", name);
                for (int q = 1; q <= r; ++q) if (part_fd[q] >= 0) close(part_fd[q]);
                return -1;
            }
        }
    }
    fflush(NULL);
    const pid_t child = fork();
    if (child < 0) { for (int r = 1; r < k; ++r) close(part_fd[r]); return -1; }
    if (child > 0) {
        int st = 0;
        for (int r = 1; r < k; ++r) close(part_fd[r]);                                  /* the child writes them */
        while (waitpid(child, &st, 0) < 0) { if (errno != EINTR) err(1, "waitpid"); }
        if (WIFEXITED(st) && WEXITSTATUS(st) == FXH_EXIT_ABANDON) {
            if (lseek(rd->fd, here, SEEK_SET) < 0) err(1, "%s", fx->input_file_name);      /* the child read through the shared descriptor */
            struct fxh_writer *w = fx->writer;
            if (w && w->fd >= 0) { if (ftruncate(w->fd, 0) != 0 || lseek(w->fd, 0, SEEK_SET) < 0) warn("%s", fx->output_file_name); }
            return -1;
        }
        if (WIFSIGNALED(st)) { signal(WTERMSIG(st), SIG_DFL); raise(WTERMSIG(st)); _exit(128 + WTERMSIG(st)); }
        _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 1);                                      /* the child printed the reports and closed the parts */
    }
    (void)prctl(PR_SET_PDEATHSIG, SIGTERM);      /* the child: a tool process that was killed takes its sharded attempt along */
    fxh_part *pt = (fxh_part *)calloc((size_t)k, sizeof(fxh_part));
    if (!pt) err(1, "out of memory");
    const char *cap_env = getenv("FXH_READ_BUFFER_MB");
    for (int r = 0; r < k; ++r) {
        pt[r].p = p; pt[r].part = r; pt[r].nparts = k; pt[r].start = cut[r]; pt[r].limit = cut[r + 1];
        fxh_part_name(fx, r, pt[r].name, sizeof pt[r].name);
        if (r == 0) { pt[r].fx = fx; rd->limit = cut[1]; continue; }
        FASTX *f = (FASTX *)malloc(sizeof(FASTX));
        if (!f) err(1, "out of memory");
        memcpy(f, fx, sizeof(FASTX));
        f->reader = fxh_reader_open_range(fx->input_file_name, cap_env && atoi(cap_env) > 0 ? (size_t)atoi(cap_env) << 20 : 0, cut[r], cut[r + 1]);
        f->writer = fxh_writer_open_fd(part_fd[r]);
        f->input_line_number = 0; f->num_input_sequences = f->num_input_reads = f->num_output_sequences = f->num_output_reads = 0;
        pt[r].fx = f;
    }
    __atomic_store_n(&g_parts_abort, 0, __ATOMIC_RELAXED);
    g_parts_mode = 1;
    for (int r = 1; r < k; ++r) if (pthread_create(&pt[r].th, NULL, fxh_part_main, &pt[r]) != 0) err(1, "pthread_create");
    fxh_part_main(&pt[0]);
    for (int r = 1; r < k; ++r) pthread_join(pt[r].th, NULL);
    int bad = FXH_ABORTED();
    for (int r = 0; r < k; ++r) if (pt[r].rc != 0) bad = 1;
    {   /* clipper: every part found reads of one length -- it has to be the SAME length in all of them (a shorter read after a longer one
         * sees the longer one's tail); otherwise the parent runs the input as one stream, which goes serial where it must */
        uint32_t len0 = 0;
        for (int r = 0; r < k && !bad; ++r) { if (!g_part_clip_len[r]) continue; if (!len0) len0 = g_part_clip_len[r]; else if (g_part_clip_len[r] != len0) bad = 1; }
    }
    if (bad) {
        /* Abandoned.  Every thread of every part has been joined and its contexts are gone (fxh_lanes_stop destroys them for a part
         * that stops), the device is idle.  The parts are emptied through their own descriptors, part 0 -- whose descriptor the parent
         * shares -- is emptied here as well, and the process leaves with _exit: no exit handler of this half-finished attempt (the
         * writers' flush-at-exit, the runtime's) gets to run.  The parent then runs the input as one stream (see the fork above). */
        for (int r = 1; r < k; ++r) { struct fxh_writer *w = pt[r].fx->writer; w->len = 0; if (ftruncate(w->fd, 0) != 0) warn("%s", pt[r].name); close(w->fd); w->fd = -1; }
        { struct fxh_writer *w = fx->writer; w->len = 0; if (ftruncate(w->fd, 0) != 0 || lseek(w->fd, 0, SEEK_SET) < 0) warn("%s", pt[0].name); }
        if (getenv("FXH_TIMING")) fprintf(stderr, "fxh timing parts: abandoned, contexts destroyed, parts emptied\n");
        fflush(NULL);
        _exit(FXH_EXIT_ABANDON);
    }
    memset(tot, 0, sizeof *tot);
    FILE *ix = NULL;
    {
        char ixname[PATH_MAX + 8];
        const char *name = fx->output_file_name, *pr = strstr(name, "%r");
        if (pr) snprintf(ixname, sizeof ixname, "%.*sparts%s", (int)(pr - name), name, pr + 2); else snprintf(ixname, sizeof ixname, "%s.parts", name);
        ix = fopen(ixname, "w");
        if (ix) fprintf(ix, "#part\tfile\tinput_bytes\tinput_records\toutput_records\toutput_bytes\n");
    }
    for (int r = 0; r < k; ++r) {
        const fxh_totals *t = &pt[r].tot;
        tot->input_sequences += t->input_sequences; tot->input_reads += t->input_reads; tot->output_sequences += t->output_sequences; tot->output_reads += t->output_reads;
        tot->clip_input += t->clip_input; tot->clip_too_short += t->clip_too_short; tot->clip_adapter_only += t->clip_adapter_only;
        tot->clip_no_adapter += t->clip_no_adapter; tot->clip_adapter_found += t->clip_adapter_found; tot->clip_n += t->clip_n;
        tot->masked_reads += t->masked_reads; tot->masked_nucleotides += t->masked_nucleotides; tot->qtrim_dropped += t->qtrim_dropped;
        if (r > 0) fxh_writer_flush(pt[r].fx->writer);
        const off_t out_bytes = r == 0 ? fx->writer->off + (off_t)fx->writer->len : pt[r].fx->writer->off;
        if (ix) fprintf(ix, "%d\t%s\t%lld\t%zu\t%zu\t%lld\n", r, pt[r].name, (long long)(pt[r].limit - pt[r].start), t->input_sequences, t->output_sequences, (long long)out_bytes);
        if (r > 0) { fxh_writer_close(pt[r].fx->writer); free(pt[r].fx); }
    }
    if (ix) fclose(ix);
    fx->num_input_sequences = tot->input_sequences; fx->num_input_reads = tot->input_reads;
    fx->num_output_sequences = tot->output_sequences; fx->num_output_reads = tot->output_reads;
    free(pt);
    return 0;
}

/* `-o out.%r.fq` without FXH_PARTS: the caller has said where parts may go, the tool picks their number -- four (what one GPU's link and
 * four writer streams take, profiles/r03/l..q_e2e_parts*.txt) for a regular input file of at least 1 GB (FXH_AUTO_PARTS_MIN_MB), where
 * the ~0.1 s of three more contexts is paid back; one otherwise (part 0 then holds everything). */
static int fxh_auto_parts(const FASTX *fx)
{
    struct stat sb;
    if (!strstr(fx->output_file_name, "%r") || strcmp(fx->output_file_name, "-") == 0) return 0;
    const char *me = getenv("FXH_AUTO_PARTS_MIN_MB");
    const long long min_bytes = (me ? atoll(me) : 1024ll) << 20;
    if (fx->reader->fd == STDIN_FILENO || fstat(fx->reader->fd, &sb) != 0 || !S_ISREG(sb.st_mode) || (long long)sb.st_size < min_bytes) return 1;
    return 4;
}

int fxh_run_tool(FASTX *fx, const fxg_params *p, fxh_totals *tot)
{
    const char *pe = getenv("FXH_PARTS");
    int k = pe ? atoi(pe) : fxh_auto_parts(fx);
    if (k > FXH_MAX_LANES) k = FXH_MAX_LANES;
    if (k > 1 && fxh_run_parts(fx, p, tot, k) == 0) return 0;
    const int rc = fxh_run_impl(fx, p, tot, NULL, NULL, NULL, 0, 1);
    if (k > 1 && strcmp(fx->output_file_name, "-") != 0) {
        /* asked for k parts but run as one stream (a pipe, a small file, -z, the serial clipper): part 0 holds everything, the others
         * exist and are empty, so that `cat` over the k names is the output either way */
        for (int r = 1; r < k; ++r) {
            char name[PATH_MAX + 16];
            fxh_part_name(fx, r, name, sizeof name);
            FILE *f = fopen(name, "w");
            if (f) fclose(f);
        }
    }
    return rc;
}

int fxh_run_quality_stats(FASTX *fx, uint64_t **hist, uint32_t *cols, fxh_totals *tot)
{
    fxg_params p;
    fxh_stats_run sr = {NULL, 0};
    fxh_default_params(&p, fx->fastq_ascii_quality_offset);
    return fxh_run_impl(fx, &p, tot, &sr, hist, cols, 0, 1);
}
