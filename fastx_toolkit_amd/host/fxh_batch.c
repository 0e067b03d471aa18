/* fxh_batch.c -- see fxh_batch.h. */
#define _GNU_SOURCE
#include "fxh_batch.h"

#include <err.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
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
    uint8_t *d_text, *d_out_text;
    uint32_t *d_ls;
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
    if (reads > st->d_cap_reads || bytes > st->d_cap_bytes) {
        if (st->d_bases) {
            fxg_free_device(st->ctx, st->d_bases); fxg_free_device(st->ctx, st->d_qual); fxg_free_device(st->ctx, st->d_len);
            fxg_free_device(st->ctx, st->d_res);
            if (st->d_out_bases) { fxg_free_device(st->ctx, st->d_out_bases); fxg_free_device(st->ctx, st->d_out_qual); }
        }
        st->d_cap_reads = st->h_cap_reads;
        st->d_cap_bytes = st->h_cap_bytes;
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
    int state;                         /* 0 idle, 1 requested, 2 done, 3 quit */
} fxh_prefetch;

static void *fxh_prefetch_main(void *arg)
{
    fxh_prefetch *pf = (fxh_prefetch *)arg;
    pthread_mutex_lock(&pf->mu);
    for (;;) {
        while (pf->state != 1 && pf->state != 3) pthread_cond_wait(&pf->cv, &pf->mu);
        if (pf->state == 3) break;
        char *buf = pf->buf; const size_t cap = pf->cap;
        pthread_mutex_unlock(&pf->mu);
        size_t got = 0; int eof = 0;
        const size_t gap = pf->gap;
        while (gap + got < cap) {
            ssize_t k = read(pf->fd, buf + gap + got, cap - gap - got);
            if (k < 0) { if (errno == EINTR) continue; err(1, "read failed"); }
            if (k == 0) { eof = 1; break; }
            got += (size_t)k;
        }
        pthread_mutex_lock(&pf->mu);
        pf->filled = got; pf->eof = eof; pf->state = 2;
        pthread_cond_broadcast(&pf->cv);
    }
    pthread_mutex_unlock(&pf->mu);
    return NULL;
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

/* ---------------------------------------------------------------------------------------------- */
/* device text path: the block is indexed, checked, packed and formatted on the GPU.   */
/* Returns 1 if the block was handled, 0 if it is irregular in any way (the caller then uses the     */
/* host parser for this block, which owns the reference's messages and corner cases).               */
/* ---------------------------------------------------------------------------------------------- */
static void fxh_register_once(fxh_state *st, void *ptr, size_t bytes)
{
    for (int i = 0; i < 4; ++i) if (st->registered[i] == ptr) return;
    for (int i = 0; i < 4; ++i)
        if (!st->registered[i]) { if (fxg_host_register(st->ctx, ptr, bytes) == 0) st->registered[i] = ptr; return; }
}

static int fxh_block_gpu_text(FASTX *fx, fxh_state *st, const fxg_params *p, fxh_totals *tot, struct fxh_writer *wr, int revcomp, uint32_t fwd_start)
{
    struct fxh_reader *rd = fx->reader;
    size_t len = rd->end - rd->beg;
    if (len == 0) return 0;
    if (rd->eof && rd->buf[rd->end - 1] != '\n') rd->buf[rd->end] = '\n', len += 1;     /* the buffer has one spare byte */
    if (st->d_text_cap < len + 32) {
        if (st->d_text) { fxg_free_device(st->ctx, st->d_text); fxg_free_device(st->ctx, st->d_out_text); }
        st->d_text_cap = len + len / 8 + 4096;
        FXG_CHECK(st, fxg_malloc_device(st->ctx, st->d_text_cap, (void **)&st->d_text));
        FXG_CHECK(st, fxg_malloc_device(st->ctx, st->d_text_cap, (void **)&st->d_out_text));
    }
    const size_t cap_lines = len / 2 + 16;                  /* the shortest record "@\nA\n+\nI\n" has 8 bytes and 4 lines */
    if (st->d_ls_cap < cap_lines) {
        if (st->d_ls) { fxg_free_device(st->ctx, st->d_ls); fxg_free_device(st->ctx, st->d_len16); }
        st->d_ls_cap = cap_lines + cap_lines / 8;
        FXG_CHECK(st, fxg_malloc_device(st->ctx, st->d_ls_cap * sizeof(uint32_t), (void **)&st->d_ls));
        FXG_CHECK(st, fxg_malloc_device(st->ctx, (st->d_ls_cap / 4 + 4) * sizeof(uint16_t), (void **)&st->d_len16));
    }
    fxh_register_once(st, rd->buf, rd->cap + 1);
    FXG_CHECK(st, fxg_memcpy_h2d(st->ctx, st->d_text, rd->buf + rd->beg, len));
    fxg_text_info info;
    FXG_CHECK(st, fxg_fastq_index(st->ctx, st->d_text, len, rd->eof, st->d_ls, st->d_ls_cap, st->d_len16, &info));
    if (info.irregular || info.records == 0) return 0;
    const uint64_t n = info.records;
    const uint32_t stride = info.max_len;
    if ((uint64_t)n * stride > (uint64_t)8 * len + (1u << 20)) return 0;   /* ragged beyond reason: let the host path split it */
    fxh_grow(st, n, (size_t)n * stride + 16, revcomp);
    uint32_t irr = 0;
    FXG_CHECK(st, fxg_fastq_pack(st->ctx, st->d_text, len, st->d_ls, n, stride, fx->fastq_ascii_quality_offset, st->d_bases, st->d_qual, &irr));
    if (irr) return 0;
    if (revcomp && st->d_off_cap < n) {
        if (st->d_out_off) fxg_free_device(st->ctx, st->d_out_off);
        st->d_off_cap = n + n / 8 + 1024;
        FXG_CHECK(st, fxg_malloc_device(st->ctx, st->d_off_cap * sizeof(uint64_t), (void **)&st->d_out_off));
    }
    const int fixed = info.min_len == info.max_len;
    fxg_batch in = {st->d_bases, st->d_qual, fixed ? NULL : st->d_len16, stride, stride, n};
    fxg_out out = {st->d_res, revcomp ? st->d_out_bases : NULL, revcomp ? st->d_out_qual : NULL, NULL, NULL, revcomp ? st->d_out_off : NULL, st->d_counters};
    fxg_params pp = *p;
    pp.qoffset = 33;
    FXG_CHECK(st, fxg_run_pipeline(st->ctx, &in, &pp, &out));
    uint64_t ctr[FXG_NCOUNTERS];
    {
        int rc = fxg_read_counters(st->ctx, st->d_counters, ctr);
        if (rc == FXG_E_DEVICE && (ctr[FXG_C_ERRORS] & FXG_DEV_ERR_BAD_BASE)) errx(1, "%s", fxg_last_error(st->ctx));
        if (rc != 0) errx(1, "GPU engine error %d: %s", rc, fxg_last_error(st->ctx));
    }
    uint64_t out_bytes = 0;
    FXG_CHECK(st, fxg_fastq_format(st->ctx, st->d_text, st->d_ls, n, st->d_res, revcomp ? 0u : fwd_start, revcomp ? st->d_out_bases : NULL,
                                   revcomp ? st->d_out_qual : NULL, revcomp ? st->d_out_off : NULL, fx->fastq_ascii_quality_offset, st->d_out_text, &out_bytes));
    char *dst = fxh_writer_reserve(wr, out_bytes + 16);
    FXG_CHECK(st, fxg_memcpy_d2h(st->ctx, dst, st->d_out_text, out_bytes));
    FXG_CHECK(st, fxg_sync(st->ctx));
    wr->len += out_bytes;
    /* FASTQ ids are never collapsed: every record counts as one read (fastx.c:480-481) */
    tot->input_sequences += n; tot->input_reads += n;
    tot->output_sequences += ctr[FXG_C_KEPT]; tot->output_reads += ctr[FXG_C_KEPT];
    tot->clip_input += (unsigned)n;
    tot->clip_too_short += (unsigned)ctr[FXG_C_CLIP_TOO_SHORT]; tot->clip_adapter_only += (unsigned)ctr[FXG_C_CLIP_ADAPTER_ONLY];
    tot->clip_no_adapter += (unsigned)ctr[FXG_C_CLIP_NO_ADAPTER]; tot->clip_adapter_found += (unsigned)ctr[FXG_C_CLIP_ADAPTER_FOUND];
    tot->clip_n += (unsigned)ctr[FXG_C_CLIP_N];
    tot->masked_reads += ctr[FXG_C_MASKED_READS]; tot->masked_nucleotides += ctr[FXG_C_MASKED_NT];
    rd->beg += (size_t)info.consumed > rd->end - rd->beg ? rd->end - rd->beg : (size_t)info.consumed;
    fx->input_line_number += 4ull * n;
    return 1;
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

static int fxh_run_impl(FASTX *fx, const fxg_params *p, fxh_totals *tot, fxh_stats_run *stats, uint64_t **hist_out, uint32_t *cols_out)
{
    fxh_state st;
    memset(&st, 0, sizeof st);
    memset(tot, 0, sizeof *tot);
    const int timing = getenv("FXH_TIMING") != NULL;
    double t_init = fxh_now(), t_read = 0, t_index = 0, t_pack = 0, t_gpu = 0, t_fmt = 0, t0;
    {
        const char *dev = getenv("FXG_DEVICE");
        int rc = fxg_ctx_create(dev ? atoi(dev) : 0, &st.ctx);
        if (rc != 0) errx(1, "no usable MI355X/HIP device (fxg_ctx_create = %d); this build has no CPU path", rc);
    }
    FXG_CHECK(&st, fxg_malloc_device(st.ctx, FXG_NCOUNTERS * sizeof(uint64_t), (void **)&st.d_counters));
    /* one fastx_clipper process = one aligner whose query buffer survives from read to read (sequence_alignment.cpp:135-136,
     * SURVEY N3): the engine reproduces that across the batches of this run */
    if (p->stages & FXG_STAGE_CLIP) FXG_CHECK(&st, fxg_set_clip_history(st.ctx, 1));
    t_init = fxh_now() - t_init;
    struct fxh_reader *rd = fx->reader;
    if (!getenv("FXH_READ_BUFFER_MB")) fxh_reader_reserve(rd, (size_t)64 << 20);   /* one engine call per 64 MB of text */
    fxh_job job;
    memset(&job, 0, sizeof job);
    job.fx = fx; job.st = &st; job.p = p;
    job.revcomp = (p->stages & (FXG_STAGE_REVCOMP | FXG_STAGE_MASK)) != 0;   /* stages whose output is not a slice of the input text */
    job.has_q = fx->read_fastq;
    job.lpr = fx->read_fastq ? 4 : 2;
    job.fwd_start = (p->stages & FXG_STAGE_FTRIM) && p->ft_first > 1 ? (uint32_t)p->ft_first - 1u : 0u;
    {
        const char *te = getenv("FXH_THREADS");
        long nt = te ? atol(te) : 16, ncpu = sysconf(_SC_NPROCESSORS_ONLN);
        if (nt < 1) nt = 1;
        if (nt > 64) nt = 64;
        if (ncpu > 0 && nt > ncpu) nt = ncpu;
        job.nworkers = (int)nt;
    }
    job.w = (fxh_worker *)calloc((size_t)job.nworkers, sizeof(fxh_worker));
    if (!job.w) err(1, "out of memory");
    for (int i = 0; i < job.nworkers; ++i) {
        fxh_worker *w = &job.w[i];
        w->id = i; w->job = &job;
        w->shadow = (FASTX *)malloc(sizeof(FASTX));
        if (!w->shadow) err(1, "out of memory");
        memcpy(w->shadow, fx, sizeof(FASTX));
        w->raw.defer_errors = 1;
    }
    char errmsg[768];
    int have_err = 0, at_eof = 0;
    fxh_prefetch pf;
    fxh_awriter aw;
    memset(&pf, 0, sizeof pf);
    memset(&aw, 0, sizeof aw);
    char *rd_spare = NULL, *wr_spare = NULL;
    size_t wr_spare_cap = 0;
    const int overlap = getenv("FXH_NO_OVERLAP") == NULL;
    /* device-side parse/format for FASTQ; FXH_HOST_PARSE=1 forces the host parser */
    const int gpu_text = !stats && fx->read_fastq && fx->write_fastq && !g_rename_ids && getenv("FXH_HOST_PARSE") == NULL;
    unsigned long n_fallback = 0;

    while (!at_eof && !have_err) {
        /* ---- 1. fill the block, split it into record-aligned ranges, index + validate them in parallel ---- */
        t0 = fxh_now();
        if (overlap) fxh_next_block(&pf, rd, &rd_spare); else fxh_reader_fill(rd);
        t_read += fxh_now() - t0; t0 = fxh_now();
        if (rd->beg == rd->end && rd->eof) break;
        if (gpu_text) {
            struct fxh_writer *wr0 = fx->writer;
            if (fxh_block_gpu_text(fx, &st, p, tot, wr0, job.revcomp, job.fwd_start)) {
                t_gpu += fxh_now() - t0;
                if (overlap) fxh_awriter_submit(&aw, wr0, &wr_spare, &wr_spare_cap); else fxh_writer_flush(wr0);
                fx->num_input_sequences = tot->input_sequences; fx->num_input_reads = tot->input_reads;
                fx->num_output_sequences = tot->output_sequences; fx->num_output_reads = tot->output_reads;
                if (rd->eof && rd->beg >= rd->end) at_eof = 1;
                continue;
            }
            n_fallback++;
        }
        const size_t beg = rd->beg, end = rd->end;
        const int T = job.nworkers;
        for (int i = 0; i < T; ++i) {
            job.w[i].a0 = beg + (size_t)((unsigned long long)(end - beg) * (unsigned)i / (unsigned)T);
            job.w[i].a1 = beg + (size_t)((unsigned long long)(end - beg) * (unsigned)(i + 1) / (unsigned)T);
        }
        fxh_parallel(&job, fxh_phase_census);
        {
            /* worker i starts at the first record boundary at or after a0: lines are counted from the block start,
             * which is itself a record boundary (the previous block stopped at one) */
            unsigned long long lines_before = 0;          /* complete lines in [beg, a0) */
            for (int i = 0; i < T; ++i) {
                fxh_worker *w = &job.w[i];
                size_t s;
                unsigned long long ls;                    /* complete lines in [beg, s) */
                if (i == 0) { s = beg; ls = 0; }
                else if (w->first_nl == (size_t)-1) { s = (size_t)-1; ls = 0; }       /* no line starts here: same boundary as the next range */
                else {
                    s = w->first_nl + 1; ls = lines_before + 1;
                    while (ls % (unsigned)job.lpr != 0) {         /* walk to the next record boundary */
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
                if (job.w[i].start == (size_t)-1) job.w[i].start = (i + 1 < T) ? job.w[i + 1].start : end;
            for (int i = 0; i < T; ++i) {
                fxh_worker *w = &job.w[i];
                w->view.buf = rd->buf; w->view.cap = rd->cap; w->view.fd = -1;
                w->view.beg = w->start;
                w->view.end = (i + 1 < T) ? job.w[i + 1].start : end;
                w->view.eof = 0;
            }
            /* the last non-empty range owns the end-of-input / incomplete-tail semantics */
            for (int i = T - 1; i >= 0; --i)
                if (job.w[i].view.beg < end || i == 0) { job.w[i].view.end = end; job.w[i].view.eof = rd->eof; break; }
        }
        fxh_parallel(&job, fxh_phase_index);
        /* merge in input order; the first error / end condition wins */
        size_t n = 0, maxlen = 0, minlen = (size_t)-1;
        int stop = -1;                                   /* worker at which the batch ends */
        for (int i = 0; i < T; ++i) {
            fxh_worker *w = &job.w[i];
            w->rec0 = n; w->use = w->nrec;
            n += w->nrec;
            if (w->nrec) { if (w->maxlen > maxlen) maxlen = w->maxlen; if (w->minlen < minlen) minlen = w->minlen; }
            const int is_last = (w->view.end == end);
            if (w->rc_end == -2) { have_err = 1; memcpy(errmsg, w->errmsg, sizeof errmsg); stop = i; }
            else if (w->rc_end == 0) { at_eof = 1; stop = i; }
            else if (is_last) stop = i;
            if (stop >= 0) { rd->beg = w->end_pos; fx->input_line_number = w->end_line; break; }
        }
        for (int i = stop + 1; i < T; ++i) { job.w[i].use = 0; job.w[i].rec0 = n; }
        if (n == 0) {
            if (!have_err && !at_eof) errx(1, "input record does not fit in the %zu MB read buffer", rd->cap >> 20);
            break;
        }
        t_index += fxh_now() - t0; t0 = fxh_now();

        /* ---- 2. pack the SoA rows (qualities normalised to Phred+33 codes) ---- */
        job.stride = (uint32_t)maxlen;
        fxh_grow(&st, n, n * (size_t)job.stride, job.revcomp);
        fxh_parallel(&job, fxh_phase_pack);
        for (int i = 0; i <= stop; ++i) {
            fxh_worker *w = &job.w[i];
            if (w->bad_q >= 0) {                       /* first bad record wins: drop it and everything after it */
                have_err = 1; at_eof = 0;
                memcpy(errmsg, w->errmsg, sizeof errmsg);
                n = w->rec0 + (size_t)w->bad_q;
                w->use = (size_t)w->bad_q;
                for (int j = i + 1; j < T; ++j) job.w[j].use = 0;
                break;
            }
        }
        t_pack += fxh_now() - t0; t0 = fxh_now();
        if (n > 0) {
            /* ---- 3. engine ---- */
            const uint32_t stride = job.stride;
            const size_t bytes = n * (size_t)stride;
            const int fixed = (minlen == maxlen);
            FXG_CHECK(&st, fxg_memcpy_h2d(st.ctx, st.d_bases, st.h_bases, bytes));
            if (job.has_q) FXG_CHECK(&st, fxg_memcpy_h2d(st.ctx, st.d_qual, st.h_qual, bytes));
            if (!fixed) FXG_CHECK(&st, fxg_memcpy_h2d(st.ctx, st.d_len, st.h_len, n * sizeof(uint16_t)));
            fxg_batch in = {st.d_bases, job.has_q ? st.d_qual : NULL, fixed ? NULL : st.d_len, (uint32_t)maxlen, stride, n};
            if (stats) {                            /* fastx_quality_stats: reduce, nothing to write */
                fxh_stats_reserve(&st, stats, stride);
                FXG_CHECK(&st, fxg_run_quality_stats(st.ctx, &in, stats->d_hist, stats->cols));
                FXG_CHECK(&st, fxg_sync(st.ctx));
                tot->input_sequences += n; tot->input_reads += n;
                fx->num_input_sequences = tot->input_sequences; fx->num_input_reads = tot->input_reads;
                t_gpu += fxh_now() - t0;
                continue;
            }
            fxg_out out = {st.d_res, job.revcomp ? st.d_out_bases : NULL, (job.revcomp && job.has_q) ? st.d_out_qual : NULL, NULL, NULL, NULL, st.d_counters};
            fxg_params pp = *p;
            pp.qoffset = 33;                        /* rows hold Phred+33 codes whatever -Q was */
            FXG_CHECK(&st, fxg_run_pipeline(st.ctx, &in, &pp, &out));
            FXG_CHECK(&st, fxg_memcpy_d2h(st.ctx, st.h_res, st.d_res, n * sizeof(uint32_t)));
            uint64_t ctr[FXG_NCOUNTERS];
            {
                int rc = fxg_read_counters(st.ctx, st.d_counters, ctr);   /* synchronises */
                if (rc == FXG_E_DEVICE && (ctr[FXG_C_ERRORS] & FXG_DEV_ERR_BAD_BASE)) errx(1, "%s", fxg_last_error(st.ctx));
                if (rc != 0) errx(1, "GPU engine error %d: %s", rc, fxg_last_error(st.ctx));
            }
            tot->masked_reads += ctr[FXG_C_MASKED_READS]; tot->masked_nucleotides += ctr[FXG_C_MASKED_NT];
            if (job.revcomp) {
                FXG_CHECK(&st, fxg_memcpy_d2h(st.ctx, st.h_out_bases, st.d_out_bases, ctr[FXG_C_KEPT_BASES]));
                if (job.has_q) FXG_CHECK(&st, fxg_memcpy_d2h(st.ctx, st.h_out_qual, st.d_out_qual, ctr[FXG_C_KEPT_BASES]));
                FXG_CHECK(&st, fxg_sync(st.ctx));
            }
            t_gpu += fxh_now() - t0; t0 = fxh_now();

            /* ---- 4. format the kept records in input order (each worker its own slice), tally the report counters ---- */
            fxh_parallel(&job, fxh_phase_count);
            {
                size_t base = tot->output_sequences + 1;
                for (int i = 0; i < T; ++i) { job.w[i].kept_base = base; base += job.w[i].kept_count; }
            }
            fxh_parallel(&job, fxh_phase_size);
            size_t total = 0, kept_total = 0;
            for (int i = 0; i < T; ++i) {
                fxh_worker *w = &job.w[i];
                w->out_off = total; w->kept_off = kept_total;
                total += w->out_bytes; kept_total += w->kept_bytes;
                tot->input_sequences += w->tot.input_sequences; tot->input_reads += w->tot.input_reads;
                tot->output_sequences += w->tot.output_sequences; tot->output_reads += w->tot.output_reads;
                tot->clip_input += w->tot.clip_input; tot->clip_too_short += w->tot.clip_too_short;
                tot->clip_adapter_only += w->tot.clip_adapter_only; tot->clip_no_adapter += w->tot.clip_no_adapter;
                tot->clip_adapter_found += w->tot.clip_adapter_found; tot->clip_n += w->tot.clip_n;
            }
            struct fxh_writer *wr = fx->writer;
            job.out_dst = fxh_writer_reserve(wr, total + 16);
            fxh_parallel(&job, fxh_phase_format);
            wr->len += total;
            if (overlap) fxh_awriter_submit(&aw, wr, &wr_spare, &wr_spare_cap); else fxh_writer_flush(wr);
            fx->num_input_sequences = tot->input_sequences; fx->num_input_reads = tot->input_reads;
            fx->num_output_sequences = tot->output_sequences; fx->num_output_reads = tot->output_reads;
            t_fmt += fxh_now() - t0;
        }
    }
    fxh_awriter_stop(&aw);
    fxh_prefetch_stop(&pf);
    if (have_err) {
        if (!stats) fxh_writer_flush(fx->writer);   /* every record before the bad one has been written, like the reference */
        errx(1, "%s", errmsg);
    }
    if (stats) {
        const size_t per_col = (size_t)FXG_QS_CLASSES * FXG_QS_BINS * sizeof(uint64_t);
        *cols_out = stats->cols;
        *hist_out = (uint64_t *)calloc(stats->cols ? stats->cols : 1, per_col);
        if (!*hist_out) err(1, "out of memory");
        if (stats->d_hist) {
            FXG_CHECK(&st, fxg_memcpy_d2h(st.ctx, *hist_out, stats->d_hist, (size_t)stats->cols * per_col));
            FXG_CHECK(&st, fxg_sync(st.ctx));
        }
    }
    if (timing)
        fprintf(stderr, "fxh timing (%d threads, %s parse, %lu host-parsed blocks): init %.3f read %.3f index %.3f pack %.3f gpu(h2d+kernel+d2h) %.3f format+write %.3f s\n",
                job.nworkers, gpu_text ? "device" : "host", n_fallback, t_init, t_read, t_index, t_pack, t_gpu, t_fmt);
    fxg_ctx_destroy(st.ctx);
    for (int i = 0; i < job.nworkers; ++i) { free(job.w[i].rec); free(job.w[i].shadow); }
    free(job.w);
    return 0;
}

int fxh_run_tool(FASTX *fx, const fxg_params *p, fxh_totals *tot) { return fxh_run_impl(fx, p, tot, NULL, NULL, NULL); }

int fxh_run_quality_stats(FASTX *fx, uint64_t **hist, uint32_t *cols, fxh_totals *tot)
{
    fxg_params p;
    fxh_stats_run sr = {NULL, 0};
    fxh_default_params(&p, fx->fastq_ascii_quality_offset);
    return fxh_run_impl(fx, &p, tot, &sr, hist, cols);
}
