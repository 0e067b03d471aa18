/* fxh_batch.c -- see fxh_batch.h: buffers and record output, the host-parsed path, the run driver and the public entry points. */
#include "fxh_priv.h"
int g_rename_ids = 0;

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

void fxh_grow_device(fxh_state *st, size_t reads, size_t bytes, int revcomp)
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

void fxh_set_rename_ids(int on) { g_rename_ids = on; }

/* Size (dst == NULL) or write one kept record; seq/qual point at `len` output bytes; qual bytes are either raw input
 * characters (raw_qual) or Phred+33 codes (engine output / numeric input).  Returns the number of bytes. */

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

static void *fxh_thread_main(void *arg) { fxh_worker *w = (fxh_worker *)arg; w->job->phase(w); return NULL; }

void fxh_parallel(fxh_job *job, void (*phase)(fxh_worker *))
{
    pthread_t th[64];
    job->phase = phase;
    for (int i = 1; i < job->nworkers; ++i)
        if (pthread_create(&th[i], NULL, fxh_thread_main, &job->w[i]) != 0) err(1, "pthread_create");
    phase(&job->w[0]);
    for (int i = 1; i < job->nworkers; ++i) pthread_join(th[i], NULL);
}

void fxh_phase_census(fxh_worker *w)
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
    fxh_note_recoveries(st);
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
void fxh_host_block(fxh_run *R)
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

void fxh_add_counters(fxh_totals *tot, const uint64_t *ctr, uint64_t n, const uint64_t *weighted)
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

int fxh_run_impl(FASTX *fx, const fxg_params *p, fxh_totals *tot, fxh_stats_run *stats, uint64_t **hist_out, uint32_t *cols_out, int part, int nparts)
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
    /* (a pipe: 16 MB -- the process downstream of another tool should start on its first block while the rest is still being produced) */
    if (!getenv("FXH_READ_BUFFER_MB")) {
        struct stat isb;
        const int fifo = fstat(rd->fd, &isb) == 0 && S_ISFIFO(isb.st_mode);
        fxh_reader_reserve(rd, (size_t)(nparts > 1 ? 8 : fifo ? 16 : 64) << 20);
    }
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

int fxh_run_tool(FASTX *fx, const fxg_params *p, fxh_totals *tot)
{
    const char *pe = getenv("FXH_PARTS");
    int k = pe ? atoi(pe) : fxh_auto_parts(fx);
    if (k > FXH_MAX_LANES) k = FXH_MAX_LANES;
    if (k > 1 && fxh_run_parts(fx, p, tot, k) == 0) return 0;
    /* `-o ONE_FILE` on a large regular input: the same parallel run into one file (fxh_strands.c); anything it does not take comes back here */
    if (k <= 1 && !pe && fxh_run_one_file(fx, p, tot) == 0) return 0;
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

