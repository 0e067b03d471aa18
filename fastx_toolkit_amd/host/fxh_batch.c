/* fxh_batch.c -- see fxh_batch.h. */
#define _GNU_SOURCE
#include "fxh_batch.h"

#include <err.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fxh_internal.h"

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
    /* record index of the current batch */
    fxh_rec *rec;
    size_t rec_cap;
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

/* emit one kept record; seq/qual point at `len` output bytes; qual bytes are either raw input characters
 * (raw_qual) or Phred+33 codes (engine output / numeric input) */
static void fxh_emit(FASTX *fx, const fxh_rec *r, const uint8_t *seq, const uint8_t *qual, size_t len, int raw_qual)
{
    struct fxh_writer *w = fx->writer;
    char *d = fxh_writer_reserve(w, (size_t)r->name_len + r->name2_len + 6 * len + 16);
    size_t k = 0;
    d[k++] = fx->output_sequence_id_prefix;
    memcpy(d + k, r->name, r->name_len); k += r->name_len; d[k++] = '\n';
    memcpy(d + k, seq, len); k += len; d[k++] = '\n';
    if (fx->write_fastq) {
        const int ascii = fx->copy_input_fastq_format_to_output ? r->is_ascii : fx->write_fastq_ascii;   /* R6 */
        d[k++] = '+';
        memcpy(d + k, r->name2, r->name2_len); k += r->name2_len; d[k++] = '\n';
        if (ascii) {
            if (raw_qual) { memcpy(d + k, qual, len); k += len; }                      /* R8: q + Q is the input byte */
            else { const int sh = fx->fastq_ascii_quality_offset - 33; for (size_t i = 0; i < len; ++i) d[k++] = (char)(qual[i] + sh); }
        } else {
            if (raw_qual) {   /* ASCII input, numeric output requested */
                unsigned char tmp[64]; size_t i = 0;
                while (i < len) {
                    size_t m = len - i < sizeof tmp ? len - i : sizeof tmp;
                    for (size_t j = 0; j < m; ++j) tmp[j] = (unsigned char)((int)(signed char)qual[i + j] - fx->fastq_ascii_quality_offset + 33);
                    if (i) d[k++] = ' ';
                    k += fxh_format_numeric(d + k, NULL, tmp, m);
                    i += m;
                }
            } else k += fxh_format_numeric(d + k, NULL, qual, len);
        }
        d[k++] = '\n';
    }
    w->len += k;
}

int fxh_run_tool(FASTX *fx, const fxg_params *p, fxh_totals *tot)
{
    fxh_state st;
    memset(&st, 0, sizeof st);
    memset(tot, 0, sizeof *tot);
    {
        const char *dev = getenv("FXG_DEVICE");
        int rc = fxg_ctx_create(dev ? atoi(dev) : 0, &st.ctx);
        if (rc != 0) errx(1, "no usable MI355X/HIP device (fxg_ctx_create = %d); this build has no CPU path", rc);
    }
    FXG_CHECK(&st, fxg_malloc_device(st.ctx, FXG_NCOUNTERS * sizeof(uint64_t), (void **)&st.d_counters));
    const int revcomp = (p->stages & FXG_STAGE_REVCOMP) != 0;
    const int has_q = fx->read_fastq;
    const uint32_t fwd_start = (p->stages & FXG_STAGE_FTRIM) && p->ft_first > 1 ? (uint32_t)p->ft_first - 1u : 0u;
    const size_t max_batch = 4u << 20;   /* reads per engine call */
    struct fxh_reader *rd = fx->reader;
    struct fxh_rawrec raw;
    memset(&raw, 0, sizeof raw);
    raw.defer_errors = 1;
    char errmsg[sizeof raw.errmsg];
    int have_err = 0, at_eof = 0;

    while (!at_eof && !have_err) {
        /* ---- 1. index the records that are completely inside the buffer ---- */
        fxh_reader_fill(rd);
        size_t n = 0, maxlen = 0, minlen = (size_t)-1;
        const unsigned long long first_line = fx->input_line_number;
        for (;;) {
            if (n == max_batch) break;
            int rc = fxh_next_raw(fx, &raw, 0);
            if (rc == 0) { at_eof = 1; break; }
            if (rc == -1) {
                if (n == 0) errx(1, "input record does not fit in the %zu MB read buffer", rd->cap >> 20);
                break;
            }
            if (rc == -2) { have_err = 1; memcpy(errmsg, raw.errmsg, sizeof errmsg); break; }
            if (n == st.rec_cap) {
                st.rec_cap = st.rec_cap ? st.rec_cap * 2 : (1u << 16);
                st.rec = (fxh_rec *)realloc(st.rec, st.rec_cap * sizeof(fxh_rec));
                if (!st.rec) err(1, "out of memory");
            }
            fxh_rec *r = &st.rec[n++];
            r->name = raw.name; r->seq = raw.seq; r->name2 = raw.name2; r->qual = raw.qual;
            r->name_len = (uint32_t)raw.name_len; r->seq_len = (uint32_t)raw.seq_len;
            r->name2_len = (uint32_t)raw.name2_len; r->qual_len = (uint32_t)raw.qual_len;
            r->is_ascii = (uint8_t)raw.is_ascii;
            r->reads_count = (uint32_t)fxh_reads_count(fx, raw.name, raw.name_len);
            if (raw.seq_len > maxlen) maxlen = raw.seq_len;
            if (raw.seq_len < minlen) minlen = raw.seq_len;
        }
        if (n == 0) break;

        /* ---- 2. pack the SoA rows (qualities normalised to Phred+33 codes) ---- */
        uint32_t stride = (uint32_t)maxlen;
        fxh_grow(&st, n, n * (size_t)stride, revcomp);
        {
            unsigned long long line = first_line;   /* line number of the quality line for deferred messages */
            for (size_t i = 0; i < n; ++i) {
                const fxh_rec *r = &st.rec[i];
                memcpy(st.h_bases + i * stride, r->seq, r->seq_len);
                st.h_len[i] = (uint16_t)r->seq_len;
                line += has_q ? 4 : 2;
                if (!has_q) continue;
                uint8_t *q = st.h_qual + i * stride;
                if (r->is_ascii && fx->fastq_ascii_quality_offset == 33) {
                    uint8_t bad = 0;
                    for (uint32_t j = 0; j < r->qual_len; ++j) { const uint8_t c = (uint8_t)r->qual[j]; bad |= (uint8_t)((c < 18) | (c > 126)); }
                    if (!bad) { memcpy(q, r->qual, r->qual_len); continue; }
                }
                raw.seq_len = r->seq_len; raw.qual = r->qual; raw.qual_len = r->qual_len; raw.is_ascii = r->is_ascii;
                const unsigned long long save = fx->input_line_number;
                fx->input_line_number = line;
                const int qrc = fxh_decode_quality(fx, &raw, NULL, q);
                fx->input_line_number = save;
                if (qrc != 0) {                 /* first bad record wins: drop it and everything after it */
                    have_err = 1; at_eof = 0;
                    memcpy(errmsg, raw.errmsg, sizeof errmsg);
                    n = i;
                    break;
                }
            }
        }
        if (n == 0) break;

        /* ---- 3. engine ---- */
        const size_t bytes = n * (size_t)stride;
        const int fixed = (minlen == maxlen);
        FXG_CHECK(&st, fxg_memcpy_h2d(st.ctx, st.d_bases, st.h_bases, bytes));
        if (has_q) FXG_CHECK(&st, fxg_memcpy_h2d(st.ctx, st.d_qual, st.h_qual, bytes));
        if (!fixed) FXG_CHECK(&st, fxg_memcpy_h2d(st.ctx, st.d_len, st.h_len, n * sizeof(uint16_t)));
        fxg_batch in = {st.d_bases, has_q ? st.d_qual : NULL, fixed ? NULL : st.d_len, (uint32_t)maxlen, stride, n};
        fxg_out out = {st.d_res, revcomp ? st.d_out_bases : NULL, (revcomp && has_q) ? st.d_out_qual : NULL, NULL, NULL, NULL, st.d_counters};
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
        if (revcomp) {
            FXG_CHECK(&st, fxg_memcpy_d2h(st.ctx, st.h_out_bases, st.d_out_bases, ctr[FXG_C_KEPT_BASES]));
            if (has_q) FXG_CHECK(&st, fxg_memcpy_d2h(st.ctx, st.h_out_qual, st.d_out_qual, ctr[FXG_C_KEPT_BASES]));
            FXG_CHECK(&st, fxg_sync(st.ctx));
        }

        /* ---- 4. write the kept records in input order, tally the report counters ---- */
        size_t opos = 0;
        for (size_t i = 0; i < n; ++i) {
            const fxh_rec *r = &st.rec[i];
            const uint32_t w = st.h_res[i], len = FXG_RES_LEN(w), rc_ = r->reads_count;
            tot->input_sequences++; tot->input_reads += rc_;
            tot->clip_input += rc_;
            if (FXG_RES_ADAPTER_ONLY(w)) tot->clip_adapter_only += rc_;
            switch (FXG_RES_REASON(w)) {
            case FXG_R_CLIP_TOO_SHORT: tot->clip_too_short += rc_; break;
            case FXG_R_CLIP_NO_ADAPTER: tot->clip_no_adapter += rc_; break;
            case FXG_R_CLIP_ADAPTER_FOUND: tot->clip_adapter_found += rc_; break;
            case FXG_R_CLIP_N: tot->clip_n += rc_; break;
            default: break;
            }
            if (!FXG_RES_KEEP(w)) continue;
            tot->output_sequences++; tot->output_reads += rc_;
            if (revcomp) {
                fxh_emit(fx, r, st.h_out_bases + opos, st.h_out_qual ? st.h_out_qual + opos : NULL, len, 0);
                opos += len;
            } else if (r->is_ascii) {
                fxh_emit(fx, r, (const uint8_t *)r->seq + fwd_start, (const uint8_t *)r->qual + fwd_start, len, 1);
            } else {
                fxh_emit(fx, r, (const uint8_t *)r->seq + fwd_start, st.h_qual + i * stride + fwd_start, len, 0);
            }
        }
        fx->num_input_sequences = tot->input_sequences; fx->num_input_reads = tot->input_reads;
        fx->num_output_sequences = tot->output_sequences; fx->num_output_reads = tot->output_reads;
    }
    if (have_err) {
        fxh_writer_flush(fx->writer);   /* every record before the bad one has been written, like the reference */
        errx(1, "%s", errmsg);
    }
    fxg_ctx_destroy(st.ctx);
    free(st.rec);
    return 0;
}
