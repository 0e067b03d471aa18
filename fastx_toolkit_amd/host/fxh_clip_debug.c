/* fxh_clip_debug.c -- fastx_clipper -D / -D -D: the reference's per-read dump of the aligner (fastx_clipper.cpp:270-275).
 *
 * -D prints every read's SequenceAlignmentResults (sequence_alignment.cpp:15-84), -D -D the whole score / origin / match matrix in front of it
 * (:169-230).  That is a developer's view of the CPU aligner's internal state: the alignment STRINGS and the MATRIX, neither of which the engine
 * has -- its DP keeps one row and a packed path summary (csrc/fxg_kernels.h).  A debug mode is not a hot path: with -D the tool runs the reference's
 * own record loop over the libfastx-compatible record API (fastx_read_next_record / fastx_write_record) with ONE aligner on the host, which exists
 * for this dump only -- the full matrix, the traceback with its strings, the never-shrinking matrix and the stale query tail of SURVEY N3 -- and
 * prints what the reference prints, byte for byte (tests/test_host_cli_emulated.py::test_clipper_debug_dump against the real libfastx).  Nothing
 * else in the product reaches this file; without -D the clipper is the GPU path and there is no CPU path.
 *
 * The dump goes to stdout like the reference's (std::cout).  With `-o FILE` that is all stdout carries; when the records go to stdout as well the
 * reference interleaves two buffered streams on one descriptor as their buffers happen to fill -- here the dump comes first, then the records. */
#include "fxh_priv.h"

enum { FROM_UPPER = 1, FROM_LEFT = 2, FROM_UPPER_LEFT = 3 };      /* sequence_alignment.h: DIRECTION */

typedef struct {
    size_t W, H;                           /* matrix_width() / matrix_height(): the longest query / target so far (resize_matrix never shrinks, :131-155) */
    float *score;                          /* [q * H + t] */
    unsigned char *origin;
    char *match;
    char *qbuf; size_t qcap, qlen;         /* _query_sequence's buffer: a shorter read leaves the tail of the longer ones behind */
    const char *target; size_t tlen;
    /* SequenceAlignmentResults */
    size_t query_start, query_end, target_start, target_end, gaps, neutral, matches, mismatches;
    float res_score;
    char *qal, *tal; size_t nal, alcap;    /* query_alignment / target_alignment */
    int fixed_fmt;                         /* std::cout after the first matrix cell: fixed, precision 1 -- for good (print_matrix never puts it back) */
} fxh_dbg_aligner;

static char dbg_q(const fxh_dbg_aligner *a, size_t i) { return a->qbuf[i]; }

static float dbg_pair(char q, char t)                         /* sequence_alignment.h:157-169 */
{
    if (q == 'N' && t == 'N') return 0.0f;
    if (q == 'N' || t == 'N') return 0.1f;
    return q == t ? 1.0f : -1.0f;
}

static void dbg_align(fxh_dbg_aligner *a, const char *query, size_t qn)
{
    /* set_sequences: std::string assignment -- the characters and the terminating NUL, into a buffer that only grows */
    if (a->qcap < qn + 1) {
        a->qcap = 2 * (qn + 1);
        a->qbuf = (char *)realloc(a->qbuf, a->qcap);
        if (!a->qbuf) err(1, "out of memory");
    }
    memcpy(a->qbuf, query, qn); a->qbuf[qn] = 0; a->qlen = qn;
    /* resize_matrix (:131-155) */
    if (!(a->W >= qn && a->H >= a->tlen)) {
        const size_t W = qn, H = a->tlen;
        float *s = (float *)calloc(W * H + 1, sizeof(float));
        unsigned char *o = (unsigned char *)calloc(W * H + 1, 1);
        char *m = (char *)calloc(W * H + 1, 1);
        if (!s || !o || !m) err(1, "out of memory");
        free(a->score); free(a->origin); free(a->match);
        a->score = s; a->origin = o; a->match = m; a->W = W; a->H = H;
    }
    const size_t W = a->W, H = a->H;
    for (size_t x = 0; x < W; ++x)                             /* populate_match_matrix (:157-162) */
        for (size_t y = 0; y < H; ++y) { const char q = dbg_q(a, x), t = a->target[y]; a->match[x * H + y] = (q == 'N' || t == 'N') ? 'N' : (q == t ? 'M' : 'x'); }
    /* populate_matrix (:365-428): borders query_border = 0, target_border[y] = y <= 3 ? 0 : -5 (y - 3); target_border[-1] reads 0 */
    float best = -1000000.0f;
    size_t bq = 0, bt = 0;
    unsigned char origin = FROM_LEFT;
    for (size_t q = 0; q < W; ++q)
        for (size_t t = 0; t < H; ++t) {
#define TB(y) ((y) <= 3 ? 0.0f : -5.0f * (float)((long)(y) - 3))
            const float up_prev = t == 0 ? 0.0f : a->score[q * H + (t - 1)];                          /* safe_score(q, t - 1): query_border[q] = 0 */
            const float left_prev = q == 0 ? TB(t) : a->score[(q - 1) * H + t];                        /* safe_score(q - 1, t): target_border[t] */
            const float ul_prev = q == 0 ? (t == 0 ? 0.0f : TB(t - 1)) : (t == 0 ? 0.0f : a->score[(q - 1) * H + (t - 1)]);
            float up = up_prev + -5.0f, left = left_prev + -5.0f;
            const float ul = ul_prev + dbg_pair(dbg_q(a, q), a->target[t]);
            if (t > 3 && t - 3 > q) left = -100000.0f;
            float sc = -100000000.0f;
            if (ul > sc) { sc = ul; origin = FROM_UPPER_LEFT; }
            if (up > sc) { sc = up; origin = FROM_UPPER; }
            if (left > sc) { sc = left; origin = FROM_LEFT; }
            a->score[q * H + t] = sc; a->origin[q * H + t] = origin;
            if (sc > best) { bq = q; bt = t; best = sc; }
        }
    /* find_optimal_alignment_from_point (:496-604) from the highest cell; the heuristics behind it all keep this result (:606-650) */
    a->nal = 0; a->gaps = a->neutral = a->matches = a->mismatches = 0; a->res_score = 0.0f;
    a->query_start = a->target_start = 0;
    a->query_end = bq; a->target_end = bt;
    long qi = (long)bq, ti = (long)bt;
    while (qi >= 0 && ti >= 0) {
        if (a->nal + 2 > a->alcap) {
            a->alcap = 2 * a->alcap + 64;
            a->qal = (char *)realloc(a->qal, a->alcap); a->tal = (char *)realloc(a->tal, a->alcap);
            if (!a->qal || !a->tal) err(1, "out of memory");
        }
        const char qc = dbg_q(a, (size_t)qi), tc = a->target[ti];
        a->query_start = (size_t)qi; a->target_start = (size_t)ti;
        switch (a->origin[(size_t)qi * H + (size_t)ti]) {
        case FROM_LEFT: a->tal[a->nal] = '-'; a->qal[a->nal++] = qc; a->gaps++; a->res_score += -5.0f; qi--; break;
        case FROM_UPPER_LEFT:
            a->tal[a->nal] = tc; a->qal[a->nal++] = qc;
            switch (a->match[(size_t)qi * H + (size_t)ti]) {
            case 'N': a->neutral++; a->res_score += 0.1f; break;
            case 'M': a->matches++; a->res_score += 1.0f; break;
            default: a->mismatches++; a->res_score += -1.0f; break;
            }
            qi--; ti--;
            break;
        default: a->tal[a->nal] = tc; a->qal[a->nal++] = '-'; a->gaps++; a->res_score += -5.0f; ti--; break;
        }
    }
    for (size_t i = 0; i < a->nal / 2; ++i) {                  /* std::reverse of both strings */
        char c = a->qal[i]; a->qal[i] = a->qal[a->nal - 1 - i]; a->qal[a->nal - 1 - i] = c;
        c = a->tal[i]; a->tal[i] = a->tal[a->nal - 1 - i]; a->tal[a->nal - 1 - i] = c;
    }
}

static void dbg_float(const fxh_dbg_aligner *a, float v, int width, int left)      /* operator<<(float) under the stream's current flags */
{
    char b[64];
    snprintf(b, sizeof b, a->fixed_fmt ? "%.1f" : "%g", (double)v);
    if (width) printf(left ? "%-*s" : "%*s", width, b); else fputs(b, stdout);
}
static void dbg_spaces(size_t n) { for (size_t i = 0; i < n; ++i) putchar(' '); }

static void dbg_print_matrix(fxh_dbg_aligner *a)               /* sequence_alignment.cpp:169-230 */
{
    const size_t W = a->W, H = a->H;
    puts("Score-Matrix:");
    printf("%-2s%-7s", "-", "-");
    for (size_t q = 0; q < W; ++q) { putchar(dbg_q(a, q)); dbg_spaces(8); }       /* setw(9) << left << char */
    putchar('\n');
    printf("%-2s%-7s", "-", "-");
    for (size_t q = 0; q < W; ++q) dbg_float(a, 0.0f, 9, 1);                        /* query_border */
    putchar('\n');
    for (size_t t = 0; t < H; ++t) {
        putchar(a->target[t]); putchar(' ');
        dbg_float(a, TB(t), 6, 0); putchar(' ');
        for (size_t q = 0; q < W; ++q) {
            const unsigned char o = a->origin[q * H + t];
            putchar(a->match[q * H + t]);
            putchar(o == FROM_UPPER ? '|' : o == FROM_LEFT ? '-' : o == FROM_UPPER_LEFT ? '\\' : '*');
            a->fixed_fmt = 1;                                    /* << fixed << setprecision(1): sticks */
            dbg_float(a, a->score[q * H + t], 7, 1);
        }
        putchar('\n');
    }
}

static void dbg_print_results(const fxh_dbg_aligner *a)        /* sequence_alignment.cpp:15-84 */
{
    const size_t qs = a->query_start, qe = a->query_end, ts = a->target_start, te = a->target_end;
    fputs("Query-Alingment = ", stdout); fwrite(a->qal, 1, a->nal, stdout); putchar('\n');
    fputs("target-Alingment= ", stdout); fwrite(a->tal, 1, a->nal, stdout); putchar('\n');
    puts("Alignment NOT found");                               /* (alignment_found is never set) */
    fputs("Score = ", stdout); dbg_float(a, a->res_score, 0, 0);
    printf(" (%zu matches, %zu neutral-matches, %zu mismatches, %zu gaps) \n", a->matches, a->neutral, a->mismatches, a->gaps);
    fputs("Query = ", stdout); fwrite(a->qbuf, 1, a->qlen, stdout); printf("(qsize %zu qstart %zu qend %zu\n", a->qlen, qs, qe);
    fputs("Target= ", stdout); fwrite(a->target, 1, a->tlen, stdout); printf("(tsize %zu tstart %zu tend %zu\n", a->tlen, ts, te);
    putchar('\n');
    const size_t delta = ts > qs ? ts : qs;
    if (delta - qs > 0) dbg_spaces(delta - qs - 1);
    if (qs > 0) fwrite(a->qbuf, 1, qs - 1 < a->qlen ? qs - 1 : a->qlen, stdout);                 /* substr(0, query_start - 1) */
    putchar('('); fwrite(a->qal, 1, a->nal, stdout); putchar(')');
    if (qe < a->qlen) fwrite(a->qbuf + qe + 1, 1, a->qlen - (qe + 1), stdout);
    putchar('\n');
    if (delta > 0) dbg_spaces(delta - 1);
    putchar('(');
    for (size_t i = 0; i < a->nal; ++i) putchar(a->qal[i] == a->tal[i] ? '*' : '|');
    putchar(')'); putchar('\n');
    if (delta - ts > 0) dbg_spaces(delta - ts);
    if (ts > 0) fwrite(a->target, 1, ts - 1 < a->tlen ? ts - 1 : a->tlen, stdout);
    putchar('('); fwrite(a->tal, 1, a->nal, stdout); putchar(')');
    if (te < a->tlen) fwrite(a->target + te + 1, 1, a->tlen - (te + 1), stdout);
    putchar('\n');
}

static int dbg_cutoff(const fxh_dbg_aligner *a, int min_adapter)           /* fastx_clipper.cpp:192-240 (size_t arithmetic) */
{
    const size_t sz = a->neutral + a->matches + a->mismatches + a->gaps;
    if (sz == 0) return -1;
    if (min_adapter > 0 && sz < (size_t)min_adapter) return -1;
    if (a->query_end == a->qlen - 1 && a->mismatches == 0) return (int)a->query_start;
    if (sz > 5 && a->target_start == 0 && (a->matches * 100 / sz) >= 75) return (int)a->query_start;
    if (sz > 11 && (a->matches * 100 / sz) >= 80) return (int)a->query_start;
    if (a->query_end >= a->qlen - 2 && sz <= 5 && a->matches >= 3) return (int)a->query_start;
    return -1;
}

/* the reference's main loop (fastx_clipper.cpp:257-320) with the dump in it; `keep_delta` as the tool folded it (-d + strlen(adapter), :153-154) */
void fxh_clipper_debug_run(FASTX *fx, const fxg_params *p, int level, fxh_totals *tot)
{
    fxh_dbg_aligner a;
    memset(&a, 0, sizeof a);
    memset(tot, 0, sizeof *tot);
    a.target = p->adapter; a.tlen = strlen(p->adapter);
    const int only_clipped = (p->clip_flags & FXG_CLIP_DISCARD_NON_CLIPPED) != 0, only_nonclipped = (p->clip_flags & FXG_CLIP_DISCARD_CLIPPED) != 0;
    const int keep_n = (p->clip_flags & FXG_CLIP_KEEP_N) != 0, adapter_only = (p->clip_flags & FXG_CLIP_ADAPTER_ONLY) != 0;
    while (fastx_read_next_record(fx)) {
        const unsigned reads = (unsigned)get_reads_count(fx);
        dbg_align(&a, fx->nucleotides, strlen(fx->nucleotides));
        if (level > 1) dbg_print_matrix(&a);
        if (level > 0) dbg_print_results(&a);
        tot->clip_input += reads;
        int i = dbg_cutoff(&a, p->clip_min_adapter_len);
        if (i != -1 && i > 0) { i += p->clip_keep_delta; fx->nucleotides[i] = 0; }                   /* :282-286 (a cut beyond the read's end changes nothing: the NUL lands behind it) */
        if (i == 0) { tot->clip_adapter_only += reads; if (adapter_only) fastx_write_record(fx); continue; }
        if (strlen(fx->nucleotides) < p->clip_min_len) { tot->clip_too_short += reads; continue; }
        if (i == -1 && only_clipped) { tot->clip_no_adapter += reads; continue; }
        if (i > 0 && only_nonclipped) { tot->clip_adapter_found += reads; continue; }
        if (!keep_n && strchr(fx->nucleotides, 'N') != NULL) { tot->clip_n += reads; continue; }
        if (!adapter_only) fastx_write_record(fx);
    }
    tot->input_sequences = num_input_sequences(fx); tot->input_reads = num_input_reads(fx);
    tot->output_sequences = num_output_sequences(fx); tot->output_reads = num_output_reads(fx);
    fflush(stdout);
    free(a.score); free(a.origin); free(a.match); free(a.qbuf); free(a.qal); free(a.tal);
}
