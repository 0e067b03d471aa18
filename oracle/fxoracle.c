/*
 * fxoracle.c -- CPU oracle for the fastx_toolkit hot path.   *** TEST INFRASTRUCTURE ONLY ***
 *
 * Plain-C restatement of the reference algorithms; see fxoracle.h for status and rules of use.
 * The code deliberately follows the reference's *formulation* (backward scan, histogram walk,
 * full DP matrix + traceback) rather than the closed forms the HIP kernels use, so that the
 * parity tests compare two independent derivations.
 */
#define _GNU_SOURCE
#include "fxoracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* synthetic generator -- SURVEY.md section 8(d) "Synthetic input spec"                        */
/* ------------------------------------------------------------------------------------------ */

uint64_t fxo_splitmix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static const char FXO_ADAPTER13[] = "AGATCGGAAGAGC";

void fxo_synth_read(uint64_t seed, uint64_t r, uint32_t L, int with_adapter, uint8_t *bases, uint8_t *qual)
{
    static const char acgt[] = "ACGT";
    const uint64_t key = fxo_splitmix(seed ^ (r * 0x9E3779B97F4A7C15ull));
#define VAL(stream, p) fxo_splitmix(key + ((uint64_t)(stream) << 32) + (uint64_t)(p))
    for (uint32_t p = 0; p < L; ++p) {
        uint64_t u = VAL(0, p);
        uint8_t b = (uint8_t)acgt[u & 3];
        if ((u >> 8) % 200 == 0) b = 'N';
        bases[p] = b;
    }
    if (qual) {
        const uint64_t d = VAL(1, 0) % (L + L / 3);
        const int noisy = (VAL(1, 1) % 4) == 0;
        for (uint32_t p = 0; p < L; ++p) {
            uint64_t u = VAL(2, p);
            uint32_t lo = 2 + (uint32_t)(u % 18);
            uint32_t hi = 25 + (uint32_t)(u % 16);
            int dip = ((u >> 16) % (noisy ? 4 : 16)) == 0;
            uint32_t q = (p < d) ? (dip ? lo : hi) : lo;
            qual[p] = (uint8_t)(33 + q);
        }
    }
    if (with_adapter) {
        const uint64_t a = VAL(3, 0);
        if (a % 2 == 0) {
            char ad[14];
            memcpy(ad, FXO_ADAPTER13, 14);
            uint32_t pos = (uint32_t)((a >> 8) % (L + 1));
            if ((a >> 40) % 8 == 0) ad[(a >> 44) % 13] = acgt[(a >> 48) & 3];
            for (uint32_t k = 0; k < 13 && pos + k < L; ++k) bases[pos + k] = (uint8_t)ad[k];
        }
    }
#undef VAL
}

void fxo_synth_batch(uint64_t seed, uint64_t first, uint64_t n, uint32_t L, int with_adapter,
                     uint8_t *bases, uint8_t *qual, uint32_t stride)
{
    for (uint64_t i = 0; i < n; ++i)
        fxo_synth_read(seed, first + i, L, with_adapter, bases + i * stride, qual ? qual + i * stride : NULL);
}

size_t fxo_synth_fastq(uint64_t seed, uint64_t first, uint64_t n, uint32_t L, int with_adapter, char *out)
{
    size_t w = 0;
    uint8_t *b = (uint8_t *)malloc(2 * (size_t)L + 2);
    uint8_t *q = b + L + 1;
    char hdr[96];
    for (uint64_t i = 0; i < n; ++i) {
        int hl = snprintf(hdr, sizeof hdr, "@SYN.%llu.%llu\n", (unsigned long long)seed, (unsigned long long)(first + i));
        if (out) {
            fxo_synth_read(seed, first + i, L, with_adapter, b, q);
            memcpy(out + w, hdr, (size_t)hl); w += (size_t)hl;
            memcpy(out + w, b, L); w += L; out[w++] = '\n';
            out[w++] = '+'; out[w++] = '\n';
            memcpy(out + w, q, L); w += L; out[w++] = '\n';
        } else {
            w += (size_t)hl + 2 * (size_t)L + 4;
        }
    }
    free(b);
    return w;
}

/* ------------------------------------------------------------------------------------------ */
/* fastq_quality_trimmer body -- src/fastq_quality_trimmer/fastq_quality_trimmer.c:91-103      */
/* ------------------------------------------------------------------------------------------ */
int fxo_qtrim_read(const uint8_t *qual, int len, int qoffset, int threshold, int min_len, int *new_len)
{
    int i;
    /* "Scan each sequence - backwards": NUL out bases while quality < threshold, stop at first >= */
    for (i = len - 1; i >= 0; --i) {
        int q = (int)(signed char)qual[i] - qoffset;   /* fastx.c:127, char is signed on x86-64 */
        if (q < threshold) continue;
        break;
    }
    *new_len = i + 1;
    return (i >= 0 && i + 1 >= min_len) ? 1 : 0;       /* :101 */
}

/* ------------------------------------------------------------------------------------------ */
/* fastq_quality_filter body -- src/fastq_quality_filter/fastq_quality_filter.c:78-129,150-156 */
/* ------------------------------------------------------------------------------------------ */
#define FXO_MINQ (-15)
#define FXO_QRANGE 108 /* fastx.h:28-30: MAX(93) - MIN(-15) */

static int fxo_index_of_nth(const int *array, int array_size, int n)
{
    int pos = 0;
    while (pos < array_size && array[pos] == 0) pos++;
    if (pos == array_size) return -1; /* reference: errx "bug: got empty array" (note N2) */
    while (n > 0) {
        if (array[pos] > n) break;
        n -= array[pos];
        pos++;
        while (pos < array_size && array[pos] == 0) pos++;
        if (pos >= array_size) break;  /* reference reads one slot past the array here; result is array_size */
    }
    return pos;
}

int fxo_qfilter_read(const uint8_t *qual, int len, int qoffset, int min_quality, int min_percent)
{
    int hist[FXO_QRANGE + 2];
    int count = 0;
    memset(hist, 0, sizeof hist);
    for (int i = 0; i < len; ++i) {
        int q = (int)(signed char)qual[i] - qoffset;
        int slot = q - FXO_MINQ;
        if (slot < 0) slot = 0;
        if (slot > FXO_QRANGE) slot = FXO_QRANGE; /* q==93 lands one past the reference's array (N2) */
        hist[slot]++;
        count++;
    }
    int n = count * (100 - min_percent) / 100;          /* :123 int arithmetic */
    int pos = fxo_index_of_nth(hist, FXO_QRANGE, n);
    if (pos < 0) return 0;
    int value = pos + FXO_MINQ;
    return value >= min_quality;                        /* :155 */
}

/* ------------------------------------------------------------------------------------------ */
/* HalfLocalSequenceAlignment -- src/libfastx/sequence_alignment.cpp:113-129,340-428,496-650   */
/* ------------------------------------------------------------------------------------------ */
enum { FROM_UPPER = 1, FROM_LEFT = 2, FROM_UPPER_LEFT = 3 };

struct fxo_aligner {
    size_t width, height;      /* matrix never shrinks (sequence_alignment.cpp:135-136) */
    float *score;              /* [width][height] */
    uint8_t *origin;
    char *qbuf;                /* emulates the std::string buffer of _query_sequence incl. stale tail (N3) */
    size_t qcap;
};

fxo_aligner *fxo_aligner_new(void) { return (fxo_aligner *)calloc(1, sizeof(fxo_aligner)); }

void fxo_aligner_free(fxo_aligner *a)
{
    if (!a) return;
    free(a->score); free(a->origin); free(a->qbuf); free(a);
}

static float fxo_pair_score(char q, char t)   /* sequence_alignment.h:157-169 */
{
    if (q == 'N' && t == 'N') return 0.0f;
    if (q == 'N' || t == 'N') return 0.1f;
    return (q == t) ? 1.0f : -1.0f;
}

void fxo_align(fxo_aligner *a, const char *query, int qn, const char *target, int tn, fxo_align_res *res)
{
    const float gap = -5.0f;
    /* set_sequences: assignment into the existing string buffer; a longer query reallocates (fresh buffer) */
    if ((size_t)qn + 1 > a->qcap) {
        size_t ncap = a->qcap * 2 > (size_t)qn + 1 ? a->qcap * 2 : (size_t)qn + 1;
        char *nb = (char *)calloc(ncap, 1);
        free(a->qbuf); a->qbuf = nb; a->qcap = ncap;
    }
    memcpy(a->qbuf, query, (size_t)qn);
    a->qbuf[qn] = '\0';
    /* resize_matrix: grow only */
    if (!(a->width >= (size_t)qn && a->height >= (size_t)tn)) {
        size_t nw = a->width > (size_t)qn ? a->width : (size_t)qn;   /* vectors resized to (width,height) args, */
        size_t nh = a->height > (size_t)tn ? a->height : (size_t)tn; /* but both only ever grow for a fixed adapter */
        /* the reference resizes to exactly (qn, tn) when either is larger; with a constant adapter
         * height is constant, so width := qn (>= old width is implied by the guard). */
        nw = (size_t)qn > a->width ? (size_t)qn : a->width;
        free(a->score); free(a->origin);
        a->score = (float *)malloc(nw * nh * sizeof(float));
        a->origin = (uint8_t *)malloc(nw * nh);
        a->width = nw; a->height = nh;
    }
    const size_t W = a->width, H = a->height;
    const char *Q = a->qbuf;       /* read up to W, i.e. possibly past qn into the stale tail */
    float *S = a->score;
    uint8_t *O = a->origin;

    /* populate_matrix (:365-428); borders from reset_matrix (:340-363) */
    float best = -1000000.0f;
    size_t best_q = 0, best_t = 0;
    for (size_t q = 0; q < W; ++q) {
        for (size_t t = 0; t < H; ++t) {
            float s_up_src   = (t == 0) ? 0.0f : S[q * H + (t - 1)];                       /* query_border[q] = 0 */
            float s_left_src = (q == 0) ? ((t <= 3) ? 0.0f : gap * (float)((long)t - 3)) : S[(q - 1) * H + t];
            float s_ul_src;
            if (q == 0 && t == 0) s_ul_src = 0.0f;      /* target_border[-1]: OOB read, 0.0 with glibc (N1) */
            else if (q == 0) s_ul_src = ((t - 1) <= 3) ? 0.0f : gap * (float)((long)(t - 1) - 3);
            else if (t == 0) s_ul_src = 0.0f;           /* query_border[q-1] */
            else s_ul_src = S[(q - 1) * H + (t - 1)];
            float up = s_up_src + gap;
            float left = s_left_src + gap;
            float ul = s_ul_src + fxo_pair_score(Q[q], target[t]);
            if (t > 3 && t - 3 > q) left = -100000.0f;
            float sc = -100000000.0f;
            uint8_t org = FROM_LEFT;
            if (ul > sc) { sc = ul; org = FROM_UPPER_LEFT; }
            if (up > sc) { sc = up; org = FROM_UPPER; }
            if (left > sc) { sc = left; org = FROM_LEFT; }
            S[q * H + t] = sc;
            O[q * H + t] = org;
            if (sc > best) { best = sc; best_q = q; best_t = t; }
        }
    }

    /* find_optimal_alignment_from_point (:496-604); the heuristics in :606-650 all keep this result */
    memset(res, 0, sizeof *res);
    res->score = best;
    long qi = (long)best_q, ti = (long)best_t;
    res->query_end = qi; res->target_end = ti;
    while (qi >= 0 && ti >= 0) {
        res->query_start = qi; res->target_start = ti;
        switch (O[(size_t)qi * H + (size_t)ti]) {
        case FROM_LEFT: res->gaps++; qi--; break;
        case FROM_UPPER: res->gaps++; ti--; break;
        default: {
            char qc = Q[qi], tc = target[ti];
            if (qc == 'N' || tc == 'N') res->neutral_matches++;     /* match_value, sequence_alignment.h:125-131 */
            else if (qc == tc) res->matches++;
            else res->mismatches++;
            qi--; ti--;
        } }
    }
    res->query_size = qn;
    res->target_size = tn;
}

/* src/fastx_clipper/fastx_clipper.cpp:192-240, all arithmetic in the reference is size_t */
int fxo_adapter_cutoff_index(const fxo_align_res *r, int min_adapter_len)
{
    size_t qsize = (size_t)r->query_size, qend = (size_t)r->query_end;
    size_t matches = (size_t)r->matches, mism = (size_t)r->mismatches;
    int alignment_size = (int)(r->neutral_matches + r->matches + r->mismatches + r->gaps);
    if (alignment_size == 0) return -1;
    if (min_adapter_len > 0 && alignment_size < min_adapter_len) return -1;
    if (qend == qsize - 1 && mism == 0) return (int)r->query_start;
    if (alignment_size > 5 && r->target_start == 0 && (matches * 100 / (size_t)alignment_size) >= 75) return (int)r->query_start;
    if (alignment_size > 11 && (matches * 100 / (size_t)alignment_size) >= 80) return (int)r->query_start;
    if (qend >= qsize - 2 && alignment_size <= 5 && matches >= 3) return (int)r->query_start;
    return -1;
}

/* ------------------------------------------------------------------------------------------ */
/* batch pipeline                                                                              */
/* ------------------------------------------------------------------------------------------ */

static int fxo_comp(uint8_t c)  /* fastx_reverse_complement.c:43-72 */
{
    switch (c) {
    case 'N': return 'N'; case 'n': return 'n';
    case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G';
    case 'a': return 't'; case 't': return 'a'; case 'g': return 'c'; case 'c': return 'g';
    default: return -1;
    }
}

int fxo_run_pipeline_h(const fxo_batch *in, const fxo_params *p, fxo_out *out, fxo_aligner *shared)
{
    const uint32_t st = p->stages;
    const int group_a = (st & (FXO_STAGE_CLIP | FXO_STAGE_QTRIM | FXO_STAGE_QFILTER)) != 0;
    const int group_b = (st & (FXO_STAGE_REVCOMP | FXO_STAGE_FTRIM | FXO_STAGE_FTRIM_END)) != 0;
    const int group_c = (st & FXO_STAGE_MASK) != 0, group_d = (st & FXO_STAGE_ARTIFACTS) != 0, group_e = (st & FXO_STAGE_NFILTER) != 0;
    if (group_a + group_b + group_c + group_d + group_e != 1) return -1;
    if (group_c && !in->qual) return -1;
    if ((st & FXO_STAGE_FTRIM) && (st & FXO_STAGE_FTRIM_END)) return -1;   /* fastx_trimmer.c:112-113 */
    if ((st & (FXO_STAGE_QTRIM | FXO_STAGE_QFILTER)) && !in->qual) return -1;

    memset(out->counters, 0, sizeof out->counters);
    /* shared: the caller's aligner, i.e. one fastx_clipper process working through several batches (N3 history) */
    fxo_aligner *al = shared ? shared : ((st & FXO_STAGE_CLIP) ? fxo_aligner_new() : NULL);
    const int alen = (int)strnlen(p->adapter, sizeof p->adapter);
    uint64_t kept = 0, obytes = 0;
    uint8_t *tmpb = (uint8_t *)malloc(70000), *tmpq = (uint8_t *)malloc(70000);
    int rc = 0;

    for (uint64_t r = 0; r < in->n; ++r) {
        const uint8_t *b = in->bases + r * in->stride;
        const uint8_t *q = in->qual ? in->qual + r * in->stride : NULL;
        int len = in->len ? in->len[r] : (int)in->fixed_len;
        int keep = 1, reason = FXO_R_KEPT, clipped = 0, aonly = 0;
        int start = 0;       /* forward view: output = b[start .. start+len) */
        int reversed = 0;
        out->counters[FXO_C_INPUT]++;

        if (st & FXO_STAGE_CLIP) {               /* fastx_clipper.cpp:257-320 */
            fxo_align_res ar;
            fxo_align(al, (const char *)b, len, p->adapter, alen, &ar);
            int i = fxo_adapter_cutoff_index(&ar, p->clip_min_adapter_len);
            int cur = len;
            if (i != -1 && i > 0) {
                i += p->clip_keep_delta;
                if (i < cur) cur = i;            /* nucleotides[i] = 0 */
                clipped = 1;
            }
            if (i == 0) {
                out->counters[FXO_C_CLIP_ADAPTER_ONLY]++; aonly = 1;
                if (p->clip_flags & FXO_CLIP_ADAPTER_ONLY) { keep = 1; cur = len; }
                else { keep = 0; reason = FXO_R_CLIP_ADAPTER_ONLY; }
            } else if ((unsigned)cur < p->clip_min_len) {
                out->counters[FXO_C_CLIP_TOO_SHORT]++; keep = 0; reason = FXO_R_CLIP_TOO_SHORT;
            } else if (i == -1 && (p->clip_flags & FXO_CLIP_DISCARD_NON_CLIPPED)) {
                out->counters[FXO_C_CLIP_NO_ADAPTER]++; keep = 0; reason = FXO_R_CLIP_NO_ADAPTER;
            } else if (i > 0 && (p->clip_flags & FXO_CLIP_DISCARD_CLIPPED)) {
                out->counters[FXO_C_CLIP_ADAPTER_FOUND]++; keep = 0; reason = FXO_R_CLIP_ADAPTER_FOUND;
            } else if (!(p->clip_flags & FXO_CLIP_KEEP_N) && memchr(b, 'N', (size_t)cur) != NULL) {
                out->counters[FXO_C_CLIP_N]++; keep = 0; reason = FXO_R_CLIP_N;
            } else if (p->clip_flags & FXO_CLIP_ADAPTER_ONLY) {
                keep = 0; reason = FXO_R_CLIP_K_MODE;
            }
            len = cur;
            if (keep) out->counters[FXO_C_CLIP_OUT]++;
        }
        if (keep && (st & FXO_STAGE_QTRIM)) {
            int nl;
            keep = fxo_qtrim_read(q, len, p->qoffset, p->qt_threshold, p->qt_min_len, &nl);
            if (!keep) { reason = FXO_R_QTRIM; out->counters[FXO_C_QTRIM_DROPPED]++; }
            else out->counters[FXO_C_QTRIM_OUT]++;
            len = nl;
        }
        if (keep && (st & FXO_STAGE_QFILTER)) {
            keep = fxo_qfilter_read(q, len, p->qoffset, p->qf_min_quality, p->qf_min_percent);
            if (!keep) { reason = FXO_R_QFILTER; out->counters[FXO_C_QFILTER_DROPPED]++; }
        }
        if (st & FXO_STAGE_ARTIFACTS) {          /* fastx_artifacts_filter.c:56-112 */
            int cnt[5] = {0, 0, 0, 0, 0}, total = 0;
            for (int k = 0; k < len; ++k) {
                total++;
                switch (b[k]) {
                case 'A': cnt[0]++; break; case 'C': cnt[1]++; break; case 'G': cnt[2]++; break; case 'T': cnt[3]++; break; case 'N': cnt[4]++; break;
                default: rc = -2; goto done;
                }
            }
            if (cnt[0] >= total - 3 || cnt[1] >= total - 3 || cnt[2] >= total - 3 || cnt[3] >= total - 3) {
                keep = 0; reason = FXO_R_ARTIFACT; out->counters[FXO_C_ARTIFACT_DROPPED]++;
            }
        }
        if ((st & FXO_STAGE_NFILTER) && !p->nf_keep_n && memchr(b, 'N', (size_t)len) != NULL) {   /* fastq_to_fasta.c:80-81 */
            keep = 0; reason = FXO_R_HAS_N;
        }
        int masked = 0;
        if (st & FXO_STAGE_MASK) {               /* fastq_masker.c:92-103: every read is written, low-quality bases replaced */
            for (int k = 0; k < len; ++k) {
                int qv = (int)(signed char)q[k] - p->qoffset;
                if (qv < p->mask_min_quality) { masked = 1; out->counters[FXO_C_MASKED_NT]++; }
            }
            if (masked) out->counters[FXO_C_MASKED_READS]++;
        }
        if (st & FXO_STAGE_REVCOMP) reversed = 1;
        if (keep && (st & FXO_STAGE_FTRIM)) {     /* fastx_trimmer.c:122-134 */
            if (p->ft_last != 0 && p->ft_last < len) len = p->ft_last;
            if (p->ft_first != 1) {
                if (len < p->ft_first) { keep = 0; reason = FXO_R_FTRIM; out->counters[FXO_C_FTRIM_DROPPED]++; }
                else { start = p->ft_first - 1; len = len - p->ft_first + 1; }
            }
        }
        if (keep && (st & FXO_STAGE_FTRIM_END)) { /* fastx_trimmer.c:136-144 */
            if ((unsigned)len <= p->ft_trim_end) keep = 0;
            else {
                unsigned i = (unsigned)len - p->ft_trim_end;
                if (i < p->ft_min_len) keep = 0; else len = (int)i;
            }
            if (!keep) { reason = FXO_R_FTRIM; out->counters[FXO_C_FTRIM_DROPPED]++; }
        }

        out->res[r] = ((uint32_t)len & 0xFFFFu) | ((uint32_t)keep << 16) | ((uint32_t)reason << 17) | ((uint32_t)clipped << 21) | ((uint32_t)aonly << 22);
        if (!keep) continue;

        /* materialise the kept read (a3: the writer emits strlen(nucleotides) bases and as many qualities) */
        const int full = in->len ? in->len[r] : (int)in->fixed_len;
        const uint8_t *sb = b, *sq = q;
        if (reversed) {                           /* fastx_reverse_complement.c:74-104 */
            for (int k = 0; k < full; ++k) {
                int c = fxo_comp(b[full - 1 - k]);
                if (c < 0) { rc = -2; goto done; }
                tmpb[k] = (uint8_t)c;
                if (q) tmpq[k] = q[full - 1 - k];
            }
            sb = tmpb; sq = q ? tmpq : NULL;
        }
        memcpy(out->out_bases + obytes, sb + start, (size_t)len);
        if (masked)
            for (int k = 0; k < len; ++k)
                if ((int)(signed char)q[k] - p->qoffset < p->mask_min_quality) out->out_bases[obytes + k] = (uint8_t)p->mask_char;
        if (q && out->out_qual) memcpy(out->out_qual + obytes, sq + start, (size_t)len);
        if (out->out_len) out->out_len[kept] = (uint16_t)len;
        if (out->kept_index) out->kept_index[kept] = (uint32_t)r;
        obytes += (uint64_t)len;
        kept++;
    }
    out->counters[FXO_C_KEPT] = kept;
    out->counters[FXO_C_KEPT_BASES] = obytes;
done:
    free(tmpb); free(tmpq);
    if (!shared) fxo_aligner_free(al);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* FASTQ text reader/writer -- src/libfastx/fastx.c:314-404 (rules R1-R9), :440-473            */
/* ------------------------------------------------------------------------------------------ */

/* one fgets()+chomp() step: returns 0 at EOF; *s/*n get the chomped line (cut at first CR or LF, chomp.c:36-41) */
static int fxo_next_line(const char *text, size_t text_len, size_t *pos, const char **s, size_t *n)
{
    if (*pos >= text_len) return 0;
    const char *p = text + *pos;
    const char *nl = (const char *)memchr(p, '\n', text_len - *pos);
    size_t raw = nl ? (size_t)(nl - p) + 1 : text_len - *pos;
    *pos += raw;
    size_t k = 0;
    while (k < raw && p[k] != '\r' && p[k] != '\n') k++;
    *s = p; *n = k;
    return 1;
}

int64_t fxo_parse_fastq(const char *text, size_t text_len, int qoffset, uint64_t max_reads, uint32_t stride,
                        uint8_t *bases, uint8_t *qual, uint16_t *len,
                        uint64_t *name_off, uint32_t *name_len, uint64_t *name2_off, uint32_t *name2_len)
{
    size_t pos = 0;
    int64_t line = 0;
    uint64_t r = 0;
    if (text_len == 0 || text[0] != '@') return -1;                     /* R1 (FASTQ only here) */
    while (r < max_reads) {
        const char *s; size_t n;
        line++;
        if (!fxo_next_line(text, text_len, &pos, &s, &n)) break;        /* EOF at a record boundary */
        /* prefix byte is the raw first byte of the line, even if it is the newline itself (R3) */
        if (s[0] != '@') return -line;
        name_off[r] = (uint64_t)(s - text) + 1; name_len[r] = (uint32_t)(n ? n - 1 : 0);
        line++;
        if (!fxo_next_line(text, text_len, &pos, &s, &n)) return -line;
        if (n == 0 || n > stride || n > 65535) return -line;            /* R4 */
        for (size_t i = 0; i < n; ++i) {
            char c = s[i];
            if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N')) return -line;
        }
        memcpy(bases + r * stride, s, n);
        len[r] = (uint16_t)n;
        size_t seqn = n;
        line++;
        if (!fxo_next_line(text, text_len, &pos, &s, &n)) return -line; /* R5: not validated, first byte dropped */
        name2_off[r] = (uint64_t)(s - text) + 1;
        {   /* chomp applies to name2 (= buffer+1): if the raw first byte is CR/LF the name is whatever follows
               in the buffer; only the common cases (first byte '+') are modelled */
            name2_len[r] = (uint32_t)(n ? n - 1 : 0);
        }
        line++;
        if (!fxo_next_line(text, text_len, &pos, &s, &n)) return -line;
        if (n != seqn) return -line;                                    /* R6: numeric qualities not modelled */
        for (size_t i = 0; i < n; ++i) {                                /* R7 */
            int q = (int)(signed char)s[i] - qoffset;
            if (q < -15 || q > 93) return -line;
        }
        memcpy(qual + r * stride, s, n);
        r++;
    }
    return (int64_t)r;
}

size_t fxo_format_fastq(const char *text, const uint64_t *name_off, const uint32_t *name_len,
                        const uint64_t *name2_off, const uint32_t *name2_len,
                        const uint8_t *out_bases, const uint8_t *out_qual, const uint16_t *out_len,
                        const uint32_t *kept_index, uint64_t kept, char *dst)
{
    size_t w = 0, src = 0;
    for (uint64_t k = 0; k < kept; ++k) {
        uint32_t r = kept_index[k];
        uint16_t l = out_len[k];
        dst[w++] = '@';
        memcpy(dst + w, text + name_off[r], name_len[r]); w += name_len[r];
        dst[w++] = '\n';
        memcpy(dst + w, out_bases + src, l); w += l;
        dst[w++] = '\n';
        dst[w++] = '+';
        memcpy(dst + w, text + name2_off[r], name2_len[r]); w += name2_len[r];
        dst[w++] = '\n';
        memcpy(dst + w, out_qual + src, l); w += l;   /* R8: q+Q round-trips to the input byte */
        dst[w++] = '\n';
        src += l;
    }
    return w;
}

int fxo_run_pipeline(const fxo_batch *in, const fxo_params *p, fxo_out *out) { return fxo_run_pipeline_h(in, p, out, NULL); }

/* ------------------------------------------------------------------------------------------ */
/* fastx_quality_stats -- src/fastx_quality_stats/fastx_quality_stats.c                         */
/* ------------------------------------------------------------------------------------------ */
struct fxo_nucdata {            /* struct nucleotide_data, :115-127 */
    int min, max, count;
    unsigned long long sum;
    int values[FXO_QS_RANGE + 1];   /* +1: quality 93 indexes one past the reference's array (N2); parity is defined below that */
};
struct fxo_qstats {
    struct fxo_nucdata (*cycles)[6];   /* [column][ALL,A,C,G,T,N], :130-133 */
    int ncols;
    int sequences;
};

static void fxo_qstats_grow(fxo_qstats *s, int ncols)
{
    if (ncols <= s->ncols) return;
    s->cycles = (struct fxo_nucdata (*)[6])realloc(s->cycles, (size_t)ncols * sizeof *s->cycles);
    for (int i = s->ncols; i < ncols; ++i)
        for (int j = 0; j < 6; ++j) {                     /* init_values, :138-163 */
            memset(&s->cycles[i][j], 0, sizeof s->cycles[i][j]);
            s->cycles[i][j].min = 100; s->cycles[i][j].max = -100;
        }
    s->ncols = ncols;
}

fxo_qstats *fxo_qstats_new(void) { return (fxo_qstats *)calloc(1, sizeof(fxo_qstats)); }
void fxo_qstats_free(fxo_qstats *s) { if (s) { free(s->cycles); free(s); } }

static int fxo_nuc_index(int c)   /* nuc_to_index, :142-155: anything else maps to 0 = ALL */
{
    switch (c) {
    case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'G': case 'g': return 3;
    case 'T': case 't': return 4; case 'N': case 'n': return 5; default: return 0;
    }
}

void fxo_qstats_add(fxo_qstats *s, const fxo_batch *in, int qoffset)
{
    fxo_qstats_grow(s, (int)in->stride);
    for (uint64_t r = 0; r < in->n; ++r) {
        const uint8_t *b = in->bases + r * in->stride;
        const uint8_t *q = in->qual ? in->qual + r * in->stride : NULL;
        const int len = in->len ? in->len[r] : (int)in->fixed_len;
        const int reads_count = 1;                        /* get_reads_count(): 1 unless a collapsed FASTA name says otherwise */
        for (int i = 0; i < len; ++i) {                   /* :179-207 */
            const int k = fxo_nuc_index(b[i]);
            s->cycles[i][0].count += reads_count;
            s->cycles[i][k].count += reads_count;
            if (q) {
                const int v = (int)q[i] - qoffset;
                struct fxo_nucdata *d[2] = { &s->cycles[i][0], &s->cycles[i][k] };
                for (int t = 0; t < 2; ++t) {
                    if (v < d[t]->min) d[t]->min = v;
                    if (v > d[t]->max) d[t]->max = v;
                    d[t]->sum += (unsigned long long)(long long)v;
                    d[t]->values[v - FXO_QS_MINQ] += reads_count;
                }
            }
        }
        s->sequences++;
    }
}

static int fxo_qs_nth(const fxo_qstats *s, int cycle, int nuc, int n)   /* get_nth_value, :218-247 */
{
    const struct fxo_nucdata *d = &s->cycles[cycle][nuc];
    if (n == 0) return d->min;
    int pos = 0;
    while (n > 0) {
        if (d->values[pos] > n) break;
        n -= d->values[pos];
        pos++;
        while (pos < FXO_QS_RANGE && d->values[pos] == 0) pos++;
    }
    return pos + FXO_QS_MINQ;
}

static void fxo_qs_box(const fxo_qstats *s, int c, int nuc, int *Q1, int *med, int *Q3, int *IQR, int *lw, int *rw)   /* :276-291, :357-372 */
{
    const struct fxo_nucdata *d = &s->cycles[c][nuc];
    *Q1 = fxo_qs_nth(s, c, nuc, d->count / 4);
    *Q3 = fxo_qs_nth(s, c, nuc, d->count * 3 / 4);
    *med = fxo_qs_nth(s, c, nuc, d->count / 2);
    *IQR = *Q3 - *Q1;
    *lw = (*Q1 - *IQR * 3 / 2) < d->min ? d->min : (*Q1 - *IQR * 3 / 2);
    *rw = (*Q3 + *IQR * 3 / 2) > d->max ? d->max : (*Q3 + *IQR * 3 / 2);
}

#define FXO_EMIT(...) do { int k__ = snprintf(dst ? dst + w : NULL, dst ? (size_t)1 << 20 : 0, __VA_ARGS__); w += (size_t)k__; } while (0)

size_t fxo_qstats_format(const fxo_qstats *s, int new_format, char *dst)
{
    size_t w = 0;
    static const char *nuc_name[6] = {"ALL", "A", "C", "G", "T", "N"};
    static const char *hdr[] = {"count", "min", "max", "sum", "mean", "Q1", "med", "Q3", "IQR", "lW", "rW"};
    int Q1, med, Q3, IQR, lw, rw;
    if (new_format) {                                                  /* print_statistics, :296-334 */
        FXO_EMIT("cycle\tmax_count");
        for (int n = 0; n < 6; ++n) for (int h = 0; h < 11; ++h) FXO_EMIT("\t%s_%s", nuc_name[n], hdr[h]);
        FXO_EMIT("\n");
        const int max_count = s->ncols ? s->cycles[0][0].count : 0;
        for (int c = 0; c < s->ncols; ++c) {
            if (s->cycles[c][0].count == 0) break;
            FXO_EMIT("%d\t%d", c + 1, max_count);
            for (int n = 0; n < 6; ++n) {                              /* print_nucleotide_statistics, :271-294 */
                const struct fxo_nucdata *d = &s->cycles[c][n];
                fxo_qs_box(s, c, n, &Q1, &med, &Q3, &IQR, &lw, &rw);
                FXO_EMIT("\t%d\t%d\t%d\t%lld\t", d->count, d->min, d->max, (long long)d->sum);
                FXO_EMIT("%3.2f\t%d\t%d\t%d\t", ((double)d->sum) / ((double)d->count), Q1, med, Q3);
                FXO_EMIT("%d\t%d\t%d", IQR, lw, rw);
            }
            FXO_EMIT("\n");
        }
        return w;
    }
    FXO_EMIT("column\tcount\tmin\tmax\tsum\tmean\tQ1\tmed\tQ3\tIQR\tlW\trW\tA_Count\tC_Count\tG_Count\tT_Count\tN_Count\tMax_count\n");   /* :347-352 */
    for (int c = 0; c < s->ncols; ++c) {
        const struct fxo_nucdata *d = &s->cycles[c][0];
        if (d->count == 0) break;
        fxo_qs_box(s, c, 0, &Q1, &med, &Q3, &IQR, &lw, &rw);
        FXO_EMIT("%d\t", c + 1);
        FXO_EMIT("%d\t%d\t%d\t%lld\t", d->count, d->min, d->max, (long long)d->sum);
        FXO_EMIT("%3.2f\t%d\t%d\t%d\t", ((double)d->sum) / ((double)d->count), Q1, med, Q3);
        FXO_EMIT("%d\t%d\t%d\t", IQR, lw, rw);
        FXO_EMIT("%d\t%d\t%d\t%d\t%d\t", s->cycles[c][1].count, s->cycles[c][2].count, s->cycles[c][3].count, s->cycles[c][4].count, s->cycles[c][5].count);
        FXO_EMIT("%d\n", s->cycles[0][0].count);
    }
    return w;
}

long long fxo_qstats_hist(const fxo_qstats *s, int col, int cls, int hist[FXO_QS_RANGE])
{
    memset(hist, 0, FXO_QS_RANGE * sizeof(int));
    if (col >= s->ncols) return 0;
    memcpy(hist, s->cycles[col][cls].values, FXO_QS_RANGE * sizeof(int));
    return s->cycles[col][cls].count;
}
