/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Multi-call driver around the REAL reference libfastx (record reader/writer, argument parser and
 * HalfLocalSequenceAlignment), which oracle/Makefile compiles from the sources where they lie under
 * /root/reference/src/libfastx/ into oracle/_ref/libfastx_ref.a.  The reference's five tool main()s
 * include an autoconf-generated <config.h>, so they are NOT buildable here without a stand-in and
 * are therefore not built; the few lines of per-tool loop body they contain are restated below, each
 * citing the reference lines it follows.  Everything else a tool run executes -- parsing, validation,
 * numeric/ASCII quality handling, output formatting, the aligner and its traceback -- is the
 * reference's own object code.
 *
 * usage:  fxref <tool> [tool flags]      tool = fastq_quality_trimmer | fastq_quality_filter | fastx_clipper | fastx_trimmer |
 *                                               fastx_reverse_complement | fastq_masker | fastx_artifacts_filter | fastq_to_fasta | fastx_quality_stats
 */
#include <err.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <iostream>

/* headers come from $(REF)/src/libfastx via -I, never copied.  Order matters: fastx.h leaves
 * #pragma pack(1) active, so the aligner header must precede it (as in fastx_clipper.cpp:29-37). */
#include "sequence_alignment.h"
#include "fastx.h"
#include "fastx_args.h"

extern "C" { const char *usage = "fxref: oracle driver around the reference libfastx (see oracle/ref_driver.cpp)\n"; }

static FASTX fx;

/* ---- fastq_quality_trimmer.c:54-124 ---- */
static int qt_threshold = 0, qt_min_len = 0;
static int qt_args(int, int c, char *arg)
{
    if (c == 'l') { qt_min_len = (int)strtoul(arg, NULL, 10); if (qt_min_len < 0) errx(1, "Invalid minimum length value (-l %s)", arg); }   /* :60-62 */
    else if (c == 't') qt_threshold = (int)strtol(arg, NULL, 10);
    else errx(1, "Unknown argument (%c)", c);
    return 1;
}
static int run_quality_trimmer(int argc, char **argv)
{
    fastx_parse_cmdline(argc, argv, "t:l:", qt_args);
    if (qt_threshold == 0) errx(1, "Missing minimum quality threshold value (-t)");
    fastx_init_reader(&fx, get_input_filename(), FASTQ_ONLY, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    while (fastx_read_next_record(&fx)) {
        int i = (int)strlen(fx.nucleotides) - 1;
        while (i >= 0 && fx.quality[i] < qt_threshold) fx.nucleotides[i--] = 0;    /* :94-99 */
        if (i >= 0 && i + 1 >= qt_min_len) fastx_write_record(&fx);               /* :101 */
    }
    if (verbose_flag()) {                                                          /* :107-121 */
        FILE *rf = get_report_file();
        fprintf(rf, "Minimum Quality Threshold: %d\n", qt_threshold);
        if (qt_min_len > 0) fprintf(rf, "Minimum Length: %d\n", qt_min_len); else fprintf(rf, "No minimum Length\n");
        size_t in = num_input_reads(&fx), out = num_output_reads(&fx);
        fprintf(rf, "Input: %zu reads.\nOutput: %zu reads.\n", in, out);
        fprintf(rf, "discarded %zu (%zu%%) too-short reads.\n", in - out, ((in - out) * 100) / in);
    }
    return 0;
}

/* ---- fastq_quality_filter.c:56-178 ---- */
static int qf_min_quality = 0, qf_min_percent = 0;
static int qf_args(int, int c, char *arg)
{
    if (c == 'q') qf_min_quality = (int)strtoul(arg, NULL, 10);
    else if (c == 'p') {
        qf_min_percent = (int)strtoul(arg, NULL, 10);
        if (qf_min_percent <= 0 || qf_min_percent > 100) errx(1, "Invalid percent value (-p %s)", arg);
    } else errx(1, "Unknown argument (%c)", c);
    return 1;
}
static int nth_bin(const int *h, int size, int n)                                  /* :78-108 */
{
    int pos = 0;
    while (pos < size && h[pos] == 0) pos++;
    if (pos == size) errx(1, "bug: got empty array");
    while (n > 0) {
        if (h[pos] > n) break;
        n -= h[pos];
        pos++;
        while (pos < size && h[pos] == 0) pos++;
        if (pos >= size) break;
    }
    return pos;
}
static int run_quality_filter(int argc, char **argv)
{
    fastx_parse_cmdline(argc, argv, "q:p:", qf_args);
    fastx_init_reader(&fx, get_input_filename(), FASTQ_ONLY, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    while (fastx_read_next_record(&fx)) {
        int hist[QUALITY_VALUES_RANGE + 2];                                        /* :110-129 (+2: note N2) */
        memset(hist, 0, sizeof hist);
        int count = 0;
        for (size_t i = 0; i < strlen(fx.nucleotides); i++) { count++; hist[fx.quality[i] - MIN_QUALITY_VALUE]++; }
        int v = nth_bin(hist, QUALITY_VALUES_RANGE, count * (100 - qf_min_percent) / 100) + MIN_QUALITY_VALUE;
        if (v >= qf_min_quality) fastx_write_record(&fx);                          /* :155 */
    }
    if (verbose_flag()) {                                                          /* :165-175 */
        FILE *rf = get_report_file();
        fprintf(rf, "Quality cut-off: %d\nMinimum percentage: %d\n", qf_min_quality, qf_min_percent);
        size_t in = num_input_reads(&fx), out = num_output_reads(&fx);
        fprintf(rf, "Input: %zu reads.\nOutput: %zu reads.\n", in, out);
        fprintf(rf, "discarded %zu (%zu%%) low-quality reads.\n", in - out, ((in - out) * 100) / in);
    }
    return 0;
}

/* ---- fastx_trimmer.c:59-162 ---- */
static int ft_first = 1, ft_last = 0, ft_by_pos = 0, ft_from_end = 0;
static unsigned ft_cut = 0, ft_minlen = 0;
static int ft_args(int, int c, char *arg)
{
    switch (c) {
    case 'f': ft_first = (int)strtoul(arg, NULL, 10);
        if (ft_first <= 0 || ft_first >= MAX_SEQ_LINE_LENGTH) errx(1, "Invalid number bases to keep (-f %s)", arg);
        ft_by_pos = 1; break;
    case 'l': ft_last = (int)strtoul(arg, NULL, 10);
        if (ft_last <= 0 || ft_last >= MAX_SEQ_LINE_LENGTH) errx(1, "Invalid number bases to keep (-l %s)", arg);
        ft_by_pos = 1; break;
    case 't': ft_cut = (unsigned)strtoul(arg, NULL, 10);
        if (ft_cut <= 0 || ft_cut >= MAX_SEQ_LINE_LENGTH) errx(1, "Invalid number bases to trim (-t %s)", arg);
        ft_from_end = 1; break;
    case 'm': ft_minlen = (unsigned)strtoul(arg, NULL, 10);
        if (ft_minlen <= 0 || ft_minlen >= MAX_SEQ_LINE_LENGTH) errx(1, "Invalid minimum length value (-m %s)", arg);
        break;
    default: errx(1, "Unknown argument (%c)", c);
    }
    return 1;
}
static int run_trimmer(int argc, char **argv)
{
    fastx_parse_cmdline(argc, argv, "l:f:t:m:", ft_args);
    if (ft_by_pos && ft_from_end) errx(1, "[-t], [-f] and [-l] options can not be used together. Use [-t] or [-l,-f]");
    fastx_init_reader(&fx, get_input_filename(), FASTA_OR_FASTQ, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    while (fastx_read_next_record(&fx)) {
        if (ft_last != 0) fx.nucleotides[ft_last] = 0;                              /* :122-124 */
        if (ft_first != 1) {                                                        /* :126-134 */
            size_t n = strlen(fx.nucleotides);
            if (n < (size_t)ft_first) continue;
            size_t keep = n - (size_t)ft_first + 1;
            memmove(fx.nucleotides, fx.nucleotides + ft_first - 1, keep);
            memmove(fx.quality, fx.quality + ft_first - 1, keep * sizeof(int));
            fx.nucleotides[keep] = 0;
        }
        if (ft_cut > 0) {                                                           /* :136-144 */
            size_t n = strlen(fx.nucleotides);
            if (n <= ft_cut) continue;
            size_t i = n - ft_cut;
            if (i < ft_minlen) continue;
            fx.nucleotides[i] = 0;
        }
        fastx_write_record(&fx);
    }
    if (verbose_flag()) {                                                          /* :150-160 */
        FILE *rf = get_report_file();
        if (ft_first != 1 || ft_last != 0) fprintf(rf, "Trimming: base %d to %d\n", ft_first, ft_last);
        if (ft_cut) {
            fprintf(rf, "Trimming %d bases from the end of the reads\n", ft_cut);
            if (ft_minlen) fprintf(rf, "Discarding reads shorter than %d bases\n", ft_minlen);
        }
        fprintf(rf, "Input: %zu reads.\nOutput: %zu reads.\n", num_input_reads(&fx), num_output_reads(&fx));
    }
    return 0;
}

/* ---- fastx_reverse_complement.c:43-128 ---- */
static char rc_base(char c)
{
    static const char from[] = "NnATGCatgc", to[] = "NnTACGtacg";
    const char *p = strchr(from, c);
    if (!p || !c) errx(1, "Invalid nucleotide value (%c) in reverse_complement_base()", c);
    return to[p - from];
}
static int run_revcomp(int argc, char **argv)
{
    fastx_parse_cmdline(argc, argv, "", NULL);
    fastx_init_reader(&fx, get_input_filename(), FASTA_OR_FASTQ, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    while (fastx_read_next_record(&fx)) {
        int n = (int)strlen(fx.nucleotides);
        for (int i = 0; i < n; i++) fx.nucleotides[i] = rc_base(fx.nucleotides[i]);
        for (int i = 0, j = n - 1; i < j; i++, j--) {
            char t = fx.nucleotides[i]; fx.nucleotides[i] = fx.nucleotides[j]; fx.nucleotides[j] = t;
            if (fx.read_fastq) { int q = fx.quality[i]; fx.quality[i] = fx.quality[j]; fx.quality[j] = q; }
        }
        fastx_write_record(&fx);
    }
    if (verbose_flag()) {
        FILE *rf = get_report_file();
        fprintf(rf, "Printing Reverse-Complement Sequences.\n");
        fprintf(rf, "Input: %zu reads.\nOutput: %zu reads.\n", num_input_reads(&fx), num_output_reads(&fx));
    }
    return 0;
}

/* ---- fastx_clipper.cpp:66-350 ---- */
static char cl_adapter[100] = "CCTTAAGG";
static unsigned cl_min_length = 5;
static int cl_discard_n = 1, cl_keep_delta = 0, cl_only_clipped = 0, cl_only_nonclipped = 0, cl_adapter_only = 0;
static int cl_min_adapter = 0, cl_debug = 0;
static unsigned n_in = 0, n_short = 0, n_adapter0 = 0, n_noadapter = 0, n_adapter = 0, n_N = 0;
static HalfLocalSequenceAlignment aligner;

static int cl_args(int, int c, char *arg)
{
    switch (c) {
    case 'M': cl_min_adapter = atoi(arg); if (cl_min_adapter <= 0) errx(1, "Invalid minimum adapter length (-M %s)", arg); break;
    case 'k': cl_adapter_only = 1; break;
    case 'D': cl_debug++; break;
    case 'c': cl_only_clipped = 1; break;
    case 'C': cl_only_nonclipped = 1; break;
    case 'd': cl_keep_delta = (int)strtoul(arg, NULL, 10); if (cl_keep_delta < 0) errx(1, "Invalid number bases to keep (-d %s)", arg); break;
    case 'a': strncpy(cl_adapter, arg, sizeof(cl_adapter) - 1); break;
    case 'l': cl_min_length = (unsigned)strtoul(arg, NULL, 10); break;
    case 'n': cl_discard_n = 0; break;
    default: errx(1, "Unknown argument (%c)", c);
    }
    return 1;
}
static int cutoff(const SequenceAlignmentResults &r)                                /* :192-240 */
{
    int sz = r.neutral_matches + r.matches + r.mismatches + r.gaps;
    if (sz == 0) return -1;
    if (cl_min_adapter > 0 && sz < cl_min_adapter) return -1;
    if (r.query_end == r.query_size - 1 && r.mismatches == 0) return r.query_start;
    if (sz > 5 && r.target_start == 0 && (r.matches * 100 / sz) >= 75) return r.query_start;
    if (sz > 11 && (r.matches * 100 / sz) >= 80) return r.query_start;
    if (r.query_end >= r.query_size - 2 && sz <= 5 && r.matches >= 3) return r.query_start;
    return -1;
}
static int run_clipper(int argc, char **argv)
{
    fastx_parse_cmdline(argc, argv, "M:kDCcd:a:s:l:n", cl_args);
    if (cl_keep_delta > 0) cl_keep_delta += strlen(cl_adapter);                      /* :153-154 */
    fastx_init_reader(&fx, get_input_filename(), FASTA_OR_FASTQ, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    while (fastx_read_next_record(&fx)) {
        int reads = get_reads_count(&fx);
        aligner.align(std::string(fx.nucleotides), std::string(cl_adapter));          /* :265-270 */
        if (cl_debug > 1) aligner.print_matrix();                                      /* :272-275 */
        if (cl_debug > 0) aligner.results().print();
        n_in += reads;
        int i = cutoff(aligner.results());
        if (i != -1 && i > 0) { i += cl_keep_delta; fx.nucleotides[i] = 0; }          /* :282-286 */
        if (i == 0) { n_adapter0 += reads; if (cl_adapter_only) fastx_write_record(&fx); continue; }
        if (strlen(fx.nucleotides) < cl_min_length) { n_short += reads; continue; }
        if (i == -1 && cl_only_clipped) { n_noadapter += reads; continue; }
        if (i > 0 && cl_only_nonclipped) { n_adapter += reads; continue; }
        if (cl_discard_n && strchr(fx.nucleotides, 'N') != NULL) { n_N += reads; continue; }
        if (!cl_adapter_only) fastx_write_record(&fx);
    }
    if (verbose_flag()) {                                                           /* :324-347 */
        FILE *rf = get_report_file();
        fprintf(rf, "Clipping Adapter: %s\nMin. Length: %d\n", cl_adapter, cl_min_length);
        if (cl_only_nonclipped) fprintf(rf, "Clipped reads - discarded.\n");
        if (cl_only_clipped) fprintf(rf, "Non-Clipped reads - discarded.\n");
        fprintf(rf, "Input: %u reads.\n", n_in);
        fprintf(rf, "Output: %u reads.\n", n_in - n_short - n_noadapter - n_adapter - n_N - n_adapter0);
        fprintf(rf, "discarded %u too-short reads.\n", n_short);
        fprintf(rf, "discarded %u adapter-only reads.\n", n_adapter0);
        if (cl_only_clipped) fprintf(rf, "discarded %u non-clipped reads.\n", n_noadapter);
        if (cl_only_nonclipped) fprintf(rf, "discarded %u clipped reads.\n", n_adapter);
        if (cl_discard_n) fprintf(rf, "discarded %u N reads.\n", n_N);
    }
    return 0;
}

/* ---- fastq_masker.c:49-123 ---- */
static int mk_min_quality = 10;
static char mk_char = 'N';
static int mk_args(int, int c, char *arg)
{
    if (c == 'q') { mk_min_quality = atoi(arg); if (mk_min_quality < -40) errx(1, "Invalid minimum length value (-q %s)", arg); }
    else if (c == 'r') { if (strlen(arg) != 1) errx(1, "[-r] parameter requires a single character as value"); mk_char = arg[0]; }
    else errx(1, "Unknown argument (%c)", c);
    return 1;
}
static int run_masker(int argc, char **argv)
{
    size_t masked_reads = 0, masked_nt = 0;
    fastx_parse_cmdline(argc, argv, "q:r:", mk_args);
    fastx_init_reader(&fx, get_input_filename(), FASTQ_ONLY, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    while (fastx_read_next_record(&fx)) {
        int any = 0, n = (int)strlen(fx.nucleotides);
        for (int i = 0; i < n; ++i)
            if (fx.quality[i] < mk_min_quality) { fx.nucleotides[i] = mk_char; any = 1; ++masked_nt; }     /* :94-99 */
        if (any) masked_reads += get_reads_count(&fx);
        fastx_write_record(&fx);
    }
    if (verbose_flag()) {                                                           /* :110-120 */
        FILE *rf = get_report_file();
        fprintf(rf, "Minimum Quality Threshold: %d\n", mk_min_quality);
        fprintf(rf, "Low-quality nucleotides replaced with '%c'\n", mk_char);
        fprintf(rf, "Input: %zu reads.\nOutput: %zu reads.\n", num_input_reads(&fx), num_output_reads(&fx));
        fprintf(rf, "Masked reads: %zu\nMasked nucleotides: %zu\n", masked_reads, masked_nt);
    }
    return 0;
}

/* ---- fastx_artifacts_filter.c:56-144 ---- */
static int run_artifacts(int argc, char **argv)
{
    fastx_parse_cmdline(argc, argv, "", NULL);
    fastx_init_reader(&fx, get_input_filename(), FASTA_OR_FASTQ, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    while (fastx_read_next_record(&fx)) {
        int cnt[256] = {0}, total = 0;
        for (const char *p = fx.nucleotides; *p; ++p) {
            if (!strchr("ACGTN", *p)) errx(1, "invalid nucleotide value (%c) at position %d", *p, total);
            cnt[(unsigned char)*p]++; total++;
        }
        const int lim = total - 3;                                                  /* :101-109 */
        if (cnt['A'] >= lim || cnt['C'] >= lim || cnt['G'] >= lim || cnt['T'] >= lim) continue;
        fastx_write_record(&fx);
    }
    if (verbose_flag()) {                                                           /* :132-140 */
        FILE *rf = get_report_file();
        size_t in = num_input_reads(&fx), out = num_output_reads(&fx);
        fprintf(rf, "Input: %zu reads.\nOutput: %zu reads.\n", in, out);
        fprintf(rf, "discarded %zu (%zu%%) artifact reads.\n", in - out, ((in - out) * 100) / in);
    }
    return 0;
}

/* ---- fastq_to_fasta.c:49-103 ---- */
static int f2a_rename = 0, f2a_discard_n = 1;
static int f2a_args(int, int c, char *)
{
    if (c == 'n') f2a_discard_n = 0; else if (c == 'r') f2a_rename = 1; else errx(1, "Unknown argument (%c)", c);
    return 1;
}
static int run_fastq_to_fasta(int argc, char **argv)
{
    fastx_parse_cmdline(argc, argv, "rn", f2a_args);
    fastx_init_reader(&fx, get_input_filename(), FASTQ_ONLY, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fx, get_output_filename(), OUTPUT_FASTA, compress_output_flag());
    while (fastx_read_next_record(&fx)) {
        if (f2a_discard_n && strchr(fx.nucleotides, 'N') != NULL) continue;          /* :80-81 */
        if (f2a_rename) snprintf(fx.name, sizeof(fx.name), "%zu", num_output_reads(&fx) + 1);   /* :83-84 */
        fastx_write_record(&fx);
    }
    if (verbose_flag()) {                                                           /* :90-100 */
        FILE *rf = get_report_file();
        size_t in = num_input_reads(&fx), out = num_output_reads(&fx);
        fprintf(rf, "Input: %zu reads.\nOutput: %zu reads.\n", in, out);
        if (f2a_discard_n) fprintf(rf, "discarded %zu (%zu%%) low-quality reads.\n", in - out, ((in - out) * 100) / in);
    }
    return 0;
}

/* debugging aid for the parity tests: "fxref align QUERY TARGET" prints the 7 result fields */
static int run_align(int argc, char **argv)
{
    if (argc < 3) errx(1, "align QUERY TARGET");
    const SequenceAlignmentResults &r = aligner.align(std::string(argv[1]), std::string(argv[2]));
    printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", r.query_start, r.query_end, r.target_start, r.target_end,
           r.matches, r.mismatches, r.neutral_matches, r.gaps);
    return 0;
}


/* ---- fastx_quality_stats.c:115-463 (struct nucleotide_data, read_file, get_nth_value, both print functions) ---- */
struct qs_nuc { int min, max, count; unsigned long long sum; int values[QUALITY_VALUES_RANGE]; };
/* [column * 6 + nucleotide], static and initialised to its end like the reference's (:136, :158-163): get_nth_value walks past a
 * class that has bases but no qualities (FASTA input) into its neighbours, so what lies there is part of the behaviour.  fastx.h
 * leaves #pragma pack(1) on (:61), here as there. */
static qs_nuc qs_store[(size_t)MAX_SEQ_LINE_LENGTH * 6];
static size_t qs_used = 0;                  /* cycles with data */
static int qs_new_format = 0;
static qs_nuc &qs_at(size_t col, int nuc)
{
    static bool init = false;
    if (!init) { for (size_t i = 0; i < (size_t)MAX_SEQ_LINE_LENGTH * 6; ++i) { qs_store[i].min = 100; qs_store[i].max = -100; } init = true; }   /* init_values */
    if (col >= (size_t)MAX_SEQ_LINE_LENGTH) errx(1, "Internal error: sequence too long. Hard-coded max. length is %d", MAX_SEQ_LINE_LENGTH);
    if (col + 1 > qs_used) qs_used = col + 1;
    return qs_store[col * 6 + (size_t)nuc];
}
static int qs_nuc_index(int c)              /* :142-155 */
{
    switch (c) { case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'G': case 'g': return 3;
                 case 'T': case 't': return 4; case 'N': case 'n': return 5; default: return 0; }
}
static int qs_nth(size_t col, int nuc, int n)   /* :218-247 */
{
    const qs_nuc &d = qs_at(col, nuc);
    if (n == 0) return d.min;
    if (n < 0 || n >= d.count) { fprintf(stderr, "Internal error at get_nth_value\n"); exit(1); }
    int pos = 0;
    while (n > 0) {
        if (d.values[pos] > n) break;
        n -= d.values[pos];
        pos++;
        while (d.values[pos] == 0) pos++;
    }
    return pos + MIN_QUALITY_VALUE;
}
static void qs_print_nuc(FILE *o, size_t col, int nuc, bool old_layout)   /* :271-294 and :357-392 */
{
    const qs_nuc &d = qs_at(col, nuc);
    const int Q1 = qs_nth(col, nuc, d.count / 4), Q3 = qs_nth(col, nuc, d.count * 3 / 4), IQR = Q3 - Q1;
    const int lw = (Q1 - IQR * 3 / 2) < d.min ? d.min : (Q1 - IQR * 3 / 2);
    const int rw = (Q3 + IQR * 3 / 2) > d.max ? d.max : (Q3 + IQR * 3 / 2);
    fprintf(o, old_layout ? "%d\t%d\t%d\t%lld\t" : "\t%d\t%d\t%d\t%lld\t", d.count, d.min, d.max, (long long)d.sum);
    fprintf(o, "%3.2f\t%d\t%d\t%d\t", ((double)d.sum) / ((double)d.count), Q1, qs_nth(col, nuc, d.count / 2), Q3);
    fprintf(o, old_layout ? "%d\t%d\t%d\t" : "%d\t%d\t%d", IQR, lw, rw);
}
static int qs_args(int, int c, char *) { if (c == 'N') qs_new_format = 1; else errx(1, "Unknown argument (%c)", c); return 1; }
static int run_quality_stats(int argc, char **argv)
{
    fastx_parse_cmdline(argc, argv, "N", qs_args);                                  /* :426-441 */
    fastx_init_reader(&fx, get_input_filename(), FASTA_OR_FASTQ, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    FILE *o = stdout;
    if (strcmp(get_output_filename(), "-") != 0) { o = fopen(get_output_filename(), "w+"); if (!o) err(1, "Failed to create output file (%s)", get_output_filename()); }
    while (fastx_read_next_record(&fx)) {                                           /* read_file :166-216 */
        const size_t L = strlen(fx.nucleotides);
        for (size_t i = 0; i < L; ++i) {
            const int k = qs_nuc_index(fx.nucleotides[i]);
            const int rc = get_reads_count(&fx);
            qs_at(i, 0).count += rc; qs_at(i, k).count += rc;
            if (fx.read_fastq) {
                const int v = fx.quality[i];
                qs_nuc *d[2] = { &qs_at(i, 0), &qs_at(i, k) };
                for (int t = 0; t < 2; ++t) {
                    if (v < d[t]->min) d[t]->min = v;
                    if (v > d[t]->max) d[t]->max = v;
                    d[t]->sum += v;
                    d[t]->values[v - MIN_QUALITY_VALUE] += rc;
                }
            }
        }
    }
    const size_t ncols = qs_used;
    if (qs_new_format) {                                                            /* print_statistics :296-334 */
        static const char *nn[6] = {"ALL", "A", "C", "G", "T", "N"};
        static const char *hd[11] = {"count", "min", "max", "sum", "mean", "Q1", "med", "Q3", "IQR", "lW", "rW"};
        fprintf(o, "cycle\tmax_count");
        for (int n = 0; n < 6; ++n) for (int h = 0; h < 11; ++h) fprintf(o, "\t%s_%s", nn[n], hd[h]);
        fprintf(o, "\n");
        const int max_count = ncols ? qs_at(0, 0).count : 0;
        for (size_t c = 0; c < ncols; ++c) {
            if (qs_at(c, 0).count == 0) break;
            fprintf(o, "%d\t%d", (int)c + 1, max_count);
            for (int n = 0; n < 6; ++n) qs_print_nuc(o, c, n, false);
            fprintf(o, "\n");
        }
    } else {                                                                        /* print_old_statistics :340-414 */
        fprintf(o, "column\tcount\tmin\tmax\tsum\tmean\tQ1\tmed\tQ3\tIQR\tlW\trW\tA_Count\tC_Count\tG_Count\tT_Count\tN_Count\tMax_count\n");
        for (size_t c = 0; c < ncols; ++c) {
            if (qs_at(c, 0).count == 0) break;
            fprintf(o, "%d\t", (int)c + 1);
            qs_print_nuc(o, c, 0, true);
            fprintf(o, "%d\t%d\t%d\t%d\t%d\t", qs_at(c, 1).count, qs_at(c, 2).count, qs_at(c, 3).count, qs_at(c, 4).count, qs_at(c, 5).count);
            fprintf(o, "%d\n", qs_at(0, 0).count);
        }
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fputs(usage, stderr); return 2; }
    const char *tool = argv[1];
    argc--; argv++;
    if (!strcmp(tool, "fastq_quality_trimmer")) return run_quality_trimmer(argc, argv);
    if (!strcmp(tool, "fastq_quality_filter")) return run_quality_filter(argc, argv);
    if (!strcmp(tool, "fastx_trimmer")) return run_trimmer(argc, argv);
    if (!strcmp(tool, "fastx_reverse_complement")) return run_revcomp(argc, argv);
    if (!strcmp(tool, "fastx_clipper")) return run_clipper(argc, argv);
    if (!strcmp(tool, "fastq_to_fasta")) return run_fastq_to_fasta(argc, argv);
    if (!strcmp(tool, "fastq_masker")) return run_masker(argc, argv);
    if (!strcmp(tool, "fastx_artifacts_filter")) return run_artifacts(argc, argv);
    if (!strcmp(tool, "fastx_quality_stats")) return run_quality_stats(argc, argv);
    if (!strcmp(tool, "align")) return run_align(argc, argv);
    fputs(usage, stderr);
    return 2;
}
