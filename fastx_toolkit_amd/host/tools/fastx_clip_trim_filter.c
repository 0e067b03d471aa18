/* fastx_clip_trim_filter -- fastx_clipper | fastq_quality_trimmer | fastq_quality_filter in one process and ONE pass over the reads
 * (BASELINE config 5, the "full pipeline").
 *
 * Not a FASTX-Toolkit program: the reference runs the three tools as a shell pipe (the text is formatted, piped and parsed again twice).
 * The engine decides all three stages in a single kernel (FXG_STAGE_CLIP | FXG_STAGE_QTRIM | FXG_STAGE_QFILTER), so this tool writes
 * exactly the bytes that
 *     fastx_clipper -a A -l L [-n] [-c|-C] [-M N] | fastq_quality_trimmer -t T -l M | fastq_quality_filter -q Q -p P
 * writes, and -v prints the three tools' reports one after the other, as the pipe would (each stage's input is its predecessor's output).
 * The trimmer's -l is spelled -m here (the clipper owns -l).  Like fastx_clipper, a run over reads of DIFFERENT lengths goes through one
 * aligner in input order; FXH_CLIP_PARALLEL=1 lifts that when the input has one fixed read length.
 */
#include <err.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "../fastx.h"
#include "../fastx_args.h"
#include "../fxh_batch.h"

const char *usage =
    "usage: fastx_clip_trim_filter [-h] [-v] [-a ADAPTER] [-l N] [-n] [-c] [-C] [-M N] -t N [-m N] [-q N] [-p N] [-z] [-i INFILE] [-o OUTFILE]\n"
    "One-pass equivalent of  fastx_clipper -a A -l N | fastq_quality_trimmer -t N -l M | fastq_quality_filter -q N -p N  on the MI355X engine.\n\n"
    "   -a ADAPTER  clipper: adapter string, default CCTTAAGG\n"
    "   -l N        clipper: discard sequences shorter than N nucleotides after clipping, default 5\n"
    "   -n          clipper: keep sequences with unknown (N) nucleotides\n"
    "   -c / -C     clipper: discard non-clipped / clipped sequences\n"
    "   -M N        clipper: require a minimum adapter alignment length of N\n"
    "   -t N        trimmer: quality threshold, trailing nucleotides with lower quality are trimmed\n"
    "   -m N        trimmer: minimum length after trimming (the trimmer's -l), default 0\n"
    "   -q N        filter: minimum quality score to keep\n"
    "   -p N        filter: minimum percent of bases that must have [-q] quality\n"
    "   -z          compress output with gzip\n"
    "   -i INFILE   FASTQ input, default stdin\n"
    "   -o OUTFILE  FASTQ output, default stdout\n"
    "   -v          verbose report (to stdout if -o is given, else to stderr)\n"
    "   -Q N        ASCII quality offset, default 33\n\n";

static char adapter[100] = "CCTTAAGG";                       /* fastx_clipper.cpp:68 */
static unsigned int clip_min_length = 5;                     /* :69 */
static int keep_n = 0, only_clipped = 0, only_non_clipped = 0, min_adapter = 0;
static int trim_threshold = 0, trim_min_length = 0, filter_min_quality = 0, filter_min_percent = 0;

static int parse_program_args(int optind_, int optc, char *optarg_)
{
    (void)optind_;
    switch (optc) {
    case 'n': keep_n = 1; return 1;
    case 'c': only_clipped = 1; return 1;
    case 'C': only_non_clipped = 1; return 1;
    default: break;
    }
    if (optarg_ == NULL) errx(1, "[-%c] parameter requires an argument value", optc);
    switch (optc) {
    case 'a': strncpy(adapter, optarg_, sizeof adapter - 1); break;
    case 'l': clip_min_length = (unsigned int)strtoul(optarg_, NULL, 10); break;
    case 'M':
        min_adapter = atoi(optarg_);
        if (min_adapter <= 0) errx(1, "Invalid minimum adapter length (-M %s)", optarg_);
        break;
    case 't': trim_threshold = (int)strtol(optarg_, NULL, 10); break;
    case 'm':
        trim_min_length = (int)strtoul(optarg_, NULL, 10);
        if (trim_min_length < 0) errx(1, "Invalid minimum length value (-m %s)", optarg_);
        break;
    case 'q': filter_min_quality = (int)strtoul(optarg_, NULL, 10); break;
    case 'p':
        filter_min_percent = (int)strtoul(optarg_, NULL, 10);
        if (filter_min_percent <= 0 || filter_min_percent > 100) errx(1, "Invalid percent value (-p %s)", optarg_);
        break;
    default: errx(1, "Unknown argument (%c)", optc);
    }
    return 1;
}

int main(int argc, char *argv[])
{
    static FASTX fastx;
    fxh_totals tot;
    fxg_params p;
    fastx_parse_cmdline(argc, argv, "a:l:ncCM:t:m:q:p:", parse_program_args);
    if (trim_threshold == 0) errx(1, "Missing minimum quality threshold value (-t)");
    fastx_init_reader(&fastx, get_input_filename(), FASTQ_ONLY, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fxh_init_writer(&fastx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    fxh_default_params(&p, get_fastq_ascii_quality_offset());
    p.stages = FXG_STAGE_CLIP | FXG_STAGE_QTRIM | FXG_STAGE_QFILTER;
    memcpy(p.adapter, adapter, sizeof p.adapter);          /* both are char[100], NUL-terminated */
    p.clip_min_len = clip_min_length;
    p.clip_min_adapter_len = min_adapter;
    p.clip_flags = (only_clipped ? FXG_CLIP_DISCARD_NON_CLIPPED : 0u) | (only_non_clipped ? FXG_CLIP_DISCARD_CLIPPED : 0u) | (keep_n ? FXG_CLIP_KEEP_N : 0u);
    p.qt_threshold = trim_threshold;
    p.qt_min_len = trim_min_length;
    p.qf_min_quality = filter_min_quality;
    p.qf_min_percent = filter_min_percent;
    fxh_run_tool(&fastx, &p, &tot);
    if (verbose_flag()) {
        FILE *rf = get_report_file();
        const unsigned clip_out = tot.clip_input - tot.clip_too_short - tot.clip_no_adapter - tot.clip_adapter_found - tot.clip_n - tot.clip_adapter_only;
        const size_t after_trim = (size_t)clip_out - tot.qtrim_dropped;
        fprintf(rf, "Clipping Adapter: %s\n", adapter);                       /* fastx_clipper.cpp:324-348 */
        fprintf(rf, "Min. Length: %d\n", (int)clip_min_length);
        if (only_non_clipped) fprintf(rf, "Clipped reads - discarded.\n");
        if (only_clipped) fprintf(rf, "Non-Clipped reads - discarded.\n");
        fprintf(rf, "Input: %u reads.\n", tot.clip_input);
        fprintf(rf, "Output: %u reads.\n", clip_out);
        fprintf(rf, "discarded %u too-short reads.\n", tot.clip_too_short);
        fprintf(rf, "discarded %u adapter-only reads.\n", tot.clip_adapter_only);
        if (only_clipped) fprintf(rf, "discarded %u non-clipped reads.\n", tot.clip_no_adapter);
        if (only_non_clipped) fprintf(rf, "discarded %u clipped reads.\n", tot.clip_adapter_found);
        if (!keep_n) fprintf(rf, "discarded %u N reads.\n", tot.clip_n);
        fprintf(rf, "Minimum Quality Threshold: %d\n", trim_threshold);        /* fastq_quality_trimmer.c:107-121 */
        if (trim_min_length > 0) fprintf(rf, "Minimum Length: %d\n", trim_min_length);
        else fprintf(rf, "No minimum Length\n");
        fprintf(rf, "Input: %zu reads.\n", (size_t)clip_out);
        fprintf(rf, "Output: %zu reads.\n", after_trim);
        if (clip_out) fprintf(rf, "discarded %zu (%zu%%) too-short reads.\n", tot.qtrim_dropped, (tot.qtrim_dropped * 100) / (size_t)clip_out);
        fprintf(rf, "Quality cut-off: %d\n", filter_min_quality);              /* fastq_quality_filter.c:165-175 */
        fprintf(rf, "Minimum percentage: %d\n", filter_min_percent);
        fprintf(rf, "Input: %zu reads.\n", after_trim);
        fprintf(rf, "Output: %zu reads.\n", tot.output_reads);
        if (after_trim) fprintf(rf, "discarded %zu (%zu%%) low-quality reads.\n", after_trim - tot.output_reads, ((after_trim - tot.output_reads) * 100) / after_trim);
    }
    fastx_finish(&fastx);
    fflush(NULL);
    if (getenv("FXH_TIMING")) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); fprintf(stderr, "fxh timing exit: _exit at %.3f (CLOCK_MONOTONIC)\n", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec); }
    if (!getenv("FXH_SLOW_EXIT")) _exit(0);      /* as fxh_tool_main: skip the HIP runtime's exit handlers */
    return 0;
}
