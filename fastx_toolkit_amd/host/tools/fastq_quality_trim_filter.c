/* fastq_quality_trim_filter -- fastq_quality_trimmer | fastq_quality_filter in one process and ONE pass over the reads.
 *
 * Not a FASTX-Toolkit program: the reference runs the two tools as a shell pipe (the text is formatted, piped, and parsed
 * again in between).  The engine decides both stages in a single kernel (FXG_STAGE_QTRIM | FXG_STAGE_QFILTER), so this tool
 * writes exactly the bytes that
 *     fastq_quality_trimmer -t T -l L | fastq_quality_filter -q Q -p P
 * writes, and -v prints the two tools' reports one after the other, as the pipe would (the filter's input is the trimmer's output).
 */
#include <err.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>

#include "../fastx.h"
#include "../fastx_args.h"
#include "../fxh_batch.h"

const char *usage =
    "usage: fastq_quality_trim_filter [-h] [-v] -t N [-l N] [-q N] [-p N] [-z] [-i INFILE] [-o OUTFILE]\n"
    "One-pass equivalent of  fastq_quality_trimmer -t N -l N | fastq_quality_filter -q N -p N  on the MI355X engine.\n\n"
    "   -t N        trimmer: quality threshold, trailing nucleotides with lower quality are trimmed\n"
    "   -l N        trimmer: minimum length after trimming, default 0\n"
    "   -q N        filter: minimum quality score to keep\n"
    "   -p N        filter: minimum percent of bases that must have [-q] quality\n"
    "   -z          compress output with gzip\n"
    "   -i INFILE   FASTQ input, default stdin\n"
    "   -o OUTFILE  FASTQ output, default stdout\n"
    "   -v          verbose report (to stdout if -o is given, else to stderr)\n"
    "   -Q N        ASCII quality offset, default 33\n\n";

static int trim_threshold = 0, trim_min_length = 0, filter_min_quality = 0, filter_min_percent = 0;

static int parse_program_args(int optind_, int optc, char *optarg_)
{
    (void)optind_;
    if (optarg_ == NULL) errx(1, "[-%c] parameter requires an argument value", optc);
    switch (optc) {
    case 't': trim_threshold = (int)strtol(optarg_, NULL, 10); break;
    case 'l':
        trim_min_length = (int)strtoul(optarg_, NULL, 10);
        if (trim_min_length < 0) errx(1, "Invalid minimum length value (-l %s)", optarg_);
        break;
    case 'q': filter_min_quality = (int)strtoul(optarg_, NULL, 10); break;
    case 'p':
        filter_min_percent = (int)strtoul(optarg_, NULL, 10);
        if (filter_min_percent <= 0 || filter_min_percent > 100) errx(1, "Invalid percent value (-p %s)", optarg_);
        break;
    default: errx(1, __FILE__ ":%d: Unknown argument (%c)", __LINE__, optc);
    }
    return 1;
}

int main(int argc, char *argv[])
{
    static FASTX fastx;
    fxh_totals tot;
    fxg_params p;
    fastx_parse_cmdline(argc, argv, "t:l:q:p:", parse_program_args);
    if (trim_threshold == 0) errx(1, "Missing minimum quality threshold value (-t)");
    fastx_init_reader(&fastx, get_input_filename(), FASTQ_ONLY, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fxh_init_writer(&fastx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    fxh_default_params(&p, get_fastq_ascii_quality_offset());
    p.stages = FXG_STAGE_QTRIM | FXG_STAGE_QFILTER;
    p.qt_threshold = trim_threshold;
    p.qt_min_len = trim_min_length;
    p.qf_min_quality = filter_min_quality;
    p.qf_min_percent = filter_min_percent;
    fxh_run_tool(&fastx, &p, &tot);
    if (verbose_flag()) {
        FILE *rf = get_report_file();
        const size_t after_trim = tot.input_reads - tot.qtrim_dropped;
        fprintf(rf, "Minimum Quality Threshold: %d\n", trim_threshold);
        if (trim_min_length > 0) fprintf(rf, "Minimum Length: %d\n", trim_min_length);
        else fprintf(rf, "No minimum Length\n");
        fprintf(rf, "Input: %zu reads.\n", tot.input_reads);
        fprintf(rf, "Output: %zu reads.\n", after_trim);
        fprintf(rf, "discarded %zu (%zu%%) too-short reads.\n", tot.qtrim_dropped, (tot.qtrim_dropped * 100) / tot.input_reads);
        fprintf(rf, "Quality cut-off: %d\n", filter_min_quality);
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
