/* fastq_quality_trimmer -- same command line, output and -v report as the reference tool
 * (src/fastq_quality_trimmer/fastq_quality_trimmer.c); the 3'-end scan runs on the GPU (FXG_STAGE_QTRIM). */
#include <err.h>
#include <stdio.h>
#include <stdlib.h>

#include "../fastx.h"
#include "../fastx_args.h"
#include "../fxh_batch.h"

const char *usage =
    "usage: fastq_quality_trimmer [-h] [-v] [-t N] [-l N] [-z] [-i INFILE] [-o OUTFILE]\n"
    "MI355X build of the FASTX-Toolkit quality trimmer (same flags as FASTX Toolkit 0.0.14).\n\n"
    "   -h          this help\n"
    "   -t N        quality threshold: trailing nucleotides with lower quality are trimmed\n"
    "   -l N        minimum length: shorter sequences (after trimming) are discarded; default 0\n"
    "   -z          compress output with gzip\n"
    "   -i INFILE   FASTQ input, default stdin\n"
    "   -o OUTFILE  FASTQ output, default stdout\n"
    "   -v          verbose report (to stdout if -o is given, else to stderr)\n"
    "   -Q N        ASCII quality offset, default 33\n\n";

static int min_quality_threshold = 0, min_length = 0;

static int parse_program_args(int optind_, int optc, char *optarg_)
{
    (void)optind_;
    switch (optc) {
    case 'l':
        if (optarg_ == NULL) errx(1, "[-l] parameter requires an argument value");
        min_length = (int)strtoul(optarg_, NULL, 10);            /* strtoul into int, as the reference (F1) */
        if (min_length < 0) errx(1, "Invalid minimum length value (-l %s)", optarg_);
        break;
    case 't':
        if (optarg_ == NULL) errx(1, "[-t] parameter requires an argument value");
        min_quality_threshold = (int)strtol(optarg_, NULL, 10);  /* negatives are accepted */
        break;
    default:
        errx(1, __FILE__ ":%d: Unknown argument (%c)", __LINE__, optc);
    }
    return 1;
}

int main(int argc, char *argv[])
{
    static FASTX fastx;
    fxh_totals tot;
    fxg_params p;
    fastx_parse_cmdline(argc, argv, "t:l:", parse_program_args);
    if (min_quality_threshold == 0) errx(1, "Missing minimum quality threshold value (-t)");
    fastx_init_reader(&fastx, get_input_filename(), FASTQ_ONLY, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fastx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    fxh_default_params(&p, get_fastq_ascii_quality_offset());
    p.stages = FXG_STAGE_QTRIM;
    p.qt_threshold = min_quality_threshold;
    p.qt_min_len = min_length;
    fxh_run_tool(&fastx, &p, &tot);
    if (verbose_flag()) {
        FILE *rf = get_report_file();
        fprintf(rf, "Minimum Quality Threshold: %d\n", min_quality_threshold);
        if (min_length > 0) fprintf(rf, "Minimum Length: %d\n", min_length);
        else fprintf(rf, "No minimum Length\n");
        fprintf(rf, "Input: %zu reads.\n", tot.input_reads);
        fprintf(rf, "Output: %zu reads.\n", tot.output_reads);
        const size_t discarded = tot.input_reads - tot.output_reads;
        fprintf(rf, "discarded %zu (%zu%%) too-short reads.\n", discarded, (discarded * 100) / tot.input_reads);
    }
    fastx_finish(&fastx);
    return 0;
}
