/* fastq_quality_filter -- same command line, output and -v report as the reference tool
 * (src/fastq_quality_filter/fastq_quality_filter.c); the percentile test runs on the GPU (FXG_STAGE_QFILTER). */
#include <err.h>
#include <stdio.h>
#include <stdlib.h>

#include "../fastx.h"
#include "../fastx_args.h"
#include "../fxh_batch.h"

const char *usage =
    "usage: fastq_quality_filter [-h] [-v] [-q N] [-p N] [-z] [-i INFILE] [-o OUTFILE]\n"
    "MI355X build of the FASTX-Toolkit quality filter (same flags as FASTX Toolkit 0.0.14).\n\n"
    "   -h          this help\n"
    "   -q N        minimum quality score to keep\n"
    "   -p N        minimum percent of bases that must have [-q] quality\n"
    "   -z          compress output with gzip\n"
    "   -i INFILE   FASTQ input, default stdin\n"
    "   -o OUTFILE  FASTQ output, default stdout\n"
    "   -v          verbose report (to stdout if -o is given, else to stderr)\n"
    "   -Q N        ASCII quality offset, default 33\n\n";

static int min_quality = 0, min_percent = 0;

static int parse_program_args(int optind_, int optc, char *optarg_)
{
    (void)optind_;
    switch (optc) {
    case 'q':
        if (optarg_ == NULL) errx(1, "[-q] parameter requires an argument value");
        min_quality = (int)strtoul(optarg_, NULL, 10);
        break;
    case 'p':
        if (optarg_ == NULL) errx(1, "[-l] parameter requires an argument value");
        min_percent = (int)strtoul(optarg_, NULL, 10);
        if (min_percent <= 0 || min_percent > 100) errx(1, "Invalid percent value (-p %s)", optarg_);
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
    fastx_parse_cmdline(argc, argv, "q:p:", parse_program_args);
    fastx_init_reader(&fastx, get_input_filename(), FASTQ_ONLY, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fastx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    fxh_default_params(&p, get_fastq_ascii_quality_offset());
    p.stages = FXG_STAGE_QFILTER;
    p.qf_min_quality = min_quality;
    p.qf_min_percent = min_percent;      /* 0 when -p was not given: every read passes unless -q > 93 (quirk F2) */
    fxh_run_tool(&fastx, &p, &tot);
    if (verbose_flag()) {
        FILE *rf = get_report_file();
        fprintf(rf, "Quality cut-off: %d\n", min_quality);
        fprintf(rf, "Minimum percentage: %d\n", min_percent);
        fprintf(rf, "Input: %zu reads.\n", tot.input_reads);
        fprintf(rf, "Output: %zu reads.\n", tot.output_reads);
        const size_t discarded = tot.input_reads - tot.output_reads;
        fprintf(rf, "discarded %zu (%zu%%) low-quality reads.\n", discarded, (discarded * 100) / tot.input_reads);
    }
    fastx_finish(&fastx);
    return 0;
}
