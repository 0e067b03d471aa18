/* fastq_masker -- same command line, output and -v report as the reference tool (src/fastq_masker/fastq_masker.c);
 * the per-base quality test and replacement run on the GPU (FXG_STAGE_MASK). */
#include <err.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../fastx.h"
#include "../fastx_args.h"
#include "../fxh_batch.h"

const char *usage =
    "usage: fastq_masker [-h] [-v] [-q N] [-r C] [-z] [-i INFILE] [-o OUTFILE]\n"
    "MI355X build of the FASTX-Toolkit quality masker (same flags as FASTX Toolkit 0.0.14).\n\n"
    "   -h          this help\n"
    "   -q N        quality threshold: nucleotides with lower quality are masked, default 10\n"
    "   -r C        replacement character, default 'N'\n"
    "   -z          compress output with gzip\n"
    "   -i INFILE   FASTQ input, default stdin\n"
    "   -o OUTFILE  FASTQ output, default stdout\n"
    "   -v          verbose report (to stdout if -o is given, else to stderr)\n\n";

static int min_quality_threshold = 10;
static char mask_character = 'N';

static int parse_program_args(int optind_, int optc, char *optarg_)
{
    (void)optind_;
    switch (optc) {
    case 'q':
        if (optarg_ == NULL) errx(1, "[-q] parameter requires an argument value");
        min_quality_threshold = atoi(optarg_);
        if (min_quality_threshold < -40) errx(1, "Invalid minimum length value (-q %s)", optarg_);
        break;
    case 'r':
        if (optarg_ == NULL) errx(1, "[-r] parameter requires an argument value");
        if (strlen(optarg_) != 1) errx(1, "[-r] parameter requires a single character as value");
        mask_character = optarg_[0];
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
    fastx_parse_cmdline(argc, argv, "q:r:", parse_program_args);
    fastx_init_reader(&fastx, get_input_filename(), FASTQ_ONLY, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fastx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    fxh_default_params(&p, get_fastq_ascii_quality_offset());
    p.stages = FXG_STAGE_MASK;
    p.mask_min_quality = min_quality_threshold;
    p.mask_char = (unsigned char)mask_character;
    fxh_run_tool(&fastx, &p, &tot);
    if (verbose_flag()) {
        FILE *rf = get_report_file();
        fprintf(rf, "Minimum Quality Threshold: %d\n", min_quality_threshold);
        fprintf(rf, "Low-quality nucleotides replaced with '%c'\n", mask_character);
        fprintf(rf, "Input: %zu reads.\n", tot.input_reads);
        fprintf(rf, "Output: %zu reads.\n", tot.output_reads);
        fprintf(rf, "Masked reads: %zu\n", tot.masked_reads);
        fprintf(rf, "Masked nucleotides: %zu\n", tot.masked_nucleotides);
    }
    fastx_finish(&fastx);
    return 0;
}
