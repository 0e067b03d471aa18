/* fastx_artifacts_filter -- same command line, output and -v report as the reference tool
 * (src/fastx_artifacts_filter/fastx_artifacts_filter.c); the base census runs on the GPU (FXG_STAGE_ARTIFACTS). */
#include <stdio.h>

#include "../fastx.h"
#include "../fastx_args.h"
#include "../fxh_batch.h"

const char *usage =
    "usage: fastx_artifacts_filter [-h] [-v] [-z] [-i INFILE] [-o OUTFILE]\n"
    "MI355X build of the FASTX-Toolkit artifacts filter (same flags as FASTX Toolkit 0.0.14).\n\n"
    "   -h          this help\n"
    "   -i INFILE   FASTA/Q input, default stdin\n"
    "   -o OUTFILE  FASTA/Q output, default stdout\n"
    "   -z          compress output with gzip\n"
    "   -v          verbose report (to stdout if -o is given, else to stderr)\n\n";

int main(int argc, char *argv[])
{
    static FASTX fastx;
    fxh_totals tot;
    fxg_params p;
    fastx_parse_cmdline(argc, argv, "", NULL);
    fastx_init_reader(&fastx, get_input_filename(), FASTA_OR_FASTQ, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fastx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    fxh_default_params(&p, get_fastq_ascii_quality_offset());
    p.stages = FXG_STAGE_ARTIFACTS;
    fxh_run_tool(&fastx, &p, &tot);
    if (verbose_flag()) {
        FILE *rf = get_report_file();
        fprintf(rf, "Input: %zu reads.\n", tot.input_reads);
        fprintf(rf, "Output: %zu reads.\n", tot.output_reads);
        const size_t discarded = tot.input_reads - tot.output_reads;
        fprintf(rf, "discarded %zu (%zu%%) artifact reads.\n", discarded, (discarded * 100) / tot.input_reads);
    }
    fastx_finish(&fastx);
    return 0;
}
