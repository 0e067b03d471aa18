/* fastq_to_fasta -- same command line, output and -v report as the reference tool (src/fastq_to_fasta/fastq_to_fasta.c);
 * the N-discard test (and the alphabet check) runs on the GPU (FXG_STAGE_NFILTER); FASTA formatting and the optional
 * renaming are host work, as in the reference. */
#include <err.h>
#include <stdio.h>

#include "../fastx.h"
#include "../fastx_args.h"
#include "../fxh_batch.h"

const char *usage =
    "usage: fastq_to_fasta [-h] [-r] [-n] [-v] [-z] [-i INFILE] [-o OUTFILE]\n"
    "MI355X build of the FASTX-Toolkit FASTQ to FASTA converter (same flags as FASTX Toolkit 0.0.14).\n\n"
    "   -h          this help\n"
    "   -r          rename sequence identifiers to numbers\n"
    "   -n          keep sequences with unknown (N) nucleotides, default is to discard them\n"
    "   -v          verbose report (to stdout if -o is given, else to stderr)\n"
    "   -z          compress output with gzip\n"
    "   -i INFILE   FASTQ input, default stdin\n"
    "   -o OUTFILE  FASTA output, default stdout\n\n";

static int flag_rename_seqid = 0, flag_discard_N = 1;

static int parse_program_args(int optind_, int optc, char *optarg_)
{
    (void)optind_; (void)optarg_;
    switch (optc) {
    case 'n': flag_discard_N = 0; break;
    case 'r': flag_rename_seqid = 1; break;
    default: errx(1, __FILE__ ":%d: Unknown argument (%c)", __LINE__, optc);
    }
    return 1;
}

int main(int argc, char *argv[])
{
    static FASTX fastx;
    fxh_totals tot;
    fxg_params p;
    fastx_parse_cmdline(argc, argv, "rn", parse_program_args);
    fastx_init_reader(&fastx, get_input_filename(), FASTQ_ONLY, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fastx, get_output_filename(), OUTPUT_FASTA, compress_output_flag());
    fxh_default_params(&p, get_fastq_ascii_quality_offset());
    p.stages = FXG_STAGE_NFILTER;
    p.nf_keep_n = flag_discard_N ? 0u : 1u;
    fxh_set_rename_ids(flag_rename_seqid);
    fxh_run_tool(&fastx, &p, &tot);
    if (verbose_flag()) {
        FILE *rf = get_report_file();
        fprintf(rf, "Input: %zu reads.\n", tot.input_reads);
        fprintf(rf, "Output: %zu reads.\n", tot.output_reads);
        if (flag_discard_N) {
            const size_t discarded = tot.input_reads - tot.output_reads;
            fprintf(rf, "discarded %zu (%zu%%) low-quality reads.\n", discarded, (discarded * 100) / tot.input_reads);
        }
    }
    fastx_finish(&fastx);
    return 0;
}
