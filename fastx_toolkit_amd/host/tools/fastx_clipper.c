/* fastx_clipper -- same command line, output and -v report as the reference tool (src/fastx_clipper/fastx_clipper.cpp);
 * the adapter alignment, accept rules and discard cascade run on the GPU (FXG_STAGE_CLIP).
 * Contract note: the reference's aligner is history dependent on variable-length input; this
 * build aligns every read independently, which is identical for fixed-length input. */
#include <err.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../fastx.h"
#include "../fastx_args.h"
#include "../fxh_batch.h"

#define MAX_ADAPTER_LEN 100

const char *usage =
    "usage: fastx_clipper [-h] [-a ADAPTER] [-D] [-l N] [-n] [-d N] [-c] [-C] [-o] [-v] [-z] [-i INFILE] [-o OUTFILE]\n"
    "MI355X build of the FASTX-Toolkit adapter clipper (same flags as FASTX Toolkit 0.0.14).\n\n"
    "   -h          this help\n"
    "   -a ADAPTER  adapter string, default CCTTAAGG (dummy adapter)\n"
    "   -l N        discard sequences shorter than N nucleotides, default 5\n"
    "   -d N        keep the adapter and N bases after it (-d 0 = not using -d)\n"
    "   -c          discard non-clipped sequences (keep only sequences which contained the adapter)\n"
    "   -C          discard clipped sequences (keep only sequences which did not contain the adapter)\n"
    "   -k          report adapter-only sequences\n"
    "   -n          keep sequences with unknown (N) nucleotides, default is to discard them\n"
    "   -v          verbose report (to stdout if -o is given, else to stderr)\n"
    "   -z          compress output with gzip\n"
    "   -D          accepted for compatibility (the GPU aligner has no matrix dump)\n"
    "   -M N        require a minimum adapter alignment length of N\n"
    "   -i INFILE   FASTA/Q input, default stdin\n"
    "   -o OUTFILE  FASTA/Q output, default stdout\n\n";

static char adapter[MAX_ADAPTER_LEN] = "CCTTAAGG";
static unsigned int min_length = 5;
static int discard_unknown_bases = 1, keep_delta = 0, discard_non_clipped = 0, discard_clipped = 0, show_adapter_only = 0;
static int minimum_adapter_length = 0;

static int parse_program_args(int optind_, int optc, char *optarg_)
{
    (void)optind_;
    switch (optc) {
    case 'M':
        if (optarg_ == NULL) errx(1, "[-M] parameter requires an argument value");
        minimum_adapter_length = atoi(optarg_);
        if (minimum_adapter_length <= 0) errx(1, "Invalid minimum adapter length (-M %s)", optarg_);
        break;
    case 'k': show_adapter_only = 1; break;
    case 'D': break;
    case 'c': discard_non_clipped = 1; break;
    case 'C': discard_clipped = 1; break;
    case 'd':
        if (optarg_ == NULL) errx(1, "[-d] parameter requires an argument value");
        keep_delta = (int)strtoul(optarg_, NULL, 10);
        if (keep_delta < 0) errx(1, "Invalid number bases to keep (-d %s)", optarg_);
        break;
    case 'a': strncpy(adapter, optarg_, sizeof(adapter) - 1); break;
    case 'l':
        if (optarg_ == NULL) errx(1, "[-l] parameter requires an argument value");
        min_length = (unsigned int)strtoul(optarg_, NULL, 10);
        break;
    case 'n': discard_unknown_bases = 0; break;
    default: errx(1, "Unknown argument (%c)", optc);     /* includes the reference's unhandled 's' (F3) */
    }
    return 1;
}

int main(int argc, char *argv[])
{
    static FASTX fastx;
    fxh_totals tot;
    fxg_params p;
    fastx_parse_cmdline(argc, argv, "M:kDCcd:a:s:l:n", parse_program_args);
    if (keep_delta > 0) keep_delta += (int)strlen(adapter);       /* fastx_clipper.cpp:153-154 */
    fastx_init_reader(&fastx, get_input_filename(), FASTA_OR_FASTQ, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fastx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    fxh_default_params(&p, get_fastq_ascii_quality_offset());
    p.stages = FXG_STAGE_CLIP;
    strncpy(p.adapter, adapter, sizeof p.adapter - 1);
    p.clip_min_len = min_length;
    p.clip_keep_delta = keep_delta;
    p.clip_min_adapter_len = minimum_adapter_length;
    p.clip_flags = (discard_non_clipped ? FXG_CLIP_DISCARD_NON_CLIPPED : 0u) | (discard_clipped ? FXG_CLIP_DISCARD_CLIPPED : 0u) |
                   (discard_unknown_bases ? 0u : FXG_CLIP_KEEP_N) | (show_adapter_only ? FXG_CLIP_ADAPTER_ONLY : 0u);
    fxh_run_tool(&fastx, &p, &tot);
    if (verbose_flag()) {
        FILE *rf = get_report_file();
        fprintf(rf, "Clipping Adapter: %s\n", adapter);
        fprintf(rf, "Min. Length: %d\n", min_length);
        if (discard_clipped) fprintf(rf, "Clipped reads - discarded.\n");
        if (discard_non_clipped) fprintf(rf, "Non-Clipped reads - discarded.\n");
        fprintf(rf, "Input: %u reads.\n", tot.clip_input);
        fprintf(rf, "Output: %u reads.\n", tot.clip_input - tot.clip_too_short - tot.clip_no_adapter - tot.clip_adapter_found -
                                               tot.clip_n - tot.clip_adapter_only);
        fprintf(rf, "discarded %u too-short reads.\n", tot.clip_too_short);
        fprintf(rf, "discarded %u adapter-only reads.\n", tot.clip_adapter_only);
        if (discard_non_clipped) fprintf(rf, "discarded %u non-clipped reads.\n", tot.clip_no_adapter);
        if (discard_clipped) fprintf(rf, "discarded %u clipped reads.\n", tot.clip_adapter_found);
        if (discard_unknown_bases) fprintf(rf, "discarded %u N reads.\n", tot.clip_n);
    }
    fastx_finish(&fastx);
    return 0;
}
