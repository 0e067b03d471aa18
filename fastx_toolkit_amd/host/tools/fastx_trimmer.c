/* fastx_trimmer -- same command line, output and -v report as the reference tool (src/fastx_trimmer/fastx_trimmer.c);
 * the per-read length arithmetic runs on the GPU (FXG_STAGE_FTRIM / FXG_STAGE_FTRIM_END). */
#include <err.h>
#include <stdio.h>
#include <stdlib.h>

#include "../fastx.h"
#include "../fastx_args.h"
#include "../fxh_batch.h"

const char *usage =
    "usage: fastx_trimmer [-h] [-f N] [-l N] [-t N] [-m MINLEN] [-z] [-v] [-i INFILE] [-o OUTFILE]\n"
    "MI355X build of the FASTX-Toolkit trimmer (same flags as FASTX Toolkit 0.0.14).\n\n"
    "   -h          this help\n"
    "   -f N        first base to keep, default 1\n"
    "   -l N        last base to keep, default the entire read\n"
    "   -t N        trim N nucleotides from the end of the read (not together with -f / -l)\n"
    "   -m MINLEN   with -t: discard reads shorter than MINLEN\n"
    "   -z          compress output with gzip\n"
    "   -i INFILE   FASTA/Q input, default stdin\n"
    "   -o OUTFILE  FASTA/Q output, default stdout\n\n";

static int keep_first_base = 1, keep_last_base = 0, trim_by_position = 0, trim_from_end = 0;
static unsigned int trim_last_bases = 0, minimum_length = 0;

static int parse_program_args(int optind_, int optc, char *optarg_)
{
    (void)optind_;
    switch (optc) {
    case 'f':
        if (optarg_ == NULL) errx(1, "[-f] parameter requires an argument value");
        keep_first_base = (int)strtoul(optarg_, NULL, 10);
        if (keep_first_base <= 0 || keep_first_base >= MAX_SEQ_LINE_LENGTH) errx(1, "Invalid number bases to keep (-f %s)", optarg_);
        trim_by_position = 1;
        break;
    case 'l':
        if (optarg_ == NULL) errx(1, "[-l] parameter requires an argument value");
        keep_last_base = (int)strtoul(optarg_, NULL, 10);
        if (keep_last_base <= 0 || keep_last_base >= MAX_SEQ_LINE_LENGTH) errx(1, "Invalid number bases to keep (-l %s)", optarg_);
        trim_by_position = 1;
        break;
    case 't':
        if (optarg_ == NULL) errx(1, "[-t] parameter requires an argument value");
        trim_last_bases = (unsigned int)strtoul(optarg_, NULL, 10);
        if (trim_last_bases <= 0 || trim_last_bases >= MAX_SEQ_LINE_LENGTH) errx(1, "Invalid number bases to trim (-t %s)", optarg_);
        trim_from_end = 1;
        break;
    case 'm':
        if (optarg_ == NULL) errx(1, "[-t] parameter requires an argument value");
        minimum_length = (unsigned int)strtoul(optarg_, NULL, 10);
        if (minimum_length <= 0 || minimum_length >= MAX_SEQ_LINE_LENGTH) errx(1, "Invalid minimum length value (-m %s)", optarg_);
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
    fastx_parse_cmdline(argc, argv, "l:f:t:m:", parse_program_args);
    if (trim_by_position && trim_from_end) errx(1, "[-t], [-f] and [-l] options can not be used together. Use [-t] or [-l,-f]");
    fastx_init_reader(&fastx, get_input_filename(), FASTA_OR_FASTQ, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fastx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    fxh_default_params(&p, get_fastq_ascii_quality_offset());
    if (trim_from_end) {
        p.stages = FXG_STAGE_FTRIM_END;
        p.ft_trim_end = trim_last_bases;
        p.ft_min_len = minimum_length;    /* -m without -t has no effect in the reference either */
    } else {
        p.stages = FXG_STAGE_FTRIM;
        p.ft_first = keep_first_base;
        p.ft_last = keep_last_base;
    }
    fxh_run_tool(&fastx, &p, &tot);
    if (verbose_flag()) {
        FILE *rf = get_report_file();
        if (keep_first_base != 1 || keep_last_base != 0) fprintf(rf, "Trimming: base %d to %d\n", keep_first_base, keep_last_base);
        if (trim_last_bases) {
            fprintf(rf, "Trimming %d bases from the end of the reads\n", trim_last_bases);
            if (minimum_length) fprintf(rf, "Discarding reads shorter than %d bases\n", minimum_length);
        }
        fprintf(rf, "Input: %zu reads.\n", tot.input_reads);
        fprintf(rf, "Output: %zu reads.\n", tot.output_reads);
    }
    fastx_finish(&fastx);
    return 0;
}
