/* fastx_copy -- per-record API demo/test: reads every record with fastx_read_next_record() and writes it back
 * with fastx_write_record() (what every reference tool does around its loop body).  Needs no GPU; used by
 * the CPU test tier to check the record API against the reference's reader/writer. */
#include "../fastx.h"
#include "../fastx_args.h"

const char *usage = "usage: fastx_copy [-h] [-v] [-z] [-Q N] [-i INFILE] [-o OUTFILE]\n";

int main(int argc, char *argv[])
{
    static FASTX fastx;
    fastx_parse_cmdline(argc, argv, "", NULL);
    fastx_init_reader(&fastx, get_input_filename(), FASTA_OR_FASTQ, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fastx_init_writer(&fastx, get_output_filename(), OUTPUT_SAME_AS_INPUT, compress_output_flag());
    while (fastx_read_next_record(&fastx)) fastx_write_record(&fastx);
    if (verbose_flag()) {
        fprintf(get_report_file(), "Input: %zu reads.\n", num_input_reads(&fastx));
        fprintf(get_report_file(), "Output: %zu reads.\n", num_output_reads(&fastx));
    }
    return 0;   /* buffered output is flushed by the atexit handler, like stdio in the reference */
}
