/* fastx_reverse_complement -- command line, output and -v report of the FASTX-Toolkit tool of that name (behaviour:
 * src/fastx_reverse_complement/fastx_reverse_complement.c); the reversal is the engine's gather with a descending anchor. */
#include <err.h>
#include <limits.h>
#include <string.h>

#include "../fxh_tool.h"

static const fxh_report_line report[] = {
    {FXH_W_ALWAYS, 0, 0, {{"Printing Reverse-Complement Sequences.\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Input: ", FXH_V_IN, 0}, {" reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Output: ", FXH_V_OUT, 0}, {" reads.\n", FXH_V_NONE, 0}}},
};
static void configure(const long *v, const char *s, fxg_params *p) { (void)v; (void)s; p->stages = FXG_STAGE_REVCOMP; }
static const fxh_tool tool = {
    "usage: fastx_reverse_complement [-h] [-r] [-z] [-v] [-i INFILE] [-o OUTFILE]\n"
    "MI355X build of the FASTX-Toolkit reverse-complement tool (same flags as FASTX Toolkit 0.0.14).\n\n"
    "   -h          this help\n"
    "   -z          compress output with gzip\n"
    "   -i INFILE   FASTA/Q input, default stdin\n"
    "   -o OUTFILE  FASTA/Q output, default stdout\n\n",
    "", NULL, 0, NULL, {0}, NULL, FASTA_OR_FASTQ, OUTPUT_SAME_AS_INPUT, NULL, configure, report, 3, NULL,
};
int main(int argc, char *argv[]) { return fxh_tool_main(&tool, argc, argv); }
