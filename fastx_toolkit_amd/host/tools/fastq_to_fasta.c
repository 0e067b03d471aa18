/* fastq_to_fasta -- command line, output and -v report of the FASTX-Toolkit tool of that name (behaviour:
 * src/fastq_to_fasta/fastq_to_fasta.c); the N-discard is the engine's base census (FXG_STAGE_NFILTER), the writer emits FASTA. */
#include <err.h>
#include <limits.h>
#include <string.h>

#include "../fxh_tool.h"

enum { RENAME, KEEP_N };

static const fxh_option options[] = {
    {'r', FXH_K_FLAG, RENAME, 1, NULL, 0, 0, 0, NULL, -1, 0},
    {'n', FXH_K_FLAG, KEEP_N, 1, NULL, 0, 0, 0, NULL, -1, 0},
};
static const fxh_report_line report[] = {
    {FXH_W_ALWAYS, 0, 0, {{"Input: ", FXH_V_IN, 0}, {" reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Output: ", FXH_V_OUT, 0}, {" reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_Z, KEEP_N, 0, {{"discarded ", FXH_V_DISCARDED, 0}, {" (", FXH_V_DISCARDED_PCT, 0}, {"%) low-quality reads.\n", FXH_V_NONE, 0}}},
};
static void configure(const long *v, const char *s, fxg_params *p)
{
    (void)s;
    p->stages = FXG_STAGE_NFILTER;
    p->nf_keep_n = v[KEEP_N] ? 1u : 0u;
    fxh_set_rename_ids((int)v[RENAME]);           /* kept records are renamed to their 1-based output index (fastq_to_fasta.c:83-84) */
}
static const fxh_tool tool = {
    "usage: fastq_to_fasta [-h] [-r] [-n] [-v] [-z] [-i INFILE] [-o OUTFILE]\n"
    "MI355X build of the FASTX-Toolkit FASTQ to FASTA converter (same flags as FASTX Toolkit 0.0.14).\n\n"
    "   -h          this help\n"
    "   -r          rename sequence identifiers to numbers\n"
    "   -n          keep sequences with unknown (N) nucleotides, default is to discard them\n"
    "   -v          verbose report (to stdout if -o is given, else to stderr)\n"
    "   -z          compress output with gzip\n"
    "   -i INFILE   FASTQ input, default stdin\n"
    "   -o OUTFILE  FASTA output, default stdout\n\n",
    "rn", options, 2, NULL, {0, 0}, NULL, FASTQ_ONLY, OUTPUT_FASTA, NULL, configure, report, 3, NULL,
};
int main(int argc, char *argv[]) { return fxh_tool_main(&tool, argc, argv); }
