/* fastq_masker -- command line, output and -v report of the FASTX-Toolkit tool of that name (behaviour: src/fastq_masker/fastq_masker.c);
 * the substitution happens inside the engine's gather (FXG_STAGE_MASK). */
#include <err.h>
#include <limits.h>
#include <string.h>

#include "../fxh_tool.h"

enum { THRESHOLD, MASK_CHAR };

static const fxh_option options[] = {
    {'q', FXH_K_ATOI, THRESHOLD, 0, "[-q] parameter requires an argument value", 1, -40, INT_MAX, "Invalid minimum length value (-q %s)", -1, 0},
    {'r', FXH_K_CHAR1, MASK_CHAR, 0, "[-r] parameter requires an argument value", 0, 0, 0, "[-r] parameter requires a single character as value", -1, 0},
};
static const fxh_report_line report[] = {
    {FXH_W_ALWAYS, 0, 0, {{"Minimum Quality Threshold: ", FXH_V_SLOT_D, THRESHOLD}, {"\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Low-quality nucleotides replaced with '", FXH_V_SLOT_C, MASK_CHAR}, {"'\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Input: ", FXH_V_IN, 0}, {" reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Output: ", FXH_V_OUT, 0}, {" reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Masked reads: ", FXH_V_MASKED_READS, 0}, {"\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Masked nucleotides: ", FXH_V_MASKED_NT, 0}, {"\n", FXH_V_NONE, 0}}},
};
static void configure(const long *v, const char *s, fxg_params *p)
{
    (void)s;
    p->stages = FXG_STAGE_MASK;
    p->mask_min_quality = (int)v[THRESHOLD];
    p->mask_char = (uint32_t)(unsigned char)v[MASK_CHAR];
}
static const fxh_tool tool = {
    "usage: fastq_masker [-h] [-v] [-q N] [-r C] [-z] [-i INFILE] [-o OUTFILE]\n"
    "MI355X build of the FASTX-Toolkit quality masker (same flags as FASTX Toolkit 0.0.14).\n\n"
    "   -h          this help\n"
    "   -q N        quality threshold: nucleotides with lower quality are masked, default 10\n"
    "   -r C        replacement character, default 'N'\n"
    "   -z          compress output with gzip\n"
    "   -i INFILE   FASTQ input, default stdin\n"
    "   -o OUTFILE  FASTQ output, default stdout\n"
    "   -v          verbose report (to stdout if -o is given, else to stderr)\n\n",
    "q:r:", options, 2, NULL, {10, 'N'}, NULL, FASTQ_ONLY, OUTPUT_SAME_AS_INPUT, NULL, configure, report, 6, NULL,
};
int main(int argc, char *argv[]) { return fxh_tool_main(&tool, argc, argv); }
