/* fastq_quality_trimmer -- command line, output and -v report of the FASTX-Toolkit tool of that name (behaviour:
 * src/fastq_quality_trimmer/fastq_quality_trimmer.c); the 3'-end scan runs on the GPU (FXG_STAGE_QTRIM).  The tool is a table. */
#include <err.h>
#include <limits.h>
#include <string.h>

#include "../fxh_tool.h"

enum { THRESHOLD, MIN_LENGTH };

static const fxh_option options[] = {
    {'t', FXH_K_STRTOL, THRESHOLD, 0, "[-t] parameter requires an argument value", 0, 0, 0, NULL, -1, 0},              /* negatives are accepted (F1) */
    {'l', FXH_K_STRTOUL_INT, MIN_LENGTH, 0, "[-l] parameter requires an argument value", 1, 0, INT_MAX, "Invalid minimum length value (-l %s)", -1, 0},
};
static const fxh_report_line report[] = {
    {FXH_W_ALWAYS, 0, 0, {{"Minimum Quality Threshold: ", FXH_V_SLOT_D, THRESHOLD}, {"\n", FXH_V_NONE, 0}}},
    {FXH_W_POS, MIN_LENGTH, 0, {{"Minimum Length: ", FXH_V_SLOT_D, MIN_LENGTH}, {"\n", FXH_V_NONE, 0}}},
    {FXH_W_NOTPOS, MIN_LENGTH, 0, {{"No minimum Length\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Input: ", FXH_V_IN, 0}, {" reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Output: ", FXH_V_OUT, 0}, {" reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"discarded ", FXH_V_DISCARDED, 0}, {" (", FXH_V_DISCARDED_PCT, 0}, {"%) too-short reads.\n", FXH_V_NONE, 0}}},
};
static void check(const long *v, const char *s) { (void)s; if (v[THRESHOLD] == 0) errx(1, "Missing minimum quality threshold value (-t)"); }
static void configure(const long *v, const char *s, fxg_params *p)
{
    (void)s;
    p->stages = FXG_STAGE_QTRIM;
    p->qt_threshold = (int)v[THRESHOLD];
    p->qt_min_len = (int)v[MIN_LENGTH];
}
static const fxh_tool tool = {
    "usage: fastq_quality_trimmer [-h] [-v] [-t N] [-l N] [-z] [-i INFILE] [-o OUTFILE]\n"
    "MI355X build of the FASTX-Toolkit quality trimmer (same flags as FASTX Toolkit 0.0.14).\n\n"
    "   -h          this help\n"
    "   -t N        quality threshold: trailing nucleotides with lower quality are trimmed\n"
    "   -l N        minimum length: shorter sequences (after trimming) are discarded; default 0\n"
    "   -z          compress output with gzip\n"
    "   -i INFILE   FASTQ input, default stdin\n"
    "   -o OUTFILE  FASTQ output, default stdout\n"
    "   -v          verbose report (to stdout if -o is given, else to stderr)\n"
    "   -Q N        ASCII quality offset, default 33\n\n",
    "t:l:", options, 2, NULL, {0, 0}, NULL, FASTQ_ONLY, OUTPUT_SAME_AS_INPUT, check, configure, report, 6, NULL,
};
int main(int argc, char *argv[]) { return fxh_tool_main(&tool, argc, argv); }
