/* fastq_quality_filter -- command line, output and -v report of the FASTX-Toolkit tool of that name (behaviour:
 * src/fastq_quality_filter/fastq_quality_filter.c); the percentile test runs on the GPU in closed form (FXG_STAGE_QFILTER). */
#include <err.h>
#include <limits.h>
#include <string.h>

#include "../fxh_tool.h"

enum { MIN_QUALITY, MIN_PERCENT };

static const fxh_option options[] = {
    {'q', FXH_K_STRTOUL_INT, MIN_QUALITY, 0, "[-q] parameter requires an argument value", 0, 0, 0, NULL, -1, 0},
    {'p', FXH_K_STRTOUL_INT, MIN_PERCENT, 0, "[-l] parameter requires an argument value", 1, 1, 100, "Invalid percent value (-p %s)", -1, 0},   /* "[-l]": the reference's wording */
};
static const fxh_report_line report[] = {
    {FXH_W_ALWAYS, 0, 0, {{"Quality cut-off: ", FXH_V_SLOT_D, MIN_QUALITY}, {"\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Minimum percentage: ", FXH_V_SLOT_D, MIN_PERCENT}, {"\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Input: ", FXH_V_IN, 0}, {" reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Output: ", FXH_V_OUT, 0}, {" reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"discarded ", FXH_V_DISCARDED, 0}, {" (", FXH_V_DISCARDED_PCT, 0}, {"%) low-quality reads.\n", FXH_V_NONE, 0}}},
};
static void configure(const long *v, const char *s, fxg_params *p)
{
    (void)s;
    p->stages = FXG_STAGE_QFILTER;
    p->qf_min_quality = (int)v[MIN_QUALITY];
    p->qf_min_percent = (int)v[MIN_PERCENT];      /* 0 when -p was not given: every read passes unless -q > 93 (quirk F2) */
}
static const fxh_tool tool = {
    "usage: fastq_quality_filter [-h] [-v] [-q N] [-p N] [-z] [-i INFILE] [-o OUTFILE]\n"
    "MI355X build of the FASTX-Toolkit quality filter (same flags as FASTX Toolkit 0.0.14).\n\n"
    "   -h          this help\n"
    "   -q N        minimum quality score to keep\n"
    "   -p N        minimum percent of bases that must have [-q] quality\n"
    "   -z          compress output with gzip\n"
    "   -i INFILE   FASTQ input, default stdin\n"
    "   -o OUTFILE  FASTQ output, default stdout\n"
    "   -v          verbose report (to stdout if -o is given, else to stderr)\n"
    "   -Q N        ASCII quality offset, default 33\n\n",
    "q:p:", options, 2, NULL, {0, 0}, NULL, FASTQ_ONLY, OUTPUT_SAME_AS_INPUT, NULL, configure, report, 5, NULL,
};
int main(int argc, char *argv[]) { return fxh_tool_main(&tool, argc, argv); }
