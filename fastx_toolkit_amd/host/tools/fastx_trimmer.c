/* fastx_trimmer -- command line, output and -v report of the FASTX-Toolkit tool of that name (behaviour:
 * src/fastx_trimmer/fastx_trimmer.c); positions are arithmetic on the read length, done on the GPU (FXG_STAGE_FTRIM / _END). */
#include <err.h>
#include <limits.h>
#include <string.h>

#include "../fxh_tool.h"

enum { FIRST, LAST, BY_POSITION, FROM_END, TRIM_END, MIN_LEN };
#define MAXLEN (MAX_SEQ_LINE_LENGTH - 1)

static const fxh_option options[] = {
    {'f', FXH_K_STRTOUL_INT, FIRST, 0, "[-f] parameter requires an argument value", 1, 1, MAXLEN, "Invalid number bases to keep (-f %s)", BY_POSITION, 1},
    {'l', FXH_K_STRTOUL_INT, LAST, 0, "[-l] parameter requires an argument value", 1, 1, MAXLEN, "Invalid number bases to keep (-l %s)", BY_POSITION, 1},
    {'t', FXH_K_STRTOUL_U32, TRIM_END, 0, "[-t] parameter requires an argument value", 1, 1, MAXLEN, "Invalid number bases to trim (-t %s)", FROM_END, 1},
    {'m', FXH_K_STRTOUL_U32, MIN_LEN, 0, "[-t] parameter requires an argument value", 1, 1, MAXLEN, "Invalid minimum length value (-m %s)", -1, 0},   /* "[-t]": the reference's wording */
};
static const fxh_report_line report[] = {
    {FXH_W_RANGE_SET, FIRST, LAST, {{"Trimming: base ", FXH_V_SLOT_D, FIRST}, {" to ", FXH_V_SLOT_D, LAST}, {"\n", FXH_V_NONE, 0}}},
    {FXH_W_NZ, TRIM_END, 0, {{"Trimming ", FXH_V_SLOT_D, TRIM_END}, {" bases from the end of the reads\n", FXH_V_NONE, 0}}},
    {FXH_W_NZ_BOTH, TRIM_END, MIN_LEN, {{"Discarding reads shorter than ", FXH_V_SLOT_D, MIN_LEN}, {" bases\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Input: ", FXH_V_IN, 0}, {" reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Output: ", FXH_V_OUT, 0}, {" reads.\n", FXH_V_NONE, 0}}},
};
static void check(const long *v, const char *s)
{
    (void)s;
    if (v[BY_POSITION] && v[FROM_END]) errx(1, "[-t], [-f] and [-l] options can not be used together. Use [-t] or [-l,-f]");   /* F4 */
}
static void configure(const long *v, const char *s, fxg_params *p)
{
    (void)s;
    if (v[FROM_END]) {
        p->stages = FXG_STAGE_FTRIM_END;
        p->ft_trim_end = (uint32_t)v[TRIM_END];
        p->ft_min_len = (uint32_t)v[MIN_LEN];        /* -m without -t has no effect in the reference either */
    } else {
        p->stages = FXG_STAGE_FTRIM;
        p->ft_first = (int)v[FIRST];
        p->ft_last = (int)v[LAST];
    }
}
static const fxh_tool tool = {
    "usage: fastx_trimmer [-h] [-f N] [-l N] [-t N] [-m MINLEN] [-z] [-v] [-i INFILE] [-o OUTFILE]\n"
    "MI355X build of the FASTX-Toolkit trimmer (same flags as FASTX Toolkit 0.0.14).\n\n"
    "   -h          this help\n"
    "   -f N        first base to keep, default 1\n"
    "   -l N        last base to keep, default the entire read\n"
    "   -t N        trim N nucleotides from the end of the read (not together with -f / -l)\n"
    "   -m MINLEN   with -t: discard reads shorter than MINLEN\n"
    "   -z          compress output with gzip\n"
    "   -i INFILE   FASTA/Q input, default stdin\n"
    "   -o OUTFILE  FASTA/Q output, default stdout\n\n",
    "l:f:t:m:", options, 4, NULL, {1, 0, 0, 0, 0, 0}, NULL, FASTA_OR_FASTQ, OUTPUT_SAME_AS_INPUT, check, configure, report, 5, NULL,
};
int main(int argc, char *argv[]) { return fxh_tool_main(&tool, argc, argv); }
