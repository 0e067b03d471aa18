/* fastx_clipper -- command line, output and -v report of the FASTX-Toolkit tool of that name (behaviour:
 * src/fastx_clipper/fastx_clipper.cpp); alignment, accept rules and the discard cascade run on the GPU (FXG_STAGE_CLIP). */
#include <err.h>
#include <limits.h>
#include <string.h>

#include "../fxh_tool.h"

enum { MIN_LENGTH, KEEP_N, KEEP_DELTA, ONLY_CLIPPED, ONLY_NON_CLIPPED, ADAPTER_ONLY, MIN_ADAPTER, DEBUG_DUMP };

static const fxh_option options[] = {
    {'M', FXH_K_ATOI, MIN_ADAPTER, 0, "[-M] parameter requires an argument value", 1, 1, INT_MAX, "Invalid minimum adapter length (-M %s)", -1, 0},
    {'k', FXH_K_FLAG, ADAPTER_ONLY, 1, NULL, 0, 0, 0, NULL, -1, 0},
    {'D', FXH_K_COUNT, DEBUG_DUMP, 1, NULL, 0, 0, 0, NULL, -1, 0},          /* -D, -D -D: the reference's debug++ (fastx_clipper.cpp:130) */
    {'c', FXH_K_FLAG, ONLY_CLIPPED, 1, NULL, 0, 0, 0, NULL, -1, 0},
    {'C', FXH_K_FLAG, ONLY_NON_CLIPPED, 1, NULL, 0, 0, 0, NULL, -1, 0},
    {'d', FXH_K_STRTOUL_INT, KEEP_DELTA, 0, "[-d] parameter requires an argument value", 1, 0, INT_MAX, "Invalid number bases to keep (-d %s)", -1, 0},
    {'a', FXH_K_STRING, 0, 0, "[-a] parameter requires an argument value", 0, 0, 0, NULL, -1, 0},
    {'l', FXH_K_STRTOUL_U32, MIN_LENGTH, 0, "[-l] parameter requires an argument value", 0, 0, 0, NULL, -1, 0},
    {'n', FXH_K_FLAG, KEEP_N, 1, NULL, 0, 0, 0, NULL, -1, 0},
};
static const fxh_report_line report[] = {
    {FXH_W_ALWAYS, 0, 0, {{"Clipping Adapter: ", FXH_V_STRING, 0}, {"\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Min. Length: ", FXH_V_SLOT_D, MIN_LENGTH}, {"\n", FXH_V_NONE, 0}}},
    {FXH_W_NZ, ONLY_NON_CLIPPED, 0, {{"Clipped reads - discarded.\n", FXH_V_NONE, 0}}},
    {FXH_W_NZ, ONLY_CLIPPED, 0, {{"Non-Clipped reads - discarded.\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Input: ", FXH_V_CLIP_IN, 0}, {" reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"Output: ", FXH_V_CLIP_OUT, 0}, {" reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"discarded ", FXH_V_CLIP_SHORT, 0}, {" too-short reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_ALWAYS, 0, 0, {{"discarded ", FXH_V_CLIP_ADAPTER_ONLY, 0}, {" adapter-only reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_NZ, ONLY_CLIPPED, 0, {{"discarded ", FXH_V_CLIP_NON_CLIPPED, 0}, {" non-clipped reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_NZ, ONLY_NON_CLIPPED, 0, {{"discarded ", FXH_V_CLIP_CLIPPED, 0}, {" clipped reads.\n", FXH_V_NONE, 0}}},
    {FXH_W_Z, KEEP_N, 0, {{"discarded ", FXH_V_CLIP_N, 0}, {" N reads.\n", FXH_V_NONE, 0}}},
};
/* -D makes the reference print every read's alignment (and with -D -D its whole score matrix) to stdout, in between the records
 * (fastx_clipper.cpp:272-275, sequence_alignment.cpp:15-84, :169-230): a developer's view of the CPU aligner's state, which the engine does not
 * have (no matrix, no alignment strings).  With -D the tool therefore runs the reference's own record loop with one aligner on the host, which
 * exists for this dump alone (host/fxh_clip_debug.c); every other run of the tool is the GPU path. */
void fxh_clipper_debug_run(FASTX *fx, const fxg_params *p, int level, fxh_totals *tot);
static int debug_dump(const long *v, FASTX *fx, const fxg_params *p, fxh_totals *tot)
{
    if (v[DEBUG_DUMP] <= 0) return 0;
    fxh_clipper_debug_run(fx, p, (int)v[DEBUG_DUMP], tot);
    return 1;
}
static void configure(const long *v, const char *s, fxg_params *p)
{
    p->stages = FXG_STAGE_CLIP;
    strncpy(p->adapter, s, sizeof p->adapter - 1);
    p->clip_min_len = (uint32_t)v[MIN_LENGTH];
    p->clip_keep_delta = v[KEEP_DELTA] > 0 ? (int)v[KEEP_DELTA] + (int)strlen(s) : (int)v[KEEP_DELTA];       /* fastx_clipper.cpp:153-154 */
    p->clip_min_adapter_len = (int)v[MIN_ADAPTER];
    p->clip_flags = (v[ONLY_CLIPPED] ? FXG_CLIP_DISCARD_NON_CLIPPED : 0u) | (v[ONLY_NON_CLIPPED] ? FXG_CLIP_DISCARD_CLIPPED : 0u) |
                    (v[KEEP_N] ? FXG_CLIP_KEEP_N : 0u) | (v[ADAPTER_ONLY] ? FXG_CLIP_ADAPTER_ONLY : 0u);
}
static const fxh_tool tool = {
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
    "   -D          debug dump of every read's alignment to stdout (twice: with the score matrix); runs on the host\n"
    "   -M N        require a minimum adapter alignment length of N\n"
    "   -i INFILE   FASTA/Q input, default stdin\n"
    "   -o OUTFILE  FASTA/Q output, default stdout\n\n",
    "M:kDCcd:a:s:l:n", options, 9, "Unknown argument (%c)",     /* 's' is in the option string but has no handler, there as here (F3) */
    {5, 0, 0, 0, 0, 0, 0}, "CCTTAAGG", FASTA_OR_FASTQ, OUTPUT_SAME_AS_INPUT, NULL, configure, report, 11, debug_dump,
};
int main(int argc, char *argv[]) { return fxh_tool_main(&tool, argc, argv); }
