/* fxh_tool.c -- see fxh_tool.h. */
#include "fxh_tool.h"

#include <err.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "fastx_args.h"

static const fxh_tool *g_tool;
static long g_v[FXH_TOOL_SLOTS];
static char g_s[100];
const char *usage = "";

static int fxh_tool_option(int optind_, int optc, char *arg)
{
    (void)optind_;
    for (int i = 0; i < g_tool->nopts; ++i) {
        const fxh_option *o = &g_tool->opts[i];
        if (o->letter != optc) continue;
        long v = 0;
        if (o->kind != FXH_K_FLAG && o->kind != FXH_K_COUNT && arg == NULL) errx(1, "%s", o->missing ? o->missing : "option requires an argument value");
        switch (o->kind) {
        case FXH_K_FLAG: v = o->value; break;
        case FXH_K_COUNT: v = g_v[o->slot] + 1; break;
        case FXH_K_STRTOL: v = (int)strtol(arg, NULL, 10); break;
        case FXH_K_STRTOUL_INT: v = (int)strtoul(arg, NULL, 10); break;
        case FXH_K_STRTOUL_U32: v = (long)(unsigned int)strtoul(arg, NULL, 10); break;
        case FXH_K_ATOI: v = atoi(arg); break;
        case FXH_K_CHAR1:
            if (strlen(arg) != 1) errx(1, "%s", o->range_fmt);
            v = (unsigned char)arg[0];
            break;
        case FXH_K_STRING:
            strncpy(g_s, arg, sizeof g_s - 1); g_s[sizeof g_s - 1] = 0;
            return 1;
        }
        if (o->ranged && (v < o->lo || v > o->hi)) errx(1, o->range_fmt, arg);
        g_v[o->slot] = v;
        if (o->also_slot >= 0) g_v[o->also_slot] = o->also_value;
        return 1;
    }
    errx(1, g_tool->unknown_fmt ? g_tool->unknown_fmt : "Unknown argument (%c)", optc);
    return 0;
}

static void fxh_put_value(FILE *f, fxh_value val, int arg, const fxh_totals *t)
{
    const size_t discarded = t->input_reads - t->output_reads;
    switch (val) {
    case FXH_V_NONE: break;
    case FXH_V_SLOT_D: fprintf(f, "%d", (int)g_v[arg]); break;
    case FXH_V_SLOT_C: fprintf(f, "%c", (int)g_v[arg]); break;
    case FXH_V_STRING: fputs(g_s, f); break;
    case FXH_V_IN: fprintf(f, "%zu", t->input_reads); break;
    case FXH_V_OUT: fprintf(f, "%zu", t->output_reads); break;
    case FXH_V_DISCARDED: fprintf(f, "%zu", discarded); break;
    case FXH_V_DISCARDED_PCT: fprintf(f, "%zu", (discarded * 100) / t->input_reads); break;
    case FXH_V_MASKED_READS: fprintf(f, "%zu", t->masked_reads); break;
    case FXH_V_MASKED_NT: fprintf(f, "%zu", t->masked_nucleotides); break;
    case FXH_V_CLIP_IN: fprintf(f, "%u", t->clip_input); break;
    case FXH_V_CLIP_OUT: fprintf(f, "%u", t->clip_input - t->clip_too_short - t->clip_no_adapter - t->clip_adapter_found - t->clip_n - t->clip_adapter_only); break;
    case FXH_V_CLIP_SHORT: fprintf(f, "%u", t->clip_too_short); break;
    case FXH_V_CLIP_ADAPTER_ONLY: fprintf(f, "%u", t->clip_adapter_only); break;
    case FXH_V_CLIP_NON_CLIPPED: fprintf(f, "%u", t->clip_no_adapter); break;
    case FXH_V_CLIP_CLIPPED: fprintf(f, "%u", t->clip_adapter_found); break;
    case FXH_V_CLIP_N: fprintf(f, "%u", t->clip_n); break;
    }
}

static int fxh_when_holds(const fxh_report_line *l)
{
    switch (l->when) {
    case FXH_W_ALWAYS: return 1;
    case FXH_W_NZ: return g_v[l->a] != 0;
    case FXH_W_Z: return g_v[l->a] == 0;
    case FXH_W_POS: return g_v[l->a] > 0;
    case FXH_W_NOTPOS: return g_v[l->a] <= 0;
    case FXH_W_NZ_BOTH: return g_v[l->a] != 0 && g_v[l->b] != 0;
    case FXH_W_RANGE_SET: return g_v[l->a] != 1 || g_v[l->b] != 0;       /* a first/last range other than "whole read" */
    }
    return 0;
}

int fxh_tool_main(const fxh_tool *tool, int argc, char *argv[])
{
    static FASTX fastx;
    fxh_totals tot;
    fxg_params p;
    g_tool = tool;
    usage = tool->usage;
    memcpy(g_v, tool->defaults, sizeof g_v);
    strncpy(g_s, tool->default_string ? tool->default_string : "", sizeof g_s - 1);
    fastx_parse_cmdline(argc, argv, tool->optstring, tool->nopts || tool->optstring[0] ? fxh_tool_option : NULL);
    if (tool->check) tool->check(g_v, g_s);
    fastx_init_reader(&fastx, get_input_filename(), tool->input_types, ALLOW_N, REQUIRE_UPPERCASE, get_fastq_ascii_quality_offset());
    fxh_init_writer(&fastx, get_output_filename(), tool->output_type, compress_output_flag());
    fxh_default_params(&p, get_fastq_ascii_quality_offset());
    tool->configure(g_v, g_s, &p);
    if (!(tool->alt_run && tool->alt_run(g_v, &fastx, &p, &tot))) fxh_run_tool(&fastx, &p, &tot);
    if (verbose_flag()) {
        FILE *rf = get_report_file();
        for (int i = 0; i < tool->nreport; ++i) {
            const fxh_report_line *l = &tool->report[i];
            if (!fxh_when_holds(l)) continue;
            for (int k = 0; k < 3; ++k) {
                if (l->seg[k].text) fputs(l->seg[k].text, rf);
                fxh_put_value(rf, l->seg[k].val, l->seg[k].arg, &tot);
            }
        }
    }
    fastx_finish(&fastx);
    /* Everything is written and flushed.  Returning would run the HIP runtime's exit handlers (tens of milliseconds of
     * teardown for a process that is gone anyway); FXH_SLOW_EXIT=1 takes that path, for leak checkers. */
    fflush(NULL);
    if (getenv("FXH_TIMING")) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); fprintf(stderr, "fxh timing exit: _exit at %.3f (CLOCK_MONOTONIC)\n", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec); }
    if (!getenv("FXH_SLOW_EXIT")) _exit(0);
    return 0;
}
