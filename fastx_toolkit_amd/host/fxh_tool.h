/*
 * fxh_tool.h -- one description-driven main() for the per-read tools.
 *
 * Every tool of this family is the same program around a different stage chain: parse the common flags and a handful of
 * tool letters, open reader and writer, run the engine over the input, print a short report.  Here a tool is DATA -- its
 * option table, its report table and two small hooks (cross-option checks, filling fxg_params) -- and fxh_tool_main() is
 * the program.  What the tables reproduce from the reference tools is observable behaviour only: letters, number parsing
 * quirks (F1-F6 of SURVEY.md 8a), messages, report text and exit codes.
 */
#ifndef FXH_TOOL_H
#define FXH_TOOL_H
#include "fastx.h"
#include "fxh_batch.h"

#define FXH_TOOL_SLOTS 8

typedef enum {
    FXH_K_FLAG,        /* no argument: v[slot] = value */
    FXH_K_COUNT,       /* no argument: v[slot] += 1 (fastx_clipper -D -D) */
    FXH_K_STRTOL,      /* (int)strtol(arg)   -- negatives accepted */
    FXH_K_STRTOUL_INT, /* (int)strtoul(arg)  -- the reference stores strtoul's result in an int */
    FXH_K_STRTOUL_U32, /* (unsigned int)strtoul(arg) */
    FXH_K_ATOI,        /* atoi(arg) */
    FXH_K_CHAR1,       /* exactly one character */
    FXH_K_STRING       /* copied into the tool's string value (up to 99 characters) */
} fxh_opt_kind;

typedef struct fxh_option {
    char letter;
    fxh_opt_kind kind;
    int slot;
    long value;                 /* FXH_K_FLAG: what to store */
    const char *missing;        /* errx text when the argument is missing */
    int ranged;                 /* 1: errx(range_fmt, arg) unless lo <= v <= hi */
    long lo, hi;
    const char *range_fmt;
    int also_slot; long also_value;   /* a second slot the option sets (-1: none) */
} fxh_option;

/* values a report can print; each carries its own conversion */
typedef enum {
    FXH_V_NONE = 0,
    FXH_V_SLOT_D,      /* v[arg] as %d */
    FXH_V_SLOT_C,      /* v[arg] as %c */
    FXH_V_STRING,      /* the tool's string value */
    FXH_V_IN, FXH_V_OUT, FXH_V_DISCARDED, FXH_V_DISCARDED_PCT,          /* %zu: reads in / out / in - out / percent of in */
    FXH_V_MASKED_READS, FXH_V_MASKED_NT,                                /* %zu */
    FXH_V_CLIP_IN, FXH_V_CLIP_OUT, FXH_V_CLIP_SHORT, FXH_V_CLIP_ADAPTER_ONLY, FXH_V_CLIP_NON_CLIPPED, FXH_V_CLIP_CLIPPED, FXH_V_CLIP_N   /* %u */
} fxh_value;

typedef enum { FXH_W_ALWAYS, FXH_W_NZ, FXH_W_Z, FXH_W_POS, FXH_W_NOTPOS, FXH_W_NZ_BOTH, FXH_W_RANGE_SET } fxh_when;

typedef struct fxh_seg { const char *text; fxh_value val; int arg; } fxh_seg;
typedef struct fxh_report_line {
    fxh_when when; int a, b;    /* condition over v[a] (and v[b]) */
    fxh_seg seg[3];             /* text, value, text, value, text */
} fxh_report_line;

typedef struct fxh_tool {
    const char *usage, *optstring;
    const fxh_option *opts; int nopts;
    const char *unknown_fmt;                    /* errx format (%c) for a letter of optstring without an entry */
    long defaults[FXH_TOOL_SLOTS];
    const char *default_string;
    ALLOWED_INPUT_FILE_TYPES input_types;
    OUTPUT_FILE_TYPE output_type;
    void (*check)(const long *v, const char *s);                        /* cross-option rules; may errx */
    void (*configure)(const long *v, const char *s, fxg_params *p);     /* stage chain and parameters */
    const fxh_report_line *report; int nreport;
    int (*alt_run)(const long *v, FASTX *fx, const fxg_params *p, fxh_totals *tot);      /* optional: a mode of the tool that is not an engine run (1 = it ran) */
} fxh_tool;

int fxh_tool_main(const fxh_tool *tool, int argc, char *argv[]);
#endif
