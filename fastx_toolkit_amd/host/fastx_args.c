/* fastx_args.c -- see fastx_args.h.  Quirks kept on purpose: -Q goes through atoi(),
 * -h prints the usage text to stdout and exits with status 1, an unknown flag prints a hint and exits 1,
 * and choosing -o moves the verbose report from stderr to stdout. */
#include "fastx_args.h"

#include <err.h>
#include <getopt.h>
#include <stdlib.h>
#include <string.h>

static struct {
    const char *in, *out;
    int verbose, gzip, qoffset;
    FILE *report;
} g_args = {"-", "-", 0, 0, 33, NULL};

const char *get_input_filename(void) { return g_args.in; }
const char *get_output_filename(void) { return g_args.out; }
int verbose_flag(void) { return g_args.verbose; }
int compress_output_flag(void) { return g_args.gzip; }
int get_fastq_ascii_quality_offset(void) { return g_args.qoffset; }
FILE *get_report_file(void) { return g_args.report ? g_args.report : stderr; }


int fastx_parse_cmdline(int argc, char *argv[], const char *program_options, parse_argument_func program_parse_arg)
{
    char spec[128];
    int c;
    snprintf(spec, sizeof spec, "Q:zhvi:o:%s", program_options);
    g_args.report = stderr;
    while ((c = getopt(argc, argv, spec)) != -1) {
        if (c != ':' && strchr(program_options, c) != NULL) {   /* tool-specific letter */
            if (!program_parse_arg(optind, c, optarg)) return 0;
            continue;
        }
        switch (c) {
        case 'h': fputs(usage, stdout); exit(1);
        case 'v': g_args.verbose = 1; break;
        case 'z': g_args.gzip = 1; break;
        case 'i':
            if (!optarg) errx(1, "[-i] option requires FILENAME argument");
            g_args.in = optarg;
            break;
        case 'o':
            if (!optarg) errx(1, "[-o] option requires FILENAME argument");
            g_args.out = optarg;
            g_args.report = stdout;
            break;
        case 'Q':
            if (!optarg) errx(1, "[-Q] option requires VALUE argument");
            g_args.qoffset = atoi(optarg);
            break;
        default:
            printf("use '-h' for usage information.\n");
            exit(1);
        }
    }
    return 1;
}
