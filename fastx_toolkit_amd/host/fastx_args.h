/*
 * fastx_args.h -- common command-line flags of the fastx tools ( -h -v -z -i -o -Q ).
 * Same entry points and semantics as the reference's src/libfastx/fastx_args.h:27-38 / fastx_args.c:76-143:
 * tool-specific letters are dispatched to a callback; the -v report goes to stderr unless -o was given.
 */
#ifndef FXH_FASTX_ARGS_H
#define FXH_FASTX_ARGS_H
#include <stdio.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef int (*parse_argument_func)(int optind, int optc, char *optarg);

const char *get_input_filename(void);
const char *get_output_filename(void);
int verbose_flag(void);
int compress_output_flag(void);
int get_fastq_ascii_quality_offset(void);
FILE *get_report_file(void);

/* Every tool defines its own help text. */
extern const char *usage;

int fastx_parse_cmdline(int argc, char *argv[], const char *program_options, parse_argument_func program_parse_arg);

#ifdef __cplusplus
}
#endif
#endif
