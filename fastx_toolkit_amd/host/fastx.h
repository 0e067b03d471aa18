/*
 * fastx.h -- record API of the MI355X fastx engine's host layer.
 *
 * Source-compatible with the reference's libfastx record API (reference src/libfastx/fastx.h:120-142):
 * the same function names, argument meaning, enums and the FASTX fields that tool code touches
 * (name, nucleotides, name2, quality[], read_fastq, ...), so a per-record caller
 *
 *      while (fastx_read_next_record(&fastx)) { ...edit fastx.nucleotides...; fastx_write_record(&fastx); }
 *
 * keeps working unchanged.  The implementation (fastx_io.c) is new: block-buffered reader and writer
 * instead of fgets()/fprintf("%c"), shared with the batch path (fxh_batch.h) that feeds the HIP engine.
 * Behavioural rules reproduced from the reference are listed as R1-R9 in SURVEY.md section 8(a).
 */
#ifndef FXH_FASTX_H
#define FXH_FASTX_H

#include <limits.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIN_QUALITY_VALUE (-15)                   /* reference fastx.h:28-30 */
#define MAX_QUALITY_VALUE 93
#define QUALITY_VALUES_RANGE (MAX_QUALITY_VALUE - MIN_QUALITY_VALUE)

#ifndef MAX_SEQ_LINE_LENGTH
#define MAX_SEQ_LINE_LENGTH (25000)               /* reference fastx.h:33-35 */
#endif

typedef enum { FASTA_ONLY = 0, FASTA_OR_FASTQ = 1, FASTQ_ONLY = 2 } ALLOWED_INPUT_FILE_TYPES;
typedef enum { DISALLOW_N = 0, ALLOW_N = 1, ALLOW_U = 2 } ALLOWED_INPUT_BASES;
typedef enum { REQUIRE_UPPERCASE = 0, ALLOW_LOWERCASE = 1 } ALLOWED_INPUT_CASE;
typedef enum {
    OUTPUT_FASTA = 0,
    OUTPUT_FASTQ_ASCII_QUAL = 1,
    OUTPUT_FASTQ_NUMERIC_QUAL = 2,
    OUTPUT_SAME_AS_INPUT = 3
} OUTPUT_FILE_TYPE;

struct fxh_reader;   /* block-buffered line source   (fastx_io.c) */
struct fxh_writer;   /* block-buffered record sink   (fastx_io.c) */

typedef struct FASTX {
    /* ---- the current record (callers read and edit these in place) ---- */
    char name[MAX_SEQ_LINE_LENGTH + 1];          /* id line without its '@' / '>' */
    char nucleotides[MAX_SEQ_LINE_LENGTH + 1];
    char name2[MAX_SEQ_LINE_LENGTH + 1];         /* '+' line without its first byte (not validated, R5) */
    int  quality[MAX_SEQ_LINE_LENGTH + 1];       /* numeric scores, -15..93 */

    /* ---- configuration / state with the reference's meaning ---- */
    int allow_input_filetype;
    int allow_N, allow_U, allow_lowercase;
    int read_fastq;                              /* 1 = input is FASTQ */
    int read_fastq_ascii;                        /* encoding of the LAST record read (R6) */
    int write_fastq, write_fastq_ascii;
    int compress_output;
    int copy_input_fastq_format_to_output;
    int fastq_ascii_quality_offset;
    char output_sequence_id_prefix;
    char input_file_name[PATH_MAX];
    char output_file_name[PATH_MAX];
    unsigned long long input_line_number;
    size_t num_input_sequences, num_output_sequences, num_input_reads, num_output_reads;

    /* ---- implementation ---- */
    unsigned char allowed_nucleotides[256];
    struct fxh_reader *reader;
    struct fxh_writer *writer;
} FASTX;

void fastx_init_reader(FASTX *pFASTX, const char *filename, ALLOWED_INPUT_FILE_TYPES allowed_input_filetype,
                       ALLOWED_INPUT_BASES allow_bases, ALLOWED_INPUT_CASE allow_lowercase, int fastq_ascii_quality_offset);
void fastx_init_writer(FASTX *pFASTX, const char *filename, OUTPUT_FILE_TYPE output_type, int compress_output);
int  fastx_read_next_record(FASTX *pFASTX);      /* 1 = record read, 0 = end of input at a record boundary */
void fastx_write_record(FASTX *pFASTX);
int  get_reads_count(const FASTX *pFASTX);       /* collapsed FASTA ids "N-count" -> count, else 1 */
size_t num_input_sequences(const FASTX *pFASTX);
size_t num_input_reads(const FASTX *pFASTX);
size_t num_output_sequences(const FASTX *pFASTX);
size_t num_output_reads(const FASTX *pFASTX);

/* Flushes buffered output and waits for the gzip child (the reference relies on exit(); callers of this
 * implementation must call it, or use fxh_run_tool which does). */
void fastx_finish(FASTX *pFASTX);

void chomp(char *string);                        /* cut at the first CR or LF (reference chomp.c:34-44) */

#ifdef __cplusplus
}
#endif
#endif
