/*
 * fxh_batch.h -- batch path of the host layer: records are parsed in bulk from the block-buffered reader,
 * packed into a Structure-of-Arrays batch, pushed through the HIP engine (include/fxg.h) and the kept
 * records are formatted back in input order.  This replaces the reference tools'
 *     while (fastx_read_next_record(&fastx)) { <loop body>; fastx_write_record(&fastx); }
 * (e.g. fastq_quality_trimmer.c:89-104) with one engine call per few hundred thousand records; the loop
 * bodies themselves run on the GPU.  There is no CPU implementation of the loop bodies in this layer.
 */
#ifndef FXH_BATCH_H
#define FXH_BATCH_H
#include <stddef.h>
#include <stdint.h>

#include "../../include/fxg.h"
#include "fastx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Totals for the tools' -v reports (a12).  *_reads are weighted by get_reads_count() like the reference's. */
typedef struct fxh_totals {
    size_t input_sequences, input_reads, output_sequences, output_reads;
    /* fastx_clipper.cpp:80-85 (32-bit unsigned there; printed with %u) */
    unsigned int clip_input, clip_too_short, clip_adapter_only, clip_no_adapter, clip_adapter_found, clip_n;
    /* fastq_masker.c:78-79 */
    size_t masked_reads, masked_nucleotides;
    /* chains of quality stages in one pass (fastq_quality_trim_filter): reads the trimmer stage dropped */
    size_t qtrim_dropped;
} fxh_totals;

/* Runs the whole input of `fx` (reader and writer already initialised) through the engine with the stage
 * chain in `p`.  On a malformed record every earlier record is processed and written first, then the
 * reference's message is printed and the process exits with status 1 (same observable order as errx()
 * inside fastx_read_next_record).  Returns 0. */
int fxh_run_tool(FASTX *fx, const fxg_params *p, fxh_totals *totals);

/* fastx_init_writer for the batch tools: the same, except that "%r" in the name stands for the part number of a sharded run (FXH_PARTS, or parts
 * chosen by the tool) -- this writer is then part 0.  fastx_init_writer itself opens the literal name, as the reference does (fastx.c:251-271). */
void fxh_init_writer(FASTX *fx, const char *filename, OUTPUT_FILE_TYPE output_type, int compress_output);

/* fastx_quality_stats over the whole FASTQ input of `fx` (reader initialised, no writer needed): the per-column histogram of
 * include/fxg.h's fxg_run_quality_stats summed over all batches, copied to a malloc'ed array hist[cols][FXG_QS_CLASSES][FXG_QS_BINS]
 * whose bin index is quality value + 33 whatever -Q was.  Malformed input ends the process like fxh_run_tool. */
int fxh_run_quality_stats(FASTX *fx, uint64_t **hist, uint32_t *cols, fxh_totals *totals);

/* fastq_to_fasta -r: kept records are renamed to their 1-based output index (fastq_to_fasta.c:83-84).  Set before fxh_run_tool. */
void fxh_set_rename_ids(int on);

/* fxg_params with the reference tools' defaults (qoffset from -Q). */
void fxh_default_params(fxg_params *p, int qoffset);

#ifdef __cplusplus
}
#endif
#endif
