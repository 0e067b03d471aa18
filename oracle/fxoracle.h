/*
 * fxoracle.h -- CPU oracle for the fastx_toolkit hot path.   *** TEST INFRASTRUCTURE ONLY ***
 *
 * This is a plain-C restatement of the reference's per-read algorithms (agordon/fastx_toolkit
 * v0.0.14).  It exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
 * check (or time) the HIP engine against an independent formulation of the same rules.  Nothing in
 * the product path (fastx_toolkit_amd/, include/, the CLI tools) may include, link or call it.
 *
 * Parity status: PINNED.  The restatement reproduces (a) every Galaxy known-answer pair the
 * reference ships for the five hot tools (tests/golden/galaxy/) and (b) the
 * outputs of the real reference libfastx (reader, writer, HalfLocalSequenceAlignment compiled from
 * /root/reference into oracle/_ref/) on seeded synthetic and fuzzed inputs; see
 * tests/test_oracle_vs_ref.py and tests/golden/make_golden.py.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 */
#ifndef FXORACLE_H
#define FXORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- stage bits (same numbering as include/fxg.h so tests can share parameter builders) ---- */
#define FXO_STAGE_CLIP      0x01u  /* fastx_clipper            */
#define FXO_STAGE_QTRIM     0x02u  /* fastq_quality_trimmer    */
#define FXO_STAGE_QFILTER   0x04u  /* fastq_quality_filter     */
#define FXO_STAGE_REVCOMP   0x08u  /* fastx_reverse_complement */
#define FXO_STAGE_FTRIM     0x10u  /* fastx_trimmer -f/-l      */
#define FXO_STAGE_FTRIM_END 0x20u  /* fastx_trimmer -t/-m      */
#define FXO_STAGE_MASK      0x40u  /* fastq_masker             */
#define FXO_STAGE_ARTIFACTS 0x80u  /* fastx_artifacts_filter   */
#define FXO_STAGE_NFILTER   0x100u /* fastq_to_fasta N-discard */

#define FXO_CLIP_DISCARD_NON_CLIPPED 0x1u /* -c */
#define FXO_CLIP_DISCARD_CLIPPED     0x2u /* -C */
#define FXO_CLIP_KEEP_N              0x4u /* -n */
#define FXO_CLIP_ADAPTER_ONLY        0x8u /* -k */

/* drop reasons stored in bits 17..20 of a result word */
enum {
    FXO_R_KEPT = 0,
    FXO_R_CLIP_TOO_SHORT = 1,
    FXO_R_CLIP_ADAPTER_ONLY = 2,
    FXO_R_CLIP_NO_ADAPTER = 3,
    FXO_R_CLIP_ADAPTER_FOUND = 4,
    FXO_R_CLIP_N = 5,
    FXO_R_QTRIM = 6,
    FXO_R_QFILTER = 7,
    FXO_R_FTRIM = 8,
    FXO_R_CLIP_K_MODE = 9, /* -k given and the read is not adapter-only */
    FXO_R_ARTIFACT = 10,
    FXO_R_HAS_N = 11
};

/* counters[] slots */
enum {
    FXO_C_INPUT = 0,
    FXO_C_KEPT = 1,
    FXO_C_KEPT_BASES = 2,
    FXO_C_CLIP_TOO_SHORT = 3,
    FXO_C_CLIP_ADAPTER_ONLY = 4,
    FXO_C_CLIP_NO_ADAPTER = 5,
    FXO_C_CLIP_ADAPTER_FOUND = 6,
    FXO_C_CLIP_N = 7,
    FXO_C_QTRIM_DROPPED = 8,
    FXO_C_QFILTER_DROPPED = 9,
    FXO_C_FTRIM_DROPPED = 10,
    FXO_C_CLIP_OUT = 11,
    FXO_C_QTRIM_OUT = 12,
    FXO_C_MASKED_READS = 13,
    FXO_C_MASKED_NT = 14,
    FXO_C_ARTIFACT_DROPPED = 16,
    FXO_NCOUNTERS = 24
};

#define FXO_RES_LEN(w)    ((uint32_t)(w) & 0xFFFFu)
#define FXO_RES_KEEP(w)   (((uint32_t)(w) >> 16) & 1u)
#define FXO_RES_REASON(w) (((uint32_t)(w) >> 17) & 0xFu)
#define FXO_RES_CLIPPED(w) (((uint32_t)(w) >> 21) & 1u)

typedef struct {
    uint32_t stages;
    int32_t  qoffset;            /* -Q (fastx_args.c:43 default 33) */
    /* fastq_quality_trimmer */
    int32_t  qt_threshold;       /* -t */
    int32_t  qt_min_len;         /* -l */
    /* fastq_quality_filter */
    int32_t  qf_min_quality;     /* -q */
    int32_t  qf_min_percent;     /* -p ; 0 = flag omitted */
    /* fastx_clipper */
    char     adapter[100];       /* -a, NUL terminated (fastx_clipper.cpp:40,68) */
    uint32_t clip_min_len;       /* -l, default 5 */
    int32_t  clip_keep_delta;    /* -d N, ALREADY increased by strlen(adapter) when N>0 (fastx_clipper.cpp:153) */
    int32_t  clip_min_adapter_len; /* -M */
    uint32_t clip_flags;
    /* fastx_trimmer */
    int32_t  ft_first;           /* -f, 1-based, default 1 */
    int32_t  ft_last;            /* -l, 0 = keep to the end */
    uint32_t ft_trim_end;        /* -t */
    uint32_t ft_min_len;         /* -m */
    /* fastq_masker */
    int32_t  mask_min_quality;   /* -q */
    uint32_t mask_char;          /* -r */
    uint32_t nf_keep_n;          /* fastq_to_fasta -n */
} fxo_params;

/* Structure-of-arrays batch: row r lives at bases + r*stride, length len[r] (or fixed_len if len==NULL).
 * qual may be NULL for FASTA input. */
typedef struct {
    const uint8_t  *bases;
    const uint8_t  *qual;
    const uint16_t *len;
    uint32_t        fixed_len;
    uint32_t        stride;
    uint64_t        n;
} fxo_batch;

typedef struct {
    uint32_t *res;         /* [n]  new_len | keep<<16 | reason<<17 | clipped<<21 | adapter_only<<22 */
    uint8_t  *out_bases;   /* packed concatenation of kept reads, input order */
    uint8_t  *out_qual;
    uint16_t *out_len;     /* [kept] (may be NULL) */
    uint32_t *kept_index;  /* [kept] (may be NULL) */
    uint64_t  counters[FXO_NCOUNTERS];
} fxo_out;

/* ---- synthetic data (SURVEY.md section 8d spec) ---- */
uint64_t fxo_splitmix(uint64_t x);
void   fxo_synth_read(uint64_t seed, uint64_t r, uint32_t L, int with_adapter, uint8_t *bases, uint8_t *qual);
void   fxo_synth_batch(uint64_t seed, uint64_t first, uint64_t n, uint32_t L, int with_adapter,
                       uint8_t *bases, uint8_t *qual, uint32_t stride);
/* writes "@SYN.<seed>.<r>\n<bases>\n+\n<qual>\n" records; returns bytes written (call with out==NULL to size) */
size_t fxo_synth_fastq(uint64_t seed, uint64_t first, uint64_t n, uint32_t L, int with_adapter, char *out);

/* ---- single-read stage restatements ---- */
int  fxo_qtrim_read(const uint8_t *qual, int len, int qoffset, int threshold, int min_len, int *new_len);
int  fxo_qfilter_read(const uint8_t *qual, int len, int qoffset, int min_quality, int min_percent);

typedef struct {
    int64_t query_size, query_start, query_end;
    int64_t target_size, target_start, target_end;
    int64_t gaps, neutral_matches, matches, mismatches;
    float   score; /* best cell's matrix score */
} fxo_align_res;

/* aligner object: carries the reference's "never shrinks" matrix width and stale query buffer (note N3) */
typedef struct fxo_aligner fxo_aligner;
fxo_aligner *fxo_aligner_new(void);
void         fxo_aligner_free(fxo_aligner *a);
void         fxo_align(fxo_aligner *a, const char *query, int qn, const char *target, int tn, fxo_align_res *res);
int          fxo_adapter_cutoff_index(const fxo_align_res *r, int min_adapter_len);

/* ---- batch pipeline: supported chains are [CLIP][QTRIM][QFILTER], [REVCOMP][FTRIM|FTRIM_END], [MASK], [ARTIFACTS] ---- */
/* returns 0, or -1 on unsupported stage combination, -2 on invalid base for REVCOMP */
int fxo_run_pipeline(const fxo_batch *in, const fxo_params *p, fxo_out *out);
/* same, with the caller's aligner: successive batches of ONE clipper run share its query buffer and matrix (N3) */
int fxo_run_pipeline_h(const fxo_batch *in, const fxo_params *p, fxo_out *out, fxo_aligner *shared);

/* ---- fastx_quality_stats (src/fastx_quality_stats/fastx_quality_stats.c) ---- */
#define FXO_QS_MINQ   (-15)   /* fastx.h:28 MIN_QUALITY_VALUE */
#define FXO_QS_RANGE  108     /* fastx.h:30 QUALITY_VALUES_RANGE */
typedef struct fxo_qstats fxo_qstats;
fxo_qstats *fxo_qstats_new(void);
void        fxo_qstats_free(fxo_qstats *s);
/* read_file (:166-216) over a batch; qual rows hold quality value + qoffset; qual == NULL = FASTA (counts only) */
void        fxo_qstats_add(fxo_qstats *s, const fxo_batch *in, int qoffset);
/* print_old_statistics (:340-414) / print_statistics (:296-334); dst == NULL returns the size only */
size_t      fxo_qstats_format(const fxo_qstats *s, int new_format, char *dst);
/* counts of nucleotide class cls (0 ALL, 1 A, 2 C, 3 G, 4 T, 5 N) at column col, by quality value - FXO_QS_MINQ; returns the class count */
long long   fxo_qstats_hist(const fxo_qstats *s, int col, int cls, int hist[FXO_QS_RANGE]);

/* ---- FASTQ text <-> SoA (reader rules R1-R9, writer a3), ASCII qualities only ---- */
/* Parses up to max_reads records from text; fills rows; names[] gets offsets into text of each '@' line.
 * Returns number of records, or -(line number) on the first invalid record. */
int64_t fxo_parse_fastq(const char *text, size_t text_len, int qoffset, uint64_t max_reads, uint32_t stride,
                        uint8_t *bases, uint8_t *qual, uint16_t *len,
                        uint64_t *name_off, uint32_t *name_len, uint64_t *name2_off, uint32_t *name2_len);
/* Formats kept reads given res[] (prefix/forward outputs) or packed out arrays. Returns bytes written. */
size_t fxo_format_fastq(const char *text, const uint64_t *name_off, const uint32_t *name_len,
                        const uint64_t *name2_off, const uint32_t *name2_len,
                        const uint8_t *out_bases, const uint8_t *out_qual, const uint16_t *out_len,
                        const uint32_t *kept_index, uint64_t kept, char *dst);

#ifdef __cplusplus
}
#endif
#endif
