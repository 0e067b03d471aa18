/* fxh_priv.h -- shared between the files of the batch path (fxh_batch.c: host-parsed path, run driver; fxh_io.c: block prefetch and writer thread;
 * fxh_lanes.c: the lanes of the device text path; fxh_parts.c: the sharded run).  Not installed. */
#ifndef FXH_PRIV_H
#define FXH_PRIV_H
#define _GNU_SOURCE
#include "fxh_batch.h"
#include <err.h>
#include <fcntl.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <errno.h>
#include <pthread.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <sys/prctl.h>
#include <sched.h>
#include <signal.h>
#include <unistd.h>

#include "fxh_internal.h"

static inline double fxh_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    const char *name, *seq, *name2, *qual;
    uint32_t name_len, seq_len, name2_len, qual_len;
    uint32_t reads_count;
    uint8_t is_ascii;
} fxh_rec;

typedef struct {
    fxg_ctx *ctx;
    /* host side (pinned) */
    uint8_t *h_bases, *h_qual;
    uint16_t *h_len;
    uint32_t *h_res;
    uint8_t *h_out_bases, *h_out_qual;
    size_t h_cap_bytes, h_cap_reads;
    /* device side */
    uint8_t *d_bases, *d_qual, *d_out_bases, *d_out_qual;
    uint16_t *d_len;
    uint32_t *d_res;
    uint64_t *d_counters;
    size_t d_cap_bytes, d_cap_reads;
    /* device text path (8f-1) */
    uint8_t *d_text, *d_out_text, *d_flags;
    uint32_t *d_ls;                        /* line starts [d_ls_cap] then line ends [d_ls_cap] */
    uint16_t *d_len16;
    uint64_t *d_out_off;
    size_t d_text_cap, d_ls_cap, d_off_cap;
    const void *registered[4];
    int recoveries_seen;                   /* fxg_scan_recoveries at the last look (fxh_note_recoveries) */
} fxh_state;

#define FXG_CHECK(st, call)                                                                     \
    do {                                                                                        \
        int rc__ = (call);                                                                      \
        if (rc__ != 0) errx(1, "GPU engine error %d: %s", rc__, fxg_last_error((st)->ctx));   \
    } while (0)

/* worker threads: every host phase (index+validate, pack, format) is split by record range */
typedef struct fxh_job fxh_job;
typedef struct fxh_worker {
    int id;
    fxh_job *job;
    FASTX *shadow;                 /* private parser state; reads the shared buffer through `view` */
    struct fxh_reader view;
    struct fxh_rawrec raw;
    size_t a0, a1, nl_count, first_nl;     /* newline census of the raw byte range [a0, a1) */
    size_t start;                          /* first record boundary at or after a0 */
    unsigned long long start_line;         /* lines before `start` (absolute input line numbering) */
    fxh_rec *rec;
    size_t nrec, rec_cap, maxlen, minlen;
    int rc_end;                            /* why indexing stopped: 0 end of input, -1 range/buffer end, -2 error */
    size_t end_pos;
    unsigned long long end_line;
    char errmsg[768];
    long bad_q;                            /* local index of the first record with an invalid quality line, or -1 */
    size_t rec0, use;                      /* global index of rec[0]; how many of this worker's records are in the batch */
    size_t out_bytes, out_off, kept_bytes, kept_off;
    size_t kept_count, kept_base;          /* kept records in this range; output index of its first kept record (1-based) */
    fxh_totals tot;
} fxh_worker;

struct fxh_job {
    FASTX *fx;
    fxh_state *st;
    const fxg_params *p;
    int nworkers, has_q, revcomp, lpr;
    uint32_t stride, fwd_start;
    char *out_dst;
    void (*phase)(fxh_worker *);
    fxh_worker *w;
};

/* block prefetch and writer thread (fxh_io.c) */
#define FXH_GAP_MAX ((size_t)1 << 20)  /* room in front of a prefetched block for the previous block's unread tail */

typedef struct {
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int fd, started;
    /* request / response, protected by mu */
    size_t gap;                        /* min(1 MB, cap / 4) */
    char *buf; size_t cap;             /* buffer to fill: data goes to buf[gap, cap) */
    size_t filled; int eof;
    size_t newlines;                   /* '\n' bytes among the `filled` bytes */
    int state;                         /* 0 idle, 1 requested, 2 done, 3 quit */
    int regular, io_threads;           /* regular file: parallel pread() from `offset` on */
    size_t io_slice;                   /* smallest piece worth a thread of its own */
    off_t offset, limit;               /* limit > 0: the input ends at this file offset (a part of a sharded run) */
    int fifo;                          /* the input is a pipe: copied out of it by several threads (fxh_io.c: fxh_fan) */
    struct fxh_fan *fan;
} fxh_prefetch;

typedef struct {
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    struct fxh_writer *w;
    int started;
    const char *buf; size_t len;
    int state;                         /* 0 idle, 1 pending, 3 quit */
} fxh_awriter;

/* lanes (fxh_lanes.c) */
#define FXH_MAX_LANES 32
#define FXH_LANE_OUT_SLOTS 8
#define FXH_RECORDS_UNKNOWN ((uint64_t)-1)      /* fxh_lane.records: take what the device index finds (it still has to be whole records) */
extern int g_parts_abort;                  /* sharded run: some part met input it does not handle (fxh_run_parts); relaxed atomics, it is only a "stop soon" */
#define FXH_ABORT_SET() __atomic_store_n(&g_parts_abort, 1, __ATOMIC_RELAXED)
#define FXH_ABORTED()   __atomic_load_n(&g_parts_abort, __ATOMIC_RELAXED)
extern pthread_mutex_t g_first_ctx_mu;     /* the HIP runtime's first-use initialisation: one thread at a time */
extern int g_hip_touched;                  /* this process has initialised the HIP runtime (a context, or the device query of fxh_bind_near_device): never fork() after that */
extern int g_parts_mode;                   /* a sharded run is under way (fxh_run_parts) */
extern int g_rename_ids;
extern uint32_t g_part_clip_len[FXH_MAX_LANES];      /* clipper parts: the one read length each part saw (0: not a clipper run / no reads) */
struct fxh_pinned { pthread_mutex_t mu; const void *ptr[FXH_MAX_LANES + 4]; int n; };

typedef struct fxh_lane {
    int id, device;
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int state;                             /* 0 idle, 1 job posted, 2 done, 3 quit */
    int ready;                             /* the context exists (created by the lane's own thread, off the main thread's path) */
    struct fxh_lane *first;                /* lane 0: the others create their contexts after it (two threads inside the runtime's first-use initialisation take twice as long as one after the other) */
    fxh_state st;
    struct fxh_pinned *pinned;             /* input buffers already page-locked (shared by the lanes) */
    char *text_base; size_t text_cap;      /* the input buffer the job's text lives in */
    const fxg_params *p;                   /* configuration, read-only */
    int revcomp, qoffset;                  /* revcomp: the output comes from the engine's packed arrays (reverse-complement, masker) */
    int reverse, lpr, has_q, out_fasta;    /* the packed output is reversed; lines per record; qualities present; write FASTA */
    uint32_t fwd_start;
    const char *text; size_t len;          /* job: whole records, every line '\n'-terminated */
    uint64_t records;
    int clip_history;                      /* this lane is the one aligner of a fastx_clipper run */
    int clip_guard;                        /* clipper run in its parallel phase (fxh_run.clip_auto): a block whose reads are not all of one length is handed back untouched */
    uint32_t fixed_len;                    /* result: the one length of the block's reads, 0 = they differ (or the block was not indexed) */
    int slot;                              /* which of out[] receives the text (the other may still be with the writer) */
    int handled;                           /* result: 0 = irregular block, parse it on the host */
    char *out[FXH_LANE_OUT_SLOTS]; size_t out_cap[FXH_LANE_OUT_SLOTS]; size_t out_len;      /* (the lanes loop uses two, the strands of the one-file run more) */
    uint64_t ctr[FXG_NCOUNTERS];
    uint64_t weighted[8];                  /* FASTA: tallies weighted by the records' read counts (fxg_fasta_weights) */
    void (*on_size)(struct fxh_lane *, uint64_t out_bytes);      /* one-file sharded run: called once the block's formatted size is known, before its download */
    int (*on_place)(struct fxh_lane *, uint64_t out_bytes);      /* one-file run in rank mode: takes the formatted text on the device (1) instead of the download; -1 = stop */
    void *owner;
    double t_busy, t_init;
    double t_call[8];                        /* FXH_TIMING: seconds inside h2d, index, pack, pipeline, counters, format, d2h+sync, blocks */
} fxh_lane;

/* fastx_quality_stats mode of the run loop: batches feed fxg_run_quality_stats instead of the pipeline, nothing is written */
typedef struct fxh_stats_run {
    uint64_t *d_hist;
    uint32_t cols;
} fxh_stats_run;

/* everything one run of a tool shares between its blocks */
typedef struct fxh_run {
    FASTX *fx;
    const fxg_params *p;
    fxh_totals *tot;
    fxh_stats_run *stats;
    fxh_state st;                          /* the host-parser path's own context and buffers (created on first use) */
    int st_device, st_shared;              /* st_shared: the context belongs to lane 0 (serial clipper run) */
    /* fastx_clipper without being told anything: the reference's aligner carries its query buffer from read to read, but the
     * stale tail only exists once a read SHORTER than the longest so far turns up (sequence_alignment.cpp:135-136).  While every block so
     * far consists of reads of ONE length (clip_len, the first block's), blocks are independent: lanes and parts run in parallel without
     * history (clip_auto).  The first block that is different -- ragged, another length, or anything the device path hands back -- switches
     * the run to the reference's mode at that block: one lane, history on, seeded with the last record before it (clip_seed), which is
     * exactly the aligner's state after reads of one length (fxh_clip_go_serial). */
    int clip_auto;
    uint32_t clip_len;
    char *clip_seed; size_t clip_seed_len, clip_seed_cap;
    fxh_job job;
    fxh_awriter aw;
    char *wr_spare; size_t wr_spare_cap;
    int overlap;
    char errmsg[768];
    int have_err, at_eof;
    int part, nparts;                      /* sharded run (FXH_PARTS): this run is part `part` of `nparts`; irregular input aborts it (fxh_run_parts) */
    int aborted;
    struct fxh_pinned pinned;              /* input buffers the lanes have page-locked */
    unsigned long n_fallback;
    double t_index, t_pack, t_gpu, t_fmt, t_init;
    double t_wait_lane, t_wait_writer, t_drain;      /* lanes loop: main thread blocked on a lane / on the writer / final drain */
} fxh_run;

/* what the four steps of the host-parser path hand to one another */
typedef struct { size_t beg, end, n, maxlen, minlen; int stop; uint64_t ctr[FXG_NCOUNTERS]; } fxh_hb;


/* the blocks in flight: where their text lives and what the host parser needs if a lane hands one back */
typedef struct fxh_block {
    char *buf; size_t beg, end;            /* whole records; buf[end - 1] == '\n' */
    int eof;                               /* the input ends with this block */
    unsigned long long line0;              /* lines read before it */
    uint64_t records;
    int lane;                              /* -1: not given to a lane (ragged end of input, oversized record): host parser */
    int posted;                            /* its lane has the job (0 only between fxh_clip_go_serial and the block's turn) */
} fxh_block;

#define FXH_EXIT_ABANDON 99

void fxh_grow_device(fxh_state *st, size_t reads, size_t bytes, int revcomp);
void fxh_parallel(fxh_job *job, void (*phase)(fxh_worker *));
void fxh_phase_census(fxh_worker *w);
void fxh_next_block(fxh_prefetch *pf, struct fxh_reader *rd, char **spare);
void fxh_next_block_ring(fxh_prefetch *pf, struct fxh_reader *rd, char *target, size_t *fresh_newlines);
void fxh_prefetch_stop(fxh_prefetch *pf);
void fxh_awriter_wait(fxh_awriter *aw);
void fxh_awriter_submit(fxh_awriter *aw, struct fxh_writer *w, char **spare, size_t *spare_cap);
void fxh_awriter_stop(fxh_awriter *aw);
void fxh_awriter_submit_ext(fxh_awriter *aw, struct fxh_writer *w, const char *buf, size_t len);
int fxh_bind_near_device(int device, cpu_set_t *before);
int fxh_device_list(int *dev, int cap);
void fxh_host_block(fxh_run *R);
void fxh_add_counters(fxh_totals *tot, const uint64_t *ctr, uint64_t n, const uint64_t *weighted);
void fxh_lane_run(fxh_lane *ln);
void fxh_note_recoveries(fxh_state *st);
void fxh_lane_open_ctx(fxh_lane *ln);
void fxh_lane_release(fxh_lane *ln);
int fxh_run_one_file(FASTX *fx, const fxg_params *p, fxh_totals *tot);
off_t fxh_find_cut(int fd, off_t from, off_t size, int lpr, size_t window);
void fxh_lanes_stop(fxh_run *R, fxh_lane *lanes, int nlanes, double *t_lane_init);
void fxh_run_lanes(fxh_run *R, fxh_prefetch *pf, int nlanes, const int *lane_dev, double *t_read, double *t_lane_init);
int fxh_run_impl(FASTX *fx, const fxg_params *p, fxh_totals *tot, fxh_stats_run *stats, uint64_t **hist_out, uint32_t *cols_out, int part, int nparts);
void fxh_part_name(const FASTX *fx, int r, char *dst, size_t cap);
int fxh_run_parts(FASTX *fx, const fxg_params *p, fxh_totals *tot, int k);
int fxh_auto_parts(const FASTX *fx);
#endif
