/* fxh_internal.h -- shared between fastx_io.c (record API) and the batch path (fxh_batch.c, fxh_io.c, fxh_lanes.c, fxh_parts.c). Not installed. */
#ifndef FXH_INTERNAL_H
#define FXH_INTERNAL_H
#include <stddef.h>
#include <sys/types.h>

#include "fastx.h"

struct fxh_reader {
    int fd;
    char *buf;
    size_t cap, beg, end;   /* unread bytes are buf[beg, end) */
    int eof;
    off_t limit;            /* > 0: the input of this reader ends at this file offset (one part of a sharded run, fxh_parts.c) */
};

struct fxh_writer {
    int fd;
    char *buf;
    size_t cap, len;
    int gz;                 /* -z: the output is a gzip stream, compressed here in parallel (one member per chunk) */
    unsigned long gz_members;
    int positional;         /* plain output to a regular file: positional writes from `off` on */
    off_t off;
    size_t pipe_size;       /* > 0: the output is a pipe of this capacity (enlarged to the system's limit when the writer was opened) */
    int fan_n, fan_off;     /* pipe output: private pipes that writer threads fill side by side and whose pages are moved on into the output pipe in order (fxh_pipe_write_all) */
    int fan_r[4], fan_w[4];
    size_t fan_piece;
};

/* one record as slices of the reader's buffer (valid until the next fill) */
struct fxh_rawrec {
    char prefix;
    const char *name, *seq, *name2, *qual;
    size_t name_len, seq_len, name2_len, qual_len;
    int is_ascii;           /* R6: quality line has as many characters as the sequence */
    int defer_errors;       /* batch mode: record the message instead of exiting, so earlier records are flushed first */
    int failed;
    char errmsg[768];
};

struct fxh_reader *fxh_reader_open(const char *filename, size_t capacity);
struct fxh_reader *fxh_reader_open_range(const char *filename, size_t capacity, off_t start, off_t limit);   /* regular file, bytes [start, limit) */
struct fxh_writer *fxh_writer_open_file(const char *filename, int gzip);
struct fxh_writer *fxh_writer_open_fd(int fd);
void   fxh_reader_reserve(struct fxh_reader *r, size_t capacity);
void   fxh_reader_fill(struct fxh_reader *r);
int    fxh_reader_peek(struct fxh_reader *r);
int    fxh_reader_line(struct fxh_reader *r, const char **p, size_t *raw, int may_refill);
size_t fxh_chomp_len(const char *s, size_t n);
int    fxh_next_raw(FASTX *fx, struct fxh_rawrec *rec, int may_refill);
int    fxh_decode_quality(FASTX *fx, struct fxh_rawrec *rec, int *out_i32, unsigned char *out_phred33);
int    fxh_reads_count(const FASTX *fx, const char *name, size_t name_len);
size_t fxh_format_numeric(char *dst, const int *q, const unsigned char *phred33, size_t n);
void   fxh_writer_flush(struct fxh_writer *w);
void   fxh_writer_emit(struct fxh_writer *w, const char *buf, size_t n);   /* raw bytes to the file, or gzip members when w->gz */
char  *fxh_writer_reserve(struct fxh_writer *w, size_t n);
void   fxh_writer_close(struct fxh_writer *w);
size_t fxh_tune_pipe(int fd);       /* a FIFO: capacity raised to /proc/sys/fs/pipe-max-size where allowed; returns the capacity, 0 = not a pipe */
#endif
