/*
 * fastx_io.c -- block-buffered FASTA/FASTQ record reader and writer behind the libfastx-compatible API
 * of fastx.h, shared with the batch path (fxh_batch.c).
 *
 * What is reproduced from the reference (src/libfastx/fastx.c) is behaviour, not code: the input rules
 * R1-R9 of SURVEY.md section 8(a), the error texts with their line numbers, and the output bytes.  What is
 * different is how: whole-block read(2)/write(2) with memchr line splitting instead of four fgets() and
 * one fprintf("%c") per quality value.
 */
#define _GNU_SOURCE
#include "fastx.h"
#include "fxh_internal.h"

#include <err.h>
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/uio.h>
#include <sys/wait.h>
#include <unistd.h>
#include <zlib.h>

/* ---------------------------------------------------------------------------------------------- */
/* reader                                                                                         */
/* ---------------------------------------------------------------------------------------------- */
/* Pipes as the reference's users run them (`trimmer | filter`, `cat in | tool > out`): the default capacity of a Linux pipe is 64 KB -- sixteen
 * pages per wake-up of the other side.  A FIFO end is raised to the system's limit (/proc/sys/fs/pipe-max-size, 1 MB unless the administrator
 * changed it; less if that is refused). */
size_t fxh_tune_pipe(int fd)
{
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISFIFO(sb.st_mode)) return 0;
    if (!getenv("FXH_NO_PIPE_TUNING")) {
        long want = 1 << 20;
        FILE *f = fopen("/proc/sys/fs/pipe-max-size", "r");
        if (f) { long v; if (fscanf(f, "%ld", &v) == 1 && v >= 4096) want = v; fclose(f); }
        if (want > (64L << 20)) want = 64L << 20;
        { const char *e = getenv("FXH_PIPE_MB"); const long v = e ? atol(e) : 8; if (v >= 1 && v <= 64 && (v << 20) > want) want = v << 20; }      /* a privileged process may go beyond the limit: 8 MB */
        for (; want >= (128 << 10); want >>= 1) if (fcntl(fd, F_SETPIPE_SZ, (int)want) >= 0) break;      /* (EPERM above the limit of an unprivileged user, EBUSY: take less) */
    }
    const int have = fcntl(fd, F_GETPIPE_SZ);
    return have > 0 ? (size_t)have : 65536u;
}

struct fxh_reader *fxh_reader_open(const char *filename, size_t capacity)
{
    struct fxh_reader *r = (struct fxh_reader *)calloc(1, sizeof *r);
    if (!r) err(1, "out of memory");
    if (strncmp(filename, "-", 1) == 0) r->fd = STDIN_FILENO;      /* reference: any name starting with '-' */
    else {
        r->fd = open(filename, O_RDONLY);
        if (r->fd < 0) err(1, "failed to open input file '%s'", filename);
    }
    (void)fxh_tune_pipe(r->fd);                 /* `cat in.fq | tool` (the Galaxy wrappers): a megabyte per wake-up instead of 64 KB */
    r->cap = capacity ? capacity : (4u << 20);
    r->buf = (char *)malloc(r->cap + 1);
    if (!r->buf) err(1, "out of memory");
    return r;
}

struct fxh_reader *fxh_reader_open_range(const char *filename, size_t capacity, off_t start, off_t limit)
{
    struct fxh_reader *r = fxh_reader_open(filename, capacity);
    if (lseek(r->fd, start, SEEK_SET) != start) err(1, "failed to seek in input file '%s'", filename);
    r->limit = limit;
    return r;
}

/* enlarge the block (batch mode wants tens of MB per engine call); existing unread bytes are kept */
void fxh_reader_reserve(struct fxh_reader *r, size_t capacity)
{
    if (capacity <= r->cap) return;
    char *nb = (char *)realloc(r->buf, capacity + 1);
    if (!nb) err(1, "out of memory");
    r->buf = nb;
    r->cap = capacity;
}

/* move the unread tail to the front and read until the buffer is full or the input ends */
void fxh_reader_fill(struct fxh_reader *r)
{
    if (r->beg > 0) {
        memmove(r->buf, r->buf + r->beg, r->end - r->beg);
        r->end -= r->beg;
        r->beg = 0;
    }
    while (!r->eof && r->end < r->cap) {
        size_t want = r->cap - r->end;
        if (r->limit > 0) {                                    /* a part of a sharded run: the input ends at `limit` */
            const off_t pos = lseek(r->fd, 0, SEEK_CUR);
            if (pos < 0 || pos >= r->limit) { r->eof = 1; break; }
            if ((off_t)want > r->limit - pos) want = (size_t)(r->limit - pos);
        }
        ssize_t k = read(r->fd, r->buf + r->end, want);
        if (k < 0) { if (errno == EINTR) continue; err(1, "read failed"); }
        if (k == 0) { r->eof = 1; break; }
        r->end += (size_t)k;
    }
}

int fxh_reader_peek(struct fxh_reader *r)
{
    if (r->beg == r->end && !r->eof) fxh_reader_fill(r);
    return r->beg < r->end ? (unsigned char)r->buf[r->beg] : -1;
}

/* One fgets() worth of input: *p/*raw = the line including its '\n' (if any).  0 at end of input.
 * If the line is not complete in the buffer: refill when allowed, otherwise report -1 (batch mode). */
int fxh_reader_line(struct fxh_reader *r, const char **p, size_t *raw, int may_refill)
{
    for (;;) {
        const char *s = r->buf + r->beg;
        const char *nl = (const char *)memchr(s, '\n', r->end - r->beg);
        if (nl) { *p = s; *raw = (size_t)(nl - s) + 1; r->beg += *raw; return 1; }
        if (r->eof) {
            if (r->beg == r->end) return 0;
            *p = s; *raw = r->end - r->beg; r->beg = r->end; return 1;
        }
        if (!may_refill) return -1;
        if (r->beg == 0 && r->end == r->cap) errx(1, "input line longer than %zu bytes", r->cap);
        fxh_reader_fill(r);
    }
}

/* chomp.c:36-41: cut at the first CR or LF.  The reference's lines are C strings (fgets into a buffer, then strlen / %s), so a NUL byte
 * ends a line's content the same way: what follows it up to the newline is read and never looked at. */
size_t fxh_chomp_len(const char *s, size_t n)
{
    size_t k = 0;
    while (k < n && s[k] != '\r' && s[k] != '\n' && s[k] != '\0') k++;
    return k;
}

void chomp(char *string)
{
    string[fxh_chomp_len(string, strlen(string))] = 0;
}

/* ---------------------------------------------------------------------------------------------- */
/* record parser shared by the per-record API and the batch packer                                */
/* ---------------------------------------------------------------------------------------------- */
static void fxh_fail(FASTX *fx, struct fxh_rawrec *rec, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    if (rec && rec->defer_errors) {
        vsnprintf(rec->errmsg, sizeof rec->errmsg, fmt, ap);
        rec->failed = 1;
        va_end(ap);
        return;
    }
    (void)fx;
    verrx(1, fmt, ap);
}

/* Returns 1 = record parsed, 0 = end of input at a record boundary, -1 = record not complete in the buffer
 * (only when may_refill == 0; nothing consumed), -2 = deferred error (rec->errmsg set). */
int fxh_next_raw(FASTX *fx, struct fxh_rawrec *rec, int may_refill)
{
    struct fxh_reader *r = fx->reader;
    const char *p;
    size_t raw;
    int rc;
    rec->failed = 0;
    if (may_refill) {   /* the buffer moves on refill: make the whole record resident before taking pointers into it */
        const size_t want = fx->read_fastq ? 4 : 2;
        for (;;) {
            size_t nl = 0, i = r->beg;
            while (nl < want && i < r->end) {
                const char *q = (const char *)memchr(r->buf + i, '\n', r->end - i);
                if (!q) break;
                nl++;
                i = (size_t)(q - r->buf) + 1;
            }
            if (nl >= want || r->eof) break;
            if (r->beg == 0 && r->end == r->cap) errx(1, "input record longer than %zu bytes", r->cap);
            fxh_reader_fill(r);
        }
    }
    const size_t save_beg = r->beg;
    const unsigned long long save_line = fx->input_line_number;

    fx->input_line_number++;
    rc = fxh_reader_line(r, &p, &raw, 0);
    if (rc == 0) return 0;
    if (rc < 0) goto incomplete;
    rec->prefix = p[0];
    rec->name = p + 1;
    rec->name_len = fxh_chomp_len(p + 1, raw - 1);
    /* the reference reads lines with fgets(.., 25000) and silently splits longer ones; here they are rejected (the FASTX record
     * buffers of the per-record API hold MAX_SEQ_LINE_LENGTH bytes) */
    if (rec->name_len >= MAX_SEQ_LINE_LENGTH - 1) { fxh_fail(fx, rec, "identifier line longer than %d on line %lld\n", MAX_SEQ_LINE_LENGTH - 2, fx->input_line_number); return -2; }
    if (fx->read_fastq && rec->prefix != '@') {
        fxh_fail(fx, rec, "Invalid input: expecting FASTQ prefix character '@' on line %lld. Is this a valid FASTQ file?\n", fx->input_line_number);
        return -2;
    }
    if (!fx->read_fastq && rec->prefix != '>') {
        /* fastx.c:327,337: only the name (from the second byte on) is chomped before the "is this a line of bases" test, so a line
         * that STARTS with CR or LF -- a blank line -- fails it, while an empty C string (leading NUL) passes */
        size_t k = p[0] ? 1 + fxh_chomp_len(p + 1, raw - 1) : 0, i = 0;
        while (i < k && fx->allowed_nucleotides[(unsigned char)p[i]]) i++;
        if (i == k)
            fxh_fail(fx, rec, "Invalid input: This looks like a multi-line FASTA file.\nLine %lld contains a nucleotides string instead of a '>' prefix.\n"
                              "FASTX-Toolkit can't handle multi-line FASTA files.\nPlease use the FASTA-Formatter tool to convert this file into a single-line FASTA.\n",
                     fx->input_line_number);
        else
            fxh_fail(fx, rec, "Invalid input: expecting FASTA prefix character '>' on line %lld. Is this a valid FASTA file?\n", fx->input_line_number);
        return -2;
    }

    fx->input_line_number++;
    rc = fxh_reader_line(r, &p, &raw, 0);
    if (rc < 0) goto incomplete;
    if (rc == 0) { fxh_fail(fx, rec, "Failed to read complete record, missing 2nd line (nucleotides), on line %lld\n", fx->input_line_number); return -2; }
    rec->seq = p;
    rec->seq_len = fxh_chomp_len(p, raw);
    if (rec->seq_len == 0) { fxh_fail(fx, rec, "found empty nucleotide sequence on line %lld\n", fx->input_line_number); return -2; }
    if (rec->seq_len >= MAX_SEQ_LINE_LENGTH - 1) { fxh_fail(fx, rec, "sequence longer than %d on line %lld\n", MAX_SEQ_LINE_LENGTH - 2, fx->input_line_number); return -2; }
    {
        size_t i = 0;
        unsigned ok = 1;
        for (; i < rec->seq_len; ++i) ok &= fx->allowed_nucleotides[(unsigned char)p[i]];
        if (!ok) { fxh_fail(fx, rec, "found invalid nucleotide sequence (%.*s) on line %lld\n", (int)rec->seq_len, p, fx->input_line_number); return -2; }
    }
    rec->name2 = NULL; rec->name2_len = 0; rec->qual = NULL; rec->qual_len = 0; rec->is_ascii = 1;
    if (fx->read_fastq) {
        fx->input_line_number++;
        rc = fxh_reader_line(r, &p, &raw, 0);
        if (rc < 0) goto incomplete;
        if (rc == 0) { fxh_fail(fx, rec, "Failed to read complete record, missing 3rd line (name-2), on line %lld\n", fx->input_line_number); return -2; }
        rec->name2 = p + 1;                                  /* first byte dropped whatever it is (R5) */
        rec->name2_len = raw > 0 ? fxh_chomp_len(p + 1, raw - 1) : 0;
        if (rec->name2_len >= MAX_SEQ_LINE_LENGTH - 1) { fxh_fail(fx, rec, "identifier line longer than %d on line %lld\n", MAX_SEQ_LINE_LENGTH - 2, fx->input_line_number); return -2; }
        fx->input_line_number++;
        rc = fxh_reader_line(r, &p, &raw, 0);
        if (rc < 0) goto incomplete;
        if (rc == 0) { fxh_fail(fx, rec, "Failed to read complete record, missing 4th line (quality), on line %lld\n", fx->input_line_number); return -2; }
        rec->qual = p;
        rec->qual_len = fxh_chomp_len(p, raw);
        rec->is_ascii = (rec->qual_len == rec->seq_len);     /* R6 */
    }
    return 1;

incomplete:
    r->beg = save_beg;
    fx->input_line_number = save_line;
    return -1;
}

/* quality line -> numeric scores; returns 0 or sets a (possibly deferred) error and returns -1 */
int fxh_decode_quality(FASTX *fx, struct fxh_rawrec *rec, int *out_i32, unsigned char *out_phred33)
{
    const int Q = fx->fastq_ascii_quality_offset;
    if (rec->is_ascii) {
        for (size_t i = 0; i < rec->qual_len; ++i) {
            const int q = (int)(signed char)rec->qual[i] - Q;          /* fastx.c:127 */
            if (q < MIN_QUALITY_VALUE || q > MAX_QUALITY_VALUE) {
                fxh_fail(fx, rec, "Invalid quality score value (char '%c' ord %d quality value %d) on line %lld",
                         rec->qual[i], rec->qual[i], q, fx->input_line_number);
                return -1;
            }
            if (out_i32) out_i32[i] = q;
            if (out_phred33) out_phred33[i] = (unsigned char)(q + 33);
        }
        return 0;
    }
    /* numeric scores (fastx.c:137-167): the reference calls strtol() on the rest of the line until the rest is empty, so a token
     * is whatever strtol takes -- leading isspace() bytes, one optional sign, digits -- and "10-5" is two values */
    {
        size_t idx = 0, pos = 0;
        const char *s = rec->qual;
        const size_t n = rec->qual_len;
        do {
            size_t j = pos;
            while (j < n && (s[j] == ' ' || (s[j] >= '\t' && s[j] <= '\r'))) j++;
            int neg = 0;
            if (j < n && (s[j] == '-' || s[j] == '+')) { neg = (s[j] == '-'); j++; }
            const size_t d0 = j;
            unsigned long long mag = 0;
            int sat = 0;
            const unsigned long long limit = neg ? 0x8000000000000000ull : 0x7FFFFFFFFFFFFFFFull;   /* strtol saturates exactly past LONG_MAX / below LONG_MIN */
            for (; j < n && s[j] >= '0' && s[j] <= '9'; ++j) {
                const unsigned long long d = (unsigned long long)(s[j] - '0');
                if (sat || mag > (limit - d) / 10) sat = 1; else mag = mag * 10 + d;
            }
            if (j == d0) {                                     /* endptr == quality_tok */
                fxh_fail(fx, rec, "Error: invalid quality score data on line %lld (quality_tok = \"%.*s\"", fx->input_line_number, (int)(n - pos), s + pos);
                return -1;
            }
            const long lv = sat ? (neg ? (-0x7FFFFFFFFFFFFFFFL - 1) : 0x7FFFFFFFFFFFFFFFL) : (long)(neg ? 0ull - mag : mag);
            const int v = (int)lv;                             /* the reference stores strtol's long in an int */
            if (v > 93 || v < -15) { fxh_fail(fx, rec, "invalid quality score value (%d) in line %lld.", v, fx->input_line_number); return -1; }
            if (idx < rec->seq_len) {
                if (out_i32) out_i32[idx] = v;
                if (out_phred33) out_phred33[idx] = (unsigned char)(v + 33);
            }
            idx++;
            pos = j;
        } while (pos < n);
        if (idx != rec->seq_len) {
            fxh_fail(fx, rec, "number of quality values (%zu) doesn't match number of nucleotides (%zu) on line %lld", idx, rec->seq_len, fx->input_line_number);
            return -1;
        }
    }
    return 0;
}

int fxh_reads_count(const FASTX *fx, const char *name, size_t name_len)   /* fastx.c:475-495 */
{
    if (fx->read_fastq) return 1;
    const char *dash = (const char *)memchr(name, '-', name_len);
    if (!dash) return 1;
    char tmp[24];
    size_t k = (size_t)(name + name_len - (dash + 1));
    if (k > sizeof tmp - 1) k = sizeof tmp - 1;
    memcpy(tmp, dash + 1, k);
    tmp[k] = 0;
    int c = atoi(tmp);
    return c > 0 ? c : 1;
}

/* ---------------------------------------------------------------------------------------------- */
/* writer                                                                                         */
/* ---------------------------------------------------------------------------------------------- */
static struct fxh_writer *g_writers[8];
static int g_fxh_batch_writer;          /* set by fxh_init_writer (the batch tools) around its call of fastx_init_writer */

static void fxh_flush_all(void)
{
    for (size_t i = 0; i < sizeof g_writers / sizeof g_writers[0]; ++i)
        if (g_writers[i]) fxh_writer_close(g_writers[i]);
}

static void fxh_write_all(int fd, const char *buf, size_t n)
{
    size_t off = 0;
    while (off < n) {
        ssize_t k = write(fd, buf + off, n - off);
        if (k < 0) { if (errno == EINTR) continue; err(1, "writing output failed"); }
        off += (size_t)k;
    }
}

/* -z.  The reference pipes its output through a `gzip` child (fastx.c:214-248): one core, however fast the tool is.  Here the
 * buffer is cut into 1 MiB chunks, worker threads deflate them independently (zlib, level 6 like gzip's default) and the
 * results are written in order as consecutive gzip members -- a valid gzip stream (RFC 1952 2.2) that gunzip/zcat
 * decompress to exactly the bytes the reference's pipe would have carried. */
#define FXH_GZ_CHUNK ((size_t)1 << 20)
struct fxh_gz_job {
    const char *in; size_t n, nchunks;
    char **out; size_t *outlen;
    int nthreads;
};
struct fxh_gz_arg { struct fxh_gz_job *job; int id; };

static void fxh_gz_member(const char *in, size_t n, char **out, size_t *outlen)
{
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, 6, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) errx(1, "deflateInit2 failed");
    const size_t cap = deflateBound(&zs, (uLong)n) + 64;
    char *o = (char *)malloc(cap);
    if (!o) err(1, "out of memory");
    zs.next_in = (Bytef *)(uintptr_t)in; zs.avail_in = (uInt)n;
    zs.next_out = (Bytef *)o; zs.avail_out = (uInt)cap;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) errx(1, "deflate failed");
    *out = o; *outlen = cap - zs.avail_out;
    deflateEnd(&zs);
}

static void *fxh_gz_worker(void *p)
{
    struct fxh_gz_arg *a = (struct fxh_gz_arg *)p;
    struct fxh_gz_job *j = a->job;
    for (size_t c = (size_t)a->id; c < j->nchunks; c += (size_t)j->nthreads) {
        const size_t off = c * FXH_GZ_CHUNK, len = j->n - off < FXH_GZ_CHUNK ? j->n - off : FXH_GZ_CHUNK;
        fxh_gz_member(j->in + off, len, &j->out[c], &j->outlen[c]);
    }
    return NULL;
}

/* Plain output to a regular file goes out as positional writes from the writer's own offset (the descriptor's position is put
 * right when the writer closes).  Measured on tmpfs (16 M reads, 2.5 GB out, profiles/r02/ab_e2e_mapped_output.txt): one pwrite()
 * stream 0.37 s; several pwrite() threads are no faster (they serialise on the inode lock, and page allocation is the cost);
 * mapping the output file instead -- one mapping over the expected output with helper threads faulting pages in ahead of the
 * copies (MADV_POPULATE_WRITE) -- brings the writes themselves to 0.0-0.1 s of waiting but costs 0.3 s to take the 600 000 page
 * mappings down again at exit, and mapping block by block is slower than pwrite() (0.6 s); allocating the pages ahead of the
 * writes with fallocate() changes nothing (ab_e2e_prealloc.txt): the copy is the cost.  So: pwrite(). */
static void fxh_pwrite_all(int fd, const char *buf, size_t n, off_t off)
{
    size_t done = 0;
    while (done < n) {
        ssize_t k = pwrite(fd, buf + done, n - done, off + (off_t)done);
        if (k < 0) { if (errno == EINTR) continue; err(1, "writing output failed"); }
        done += (size_t)k;
    }
}

/* A block into a pipe.  write() copies every byte into the pipe's pages under the pipe's lock -- which the reader's copy out of it needs as well, so the two
 * copies take turns: 3.4 GB/s for `trimmer | filter`, one thread on either side.  The copies themselves cannot be avoided safely (vmsplice() would hand the
 * pipe pages of the output buffers, and pages handed over belong to the pipe -- and to whatever pipe or socket a downstream process splices them on into,
 * pv does -- until their last reader is done, which a writer cannot observe: the buffers are written to again).  But they can be made side by side and outside
 * the shared pipe's lock: a few threads write() the block's pieces into PRIVATE pipes, one piece of a pipe capacity each, round robin, and this thread moves the
 * pieces on into the output pipe in order with splice(), which moves page references and copies nothing.  The pages that travel are the kernel's own pipe
 * pages: correct whatever the reader does with them.  The reader does the same in reverse (fxh_io.c: fxh_fan).  (bench.py e2e.pipe) */
typedef struct { int wfd; const char *buf; size_t n, piece; int first, stride; } fxh_fanin_job;
static void *fxh_fanin_main(void *arg)
{
    fxh_fanin_job *j = (fxh_fanin_job *)arg;
    for (size_t o = (size_t)j->first * j->piece; o < j->n; o += (size_t)j->stride * j->piece)
        fxh_write_all(j->wfd, j->buf + o, j->n - o < j->piece ? j->n - o : j->piece);
    return NULL;
}

static int fxh_fanin_open(struct fxh_writer *w)
{
    if (w->fan_n) return 1;
    if (w->fan_off || getenv("FXH_NO_PIPE_FANOUT")) { w->fan_off = 1; return 0; }
    const char *e = getenv("FXH_PIPE_WRITERS");
    long n = e ? atol(e) : 3;
    if (n > 4) n = 4;
    w->fan_piece = w->pipe_size;
    for (int i = 0; i < (int)n; ++i) {
        int pfd[2];
        if (pipe2(pfd, O_CLOEXEC) != 0) break;
        const size_t cap = fxh_tune_pipe(pfd[1]);
        if (cap && cap < w->fan_piece) w->fan_piece = cap;
        w->fan_r[i] = pfd[0]; w->fan_w[i] = pfd[1];
        w->fan_n = i + 1;
    }
    if (w->fan_n < 2 || w->fan_piece < 65536) {
        for (int i = 0; i < w->fan_n; ++i) { close(w->fan_r[i]); close(w->fan_w[i]); }
        w->fan_n = 0; w->fan_off = 1;
        return 0;
    }
    return 1;
}

static void fxh_pipe_write_all(struct fxh_writer *w, const char *buf, size_t n)
{
    if (n < ((size_t)1 << 20) || !fxh_fanin_open(w)) { fxh_write_all(w->fd, buf, n); return; }
    const int K = w->fan_n;
    /* pieces of a pipe capacity, smaller where the block would otherwise not keep every writer busy twice over */
    size_t piece = (n / (size_t)(2 * K)) & ~(size_t)4095u;
    if (piece < ((size_t)256 << 10)) piece = (size_t)256 << 10;
    if (piece > w->fan_piece) piece = w->fan_piece;
    pthread_t th[4];
    fxh_fanin_job job[4];
    int started = 0;
    for (int i = 0; i < K; ++i) {
        job[i].wfd = w->fan_w[i]; job[i].buf = buf; job[i].n = n; job[i].piece = piece; job[i].first = i; job[i].stride = K;
        if (pthread_create(&th[i], NULL, fxh_fanin_main, &job[i]) != 0) err(1, "pthread_create");
        started = i + 1;
    }
    int lane = 0, copying = 0;
    for (size_t o = 0; o < n; o += piece, lane = (lane + 1) % K) {
        size_t left = n - o < piece ? n - o : piece;
        while (left) {
            ssize_t k = copying ? -1 : splice(w->fan_r[lane], NULL, w->fd, NULL, left, SPLICE_F_MOVE);
            if (k < 0 && !copying) {
                if (errno == EINTR) continue;
                if (errno != EINVAL && errno != ENOSYS) err(1, "writing output failed");
                copying = 1;                             /* an output pipe that takes no splice (opened O_APPEND): the pieces are copied across, later blocks written directly */
            }
            if (copying) {
                char tmp[65536];
                k = read(w->fan_r[lane], tmp, left < sizeof tmp ? left : sizeof tmp);
                if (k < 0) { if (errno == EINTR) continue; err(1, "writing output failed"); }
                fxh_write_all(w->fd, tmp, (size_t)k);
            }
            if (k == 0) errx(1, "writing output failed");
            left -= (size_t)k;
        }
    }
    for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
    if (copying) { for (int i = 0; i < w->fan_n; ++i) { close(w->fan_r[i]); close(w->fan_w[i]); } w->fan_n = 0; w->fan_off = 1; }
}

void fxh_writer_emit(struct fxh_writer *w, const char *buf, size_t n)
{
    if (!w->gz) {
        if (w->positional) { fxh_pwrite_all(w->fd, buf, n, w->off); w->off += (off_t)n; }
        else if (w->pipe_size) fxh_pipe_write_all(w, buf, n);
        else fxh_write_all(w->fd, buf, n);
        return;
    }
    if (n == 0) return;
    struct fxh_gz_job job;
    job.in = buf; job.n = n; job.nchunks = (n + FXH_GZ_CHUNK - 1) / FXH_GZ_CHUNK;
    job.out = (char **)calloc(job.nchunks, sizeof(char *));
    job.outlen = (size_t *)calloc(job.nchunks, sizeof(size_t));
    if (!job.out || !job.outlen) err(1, "out of memory");
    const char *te = getenv("FXH_THREADS");
    long nt = te ? atol(te) : 16, ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    if (ncpu > 0 && nt > ncpu) nt = ncpu;
    if ((size_t)nt > job.nchunks) nt = (long)job.nchunks;
    job.nthreads = (int)nt;
    pthread_t th[64];
    struct fxh_gz_arg arg[64];
    for (int i = 0; i < job.nthreads; ++i) { arg[i].job = &job; arg[i].id = i; }
    for (int i = 1; i < job.nthreads; ++i) if (pthread_create(&th[i], NULL, fxh_gz_worker, &arg[i]) != 0) err(1, "pthread_create");
    fxh_gz_worker(&arg[0]);
    for (int i = 1; i < job.nthreads; ++i) pthread_join(th[i], NULL);
    for (size_t c = 0; c < job.nchunks; ++c) { fxh_write_all(w->fd, job.out[c], job.outlen[c]); free(job.out[c]); }
    w->gz_members += job.nchunks;
    free(job.out); free(job.outlen);
}

void fxh_writer_flush(struct fxh_writer *w)
{
    fxh_writer_emit(w, w->buf, w->len);
    w->len = 0;
}

char *fxh_writer_reserve(struct fxh_writer *w, size_t n)
{
    if (w->len + n > w->cap) {
        fxh_writer_flush(w);
        if (n > w->cap) {
            w->buf = (char *)realloc(w->buf, n);
            if (!w->buf) err(1, "out of memory");
            w->cap = n;
        }
    }
    return w->buf + w->len;
}

void fxh_writer_close(struct fxh_writer *w)
{
    if (!w || w->fd < 0) return;
    fxh_writer_flush(w);
    if (w->gz && w->gz_members == 0) {           /* nothing was written: still a valid (empty) gzip file, as `gzip < /dev/null` gives */
        char *o; size_t on;
        fxh_gz_member("", 0, &o, &on);
        fxh_write_all(w->fd, o, on);
        free(o);
    }
    if (w->positional) (void)lseek(w->fd, w->off, SEEK_SET);     /* leave the descriptor where a write() stream would have */
    if (w->fd != STDOUT_FILENO) close(w->fd);
    w->fd = -1;
}

static int fxh_open_output(const char *filename)
{
    if (strcmp(filename, "-") == 0) return STDOUT_FILENO;
    /* read-write where that is allowed: the many-strand run maps its one output file (fxh_strands.c); a file that may only be written still opens */
    /* one process per GPU over one job (FXH_WORLD > 1, fxh_strands.c): the file is rank 0's to empty; the others only open it */
    const char *we = getenv("FXH_WORLD"), *re = getenv("FXH_RANK");
    const int trunc = (g_fxh_batch_writer && we && atoi(we) > 1 && re && atoi(re) > 0) ? 0 : O_TRUNC;
    int fd = open(filename, O_CREAT | O_RDWR | trunc, 0666);
    if (fd == -1 && errno == EACCES) fd = open(filename, O_CREAT | O_WRONLY | trunc, 0666);
    if (fd == -1) err(1, "Failed to create output file (%s)", filename);
    return fd;
}

static struct fxh_writer *fxh_writer_open(const char *filename, int gzip);
struct fxh_writer *fxh_writer_open_file(const char *filename, int gzip) { return fxh_writer_open(filename, gzip); }
static struct fxh_writer *fxh_writer_from_fd(int fd, int gzip);
/* a writer over a descriptor the caller has opened (the parts of a sharded run are opened before anything can fail halfway) */
struct fxh_writer *fxh_writer_open_fd(int fd) { return fxh_writer_from_fd(fd, 0); }

static struct fxh_writer *fxh_writer_open(const char *filename, int gzip) { return fxh_writer_from_fd(fxh_open_output(filename), gzip); }

static struct fxh_writer *fxh_writer_from_fd(int fd, int gzip)
{
    struct fxh_writer *w = (struct fxh_writer *)calloc(1, sizeof *w);
    if (!w) err(1, "out of memory");
    w->cap = 8u << 20;
    w->buf = (char *)malloc(w->cap);
    if (!w->buf) err(1, "out of memory");
    w->fd = fd;
    w->gz = gzip ? 1 : 0;
    {
        struct stat sb;
        const off_t pos = lseek(w->fd, 0, SEEK_CUR);
        const int fl = fcntl(w->fd, F_GETFL);
        w->positional = (!w->gz && pos >= 0 && fl >= 0 && !(fl & O_APPEND) && fstat(w->fd, &sb) == 0 && S_ISREG(sb.st_mode)) ? 1 : 0;
        w->off = pos;
        w->pipe_size = fxh_tune_pipe(w->fd);
    }
    static int registered;
    if (!registered) { atexit(fxh_flush_all); registered = 1; }
    for (size_t i = 0; i < sizeof g_writers / sizeof g_writers[0]; ++i)
        if (!g_writers[i]) { g_writers[i] = w; break; }
    return w;
}

/* ---------------------------------------------------------------------------------------------- */
/* libfastx-compatible API                                                                        */
/* ---------------------------------------------------------------------------------------------- */
void fastx_init_reader(FASTX *fx, const char *filename, ALLOWED_INPUT_FILE_TYPES allowed_input_filetype,
                       ALLOWED_INPUT_BASES allow_bases, ALLOWED_INPUT_CASE allow_lowercase, int fastq_ascii_quality_offset)
{
    if (fx == NULL) errx(1, "Internal error: pFASTX==NULL (%s:%d)", __FILE__, __LINE__);
    memset(fx, 0, sizeof *fx);
    const char *cap_env = getenv("FXH_READ_BUFFER_MB");
    fx->reader = fxh_reader_open(filename, cap_env && atoi(cap_env) > 0 ? (size_t)atoi(cap_env) << 20 : 0);
    strncpy(fx->input_file_name, filename, sizeof fx->input_file_name - 1);
    fx->allow_input_filetype = allowed_input_filetype;
    fx->allow_lowercase = allow_lowercase;
    fx->allow_N = (allow_bases & ALLOW_N) != 0;
    fx->allow_U = (allow_bases & ALLOW_U) != 0;
    fx->fastq_ascii_quality_offset = fastq_ascii_quality_offset;
    {   /* alphabet table (fastx.c:56-84) */
        const char *up = "ACGT", *p;
        for (p = up; *p; ++p) { fx->allowed_nucleotides[(unsigned char)*p] = 1; if (allow_lowercase) fx->allowed_nucleotides[(unsigned char)(*p | 0x20)] = 1; }
        if (fx->allow_N) { fx->allowed_nucleotides['N'] = 1; if (allow_lowercase) fx->allowed_nucleotides['n'] = 1; }
        if (fx->allow_U) { fx->allowed_nucleotides['U'] = 1; if (allow_lowercase) fx->allowed_nucleotides['u'] = 1; }
    }
    const int c = fxh_reader_peek(fx->reader);    /* format sniff on the first byte (R1) */
    if (c == '>') {
        if (allowed_input_filetype == FASTQ_ONLY) errx(1, "input file (%s) is FASTA, but only FASTQ input is allowed.", fx->input_file_name);
        fx->read_fastq = 0;
    } else if (c == '@') {
        if (allowed_input_filetype == FASTA_ONLY) errx(1, "input file (%s) is FASTQ, but only FASTA input is allowed.", fx->input_file_name);
        fx->read_fastq = 1;
    } else if (c == -1) {
        errx(1, "Premature End-Of-File (filename ='%s')", fx->input_file_name);
    } else {
        errx(1, "input file (%s) has unknown file format (not FASTA or FASTQ), first character = %c (%d)", fx->input_file_name, c, c);
    }
}

/* Set by fxh_init_writer (the batch tools) around its call of fastx_init_writer: only there does "%r" in the output name stand for the part
 * number of a sharded run.  A per-record caller of the libfastx API opens exactly the name it gave. */
void fxh_init_writer(FASTX *fx, const char *filename, OUTPUT_FILE_TYPE output_type, int compress_output)
{
    g_fxh_batch_writer = 1;
    fastx_init_writer(fx, filename, output_type, compress_output);
    g_fxh_batch_writer = 0;
}

void fastx_init_writer(FASTX *fx, const char *filename, OUTPUT_FILE_TYPE output_type, int compress_output)
{
    if (fx == NULL) errx(1, "Internal error: pFASTX==NULL (%s:%d)", __FILE__, __LINE__);
    if (fx->reader == NULL) errx(1, "Internal error: pFASTX not initialized (%s:%d)", __FILE__, __LINE__);
    fx->compress_output = compress_output;
    strncpy(fx->output_file_name, filename, sizeof fx->output_file_name - 1);
    {   /* "-o out.%r.fq" names the output parts of a sharded run (FXH_PARTS=k, or chosen by the tool: fxh_run_tool); this writer is
         * part 0 (fxh_parts.c opens the others) */
        char first[PATH_MAX];
        const char *pe = getenv("FXH_PARTS"), *pr = g_fxh_batch_writer ? strstr(filename, "%r") : NULL;     /* record-API callers get the literal name, like fastx.c:251-271 */
        if ((!pe || atoi(pe) >= 1) && pr && strlen(filename) < sizeof first - 8) {
            snprintf(first, sizeof first, "%.*s0%s", (int)(pr - filename), filename, pr + 2);
            fx->writer = fxh_writer_open(first, compress_output);
        } else fx->writer = fxh_writer_open(filename, compress_output);
    }
    switch (output_type) {
    case OUTPUT_FASTA:
        fx->write_fastq = 0; fx->output_sequence_id_prefix = '>';
        break;
    case OUTPUT_FASTQ_ASCII_QUAL:
    case OUTPUT_FASTQ_NUMERIC_QUAL:
        if (!fx->read_fastq) errx(1, "Can't output FASTQ when input is FASTA.");
        fx->write_fastq = 1; fx->write_fastq_ascii = (output_type == OUTPUT_FASTQ_ASCII_QUAL); fx->output_sequence_id_prefix = '@';
        break;
    case OUTPUT_SAME_AS_INPUT:
        fx->write_fastq = fx->read_fastq; fx->write_fastq_ascii = 1; fx->copy_input_fastq_format_to_output = 1;
        fx->output_sequence_id_prefix = fx->write_fastq ? '@' : '>';
        break;
    default:
        errx(1, __FILE__ ":%d: Unknown output_type (%d)", __LINE__, output_type);
    }
}

int fastx_read_next_record(FASTX *fx)
{
    struct fxh_rawrec rec;
    if (fx == NULL) errx(1, "Internal error: pFASTX==NULL (%s:%d)", __FILE__, __LINE__);
    memset(&rec, 0, sizeof rec);
    if (fxh_next_raw(fx, &rec, 1) == 0) return 0;
    memcpy(fx->name, rec.name, rec.name_len); fx->name[rec.name_len] = 0;
    memcpy(fx->nucleotides, rec.seq, rec.seq_len); fx->nucleotides[rec.seq_len] = 0;
    if (fx->read_fastq) {
        memcpy(fx->name2, rec.name2, rec.name2_len); fx->name2[rec.name2_len] = 0;
        fxh_decode_quality(fx, &rec, fx->quality, NULL);
        fx->read_fastq_ascii = rec.is_ascii;
        if (fx->copy_input_fastq_format_to_output) fx->write_fastq_ascii = fx->read_fastq_ascii;
    }
    fx->num_input_sequences++;
    fx->num_input_reads += (size_t)get_reads_count(fx);
    return 1;
}

size_t fxh_format_numeric(char *dst, const int *q, const unsigned char *phred33, size_t n)
{
    size_t w = 0;
    for (size_t i = 0; i < n; ++i) {
        int v = q ? q[i] : (int)phred33[i] - 33;
        if (i) dst[w++] = ' ';
        if (v < 0) { dst[w++] = '-'; v = -v; }
        if (v >= 10) dst[w++] = (char)('0' + v / 10);
        dst[w++] = (char)('0' + v % 10);
    }
    return w;
}

void fastx_write_record(FASTX *fx)
{
    if (fx == NULL) errx(1, "Internal error: pFASTX==NULL (%s:%d)", __FILE__, __LINE__);
    struct fxh_writer *w = fx->writer;
    const size_t nl = strlen(fx->name), sl = strlen(fx->nucleotides), n2 = strlen(fx->name2);
    char *d = fxh_writer_reserve(w, nl + sl + n2 + 5 * sl + 16);
    size_t k = 0;
    d[k++] = fx->output_sequence_id_prefix;
    memcpy(d + k, fx->name, nl); k += nl; d[k++] = '\n';
    memcpy(d + k, fx->nucleotides, sl); k += sl; d[k++] = '\n';
    if (fx->write_fastq) {
        d[k++] = '+';
        memcpy(d + k, fx->name2, n2); k += n2; d[k++] = '\n';
        if (fx->write_fastq_ascii)
            for (size_t i = 0; i < sl; ++i) d[k++] = (char)(fx->quality[i] + fx->fastq_ascii_quality_offset);   /* fastx.c:412 */
        else
            k += fxh_format_numeric(d + k, fx->quality, NULL, sl);
        d[k++] = '\n';
    }
    w->len += k;
    fx->num_output_sequences++;
    fx->num_output_reads += (size_t)get_reads_count(fx);
}

int get_reads_count(const FASTX *fx) { return fxh_reads_count(fx, fx->name, strlen(fx->name)); }
size_t num_input_sequences(const FASTX *fx) { return fx->num_input_sequences; }
size_t num_input_reads(const FASTX *fx) { return fx->num_input_reads; }
size_t num_output_sequences(const FASTX *fx) { return fx->num_output_sequences; }
size_t num_output_reads(const FASTX *fx) { return fx->num_output_reads; }

void fastx_finish(FASTX *fx)
{
    if (fx && fx->writer) fxh_writer_close(fx->writer);
}
