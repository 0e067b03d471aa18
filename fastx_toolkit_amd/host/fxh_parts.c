/* fxh_parts.c -- the sharded run of the tools: k byte ranges of the input side by side, k output parts (fxh_priv.h). */
#include "fxh_priv.h"
uint32_t g_part_clip_len[FXH_MAX_LANES];

/* ---------------------------------------------------------------------------------------------- */
/* Sharded run (FXH_PARTS=k): the text-level analogue of fxg_shard_range / fxg_epilogue / fxg_concat_pwrite.  The input file is   */
/* cut into k contiguous byte ranges at record boundaries; k runs (each the lanes loop above: own reader threads, own lanes, own    */
/* writer thread) work through them at the same time, part r on GPU r mod #GPUs, and write k output parts whose concatenation in    */
/* part order is the output of the unsharded run: one writer stream is what caps a single run (a tmpfs or page-cache write is one   */
/* thread under the inode lock), k parts are k streams.  `-o NAME` names part 0 NAME and part r NAME.r; `-o out.%r.fq` substitutes. */
/* An index NAME.parts lists (part, file, input bytes, records in, records out, output bytes).                                       */
/*   A record is four (two) lines counted from the start of the input, so a cut is only KNOWN to be a record boundary when the line */
/* count before it is: the cut points are found by pattern (an '@' line, a '+' line two below it, equal lengths, the same again)    */
/* and then PROVEN -- part r ends exactly at part r+1's cut, so if its lines are a whole number of records and part r started at a  */
/* boundary, so does part r+1 (induction from offset 0).  A part that meets anything the device path does not take (a ragged end =   */
/* a wrong cut, a malformed record, CR-less oddities the host parser owns) stops all parts; the attempt ran in a child process, which */
/* empties the parts and exits, and the parent runs the input as one stream: messages, exit codes and partial output are the         */
/* reference's in every case.                                                                                                         */
/* ---------------------------------------------------------------------------------------------- */
/* first record start after `from`, looked for in `window` bytes of text (-1: none found there) */
off_t fxh_find_cut(int fd, off_t from, off_t size, int lpr, size_t window)
{
    const size_t W = window ? window : (size_t)4 << 20;
    char *w = (char *)malloc(W);
    if (!w) err(1, "out of memory");
    ssize_t got = pread(fd, w, W, from);
    off_t found = -1;
    if (got > 0) {
        size_t n = (size_t)got, ls[12];
        const char *nl = (const char *)memchr(w, '\n', n);
        size_t pos = nl ? (size_t)(nl - w) + 1 : n;                         /* first line start after `from` */
        while (pos < n && found < 0) {
            int k = 0;                                                       /* starts of this line and the next 2 lpr */
            size_t q = pos;
            while (k < 2 * lpr + 1 && q < n) { ls[k++] = q; const char *e = (const char *)memchr(w + q, '\n', n - q); if (!e) { q = n; break; } q = (size_t)(e - w) + 1; }
            if (k < 2 * lpr + 1) break;                                     /* not enough text in the window */
            int ok;
            if (lpr == 2) ok = w[ls[0]] == '>' && w[ls[2]] == '>';
            else ok = w[ls[0]] == '@' && w[ls[2]] == '+' && (ls[2] - ls[1]) == (ls[4] - ls[3]) &&
                      w[ls[4]] == '@' && w[ls[6]] == '+' && (ls[6] - ls[5]) == (ls[8] - ls[7]);
            if (ok) found = from + (off_t)ls[0];
            else pos = ls[1];
        }
    }
    free(w);
    return (found > 0 && found < size) ? found : -1;
}

typedef struct { FASTX *fx; const fxg_params *p; fxh_totals tot; int part, nparts, rc; pthread_t th; off_t start, limit; char name[PATH_MAX + 16]; } fxh_part;
static void *fxh_part_main(void *arg)
{
    fxh_part *pt = (fxh_part *)arg;
    pt->rc = fxh_run_impl(pt->fx, pt->p, &pt->tot, NULL, NULL, NULL, pt->part, pt->nparts);
    return NULL;
}

void fxh_part_name(const FASTX *fx, int r, char *dst, size_t cap)
{
    const char *name = fx->output_file_name, *pr = strstr(name, "%r");
    if (pr) snprintf(dst, cap, "%.*s%d%s", (int)(pr - name), name, r, pr + 2);
    else if (r == 0) snprintf(dst, cap, "%s", name);
    else snprintf(dst, cap, "%s.%d", name, r);
}

/* 0 = done (in the child of the fork below: the caller goes on to print its reports); -1 = run unsharded (not eligible, or the sharded attempt was abandoned) */
int fxh_run_parts(FASTX *fx, const fxg_params *p, fxh_totals *tot, int k)
{
    struct fxh_reader *rd = fx->reader;
    struct stat sb;
    if (k > FXH_MAX_LANES) k = FXH_MAX_LANES;
    if (rd->fd == STDIN_FILENO || fstat(rd->fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return -1;
    if (strcmp(fx->output_file_name, "-") == 0 || fx->compress_output || g_rename_ids || getenv("FXH_HOST_PARSE")) return -1;
    if ((p->stages & FXG_STAGE_CLIP) && getenv("FXH_CLIP_SERIAL") != NULL && getenv("FXH_CLIP_PARALLEL") == NULL) return -1;      /* one aligner asked for */
    const off_t size = sb.st_size, here = lseek(rd->fd, 0, SEEK_CUR);
    const int lpr = fx->read_fastq ? 4 : 2;
    off_t cut[FXH_MAX_LANES + 1];
    cut[0] = 0; cut[k] = size;
    for (int r = 1; r < k; ++r) {
        cut[r] = fxh_find_cut(rd->fd, (off_t)((unsigned long long)size * (unsigned)r / (unsigned)k), size, lpr, 0);
        if (cut[r] < 0 || cut[r] <= cut[r - 1] || (r == 1 && cut[r] < here)) return -1;       /* small or odd input: one run */
    }
    /* The sharded attempt runs in a CHILD process.  Irregular input anywhere (or a cut that was no record boundary) abandons it: the
     * reference's behaviour -- message, exit code, what has been written before the bad record -- is defined for ONE stream, so the
     * child empties the parts and exits with FXH_EXIT_ABANDON, and this process -- which has not touched the GPU yet -- runs the same
     * input unsharded (part 0 then receives everything).  Nothing is ever exec'd or killed with device work in flight: the child ends
     * like any tool run, after its threads have been joined and its contexts destroyed. */
    if (g_hip_touched) return -1;                /* this process has used the HIP runtime already (a host that calls in twice): no fork over a live runtime */
    /* Every part is opened HERE, before anything has run: an output that cannot take parts -- /dev/null, a FIFO, a directory where the
     * sibling names cannot be created -- means one stream (part 0 alone, as named by the caller), never a failure halfway. */
    int part_fd[FXH_MAX_LANES];
    {
        struct stat ob;
        struct fxh_writer *w0 = fx->writer;
        if (!w0 || w0->fd < 0 || fstat(w0->fd, &ob) != 0 || !S_ISREG(ob.st_mode)) return -1;
        for (int r = 1; r < k; ++r) {
            char name[PATH_MAX + 16];
            fxh_part_name(fx, r, name, sizeof name);
            part_fd[r] = open(name, O_CREAT | O_WRONLY | O_TRUNC, 0666);
            if (part_fd[r] < 0 || fstat(part_fd[r], &ob) != 0 || !S_ISREG(ob.st_mode)) {
                warn("%s: cannot be an output part, running as one stream", name);
                for (int q = 1; q <= r; ++q) if (part_fd[q] >= 0) close(part_fd[q]);
                return -1;
            }
        }
    }
    fflush(NULL);
    const pid_t child = fork();
    if (child < 0) { for (int r = 1; r < k; ++r) close(part_fd[r]); return -1; }
    if (child > 0) {
        int st = 0;
        for (int r = 1; r < k; ++r) close(part_fd[r]);                                  /* the child writes them */
        while (waitpid(child, &st, 0) < 0) { if (errno != EINTR) err(1, "waitpid"); }
        if (WIFEXITED(st) && WEXITSTATUS(st) == FXH_EXIT_ABANDON) {
            if (lseek(rd->fd, here, SEEK_SET) < 0) err(1, "%s", fx->input_file_name);      /* the child read through the shared descriptor */
            struct fxh_writer *w = fx->writer;
            if (w && w->fd >= 0) { if (ftruncate(w->fd, 0) != 0 || lseek(w->fd, 0, SEEK_SET) < 0) warn("%s", fx->output_file_name); }
            return -1;
        }
        if (WIFSIGNALED(st)) { signal(WTERMSIG(st), SIG_DFL); raise(WTERMSIG(st)); _exit(128 + WTERMSIG(st)); }
        _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 1);                                      /* the child printed the reports and closed the parts */
    }
    (void)prctl(PR_SET_PDEATHSIG, SIGTERM);      /* the child: a tool process that was killed takes its sharded attempt along */
    fxh_part *pt = (fxh_part *)calloc((size_t)k, sizeof(fxh_part));
    if (!pt) err(1, "out of memory");
    const char *cap_env = getenv("FXH_READ_BUFFER_MB");
    for (int r = 0; r < k; ++r) {
        pt[r].p = p; pt[r].part = r; pt[r].nparts = k; pt[r].start = cut[r]; pt[r].limit = cut[r + 1];
        fxh_part_name(fx, r, pt[r].name, sizeof pt[r].name);
        if (r == 0) { pt[r].fx = fx; rd->limit = cut[1]; continue; }
        FASTX *f = (FASTX *)malloc(sizeof(FASTX));
        if (!f) err(1, "out of memory");
        memcpy(f, fx, sizeof(FASTX));
        f->reader = fxh_reader_open_range(fx->input_file_name, cap_env && atoi(cap_env) > 0 ? (size_t)atoi(cap_env) << 20 : 0, cut[r], cut[r + 1]);
        f->writer = fxh_writer_open_fd(part_fd[r]);
        f->input_line_number = 0; f->num_input_sequences = f->num_input_reads = f->num_output_sequences = f->num_output_reads = 0;
        pt[r].fx = f;
    }
    __atomic_store_n(&g_parts_abort, 0, __ATOMIC_RELAXED);
    g_parts_mode = 1;
    for (int r = 1; r < k; ++r) if (pthread_create(&pt[r].th, NULL, fxh_part_main, &pt[r]) != 0) err(1, "pthread_create");
    fxh_part_main(&pt[0]);
    for (int r = 1; r < k; ++r) pthread_join(pt[r].th, NULL);
    int bad = FXH_ABORTED();
    for (int r = 0; r < k; ++r) if (pt[r].rc != 0) bad = 1;
    {   /* clipper: every part found reads of one length -- it has to be the SAME length in all of them (a shorter read after a longer one
         * sees the longer one's tail); otherwise the parent runs the input as one stream, which goes serial where it must */
        uint32_t len0 = 0;
        for (int r = 0; r < k && !bad; ++r) { if (!g_part_clip_len[r]) continue; if (!len0) len0 = g_part_clip_len[r]; else if (g_part_clip_len[r] != len0) bad = 1; }
    }
    if (bad) {
        /* Abandoned.  Every thread of every part has been joined and its contexts are gone (fxh_lanes_stop destroys them for a part
         * that stops), the device is idle.  The parts are emptied through their own descriptors, part 0 -- whose descriptor the parent
         * shares -- is emptied here as well, and the process leaves with _exit: no exit handler of this half-finished attempt (the
         * writers' flush-at-exit, the runtime's) gets to run.  The parent then runs the input as one stream (see the fork above). */
        for (int r = 1; r < k; ++r) { struct fxh_writer *w = pt[r].fx->writer; w->len = 0; if (ftruncate(w->fd, 0) != 0) warn("%s", pt[r].name); close(w->fd); w->fd = -1; }
        { struct fxh_writer *w = fx->writer; w->len = 0; if (ftruncate(w->fd, 0) != 0 || lseek(w->fd, 0, SEEK_SET) < 0) warn("%s", pt[0].name); }
        if (getenv("FXH_TIMING")) fprintf(stderr, "fxh timing parts: abandoned, contexts destroyed, parts emptied\n");
        fflush(NULL);
        _exit(FXH_EXIT_ABANDON);
    }
    memset(tot, 0, sizeof *tot);
    FILE *ix = NULL;
    {
        char ixname[PATH_MAX + 8];
        const char *name = fx->output_file_name, *pr = strstr(name, "%r");
        if (pr) snprintf(ixname, sizeof ixname, "%.*sparts%s", (int)(pr - name), name, pr + 2); else snprintf(ixname, sizeof ixname, "%s.parts", name);
        ix = fopen(ixname, "w");
        if (ix) fprintf(ix, "#part\tfile\tinput_bytes\tinput_records\toutput_records\toutput_bytes\n");
    }
    for (int r = 0; r < k; ++r) {
        const fxh_totals *t = &pt[r].tot;
        tot->input_sequences += t->input_sequences; tot->input_reads += t->input_reads; tot->output_sequences += t->output_sequences; tot->output_reads += t->output_reads;
        tot->clip_input += t->clip_input; tot->clip_too_short += t->clip_too_short; tot->clip_adapter_only += t->clip_adapter_only;
        tot->clip_no_adapter += t->clip_no_adapter; tot->clip_adapter_found += t->clip_adapter_found; tot->clip_n += t->clip_n;
        tot->masked_reads += t->masked_reads; tot->masked_nucleotides += t->masked_nucleotides; tot->qtrim_dropped += t->qtrim_dropped;
        if (r > 0) fxh_writer_flush(pt[r].fx->writer);
        const off_t out_bytes = r == 0 ? fx->writer->off + (off_t)fx->writer->len : pt[r].fx->writer->off;
        if (ix) fprintf(ix, "%d\t%s\t%lld\t%zu\t%zu\t%lld\n", r, pt[r].name, (long long)(pt[r].limit - pt[r].start), t->input_sequences, t->output_sequences, (long long)out_bytes);
        if (r > 0) { fxh_writer_close(pt[r].fx->writer); free(pt[r].fx); }
    }
    if (ix) fclose(ix);
    fx->num_input_sequences = tot->input_sequences; fx->num_input_reads = tot->input_reads;
    fx->num_output_sequences = tot->output_sequences; fx->num_output_reads = tot->output_reads;
    free(pt);
    return 0;
}

/* `-o out.%r.fq` without FXH_PARTS: the caller has said where parts may go, the tool picks their number -- four (what one GPU's link and
 * four writer streams take, profiles/r03/l..q_e2e_parts*.txt) for a regular input file of at least 1 GB (FXH_AUTO_PARTS_MIN_MB), where
 * the ~0.1 s of three more contexts is paid back; one otherwise (part 0 then holds everything). */
int fxh_auto_parts(const FASTX *fx)
{
    struct stat sb;
    if (!strstr(fx->output_file_name, "%r") || strcmp(fx->output_file_name, "-") == 0) return 0;
    const char *me = getenv("FXH_AUTO_PARTS_MIN_MB");
    const long long min_bytes = (me ? atoll(me) : 1024ll) << 20;
    if (fx->reader->fd == STDIN_FILENO || fstat(fx->reader->fd, &sb) != 0 || !S_ISREG(sb.st_mode) || (long long)sb.st_size < min_bytes) return 1;
    return 4;
}

