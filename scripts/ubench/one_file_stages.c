/* one_file_stages.c -- ONE tmpfs file filled by a pipeline of stages, each with its own thread count (measurement aid, not product):
 *   F: fallocate() of windows ahead (allocation only: no zeroing, no copy)          nF = 0 | 1
 *   M: MADV_POPULATE_WRITE of blocks (zero + map; allocates too when F is absent)   nM >= 0
 *   C: memcpy into the mapping + MADV_DONTNEED (drop the page-table entries)        nC >= 1
 * A stage works on block b only when the stage before it has finished every block up to b.  Blocks are drawn in order.
 * gate = 1: allocation and copies exclude each other (a writer-preferring rwlock: F holds it exclusively around each fallocate, C shared around each copy);
 * head ms: the copy threads start that much later (the tool's device start-up, during which nothing is there to be written yet).
 * usage: one_file_stages <dir> <GiB> <nF> <nM> <nC> [block MiB] [window blocks] [zap 0|1] [gate 0|1] [head ms]                                   */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static int fd, zap = 1, gate = 0, head_ms = 0; static pthread_rwlock_t gl; static double t_copy0; static size_t total, blk, nblk, win; static char *map;
typedef struct { pthread_mutex_t mu; pthread_cond_t cv; size_t next, done_upto; unsigned char *done; double busy; } stage;
static stage SF, SM, SC;
static int nF, nM, nC;
static void stage_init(stage *s) { pthread_mutex_init(&s->mu, NULL); pthread_cond_init(&s->cv, NULL); s->done = calloc(nblk + 1, 1); }
static void stage_finish(stage *s, size_t b0, size_t b1)
{
    pthread_mutex_lock(&s->mu);
    for (size_t b = b0; b < b1; ++b) s->done[b] = 1;
    while (s->done_upto < nblk && s->done[s->done_upto]) s->done_upto++;
    pthread_cond_broadcast(&s->cv);
    pthread_mutex_unlock(&s->mu);
}
static void stage_wait(stage *s, size_t upto) { pthread_mutex_lock(&s->mu); while (s->done_upto < upto) pthread_cond_wait(&s->cv, &s->mu); pthread_mutex_unlock(&s->mu); }
static size_t draw(stage *s, size_t step) { pthread_mutex_lock(&s->mu); size_t b = s->next; s->next += step; pthread_mutex_unlock(&s->mu); return b; }

static void *run_F(void *a)
{
    (void)a;
    for (;;) {
        size_t b = draw(&SF, win);
        if (b >= nblk) break;
        size_t e = b + win < nblk ? b + win : nblk, off = b * blk, n = (e * blk < total ? e * blk : total) - off;
        if (gate) pthread_rwlock_wrlock(&gl);
        double t = now();
        if (fallocate(fd, 0, (off_t)off, (off_t)n) != 0) { perror("fallocate"); exit(1); }
        SF.busy += now() - t;
        if (gate) pthread_rwlock_unlock(&gl);
        stage_finish(&SF, b, e);
    }
    return NULL;
}
static void *run_M(void *a)
{
    (void)a;
    double busy = 0;
    for (;;) {
        size_t b = draw(&SM, 1);
        if (b >= nblk) break;
        if (nF) stage_wait(&SF, b + 1);
        size_t off = b * blk, n = total - off < blk ? total - off : blk;
        double t = now();
        if (madvise(map + off, n, MADV_POPULATE_WRITE) != 0) { perror("populate"); exit(1); }
        busy += now() - t;
        stage_finish(&SM, b, b + 1);
    }
    pthread_mutex_lock(&SM.mu); SM.busy += busy; pthread_mutex_unlock(&SM.mu);
    return NULL;
}
static void *run_C(void *a)
{
    const int id = (int)(long)a;
    char *src = (char *)malloc(blk);
    memset(src, 'A' + id, blk);
    double busy = 0;
    if (head_ms) usleep((useconds_t)head_ms * 1000);
    for (;;) {
        size_t b = draw(&SC, 1);
        if (b >= nblk) break;
        if (nM) stage_wait(&SM, b + 1); else if (nF) stage_wait(&SF, b + 1);
        size_t off = b * blk, n = total - off < blk ? total - off : blk;
        if (gate) pthread_rwlock_rdlock(&gl);
        double t = now();
        memcpy(map + off, src, n);
        if (zap) madvise(map + off, n, MADV_DONTNEED);
        busy += now() - t;
        if (gate) pthread_rwlock_unlock(&gl);
    }
    pthread_mutex_lock(&SC.mu); SC.busy += busy; pthread_mutex_unlock(&SC.mu);
    free(src);
    return NULL;
}
int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage\n"); return 2; }
    const char *dir = argv[1]; total = (size_t)(atof(argv[2]) * (1ull << 30)); nF = atoi(argv[3]); nM = atoi(argv[4]); nC = atoi(argv[5]);
    blk = (size_t)(argc > 6 ? atoi(argv[6]) : 8) << 20; win = argc > 7 ? (size_t)atoi(argv[7]) : 8; zap = argc > 8 ? atoi(argv[8]) : 1; gate = argc > 9 ? atoi(argv[9]) : 0; head_ms = argc > 10 ? atoi(argv[10]) : 0;
    { pthread_rwlockattr_t ra; pthread_rwlockattr_init(&ra); pthread_rwlockattr_setkind_np(&ra, PTHREAD_RWLOCK_PREFER_WRITER_NONRECURSIVE_NP); pthread_rwlock_init(&gl, &ra); }
    nblk = (total + blk - 1) / blk;
    char name[512]; snprintf(name, sizeof name, "%s/ofs.out", dir);
    stage_init(&SF); stage_init(&SM); stage_init(&SC);
    double t0 = now();
    fd = open(name, O_CREAT | O_RDWR | O_TRUNC, 0666);
    const size_t span = total + total / 7;
    if (ftruncate(fd, (off_t)span) != 0) { perror("ftruncate"); return 1; }
    map = (char *)mmap(NULL, span, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (map == MAP_FAILED) { perror("mmap"); return 1; }
    pthread_t th[128]; int k = 0;
    for (int i = 0; i < nF; ++i) pthread_create(&th[k++], NULL, run_F, NULL);
    for (int i = 0; i < nM; ++i) pthread_create(&th[k++], NULL, run_M, NULL);
    for (int i = 0; i < nC; ++i) pthread_create(&th[k++], NULL, run_C, (void *)(long)i);
    for (int i = 0; i < k; ++i) pthread_join(th[i], NULL);
    double t1 = now();
    munmap(map, span); if (ftruncate(fd, (off_t)total) != 0) perror("ftruncate"); close(fd);
    double t2 = now();
    printf("gate %d head %3d ms: after the head start %.3f s = %.2f GB/s | ", gate, head_ms, t2 - t0 - 1e-3 * head_ms, total / (t2 - t0 - 1e-3 * head_ms) / 1e9);
    printf("F %d M %d C %2d block %zu MiB window %zu zap %d: stages %.3f s (busy: F %.3f, M %.3f summed, C %.3f summed), unmap+trim %.3f s, total %.3f s = %.2f GB/s\n",
           nF, nM, nC, blk >> 20, win, zap, t1 - t0, SF.busy, SM.busy, SC.busy, t2 - t1, t2 - t0, total / (t2 - t0) / 1e9);
    unlink(name);
    return 0;
}
