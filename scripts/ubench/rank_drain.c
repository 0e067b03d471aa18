/* rank_drain.c -- P processes (the ranks of a job), T writer threads each, put their contiguous slices of G bytes into ONE file of a tmpfs.  (measurement aid, not product)
 *   w: pwrite() at exact offsets into a file that has its size but no pages (every write allocates under the ONE inode lock, whatever process it comes from)
 *   a: process 0 fallocate()s the whole file first (timed on its own: in a job it runs while the GPUs still compute, nothing else touches the file),
 *      then every process maps ITS slice and copies into pages that exist (separate processes: separate page tables, separate mmap_locks), MADV_DONTNEED per block
 *   x: as a, but copies by pwrite() into the allocated file
 * usage: rank_drain <mode> <dir> <GiB> <processes> <threads> [block MiB]                                   */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static char mode; static int fd, T; static size_t blk, lo, hi, next_blk; static char *map, *src;
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;

static void *worker(void *arg)
{
    (void)arg;
    for (;;) {
        pthread_mutex_lock(&mu);
        const size_t o = next_blk; next_blk += blk;
        pthread_mutex_unlock(&mu);
        if (o >= hi) break;
        const size_t n = hi - o < blk ? hi - o : blk;
        if (mode == 'a') { memcpy(map + (o - lo), src, n); madvise(map + (o - lo), n, MADV_DONTNEED); }
        else { size_t d = 0; while (d < n) { ssize_t k = pwrite(fd, src + d, n - d, (off_t)(o + d)); if (k <= 0) { perror("pwrite"); exit(1); } d += (size_t)k; } }
    }
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: rank_drain <w|a|x> <dir> <GiB> <processes> <threads> [block MiB]\n"); return 2; }
    mode = argv[1][0];
    const size_t total = (size_t)(atof(argv[3]) * (double)(1ull << 30)) & ~(size_t)4095;
    const int P = atoi(argv[4]); T = atoi(argv[5]);
    blk = (size_t)(argc > 6 ? atoi(argv[6]) : 32) << 20;
    char path[4096]; snprintf(path, sizeof path, "%s/rank_drain.%d", argv[2], (int)getpid());
    fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)total) != 0) { perror(path); return 1; }
    double t_alloc = 0;
    if (mode != 'w') { const double t0 = now(); if (fallocate(fd, 0, 0, (off_t)total) != 0) { perror("fallocate"); return 1; } t_alloc = now() - t0; }
    src = malloc(blk); memset(src, 'x', blk);
    const double t0 = now();
    for (int p = 0; p < P; ++p) {
        if (fork() == 0) {
            lo = (total / (size_t)P * (size_t)p) & ~(size_t)4095; hi = p + 1 == P ? total : (total / (size_t)P * (size_t)(p + 1)) & ~(size_t)4095; next_blk = lo;
            if (mode == 'a') { map = mmap(NULL, hi - lo, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)lo); if (map == MAP_FAILED) { perror("mmap"); _exit(1); } }
            pthread_t th[64];
            for (int i = 0; i < T; ++i) pthread_create(&th[i], NULL, worker, NULL);
            for (int i = 0; i < T; ++i) pthread_join(th[i], NULL);
            _exit(0);
        }
    }
    for (int p = 0; p < P; ++p) { int st; wait(&st); }
    const double dt = now() - t0;
    printf("mode %c: %.2f GiB, %d processes x %d threads, %zu MiB blocks: copies %.3f s = %.2f GB/s", mode, (double)total / (1ull << 30), P, T, blk >> 20, dt, 1e-9 * (double)total / dt);
    if (mode != 'w') printf("; fallocate before them %.3f s = %.2f GB/s", t_alloc, 1e-9 * (double)total / t_alloc);
    printf("\n");
    unlink(path);
    return 0;
}
