/* one_file_write.c -- how fast can T threads put G bytes into ONE file of a tmpfs / page cache?  (measurement aid, not product)
 *   p: pwrite() at exact offsets (every write takes the inode lock for allocation AND copy)
 *   f: T separate files (the sharded run's parts), one pwrite stream each
 *   m: one shared mapping over the final size, memcpy by the threads, one munmap at the end
 *   d: the same, each thread drops its finished block's page-table entries itself (MADV_DONTNEED takes mmap_lock shared)
 *   P: as d, the block's pages faulted in with one MADV_POPULATE_WRITE before the copy
 *   a: fallocate() of the whole file first (allocation without zeroing or copying, under the inode lock), then as d
 *   Q: fallocate() first, then as P
 *   A: ONE thread runs fallocate() ahead of the copies in 64 MiB steps; the T copy threads (as d) wait for their block to be allocated
 *   B: as A with MADV_POPULATE_WRITE before each copy
 *   w: fallocate() first, then parallel pwrite()
 *   r: pread() of a file written by a first pass (untimed)
 * usage: one_file_write <mode> <dir> <GiB> <threads> [block MiB]                                   */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static char mode; static int fd, T; static size_t total, blk; static char *map; static const char *dir;
static size_t next_blk; static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static size_t allocated; static pthread_mutex_t amu = PTHREAD_MUTEX_INITIALIZER; static pthread_cond_t acv = PTHREAD_COND_INITIALIZER;
static double t_falloc;

static void *allocator(void *arg)
{
    (void)arg;
    const size_t step = (size_t)64 << 20;
    double t0 = now();
    for (size_t o = 0; o < total; o += step) {
        size_t n = total - o < step ? total - o : step;
        if (fallocate(fd, 0, (off_t)o, (off_t)n) != 0) { perror("fallocate"); exit(1); }
        pthread_mutex_lock(&amu); allocated = o + n; pthread_cond_broadcast(&acv); pthread_mutex_unlock(&amu);
    }
    t_falloc = now() - t0;
    return NULL;
}

static void *worker(void *arg)
{
    const int id = (int)(long)arg;
    char *src = (char *)malloc(blk);
    memset(src, 'A' + id, blk);
    int myfd = fd;
    size_t myoff = 0;
    if (mode == 'f') { char name[512]; snprintf(name, sizeof name, "%s/ofw.%d", dir, id); myfd = open(name, O_CREAT | O_WRONLY | O_TRUNC, 0666); }
    for (;;) {
        pthread_mutex_lock(&mu); size_t b = next_blk++; pthread_mutex_unlock(&mu);
        size_t off = b * blk;
        if (off >= total) break;
        size_t n = total - off < blk ? total - off : blk;
        if (mode == 'r') { size_t d = 0; while (d < n) { ssize_t k = pread(myfd, src + d, n - d, (off_t)(off + d)); if (k <= 0) { perror("pread"); exit(1); } d += (size_t)k; } }
        else if (mode == 'p' || mode == 'w') { size_t d = 0; while (d < n) { ssize_t k = pwrite(myfd, src + d, n - d, (off_t)(off + d)); if (k <= 0) { perror("pwrite"); exit(1); } d += (size_t)k; } }
        else if (mode == 'f') { size_t d = 0; while (d < n) { ssize_t k = pwrite(myfd, src + d, n - d, (off_t)(myoff + d)); if (k <= 0) { perror("pwrite"); exit(1); } d += (size_t)k; } myoff += n; }
        else {
            if (mode == 'A' || mode == 'B') { pthread_mutex_lock(&amu); while (allocated < off + n) pthread_cond_wait(&acv, &amu); pthread_mutex_unlock(&amu); }
            if (mode == 'P' || mode == 'Q' || mode == 'B') if (madvise(map + off, n, MADV_POPULATE_WRITE) != 0) { perror("MADV_POPULATE_WRITE"); exit(1); }
            memcpy(map + off, src, n);
            if (mode != 'm') madvise(map + off, n, MADV_DONTNEED);
        }
    }
    if (mode == 'f') close(myfd);
    free(src);
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage\n"); return 2; }
    mode = argv[1][0]; dir = argv[2]; total = (size_t)(atof(argv[3]) * (1ull << 30)); T = atoi(argv[4]); blk = (size_t)(argc > 5 ? atoi(argv[5]) : 8) << 20;
    char name[512]; snprintf(name, sizeof name, "%s/ofw.out", dir);
    double t0 = now();
    fd = open(name, O_CREAT | O_RDWR | O_TRUNC, 0666);
    if (mode == 'r') {   /* fill the file first, untimed */
        char *b = (char *)malloc(blk); memset(b, 'x', blk);
        for (size_t o = 0; o < total; o += blk) if (pwrite(fd, b, blk, (off_t)o) < 0) { perror("fill"); return 1; }
        free(b); t0 = now();
    }
    const int mapped = strchr("mdPaQAB", mode) != NULL;
    const size_t span = total + total / 7;
    if (mapped) {
        if (ftruncate(fd, (off_t)span) != 0) { perror("ftruncate"); return 1; }
        map = (char *)mmap(NULL, span, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (map == MAP_FAILED) { perror("mmap"); return 1; }
    }
    if (mode == 'a' || mode == 'Q' || mode == 'w') { double a = now(); if (fallocate(fd, 0, 0, (off_t)total) != 0) { perror("fallocate"); return 1; } t_falloc = now() - a; }
    pthread_t th[64], ath;
    if (mode == 'A' || mode == 'B') pthread_create(&ath, NULL, allocator, NULL);
    for (int i = 0; i < T; ++i) pthread_create(&th[i], NULL, worker, (void *)(long)i);
    for (int i = 0; i < T; ++i) pthread_join(th[i], NULL);
    if (mode == 'A' || mode == 'B') pthread_join(ath, NULL);
    double t1 = now();
    if (map) { munmap(map, span); if (ftruncate(fd, (off_t)total) != 0) perror("ftruncate"); }
    close(fd);
    double t2 = now();
    printf("mode %c threads %2d block %2zu MiB: fallocate %.3f s, copy phase %.3f s, unmap+trim %.3f s, total %.3f s = %.2f GB/s\n", mode, T, blk >> 20, t_falloc, t1 - t0, t2 - t1, t2 - t0, total / (t2 - t0) / 1e9);
    unlink(name);
    if (mode == 'f') for (int i = 0; i < T; ++i) { snprintf(name, sizeof name, "%s/ofw.%d", dir, i); unlink(name); }
    return 0;
}
