/* fake_rccl.c -- TEST-ONLY stand-in for librccl.so.1, so that the product's RCCL transport (csrc/fxg_comm.h: fxg_comm_create,
 * fxg_epilogue_rccl) runs with world > 1 on a machine without GPUs.  The five NCCL entry points the transport binds, over one
 * shared-memory file per communicator; "device" pointers are host pointers (the emulation stub's), the stream is ignored.
 * Built into tests/emu/fakerccl/librccl.so.1 by tests/test_comm_cpu.py and reached only through LD_LIBRARY_PATH.  Never shipped.
 *
 * Under the REAL engine (GPU tier: two ranks of a job sharing the test box's one GPU, which RCCL itself refuses) the blocks are real device memory:
 * with FXG_FAKE_RCCL_HIP=1 copies go through the process's own HIP runtime (dlopen RTLD_NOLOAD): hipMemcpy behind a hipStreamSynchronize of the caller's stream.
 *
 *   FXG_FAKE_RCCL_LOG=<file>      one line per call (the test counts them)
 *   FXG_FAKE_RCCL_FAIL=id|init|gather   the named call returns ncclInternalError (error paths of the transport)
 *   FXG_FAKE_RCCL_TIMEOUT_S=<n>   how long a gather waits for the other ranks before it fails (default 30; 0 = for ever, like the real library)
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
#define SLOT_BYTES 4096
struct shared { volatile uint32_t arrived, generation, attached; uint32_t world; unsigned char slot[][SLOT_BYTES]; };
struct comm { struct shared *sh; size_t map_bytes; int rank, world; char path[128]; };

static void logline(const char *fmt, int a, int b)
{
    const char *f = getenv("FXG_FAKE_RCCL_LOG");
    if (!f) return;
    char line[96];
    const int n = snprintf(line, sizeof line, fmt, a, b);
    const int fd = open(f, O_WRONLY | O_CREAT | O_APPEND, 0600);
    if (fd >= 0) { if (write(fd, line, (size_t)n) < 0) { } close(fd); }
}
static int failing(const char *what) { const char *e = getenv("FXG_FAKE_RCCL_FAIL"); return e && strcmp(e, what) == 0; }

const char *ncclGetErrorString(int rc) { return rc == 0 ? "no error" : rc == 3 ? "internal error (fake)" : "unhandled error (fake)"; }

int ncclGetUniqueId(ncclUniqueId *id)
{
    logline("getid pid %d\n", (int)getpid(), 0);
    if (failing("id")) return 3;
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/tmp/fxg_fake_rccl_%d_%ld_%ld", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec);
    return 0;
}

int ncclCommInitRank(void **out, int world, ncclUniqueId id, int rank)
{
    logline("init rank %d world %d\n", rank, world);
    if (failing("init")) return 3;
    if (id.internal[0] != '/' || world < 1 || rank < 0 || rank >= world) return 4;
    struct comm *c = calloc(1, sizeof *c);
    if (!c) return 2;
    c->rank = rank; c->world = world;
    memcpy(c->path, id.internal, sizeof c->path);
    c->map_bytes = sizeof(struct shared) + (size_t)world * SLOT_BYTES;
    const int fd = open(c->path, O_RDWR | O_CREAT, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { if (fd >= 0) close(fd); free(c); return 2; }
    c->sh = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->sh == MAP_FAILED) { free(c); return 2; }
    c->sh->world = (uint32_t)world;
    __sync_fetch_and_add(&c->sh->attached, 1u);
    /* like the real call, return only once every rank has joined */
    for (int tries = 0; c->sh->attached < (uint32_t)world; ++tries) { if (tries > 20 * 30) { munmap(c->sh, c->map_bytes); free(c); return 3; } usleep(50000); }
    *out = c;
    return 0;
}

static int barrier(struct comm *c)
{
    const uint32_t gen = c->sh->generation;
    if (__sync_add_and_fetch(&c->sh->arrived, 1u) == (uint32_t)c->world) { c->sh->arrived = 0; __sync_synchronize(); c->sh->generation = gen + 1u; return 0; }
    const char *e = getenv("FXG_FAKE_RCCL_TIMEOUT_S");
    const long limit = e && *e ? atol(e) : 30;
    for (long tries = 0; c->sh->generation == gen; ++tries) { if (limit > 0 && tries > 2000 * limit) return 3; usleep(500); }
    return 0;
}

static int (*hip_memcpy)(void *, const void *, size_t, int);
static int (*hip_stream_sync)(void *);
static void hip_lookup(void)
{
    static int done;
    if (done) return;
    done = 1;
    if (!getenv("FXG_FAKE_RCCL_HIP")) return;            /* (the emulation stub links the runtime too, but its "device" memory is host memory) */
    void *h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("libamdhip64.so.6", RTLD_NOW | RTLD_NOLOAD);
    if (!h) return;
    hip_memcpy = (int (*)(void *, const void *, size_t, int))dlsym(h, "hipMemcpy");
    hip_stream_sync = (int (*)(void *))dlsym(h, "hipStreamSynchronize");
    if (!hip_memcpy || !hip_stream_sync) hip_memcpy = NULL;
}
static int copy(void *dst, const void *src, size_t n)
{
    if (hip_memcpy) return hip_memcpy(dst, src, n, 4 /* hipMemcpyDefault */) == 0 ? 0 : 3;
    memcpy(dst, src, n);
    return 0;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, void *stream)
{
    struct comm *c = comm;
    hip_lookup();
    if (hip_memcpy && hip_stream_sync(stream) != 0) return 3;          /* "behind the pass on the same stream" */
    logline("gather rank %d count %d\n", c ? c->rank : -1, (int)count);
    if (failing("gather")) return 3;
    static const size_t width[] = {1, 1, 4, 4, 8, 8, 2, 4, 8};          /* ncclInt8 .. ncclFloat64 */
    if (!c || dtype < 0 || dtype > 8 || count * width[dtype] > SLOT_BYTES) return 4;
    const size_t bytes = count * width[dtype];
    unsigned char tmp[SLOT_BYTES];
    if (copy(tmp, send, bytes)) return 3;
    memcpy((void *)c->sh->slot[c->rank], tmp, bytes);
    __sync_synchronize();
    if (barrier(c)) return 3;
    for (int g = 0; g < c->world; ++g) { memcpy(tmp, (const void *)c->sh->slot[g], bytes); if (copy((char *)recv + (size_t)g * bytes, tmp, bytes)) return 3; }
    return barrier(c);                                                     /* nobody overwrites a slot another rank still reads */
}

int ncclCommDestroy(void *comm)
{
    struct comm *c = comm;
    if (!c) return 4;
    logline("destroy rank %d\n", c->rank, 0);
    if (__sync_sub_and_fetch(&c->sh->attached, 1u) == 0) unlink(c->path);
    munmap(c->sh, c->map_bytes);
    free(c);
    return 0;
}
