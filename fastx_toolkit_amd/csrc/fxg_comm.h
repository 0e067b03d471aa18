// fxg_comm.h -- the multi-GPU host code of the C-ABI (include/fxg.h): shard ranges, the epilogue arithmetic, the
// concatenation of the ranks' packed slices and the RCCL transport of the counter blocks.  Host code only.
//
// Reads shard by contiguous index range and no read depends on another (the reference is one process over one stream,
// fastq_quality_trimmer.c:76-124), so the only exchange of a job is each rank's counter block: one ncclAllGather of
// FXG_NCOUNTERS u64 behind the pass, from which every rank derives the job totals and where its kept reads / kept bytes start.
//
// Included by fxg_engine.hip and -- so that the CPU tier runs this very code with world > 1 -- by tests/emu/fxg_stub.cpp.  It reaches
// device memory only through hooks the includer defines first:
//   FXG_COMM_FAIL(ctx, code, fmt, ...)   record a message, return code          FXG_COMM_SET_DEVICE(ctx)          -> bool
//   FXG_COMM_MALLOC(pp, bytes) -> bool   FXG_COMM_FREE(p)                       FXG_COMM_STREAM(ctx)              -> void * (hipStream_t)
//   FXG_COMM_SCRATCH(ctx) -> const uint64_t * (the context's own counter block) fxg_comm_d2h_sync(ctx, dst, src, bytes) -> error text or null
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>

extern "C" int fxg_shard_range(uint64_t n, uint32_t rank, uint32_t world, uint64_t *lo, uint64_t *hi)
{
    if (!lo || !hi || world == 0 || rank >= world) return FXG_E_INVALID;
    *lo = (uint64_t)(((unsigned __int128)n * rank) / world);
    *hi = (uint64_t)(((unsigned __int128)n * (rank + 1u)) / world);
    return FXG_OK;
}

extern "C" int fxg_epilogue(const uint64_t *gathered, uint32_t world, uint32_t rank, uint64_t totals[FXG_NCOUNTERS],
                            uint64_t *read_off, uint64_t *byte_off)
{
    if (!gathered || world == 0 || rank >= world) return FXG_E_INVALID;
    uint64_t ro = 0, bo = 0;
    if (totals) memset(totals, 0, FXG_NCOUNTERS * sizeof(uint64_t));
    for (uint32_t g = 0; g < world; ++g) {
        const uint64_t *c = gathered + (size_t)g * FXG_NCOUNTERS;
        if (g < rank) { ro += c[FXG_C_KEPT]; bo += c[FXG_C_KEPT_BASES]; }
        if (totals)
            for (int i = 0; i < FXG_NCOUNTERS; ++i) { if (i == FXG_C_ERRORS) totals[i] |= c[i]; else totals[i] += c[i]; }
    }
    if (read_off) *read_off = ro;
    if (byte_off) *byte_off = bo;
    return FXG_OK;
}

extern "C" int fxg_concat_pwrite(int fd, const void *host_buf, uint64_t bytes, uint64_t offset)
{
    if (fd < 0 || (!host_buf && bytes)) return FXG_E_INVALID;
    const char *p = (const char *)host_buf;
    while (bytes) {
        const ssize_t k = pwrite(fd, p, bytes > ((uint64_t)1 << 30) ? ((size_t)1 << 30) : (size_t)bytes, (off_t)offset);
        if (k < 0) { if (errno == EINTR) continue; return FXG_E_INVALID; }
        p += k; offset += (uint64_t)k; bytes -= (uint64_t)k;
    }
    return FXG_OK;
}

// ------------------------------------------------------------------------------------------------
// RCCL transport of the counter blocks (one process per GPU, C hosts).  librccl.so is opened at run time.
// Rendezvous: rank 0 publishes the ncclUniqueId through a file (written to a temporary name and renamed into place, so a reader
// sees a whole record or no file); the other ranks poll for it.  The file belongs to ONE job: rank 0 removes whatever lies under the
// name BEFORE it makes its id, and removes its own record once the communicator is up (every rank has read it by then).  A job that
// died between the two leaves a record behind; what keeps a later job's early ranks from taking it: the record carries the world size
// and a job token (FXG_COMM_JOB in the environment, hashed; launchers export one per job), a reader accepts only a record it has read
// twice, 100 ms apart, unchanged (rank 0 of its own job replaces or removes a stale one in between), and without a token the name
// itself has to be the job's own -- INTEGRATION.md says so.
// ------------------------------------------------------------------------------------------------
typedef struct { char internal[128]; } fxg_nccl_id;          // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef struct { char magic[8]; uint32_t world, reserved; uint64_t job; fxg_nccl_id id; } fxg_rdv_record;
static inline uint64_t fxg_rdv_job_token(void)
{
    const char *j = getenv("FXG_COMM_JOB");
    uint64_t h = 0xcbf29ce484222325ull;                       // FNV-1a; 0 = no token given
    if (!j || !*j) return 0;
    for (; *j; ++j) h = (h ^ (unsigned char)*j) * 0x100000001b3ull;
    return h ? h : 1;
}
static inline bool fxg_rdv_read(const char *file, uint32_t world, uint64_t job, fxg_rdv_record *r)
{
    const int fd = open(file, O_RDONLY);
    if (fd < 0) return false;
    char extra;
    const bool whole = read(fd, r, sizeof *r) == (ssize_t)sizeof *r && read(fd, &extra, 1) == 0;
    close(fd);
    return whole && memcmp(r->magic, "FXGRDV1", 8) == 0 && r->world == world && r->job == job;
}
struct fxg_comm {
    void *lib, *comm;
    uint32_t rank, world;
    uint64_t *d_gather;                                       // world * FXG_NCOUNTERS
    int (*get_id)(fxg_nccl_id *);
    int (*init_rank)(void **, int, fxg_nccl_id, int);
    int (*all_gather)(const void *, void *, size_t, int, void *, void *);
    int (*destroy)(void *);
    const char *(*err_string)(int);
};

extern "C" void fxg_comm_destroy(fxg_comm *m)
{
    if (!m) return;
    if (m->comm && m->destroy) (void)m->destroy(m->comm);
    FXG_COMM_FREE(m->d_gather);
    if (m->lib) dlclose(m->lib);
    free(m);
}

extern "C" int fxg_comm_create(fxg_ctx *c, const char *file, uint32_t rank, uint32_t world, int timeout_s, fxg_comm **out)
{
    if (!c || !file || !out || world == 0 || rank >= world) return FXG_E_INVALID;
    *out = nullptr;
    fxg_comm *m = (fxg_comm *)calloc(1, sizeof(fxg_comm));
    if (!m) return FXG_E_NOMEM;
    m->rank = rank; m->world = world;
    m->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!m->lib) m->lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!m->lib) { free(m); return FXG_COMM_FAIL(c, FXG_E_HIP, "RCCL is not installed (dlopen librccl.so: %s)", dlerror()); }
    m->get_id = (int (*)(fxg_nccl_id *))dlsym(m->lib, "ncclGetUniqueId");
    m->init_rank = (int (*)(void **, int, fxg_nccl_id, int))dlsym(m->lib, "ncclCommInitRank");
    m->all_gather = (int (*)(const void *, void *, size_t, int, void *, void *))dlsym(m->lib, "ncclAllGather");
    m->destroy = (int (*)(void *))dlsym(m->lib, "ncclCommDestroy");
    m->err_string = (const char *(*)(int))dlsym(m->lib, "ncclGetErrorString");
    if (!m->get_id || !m->init_rank || !m->all_gather || !m->destroy) { fxg_comm_destroy(m); return FXG_COMM_FAIL(c, FXG_E_HIP, "librccl.so lacks the NCCL entry points"); }
    const int wait_s = timeout_s > 0 ? timeout_s : 60;
    fxg_nccl_id id;
    memset(&id, 0, sizeof id);
    const uint64_t job = fxg_rdv_job_token();
    if (rank == 0) {                                          // publish the id atomically: write a temporary, rename it into place
        (void)unlink(file);                                   // whatever a dead job left under the name goes first
        const int rc = m->get_id(&id);
        if (rc != 0) {                                        // (the message is formatted before the library that owns the text is closed)
            const int frc = FXG_COMM_FAIL(c, FXG_E_HIP, "ncclGetUniqueId: %s", m->err_string ? m->err_string(rc) : "error");
            fxg_comm_destroy(m);
            return frc;
        }
        fxg_rdv_record rec;
        memset(&rec, 0, sizeof rec);
        memcpy(rec.magic, "FXGRDV1", 8); rec.world = world; rec.job = job; rec.id = id;
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp.%d", file, (int)getpid());
        errno = 0;
        const int fd = open(tmp, O_CREAT | O_WRONLY | O_TRUNC, 0600);
        int e = errno;
        bool ok = fd >= 0;
        if (ok) { const ssize_t k = write(fd, &rec, sizeof rec); if (k != (ssize_t)sizeof rec) { ok = false; e = k < 0 ? errno : EIO; } }      // a short write sets no errno
        if (fd >= 0 && close(fd) != 0 && ok) { ok = false; e = errno; }
        if (ok && rename(tmp, file) != 0) { ok = false; e = errno; }
        if (!ok) {
            (void)unlink(tmp);
            fxg_comm_destroy(m);
            return FXG_COMM_FAIL(c, FXG_E_INVALID, "cannot write the rendezvous file %s: %s", file, strerror(e));
        }
    } else {
        bool got = false;
        fxg_rdv_record a, b;
        for (int tries = 0; tries < wait_s * 20 && !got; ++tries) {
            if (fxg_rdv_read(file, world, job, &a)) {         // twice, 100 ms apart, unchanged: not a leftover that this job's rank 0 is about to replace
                usleep(100000);
                got = fxg_rdv_read(file, world, job, &b) && memcmp(&a, &b, sizeof a) == 0;
                tries += 2;
            }
            if (!got) usleep(50000);
        }
        if (!got) { fxg_comm_destroy(m); return FXG_COMM_FAIL(c, FXG_E_INVALID, "rank %u: no RCCL id in %s after %d s", rank, file, wait_s); }
        id = a.id;
    }
    if (!FXG_COMM_SET_DEVICE(c) || !FXG_COMM_MALLOC(&m->d_gather, (size_t)world * FXG_NCOUNTERS * sizeof(uint64_t))) {
        m->d_gather = nullptr;
        fxg_comm_destroy(m);
        if (rank == 0) (void)unlink(file);
        return FXG_COMM_FAIL(c, FXG_E_HIP, "allocating the gather buffer failed");
    }
    const int rc = m->init_rank(&m->comm, (int)world, id, (int)rank);
    if (rank == 0) (void)unlink(file);                        // every rank has joined (or the job has failed): the id is spent
    if (rc != 0) {
        const char *es = m->err_string ? m->err_string(rc) : "error";
        m->comm = nullptr;
        const int frc = FXG_COMM_FAIL(c, FXG_E_HIP, "ncclCommInitRank(rank %u of %u): %s", rank, world, es);
        fxg_comm_destroy(m);
        return frc;
    }
    *out = m;
    return FXG_OK;
}

extern "C" int fxg_epilogue_rccl(fxg_ctx *c, fxg_comm *m, const uint64_t *d_counters, uint64_t totals[FXG_NCOUNTERS], uint64_t *read_off,
                                 uint64_t *byte_off, uint64_t *gathered)
{
    if (!c || !m || !m->comm) return FXG_E_INVALID;
    if (!FXG_COMM_SET_DEVICE(c)) return FXG_COMM_FAIL(c, FXG_E_HIP, "selecting the context's device failed");
    const uint64_t *src = d_counters ? d_counters : FXG_COMM_SCRATCH(c);
    const int rc = m->all_gather(src, m->d_gather, FXG_NCOUNTERS, 5 /* ncclUint64 */, m->comm, FXG_COMM_STREAM(c));     // behind the pass on the same stream
    if (rc != 0) return FXG_COMM_FAIL(c, FXG_E_HIP, "ncclAllGather: %s", m->err_string ? m->err_string(rc) : "error");
    uint64_t *host = (uint64_t *)malloc((size_t)m->world * FXG_NCOUNTERS * sizeof(uint64_t));
    if (!host) return FXG_E_NOMEM;
    const char *e = fxg_comm_d2h_sync(c, host, m->d_gather, (size_t)m->world * FXG_NCOUNTERS * sizeof(uint64_t));
    if (e) { free(host); return FXG_COMM_FAIL(c, FXG_E_HIP, "copying the gathered counters failed: %s", e); }
    const int erc = fxg_epilogue(host, m->world, m->rank, totals, read_off, byte_off);
    if (gathered) memcpy(gathered, host, (size_t)m->world * FXG_NCOUNTERS * sizeof(uint64_t));
    free(host);
    return erc;
}
