// register_filemap.hip -- can the DMA engines read and write the PAGE CACHE of a tmpfs file directly?  (measurement aid, not product)
// A file mapping (MAP_SHARED) is page-locked window by window with hipHostRegister and used as the source / destination of
// hipMemcpyAsync: no pread() into a staging buffer on the way up, no pwrite()/memcpy on the way down.  Prints the cost of
// register + unregister per window and the copy rates, next to pread() / memcpy into hipHostMalloc staging.
//   hipcc --offload-arch=gfx950 -O3 register_filemap.hip -o register_filemap && ./register_filemap /dev/shm 4
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t total = (size_t)(argc > 2 ? atoi(argv[2]) : 2) << 30, win = (size_t)64 << 20;
    char in[512], out[512];
    snprintf(in, sizeof in, "%s/rfm.in", dir); snprintf(out, sizeof out, "%s/rfm.out", dir);
    {   // input file: written with write()
        int fd = open(in, O_CREAT | O_WRONLY | O_TRUNC, 0666);
        char *b = (char *)malloc(win); for (size_t i = 0; i < win; ++i) b[i] = (char)(i * 2654435761u >> 13);
        for (size_t o = 0; o < total; o += win) if (write(fd, b, win) != (ssize_t)win) { perror("write"); return 1; }
        close(fd); free(b);
    }
    void *dbuf; CK(hipMalloc(&dbuf, win));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    void *stage; CK(hipHostMalloc(&stage, win, hipHostMallocPortable)); memset(stage, 0, win);
    // ---- up: pread into pinned staging + H2D
    {
        int fd = open(in, O_RDONLY);
        double t0 = now(), tr = 0;
        for (size_t o = 0; o < total; o += win) { double a = now(); if (pread(fd, stage, win, (off_t)o) != (ssize_t)win) { perror("pread"); return 1; } tr += now() - a; CK(hipMemcpyAsync(dbuf, stage, win, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
        double dt = now() - t0;
        printf("up   pread -> pinned staging -> H2D      : %.3f s (pread %.3f s = %.1f GB/s one thread), %.1f GB/s overall\n", dt, tr, total / tr / 1e9, total / dt / 1e9);
        close(fd);
    }
    // ---- up: register windows of the file mapping, H2D straight from the page cache
    {
        int fd = open(in, O_RDONLY);
        char *m = (char *)mmap(NULL, total, PROT_READ, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { perror("mmap in"); return 1; }
        double t0 = now(), treg = 0, tcp = 0, tun = 0; int ok = 1;
        for (size_t o = 0; o < total && ok; o += win) {
            double a = now();
            hipError_t e = hipHostRegister(m + o, win, hipHostRegisterPortable | hipHostRegisterReadOnly);
            if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostRegister(m + o, win, hipHostRegisterPortable); }
            if (e != hipSuccess) { printf("up   hipHostRegister(file mapping, PROT_READ) FAILED: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); ok = 0; break; }
            double b = now(); treg += b - a;
            CK(hipMemcpyAsync(dbuf, m + o, win, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
            double c = now(); tcp += c - b;
            CK(hipHostUnregister(m + o)); tun += now() - c;
        }
        double dt = now() - t0;
        if (ok) printf("up   register window of the mapping -> H2D : %.3f s: register %.3f, copy %.3f (%.1f GB/s), unregister %.3f; %.1f GB/s overall\n", dt, treg, tcp, total / tcp / 1e9, tun, total / dt / 1e9);
        if (!ok) {   // PROT_READ|PROT_WRITE private? try a writable shared mapping
            munmap(m, total); close(fd); fd = open(in, O_RDWR);
            m = (char *)mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            hipError_t e = hipHostRegister(m, win, hipHostRegisterPortable);
            printf("up   hipHostRegister(file mapping, PROT_READ|PROT_WRITE): %s\n", hipGetErrorString(e)); (void)hipGetLastError();
            if (e == hipSuccess) CK(hipHostUnregister(m));
        }
        munmap(m, total); close(fd);
    }
    // ---- down: D2H into staging + pwrite
    {
        int fd = open(out, O_CREAT | O_RDWR | O_TRUNC, 0666);
        double t0 = now(), tw = 0;
        for (size_t o = 0; o < total; o += win) { CK(hipMemcpyAsync(stage, dbuf, win, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); double a = now(); if (pwrite(fd, stage, win, (off_t)o) != (ssize_t)win) { perror("pwrite"); return 1; } tw += now() - a; }
        double dt = now() - t0;
        printf("down D2H -> pinned staging -> pwrite       : %.3f s (pwrite %.3f s = %.1f GB/s one thread), %.1f GB/s overall\n", dt, tw, total / tw / 1e9, total / dt / 1e9);
        close(fd);
    }
    // ---- down: register windows of the (sparse) output mapping, D2H straight into the page cache
    {
        int fd = open(out, O_CREAT | O_RDWR | O_TRUNC, 0666);
        if (ftruncate(fd, (off_t)total) != 0) { perror("ftruncate"); return 1; }
        char *m = (char *)mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { perror("mmap out"); return 1; }
        double t0 = now(), treg = 0, tcp = 0, tun = 0; int ok = 1;
        for (size_t o = 0; o < total && ok; o += win) {
            double a = now();
            hipError_t e = hipHostRegister(m + o, win, hipHostRegisterPortable);
            if (e != hipSuccess) { printf("down hipHostRegister(output mapping) FAILED: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); ok = 0; break; }
            double b = now(); treg += b - a;
            CK(hipMemcpyAsync(m + o, dbuf, win, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
            double c = now(); tcp += c - b;
            CK(hipHostUnregister(m + o)); tun += now() - c;
            madvise(m + o, win, MADV_DONTNEED);
        }
        double dt = now() - t0, t1 = now();
        munmap(m, total); close(fd);
        if (ok) printf("down register window of the mapping <- D2H : %.3f s: register %.3f, copy %.3f (%.1f GB/s), unregister %.3f; %.1f GB/s overall; munmap %.3f s\n", dt, treg, tcp, total / tcp / 1e9, tun, total / dt / 1e9, now() - t1);
        if (ok) {   // did the bytes land in the file?
            int fd2 = open(out, O_RDONLY); char *chk = (char *)malloc(win); char *src = (char *)malloc(win);
            CK(hipMemcpy(src, dbuf, win, hipMemcpyDeviceToHost));
            int same = pread(fd2, chk, win, (off_t)(total - win)) == (ssize_t)win && memcmp(chk, src, win) == 0;
            printf("down file content == device buffer: %s\n", same ? "yes" : "NO");
            close(fd2); free(chk); free(src);
        }
    }
    unlink(in); unlink(out);
    return 0;
}
