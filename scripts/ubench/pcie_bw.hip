// pcie_bw.hip -- host <-> device copy rates of this box for the block sizes the tools move (not part of the product: a measurement aid).
// hipMemcpyAsync of 64 MB blocks from/to page-locked host memory (hipHostMalloc, or malloc + hipHostRegister as the tools' input buffers),
// on 1..8 streams, one direction or both at once.   hipcc --offload-arch=gfx950 -O3 pcie_bw.hip -o pcie_bw && ./pcie_bw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double run(int nstreams, int up, int down, bool registered, size_t blk, int reps)
{
    std::vector<hipStream_t> s(nstreams);
    std::vector<void *> hin(nstreams), hout(nstreams), din(nstreams), dout(nstreams);
    for (int i = 0; i < nstreams; ++i) {
        CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
        if (registered) { if (posix_memalign(&hin[i], 4096, blk)) exit(1); memset(hin[i], 1, blk); CK(hipHostRegister(hin[i], blk, hipHostRegisterPortable)); }
        else { CK(hipHostMalloc(&hin[i], blk, hipHostMallocPortable)); memset(hin[i], 1, blk); }
        CK(hipHostMalloc(&hout[i], blk, hipHostMallocPortable));
        memset(hout[i], 2, blk);
        CK(hipMalloc(&din[i], blk)); CK(hipMalloc(&dout[i], blk));
    }
    for (int i = 0; i < nstreams; ++i) { CK(hipMemcpyAsync(din[i], hin[i], blk, hipMemcpyHostToDevice, s[i])); CK(hipMemcpyAsync(hout[i], dout[i], blk, hipMemcpyDeviceToHost, s[i])); }
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r)
        for (int i = 0; i < nstreams; ++i) {
            // both directions: odd streams start with the download so that the two directions are in flight together
            if (up && down && (i & 1)) { CK(hipMemcpyAsync(hout[i], dout[i], blk, hipMemcpyDeviceToHost, s[i])); CK(hipMemcpyAsync(din[i], hin[i], blk, hipMemcpyHostToDevice, s[i])); continue; }
            if (up) CK(hipMemcpyAsync(din[i], hin[i], blk, hipMemcpyHostToDevice, s[i]));
            if (down) CK(hipMemcpyAsync(hout[i], dout[i], blk, hipMemcpyDeviceToHost, s[i]));
        }
    CK(hipDeviceSynchronize());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < nstreams; ++i) {
        if (registered) { CK(hipHostUnregister(hin[i])); free(hin[i]); } else CK(hipHostFree(hin[i]));
        CK(hipHostFree(hout[i])); CK(hipFree(din[i])); CK(hipFree(dout[i])); CK(hipStreamDestroy(s[i]));
    }
    return dt;
}

int main()
{
    const size_t blk = (size_t)64 << 20;
    const int reps = 24;
    printf("%-34s %8s %8s %8s\n", "64 MB blocks, GB/s", "up", "down", "sum");
    for (int reg = 0; reg < 2; ++reg)
        for (int ns : {1, 2, 4, 8}) {
            const double u = run(ns, 1, 0, reg, blk, reps), d = run(ns, 0, 1, reg, blk, reps), b = run(ns, 1, 1, reg, blk, reps);
            const double gb = (double)blk * reps * ns / 1e9;
            printf("%d stream(s), input %-14s %8.1f %8.1f   both at once: %.1f up + %.1f down = %.1f\n", ns, reg ? "hipHostRegister" : "hipHostMalloc", gb / u, gb / d, gb / b, gb / b, 2 * gb / b);
        }
    return 0;
}
