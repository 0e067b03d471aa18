// mix_rw.hip -- what the memory system gives a perfectly regular streaming kernel at cfg2's read : write mix.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mix_rw.hip -o scripts/ubench/mix_rw && scripts/ubench/mix_rw
// Each workgroup streams 16-byte units, fully coalesced, no dependencies between workgroups, nothing computed:
//   read  : sum of two 7.5 GB arrays into a register (kept alive by a never-true store)           15 GB read
//   write : 7.3 GB written                                                                         7.3 GB written
//   mix   : read both arrays AND write 7.3 GB (every lane writes ~0.49 of what it reads)          15 GB read + 7.3 GB written = cfg2's algorithmic bytes
//   copy  : 7.5 GB -> 7.5 GB                                                                       7.5 GB read + 7.5 GB written
// The cfg2 kernels (fxg_kernel_tiles<0,0> and fxg_kernel_rows<38>) cannot beat `mix`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>   // 0 read, 1 write, 2 mix, 3 copy; NT: non-temporal stores
__global__ __launch_bounds__(256) void k(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b, u32x4 *__restrict__ o, size_t n, size_t nout)
{
    const size_t stride = (size_t)gridDim.x * 256u;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += stride) {
        if (MODE == 0 || MODE == 2) { const u32x4 x = a[i], y = b[i]; acc += x ^ y; }
        if (MODE == 3) acc = a[i];
        if (MODE == 1) { if (NT) __builtin_nontemporal_store(acc + (u32)i, o + i); else o[i] = acc + (u32)i; }
        if (MODE == 2) { const size_t g = i / 75u, r = i - g * 75u; if (r < 73u) { if (NT) __builtin_nontemporal_store(acc + (u32)i, o + g * 73u + r); else o[g * 73u + r] = acc + (u32)i; } }   // 73 of 75 units: 7.3 GB of 7.5
        if (MODE == 3) { if (NT) __builtin_nontemporal_store(acc, o + i); else o[i] = acc; }
    }
    if (MODE == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) o[0] = acc;    // never true in practice: keeps the loads
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const size_t n = 7500000000ull / 16 / 75 * 75, nout = n / 75 * 73;
    u32x4 *a, *b, *o;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&o, n * 16));
    CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 2, n * 16)); CK(hipMemset(o, 0, n * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"read 15 GB", "write 7.3 GB", "mix 15 GB read + 7.3 GB written", "copy 7.5 GB -> 7.5 GB"};
    const double gb[] = {15.0, 7.3, 22.3, 15.0};
    for (int nt = 0; nt < 2; ++nt)
    for (int blocks_per_cu = 2; blocks_per_cu <= 8; blocks_per_cu *= 2) {
        const int grid = 256 * blocks_per_cu;
        for (int mode = 0; mode < 4; ++mode) {
            if (nt && mode == 0) continue;
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL((k<0, false>), dim3(grid), dim3(256), 0, 0, a, b, o, n, nout);
                if (mode == 1 && !nt) hipLaunchKernelGGL((k<1, false>), dim3(grid), dim3(256), 0, 0, a, b, o, nout, nout);
                if (mode == 2 && !nt) hipLaunchKernelGGL((k<2, false>), dim3(grid), dim3(256), 0, 0, a, b, o, n, nout);
                if (mode == 3 && !nt) hipLaunchKernelGGL((k<3, false>), dim3(grid), dim3(256), 0, 0, a, b, o, n, nout);
                if (mode == 1 && nt) hipLaunchKernelGGL((k<1, true>), dim3(grid), dim3(256), 0, 0, a, b, o, nout, nout);
                if (mode == 2 && nt) hipLaunchKernelGGL((k<2, true>), dim3(grid), dim3(256), 0, 0, a, b, o, n, nout);
                if (mode == 3 && nt) hipLaunchKernelGGL((k<3, true>), dim3(grid), dim3(256), 0, 0, a, b, o, n, nout);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("grid %5d %s %-34s %7.3f ms  %6.2f TB/s\n", grid, nt ? "nt-stores" : "plain    ", names[mode], best, gb[mode] / best);
        }
    }
    return 0;
}
