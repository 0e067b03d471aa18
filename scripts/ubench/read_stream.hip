// read_stream.hip -- what the part gives a READ-ONLY stream of 15 GB (two 7.5 GB arrays) by the way it is asked for (round 6; not part of the product):
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/read_stream.hip -o scripts/ubench/read_stream && scripts/ubench/read_stream
//   vgpr  : global_load_dwordx4 into registers, D loads per array in flight per lane, default / non-temporal policy
//   lds   : LDS-DMA (buffer_load_dwordx4 ... lds, 1 KB per wave instruction, no registers), D KB per array in flight per wave, default / nt; the consumer
//           reads one dword per lane of what landed (the statistics kernel would read all 16 bytes)
// geometry: 256 workgroups of 960 threads (the statistics kernel's) and 1024 of 256; every workgroup streams chunks dealt round-robin (chunk = one trip of
// the workgroup).  The question behind it: is the statistics kernel's ceiling (its loads alone = 2.49 ms = what `mix_rw` calls a plain read) the part's, or the
// access form's?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lptr_t;
typedef __attribute__((address_space(3))) unsigned char lds_u8;

template <int D, bool NT>
__global__ void k_vgpr(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b, u32x4 *o, u64 n)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 xa[D], xb[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { const u64 j = i + d * stride < n ? i + d * stride : 0; xa[d] = NT ? __builtin_nontemporal_load(a + j) : a[j]; xb[d] = NT ? __builtin_nontemporal_load(b + j) : b[j]; }
    for (; i < n; i += D * stride) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const u32x4 ca = xa[d], cb = xb[d];
            const u64 j = i + (u64)(D + d) * stride < n ? i + (u64)(D + d) * stride : 0;
            xa[d] = NT ? __builtin_nontemporal_load(a + j) : a[j]; xb[d] = NT ? __builtin_nontemporal_load(b + j) : b[j];
            if (i + d * stride < n) acc += ca ^ cb;
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) o[0] = acc;
}

// the same stream with cfg2's share of it written back (73 of 75 units of ONE array's worth: 7.3 GB), non-temporal stores; NTL: non-temporal loads
template <bool NTL>
__global__ void k_mix(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b, u32x4 *o, u64 n)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 xa = NTL ? __builtin_nontemporal_load(a + i) : a[i], xb = NTL ? __builtin_nontemporal_load(b + i) : b[i];
    for (; i < n; i += stride) {
        const u32x4 ca = xa, cb = xb;
        const u64 j = i + stride < n ? i + stride : 0;
        xa = NTL ? __builtin_nontemporal_load(a + j) : a[j]; xb = NTL ? __builtin_nontemporal_load(b + j) : b[j];
        const u64 g = i / 75u, r = i - g * 75u;
        if (r < 73u) __builtin_nontemporal_store(ca ^ cb, o + g * 73u + r);
    }
}

// cfg4's shape: both arrays read, 141 of every 150 units written back to two outputs (revcomp + trim -f 5 -l 145 keeps 141 of 150 bytes), non-temporal stores
template <bool NTL>
__global__ void k_copy2(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b, u32x4 *o1, u32x4 *o2, u64 n)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 xa = NTL ? __builtin_nontemporal_load(a + i) : a[i], xb = NTL ? __builtin_nontemporal_load(b + i) : b[i];
    for (; i < n; i += stride) {
        const u32x4 ca = xa, cb = xb;
        const u64 j = i + stride < n ? i + stride : 0;
        xa = NTL ? __builtin_nontemporal_load(a + j) : a[j]; xb = NTL ? __builtin_nontemporal_load(b + j) : b[j];
        const u64 g = i / 150u, r = i - g * 150u;
        if (r < 141u) { __builtin_nontemporal_store(ca, o1 + g * 141u + r); __builtin_nontemporal_store(cb, o2 + g * 141u + r); }
    }
}

// every wave owns a ring of D slots of 2 KB (1 KB per array); trip t of the workgroup = units [t * W * 64, (t + 1) * W * 64) (W waves), wave w takes its 64
template <int D, int AUX>
__global__ void k_lds(const unsigned char *__restrict__ a, const unsigned char *__restrict__ b, u32 *o, u64 nunits)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    lds_u8 *ring = (lds_u8 *)smem + wave * (D * 2048u);
    const u64 per_trip = (u64)W * 64u, trips = (nunits + per_trip - 1) / per_trip;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)a, 0, 0x7FFFFFFC, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)b, 0, 0x7FFFFFFC, 0x00020000);
    (void)ra; (void)rb;
    u32 acc = 0;
    // a descriptor covers 2 GB: rebase it per trip (scalar), the lane offset stays below 1 KB
    auto issue = [&](u64 t, u32 slot) {
        const u64 unit0 = (t * gridDim.x + blockIdx.x) * per_trip + (u64)wave * 64u;        // round-robin trips over the workgroups
        if (unit0 + 64u <= nunits) {
            const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void *)(a + unit0 * 16u), 0, 1024, 0x00020000);
            const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void *)(b + unit0 * 16u), 0, 1024, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lptr_t *)(ring + slot * 2048u), 16, (int)(lane << 4), 0, 0, AUX);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lptr_t *)(ring + slot * 2048u + 1024u), 16, (int)(lane << 4), 0, 0, AUX);
        }
    };
    const u64 my_trips = (trips + gridDim.x - 1 - blockIdx.x) / gridDim.x;
#pragma unroll
    for (int d = 0; d < D; ++d) if ((u64)d < my_trips) issue(d, d);
    for (u64 t = 0; t < my_trips; t += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (t + d >= my_trips) break;
            // slot d landed when at most 2 (D - 1) of the wave's DMA loads are still out
            if (t + d + D > my_trips + 0ull) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail: fewer loads behind this slot than the counts below assume
            if (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (D == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (D == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            if (D == 8) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            const u32 x = *(const volatile u32 *)(smem + wave * (D * 2048u) + d * 2048u + lane * 16u) ^ *(const volatile u32 *)(smem + wave * (D * 2048u) + d * 2048u + 1024u + lane * 16u);
            acc ^= x;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + d + D < my_trips) issue(t + d + D, d);
            else { /* keep the counter pattern: nothing more to ask for */ }
        }
        if (t + D >= my_trips) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc == 0x12345678u) o[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename F> static float timeit(F launch)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main()
{
    const u64 bytes = 7500000000ull / 61440 * 61440, n = bytes / 16;       // whole trips of 960 and of 256 lanes
    unsigned char *a, *b; u32x4 *o, *o2;
    CK(hipMalloc(&a, bytes + 4096)); CK(hipMalloc(&b, bytes + 4096)); CK(hipMalloc(&o, bytes + 4096)); CK(hipMalloc(&o2, bytes + 4096));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    for (int geo = 0; geo < 3; ++geo) {
        const int grid = geo == 2 ? 2048 : geo ? 1024 : 256, block = geo ? 256 : 960, W = block / 64;
#define VG(D, NT) { float ms = timeit([&] { hipLaunchKernelGGL((k_vgpr<D, NT>), dim3(grid), dim3(block), 0, 0, (const u32x4 *)a, (const u32x4 *)b, o, n); }); \
                    printf("grid %4d x %3d  vgpr depth %d %-7s  %7.3f ms  %5.2f TB/s\n", grid, block, D, NT ? "nt" : "default", ms, 2.0 * bytes / ms / 1e9); }
        VG(1, false) VG(2, false) VG(4, false) VG(1, true) VG(2, true) VG(4, true)
#define LD(D, AUX) { CK(hipFuncSetAttribute((const void *)k_lds<D, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, W * D * 2048)); \
                     float ms = timeit([&] { hipLaunchKernelGGL((k_lds<D, AUX>), dim3(grid), dim3(block), W * D * 2048, 0, a, b, (u32 *)o, n); }); \
                     printf("grid %4d x %3d  lds  depth %d %-7s  %7.3f ms  %5.2f TB/s\n", grid, block, D, AUX ? "nt" : "default", ms, 2.0 * bytes / ms / 1e9); }
        LD(1, 0) LD(2, 0) LD(4, 0) LD(1, 2) LD(2, 2) LD(4, 2)
#define MX(NTL) { float ms = timeit([&] { hipLaunchKernelGGL((k_mix<NTL>), dim3(grid), dim3(block), 0, 0, (const u32x4 *)a, (const u32x4 *)b, o, n); }); \
                  printf("grid %4d x %3d  mix: 15 GB read (%s) + 7.3 GB written (nt)  %7.3f ms  %5.2f TB/s\n", grid, block, NTL ? "nt" : "default", ms, (2.0 + 73.0 / 75.0) * bytes / ms / 1e9); }
        MX(false) MX(true)
#define CP(NTL) { float ms = timeit([&] { hipLaunchKernelGGL((k_copy2<NTL>), dim3(grid), dim3(block), 0, 0, (const u32x4 *)a, (const u32x4 *)b, o, o2, n); }); \
                  printf("grid %4d x %3d  copy: 15 GB read (%s) + 14.1 GB written (nt), cfg4's shape  %7.3f ms  %5.2f TB/s\n", grid, block, NTL ? "nt" : "default", ms, (2.0 + 2.0 * 141.0 / 150.0) * bytes / ms / 1e9); }
        CP(false) CP(true)
    }
    CK(hipDeviceSynchronize());
    return 0;
}
