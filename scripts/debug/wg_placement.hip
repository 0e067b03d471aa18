// Which workgroups of a 4-per-CU launch share a CU?  (round 4: the clip kernel's workgroups run in lock-step; staggering needs to know)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 4) void probe(unsigned *out, unsigned long long *t)
{
    extern __shared__ unsigned char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; t[blockIdx.x] = __builtin_amdgcn_s_memrealtime(); smem[0] = 1; }
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(127);     // keep every workgroup resident while the others arrive
}
int main()
{
    const int n = 1025;
    unsigned *out; unsigned long long *t;
    hipMalloc(&out, n * 8); hipMalloc(&t, n * 8);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 36240);
    hipLaunchKernelGGL(probe, n, 256, 36240, 0, out, t);
    std::vector<unsigned> h(2 * n); std::vector<unsigned long long> ht(n);
    hipMemcpy(h.data(), out, n * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), t, n * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < n; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned key = (xcc << 16) | (hw & 0xff00);          // xcc, se/sh/cu bits of HW_ID
        cu[key].push_back(b);
    }
    printf("%zu distinct (xcc, se, sh, cu); first 12:\n", cu.size());
    int k = 0;
    for (auto &e : cu) { if (k++ < 12) { printf("  key %05x:", e.first); for (int b : e.second) printf(" %d", b); printf("\n"); } }
    std::map<int, int> hist; std::map<int,int> d256, mod4;
    for (auto &e : cu) { hist[(int)e.second.size()]++; std::map<int,int> c1, c2; for (int b : e.second) { c1[(b >> 8) & 3]++; c2[b & 3]++; }
        for (auto &x : c1) if (x.second > 1) d256[x.second]++; for (auto &x : c2) if (x.second > 1) mod4[x.second]++; }
    for (auto &x : hist) printf("CUs with %d workgroups: %d\n", x.first, x.second);
    for (auto &x : d256) printf("class (b>>8)&3: %d CUs-classes with %d workgroups of one class\n", x.second, x.first);
    for (auto &x : mod4) printf("class b&3: %d CU-classes with %d workgroups of one class\n", x.second, x.first);
    return 0;
}
