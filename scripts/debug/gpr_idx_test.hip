// Does the VGPR index mode (s_set_gpr_idx_on) work on gfx950, and what does an indexed v_add_f32 cost?  (round 4 experiment)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void pick(const int *idx, float *out, int mode)
{
    f4 pv = {10.0f + threadIdx.x, 20.0f + threadIdx.x, 30.0f + threadIdx.x, 40.0f + threadIdx.x};
    int i0 = __builtin_amdgcn_readfirstlane(idx[0]), i1 = __builtin_amdgcn_readfirstlane(idx[1]);
    float a = 1000.0f, b = 2000.0f, o0, o1;
    if (mode == 0)
        asm volatile("s_set_gpr_idx_on %5, gpr_idx(SRC0)\n\tv_add_f32 %0, v120, %3\n\ts_set_gpr_idx_idx %6\n\tv_add_f32 %1, v120, %4\n\ts_set_gpr_idx_off"
                     : "=&v"(o0), "=&v"(o1) : "{v[120:123]}"(pv), "v"(a), "v"(b), "s"(i0), "s"(i1) : "m0");
    else
        asm volatile("s_set_gpr_idx_on %5, gpr_idx(SRC0)\n\ts_nop 4\n\tv_add_f32 %0, v120, %3\n\ts_nop 4\n\ts_set_gpr_idx_idx %6\n\ts_nop 4\n\tv_add_f32 %1, v120, %4\n\ts_nop 4\n\ts_set_gpr_idx_off\n\ts_nop 4"
                     : "=&v"(o0), "=&v"(o1) : "{v[120:123]}"(pv), "v"(a), "v"(b), "s"(i0), "s"(i1) : "m0");
    out[threadIdx.x * 2] = o0; out[threadIdx.x * 2 + 1] = o1;
}
__global__ __launch_bounds__(256, 4) void rate(float *out, int n, int i0, int variant)
{
    f4 pv = {1.0f, -1.0f, 0.1f, 0.0f};
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    for (int k = 0; k < n; ++k) {
        if (variant == 0)
            asm volatile("s_set_gpr_idx_on %5, gpr_idx(SRC0)\n\tv_add_f32 %0, v120, %0\n\ts_set_gpr_idx_idx %5\n\tv_add_f32 %1, v120, %1\n\ts_set_gpr_idx_idx %5\n\tv_add_f32 %2, v120, %2\n\ts_set_gpr_idx_idx %5\n\tv_add_f32 %3, v120, %3\n\ts_set_gpr_idx_off"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "{v[120:123]}"(pv), "s"(i0) : "m0");
        else if (variant == 1)
            asm volatile("v_add_f32 %0, v120, %0\n\tv_add_f32 %1, v121, %1\n\tv_add_f32 %2, v122, %2\n\tv_add_f32 %3, v123, %3"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "{v[120:123]}"(pv), "s"(i0));
        else
            asm volatile("v_cmp_eq_u32 vcc, %5, %0\n\tv_cndmask_b32 v119, v120, v121, vcc\n\tv_add_f32 %0, v119, %0\n\tv_cmp_eq_u32 vcc, %5, %1\n\tv_cndmask_b32 v119, v120, v121, vcc\n\tv_add_f32 %1, v119, %1\n\t"
                         "v_cmp_eq_u32 vcc, %5, %2\n\tv_cndmask_b32 v119, v120, v121, vcc\n\tv_add_f32 %2, v119, %2\n\tv_cmp_eq_u32 vcc, %5, %3\n\tv_cndmask_b32 v119, v120, v121, vcc\n\tv_add_f32 %3, v119, %3"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "{v[120:123]}"(pv), "s"(i0) : "vcc", "v119");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
int main()
{
    int *idx; float *out;
    hipMalloc(&idx, 8); hipMalloc(&out, 1 << 24);
    int bad = 0;
    for (int mode = 0; mode < 2; ++mode)
        for (int i0 = 0; i0 < 4; ++i0) for (int i1 = 0; i1 < 4; ++i1) {
            int h[2] = {i0, i1}; hipMemcpy(idx, h, 8, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(pick, 1, 64, 0, 0, idx, out, mode);
            std::vector<float> r(128); hipMemcpy(r.data(), out, 512, hipMemcpyDeviceToHost);
            for (int t = 0; t < 64; ++t) {
                const float e0 = 10.0f * (i0 + 1) + t + 1000.0f, e1 = 10.0f * (i1 + 1) + t + 2000.0f;
                if (r[2 * t] != e0 || r[2 * t + 1] != e1) { if (bad < 6) printf("mode %d idx %d %d lane %d: got %g %g want %g %g\n", mode, i0, i1, t, r[2 * t], r[2 * t + 1], e0, e1); ++bad; }
            }
        }
    printf("index mode: %d wrong values\n", bad);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[3] = {"4 indexed v_add_f32 (on, 3 idx, off)", "4 plain v_add_f32", "4 x (v_cmp, v_cndmask, v_add_f32)"};
    for (int v = 0; v < 3; ++v) {
        const int n = 20000, blocks = 256 * 4;
        hipLaunchKernelGGL(rate, blocks, 256, 0, 0, out, 100, 1, v);
        hipEventRecord(e0); hipLaunchKernelGGL(rate, blocks, 256, 0, 0, out, n, 1, v); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %.3f ms  = %.2f ns per loop trip per wave set (4 waves/SIMD)\n", names[v], ms, ms * 1e6 / n);
    }
    return bad != 0;
}
