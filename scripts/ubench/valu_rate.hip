// valu_rate.hip -- issue rate of the VALU / LDS instructions the clip and quality-stats kernels are built from (gfx950).
// Not part of the product: a measurement aid (`hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate`).
// Every test runs REPS x 64 copies of one instruction over eight independent register chains in every wave and reports
// shader cycles (s_memtime) per wave-instruction, for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "clip_loop_variants.inc"
#define REPS 256
#define X8(s) s s s s s s s s
#define X64(s) X8(X8(s))

#define BODY8(INS)                                    \
    INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)


#define T2F(ID, INS) if (OP == ID) asm volatile(X8(INS " %0, %0, %8\n " INS " %1, %1, %8\n " INS " %2, %2, %8\n " INS " %3, %3, %8\n " INS " %4, %4, %8\n " INS " %5, %5, %8\n " INS " %6, %6, %8\n " INS " %7, %7, %8\n") \
                                  : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0f))
#define T2U(ID, INS) if (OP == ID) asm volatile(X8(INS " %0, %0, %8\n " INS " %1, %1, %8\n " INS " %2, %2, %8\n " INS " %3, %3, %8\n " INS " %4, %4, %8\n " INS " %5, %5, %8\n " INS " %6, %6, %8\n " INS " %7, %7, %8\n") \
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(3u))
#define T3F(ID, INS) if (OP == ID) asm volatile(X8(INS " %0, %0, %8, %1\n " INS " %1, %1, %8, %2\n " INS " %2, %2, %8, %3\n " INS " %3, %3, %8, %4\n " INS " %4, %4, %8, %5\n " INS " %5, %5, %8, %6\n " INS " %6, %6, %8, %7\n " INS " %7, %7, %8, %0\n") \
                                  : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0f))
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, unsigned *sink, int reps)
{
    __shared__ unsigned lds[8192];
    for (unsigned i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    float f0 = threadIdx.x, f1 = 1.5f, f2 = 2.5f, f3 = 3.5f, f4 = 4.5f, f5 = 5.5f, f6 = 6.5f, f7 = 7.5f;
    unsigned u0 = threadIdx.x, u1 = 11, u2 = 12, u3 = 13, u4 = 14, u5 = 15, u6 = 16, u7 = 17;
    double d0 = 1.0, d1 = 2.0, d2 = 3.0, d3 = 4.0;
    unsigned a0 = (threadIdx.x * 4u) & 32764u, a1 = ((threadIdx.x * 32u) & 8191u) * 4u;   // a1: every lane of a 32-lane group on bank 0, different words
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        if (OP == 0) asm volatile(X8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                                  : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0f));
        if (OP == 1) asm volatile(X8("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n")
                                  : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0f));
        if (OP == 2) asm volatile(X8("v_max3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_max3_f32 %3, %3, %8, %4\n v_max3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %8, %0\n")
                                  : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0f));
        if (OP == 3) asm volatile(X8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(3u) : "vcc");
        if (OP == 4) asm volatile(X8("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n")
                                  : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0f) : "vcc");
        if (OP == 5) asm volatile(X8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(3u));
        if (OP == 6) asm volatile(X8("v_addc_co_u32 %0, vcc, %0, %8, vcc\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_addc_co_u32 %4, vcc, %4, %8, vcc\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_addc_co_u32 %6, vcc, %6, %8, vcc\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n")
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(3u) : "vcc");
        if (OP == 7) asm volatile(X8("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(0xFFFFFFF3u));
        if (OP == 8) asm volatile(X8("v_perm_b32 %0, %0, %8, %1\n v_perm_b32 %1, %1, %8, %2\n v_perm_b32 %2, %2, %8, %3\n v_perm_b32 %3, %3, %8, %4\n v_perm_b32 %4, %4, %8, %5\n v_perm_b32 %5, %5, %8, %6\n v_perm_b32 %6, %6, %8, %7\n v_perm_b32 %7, %7, %8, %0\n")
                                  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(0x07060100u));
        if (OP == 9) asm volatile(X8("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                                  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0));
        if (OP == 10) asm volatile(X8("v_add3_u32 %0, %0, %8, %1\n v_add3_u32 %1, %1, %8, %2\n v_add3_u32 %2, %2, %8, %3\n v_add3_u32 %3, %3, %8, %4\n v_add3_u32 %4, %4, %8, %5\n v_add3_u32 %5, %5, %8, %6\n v_add3_u32 %6, %6, %8, %7\n v_add3_u32 %7, %7, %8, %0\n")
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(3u));
        if (OP == 11) asm volatile(X8("v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8\n v_bfe_u32 %4, %4, 8, 8\n v_bfe_u32 %5, %5, 8, 8\n v_bfe_u32 %6, %6, 8, 8\n v_bfe_u32 %7, %7, 8, 8\n")
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
        if (OP == 12) asm volatile(X8("v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %0\n")
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(3u));
        if (OP == 13) asm volatile(X8("v_add_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n")
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(0x00030201u));
        if (OP == 14) asm volatile(X8("v_cmp_eq_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_eq_u32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_eq_u32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_eq_u32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n")
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(3u) : "vcc");
        if (OP == 15) asm volatile(X8("ds_add_u32 %0, %2\n ds_add_u32 %0, %2 offset:4096\n ds_add_u32 %0, %2 offset:8192\n ds_add_u32 %0, %2 offset:12288\n ds_add_u32 %0, %2 offset:16384\n ds_add_u32 %0, %2 offset:20480\n ds_add_u32 %0, %2 offset:24576\n ds_add_u32 %0, %2 offset:28672\n") "s_waitcnt lgkmcnt(0)\n"
                                   : : "v"(a0 & 4095u), "v"(a1), "v"(1u) : "memory");     // conflict-free: lane i -> word i
        if (OP == 16) asm volatile(X8("ds_add_u32 %1, %2\n ds_add_u32 %1, %2 offset:4\n ds_add_u32 %1, %2 offset:8\n ds_add_u32 %1, %2 offset:12\n ds_add_u32 %1, %2 offset:16\n ds_add_u32 %1, %2 offset:20\n ds_add_u32 %1, %2 offset:24\n ds_add_u32 %1, %2 offset:28\n") "s_waitcnt lgkmcnt(0)\n"
                                   : : "v"(a0), "v"(a1 & 32764u), "v"(1u) : "memory");    // same bank, different words: the worst bank conflict
        if (OP == 17) asm volatile(X8("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n") "s_waitcnt lgkmcnt(0)\n"
                                   : "=v"(*(uint4 *)&d0), "=v"(*(uint4 *)&d2), "=v"(*(uint4 *)&f0), "=v"(*(uint4 *)&f4) : "v"((threadIdx.x & 63u) * 16u) : "memory");   // aligned 16 B per lane
        if (OP == 18) asm volatile(X8("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n") "s_waitcnt lgkmcnt(0)\n"
                                   : "=v"(*(uint4 *)&d0), "=v"(*(uint4 *)&d2), "=v"(*(uint4 *)&f0), "=v"(*(uint4 *)&f4) : "v"((threadIdx.x & 63u) * 16u + 5u) : "memory");   // the same, 5 bytes off alignment
        // ---- lane masks: where the mask of a v_cndmask comes from decides what it costs ----
        if (OP == 20) asm volatile(X8("v_cmp_gt_f32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cmp_gt_f32_e64 s[22:23], %2, %8\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n v_cmp_gt_f32_e64 s[24:25], %4, %8\n v_cndmask_b32_e64 %5, %5, %8, s[24:25]\n v_cmp_gt_f32_e64 s[26:27], %6, %8\n v_cndmask_b32_e64 %7, %7, %8, s[26:27]\n")
                                   : "+v"(f0), "+v"(u1), "+v"(f2), "+v"(u3), "+v"(f4), "+v"(u5), "+v"(f6), "+v"(u7) : "v"(1.0f) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        if (OP == 21) asm volatile(X8("v_cmp_gt_f32_e64 s[20:21], %0, %8\n v_cmp_gt_f32_e64 s[22:23], %2, %8\n v_cmp_gt_f32_e64 s[24:25], %4, %8\n v_cmp_gt_f32_e64 s[26:27], %6, %8\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n v_cndmask_b32_e64 %5, %5, %8, s[24:25]\n v_cndmask_b32_e64 %7, %7, %8, s[26:27]\n")
                                   : "+v"(f0), "+v"(u1), "+v"(f2), "+v"(u3), "+v"(f4), "+v"(u5), "+v"(f6), "+v"(u7) : "v"(1.0f) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        if (OP == 22) asm volatile(X8("v_cmp_gt_f32 vcc, %0, %8\n s_and_b64 vcc, vcc, s[40:41]\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_gt_f32 vcc, %2, %8\n s_and_b64 vcc, vcc, s[40:41]\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n")
                                   : "+v"(f0), "+v"(u1), "+v"(f2), "+v"(u3), "+v"(f4), "+v"(u5) : "v"(f6), "v"(u7), "v"(1.0f) : "vcc", "s40", "s41");
        if (OP == 23) asm volatile(X8("v_cndmask_b32_e64 %0, %0, %8, s[40:41]\n v_cndmask_b32_e64 %1, %1, %8, s[40:41]\n v_cndmask_b32_e64 %2, %2, %8, s[42:43]\n v_cndmask_b32_e64 %3, %3, %8, s[42:43]\n v_cndmask_b32_e64 %4, %4, %8, s[40:41]\n v_cndmask_b32_e64 %5, %5, %8, s[42:43]\n v_cndmask_b32_e64 %6, %6, %8, s[40:41]\n v_cndmask_b32_e64 %7, %7, %8, s[42:43]\n")
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(3u) : "s40", "s41", "s42", "s43");
        if (OP == 24) asm volatile(X8("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %7, %7, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n")
                                   : "+v"(f0), "+v"(u1), "+v"(f2), "+v"(u3), "+v"(f4), "+v"(u5), "+v"(f6), "+v"(u7) : "v"(1.0f) : "vcc");
        if (OP == 25) asm volatile(X8("v_cmp_gt_f32 vcc, %0, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %6, %6, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_add_f32 %2, %2, %8\n v_add_f32 %4, %4, %8\n v_cndmask_b32 %3, %3, %8, vcc\n")
                                   : "+v"(f0), "+v"(u1), "+v"(f2), "+v"(u3), "+v"(f4), "+v"(u5), "+v"(f6), "+v"(u7) : "v"(1.0f) : "vcc");
        if (OP == 26) asm volatile(X8("v_cmp_eq_f32 vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_cmp_eq_f32 vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_cmp_eq_f32 vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_cmp_eq_f32 vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n")
                                   : "+v"(f0), "+v"(u1), "+v"(f2), "+v"(u3), "+v"(f4), "+v"(u5), "+v"(f6), "+v"(u7) : "v"(1.0f) : "vcc");
        if (OP == 27) asm volatile(X8("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4\n ds_read_b32 %2, %4 offset:8\n ds_read_b32 %3, %4 offset:12\n ds_read_b32 %0, %4 offset:16\n ds_read_b32 %1, %4 offset:1028\n ds_read_b32 %2, %4 offset:1032\n ds_read_b32 %3, %4 offset:1036\n") "s_waitcnt lgkmcnt(0)\n"
                                   : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"((threadIdx.x & 63u) * 16u + 4u) : "memory");   // dword reads, 16 B stride between lanes
        if (OP == 28) asm volatile(X8("ds_read2_b32 %0, %4 offset1:1\n ds_read2_b32 %1, %4 offset0:2 offset1:3\n ds_read2_b32 %2, %4 offset0:4 offset1:5\n ds_read2_b32 %3, %4 offset0:6 offset1:7\n ds_read2_b32 %0, %4 offset0:8 offset1:9\n ds_read2_b32 %1, %4 offset0:10 offset1:11\n ds_read2_b32 %2, %4 offset0:12 offset1:13\n ds_read2_b32 %3, %4 offset0:14 offset1:15\n") "s_waitcnt lgkmcnt(0)\n"
                                   : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"((threadIdx.x & 63u) * 16u + 4u) : "memory");
        if (OP == 29) asm volatile(X8("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:1024\n ds_read_b64 %2, %4 offset:2048\n ds_read_b64 %3, %4 offset:3072\n ds_read_b64 %0, %4 offset:4096\n ds_read_b64 %1, %4 offset:5120\n ds_read_b64 %2, %4 offset:6144\n ds_read_b64 %3, %4 offset:7168\n") "s_waitcnt lgkmcnt(0)\n"
                                   : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"((threadIdx.x & 63u) * 16u + 4u) : "memory");   // 8-byte reads at 4-byte alignment
        if (OP == 30) asm volatile(X8("v_alignbyte_b32 %0, %1, %0, %8\n v_alignbyte_b32 %1, %2, %1, %8\n v_alignbyte_b32 %2, %3, %2, %8\n v_alignbyte_b32 %3, %4, %3, %8\n v_alignbyte_b32 %4, %5, %4, %8\n v_alignbyte_b32 %5, %6, %5, %8\n v_alignbyte_b32 %6, %7, %6, %8\n v_alignbyte_b32 %7, %0, %7, %8\n")
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(threadIdx.x & 3u));
        // ---- dependent issue: how far apart must two instructions of ONE wave be when the second reads the first's result ----
        if (OP == 31) asm volatile(X64("v_max3_f32 %0, %0, %1, %2\n") : "+v"(f0) : "v"(f1), "v"(f2));                                   // one chain
        if (OP == 32) asm volatile(X8("v_max3_f32 %0, %0, %2, %3\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %0, %0, %2, %3\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %0, %0, %2, %3\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %0, %0, %2, %3\n v_max3_f32 %1, %1, %2, %3\n")
                                   : "+v"(f0), "+v"(f1) : "v"(f2), "v"(f3));                                                                // two chains
        if (OP == 33) asm volatile(X8("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f4), "v"(f5));                                            // four chains
        if (OP == 34) asm volatile(X64("v_add_f32 %0, %0, %1\n") : "+v"(f0) : "v"(f1));                                                    // one chain of adds
        // the score row of the clip kernel's first pass, one cell after the other: cmp -> cndmask -> add -> max3 -> add, each feeding the next
        if (OP == 35) asm volatile(X8("v_cmp_eq_u32 vcc, %4, %5\n v_cndmask_b32 %2, %6, %7, vcc\n v_add_f32 %2, %2, %1\n v_max3_f32 %1, %2, %0, %3\n v_add_f32 %0, %8, %1\n v_cmp_eq_u32 vcc, %4, %5\n v_cndmask_b32 %2, %6, %7, vcc\n v_add_f32 %2, %2, %1\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2) : "v"(f3), "v"(u0), "v"(u1), "v"(f4), "v"(f5), "v"(f6) : "vcc");
        // the same with two independent cells interleaved
        if (OP == 36) asm volatile(X8("v_cmp_eq_u32 vcc, %7, %8\n v_cmp_eq_u32 s[20:21], %7, %9\n v_cndmask_b32 %2, %10, %11, vcc\n v_cndmask_b32 %5, %10, %11, s[20:21]\n v_add_f32 %2, %2, %1\n v_add_f32 %5, %5, %4\n v_max3_f32 %1, %2, %0, %6\n v_max3_f32 %4, %5, %3, %6\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(f6), "v"(u0), "v"(u1), "v"(u2), "v"(f7), "v"(f7) : "vcc", "s20", "s21");
        // ---- round 4: which instructions issue at the v_add_f32 rate, which at the v_max_f32 rate; do the two overlap ----
        T2F(40, "v_mul_f32"); T2F(41, "v_sub_f32"); T2F(42, "v_min_f32"); T3F(43, "v_fma_f32"); T3F(44, "v_med3_f32"); T2F(45, "v_fmac_f32");
        T2U(46, "v_or_b32"); T2U(47, "v_xor_b32"); T2U(48, "v_lshlrev_b32"); T2U(49, "v_max_u32"); T2U(50, "v_max_i32"); T2U(51, "v_sub_u32"); T2U(52, "v_min_u32");
        if (OP == 53) asm volatile(X8("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
        // one slow and one fast instruction alternating (two pipes, or one?)
        if (OP == 54) asm volatile(X8("v_max_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0f));
        if (OP == 55) asm volatile(X8("v_max3_f32 %0, %0, %8, %1\n v_add_f32 %1, %1, %8\n v_max3_f32 %2, %2, %8, %3\n v_add_f32 %3, %3, %8\n v_max3_f32 %4, %4, %8, %5\n v_add_f32 %5, %5, %8\n v_max3_f32 %6, %6, %8, %7\n v_add_f32 %7, %7, %8\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0f));
        // three fast to one slow
        if (OP == 56) asm volatile(X8("v_max3_f32 %0, %0, %8, %1\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_max3_f32 %4, %4, %8, %5\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0f));
        // v_max_f32 whose operands change (is the rate data dependent?): chains fed by adds
        if (OP == 57) asm volatile(X8("v_max_f32 %0, %1, %2\n v_max_f32 %1, %2, %3\n v_max_f32 %2, %3, %4\n v_max_f32 %3, %4, %5\n v_max_f32 %4, %5, %6\n v_max_f32 %5, %6, %7\n v_max_f32 %6, %7, %0\n v_max_f32 %7, %0, %1\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0f));
        // v_add_f32 with a literal (8 bytes) and with an SGPR operand
        if (OP == 58) asm volatile(X8("v_add_f32 %0, 0xc0a00000, %0\n v_add_f32 %1, 0xc0a00000, %1\n v_add_f32 %2, 0xc0a00000, %2\n v_add_f32 %3, 0xc0a00000, %3\n v_add_f32 %4, 0xc0a00000, %4\n v_add_f32 %5, 0xc0a00000, %5\n v_add_f32 %6, 0xc0a00000, %6\n v_add_f32 %7, 0xc0a00000, %7\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
        // the pass-1 cell as the shipped kernel has it: e64 compare into an SGPR pair two cells ahead, select, add, max3, add with a literal
        if (OP == 59) asm volatile(X8("v_cmp_eq_u32_e64 s[20:21], %7, %8\n v_cndmask_b32_e64 %2, %10, %11, s[22:23]\n v_add_f32 %2, %2, %1\n v_max3_f32 %1, %2, %0, %3\n v_add_f32 %0, 0xc0a00000, %1\n v_cmp_eq_u32_e64 s[22:23], %7, %9\n v_cndmask_b32_e64 %5, %10, %11, s[20:21]\n v_add_f32 %5, %5, %4\n v_max3_f32 %4, %5, %0, %3\n v_add_f32 %0, 0xc0a00000, %4\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(f6), "v"(u0), "v"(u1), "v"(u2), "v"(f7), "v"(f7) : "s20", "s21", "s22", "s23");
        // the asm row of experiment v (24 bytes per cell): vcc masks, -5.0 from an SGPR, the compiler's s_nop 0 between the statements
        if (OP == 60) asm volatile(X8("v_cndmask_b32 %2, %10, %11, vcc\n v_add_f32 %2, %2, %1\n v_cmp_eq_u32 vcc, %7, %8\n v_max3_f32 %1, %5, %0, %3\n v_add_f32 %0, s30, %1\n s_nop 0\n v_cndmask_b32 %5, %10, %11, vcc\n v_add_f32 %5, %5, %4\n v_cmp_eq_u32 vcc, %7, %9\n v_max3_f32 %4, %2, %0, %3\n v_add_f32 %0, s30, %4\n s_nop 0\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(f6), "v"(u0), "v"(u1), "v"(u2), "v"(f7), "v"(f7) : "vcc", "s30");
        // the same without the s_nops and with -5.0 in a VGPR
        if (OP == 61) asm volatile(X8("v_cndmask_b32 %2, %10, %11, vcc\n v_add_f32 %2, %2, %1\n v_cmp_eq_u32 vcc, %7, %8\n v_max3_f32 %1, %5, %0, %3\n v_add_f32 %0, %6, %1\n v_cndmask_b32 %5, %10, %11, vcc\n v_add_f32 %5, %5, %4\n v_cmp_eq_u32 vcc, %7, %9\n v_max3_f32 %4, %2, %0, %3\n v_add_f32 %0, %6, %4\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(f6), "v"(u0), "v"(u1), "v"(u2), "v"(f7), "v"(f7) : "vcc");
        // ... and -5.0 as an SGPR operand, no s_nops
        if (OP == 62) asm volatile(X8("v_cndmask_b32 %2, %10, %11, vcc\n v_add_f32 %2, %2, %1\n v_cmp_eq_u32 vcc, %7, %8\n v_max3_f32 %1, %5, %0, %3\n v_add_f32 %0, s30, %1\n v_cndmask_b32 %5, %10, %11, vcc\n v_add_f32 %5, %5, %4\n v_cmp_eq_u32 vcc, %7, %9\n v_max3_f32 %4, %2, %0, %3\n v_add_f32 %0, s30, %4\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(f6), "v"(u0), "v"(u1), "v"(u2), "v"(f7), "v"(f7) : "vcc", "s30");
        // the compare against an SGPR (the adapter letter) instead of a VGPR
        if (OP == 63) asm volatile(X8("v_cndmask_b32 %2, %10, %11, vcc\n v_add_f32 %2, %2, %1\n v_cmp_eq_u32 vcc, s31, %8\n v_max3_f32 %1, %5, %0, %3\n v_add_f32 %0, %6, %1\n v_cndmask_b32 %5, %10, %11, vcc\n v_add_f32 %5, %5, %4\n v_cmp_eq_u32 vcc, s31, %9\n v_max3_f32 %4, %2, %0, %3\n v_add_f32 %0, %6, %4\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(f6), "v"(u0), "v"(u1), "v"(u2), "v"(f7), "v"(f7) : "vcc", "s31");
        // the pass-2 cell: score part + the two direction compares, selects, the two summary adds
        if (OP == 64) asm volatile(X8("v_cmp_eq_u32 vcc, %7, %8\n v_max3_f32 %1, %5, %0, %3\n v_add_f32 %0, %6, %1\n v_cndmask_b32 %2, %10, %11, vcc\n v_addc_co_u32 %9, vcc, %9, %8, vcc\n v_add_f32 %2, %2, %1\n v_cmp_eq_f32 vcc, %1, %0\n v_add_f32 %4, %6, %4\n v_add_f32 %5, %6, %5\n v_cndmask_b32 %8, %8, %9, vcc\n v_cmp_eq_f32 vcc, %1, %5\n v_add_f32 %4, %6, %4\n v_add_f32 %5, %6, %5\n v_cndmask_b32 %8, %8, %9, vcc\n v_add_u32 %8, 32, %8\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(u0), "+v"(u1), "+v"(u2) : "v"(f7), "v"(f7) : "vcc");
        // VGPR banks: the same instruction with its sources in one bank (register number mod 4) and in different banks
        if (OP == 65) asm volatile(X8("v_max3_f32 v40, v44, v48, v52\n v_max3_f32 v56, v60, v64, v68\n v_max3_f32 v44, v48, v52, v56\n v_max3_f32 v60, v64, v68, v40\n v_max3_f32 v48, v52, v56, v60\n v_max3_f32 v64, v68, v40, v44\n v_max3_f32 v52, v56, v60, v64\n v_max3_f32 v68, v40, v44, v48\n")
                                   : : : "v40", "v44", "v48", "v52", "v56", "v60", "v64", "v68");
        if (OP == 66) asm volatile(X8("v_max3_f32 v40, v45, v50, v55\n v_max3_f32 v56, v61, v66, v71\n v_max3_f32 v45, v50, v55, v56\n v_max3_f32 v61, v66, v71, v40\n v_max3_f32 v50, v55, v56, v61\n v_max3_f32 v66, v71, v40, v45\n v_max3_f32 v55, v56, v61, v66\n v_max3_f32 v71, v40, v45, v50\n")
                                   : : : "v40", "v45", "v50", "v55", "v56", "v61", "v66", "v71");
        if (OP == 67) asm volatile(X8("v_add_f32 v40, v44, v48\n v_add_f32 v52, v56, v60\n v_add_f32 v44, v48, v52\n v_add_f32 v56, v60, v40\n v_add_f32 v48, v52, v56\n v_add_f32 v60, v40, v44\n v_add_f32 v64, v68, v40\n v_add_f32 v68, v64, v44\n")
                                   : : : "v40", "v44", "v48", "v52", "v56", "v60", "v64", "v68");
        if (OP == 68) asm volatile(X8("v_add_f32 v40, v45, v50\n v_add_f32 v52, v57, v62\n v_add_f32 v45, v50, v52\n v_add_f32 v57, v62, v40\n v_add_f32 v50, v52, v57\n v_add_f32 v62, v40, v45\n v_add_f32 v64, v69, v42\n v_add_f32 v69, v64, v47\n")
                                   : : : "v40", "v45", "v50", "v52", "v57", "v62", "v64", "v69", "v42", "v47");
        // the pass-1 cell (test 35 order) with every register of a cell in one bank / spread over the banks, high register numbers
        if (OP == 69) asm volatile(X8("v_cmp_eq_u32 vcc, v100, v104\n v_cndmask_b32 v108, v112, v116, vcc\n v_add_f32 v108, v108, v120\n v_max3_f32 v120, v108, v124, v96\n v_add_f32 v124, v92, v120\n")
                                   : : : "vcc", "v92", "v96", "v100", "v104", "v108", "v112", "v116", "v120", "v124");
        if (OP == 70) asm volatile(X8("v_cmp_eq_u32 vcc, v100, v105\n v_cndmask_b32 v110, v113, v118, vcc\n v_add_f32 v110, v110, v121\n v_max3_f32 v121, v110, v127, v96\n v_add_f32 v127, v93, v121\n")
                                   : : : "vcc", "v93", "v96", "v100", "v105", "v110", "v113", "v118", "v121", "v127");
        // ---- round 4: the pass-1 row loop of the shipped clip kernel, registers as allocated and renamed by bank (clip_loop_variants.inc) ----
#define CLIP_SGPRS "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s28", "s29", "s33", "s66", "s67", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s88"
        if (OP == 71) asm volatile("v_mov_b32 " CLIP_ADDR_SHIPPED ", 0\n" X8(CLIP_LOOP_SHIPPED) : : : "vcc", "memory", CLIP_SGPRS, CLIP_CLOB_SHIPPED);
        if (OP == 72) asm volatile("v_mov_b32 " CLIP_ADDR_SRC ", 0\n" X8(CLIP_LOOP_SRC) : : : "vcc", "memory", CLIP_SGPRS, CLIP_CLOB_SRC);
        if (OP == 73) asm volatile("v_mov_b32 " CLIP_ADDR_SRCDST ", 0\n" X8(CLIP_LOOP_SRCDST) : : : "vcc", "memory", CLIP_SGPRS, CLIP_CLOB_SRCDST);
        if (OP == 74) asm volatile("v_mov_b32 " CLIP_ADDR_ALL ", 0\n" X8(CLIP_LOOP_ALL) : : : "vcc", "memory", CLIP_SGPRS, CLIP_CLOB_ALL);
        if (OP == 75) asm volatile("v_mov_b32 " CLIP_ADDR_BANK0 ", 0\n" X8(CLIP_LOOP_BANK0) : : : "vcc", "memory", CLIP_SGPRS, CLIP_CLOB_BANK0);
        if (OP == 76) asm volatile("v_mov_b32 v123, 0\n v_mov_b32 v17, 0\n" X8(CLIP_LOOP2) : : : "vcc", "memory", CLIP_CLOB2);
        // ---- round 6: the mixed-precision fma that takes a pair value out of a packed half (pass 1 of the clip DP with its LDS pair table) ----
        if (OP == 77) asm volatile(X8("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %8, %9, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %8, %9, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(0x3C00BC00u), "v"(1.0f));
        if (OP == 78) asm volatile(X8("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %8\n v_cvt_f32_f16 %2, %8\n v_cvt_f32_f16 %3, %8\n v_cvt_f32_f16 %4, %8\n v_cvt_f32_f16 %5, %8\n v_cvt_f32_f16 %6, %8\n v_cvt_f32_f16 %7, %8\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(0x3C00BC00u));
        // the table cell: fma_mix (diagonal candidate) ; max3 ; add -- two cells, as the sweep chains them
        if (OP == 79) asm volatile(X8("v_fma_mix_f32 %2, %6, %7, %1 op_sel_hi:[1,0,0]\n v_max3_f32 %1, %2, %0, %3\n v_add_f32 %0, %8, %1\n v_fma_mix_f32 %5, %6, %7, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_max3_f32 %4, %5, %0, %3\n v_add_f32 %0, %8, %4\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(0x3C00BC00u), "v"(f6), "v"(f7));
        // the same cell with the pair value as an f32 in a register: add ; max3 ; add
        if (OP == 80) asm volatile(X8("v_add_f32 %2, %6, %1\n v_max3_f32 %1, %2, %0, %3\n v_add_f32 %0, %8, %1\n v_add_f32 %5, %7, %4\n v_max3_f32 %4, %5, %0, %3\n v_add_f32 %0, %8, %4\n")
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : "v"(f6), "v"(f6), "v"(f7));
        // ... and with the f32 table's fetch beside it: 6 cells per ds_read_b128 (13 cells : 4 reads would be 3.25), five distinct rows per wave
        if (OP == 81) asm volatile(X8("ds_read_b128 %[q0], %[ad]\n v_add_f32 %[c], %[p], %[b]\n v_max3_f32 %[b], %[c], %[a], %[d]\n v_add_f32 %[a], %[m5], %[b]\n v_add_f32 %[f], %[p], %[e]\n v_max3_f32 %[e], %[f], %[a], %[d]\n v_add_f32 %[a], %[m5], %[e]\n v_add_f32 %[c], %[p], %[b]\n v_max3_f32 %[b], %[c], %[a], %[d]\n v_add_f32 %[a], %[m5], %[b]\n ds_read_b128 %[q1], %[ad] offset:16\n v_add_f32 %[f], %[p], %[e]\n v_max3_f32 %[e], %[f], %[a], %[d]\n v_add_f32 %[a], %[m5], %[e]\n v_add_f32 %[c], %[p], %[b]\n v_max3_f32 %[b], %[c], %[a], %[d]\n v_add_f32 %[a], %[m5], %[b]\n v_add_f32 %[f], %[p], %[e]\n v_max3_f32 %[e], %[f], %[a], %[d]\n v_add_f32 %[a], %[m5], %[e]\n s_waitcnt lgkmcnt(0)\n")
                                   : [a] "+v"(f0), [b] "+v"(f1), [c] "+v"(f2), [d] "+v"(f3), [e] "+v"(f4), [f] "+v"(f5), [q0] "=&v"(*(uint4 *)&d0), [q1] "=&v"(*(uint4 *)&d2)
                                   : [p] "v"(f6), [m5] "v"(f7), [ad] "v"(((threadIdx.x * 7u) % 5u) * 64u) : "memory");
        // ... the half table's fetch: 4 cells per ds_read_b128 (13 cells : 2 reads would be 6.5)
        if (OP == 82) asm volatile(X8("ds_read_b128 %[q0], %[ad]\n v_fma_mix_f32 %[c], %[h], %[p], %[b] op_sel_hi:[1,0,0]\n v_max3_f32 %[b], %[c], %[a], %[d]\n v_add_f32 %[a], %[m5], %[b]\n v_fma_mix_f32 %[f], %[h], %[p], %[e] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_max3_f32 %[e], %[f], %[a], %[d]\n v_add_f32 %[a], %[m5], %[e]\n v_fma_mix_f32 %[c], %[h], %[p], %[b] op_sel_hi:[1,0,0]\n v_max3_f32 %[b], %[c], %[a], %[d]\n v_add_f32 %[a], %[m5], %[b]\n v_fma_mix_f32 %[f], %[h], %[p], %[e] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_max3_f32 %[e], %[f], %[a], %[d]\n v_add_f32 %[a], %[m5], %[e]\n s_waitcnt lgkmcnt(0)\n")
                                   : [a] "+v"(f0), [b] "+v"(f1), [c] "+v"(f2), [d] "+v"(f3), [e] "+v"(f4), [f] "+v"(f5), [q0] "=&v"(*(uint4 *)&d0)
                                   : [h] "v"(0x3C00BC00u), [p] "v"(f6), [m5] "v"(f7), [ad] "v"(((threadIdx.x * 7u) % 5u) * 32u) : "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63u) == 0u) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if (blockIdx.x == 7u && threadIdx.x == 0u) { out[256 * 16 - 2] = t1 - t0; out[256 * 16 - 1] = r1 - r0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) + u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7 + (unsigned)(d0 + d1 + d2 + d3) + lds[threadIdx.x];
}

template <int OP>
static void run(const char *name, int per_rep)
{
    unsigned long long *out;
    unsigned *sink;
    hipMalloc(&out, 256 * 16 * sizeof *out);
    hipMalloc(&sink, 256 * 1024 * sizeof *sink);
    printf("%-44s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {                  // waves per SIMD: workgroup of 256 * wps threads, one workgroup per CU
        const int threads = 256 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, sink, 8);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, sink, REPS);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * threads / 64);
        hipMemcpy(h.data(), out, h.size() * sizeof h[0], hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        const double per_wave = s / h.size() / ((double)REPS * per_rep);     // cycles between two instructions of ONE wave
        printf("  %dw/SIMD: %6.2f cyc/instr/wave = %5.2f cyc/instr/SIMD", wps, per_wave, per_wave / wps);
        if (wps == 4) { unsigned long long ck[2]; hipMemcpy(ck, out + 256 * 16 - 2, 16, hipMemcpyDeviceToHost); printf("  [%.0f ticks per us]", ck[1] ? (double)ck[0] / ((double)ck[1] / 100.0) : 0.0); }
    }
    printf("\n");
    hipFree(out); hipFree(sink);
}

#define RUN(id, name) if (only < 0 || only == id) run<id>(name, 64)
int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int only = argc > 1 ? atoi(argv[1]) : -1;      // one test per process: `for i in $(seq 0 30); do timeout 20 ./valu_rate $i; done`
    RUN(0, "v_add_f32"); RUN(1, "v_max_f32"); RUN(2, "v_max3_f32"); RUN(3, "v_cndmask_b32 (vcc never written in the loop)"); RUN(4, "v_cmp_gt_f32");
    RUN(5, "v_add_u32"); RUN(6, "v_addc_co_u32"); RUN(7, "v_and_b32"); RUN(8, "v_perm_b32"); RUN(9, "v_pk_add_f32 (2 adds each)");
    RUN(10, "v_add3_u32"); RUN(11, "v_bfe_u32"); RUN(12, "v_mad_u32_u24"); RUN(13, "v_add_u32_sdwa"); RUN(14, "v_cmp_eq_u32 + v_cndmask (pairs)");
    RUN(20, "v_cmp_e64 s[] ; v_cndmask_e64 s[] (adjacent)"); RUN(21, "4 x v_cmp_e64 s[] then 4 x v_cndmask_e64 s[]");
    RUN(22, "v_cmp vcc ; s_and_b64 vcc ; v_cndmask vcc"); RUN(23, "v_cndmask_e64 with loop-invariant s[] masks");
    RUN(24, "v_cmp vcc ; 3 x v_cndmask vcc"); RUN(25, "v_cmp vcc ; 3 VALU ; v_cndmask vcc ; 2 VALU ; v_cndmask"); RUN(26, "v_cmp vcc ; v_addc vcc (pairs)");
    RUN(27, "ds_read_b32 (dword-aligned windows)"); RUN(28, "ds_read2_b32 (dword-aligned)"); RUN(29, "ds_read_b64 at 4-byte alignment"); RUN(30, "v_alignbyte_b32");
    RUN(31, "v_max3_f32, ONE dependent chain"); RUN(32, "v_max3_f32, two chains"); RUN(33, "v_max3_f32, four chains"); RUN(34, "v_add_f32, ONE dependent chain");
    RUN(35, "clip score cell (cmp,cndmask,add,max3,add dependent)"); RUN(36, "two clip score cells interleaved");
    RUN(40, "v_mul_f32"); RUN(41, "v_sub_f32"); RUN(42, "v_min_f32"); RUN(43, "v_fma_f32"); RUN(44, "v_med3_f32"); RUN(45, "v_fmac_f32");
    RUN(46, "v_or_b32"); RUN(47, "v_xor_b32"); RUN(48, "v_lshlrev_b32"); RUN(49, "v_max_u32"); RUN(50, "v_max_i32"); RUN(51, "v_sub_u32"); RUN(52, "v_min_u32"); RUN(53, "v_mov_b32");
    RUN(54, "v_max_f32 / v_add_f32 alternating"); RUN(55, "v_max3_f32 / v_add_f32 alternating"); RUN(56, "v_max3_f32 : v_add_f32 = 1 : 3"); RUN(57, "v_max_f32, three different registers");
    RUN(58, "v_add_f32 with a 32-bit literal"); RUN(59, "pass-1 cell as shipped (e64 masks two cells ahead, literal)");
    if (only == 60) run<60>("asm row v: vcc, SGPR -5, s_nop 0 per cell", 96); if (only == 61) run<61>("same, no s_nop, VGPR -5", 80); if (only == 62) run<62>("same, no s_nop, SGPR -5", 80);
    if (only == 63) run<63>("same, VGPR -5, compare against an SGPR", 80); if (only == 64) run<64>("pass-2 cell (15 instructions, 2 wait states kept)", 120);
    RUN(65, "v_max3_f32, all registers in bank 0"); RUN(66, "v_max3_f32, sources in three banks"); RUN(67, "v_add_f32, all registers in bank 0"); RUN(68, "v_add_f32, sources in two banks");
    if (only == 69) run<69>("pass-1 cell, every register in bank 0 (v92..v124)", 40); if (only == 70) run<70>("pass-1 cell, registers spread over the banks", 40);
    if (only == 71) run<71>("shipped pass-1 row loop, registers as allocated (CYCLES PER ROW: 80 VALU + 7)", 8); if (only == 72) run<72>("  renamed: no two sources of an instruction in one bank", 8);
    if (only == 73) run<73>("  renamed: sources and destination in different banks", 8); if (only == 74) run<74>("  renamed: ... and not the bank the previous instruction wrote", 8); if (only == 75) run<75>("  renamed: every register in bank 0", 8);
    if (only == 76) run<76>("shipped pass-2 row loop (CYCLES PER ROW: 153 VALU, 24 s_nop)", 8);
    RUN(77, "v_fma_mix_f32 (f16 x f32 + f32)"); RUN(78, "v_cvt_f32_f16");
    if (only == 79) run<79>("table cell: fma_mix, max3, add (chained)", 48); if (only == 80) run<80>("f32-table cell: add, max3, add (chained)", 48);
    if (only == 81) run<81>("f32-table cells, 6 per ds_read_b128 (VALU instr counted)", 36); if (only == 82) run<82>("half-table cells, 4 per ds_read_b128... (VALU instr counted)", 24);
    RUN(15, "ds_add_u32 conflict-free"); RUN(16, "ds_add_u32 all lanes on one bank"); RUN(17, "ds_read_b128 aligned"); RUN(18, "ds_read_b128 unaligned (+5 B)");
    return 0;
}
