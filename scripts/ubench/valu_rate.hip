// valu_rate.hip -- issue rate of the VALU / LDS instructions the clip and quality-stats kernels are built from (gfx950).
// Not part of the product: a measurement aid (`hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate`).
// Every test runs REPS x 64 copies of one instruction over eight independent register chains in every wave and reports
// shader cycles (s_memtime) per wave-instruction, for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define REPS 256
#define X8(s) s s s s s s s s
#define X64(s) X8(X8(s))

#define BODY8(INS)                                    \
    INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

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
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
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
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63u) == 0u) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
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
    RUN(15, "ds_add_u32 conflict-free"); RUN(16, "ds_add_u32 all lanes on one bank"); RUN(17, "ds_read_b128 aligned"); RUN(18, "ds_read_b128 unaligned (+5 B)");
    return 0;
}
