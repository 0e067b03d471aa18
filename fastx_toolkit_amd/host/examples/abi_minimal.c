/* abi_minimal.c -- the smallest C caller of the engine's C-ABI (include/fxg.h): no HIP headers, no Python.
 *
 *   gcc -O2 abi_minimal.c -I../../../include -L../.. -lfxg -Wl,-rpath,'$ORIGIN/../../..' -o abi_minimal
 *   ./abi_minimal [reads] [length]
 *
 * Generates `reads` synthetic reads on the device (SURVEY 8d generator, seed 2), runs
 *   fastq_quality_trimmer -t 20 -l 30 | fastq_quality_filter -q 20 -p 80
 * as one fused pass with order-preserving compaction and prints the counters the tools' -v reports are made of. */
#include <stdio.h>
#include <stdlib.h>

#include "fxg.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, fxg_last_error(ctx)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const uint64_t n = argc > 1 ? strtoull(argv[1], NULL, 10) : 1000000;
    const uint32_t L = argc > 2 ? (uint32_t)atoi(argv[2]) : 150;
    fxg_ctx *ctx = NULL;
    if (fxg_ctx_create(0, &ctx) != 0) { fprintf(stderr, "no usable HIP device\n"); return 1; }

    uint8_t *bases, *qual, *out_bases, *out_qual;
    uint32_t *res;
    uint64_t *counters;
    const size_t bytes = (size_t)n * L;
    CHECK(fxg_malloc_device(ctx, bytes + 16, (void **)&bases));
    CHECK(fxg_malloc_device(ctx, bytes + 16, (void **)&qual));
    CHECK(fxg_malloc_device(ctx, bytes + 16, (void **)&out_bases));
    CHECK(fxg_malloc_device(ctx, bytes + 16, (void **)&out_qual));
    CHECK(fxg_malloc_device(ctx, n * sizeof(uint32_t), (void **)&res));
    CHECK(fxg_malloc_device(ctx, FXG_NCOUNTERS * sizeof(uint64_t), (void **)&counters));
    CHECK(fxg_synth_generate(ctx, 2, 0, n, L, 0, bases, qual, L));

    fxg_batch in = { bases, qual, NULL /* every read has fixed_len bases */, L, L, n };
    fxg_out out = { res, out_bases, out_qual, NULL, NULL, NULL, counters };
    float ms = 0;
    CHECK(fxg_timer_start(ctx));
    CHECK(fxg_run_qtrim_qfilter(ctx, &in, /* -Q */ 33, 1, /* -t */ 20, /* -l */ 30, 1, /* -q */ 20, /* -p */ 80, &out));
    CHECK(fxg_timer_stop(ctx, &ms));

    uint64_t c[FXG_NCOUNTERS];
    CHECK(fxg_read_counters(ctx, counters, c));
    printf("input %llu kept %llu kept_bases %llu qtrim_dropped %llu qfilter_dropped %llu\n", (unsigned long long)c[FXG_C_INPUT],
           (unsigned long long)c[FXG_C_KEPT], (unsigned long long)c[FXG_C_KEPT_BASES], (unsigned long long)c[FXG_C_QTRIM_DROPPED],
           (unsigned long long)c[FXG_C_QFILTER_DROPPED]);
    fprintf(stderr, "%.3f ms on the device stream (%.1f Mreads/s)\n", ms, (double)n / ms / 1e3);
    fxg_free_device(ctx, bases); fxg_free_device(ctx, qual); fxg_free_device(ctx, out_bases); fxg_free_device(ctx, out_qual);
    fxg_free_device(ctx, res); fxg_free_device(ctx, counters);
    fxg_ctx_destroy(ctx);
    return 0;
}
