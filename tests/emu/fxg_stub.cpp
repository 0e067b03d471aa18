// fxg_stub.cpp -- TEST-ONLY stand-in for libfxg.so so that the host C layer (fastx_toolkit_amd/host) can be exercised
// on machines without a GPU.  It exports the same C-ABI, keeps "device" memory in host RAM and runs the kernels'
// per-thread code through the serial emulator (fxg_emu.cpp).  The device text path reports every block as irregular,
// so the tools use their host parser.  Never installed next to the product: tests put tests/emu/stub first on
// LD_LIBRARY_PATH.  The product library has no CPU path and this file is not part of it.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/fxg.h"

struct fxg_emu_hist;
extern "C" int fxg_emu_run_pipeline_hist(const fxg_batch *in, const fxg_params *p, const fxg_out *out, char *err, size_t cap, fxg_emu_hist *h);
extern "C" fxg_emu_hist *fxg_emu_hist_new(void);
extern "C" int fxg_emu_run_quality_stats(const fxg_batch *in, uint64_t *hist, uint32_t hist_cols);
extern "C" void fxg_emu_hist_free(fxg_emu_hist *h);

struct fxg_ctx { char err[512]; uint64_t scratch[FXG_NCOUNTERS]; fxg_emu_hist *hist; };

extern "C" {
int fxg_abi_version(void) { return FXG_ABI_VERSION; }
int fxg_ctx_create(int, fxg_ctx **out) { *out = (fxg_ctx *)calloc(1, sizeof(fxg_ctx)); return *out ? 0 : FXG_E_NOMEM; }
void fxg_ctx_destroy(fxg_ctx *c) { if (c) fxg_emu_hist_free(c->hist); free(c); }
int fxg_set_clip_history(fxg_ctx *c, int on) { fxg_emu_hist_free(c->hist); c->hist = on ? fxg_emu_hist_new() : nullptr; return 0; }
const char *fxg_last_error(const fxg_ctx *c) { return c ? c->err : "null"; }
int fxg_set_stream(fxg_ctx *, void *) { return 0; }
int fxg_sync(fxg_ctx *) { return 0; }
int fxg_device_info(fxg_ctx *, int *cus, size_t *mem, char *name, size_t cap) { if (cus) *cus = 1; if (mem) *mem = 0; if (name && cap) snprintf(name, cap, "cpu-emulation stub"); return 0; }
static void *amalloc(size_t n) { void *p = nullptr; if (posix_memalign(&p, 64, n ? n : 16) != 0) return nullptr; memset(p, 0, n ? n : 16); return p; }
int fxg_malloc_device(fxg_ctx *, size_t n, void **p) { *p = amalloc(n); return *p ? 0 : FXG_E_NOMEM; }
int fxg_free_device(fxg_ctx *, void *p) { free(p); return 0; }
int fxg_malloc_host(fxg_ctx *, size_t n, void **p) { *p = amalloc(n); return *p ? 0 : FXG_E_NOMEM; }
int fxg_free_host(fxg_ctx *, void *p) { free(p); return 0; }
int fxg_memcpy_h2d(fxg_ctx *, void *d, const void *s, size_t n) { memcpy(d, s, n); return 0; }
int fxg_memcpy_d2h(fxg_ctx *, void *d, const void *s, size_t n) { memcpy(d, s, n); return 0; }
int fxg_memset_device(fxg_ctx *, void *d, int v, size_t n) { memset(d, v, n); return 0; }
int fxg_timer_start(fxg_ctx *) { return 0; }
int fxg_timer_stop(fxg_ctx *, float *ms) { *ms = 0; return 0; }
int fxg_run_pipeline(fxg_ctx *c, const fxg_batch *in, const fxg_params *p, const fxg_out *out)
{
    fxg_out o = *out;
    if (!o.counters) o.counters = c->scratch;
    return fxg_emu_run_pipeline_hist(in, p, &o, c->err, sizeof c->err, c->hist);
}
int fxg_run_quality_stats(fxg_ctx *, const fxg_batch *in, uint64_t *h, uint32_t cols) { return fxg_emu_run_quality_stats(in, h, cols); }
int fxg_run_qtrim_qfilter(fxg_ctx *, const fxg_batch *, int, int, int, int, int, int, int, const fxg_out *) { return FXG_E_INVALID; }
int fxg_run_clip(fxg_ctx *, const fxg_batch *, const char *, uint32_t, int, int, uint32_t, const fxg_out *) { return FXG_E_INVALID; }
int fxg_run_revcomp_trim(fxg_ctx *, const fxg_batch *, int, int, int, const fxg_out *) { return FXG_E_INVALID; }
int fxg_read_counters(fxg_ctx *c, const uint64_t *d, uint64_t host[FXG_NCOUNTERS])
{
    memcpy(host, d ? d : c->scratch, FXG_NCOUNTERS * sizeof(uint64_t));
    if (host[FXG_C_ERRORS] & FXG_DEV_ERR_BAD_BASE) { snprintf(c->err, sizeof c->err, "Invalid nucleotide value in reverse_complement_base()"); return FXG_E_DEVICE; }
    return 0;
}
int fxg_synth_generate(fxg_ctx *, uint64_t, uint64_t, uint64_t, uint32_t, int, uint8_t *, uint8_t *, uint32_t) { return FXG_E_INVALID; }
int fxg_fastq_index(fxg_ctx *, const uint8_t *, uint64_t, int, int, uint32_t *, uint64_t, uint16_t *, uint8_t *, fxg_text_info *info) { memset(info, 0, sizeof *info); info->irregular = FXG_TEXT_IRR_TAIL; return 0; }
int fxg_fastq_pack(fxg_ctx *, const uint8_t *, uint64_t, int, const uint32_t *, uint64_t, const uint8_t *, uint64_t, uint32_t, int, uint8_t *, uint8_t *, uint32_t *irr) { *irr = 1; return 0; }
int fxg_fastq_format(fxg_ctx *, const uint8_t *, int, const uint32_t *, uint64_t, const uint8_t *, uint64_t, const uint32_t *, uint32_t, int, const uint8_t *, const uint8_t *, const uint64_t *, const uint8_t *, uint32_t, int, int, uint8_t *, uint64_t *n) { *n = 0; return FXG_E_INVALID; }
int fxg_fasta_weights(fxg_ctx *, const uint8_t *, const uint32_t *, uint64_t, uint64_t, const uint32_t *, uint64_t *w) { memset(w, 0, 8 * sizeof(uint64_t)); return FXG_E_INVALID; }
int fxg_device_count(void) { return 1; }
int fxg_shard_range(uint64_t n, uint32_t rank, uint32_t world, uint64_t *lo, uint64_t *hi) { *lo = n * rank / world; *hi = n * (rank + 1) / world; return 0; }
int fxg_epilogue(const uint64_t *, uint32_t, uint32_t, uint64_t *, uint64_t *, uint64_t *) { return FXG_E_INVALID; }
int fxg_concat_pwrite(int, const void *, uint64_t, uint64_t) { return FXG_E_INVALID; }
int fxg_host_register(fxg_ctx *, void *, size_t) { return 0; }
int fxg_host_unregister(fxg_ctx *, void *) { return 0; }
int fxg_set_profiling(fxg_ctx *, int) { return 0; }
int fxg_last_kernel_ms(fxg_ctx *, float *ms) { *ms = 0; return 0; }
int fxg_last_launch_info(const fxg_ctx *, char *name, size_t cap, uint32_t *g, uint32_t *b, uint32_t *l, uint32_t *t) { if (name && cap) name[0] = 0; if (g) *g = 0; if (b) *b = 0; if (l) *l = 0; if (t) *t = 0; return 0; }
}
