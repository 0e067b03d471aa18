// fxg_stub.cpp -- TEST-ONLY stand-in for libfxg.so so that the host C layer (fastx_toolkit_amd/host) can be exercised
// on machines without a GPU.  It exports the same C-ABI, keeps "device" memory in host RAM and runs the kernels'
// per-thread code through the serial emulator (fxg_emu.cpp), the device text path (fxg_text.h) included, so the tools' lanes loop,
// block cutting and sharded runs execute as on a GPU box.  Never installed next to the product: tests put tests/emu/stub first on
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

extern "C" int fxg_emu_fastq_index(void *st, const uint8_t *text, uint64_t text_len, int at_eof, int lpr, uint32_t *d_line, uint64_t cap_lines, uint16_t *d_len, uint8_t *d_flags, fxg_text_info *info);
extern "C" int fxg_emu_fastq_pack(const uint8_t *text, uint64_t text_len, int lpr, const uint32_t *d_line, uint64_t cap_lines, const uint8_t *flags, uint64_t n, uint32_t stride, int qoffset, uint8_t *bases, uint8_t *qual, uint32_t *irregular);
extern "C" int fxg_emu_fastq_format(const uint8_t *text, int lpr, const uint32_t *d_line, uint64_t cap_lines, const uint8_t *flags, uint64_t n, const uint32_t *res, uint32_t fwd_start, int reverse, const uint8_t *pk_bases, const uint8_t *pk_qual, const uint64_t *pk_off, const uint8_t *rows_qual, uint32_t stride, int qoffset, int out_fasta, uint8_t *out, uint64_t *out_bytes);
extern "C" int fxg_emu_fasta_weights(const uint8_t *text, const uint32_t *d_line, uint64_t cap_lines, uint64_t n, const uint32_t *res, uint64_t weighted[8]);

struct fxg_ctx { char err[512]; uint64_t scratch[FXG_NCOUNTERS]; fxg_emu_hist *hist; uint64_t text_state[16]; int device; };

extern "C" {
int fxg_abi_version(void) { return FXG_ABI_VERSION; }
// FXG_EMU_DEVICES (default 1) fake devices, so that FXG_DEVICES=0,1 exercises the per-device bookkeeping of the lanes
static int emu_device_count(void) { const char *e = getenv("FXG_EMU_DEVICES"); return e && atoi(e) > 0 ? atoi(e) : 1; }
int fxg_ctx_create(int dev, fxg_ctx **out)
{
    *out = nullptr;
    if (dev < 0 || dev >= emu_device_count()) return FXG_E_HIP;
    *out = (fxg_ctx *)calloc(1, sizeof(fxg_ctx));
    if (*out) { (*out)->device = dev; if (const char *log = getenv("FXG_EMU_LOG")) { FILE *f = fopen(log, "a"); if (f) { fprintf(f, "ctx device %d\n", dev); fclose(f); } } }
    return *out ? 0 : FXG_E_NOMEM;
}
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
int fxg_fastq_index(fxg_ctx *c, const uint8_t *t, uint64_t len, int eof, int lpr, uint32_t *line, uint64_t cap, uint16_t *l16, uint8_t *fl, fxg_text_info *info)
{
    if (getenv("FXG_EMU_NO_TEXT")) { memset(info, 0, sizeof *info); info->irregular = FXG_TEXT_IRR_TAIL; return 0; }     // every block to the host parser
    return fxg_emu_fastq_index(c->text_state, t, len, eof, lpr, line, cap, l16, fl, info);
}
int fxg_fastq_pack(fxg_ctx *, const uint8_t *t, uint64_t len, int lpr, const uint32_t *line, uint64_t cap, const uint8_t *fl, uint64_t n, uint32_t stride, int qo, uint8_t *b, uint8_t *q, uint32_t *irr)
{ return fxg_emu_fastq_pack(t, len, lpr, line, cap, fl, n, stride, qo, b, q, irr); }
int fxg_fastq_format(fxg_ctx *, const uint8_t *t, int lpr, const uint32_t *line, uint64_t cap, const uint8_t *fl, uint64_t n, const uint32_t *res, uint32_t fs, int rev, const uint8_t *pb, const uint8_t *pq,
                     const uint64_t *po, const uint8_t *rq, uint32_t stride, int qo, int fa, uint8_t *out, uint64_t *nb)
{ return fxg_emu_fastq_format(t, lpr, line, cap, fl, n, res, fs, rev, pb, pq, po, rq, stride, qo, fa, out, nb); }
int fxg_fasta_weights(fxg_ctx *, const uint8_t *t, const uint32_t *line, uint64_t cap, uint64_t n, const uint32_t *res, uint64_t *w) { return fxg_emu_fasta_weights(t, line, cap, n, res, w); }
int fxg_device_count(void) { return emu_device_count(); }
int fxg_device_numa_node(int device) { (void)device; const char *e = getenv("FXG_EMU_NUMA_NODE"); return e ? atoi(e) : -1; }   /* no GPU, no node; the env lets a test walk the binding code */
int fxg_concat_peer(fxg_ctx *dst, void *d, uint64_t off, fxg_ctx *src, const void *s, uint64_t n) { if (!dst || !src) return FXG_E_INVALID; if (n) memcpy((char *)d + off, s, n); return 0; }
int fxg_host_register(fxg_ctx *, void *, size_t) { return 0; }
int fxg_host_unregister(fxg_ctx *, void *) { return 0; }
int fxg_set_profiling(fxg_ctx *, int) { return 0; }
int fxg_last_kernel_ms(fxg_ctx *, float *ms) { *ms = 0; return 0; }
int fxg_profiled_kernel_ms(fxg_ctx *, float *, uint32_t, uint32_t *n) { *n = 0; return 0; }
int fxg_scan_recoveries(const fxg_ctx *) { return 0; }      /* (nothing waits in the serial emulation) */
int fxg_last_launch_info(const fxg_ctx *, char *name, size_t cap, uint32_t *g, uint32_t *b, uint32_t *l, uint32_t *t) { if (name && cap) name[0] = 0; if (g) *g = 0; if (b) *b = 0; if (l) *l = 0; if (t) *t = 0; return 0; }
}

// The multi-GPU host code is the product's own (csrc/fxg_comm.h) over this stub's host-memory "device": shard ranges, epilogue,
// concatenation and the RCCL transport -- which dlopens librccl.so.1, i.e. tests/emu/fakerccl's shared-memory stand-in when the
// test puts it first on LD_LIBRARY_PATH -- run here with world > 1 (tests/test_comm_cpu.py).
#include <cstdarg>
static int stub_fail(fxg_ctx *c, int code, const char *fmt, ...)
{
    if (c) { va_list ap; va_start(ap, fmt); vsnprintf(c->err, sizeof c->err, fmt, ap); va_end(ap); }
    return code;
}
#define FXG_COMM_FAIL(c, code, ...) stub_fail(c, code, __VA_ARGS__)
#define FXG_COMM_SET_DEVICE(c) ((c)->device >= 0)
#define FXG_COMM_MALLOC(pp, bytes) ((*(void **)(pp) = getenv("FXG_EMU_COMM_NOMEM") ? nullptr : amalloc(bytes)) != nullptr)
#define FXG_COMM_FREE(p) free(p)
#define FXG_COMM_STREAM(c) ((void *)nullptr)
#define FXG_COMM_SCRATCH(c) ((const uint64_t *)(c)->scratch)
static const char *fxg_comm_d2h_sync(fxg_ctx *, void *dst, const void *src, size_t bytes) { memcpy(dst, src, bytes); return nullptr; }
#include "../../fastx_toolkit_amd/csrc/fxg_comm.h"

