// fxg_emu.cpp -- TEST-ONLY serial emulator of the tile kernels.
//
// Compiled for the HOST only (hipcc --cuda-host-only); it runs the very same __host__ __device__
// per-thread phase bodies the GPU kernels run (bitmap build, per-read decision incl. the clipper DP,
// chunk gather) with threadIdx replaced by a loop, and replaces only the wave-level pieces (workgroup
// scan, decoupled look-back) by serial prefix sums.  It lets the CPU-only test tier check the device
// logic against the oracle without a GPU.  It is NOT part of the product and is never loaded by
// fastx_toolkit_amd; the product path has no CPU fallback.
#include <cstdlib>
#include <vector>

#include "../../fastx_toolkit_amd/csrc/fxg_plan.h"
#include "../../fastx_toolkit_amd/csrc/fxg_history.h"
#include "../../fastx_toolkit_amd/csrc/fxg_stats.h"
#include "../../fastx_toolkit_amd/csrc/fxg_text.h"

// fxg_kernel_rows: a lane holds NW registers of its read's quality row from byte `from` on (bytes past the batch: whatever follows in
// the staging buffer); with two lanes per read the pieces' results are combined as the kernel's DPP swaps do
template <int NW>
static void emu_rows_piece(const FxgKArgs &a, u32 read, u32 from, u32 (&q)[NW])
{
    for (int k = 0; k < NW; ++k) {
        u32 w = 0;
        for (int i = 0; i < 4; ++i) {
            const u64 at = (u64)read * a.stride + from + 4u * (u32)k + (u32)i;
            w |= (u32)(at < a.total_bytes ? a.qual[at] : 0xA5u) << (8 * i);
        }
        q[k] = w;
    }
}
template <int NW>
static void emu_rows_decide_nw(const FxgKArgs &a, int h, u32 read, u32 *keep, u32 *olen)
{
    constexpr int NM = (NW * 4 + 31) / 32;
    constexpr u32 HB = 4u * NW;
    u32 q0[NW], q1[NW];
    emu_rows_piece<NW>(a, read, 0, q0);
    if (h == 1) { fxg_rows_decide<NW>(a, q0, read, keep, olen); return; }
    emu_rows_piece<NW>(a, read, HB, q1);
    const u32 rl = a.len ? (u32)a.len[read] : a.fixed_len;
    u32 G0[NM], G1[NM];
    const u32 k_lo = fxg_rows_piece_last<NW>(a, q0, fxg_rows_piece_len(rl, HB, 0), G0), k_hi = fxg_rows_piece_last<NW>(a, q1, fxg_rows_piece_len(rl, HB, 1), G1);
    const u32 k = k_hi ? HB + k_hi : k_lo;
    const u32 cl = (a.stages & FXG_STAGE_QTRIM) ? k : rl;
    const u32 low = fxg_rows_piece_low<NW>(a, q0, G0, fxg_rows_piece_len(cl, HB, 0)) + fxg_rows_piece_low<NW>(a, q1, G1, fxg_rows_piece_len(cl, HB, 1));
    a.res[read] = fxg_rows_verdict(a, rl, k, low, keep, olen);
}
static void emu_rows_decide(int nw, int h, const FxgKArgs &a, u32 read, u32 *keep, u32 *olen)
{
    if (nw == 10) emu_rows_decide_nw<10>(a, h, read, keep, olen);           // (fxg_kernel_rows_multi: the same decision, shorter register rows)
    else if (nw == 14) emu_rows_decide_nw<14>(a, h, read, keep, olen);
    else if (nw == 20) emu_rows_decide_nw<20>(a, h, read, keep, olen);
    else if (nw == 26) emu_rows_decide_nw<26>(a, h, read, keep, olen);
    else emu_rows_decide_nw<38>(a, h, read, keep, olen);
}

template <int AMAX, bool REV, int MODE = 0>
static int emu_run(const FxgPlan &pl, uint64_t *counters, char *err, size_t cap)
{
    const FxgKArgs &a = pl.ka;
    const u32 T = a.tile_reads, stride = a.stride, NT = FXG_TBLOCK;
    std::vector<unsigned char> lds(fxg_plan_lds(&pl) + 64, 0);   // the general tile layout, also where the engine would pick fxg_kernel_rows
    unsigned char *smem = lds.data();
    const FxgLds L = fxg_plan_layout(&pl);
    u64 m_reads = 0, m_nt = 0;
    u32 *k_off = reinterpret_cast<u32 *>(smem);
    u32 *k_src = reinterpret_cast<u32 *>(smem + L.so_ksrc);
    uint16_t *k_tab = L.has_tab ? reinterpret_cast<uint16_t *>(smem + L.so_ktab) : nullptr;
    u32 *bm_g = reinterpret_cast<u32 *>(smem + L.off_bm_g);
    u32 *bm_l = reinterpret_cast<u32 *>(smem + L.off_bm_l);
    uint8_t *sb = smem + L.off_bases;
    FxgCounts cnt = {};
    u64 base_c = 0, base_b = 0;
    u32 bad = 0;
    std::vector<u32> keep(T), olen(T), anchor(T);
    if (pl.group_a && fxg_clip_uses_ptab(AMAX)) { for (u32 tid = 0; tid < NT; ++tid) fxg_clip_ptab_build(a, smem + L.off_ptab, tid, NT); }
    for (u32 tile = 0; tile < a.ntiles; ++tile) {
        const u32 r0 = tile * T;
        const u64 left = a.n - (u64)r0;
        const u32 nreads = left < (u64)T ? (u32)left : T;
        const u64 tb = (u64)r0 * stride;
        const u32 tbytes = nreads * stride;
        if (pl.group_a) {
            for (u32 tid = 0; tid < NT; ++tid) {
                if (pl.use_q) fxg_phase_bitmaps(a, tb, tbytes, bm_g, bm_l, tid, NT);
                if constexpr (AMAX != 0) { if (!a.clip_global) fxg_phase_stage_bases(a.clip_src, a.clip_total, (u64)r0 * a.clip_stride, nreads * a.clip_stride, sb, tid, NT); }
            }
            for (u32 tid = 0; tid < nreads; ++tid) {
                if (AMAX == 0 && pl.rows_nw) emu_rows_decide(pl.rows_nw, pl.rows_h, a, r0 + tid, &keep[tid], &olen[tid]);   // what a lane of fxg_kernel_rows does
                else if (AMAX < -16 && pl.ck_per_wg) {        // the two-pass form with its checkpoint scratch (here: one thread's, stride 1)
                    std::vector<float> ck((size_t)FXG_CK_SLOTS * (size_t)(AMAX < 0 ? fxg_clip_cols(AMAX) : 1), (getenv("FXG_EMU_CK_FILL") ? (float)atof(getenv("FXG_EMU_CK_FILL")) : 1.0e30f));   // the device's scratch is not cleared either
                    const uint8_t *pt = fxg_clip_uses_ptab(AMAX) ? smem + L.off_ptab : nullptr;
                    if (a.clip_global) { if constexpr (AMAX < -16) fxg_decide_a<AMAX, true>(a, bm_g, bm_l, sb, r0, tid, &keep[tid], &olen[tid], ck.data(), 1u, pt); }
                    else fxg_decide_a<AMAX>(a, bm_g, bm_l, sb, r0, tid, &keep[tid], &olen[tid], ck.data(), 1u, pt);
                } else if (AMAX < 0 && AMAX >= -16 && a.clip_global) {     // the DP straight over the batch (fxg_plan.h: clip_global), as the kernel calls it
                    if constexpr (AMAX < 0 && AMAX >= -16) fxg_decide_a<AMAX, true>(a, bm_g, bm_l, sb, r0, tid, &keep[tid], &olen[tid], nullptr, 0u, fxg_clip_uses_ptab(AMAX) ? smem + L.off_ptab : nullptr);
                } else if (fxg_clip_uses_ptab(AMAX)) {                     // the staged register form: pair values out of the workgroup's table
                    fxg_decide_a<AMAX>(a, bm_g, bm_l, sb, r0, tid, &keep[tid], &olen[tid], nullptr, 0u, smem + L.off_ptab);
                } else fxg_decide_a<AMAX>(a, bm_g, bm_l, sb, r0, tid, &keep[tid], &olen[tid]);
                anchor[tid] = tid * stride;
            }
        } else if (MODE == 3) {
            for (u32 tid = 0; tid < NT; ++tid) fxg_phase_bitmaps(a, tb, tbytes, bm_g, bm_l, tid, NT);
            for (u32 tid = 0; tid < nreads; ++tid) {
                u32 nl;
                fxg_decide_mask(a, bm_l, r0, tid, &keep[tid], &olen[tid], &nl);
                anchor[tid] = tid * stride; m_nt += nl; m_reads += (nl != 0u);
            }
        } else if (MODE == 4) {
            const FxgCensusBm cb = fxg_census_bitmaps(bm_g, T, stride);
            for (u32 tid = 0; tid < NT; ++tid) fxg_phase_census(a, tb, tbytes, cb, tid, NT);
            for (u32 tid = 0; tid < nreads; ++tid) { fxg_decide_census(a, cb, tid * stride, r0, tid, &keep[tid], &olen[tid], &bad); anchor[tid] = tid * stride; }
        } else {
            for (u32 tid = 0; tid < nreads; ++tid) fxg_decide_b<REV>(a, r0, tid, &keep[tid], &olen[tid], &anchor[tid]);
        }
        if (!a.compact) continue;
        u32 exb = 0, exc = 0;
        for (u32 tid = 0; tid < nreads; ++tid) {
            if (!keep[tid]) continue;
            k_off[exc] = exb; k_src[exc] = anchor[tid];
            if (k_tab) fxg_tab_fill(k_tab, exc, exb, olen[tid]);
            fxg_write_kept_meta(a, base_c + exc, olen[tid], r0 + tid, base_b + exb);
            exb += olen[tid]; exc++;
        }
        k_off[exc] = exb;
        for (u32 tid = 0; tid < NT; ++tid) bad |= fxg_tile_gather<REV, MODE == 3>(a, k_off, k_src, k_tab, exc, tb, tbytes, base_b, exb, tid, NT);
        base_c += exc; base_b += exb;
    }
    for (u64 i = 0; i < a.n; ++i) fxg_count_res(a.res[i], cnt);   // same reduction the counting kernel performs
    if (counters) {
        u64 slot[FXG_NCOUNTERS];
        fxg_counts_to_slots(cnt, a.stages, slot);
        for (int i = 0; i < FXG_NCOUNTERS; ++i) counters[i] = slot[i];
        counters[FXG_C_ERRORS] = ((REV || MODE == 4) && bad) ? FXG_DEV_ERR_BAD_BASE : 0;
        counters[FXG_C_MASKED_READS] = m_reads; counters[FXG_C_MASKED_NT] = m_nt;
    }
    (void)err; (void)cap;
    return FXG_OK;
}

// The clipper's instances, in groups that tests/emu_py.py compiles side by side (-DFXG_EMU_TU=k: group k alone, 0: everything else; no define:
// the whole emulator in one unit).  A group answers EMU_NOT_MINE for a plan it holds no instance of.
enum { EMU_NOT_MINE = -12345, FXG_EMU_CLIP_GROUPS = 7 };
typedef int emu_clip_group_fn(const FxgPlan &, uint64_t *, char *, size_t);
#define EMU_HIDDEN __attribute__((visibility("hidden")))
EMU_HIDDEN emu_clip_group_fn emu_clip_rest;         // the general instance (any adapter length)
EMU_HIDDEN emu_clip_group_fn emu_clip_group1;
EMU_HIDDEN emu_clip_group_fn emu_clip_group2;
EMU_HIDDEN emu_clip_group_fn emu_clip_group3;
EMU_HIDDEN emu_clip_group_fn emu_clip_group4;
EMU_HIDDEN emu_clip_group_fn emu_clip_group5;
EMU_HIDDEN emu_clip_group_fn emu_clip_group6;
EMU_HIDDEN emu_clip_group_fn emu_clip_group7;
#if !defined(FXG_EMU_TU) || FXG_EMU_TU == 1
int emu_clip_group1(const FxgPlan &pl, uint64_t *ctr, char *err, size_t cap)
{
    switch (pl.amax) {
    case -4: return emu_run<-4, false>(pl, ctr, err, cap);
    case -8: return emu_run<-8, false>(pl, ctr, err, cap);
    case -9: return emu_run<-9, false>(pl, ctr, err, cap);
    case -10: return emu_run<-10, false>(pl, ctr, err, cap);
    case -11: return emu_run<-11, false>(pl, ctr, err, cap);
    case -12: return emu_run<-12, false>(pl, ctr, err, cap);
    case -13: return emu_run<-13, false>(pl, ctr, err, cap);
    case -14: return emu_run<-14, false>(pl, ctr, err, cap);
    case -15: return emu_run<-15, false>(pl, ctr, err, cap);
    case -16: return emu_run<-16, false>(pl, ctr, err, cap);
    case 16: return emu_run<16, false>(pl, ctr, err, cap);
    case 32: return emu_run<32, false>(pl, ctr, err, cap);
    case 64: return emu_run<64, false>(pl, ctr, err, cap);
#ifdef FXG_CLIP_ONE_PASS
    case -216: return emu_run<-216, false>(pl, ctr, err, cap);
#endif
    default: return EMU_NOT_MINE;
    }
}
int emu_clip_rest(const FxgPlan &pl, uint64_t *ctr, char *err, size_t cap) { return emu_run<100, false>(pl, ctr, err, cap); }
#endif
#if !defined(FXG_EMU_TU) || FXG_EMU_TU == 2
int emu_clip_group2(const FxgPlan &pl, uint64_t *ctr, char *err, size_t cap)
{
    switch (pl.amax) {
    case -20: return emu_run<-20, false>(pl, ctr, err, cap);
    case -24: return emu_run<-24, false>(pl, ctr, err, cap);
    case -28: return emu_run<-28, false>(pl, ctr, err, cap);
    case -32: return emu_run<-32, false>(pl, ctr, err, cap);
    case -36: return emu_run<-36, false>(pl, ctr, err, cap);
    case -40: return emu_run<-40, false>(pl, ctr, err, cap);
    default: return EMU_NOT_MINE;
    }
}
#endif
#if !defined(FXG_EMU_TU) || FXG_EMU_TU == 3
int emu_clip_group3(const FxgPlan &pl, uint64_t *ctr, char *err, size_t cap)
{
    switch (pl.amax) {
    case -48: return emu_run<-48, false>(pl, ctr, err, cap);
    case -56: return emu_run<-56, false>(pl, ctr, err, cap);
    case -64: return emu_run<-64, false>(pl, ctr, err, cap);
    default: return EMU_NOT_MINE;
    }
}
#endif
#if !defined(FXG_EMU_TU) || FXG_EMU_TU == 4
int emu_clip_group4(const FxgPlan &pl, uint64_t *ctr, char *err, size_t cap)
{
    switch (pl.amax) {
    case -80: return emu_run<-80, false>(pl, ctr, err, cap);
    case -100: return emu_run<-100, false>(pl, ctr, err, cap);
    default: return EMU_NOT_MINE;
    }
}
#endif
#if !defined(FXG_EMU_TU) || FXG_EMU_TU == 5
int emu_clip_group5(const FxgPlan &pl, uint64_t *ctr, char *err, size_t cap)
{
    switch (pl.amax) {
    case -44: return emu_run<-44, false>(pl, ctr, err, cap);       // (the buckets of round 6, where the N instances used to be: an N is a column pattern of the pair table now)
    case -52: return emu_run<-52, false>(pl, ctr, err, cap);
    default: return EMU_NOT_MINE;
    }
}
#endif
#if !defined(FXG_EMU_TU) || FXG_EMU_TU == 6
int emu_clip_group6(const FxgPlan &pl, uint64_t *ctr, char *err, size_t cap)
{
    switch (pl.amax) {
    case -60: return emu_run<-60, false>(pl, ctr, err, cap);
    case -72: return emu_run<-72, false>(pl, ctr, err, cap);
    default: return EMU_NOT_MINE;
    }
}
#endif
#if !defined(FXG_EMU_TU) || FXG_EMU_TU == 7
int emu_clip_group7(const FxgPlan &pl, uint64_t *ctr, char *err, size_t cap)
{
    switch (pl.amax) {
    case -88: return emu_run<-88, false>(pl, ctr, err, cap);
    default: return EMU_NOT_MINE;
    }
}
#endif

#if !defined(FXG_EMU_TU) || FXG_EMU_TU == 0
static emu_clip_group_fn *const emu_clip_groups[FXG_EMU_CLIP_GROUPS] = {emu_clip_group1, emu_clip_group2, emu_clip_group3, emu_clip_group4, emu_clip_group5, emu_clip_group6, emu_clip_group7};

// clip history (fxg_history.h): the same per-column bodies the pre-pass kernels run, serially
struct fxg_emu_hist {
    std::vector<uint8_t> buf[2];
    u32 w[2];
    int cur;
    u32 wcap;
    std::vector<u32> M, BT;
    std::vector<uint8_t> ext;
    std::vector<uint16_t> wlen;
};
extern "C" fxg_emu_hist *fxg_emu_hist_new(void)
{
    fxg_emu_hist *h = new fxg_emu_hist();
    h->buf[0].assign(FXG_HIST_CAP, 0); h->buf[1].assign(FXG_HIST_CAP, 0);
    h->w[0] = h->w[1] = 0; h->cur = 0; h->wcap = 0;
    return h;
}
extern "C" void fxg_emu_hist_free(fxg_emu_hist *h) { delete h; }

static void emu_hist_prepass(fxg_emu_hist *hs, const fxg_batch *in, u32 T, u32 estride, FxgKArgs *ka, int *use)
{
    const u32 lmax = in->len ? in->stride : in->fixed_len;
    const int cur = hs->cur;
    *use = 0;
    if (!in->len && hs->wcap <= in->fixed_len) {
        const u32 L = in->fixed_len;
        for (u32 x = 0; x < FXG_HIST_CAP; ++x)
            hs->buf[cur ^ 1][x] = x < L ? in->bases[(in->n - 1u) * in->stride + x] : (x == L ? (uint8_t)0 : hs->buf[cur][x]);
        hs->w[cur ^ 1] = hs->w[cur] > L ? hs->w[cur] : L;
    } else {
        const u32 S2 = in->stride + 2u;
        const u32 ntiles = (u32)((in->n + T - 1) / T), nblk = (ntiles + FXG_HIST_BLOCK - 1) / FXG_HIST_BLOCK;
        hs->M.assign((size_t)ntiles * S2, 0); hs->BT.assign((size_t)nblk * S2, 0);
        hs->ext.assign((size_t)in->n * estride + 16, 0xEE); hs->wlen.assign(in->n, 0);
        FxgHist h;
        h.bases = in->bases; h.len = in->len; h.fixed_len = in->fixed_len; h.stride = in->stride; h.n = in->n;
        h.tile_reads = T; h.ntiles = ntiles; h.M = hs->M.data(); h.BT = hs->BT.data();
        h.ext = hs->ext.data(); h.estride = estride; h.wlen = hs->wlen.data();
        h.hist_in = hs->buf[cur].data(); h.w_in = &hs->w[cur]; h.hist_out = hs->buf[cur ^ 1].data(); h.w_out = &hs->w[cur ^ 1];
        for (u32 t = 0; t < ntiles; ++t) for (u32 x = 0; x < S2; ++x) fxg_hist_tile_column(h, t, x);
        for (u32 b = 0; b < nblk; ++b) for (u32 x = 0; x < S2; ++x) fxg_hist_block_column(h, b, x);
        for (u32 x = 0; x < S2; ++x) fxg_hist_top_column(h, nblk, x);
        const u32 ncol = fxg_hist_columns(h);
        for (u32 t = 0; t < ntiles; ++t) for (u32 x = 0; x < ncol; ++x) fxg_hist_extend_column(h, t, x);
        ka->clip_src = h.ext; ka->clip_stride = estride; ka->clip_total = (u64)in->n * estride; ka->wlen = h.wlen;
        *use = 1;
    }
    hs->cur = cur ^ 1;
    if (lmax > hs->wcap) hs->wcap = lmax;
}

// what fxg_make_plan chose for the last pipeline call: the clip instance (FxgPlan.amax) and whether it runs its two-pass form with checkpoints in scratch
static int g_last_amax, g_last_two_pass, g_last_clip_global, g_last_tile;
extern "C" void fxg_emu_last_plan(int *amax, int *two_pass) { *amax = g_last_amax; *two_pass = g_last_two_pass; }
extern "C" void fxg_emu_last_plan_clip_global(int *on, int *tile_reads) { *on = g_last_clip_global; *tile_reads = g_last_tile; }

#ifdef FXG_CLIP_DEBUG
static std::vector<u32> g_clip_dbg;
extern "C" size_t fxg_emu_clip_debug(u32 *out, size_t cap_words) { const size_t k = g_clip_dbg.size() < cap_words ? g_clip_dbg.size() : cap_words; memcpy(out, g_clip_dbg.data(), k * 4); return k; }
#endif
extern "C" int fxg_emu_run_pipeline_hist(const fxg_batch *in, const fxg_params *p, const fxg_out *out, char *err, size_t cap, fxg_emu_hist *hs)
{
    FxgPlan pl;
    const bool hist = hs && (p->stages & FXG_STAGE_CLIP) && in->n != 0;
    const u32 estride = hist && hs->wcap > in->stride ? hs->wcap : in->stride;
    const int rc = fxg_make_plan(in, p, out, &pl, err, cap, hist ? estride : 0u);
    if (rc != FXG_OK) return rc;
    g_last_amax = pl.amax; g_last_two_pass = pl.ck_per_wg != 0; g_last_clip_global = (int)pl.ka.clip_global; g_last_tile = (int)pl.ka.tile_reads;
    if (in->n == 0) return FXG_OK;
#ifdef FXG_CLIP_DEBUG      // debug builds (scripts/debug/clip64_bisect.py): the per-read dump of fxg_clip_two_pass_k, read back through fxg_emu_clip_debug
    g_clip_dbg.assign((size_t)in->n * FXG_CLIP_DBG_WORDS, 0xEEEEEEEEu);
    pl.ka.clip_dbg = g_clip_dbg.data();
#endif
    if (hist) {
        int use = 0;
        emu_hist_prepass(hs, in, pl.ka.tile_reads, estride, &pl.ka, &use);
        if (!use) { pl.ka.clip_src = in->bases; pl.ka.clip_stride = in->stride; pl.ka.clip_total = in->n * (u64)in->stride; pl.ka.wlen = nullptr; pl.lds = fxg_plan_lds(&pl); }
    }
    uint64_t *ctr = out->counters;
    if (pl.group_a) {
        if (pl.amax == 0) return emu_run<0, false>(pl, ctr, err, cap);
        for (int g = 0; g < FXG_EMU_CLIP_GROUPS; ++g) {
            const int r = emu_clip_groups[g](pl, ctr, err, cap);
            if (r != EMU_NOT_MINE) return r;
        }
        return emu_clip_rest(pl, ctr, err, cap);
    }
    if (pl.mask) return emu_run<0, false, 3>(pl, ctr, err, cap);
    if (pl.artifacts) return emu_run<0, false, 4>(pl, ctr, err, cap);
    return pl.rev ? emu_run<0, true>(pl, ctr, err, cap) : emu_run<0, false>(pl, ctr, err, cap);
}

extern "C" int fxg_emu_run_pipeline(const fxg_batch *in, const fxg_params *p, const fxg_out *out, char *err, size_t cap)
{
    return fxg_emu_run_pipeline_hist(in, p, out, err, cap, nullptr);
}

extern "C" unsigned fxg_emu_tile_reads(unsigned stride, int clip) { return fxg_pick_tile(stride, clip != 0); }
// the GL window's clamp (fxg_kernels.h: fxg_gl_last_dword), so that batches of 8 GiB and more can be checked without allocating one
extern "C" int fxg_emu_gl_last_dword(uint64_t total, uint64_t off) { return fxg_gl_last_dword(total, off); }

// fastx_quality_stats: the per-thread bodies of fxg_kernel_quality_stats / _fold, one "workgroup" after the other
static uint64_t g_qs_piece_trips = 0, g_qs_piece_moved = 0;      // trips the piece form has run since the library was loaded (the tests ask whether the form under test ran)
extern "C" uint64_t fxg_emu_quality_stats_piece_trips(void) { return g_qs_piece_trips; }
extern "C" uint64_t fxg_emu_quality_stats_piece_moved(void) { return g_qs_piece_moved; }      // ... whose first cut was moved onto the line grid
extern "C" int fxg_emu_run_quality_stats(const fxg_batch *in, uint64_t *hist, uint32_t hist_cols)
{
    if (!in || !hist || !in->bases || in->stride == 0 || hist_cols < in->stride) return FXG_E_INVALID;
    if (in->n == 0) return FXG_OK;
    FxgStatsArgs a;
    a.bases = in->bases; a.qual = in->qual; a.len = in->len; a.n = in->n; a.total_bytes = in->n * (u64)in->stride;
    a.fixed_len = in->fixed_len; a.stride = in->stride; a.hist = (u64 *)hist; a.hist_cols = hist_cols;
    a.nwg = (u32)((in->n + 255) / 256 < 3 ? (in->n + 255) / 256 : 3);
    std::vector<u32> partial((size_t)a.nwg * FXG_QS_PART_WORDS), lds(FXG_QS_LDS_WORDS);
    a.partial = partial.data();
    const u32 nstrips = (in->stride + FXG_QS_STRIP - 1) / FXG_QS_STRIP;
    const u32 trip_reads = (FXG_QS_TBLOCK * FXG_QS_UNROLL + FXG_QS_WAVES - 1u) / FXG_QS_WAVES + 1u;
    for (u32 s0 = 0; s0 < nstrips; s0 += FXG_QS_WAVES) {
        a.strip0 = s0;
        std::fill(partial.begin(), partial.end(), 0xDEADBEEFu);     // the kernel does not clear its partial: a workgroup's first flush stores every counter
        // the piece form (dense even-length batches of 16 .. 160 bytes): workgroup g takes trips g, g + nwg, ... as the kernel's round-robin loop does, one piece per
        // "lane"; the reads behind the last whole trip go through workgroup 0's row-strip loop.  FXG_EMU_QS_ROWS=1 keeps the row-strip form (the kernel's round_robin 3).
        u32 pR = 0, pP = 0;
        a.round_robin = 1u;
        const bool piece = a.n && !getenv("FXG_EMU_QS_ROWS") && fxg_stats_piece_plan(a, &pR, &pP);
        const u64 ntrip = piece ? a.n / pR : 0u;
        if (piece && ntrip >= 8u && !getenv("FXG_EMU_QS_NOGRID")) a.nwg = 8u;      // (a multiple of eight workgroups: the cuts between trips go onto the line grid)
        if (partial.size() < (size_t)a.nwg * FXG_QS_PART_WORDS) { partial.resize((size_t)a.nwg * FXG_QS_PART_WORDS); a.partial = partial.data(); std::fill(partial.begin(), partial.end(), 0xDEADBEEFu); }
        const u32 wrap = getenv("FXG_EMU_QS_FLUSH") ? 600u : 65535u;      // the env knob exercises the wrap guard on small inputs
        for (u32 g = 0; g < a.nwg; ++g) {
            std::fill(lds.begin(), lds.end(), 0u);
            u32 *part = a.partial + (u64)g * FXG_QS_PART_WORDS;
            u32 since = 0, nflush = 0;
            auto flush = [&]() { for (u32 t = 0; t < FXG_QS_TBLOCK; ++t) fxg_stats_flush(lds.data(), part, t, FXG_QS_TBLOCK, nflush == 0u); since = 0; ++nflush; };
            auto rows = [&](u64 lo, u64 hi) {
                const u64 nitems = (hi - lo) * FXG_QS_WAVES;
                for (u64 g0 = 0; g0 < nitems; g0 += (u64)FXG_QS_TBLOCK * FXG_QS_UNROLL) {
                    if (since + trip_reads > wrap) flush();
                    for (u32 t = 0; t < FXG_QS_TBLOCK; ++t)
                        for (u32 u = 0; u < FXG_QS_UNROLL; ++u) {
                            const u64 it = g0 + (u64)u * FXG_QS_TBLOCK + t;
                            if (it >= nitems) continue;
                            u64 r; u32 sl;
                            FxgStripRow row;
                            fxg_stats_item(lo, it, &r, &sl);
                            fxg_stats_load(a, r, s0 + sl, row);
                            u32 m[4];
                            fxg_stats_masks(row.nb, m);
                            fxg_stats_accumulate(a, row, sl, (s0 + sl) * FXG_QS_STRIP, m, lds.data());
                        }
                    since += trip_reads;
                }
            };
            if (piece) {
                if (g == 0) rows(ntrip * pR, a.n);
                const u64 RL = (u64)pR * a.stride;
                const bool grid = ((u64)a.nwg * RL) % 128u == 0u;                // as the kernel: the workgroups' trips keep their distance to the line grid
                for (u64 T = g; T < ntrip; T += a.nwg) {
                    if (since + pR + 8u > wrap) flush();
                    const u64 c0 = fxg_stats_piece_cut(T, ntrip, RL, grid), c1 = fxg_stats_piece_cut(T + 1, ntrip, RL, grid);
                    if ((c1 - c0) / 16u > FXG_QS_TBLOCK) return FXG_E_DEVICE;    // more pieces than the workgroup has lanes
                    for (u64 at = c0; at < c1; at += 16u) {
                        FxgStripRow row;
                        row.nb = FXG_QS_STRIP; row.vb = fxg_ld16(a.bases + at); row.vq = fxg_ld16(a.qual + at);
                        if (a.fixed_len & 1u) { FxgPieceLane<true> pc; fxg_stats_piece_lane(a.fixed_len, (u32)(at % a.fixed_len), pc); fxg_stats_accumulate_piece(a, row, pc, lds.data()); }
                        else { FxgPieceLane<false> pc; fxg_stats_piece_lane(a.fixed_len, (u32)(at % a.fixed_len), pc); fxg_stats_accumulate_piece(a, row, pc, lds.data()); }
                    }
                    since += pR + 8u; ++g_qs_piece_trips;
                    if (grid && (c0 & 127u) == 0u && c0 != T * RL) ++g_qs_piece_moved;
                }
            } else {
                u64 lo, hi;
                fxg_stats_slice(a, g, &lo, &hi);
                rows(lo, hi);
            }
            flush();
        }
        for (u32 e = 0; e < FXG_QS_PART_WORDS; ++e) fxg_stats_fold(a, e);
    }
    return FXG_OK;
}


// ------------------------------------------------------------------------------------------------
// FASTA/FASTQ text on the "device": the per-thread bodies of fxg_text.h, one lane after the other; the wave-level reductions and
// the scans between the kernels (fxg_engine.hip) are serial sums here.  Same contracts as fxg_fastq_index / _pack / _format / fxg_fasta_weights.
// ------------------------------------------------------------------------------------------------
extern "C" int fxg_emu_fastq_index(FxgTextState *st, const uint8_t *text, uint64_t text_len, int at_eof, int lpr, uint32_t *d_line, uint64_t cap_lines,
                                   uint16_t *d_len, uint8_t *d_flags, fxg_text_info *info)
{
    if (!text || !d_line || !d_len || !d_flags || !info || (lpr != 4 && lpr != 2)) return FXG_E_INVALID;
    memset(info, 0, sizeof *info);
    info->first_bad = 0xFFFFFFFFu;
    if (text_len == 0) return FXG_OK;
    if (text_len > 0xFFFFFFF0ull) return FXG_E_INVALID;
    u32 *ls = d_line, *le = d_line + cap_lines;
    memset(st, 0, sizeof *st);
    st->min_len = 0xFFFFFFFFu; st->first_bad = 0xFFFFFFFFu;
    const u64 nseg = (text_len + FXG_TEXT_SEG - 1) / FXG_TEXT_SEG;
    u64 j = 0;
    ls[0] = 0u;
    for (u64 seg = 0; seg < nseg; ++seg)
        for (u32 t = 0; t < FXG_BLOCK; ++t) {
            const u64 off = seg * FXG_TEXT_SEG + (u64)t * 16;
            u32 cr = 0;
            const u32 m = fxg_text_nl_mask(text, off, text_len, &cr);
            if (cr & 0xFFFFu) st->has_cr = 1u;
            if (cr >> 16) st->irregular |= FXG_TEXT_IRR_NUL;
            fxg_text_nl_store(m, off, j, ls, le, cap_lines);
            j += (u64)__builtin_popcount(m);
        }
    const u64 lines = j;
    info->lines = lines;
    u64 n = lines / (u64)lpr;
    if ((u64)lpr * n + 1 > cap_lines) n = (cap_lines - 1) / (u64)lpr;
    info->records = n;
    if (n == 0) { if (at_eof && lines % (u64)lpr != 0) info->irregular |= FXG_TEXT_IRR_TAIL; return FXG_OK; }
    for (u64 r = 0; r < n; ++r) {
        u32 sl = 0, fl = 0;
        const u32 irr = lpr == 4 ? fxg_text_record<4>(text, ls, le, r, st->has_cr, d_len, d_flags, &sl, &fl) : fxg_text_record<2>(text, ls, le, r, st->has_cr, d_len, d_flags, &sl, &fl);
        if (fl) st->n_numeric++;
        if (irr) { st->irregular |= irr; if ((u32)r < st->first_bad) st->first_bad = (u32)r; }
        else { if (sl > st->max_len) st->max_len = sl; if (sl < st->min_len) st->min_len = sl; }
    }
    info->consumed = ls[(u64)lpr * n];
    info->max_len = st->max_len; info->min_len = st->min_len; info->irregular = st->irregular; info->first_bad = st->first_bad;
    info->numeric_records = st->n_numeric; info->has_cr = st->has_cr;
    if (at_eof && (lines % (u64)lpr != 0 || info->consumed != text_len)) info->irregular |= FXG_TEXT_IRR_TAIL;
    return FXG_OK;
}

extern "C" int fxg_emu_fastq_pack(const uint8_t *text, uint64_t text_len, int lpr, const uint32_t *d_line, uint64_t cap_lines, const uint8_t *flags, uint64_t n,
                                  uint32_t stride, int qoffset, uint8_t *bases, uint8_t *qual, uint32_t *irregular)
{
    if (!text || !d_line || !flags || !bases || !irregular || stride == 0 || (lpr != 4 && lpr != 2)) return FXG_E_INVALID;
    *irregular = 0;
    if (n == 0) return FXG_OK;
    if (lpr == 2 && qual) return FXG_E_INVALID;
    const u32 *ls = d_line, *le = d_line + cap_lines;
    const u64 nchunks = (n * (u64)stride + 15) >> 4;
    u32 badb = 0, badq = 0;
    for (u64 c = 0; c < nchunks; ++c) {
        badb |= lpr == 4 ? fxg_text_pack_chunk<false, 4>(text, text_len, ls, le, flags, n, stride, qoffset, bases, c) : fxg_text_pack_chunk<false, 2>(text, text_len, ls, le, flags, n, stride, qoffset, bases, c);
        if (qual) badq |= fxg_text_pack_chunk<true, 4>(text, text_len, ls, le, flags, n, stride, qoffset, qual, c);
    }
    if (qual) for (u64 r = 0; r < n; ++r) fxg_text_numeric_row(text, ls, le, flags, stride, qual, r);
    if (badb) *irregular |= FXG_TEXT_IRR_BASE;
    if (badq) *irregular |= FXG_TEXT_IRR_QUAL;
    return FXG_OK;
}

extern "C" int fxg_emu_fastq_format(const uint8_t *text, int lpr, const uint32_t *d_line, uint64_t cap_lines, const uint8_t *flags, uint64_t n, const uint32_t *res,
                                    uint32_t fwd_start, int reverse, const uint8_t *pk_bases, const uint8_t *pk_qual, const uint64_t *pk_off, const uint8_t *rows_qual,
                                    uint32_t stride, int qoffset, int out_fasta, uint8_t *out, uint64_t *out_bytes)
{
    if (!text || !d_line || !flags || !res || !out || !out_bytes || (lpr != 4 && lpr != 2)) return FXG_E_INVALID;
    *out_bytes = 0;
    if (n == 0) return FXG_OK;
    const bool fastq_out = lpr == 4 && !out_fasta;
    if (pk_bases && (!pk_off || (fastq_out && !pk_qual))) return FXG_E_INVALID;
    if (fastq_out && !rows_qual) return FXG_E_INVALID;
    std::vector<u64> item(n);
    FxgFormatArgs a;
    a.text = text; a.ls = d_line; a.le = d_line + cap_lines; a.res = res; a.flags = flags; a.item_scan = item.data(); a.n = n;
    a.fwd_start = fwd_start; a.rev = reverse ? 1u : 0u; a.pk_bases = pk_bases; a.pk_qual = pk_qual; a.pk_off = (const u64 *)pk_off;
    a.rows_qual = rows_qual; a.stride = stride; a.qoffset = qoffset; a.out_fasta = out_fasta ? 1u : 0u; a.out = out;
    u64 run = 0;
    for (u64 r = 0; r < n; ++r) {                               // sizes, then the exclusive scan (offset in the low 40 bits, rank above)
        const u64 v = lpr == 4 ? fxg_text_size_record<4>(a, r) : fxg_text_size_record<2>(a, r);
        item[r] = run;
        run += v;
    }
    for (u64 r = 0; r < n; ++r)
        for (u32 l = 0; l < 16; ++l) { if (lpr == 4) fxg_text_format_record<4>(a, r, l); else fxg_text_format_record<2>(a, r, l); }
    *out_bytes = run & ((1ull << 40) - 1ull);
    return FXG_OK;
}

extern "C" int fxg_emu_fasta_weights(const uint8_t *text, const uint32_t *d_line, uint64_t cap_lines, uint64_t n, const uint32_t *res, uint64_t weighted[8])
{
    if (!text || !d_line || !res || !weighted) return FXG_E_INVALID;
    memset(weighted, 0, 8 * sizeof(uint64_t));
    for (u64 r = 0; r < n; ++r) {
        u64 v[7] = {0, 0, 0, 0, 0, 0, 0};
        fxg_text_weights_record(text, d_line, d_line + cap_lines, res, r, v);
        for (int i = 0; i < 7; ++i) weighted[i] += v[i];
    }
    return FXG_OK;
}
#endif      // FXG_EMU_TU == 0
