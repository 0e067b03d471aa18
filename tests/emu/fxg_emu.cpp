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

// fxg_kernel_rows: the lane holds its read's quality row in NW registers (bytes past the row: whatever follows in the staging buffer)
template <int NW>
static void emu_rows_decide_nw(const FxgKArgs &a, u32 read, u32 *keep, u32 *olen)
{
    u32 q[NW];
    for (int k = 0; k < NW; ++k) {
        u32 w = 0;
        for (int i = 0; i < 4; ++i) {
            const u64 at = (u64)read * a.stride + 4u * (u32)k + (u32)i;
            w |= (u32)(at < a.total_bytes ? a.qual[at] : 0xA5u) << (8 * i);
        }
        q[k] = w;
    }
    fxg_rows_decide<NW>(a, q, read, keep, olen);
}
static void emu_rows_decide(int nw, const FxgKArgs &a, u32 read, u32 *keep, u32 *olen)
{
    if (nw == 26) emu_rows_decide_nw<26>(a, read, keep, olen);
    else emu_rows_decide_nw<38>(a, read, keep, olen);
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
    for (u32 tile = 0; tile < a.ntiles; ++tile) {
        const u32 r0 = tile * T;
        const u64 left = a.n - (u64)r0;
        const u32 nreads = left < (u64)T ? (u32)left : T;
        const u64 tb = (u64)r0 * stride;
        const u32 tbytes = nreads * stride;
        if (pl.group_a) {
            for (u32 tid = 0; tid < NT; ++tid) {
                if (pl.use_q) fxg_phase_bitmaps(a, tb, tbytes, bm_g, bm_l, tid, NT);
                if constexpr (AMAX != 0) fxg_phase_stage_bases(a.clip_src, a.clip_total, (u64)r0 * a.clip_stride, nreads * a.clip_stride, sb, tid, NT);
            }
            for (u32 tid = 0; tid < nreads; ++tid) {
                if (AMAX == 0 && pl.rows_nw) emu_rows_decide(pl.rows_nw, a, r0 + tid, &keep[tid], &olen[tid]);   // what a lane of fxg_kernel_rows does
                else fxg_decide_a<AMAX>(a, bm_g, bm_l, sb, r0, tid, &keep[tid], &olen[tid]);
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
            for (u32 tid = 0; tid < NT; ++tid) fxg_phase_stage_bases(a.bases, a.total_bytes, tb, tbytes, sb, tid, NT);
            for (u32 tid = 0; tid < nreads; ++tid) { fxg_decide_census(a, sb + tid * stride, r0, tid, &keep[tid], &olen[tid], &bad); anchor[tid] = tid * stride; }
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

extern "C" int fxg_emu_run_pipeline_hist(const fxg_batch *in, const fxg_params *p, const fxg_out *out, char *err, size_t cap, fxg_emu_hist *hs)
{
    FxgPlan pl;
    const bool hist = hs && (p->stages & FXG_STAGE_CLIP) && in->n != 0;
    const u32 estride = hist && hs->wcap > in->stride ? hs->wcap : in->stride;
    const int rc = fxg_make_plan(in, p, out, &pl, err, cap, hist ? estride : 0u);
    if (rc != FXG_OK) return rc;
    if (in->n == 0) return FXG_OK;
    if (hist) {
        int use = 0;
        emu_hist_prepass(hs, in, pl.ka.tile_reads, estride, &pl.ka, &use);
        if (!use) { pl.ka.clip_src = in->bases; pl.ka.clip_stride = in->stride; pl.ka.clip_total = in->n * (u64)in->stride; pl.ka.wlen = nullptr; pl.lds = fxg_plan_lds(&pl); }
    }
    uint64_t *ctr = out->counters;
    if (pl.group_a) {
        switch (pl.amax) {
        case 0: return emu_run<0, false>(pl, ctr, err, cap);
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
        case -20: return emu_run<-20, false>(pl, ctr, err, cap);
        case -24: return emu_run<-24, false>(pl, ctr, err, cap);
        case -28: return emu_run<-28, false>(pl, ctr, err, cap);
        case -32: return emu_run<-32, false>(pl, ctr, err, cap);
        case 16: return emu_run<16, false>(pl, ctr, err, cap);
        case 32: return emu_run<32, false>(pl, ctr, err, cap);
        case 64: return emu_run<64, false>(pl, ctr, err, cap);
        default: return emu_run<100, false>(pl, ctr, err, cap);
        }
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

// fastx_quality_stats: the per-thread bodies of fxg_kernel_quality_stats / _fold, one "workgroup" after the other
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
        std::fill(partial.begin(), partial.end(), 0u);
        for (u32 g = 0; g < a.nwg; ++g) {
            std::fill(lds.begin(), lds.end(), 0u);
            u32 *part = a.partial + (u64)g * FXG_QS_PART_WORDS;
            u64 lo, hi;
            fxg_stats_slice(a, g, &lo, &hi);
            const u64 nitems = (hi - lo) * FXG_QS_WAVES;
            u32 since = 0;
            for (u64 g0 = 0; g0 < nitems; g0 += (u64)FXG_QS_TBLOCK * FXG_QS_UNROLL) {
                if (since + trip_reads > (getenv("FXG_EMU_QS_FLUSH") ? 600u : 65535u)) {      // the env knob exercises the wrap guard on small inputs
                    for (u32 t = 0; t < FXG_QS_TBLOCK; ++t) fxg_stats_flush(lds.data(), part, t, FXG_QS_TBLOCK);
                    since = 0;
                }
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
            for (u32 t = 0; t < FXG_QS_TBLOCK; ++t) fxg_stats_flush(lds.data(), part, t, FXG_QS_TBLOCK);
        }
        for (u32 e = 0; e < FXG_QS_PART_WORDS; ++e) fxg_stats_fold(a, e);
    }
    return FXG_OK;
}
