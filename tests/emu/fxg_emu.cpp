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

template <int AMAX, bool REV, int MODE = 0>
static int emu_run(const FxgPlan &pl, uint64_t *counters, char *err, size_t cap)
{
    const FxgKArgs &a = pl.ka;
    const u32 T = a.tile_reads, stride = a.stride, NT = FXG_TBLOCK;
    std::vector<unsigned char> lds(pl.lds + 64, 0);
    unsigned char *smem = lds.data();
    const FxgLds L = pl.group_a ? fxg_lds_layout(T, stride, pl.use_q, pl.clip) : fxg_lds_layout(T, stride, MODE == 3, MODE == 4);
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
                if constexpr (AMAX != 0) fxg_phase_stage_bases(a, tb, tbytes, sb, tid, NT);
            }
            for (u32 tid = 0; tid < nreads; ++tid) {
                fxg_decide_a<AMAX>(a, bm_g, bm_l, sb, r0, tid, &keep[tid], &olen[tid]);
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
            for (u32 tid = 0; tid < NT; ++tid) fxg_phase_stage_bases(a, tb, tbytes, sb, tid, NT);
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

extern "C" int fxg_emu_run_pipeline(const fxg_batch *in, const fxg_params *p, const fxg_out *out, char *err, size_t cap)
{
    FxgPlan pl;
    const int rc = fxg_make_plan(in, p, out, &pl, err, cap);
    if (rc != FXG_OK) return rc;
    if (in->n == 0) return FXG_OK;
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

extern "C" unsigned fxg_emu_tile_reads(unsigned stride, int clip) { return fxg_pick_tile(stride, clip != 0); }
