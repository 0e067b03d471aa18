// fxg_plan.h -- host-side validation of a request and folding of the tool parameters into launch
// arguments.  Shared by the engine (fxg_engine.hip) and by the CPU emulator used in tests/emu.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fxg_kernels.h"
#include "fxg_rows.h"

struct FxgPlan {
    FxgKArgs ka;
    bool group_a;   // [CLIP][QTRIM][QFILTER] chain (else [REVCOMP][FTRIM*] unless mode3/mode4)
    bool clip, use_q, rev;
    bool mask, artifacts;
    int  amax;      // adapter bucket of the clip kernel instance (0 = no clip)
    u32  lds;       // dynamic LDS bytes per workgroup
    int  rows_nw;   // != 0: the quality stages run as fxg_kernel_rows<rows_nw, rows_h> (fxg_rows.h): dwords of one lane's piece of a row
    int  rows_h;    // lanes per read of that instance (1: rows up to 152 bytes, 2: up to 304)
    int  rows_r;    // reads per lane (fxg_kernel_rows_multi: rows of 28..79 bytes; else 1)
    u32  block;     // threads per workgroup of the instance (FxgTileBlock)
    u64  ck_per_wg; // floats of checkpoint scratch per workgroup (fxg_clip_two_pass_k), 0 = the instance runs its one-pass form
};

static inline int fxg_clampi(long long v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : (int)v); }

static inline u32 fxg_pick_tile(u32 stride, bool clip, u32 block = FXG_TBLOCK)
{
    // Rows one workgroup covers per tile.  Streaming kernels: about 20 KB (128 reads of 150 bases) -- with the central scanner the
    // per-tile cost is small enough that the shorter pipeline step wins (cfg2: 4.13 ms at 128 reads, 4.19-4.31 at 256, 5.9 at 64;
    // profiles/r02/variants_*.txt).  Kernels that stage the tile's bases in LDS (the clipper): up to 60 KB so that two or
    // more workgroups share a CU.
    const u64 budget = clip ? 60ull * 1024 : 20ull * 1024;
    u32 T = block;                          // one thread decides one read
    const char *env = getenv("FXG_TILE");   // tuning knob: 1..block, power of two
    if (env && atoi(env) >= 1 && atoi(env) <= (int)block && (atoi(env) & (atoi(env) - 1)) == 0) return (u32)atoi(env);
    while (T > 1 && (u64)T * stride > budget) T >>= 1;
    return T;
}

static inline FxgLds fxg_plan_layout(const FxgPlan *pl)
{
    const FxgKArgs &ka = pl->ka;
    return pl->group_a ? fxg_lds_layout(ka.tile_reads, ka.stride, fxg_bitmap_count(ka, pl->use_q, pl->clip), (pl->clip && !ka.clip_global) ? ka.clip_stride : 0u, pl->clip ? ka.depth : 2u, pl->clip,
                                        (pl->clip && fxg_clip_uses_ptab(pl->amax)) ? fxg_ptab_bytes(ka.clip_ptab_rows, ka.clip_ptab_stride) : 0u)
         : pl->mask ? fxg_lds_layout(ka.tile_reads, ka.stride, 2u, 0u)
         : pl->artifacts ? fxg_lds_layout(ka.tile_reads, ka.stride, 4u, 0u) : fxg_lds_layout(ka.tile_reads, ka.stride, 0u, 0u);
}
static inline u32 fxg_plan_lds(const FxgPlan *pl) { return fxg_plan_layout(pl).total; }

#define FXG_PLAN_FAIL(...) do { snprintf(err, cap, __VA_ARGS__); return FXG_E_INVALID; } while (0)

// clip_stride: row stride of the array the clipper's DP reads when it is not the batch itself (clip history), else 0
static inline int fxg_make_plan(const fxg_batch *in, const fxg_params *p, const fxg_out *out, FxgPlan *pl, char *err, size_t cap, u32 clip_stride = 0)
{
    const u32 st = p->stages;
    const bool ga = (st & (FXG_STAGE_CLIP | FXG_STAGE_QTRIM | FXG_STAGE_QFILTER)) != 0;
    const bool gb = (st & (FXG_STAGE_REVCOMP | FXG_STAGE_FTRIM | FXG_STAGE_FTRIM_END)) != 0;
    const bool gm = (st & FXG_STAGE_MASK) != 0, gf = (st & (FXG_STAGE_ARTIFACTS | FXG_STAGE_NFILTER)) != 0;
    if ((int)ga + (int)gb + (int)gm + (int)gf != 1 || (st & (FXG_STAGE_ARTIFACTS | FXG_STAGE_NFILTER)) == (FXG_STAGE_ARTIFACTS | FXG_STAGE_NFILTER))
        FXG_PLAN_FAIL("unsupported stage chain 0x%x: use [CLIP][QTRIM][QFILTER], [REVCOMP][FTRIM|FTRIM_END], [MASK], [ARTIFACTS] or [NFILTER]", st);
    if (gm && !in->qual) FXG_PLAN_FAIL("fastq_masker needs qualities");
    if ((st & FXG_STAGE_FTRIM) && (st & FXG_STAGE_FTRIM_END))
        FXG_PLAN_FAIL("[-t], [-f] and [-l] options can not be used together");   // fastx_trimmer.c:112-113
    if (!in->bases || !out->res) FXG_PLAN_FAIL("bases and res are mandatory");
    if ((st & (FXG_STAGE_QTRIM | FXG_STAGE_QFILTER)) && !in->qual) FXG_PLAN_FAIL("quality stages need qual");
    if (in->stride == 0 || in->stride > FXG_MAX_READ_LEN || (!in->len && (in->fixed_len == 0 || in->fixed_len > in->stride)))
        FXG_PLAN_FAIL("bad stride/fixed_len (%u/%u)", in->stride, in->fixed_len);
    if ((((uintptr_t)in->bases | (uintptr_t)in->qual | (uintptr_t)out->out_bases | (uintptr_t)out->out_qual | (uintptr_t)out->res) & 15u) != 0)
        FXG_PLAN_FAIL("bases/qual/res/out_bases/out_qual must be 16-byte aligned");
    if (out->out_bases && in->qual && !out->out_qual) FXG_PLAN_FAIL("out_qual missing");

    FxgKArgs &ka = pl->ka;
    memset(&ka, 0, sizeof ka);
    ka.bases = in->bases; ka.qual = in->qual; ka.len = in->len;
    ka.clip_src = in->bases; ka.clip_stride = clip_stride ? clip_stride : in->stride; ka.clip_total = in->n * (u64)ka.clip_stride; ka.wlen = nullptr;
    ka.n = in->n; ka.total_bytes = in->n * (u64)in->stride;
    ka.fixed_len = in->fixed_len; ka.stride = in->stride;
    ka.res = out->res; ka.out_bases = out->out_bases; ka.out_qual = out->out_qual;
    ka.out_len = out->out_len; ka.kept_index = out->kept_index; ka.out_off = (u64 *)out->out_off;
    ka.compact = out->out_bases ? 1u : 0u;
    ka.stages = st;
    ka.tq = (u32)fxg_clampi((long long)p->qt_threshold + p->qoffset, 0, 128);
    ka.fq = (u32)fxg_clampi((long long)(gm ? p->mask_min_quality : p->qf_min_quality) + p->qoffset, 0, 128);
    ka.mask_char = p->mask_char & 0xFFu;
    ka.nf_keep_n = p->nf_keep_n;
    ka.qt_min_len = p->qt_min_len;
    ka.qf_keep_pct = 100 - p->qf_min_percent;
    ka.qf_drop_all = (p->qf_min_percent == 0 && p->qf_min_quality > 93) ? 1u : 0u;   // quirk F2
    ka.clip_min_len = p->clip_min_len; ka.clip_keep_delta = p->clip_keep_delta;
    ka.clip_min_adapter_len = p->clip_min_adapter_len; ka.clip_flags = p->clip_flags;
    ka.ft_first = p->ft_first; ka.ft_last = p->ft_last; ka.ft_trim_end = p->ft_trim_end; ka.ft_min_len = p->ft_min_len;
    memcpy(ka.adapter, p->adapter, sizeof ka.adapter);
    ka.adapter[sizeof ka.adapter - 1] = 0;
    ka.alen = (int)strlen(ka.adapter);
    ka.adapter_has_n = strchr(ka.adapter, 'N') != nullptr;
    ka.clip_ptab_rows = fxg_ptab_rows(ka.adapter, ka.alen);
    fxg_ptab_row_map(ka.adapter, ka.alen, ka.clip_ptab_row);

    pl->group_a = ga;
    pl->mask = gm; pl->artifacts = gf;
    pl->clip = (st & FXG_STAGE_CLIP) != 0;
    pl->use_q = (st & (FXG_STAGE_QTRIM | FXG_STAGE_QFILTER)) != 0;
    pl->rev = (st & FXG_STAGE_REVCOMP) != 0;
    if (pl->clip && (ka.alen < 1 || ka.alen > FXG_MAX_ADAPTER)) FXG_PLAN_FAIL("adapter length %d out of range", ka.alen);
    if (pl->clip && ka.clip_stride > 65000u) FXG_PLAN_FAIL("clip: reads longer than 65000 are not supported");
    if ((st & FXG_STAGE_FTRIM) && p->ft_first < 1) FXG_PLAN_FAIL("-f must be >= 1");
    pl->amax = !pl->clip ? 0 : ka.alen <= 16 ? 16 : ka.alen <= 32 ? 32 : ka.alen <= 64 ? 64 : 100;
    // Packed path summary (one u32 per cell); buckets are fine-grained because every padded column costs a full cell.
    //   up to 16 columns, no 'N' in the adapter: two passes in registers (fxg_clip_two_pass), reads of any length (the start of a path
    //   is recorded relative to the second pass' first row);
    //   everything else: the form with ONE start field (fxg_clip_row_k) -- 17..99 columns, adapters that contain 'N' (instances
    //   -(300 + columns), two more instructions per cell).
    // That form runs two passes with its checkpoints in global scratch once the read is long enough to pay for the second one (one pass
    // ~15 VALU instructions per cell; two: ~6.5 + the <= SPAN + clip_ck_rows rows of the second pass), and always beyond 255 bases.
#ifdef FXG_CLIP_ONE_PASS
    const bool reg_any_len = false;              // (ablation build: the register form in one pass records absolute rows, 8 bits)
#else
    const bool reg_any_len = true;
#endif
    // (an 'N' in the adapter is one more column pattern of the pair table: the register form serves it like any other adapter)
    const bool kform = pl->clip && (ka.alen > 16 || (ka.adapter_has_n && !fxg_clip_uses_ptab(-16)) || (ka.clip_stride > 255u && !reg_any_len));
    // Two passes pay while the summary rows of the second one (at most SPAN = A + (A + 1) / 5 rows up to the best row, plus the scores re-run from the last
    // checkpoint: < clip_ck_rows) are fewer than the read's rows: a score row is ~5 VALU instructions per cell with the pair table, a summary row ~10, and the
    // one-pass form tracks the best cell in every row (~15).  (Round 5: from 20 + 2 A bases on -- its score rows cost 7 per cell and a 44-column adapter on
    // 100-base reads ran in one pass: 17.1 ms per 10 M reads, 12.6 with the table, two passes: profiles/r06/.)
    ka.clip_ck_rows = (ka.clip_stride + 7u) / 8u < 4u ? 4u : (ka.clip_stride + 7u) / 8u;      // (rows - 1) / clip_ck_rows <= FXG_CK_SLOTS
    const u32 span_k = (u32)ka.alen + ((u32)ka.alen + 1u) / 5u;
    // (a tenth of the read as margin: with 65..73 columns on 100-base reads -- 91..99 of 100 rows -- the two passes were 7-13 % behind the one)
    const bool two_pass_k = kform && (span_k + ka.clip_ck_rows + ka.clip_stride / 10u <= ka.clip_stride || ka.clip_stride > 255u) && !getenv("FXG_CLIP_K_ONE_PASS");
    pl->ck_per_wg = 0;
    ka.clip_ck = nullptr;
    if (pl->clip && (ka.clip_stride <= 255u || two_pass_k || (!kform && reg_any_len)) && !getenv("FXG_NO_PACKED_CLIP")) {
        // 36: the 33/34-base TruSeq adapters; 56 and 80 (round 5): 49..56 columns no longer pay for 64 (17.4 -> 27.7 ms between 48 and 49 bases,
        // profiles/r04/p_clip_waves_by_adapter_len.txt) and 65..80 no longer for 100
        // 44, 52, 60, 72, 88 (round 6): a bucket every 4 columns to 64 and every 8 to 88, so that no adapter pays for more than 8 % .. 12 % of padding columns
        static const int pk[] = {4, 8, 9, 10, 11, 12, 13, 14, 15, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64, 72, 80, 88, 100};
        static const int pn[] = {16, 24, 36, 48, 56, 64, 80, 100};
        int b = 100;
        const bool n_inst = ka.adapter_has_n && kform && !fxg_clip_uses_ptab(-20);      // (builds without the pair table: the instances with per-column neutral selects)
        if (n_inst) { for (unsigned i = 0; i < sizeof pn / sizeof pn[0]; ++i) if (ka.alen <= pn[i]) { b = pn[i]; break; } }
        else { for (unsigned i = 0; i < sizeof pk / sizeof pk[0]; ++i) if (ka.alen <= pk[i]) { b = pk[i]; break; } }
        if (b < 16 && kform) b = 16;
        pl->amax = n_inst ? -(300 + b) : (b == 16 && kform) ? -216 : -b;
        if (two_pass_k) pl->ck_per_wg = (u64)FXG_CK_SLOTS * (u64)b * FXG_TBLOCK;
        // the pair table of the instance (fxg_kernels.h: fxg_clip_ptab_build): as many columns as the bucket, rounded up to whole 16-byte blocks
        if (fxg_clip_uses_ptab(pl->amax)) {
            const bool kf = fxg_clip_kform(pl->amax);
            ka.clip_ptab_cols = kf ? ((u32)b + 3u) & ~3u : 16u;
            ka.clip_ptab_stride = fxg_ptab_stride(ka.clip_ptab_cols);
            ka.clip_ptab_dia1 = kf ? FXG_K_DIA1 : FXG_PK_DIA1;
            if (kf && ka.clip_ptab_rows > FXG_PTAB_MAX_ROWS_K) {      // an adapter of more than six distinct bytes and more than 16 columns: the general form
                pl->amax = ka.alen <= 32 ? 32 : ka.alen <= 64 ? 64 : 100;
                pl->ck_per_wg = 0;
            }
        }
    }
    // quality trim / filter with compaction over rows of 80..152 bytes: one lane per read, 64 reads per tile (fxg_rows.h).  Shorter
    // rows make its 64-read tiles too small (the tile kernel is 10-15 % ahead at 36-50 bases, ahead at 72, level at 76, 5-10 % behind from 88 on:
    // profiles/r02/af_rows_vs_tiles_by_length.txt)
    // Rows of 153..304 bytes: the same kernel with TWO lanes per read, 32 reads per tile.  Measured (profiles/r03/s_rows_vs_tiles_long.txt):
    // 2 % ahead of the tile kernel at 200 bases, 4 % / 2 % BEHIND at 250 / 300 (a 250-byte row fills 82 % of its two register pieces, and
    // the tile kernel's per-tile costs shrink with the row length) -- so it is the default up to 208 bytes only; FXG_ROWS=2 selects it
    // wherever it exists (tests, measurements), FXG_ROWS=0 never.
    // Rows of 28..79 bytes (round 6): the same kernel with several reads per lane (fxg_kernel_rows_multi: 4 x 40, 3 x 56, 2 x 80 bytes), so that a tile is ~10 KB
    // again -- 36-base reads 0.44 -> ... of the HBM peak (profiles/r06/rows_multi_short_reads.txt).
    pl->rows_nw = 0; pl->rows_h = 1; pl->rows_r = 1;
    {
        const int want = getenv("FXG_ROWS") ? atoi(getenv("FXG_ROWS")) : 1;
        const u32 top = want >= 2 ? 304u : 208u;
        if (ga && !pl->clip && ka.compact && in->stride >= 80u && in->stride <= top && want != 0) {
            pl->rows_h = in->stride <= 152u ? 1 : 2;
            pl->rows_nw = in->stride <= 104u * (u32)pl->rows_h ? 26 : 38;
        } else if (ga && !pl->clip && ka.compact && in->stride >= 28u && in->stride < 80u && want != 0) {
            if (in->stride <= 40u) { pl->rows_nw = 10; pl->rows_r = 4; }
            else if (in->stride <= 56u) { pl->rows_nw = 14; pl->rows_r = 3; }
            else { pl->rows_nw = 20; pl->rows_r = 2; }
        }
    }
    pl->block = pl->rows_nw ? 64u : (ga && pl->amax < 0 && pl->amax >= -16) ? (u32)FXG_CLIP_TBLOCK : (u32)FXG_TBLOCK;     // FxgTileBlock
    // The two-pass clip forms can run their DP straight over the batch in global memory (fxg_clip_two_pass<.., GL>, fxg_clip_two_pass_k<.., GL>): no tile of
    // bases in LDS, so the tile stays at one read per thread whatever the read length.  Staged, a 256-thread workgroup holds 128 reads at
    // 250-300 bases and 32 at 1 000 (the other lanes idle), and from about 180 bases on a CU holds two workgroups instead of three or four.
    // The staged form is the faster one while it keeps three workgroups per CU (the window costs the row loop a branch and a shift:
    // 100 bases 5.56 against 6.13 ms per 20 M reads, 176 bases 4.37 / 4.81 per 10 M), the other one from there on (188 bases 5.81 / 5.14,
    // 300 bases 6.26 / 4.66 per 6 M, 1 000 bases 13.1 / 5.95 per 2 M: profiles/r04/ae_clip_global_vs_staged*.txt).  Rows must start on dword
    // boundaries; runs with clip history (ragged input of the tools) keep the staged form.  FXG_CLIP_GLOBAL=0 / 1 overrides (tests run both).
    ka.clip_global = 0u;
    u32 T = pl->rows_nw ? 64u * (u32)pl->rows_r / (u32)pl->rows_h : fxg_pick_tile(pl->clip ? ka.clip_stride : in->stride, pl->clip, pl->block);
#ifndef FXG_CLIP_ONE_PASS
    if (pl->clip && pl->amax < 0 && (pl->amax >= -16 || pl->ck_per_wg != 0) && clip_stride == 0u && (ka.clip_stride & 3u) == 0u && ((uintptr_t)ka.clip_src & 3u) == 0u) {      // (clip_stride != 0: a run with clip history, whose rows are settled after the plan)
        ka.tile_reads = T; ka.depth = 2u;
        const bool cramped = T < pl->block || (156u * 1024u) / fxg_plan_lds(pl) < 3u;
        const char *e = getenv("FXG_CLIP_GLOBAL");
        ka.clip_global = (e ? atoi(e) != 0 : cramped) ? 1u : 0u;
        if (ka.clip_global) T = pl->block;
    }
#endif
    const u64 ntiles = (in->n + T - 1) / T;
    if (ntiles > 0x7FFFFFFFull || in->n > 0xFFFFFFFFull) FXG_PLAN_FAIL("batch too large (%llu reads): split it", (unsigned long long)in->n);
    ka.tile_reads = T; ka.ntiles = (u32)ntiles;
    // Reverse complement of a fixed-length batch: the bytes of an output chunk of read k come from the window that starts at  k (stride + kept length) +
    // (length - start)  mod 4 (tiles of a multiple of four reads, every read kept alike: fxg_decide_b), so either some residue class of k has its windows on dword
    // boundaries or none has -- and only then do dword-aligned loads pay (fxg_ld16_dw, fxg_kernel_tiles<0, 5>).  FXG_REV_DW=0/1 forces it (measurement knob).
    ka.rev_dw = 0u;
    if ((st & FXG_STAGE_REVCOMP) && !in->len && (T & 3u) == 0u) {
        u32 start = 0, cur = in->fixed_len;
        if (st & FXG_STAGE_FTRIM) {
            if (p->ft_last != 0 && (u32)p->ft_last < cur) cur = (u32)p->ft_last;
            if (p->ft_first != 1 && cur >= (u32)p->ft_first) { start = (u32)p->ft_first - 1u; cur -= start; }
        }
        if ((st & FXG_STAGE_FTRIM_END) && cur > p->ft_trim_end) cur -= p->ft_trim_end;
        const u32 A = (in->stride + cur) & 3u, C = (in->fixed_len - start) & 3u;
        ka.rev_dw = ((A == 0u && C != 0u) || (A == 2u && (C & 1u))) ? 1u : 0u;
    }
    if (const char *e = getenv("FXG_REV_DW")) ka.rev_dw = atoi(e) != 0 ? 1u : 0u;
    // clip instances: the write-out runs two steps behind the decision (three slots) unless the extra slot costs a workgroup per CU
    ka.depth = 2u;
    if (pl->clip && FXG_CLIP_DEPTH >= 3u) {
        // (156 KB: three workgroups of 53.9 KB -- 161.6 KB, nominally inside the CU's 160 KB = 163 840 bytes -- ran as TWO per CU: cfg5 9.7 against 7.7 ms, profiles/r04/m_clip_depth.txt)
        const u32 cu_lds = 156u * 1024u, regs_wg = pl->block == 64u ? 16u : 4u;      // workgroups per CU the registers allow (128 VGPRs)
        const u32 lds2 = fxg_plan_lds(pl);
        const u32 wg2 = cu_lds / lds2 < regs_wg ? cu_lds / lds2 : regs_wg;
        u32 want = FXG_CLIP_DEPTH;
        { const char *e = getenv("FXG_CLIP_DEPTH_RT"); if (e && atoi(e) >= 2 && atoi(e) <= 4) want = (u32)atoi(e); }      // tuning knob
        for (u32 d = want; d >= 3u; --d) {           // the deepest pipeline that does not cost a workgroup per CU
            ka.depth = d;
            const u32 ldsd = fxg_plan_lds(pl);
            if ((cu_lds / ldsd < regs_wg ? cu_lds / ldsd : regs_wg) >= wg2) break;
            ka.depth = 2u;
        }
        if (want == 2u) ka.depth = 2u;
    }
    pl->lds = pl->rows_nw ? fxg_rows_lds(ka.stride, (u32)pl->rows_h, (u32)pl->rows_r) : fxg_plan_lds(pl);
    return FXG_OK;
}
