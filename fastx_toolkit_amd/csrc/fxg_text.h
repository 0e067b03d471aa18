// fxg_text.h -- FASTA/FASTQ text <-> Structure-of-Arrays on the device.
//
// Replaces the reference's record reader (src/libfastx/fastx.c:314-404) and writer (:440-473): a block of text already
// resident in HBM is indexed (newline positions by SWAR compare/popcount + prefix sums), checked record by record, packed into
// the engine's SoA batch, and after the pipeline the kept records are formatted back to text on the device.  Handled here:
//   * FASTQ (four lines per record, '@') and FASTA (two lines, '>'), fastx.c:86-116;
//   * line ends: every line is cut at its first CR or LF like chomp() does (chomp.c:36-41), so CRLF input gives LF output;
//   * quality lines: as many characters as bases = ASCII; otherwise whitespace-separated integers parsed with strtol()'s
//     rules (fastx.c:137-167), per record; kept records are written in the encoding they came in (fastx.c:393-395);
//   * collapsed FASTA identifiers (">id-count", fastx.c:475-495): every record's read count, for the -v reports.
// Anything malformed (wrong prefix, empty or over-long line, bad base or quality, wrong number of values) is only DETECTED
// (info.irregular); the caller then runs that block through the host parser, which owns the reference's exact messages.
#pragma once
#include "fxg_device.h"

#define FXG_TEXT_SEG 4096u            // bytes of text per workgroup in the newline passes (256 lanes x 16 B)
#define FXG_REC_NUMERIC 1u            // per-record flag: the quality line holds numbers, not characters

struct FxgTextState {                 // device-resident scalars of one block of text
    u32 has_cr;                       // any '\r' byte
    u32 max_len, min_len;
    u32 irregular;                    // FXG_TEXT_IRR_* bits
    u32 first_bad;                    // smallest irregular record index
    u32 n_numeric;                    // records with numeric quality lines
    u32 pad[2];
    u64 weighted[8];                  // fxg_kernel_text_weights: read-count weighted tallies (FASTA with collapsed ids)
};

// ---------------------------------------------------------------------------------------------------------
// generic exclusive scan of u64 (three small kernels, recursive on the block sums)
// ---------------------------------------------------------------------------------------------------------
#define FXG_SCAN_PER_BLOCK 1024u      // 256 threads x 4 items

#ifndef FXG_HOST_EMULATION
__device__ __forceinline__ u64 fxg_wave_incl_scan64(u64 v)
{
    const u32 lane = fxg_lane();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 lo = __shfl_up((u32)v, d, 64), hi = __shfl_up((u32)(v >> 32), d, 64);
        if ((int)lane >= d) v += ((u64)hi << 32) | lo;
    }
    return v;
}

// in-place exclusive scan of each 1024-item block; block totals to sums[]
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_scan_blocks(u64 *data, u64 n, u64 *sums)
{
    __shared__ u64 wsum[FXG_WAVES];
    const u64 base = (u64)blockIdx.x * FXG_SCAN_PER_BLOCK + (u64)threadIdx.x * 4;
    u64 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (base + i < n) ? data[base + i] : 0ull;
    const u64 mine = v[0] + v[1] + v[2] + v[3];
    const u64 incl = fxg_wave_incl_scan64(mine);
    const u32 wave = threadIdx.x >> 6;
    if (fxg_lane() == 63) wsum[wave] = incl;
    __syncthreads();
    u64 off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < FXG_WAVES; ++w) { if (w < (int)wave) off += wsum[w]; tot += wsum[w]; }
    u64 run = off + incl - mine;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (base + i < n) data[base + i] = run; run += v[i]; }
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_scan_add(u64 *data, u64 n, const u64 *block_off)
{
    const u64 base = (u64)blockIdx.x * FXG_SCAN_PER_BLOCK + (u64)threadIdx.x * 4;
    const u64 add = block_off[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i) if (base + i < n) data[base + i] += add;
}

#endif  // FXG_HOST_EMULATION

// ---------------------------------------------------------------------------------------------------------
// newline census and scatter.  Per-thread bodies are __host__ __device__ (tests/emu runs them serially on the CPU tier); the
// kernels around them add only the wave-level reductions.
// ---------------------------------------------------------------------------------------------------------
// bit i of the result = byte i of v equals ch
FXG_HD u32 fxg_eq_mask16(u32x4 v, u32 ch)
{
    const u32 c4 = ch * 0x01010101u;
    u32 m = 0;
    const u32 w[4] = {v.x ^ c4, v.y ^ c4, v.z ^ c4, v.w ^ c4};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // exact zero-byte detector: bit 7 of each byte is set iff the byte is non-zero
        const u32 nz = (((w[i] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w[i]) & 0x80808080u;
        m |= fxg_pack4(nz ^ 0x80808080u) << (4 * i);
    }
    return m;
}

// 16 bytes of text at offset off (zero beyond text_len); the buffer itself is readable 16 bytes past text_len
FXG_HD u32x4 fxg_text16(const uint8_t *text, u64 off, u64 text_len)
{
    u32x4 v = fxg_ld16(text + off);
    if (off + 16 > text_len) {
        const int keep = off >= text_len ? 0 : (int)(text_len - off);
        v = fxg_keep_bytes(v, 0, keep);
    }
    return v;
}

// one lane's 16 bytes of a segment: newline mask of text[off, off + 16); *cr (optional): carriage-return mask in the low half, NUL
// bytes of the text in the high half (the reference's lines are C strings: a NUL ends a line's content, which only the host reader does)
FXG_HD u32 fxg_text_nl_mask(const uint8_t *text, u64 off, u64 text_len, u32 *cr)
{
    if (off >= text_len) { if (cr) *cr = 0u; return 0u; }
    const u32x4 v = fxg_text16(text, off, text_len);
    if (cr) {
        const u32 valid = text_len - off >= 16u ? 0xFFFFu : ((1u << (u32)(text_len - off)) - 1u);      // fxg_text16 zero-fills past the end
        *cr = fxg_eq_mask16(v, '\r') | ((fxg_eq_mask16(v, 0) & valid) << 16);
    }
    return fxg_eq_mask16(v, '\n');
}
// the newlines of one lane's mask are lines j, j + 1, ...: line_end[j] = their position, line_start[j + 1] = the byte after
FXG_HD void fxg_text_nl_store(u32 m, u64 off, u64 j, u32 *line_start, u32 *line_end, u64 cap_lines)
{
    while (m) {
        const u32 b = (u32)__builtin_ctz(m);
        m &= m - 1u;
        if (j + 1 < cap_lines) { line_start[j + 1] = (u32)(off + b + 1); line_end[j] = (u32)(off + b); }
        ++j;
    }
}

#ifndef FXG_HOST_EMULATION
// pass 1: newlines per 4 KB segment (+ carriage-return detection)
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_nl_count(const uint8_t *text, u64 text_len, u64 *seg_count, FxgTextState *st)
{
    __shared__ u32 wsum[FXG_WAVES];
    const u64 off = (u64)blockIdx.x * FXG_TEXT_SEG + (u64)threadIdx.x * 16;
    u32 cr = 0;
    u32 c = (u32)__builtin_popcount(fxg_text_nl_mask(text, off, text_len, &cr));
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if (fxg_lane() == 0) wsum[threadIdx.x >> 6] = c;
    if (__ballot((cr & 0xFFFFu) != 0u) != 0ull && fxg_lane() == 0) atomicOr(&st->has_cr, 1u);
    if (__ballot((cr >> 16) != 0u) != 0ull && fxg_lane() == 0) atomicOr(&st->irregular, FXG_TEXT_IRR_NUL);
    __syncthreads();
    if (threadIdx.x == 0) seg_count[blockIdx.x] = (u64)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// pass 2: line_end[j] = position of the j-th newline, line_start[j + 1] = the byte after it (seg_off = exclusive scan of seg_count)
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_nl_scatter(const uint8_t *text, u64 text_len, const u64 *seg_off, u32 *line_start, u32 *line_end, u64 cap_lines)
{
    __shared__ u32 wsum[FXG_WAVES];
    const u64 off = (u64)blockIdx.x * FXG_TEXT_SEG + (u64)threadIdx.x * 16;
    const u32 m = fxg_text_nl_mask(text, off, text_len, nullptr);
    const u32 mine = (u32)__builtin_popcount(m);
    u32 incl = mine;
    const u32 lane = fxg_lane(), wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(incl, d, 64); if ((int)lane >= d) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 woff = 0;
#pragma unroll
    for (int w = 0; w < FXG_WAVES; ++w) if (w < (int)wave) woff += wsum[w];
    if (blockIdx.x == 0 && threadIdx.x == 0) line_start[0] = 0u;
    fxg_text_nl_store(m, off, seg_off[blockIdx.x] + woff + incl - mine, line_start, line_end, cap_lines);
}
#endif  // FXG_HOST_EMULATION

// ---------------------------------------------------------------------------------------------------------
// per-record checks on the line index; chomp; quality-line encoding; writes len[], flags[] and the batch extrema
// ---------------------------------------------------------------------------------------------------------
// first '\r' in text[s, e), or e
FXG_HD u32 fxg_first_cr(const uint8_t *text, u32 s, u32 e)
{
    for (u32 p = s; p < e; p += 16u) {
        u32 m = fxg_eq_mask16(fxg_ld16(text + p), '\r');
        if (e - p < 16u) m &= (1u << (e - p)) - 1u;
        if (m) return p + (u32)__builtin_ctz(m);
    }
    return e;
}

// Numeric quality line text[s, e) with the reference's token rules (fastx.c:137-167): strtol() on the rest of the line until the
// rest is empty -- leading isspace() bytes, one optional sign, digits; the long lands in an int.  Returns the number of values, or
// -1 if a token is not a number or a value lies outside -15..93.  out (optional) receives value + 33 for the first cap values.
FXG_HD int fxg_parse_numeric(const uint8_t *text, u32 s, u32 e, uint8_t *out, u32 cap)
{
    u32 p = s;
    int cnt = 0;
    do {
        while (p < e && (text[p] == ' ' || (text[p] >= '\t' && text[p] <= '\r'))) ++p;
        bool neg = false;
        if (p < e && (text[p] == '-' || text[p] == '+')) { neg = (text[p] == '-'); ++p; }
        const u32 d0 = p;
        u64 mag = 0;
        bool sat = false;
        const u64 limit = neg ? 0x8000000000000000ull : 0x7FFFFFFFFFFFFFFFull;      // strtol saturates exactly past LONG_MAX / below LONG_MIN
        for (; p < e && text[p] >= '0' && text[p] <= '9'; ++p) {
            const u64 d = (u64)(text[p] - '0');
            if (sat || mag > (limit - d) / 10ull) sat = true; else mag = mag * 10ull + d;
        }
        if (p == d0) return -1;
        const long long lv = sat ? (neg ? (-0x7FFFFFFFFFFFFFFFll - 1ll) : 0x7FFFFFFFFFFFFFFFll) : (long long)(neg ? 0ull - mag : mag);
        const int v = (int)lv;
        if (v > 93 || v < -15) return -1;
        if (out && (u32)cnt < cap) out[cnt] = (uint8_t)(v + 33);
        ++cnt;
    } while (p < e);
    return cnt;
}

// LPR: lines per record (4 FASTQ, 2 FASTA).  One record: chomp its lines, check them, classify its quality line; writes le[] (when the
// block has CR bytes), len[r], flags[r]; returns the FXG_TEXT_IRR_* bits, the sequence length and whether the quality line is numeric.
template <int LPR>
FXG_HD u32 fxg_text_record(const uint8_t *text, const u32 *ls, u32 *le, u64 r, u32 has_cr, uint16_t *len, uint8_t *flags, u32 *seq_len, u32 *numeric)
{
    const u64 b = (u64)LPR * r;
    u32 irr = 0;
    u32 s[LPR], e[LPR];
#pragma unroll
    for (int k = 0; k < LPR; ++k) { s[k] = ls[b + k]; e[k] = le[b + k]; }
    if (has_cr) {                                            // chomp: a line ends at its first CR (chomp.c:36-41)
#pragma unroll
        for (int k = 0; k < LPR; ++k) { e[k] = fxg_first_cr(text, s[k], e[k]); le[b + k] = e[k]; }
    }
    const u32 sl = e[1] - s[1];
    if (s[0] == e[0] || text[s[0]] != (LPR == 4 ? '@' : '>')) irr |= FXG_TEXT_IRR_PREFIX;      // (an empty first line has no prefix either)
    if (sl == 0u || sl >= 24998u || e[0] - s[0] >= 24999u) irr |= FXG_TEXT_IRR_SEQLEN;
    u32 fl = 0;
    if (LPR == 4) {
        const u32 ql = e[3] - s[3];
        if (e[2] - s[2] >= 24999u) irr |= FXG_TEXT_IRR_SEQLEN;
        if (ql != sl && !irr) {                                // R6: not one character per base -> numbers
            if (fxg_parse_numeric(text, s[3], e[3], nullptr, 0u) == (int)sl) fl = FXG_REC_NUMERIC;
            else irr |= FXG_TEXT_IRR_QUALLEN;
        }
    }
    flags[r] = (uint8_t)fl;
    len[r] = (uint16_t)(sl > 65535u ? 65535u : sl);
    *seq_len = sl; *numeric = fl;
    return irr;
}

#ifndef FXG_HOST_EMULATION
template <int LPR>
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_text_records(const uint8_t *text, const u32 *ls, u32 *le, u64 n, uint16_t *len, uint8_t *flags, FxgTextState *st)
{
    const u64 r = (u64)blockIdx.x * FXG_BLOCK + threadIdx.x;
    u32 irr = 0, sl = 0;
    if (r < n) {
        u32 fl = 0;
        irr = fxg_text_record<LPR>(text, ls, le, r, st->has_cr, len, flags, &sl, &fl);
        if (fl) atomicAdd(&st->n_numeric, 1u);
        if (irr) { atomicOr(&st->irregular, irr); atomicMin(&st->first_bad, (u32)r); }
    }
    // batch extrema: wave reduction, then one pair of atomics per workgroup (a single address takes ~11 ns per atomic)
    __shared__ u32 smx[FXG_WAVES], smn[FXG_WAVES];
    u32 mx = (r < n && !irr) ? sl : 0u, mn = (r < n && !irr) ? sl : 0xFFFFFFFFu;
    for (int d = 32; d >= 1; d >>= 1) { mx = max(mx, (u32)__shfl_xor(mx, d, 64)); mn = min(mn, (u32)__shfl_xor(mn, d, 64)); }
    if (fxg_lane() == 0) { smx[threadIdx.x >> 6] = mx; smn[threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = max(max(smx[0], smx[1]), max(smx[2], smx[3]));
        mn = min(min(smn[0], smn[1]), min(smn[2], smn[3]));
        if (mx) atomicMax(&st->max_len, mx);
        if (mn != 0xFFFFFFFFu) atomicMin(&st->min_len, mn);
    }
}
#endif  // FXG_HOST_EMULATION

// ---------------------------------------------------------------------------------------------------------
// text -> SoA rows.  One lane owns one 16-byte aligned chunk of the row array and assembles it from the
// sequence (or quality) lines of the records that intersect it; bytes past a read's length are zero.
// Qualities are normalised to Phred+33 codes (byte - (Q - 33)) and range-checked; bases are alphabet-checked.
// Records with numeric quality lines are left zero here and filled by fxg_kernel_text_numeric.
// ---------------------------------------------------------------------------------------------------------
// one 16-byte chunk c of the row array; returns nonzero if a byte of it is not a valid base / quality character
template <bool QUAL, int LPR>
FXG_HD u32 fxg_text_pack_chunk(const uint8_t *text, u64 text_len, const u32 *ls, const u32 *le, const uint8_t *flags, u64 n, u32 stride, int qoffset, uint8_t *rows, u64 c)
{
    const int qlo = qoffset - 15 < 0 ? 0 : qoffset - 15, qhi = qoffset + 93 > 127 ? 127 : qoffset + 93;   // valid raw characters
    const u32 Klo = (u32)(128 - qlo) * 0x01010101u, Khi = (u32)(128 - (qhi + 1)) * 0x01010101u;
    const int dq = qoffset - 33;                                     // raw character -> Phred+33 code
    const u32 adj4 = (u32)(dq >= 0 ? dq : -dq) * 0x01010101u;
    u32 bad = 0;
    {
        const u64 b0 = c << 4;
        u64 r = b0 / stride;
        u32 pos = (u32)(b0 - r * stride);            // position inside row r of chunk byte 0
        u32x4 acc = {0u, 0u, 0u, 0u};
        int filled = 0;
        while (filled < 16 && r < n) {
            const u64 lb = (u64)LPR * r;
            const u32 o = QUAL ? ls[lb + 3] : ls[lb + 1];
            const u32 rl = (QUAL && (flags[r] & FXG_REC_NUMERIC)) ? 0u : le[lb + 1] - ls[lb + 1];
            const int take_row = (int)(stride - pos) < 16 - filled ? (int)(stride - pos) : 16 - filled;    // bytes of this row in the chunk
            const int have = pos < rl ? ((int)(rl - pos) < take_row ? (int)(rl - pos) : take_row) : 0;    // of which real data
            if (have > 0) {
                // window whose byte `filled` is text[o + pos]
                const long long src = (long long)o + pos - filled;
                u32x4 w;
                if (src >= 0 && (u64)src + 16 <= text_len + 16) w = fxg_ld16(text + src);
                else {                                      // window would start before the buffer: assemble the needed bytes
                    u64 lo = 0, hi = 0;
                    for (int i = 0; i < have; ++i) {
                        const u64 by = text[o + pos + i];
                        const int k = filled + i;
                        if (k < 8) lo |= by << (8 * k); else hi |= by << (8 * (k - 8));
                    }
                    w = (u32x4){(u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32)};
                }
                w = fxg_keep_bytes(w, filled, filled + have);
                const u32x4 m = fxg_keep_bytes((u32x4){~0u, ~0u, ~0u, ~0u}, filled, filled + have);
                if (QUAL) {
                    // valid iff qlo <= byte <= qhi (bytes >= 128 are negative chars in the reference: invalid)
                    const u32 ws[4] = {w.x, w.y, w.z, w.w}, ms[4] = {m.x, m.y, m.z, m.w};
                    u32 outw[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32 ge_lo = (((ws[i] & 0x7f7f7f7fu) + Klo)) & 0x80808080u;                 // byte(7 bits) >= qlo
                        const u32 ge_hi1 = (((ws[i] & 0x7f7f7f7fu) + Khi)) & 0x80808080u;               // byte(7 bits) >= qhi + 1
                        const u32 hib = ws[i] & 0x80808080u;                                             // byte >= 128
                        const u32 ok = ge_lo & ~ge_hi1 & ~hib;
                        bad |= (~ok & 0x80808080u) & ms[i];
                        // no borrow / carry between bytes when the bytes are valid: Q-33 <= qlo <= byte and byte + 33 - Q <= 126
                        outw[i] = (dq >= 0 ? ws[i] - (adj4 & ms[i]) : ws[i] + (adj4 & ms[i])) & ms[i];
                    }
                    w = (u32x4){outw[0], outw[1], outw[2], outw[3]};
                } else {
                    // upper-case ACGTN only (fastx.c:56-84 with ALLOW_N, REQUIRE_UPPERCASE)
                    const u32 ws[4] = {w.x, w.y, w.z, w.w}, ms[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) bad |= fxg_invalid_bases4(ws[i], ms[i]) | ((ws[i] & 0x20202020u) & ms[i]);
                }
                acc |= w;
            }
            filled += take_row;
            pos += (u32)take_row;
            if (pos >= stride) { pos = 0; ++r; }
        }
        *reinterpret_cast<u32x4 *>(rows + b0) = acc;       // rows[] is padded to a multiple of 16 bytes by the caller
    }
    return bad;
}

// quality row of record r when its quality line is numeric (fxg_text_pack_chunk<true> left it zero)
FXG_HD void fxg_text_numeric_row(const uint8_t *text, const u32 *ls, const u32 *le, const uint8_t *flags, u32 stride, uint8_t *rows, u64 r)
{
    if (!(flags[r] & FXG_REC_NUMERIC)) return;
    (void)fxg_parse_numeric(text, ls[4 * r + 3], le[4 * r + 3], rows + r * (u64)stride, stride);
}

#ifndef FXG_HOST_EMULATION
template <bool QUAL, int LPR>
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_text_pack(const uint8_t *text, u64 text_len, const u32 *ls, const u32 *le, const uint8_t *flags, u64 n, u32 stride,
                                                                  int qoffset, uint8_t *rows, FxgTextState *st)
{
    const u64 nchunks = (n * (u64)stride + 15) >> 4;
    u32 bad = 0;
    for (u64 c = (u64)blockIdx.x * FXG_BLOCK + threadIdx.x; c < nchunks; c += (u64)gridDim.x * FXG_BLOCK)
        bad |= fxg_text_pack_chunk<QUAL, LPR>(text, text_len, ls, le, flags, n, stride, qoffset, rows, c);
    if (bad) atomicOr(&st->irregular, QUAL ? FXG_TEXT_IRR_QUAL : FXG_TEXT_IRR_BASE);
}

// quality rows of the records whose quality line is numeric (after fxg_kernel_text_pack<true> left them zero)
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_text_numeric(const uint8_t *text, const u32 *ls, const u32 *le, const uint8_t *flags, u64 n, u32 stride, uint8_t *rows)
{
    const u64 r = (u64)blockIdx.x * FXG_BLOCK + threadIdx.x;
    if (r < n) fxg_text_numeric_row(text, ls, le, flags, stride, rows, r);
}
#endif  // FXG_HOST_EMULATION

// read count of a FASTA record: the number after the first '-' of its identifier, 1 if there is none (fastx.c:475-495)
FXG_HD u32 fxg_reads_count(const uint8_t *text, u32 s, u32 e)
{
    u32 p = s;
    while (p < e && text[p] != '-') ++p;
    if (p >= e) return 1u;
    ++p;
    while (p < e && (text[p] == ' ' || (text[p] >= '\t' && text[p] <= '\r'))) ++p;     // atoi skips blanks and takes one sign
    bool neg = false;
    if (p < e && (text[p] == '-' || text[p] == '+')) { neg = (text[p] == '-'); ++p; }
    u64 v = 0;
    for (u32 k = 0; p < e && text[p] >= '0' && text[p] <= '9' && k < 23u; ++p, ++k) v = v * 10u + (u64)(text[p] - '0');
    const int c = neg ? -(int)(u32)v : (int)(u32)v;
    return c > 0 ? (u32)c : 1u;
}

// -v report tallies weighted by read count (FASTA input): [0] input reads, [1] output reads, then the clipper's five reasons
FXG_HD void fxg_text_weights_record(const uint8_t *text, const u32 *ls, const u32 *le, const u32 *res, u64 r, u64 (&v)[7])
{
    const u64 w = fxg_reads_count(text, ls[2 * r] + 1u, le[2 * r]);
    const u32 x = res[r], why = (x >> 17) & 0xFu;
    v[0] = w;
    if ((x >> 16) & 1u) v[1] = w;
    if (why == FXG_R_CLIP_TOO_SHORT) v[2] = w;
    if ((x >> FXG_RES_ADAPTER_ONLY_BIT) & 1u) v[3] = w;
    if (why == FXG_R_CLIP_NO_ADAPTER) v[4] = w;
    if (why == FXG_R_CLIP_ADAPTER_FOUND) v[5] = w;
    if (why == FXG_R_CLIP_N) v[6] = w;
}

#ifndef FXG_HOST_EMULATION
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_text_weights(const uint8_t *text, const u32 *ls, const u32 *le, const u32 *res, u64 n, FxgTextState *st)
{
    const u64 r = (u64)blockIdx.x * FXG_BLOCK + threadIdx.x;
    u64 v[7] = {0, 0, 0, 0, 0, 0, 0};
    if (r < n) fxg_text_weights_record(text, ls, le, res, r, v);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        u64 x = v[i];
        for (int d = 32; d >= 1; d >>= 1) x += ((u64)__shfl_xor((u32)(x >> 32), d, 64) << 32) | __shfl_xor((u32)x, d, 64);
        if (fxg_lane() == 0 && x) atomicAdd(&st->weighted[i], x);
    }
}
#endif  // FXG_HOST_EMULATION

// ---------------------------------------------------------------------------------------------------------
// formatting: sizes -> (scan on the host side of this header) -> copy
// ---------------------------------------------------------------------------------------------------------
struct FxgFormatArgs {
    const uint8_t *text;
    const u32 *ls, *le, *res;
    const uint8_t *flags;
    const u64 *item_scan;
    u64 n;
    u32 fwd_start;                    // first kept base of forward outputs (fastx_trimmer -f)
    u32 rev;                          // packed outputs are reverse-complemented: their qualities are those of input positions [rl - fwd_start - len, rl - fwd_start)
    const uint8_t *pk_bases, *pk_qual; const u64 *pk_off;    // packed (reverse-complemented / masked) outputs, or null
    const uint8_t *rows_qual; u32 stride;                   // the batch's quality rows (Phred+33 codes): numeric output of forward records
    int qoffset;
    u32 out_fasta;                    // write FASTA whatever the input was (fastq_to_fasta)
    uint8_t *out;
};

FXG_HD u32 fxg_num_width(int v) { return (v < 0 ? 1u : 0u) + ((v <= -10 || v >= 10) ? 2u : 1u); }     // -15..93

// item r = output bytes of record r in the low 40 bits, keep flag above (the scan then yields offset and rank)
template <int LPR>
FXG_HD u64 fxg_text_size_record(const FxgFormatArgs &a, u64 r)
{
    const u32 w = a.res[r];
    u64 v = 0;
    if ((w >> 16) & 1u) {
        const u64 b = (u64)LPR * r;
        const u32 name_len = a.le[b] - a.ls[b] - 1u;                                // without the prefix character
        const u32 len = w & 0xFFFFu;
        u64 bytes = (u64)name_len + len + 3u;                                       // prefix, name, LF, bases, LF
        if (LPR == 4 && !a.out_fasta) {
            const u32 l2 = a.le[b + 2] - a.ls[b + 2];
            const u32 name2_len = l2 ? l2 - 1u : 0u;                                // first byte dropped (R5)
            u32 qbytes = len;
            if (a.flags[r] & FXG_REC_NUMERIC) {                                     // "%d" joined by blanks: only the digit counts depend on the values
                const u32 rl = a.le[b + 1] - a.ls[b + 1];
                const u32 q0 = a.rev ? rl - a.fwd_start - len : (a.pk_bases ? 0u : a.fwd_start);
                const uint8_t *q = a.rows_qual + r * (u64)a.stride + q0;
                qbytes = len ? len - 1u : 0u;
                for (u32 i = 0; i < len; ++i) qbytes += fxg_num_width((int)q[i] - 33);
            }
            bytes += (u64)name2_len + qbytes + 3u;                                  // '+', name2, LF, qualities, LF
        }
        v = bytes | (1ull << 40);
    }
    return v;
}

// n bytes from src to dst, both arbitrarily aligned, by `lanes` cooperating lanes (lane id `l`); add is applied per byte
FXG_HD void fxg_copy_bytes(uint8_t *dst, const uint8_t *src, u32 n, u32 l, u32 lanes, int add)
{
    const u32 full = n >> 4;
    const u32 a4 = (u32)(add >= 0 ? add : -add) * 0x01010101u;
    for (u32 k = l; k < full; k += lanes) {
        u32x4 v = fxg_ld16(src + (k << 4));
        if (add > 0) { v.x += a4; v.y += a4; v.z += a4; v.w += a4; }        // Phred+33 code -> character: stays below 256 per byte
        else if (add < 0) { v.x -= a4; v.y -= a4; v.z -= a4; v.w -= a4; }
        __builtin_memcpy(dst + (k << 4), &v, 16);
    }
    for (u32 i = (full << 4) + l; i < n; i += lanes) dst[i] = (uint8_t)((int)src[i] + add);
}

// 16 lanes format one kept record: "@name\nSEQ\n+name2\nQUAL\n" (FASTQ) or ">name\nSEQ\n" (FASTA).
// Forward outputs are slices of the input lines; reverse-complemented / masked outputs come from the engine's packed arrays at
// pk_off[rank] and hold Phred+33 codes.  Numeric quality lines are printed from the codes by one lane.
template <int LPR>
FXG_HD void fxg_text_format_record(const FxgFormatArgs &a, u64 r, u32 l)
{
    const u32 w = a.res[r];
    if (!((w >> 16) & 1u)) return;
    const u64 sc = a.item_scan[r];
    const u64 off = sc & ((1ull << 40) - 1ull), rank = sc >> 40;
    const u64 b = (u64)LPR * r;
    const u32 o0 = a.ls[b], o1 = a.ls[b + 1];
    const u32 name_len = a.le[b] - o0 - 1u;
    const u32 len = w & 0xFFFFu;
    const bool fastq = (LPR == 4 && !a.out_fasta);
    uint8_t *d = a.out + off;
    const u64 po = a.pk_bases ? a.pk_off[rank] : 0ull;
    if (l == 0) { d[0] = fastq ? '@' : '>'; d[1 + name_len] = '\n'; d[2 + name_len + len] = '\n'; }
    fxg_copy_bytes(d + 1, a.text + o0 + 1, name_len, l, 16, 0);
    if (a.pk_bases) fxg_copy_bytes(d + 2 + name_len, a.pk_bases + po, len, l, 16, 0);
    else fxg_copy_bytes(d + 2 + name_len, a.text + o1 + a.fwd_start, len, l, 16, 0);
    if (!fastq) return;
    const u32 o2 = a.ls[b + 2], o3 = a.ls[b + 3];
    const u32 l2 = a.le[b + 2] - o2, name2_len = l2 ? l2 - 1u : 0u;
    uint8_t *q = d + 3 + name_len + len;                 // '+'
    if (l == 0) { q[0] = '+'; q[1 + name2_len] = '\n'; }
    fxg_copy_bytes(q + 1, a.text + o2 + 1, name2_len, l, 16, 0);
    uint8_t *qd = q + 2 + name2_len;
    if (!(a.flags[r] & FXG_REC_NUMERIC)) {
        if (a.pk_bases) fxg_copy_bytes(qd, a.pk_qual + po, len, l, 16, a.qoffset - 33);
        else fxg_copy_bytes(qd, a.text + o3 + a.fwd_start, len, l, 16, 0);    // R8: q + Q is the input byte
        if (l == 0) qd[len] = '\n';
    } else if (l == 0) {                                  // numbers separated by blanks, as write_ascii / numeric output does (fastx.c:421-438)
        const uint8_t *src = a.pk_bases ? a.pk_qual + po : a.rows_qual + r * (u64)a.stride + a.fwd_start;
        u32 k = 0;
        for (u32 i = 0; i < len; ++i) {
            int v = (int)src[i] - 33;
            if (i) qd[k++] = ' ';
            if (v < 0) { qd[k++] = '-'; v = -v; }
            if (v >= 10) qd[k++] = (uint8_t)('0' + v / 10);
            qd[k++] = (uint8_t)('0' + v % 10);
        }
        qd[k] = '\n';
    }
}

#ifndef FXG_HOST_EMULATION
template <int LPR>
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_text_sizes(const FxgFormatArgs a, u64 *item)
{
    const u64 r = (u64)blockIdx.x * FXG_BLOCK + threadIdx.x;
    if (r < a.n) item[r] = fxg_text_size_record<LPR>(a, r);
}

template <int LPR>
__global__ __launch_bounds__(FXG_BLOCK) void fxg_kernel_text_format(const FxgFormatArgs a)
{
    const u64 r = ((u64)blockIdx.x * FXG_BLOCK + threadIdx.x) >> 4;
    if (r < a.n) fxg_text_format_record<LPR>(a, r, threadIdx.x & 15u);
}
#endif  // FXG_HOST_EMULATION
