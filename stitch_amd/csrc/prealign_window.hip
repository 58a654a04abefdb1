// Banded Smith-Waterman score (prealign.h) with the column state in REGISTERS: the band rows of a column live in a window of 256
// consecutive read rows, four per lane (row = base + 4 * lane + u), that slides down the read as the band does.  One wavefront per
// pair, no LDS; a column is ONE block (the LDS-ring kernel of prealign_kernel.hip walks a column in 64-row blocks, 3.3 of them
// per column at w = 50, each with its own prefix scan and three LDS round trips):
//   * H and D of the previous column are the lane's own registers; the diagonal of a lane's first row is the last row of the lane
//     before (one DPP shift);
//   * the vertical (insertion) chain is the four rows of a lane in sequence plus ONE 64-lane DPP prefix maximum of the lanes'
//     maxima of T - ge * i (as in the other kernels: I(i) = go + ge * i + max_{k < i} (T(k) - ge * k));
//   * when the band's first row passes the next multiple of four, every register moves one lane down (DPP wave_shl:1; the lane
//     that comes free holds "no value"), and the lanes' read bases (one aligned word each: a read's first base sits at an offset
//     congruent to 1 modulo 4, row i compares base i - 1) move with them, the incoming word taken from a reserve of 64 words that
//     is refilled every 64 shifts.
// Rows outside the band hold NONE_V = -2^29 after every column ("not in the band" for the next one), so nothing needs the band
// ranges of the previous column (row 0 is set to 0 before every column: the reference reads H(0, j) = 0 whatever the band).  No value is clamped: a score that is not positive never wins (T >= 0), and the sums stay
// within 32 bits (checked by the host).  The host sends a pair here when its band's first row never decreases and every column
// fits the window (band_fits_window); the other pairs take the LDS-ring or the global-state kernel.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "prealign.h"

namespace stitch {

namespace {
constexpr int32_t NONE_V = -(1 << 29), NONE_KEY = -(1 << 30);
constexpr uint32_t WIN_ROWS = 256;

__device__ __forceinline__ int32_t prefix_max_incl(int32_t v) {            // inclusive prefix maximum over the 64 lanes
    // v_max_i32 with a DPP source and bound_ctrl off: a lane whose source does not exist keeps its value, which is what a maximum with
    // "nothing" is - one instruction per step where `old value + v_mov_dpp + v_max` are three.  (The s_nop 1 are the two wait states
    // between a vector write and a DPP read of the same register: the compiler does not look into inline assembly.)
    asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" : "+v"(v));
    return v;
}
__device__ __forceinline__ int32_t from_lane_above(int32_t v, int32_t first) {      // lane l takes lane l - 1's value, lane 0 `first`
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xF, 0xF, false);            // wave_shr:1
}
__device__ __forceinline__ int32_t from_lane_below(int32_t v, int32_t last) {       // lane l takes lane l + 1's value, lane 63 `last`
    return __builtin_amdgcn_update_dpp(last, v, 0x130, 0xF, 0xF, false);             // wave_shl:1
}
}  // namespace

__global__ __launch_bounds__(64) void banded_score_window_kernel(const BandPair* __restrict__ pairs, const uint32_t* __restrict__ which, BandScoring sc,
                                                                 const uint8_t* __restrict__ reads, const uint8_t* __restrict__ contigs,
                                                                 const uint16_t* __restrict__ bands, int32_t* __restrict__ scores, const uint32_t* __restrict__ cls) {
    const uint32_t pid = which[blockIdx.x];
    if (cls && cls[pid] != BAND_CLASS_WINDOW) return;                    // (the device drew the bands and chose the kernels: prealign_band.hip)
    const BandPair P = pairs[pid];
    const uint32_t lane = threadIdx.x;
    const uint32_t m = P.m, n = P.n;
    const uint32_t* qwords = (const uint32_t*)(reads + P.q_off - 1);     // word k = the bases of rows 4k .. 4k + 3 (row 0 has none: the byte before the read)
    const uint32_t n_qwords = (m + 4) / 4;
    const uint8_t* t = contigs + P.t_off;
    const uint16_t* lo = bands + P.band_off; const uint16_t* hi = lo + (n + 1);
    const int32_t ge = sc.gap_extend, go = sc.gap_open, goe = go + ge;

    int32_t H[4], D[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { H[u] = 0; D[u] = NONE_V; }            // column 0: H = 0 in every row, no deletion yet
    uint32_t base = 0;                                                   // first row of the window (a multiple of 4); set before column 1
    bool placed = false;
    uint32_t ib = 4 * lane;                                              // the lane's first row
    uint32_t qw = 0, reserve = 0, used = 64;                             // the lane's four bases; 64 words beyond the window; how many of them are spent
    int32_t best = 0;
    uint32_t vlo = 0, vhi = 0, vt = 0;
    for (uint32_t j = 1; j <= n; ++j) {
        const uint32_t jl = (j - 1) & 63u;
        if (jl == 0) {                                                    // ranges and target bases of columns j .. j + 63
            const uint32_t c = j + lane;
            vlo = c <= n ? lo[c] : 0u; vhi = c <= n ? hi[c] : 0u; vt = c <= n ? t[c - 1] : 0u;
        }
        const uint32_t clo = (uint32_t)__builtin_amdgcn_readlane((int)vlo, (int)jl), chi = (uint32_t)__builtin_amdgcn_readlane((int)vhi, (int)jl);
        const uint32_t tj = (uint32_t)__builtin_amdgcn_readlane((int)vt, (int)jl);
        uint32_t r0 = max(clo, 1u), r1 = min(chi, m + 1);
        uint32_t v0 = r0 == 1 ? 0u : r0;                                  // row 0 (H = 0) belongs to a column that starts in row 1
        if (r0 >= r1) { v0 = 0; r1 = 0; }                                 // an empty column: every row leaves it as "no value"
        else {
            const uint32_t want = (r0 - 1) & ~3u;
            if (!placed) {                                                // the first column with rows: put the window there
                placed = true; base = want; ib = base + 4 * lane;
                const uint32_t w = base / 4 + lane;
                qw = w < n_qwords ? qwords[w] : 0u; used = 64;
                if (j > 1) {                                              // (only column 1 sees column 0; behind empty columns there is nothing)
#pragma unroll
                    for (int u = 0; u < 4; ++u) { H[u] = NONE_V; D[u] = NONE_V; }
                }
            }
            while (base < want) {                                         // one lane down per step
                if (used == 64) { const uint32_t w = base / 4 + 64 + lane; reserve = w < n_qwords ? qwords[w] : 0u; used = 0; }
                const uint32_t incoming = (uint32_t)__builtin_amdgcn_readlane((int)reserve, (int)used);
                ++used;
                qw = (uint32_t)from_lane_below((int32_t)qw, (int32_t)incoming);
#pragma unroll
                for (int u = 0; u < 4; ++u) { H[u] = from_lane_below(H[u], NONE_V); D[u] = from_lane_below(D[u], NONE_V); }
                base += 4; ib += 4;
            }
        }
        const uint32_t span = r1 - v0, rel = ib - v0;                     // row ib + u is in the band when rel + u < span (unsigned)
        const int32_t kc = -ge * (int32_t)ib, ic = go + ge * (int32_t)ib;
        H[0] = (base == 0 && lane == 0) ? 0 : H[0];                         // row 0: H = 0 in every column, in the band or not
        int32_t hd = from_lane_above(H[3], NONE_V);
        int32_t T[4], key[4], dn[4]; bool in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            in[u] = rel + (uint32_t)u < span;
            const int32_t hl = H[u];
            dn[u] = max(D[u] + ge, hl + goe);
            const int32_t s = ((qw >> (8 * u)) & 0xFFu) == tj ? sc.match : sc.mismatch;
            T[u] = max(max(hd + s, dn[u]), 0);
            hd = hl;
            key[u] = in[u] ? T[u] + (kc - ge * u) : NONE_KEY;
        }
        const int32_t mine = max(max(key[0], key[1]), max(key[2], key[3]));
        int32_t pre = from_lane_above(prefix_max_incl(mine), NONE_KEY);    // the rows of the lanes before this one
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int32_t h = max(T[u], pre + (ic + ge * u));
            H[u] = in[u] ? h : NONE_V; D[u] = in[u] ? dn[u] : NONE_V;
            best = max(best, H[u]);
            pre = max(pre, key[u]);
        }
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) best = max(best, __shfl_xor(best, dd, 64));
    if (lane == 0) scores[pid] = best;
}

// Can the window kernel take this band?  (lo, hi: rows [lo[c], hi[c]) of column c = 0 .. n)
bool band_fits_window(const uint16_t* lo, const uint16_t* hi, uint32_t m, uint32_t n) {
    uint32_t prev_r0 = 0;
    for (uint32_t c = 1; c <= n; ++c) {
        const uint32_t r0 = std::max<uint32_t>(lo[c], 1u), r1 = std::min<uint32_t>(hi[c], m + 1);
        if (r0 >= r1) continue;
        if (r0 < prev_r0) return false;
        prev_r0 = r0;
        if (r1 - ((r0 - 1) & ~3u) > WIN_ROWS) return false;
    }
    return true;
}
// false = not applicable to this scoring / these lengths (sums beyond 32 bits, positive gap scores)
bool window_scoring_ok(const BandScoring& sc, uint32_t max_m) {
    const long long big = (long long)1 << 28;
    return (long long)std::abs(sc.match) * (max_m + 1) < big && (long long)std::abs(sc.gap_extend) * (max_m + 260) + std::abs(sc.gap_open) < big &&
           std::abs((long long)sc.mismatch) < big && sc.gap_extend <= 0 && sc.gap_open <= 0;
}
void launch_banded_scores_window(const BandPair* d_pairs, const uint32_t* d_which, uint32_t n_pairs, const BandScoring& sc, const uint8_t* d_reads,
                                 const uint8_t* d_contigs, const uint16_t* d_bands, int32_t* d_scores, const uint32_t* d_cls, hipStream_t stream) {
    if (n_pairs) hipLaunchKernelGGL(banded_score_window_kernel, dim3(n_pairs), dim3(64), 0, stream, d_pairs, d_which, sc, d_reads, d_contigs, d_bands, d_scores, d_cls);
}

}  // namespace stitch
