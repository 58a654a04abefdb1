// Full-matrix Smith-Waterman SCORE of (read, target strand) pairs that have no k-mer match (prealign.h: the band of such a pair is
// the whole matrix), as an anti-diagonal sweep in packed 16-bit arithmetic.  One wavefront per pair, no LDS, no barriers, no scans.
//
// The local score is symmetric in its two sequences (match / mismatch by equality of bytes, one gap_open / gap_extend for both
// gap kinds), so the matrix is laid out with the TARGET along the rows and the read along the columns: the 128 * RP target rows
// are dealt in strips of RP consecutive rows to 128 "virtual lanes" - the low and the high 16 bits of the 64 lanes' registers,
// virtual lane 2l = low half of lane l, 2l + 1 = high half - and virtual lane v works on column t - v in step t.  A strip needs
// from the strip above only that strip's results of the step before (H of its last row, the vertical gap entering this strip's first
// row), which are one DPP shift and one v_alignbit away; the read's bases travel down the virtual lanes the same way.  The rows
// of a strip are an unrolled loop over packed registers: per pair of cells 3 instructions for the horizontal gap, 4 for the
// substitution score, 2 for T = max(diagonal, horizontal, 0), 3 for the vertical gap, 1 for H, 1 for the running maximum.
//
// Columns outside the read (before a strip starts, after it ends) and rows beyond the target carry byte codes that equal
// nothing: every cell there is a mismatch cell, scores at most what a real neighbour holds minus a penalty, and never raises the
// maximum; a never-opened gap is held as gap_open + gap_extend instead of minus infinity, which is equivalent wherever it is
// negative (a negative gap score never wins against T >= 0 and only decreases when extended).
//
// Parity: the value equals full_score_reg_kernel's / the oracle's banded score with a full band; tests/test_prealign.py
// compares all of them.  Range (checked by the launcher): match * min(m, n) <= 32000, penalties within 16 bits, n <= 128 * 40.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "prealign.h"

namespace stitch {

namespace {
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x2 as_s(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s16x2 pmax(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ uint32_t both(int32_t v) { return ((uint32_t)v & 0xFFFFu) * 0x10001u; }
// [low half of this lane's word as the new high half, high half of the lane before as the new low half]; lane 0 takes `first` low
__device__ __forceinline__ uint32_t pass_down(uint32_t w, uint32_t first) {
    const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp((int)(first << 16), (int)w, 0x138, 0xF, 0xF, false);      // wave_shr:1
    return __builtin_amdgcn_alignbit(w, up, 16);
}
constexpr uint32_t PAD_READ = 0x100, PAD_TARGET = 0x200;      // 16-bit codes no byte equals
// min(x, 1) per half, then that * delta + addend per half: the compiler turns the plain expression into two compares, two selects
// and a permute
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b)); return r; }
__device__ __forceinline__ s16x2 pk_mad_i16(uint32_t a, s16x2 b, s16x2 c) {
    uint32_t r; asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(as_u(b)), "v"(as_u(c))); return as_s(r);
}
}  // namespace

template <int RP>
__global__ __launch_bounds__(64) void full_score_skew16_kernel(const BandPair* __restrict__ pairs, const uint32_t* __restrict__ which, BandScoring sc,
                                                               const uint8_t* __restrict__ reads, const uint8_t* __restrict__ contigs,
                                                               int32_t* __restrict__ scores) {
    const uint32_t pid = which[blockIdx.x];
    const BandPair P = pairs[pid];
    const uint32_t m = P.m, n = P.n;
    const uint8_t* q = reads + P.q_off; const uint8_t* t = contigs + P.t_off;
    const uint32_t lane = threadIdx.x;
    const s16x2 ge2 = as_s(both(sc.gap_extend)), goe2 = as_s(both(sc.gap_open + sc.gap_extend)), match2 = as_s(both(sc.match)),
                delta2 = as_s(both(sc.mismatch - sc.match)), zero2 = as_s(0u);
    const uint32_t one2 = 0x00010001u;

    uint32_t tb[RP]; s16x2 H[RP], E[RP];
    {
        const uint32_t r_lo = 2 * lane * RP, r_hi = r_lo + RP;      // 0-based first target position of the two strips
#pragma unroll
        for (int u = 0; u < RP; ++u) {
            const uint32_t a = r_lo + u < n ? (uint32_t)t[r_lo + u] : PAD_TARGET, b = r_hi + u < n ? (uint32_t)t[r_hi + u] : PAD_TARGET;
            tb[u] = a | (b << 16); H[u] = zero2; E[u] = goe2;
        }
    }
    uint32_t x2 = PAD_READ * 0x10001u;                               // read codes of this lane's two strips in this step
    uint32_t in_h = 0, in_f = as_u(goe2), diag0 = 0;                  // from the strips above: H of their last row, the vertical gap into row one; H a step earlier
    s16x2 best2 = zero2;
    uint32_t vq = 0;
    const uint32_t steps = m + 127;                                   // virtual lane 127 works on column m in step m + 127
    for (uint32_t s = 1; s <= steps; ++s) {
        const uint32_t sl = (s - 1) & 63u;
        if (sl == 0) { const uint32_t c = s + lane; vq = c <= m ? (uint32_t)q[c - 1] : PAD_READ; }      // the read's bases of steps s .. s + 63
        const uint32_t qn = (uint32_t)__builtin_amdgcn_readlane((int)vq, (int)sl);
        x2 = pass_down(x2, qn);
        s16x2 diag = as_s(diag0), f = as_s(in_f), tprev = zero2;
#pragma unroll
        for (int u = 0; u < RP; ++u) {
            const s16x2 hp = H[u];
            const s16x2 e = pmax(E[u] + ge2, hp + goe2);
            const s16x2 dg = pk_mad_i16(pk_min_u16(x2 ^ tb[u], one2), delta2, diag + match2);      // diagonal + (equal ? match : mismatch)
            const s16x2 T = pmax(pmax(dg, e), zero2);
            if (u > 0) f = pmax(f + ge2, tprev + goe2);
            const s16x2 h = pmax(T, f);
            best2 = pmax(best2, h);
            diag = hp; H[u] = h; E[u] = e; tprev = T;
        }
        const uint32_t out_f = as_u(pmax(f + ge2, tprev + goe2));
        diag0 = in_h;
        in_h = pass_down(as_u(H[RP - 1]), 0u);
        in_f = pass_down(out_f, (uint32_t)(sc.gap_open + sc.gap_extend) & 0xFFFFu);
    }
    int32_t best = max((int32_t)best2.x, (int32_t)best2.y);
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) best = max(best, __shfl_xor(best, dd, 64));
    if (lane == 0) scores[pid] = best;
}

// true when the pairs were launched here; false = not applicable (the caller uses launch_full_scores)
bool launch_full_scores_skew16(const BandPair* d_pairs, const uint32_t* d_which, uint32_t n_full, uint32_t max_m, uint32_t max_n, const BandScoring& sc,
                               const uint8_t* d_reads, const uint8_t* d_contigs, int32_t* d_scores, hipStream_t stream) {
    if (!n_full) return true;
    const long long top = (long long)std::max(sc.match, 0) * std::min(max_m, max_n);
    const bool fits = top <= 32000 && std::abs((long long)sc.match) <= 8000 && std::abs((long long)sc.mismatch) <= 8000 &&
                      std::abs((long long)sc.gap_open) + std::abs((long long)sc.gap_extend) <= 8000 && sc.gap_open <= 0 && sc.gap_extend <= 0 && max_n <= 128u * 40u;
    if (!fits) return false;
#define STITCH_SKEW16(RP_) hipLaunchKernelGGL(full_score_skew16_kernel<RP_>, dim3(n_full), dim3(64), 0, stream, d_pairs, d_which, sc, d_reads, d_contigs, d_scores)
    const uint32_t rp = (max_n + 127) / 128;
    if (rp <= 8) STITCH_SKEW16(8);
    else if (rp <= 16) STITCH_SKEW16(16);
    else if (rp <= 24) STITCH_SKEW16(24);
    else if (rp <= 32) STITCH_SKEW16(32);
    else STITCH_SKEW16(40);
#undef STITCH_SKEW16
    return true;
}

}  // namespace stitch
