// Banded Smith-Waterman score of (read, target strand) pairs for the pre-alignment filter (prealign.h).  One wavefront per
// pair, columns in order, the band rows of a column 64 at a time; the vertical (insertion) chain of a column is a
// prefix maximum: with T = max(0, diagonal, deletion), I(i) = go + ge*i + max_{k<i}(T(k) - ge*k) over the in-band rows
// above i (an insertion opened from an insertion never beats extending it because go <= 0).  Only the score is needed,
// so there is no traceback; H and D of the previous column live in global memory (a few hundred KB per pair, L2
// resident) and the band ranges decide which of their entries are valid.  The kernel is latency-bound per pair and runs
// thousands of pairs at once.
#include <hip/hip_runtime.h>

#include <climits>

#include "dp_core.h"
#include "prealign.h"

namespace stitch {

namespace {
constexpr long long NO_KEY = LLONG_MIN / 4;

__device__ __forceinline__ long long shfl_up_ll(long long v, int d) {
    const int lo = __shfl_up((int)(unsigned)(unsigned long long)v, d, 64), hi = __shfl_up((int)(unsigned)((unsigned long long)v >> 32), d, 64);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ int32_t floor_min(long long v) { return v < (long long)MIN_SCORE ? MIN_SCORE : (int32_t)v; }
}  // namespace

__global__ __launch_bounds__(64) void banded_score_kernel(const BandPair* __restrict__ pairs, BandScoring sc, const uint8_t* __restrict__ reads,
                                                          const uint8_t* __restrict__ contigs, const uint16_t* __restrict__ bands,
                                                          int32_t* __restrict__ state, int32_t* __restrict__ scores) {
    const BandPair P = pairs[blockIdx.x];
    const int lane = threadIdx.x;
    const uint32_t m = P.m, n = P.n;
    const uint8_t* q = reads + P.q_off; const uint8_t* t = contigs + P.t_off;
    const uint16_t* lo = bands + P.band_off; const uint16_t* hi = lo + (n + 1);
    int32_t* H0 = state + P.state_off; int32_t* H1 = H0 + (m + 1); int32_t* D = H1 + (m + 1);
    const long long go = sc.gap_open, ge = sc.gap_extend;
    int32_t best = 0;
    for (uint32_t j = 1; j <= n; ++j) {
        const uint32_t r0 = max((uint32_t)lo[j], 1u), r1 = min((uint32_t)hi[j], m + 1);
        if (r0 >= r1) continue;
        const uint32_t plo = lo[j - 1], phi = hi[j - 1];
        const uint8_t tj = t[j - 1];
        const int32_t* Hp = (j & 1) ? H0 : H1; int32_t* Hc = (j & 1) ? H1 : H0;
        long long carry = r0 == 1 ? 0 : NO_KEY;                      // row 0 holds H = 0: T(0) - ge*0
        for (uint32_t base = r0; base < r1; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            const bool valid = i < r1;
            int32_t T = 0, d = MIN_SCORE; long long key = NO_KEY;
            if (valid) {
                int32_t hd, hl, dl;
                if (j == 1) { hd = 0; hl = 0; dl = MIN_SCORE; }
                else {
                    hd = i == 1 ? 0 : ((i - 1 >= plo && i - 1 < phi) ? Hp[i - 1] : MIN_SCORE);
                    const bool in = i >= plo && i < phi;
                    hl = in ? Hp[i] : MIN_SCORE; dl = in ? D[i] : MIN_SCORE;
                }
                d = floor_min(max((long long)dl + ge, (long long)hl + go + ge));
                const int32_t s = q[i - 1] == tj ? sc.match : sc.mismatch;
                T = max(max(floor_min((long long)hd + s), d), 0);
                key = (long long)T - ge * (long long)i;
            }
            long long incl = key;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) { const long long o = shfl_up_ll(incl, dd); if (lane >= dd && o > incl) incl = o; }
            long long pre = shfl_up_ll(incl, 1); if (lane == 0) pre = NO_KEY;
            if (carry > pre) pre = carry;
            if (valid) {
                const int32_t I = pre == NO_KEY ? MIN_SCORE : floor_min(pre + go + ge * (long long)i);
                const int32_t h = max(T, I);
                Hc[i] = h; D[i] = d;
                best = max(best, h);
            }
            const int lo32 = __shfl((int)(unsigned)(unsigned long long)incl, 63, 64), hi32 = __shfl((int)(unsigned)((unsigned long long)incl >> 32), 63, 64);
            const long long tail = (long long)(((unsigned long long)(unsigned)hi32 << 32) | (unsigned)lo32);
            if (tail > carry) carry = tail;
        }
        // the next column reads this one: make the stores of all lanes visible to the wave's later loads
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) best = max(best, __shfl_xor(best, dd, 64));
    if (lane == 0) scores[blockIdx.x] = best;
}

void launch_banded_scores(const BandPair* d_pairs, uint32_t n_pairs, const BandScoring& sc, const uint8_t* d_reads, const uint8_t* d_contigs,
                          const uint16_t* d_bands, int32_t* d_state, int32_t* d_scores, hipStream_t stream) {
    if (n_pairs) hipLaunchKernelGGL(banded_score_kernel, dim3(n_pairs), dim3(64), 0, stream, d_pairs, sc, d_reads, d_contigs, d_bands, d_state, d_scores);
}

}  // namespace stitch
