// Banded Smith-Waterman score of (read, target strand) pairs for the pre-alignment filter (prealign.h).  One wavefront per
// pair, columns in order, the band rows of a column 64 at a time; the vertical (insertion) chain of a column is a
// prefix maximum: with T = max(0, diagonal, deletion), I(i) = go + ge*i + max_{k<i}(T(k) - ge*k) over the in-band rows
// above i (an insertion opened from an insertion never beats extending it because go <= 0).  Only the score is needed,
// so there is no traceback; H and D of the previous column live in global memory (a few hundred KB per pair, L2
// resident) and the band ranges decide which of their entries are valid.  The kernel is latency-bound per pair and runs
// thousands of pairs at once.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdlib>

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

__global__ __launch_bounds__(64) void banded_score_kernel(const BandPair* __restrict__ pairs, const uint32_t* __restrict__ which, BandScoring sc,
                                                          const uint8_t* __restrict__ reads, const uint8_t* __restrict__ contigs,
                                                          const uint16_t* __restrict__ bands, int32_t* __restrict__ state, int32_t* __restrict__ scores,
                                                          const uint32_t* __restrict__ cls, uint32_t my_class) {
    const uint32_t pid = which[blockIdx.x];
    if (cls && cls[pid] != my_class) return;                           // (the device drew the bands and chose the kernels: prealign_band.hip)
    const BandPair P = pairs[pid];
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
    if (lane == 0) scores[pid] = best;
}

// The banded score in EVERY clipping mode and for reads of any length (32-bit band ranges): the filter of `--pre-align` follows `--mode`
// in the reference (Options::banded_scoring, aligners/mod.rs:133-141), and its reads have no length limit (:246-284).  Same organisation
// as banded_score_kernel above — one wavefront per pair, columns in order, H and D of the previous column in global memory — with bio's
// four clip penalties (0 = free, MIN_SCORE = forbidden; x = read, y = target).  PARITY UNPINNED like the local filter: crate bio 1.1.0 is not
// in the reference tree and no reference test touches it; what is restated is the documented meaning of the penalties ("xclip_prefix:
// the penalty for clipping a prefix of x", and so on for the other three), with the band of the local filter.  Definitions this
// repository chose where the description is silent (the oracle, oracle/prealign_oracle.cpp, states the same): row 0 and column 0 are
// always in the band and hold  S(0, j) = max(yclip_prefix, go + ge j),  S(i, 0) = max(xclip_prefix, go + ge i)  (the skipped prefix is
// clipped or gapped); a cell may also start the alignment,  xclip_prefix + S(0, j)  or  yclip_prefix + go + ge i;  the score is the best
// in-band cell plus what the rest of both sequences costs there,  max(xclip_suffix, go + ge (m - i))  for i < m and
// max(yclip_suffix, go + ge (n - j))  for j < n, or the empty alignment  max(xp, xs, go + ge m) + max(yp, ys, go + ge n).  With all four
// penalties 0 this is the local filter's score cell for cell.
__global__ __launch_bounds__(64) void banded_score_general_kernel(const BandPair32* __restrict__ pairs, uint32_t n_pairs, BandScoringClip sc,
                                                                  const uint8_t* __restrict__ reads, const uint8_t* __restrict__ contigs,
                                                                  const uint32_t* __restrict__ bands, int32_t* __restrict__ state, int32_t* __restrict__ scores) {
    const uint32_t pid = blockIdx.x;
    if (pid >= n_pairs) return;
    const BandPair32 P = pairs[pid];
    const int lane = threadIdx.x;
    const uint32_t m = P.m, n = P.n;
    const uint8_t* q = reads + P.q_off; const uint8_t* t = contigs + P.t_off;
    const uint32_t* lo = bands + P.band_off; const uint32_t* hi = lo + (n + 1);
    int32_t* H0 = state + P.state_off; int32_t* H1 = H0 + (m + 1); int32_t* D = H1 + (m + 1);
    const long long go = sc.gap_open, ge = sc.gap_extend;
    const long long xp = sc.xclip_prefix, xs = sc.xclip_suffix, yp = sc.yclip_prefix, ys = sc.yclip_suffix;
    auto row0 = [&](uint32_t j) -> long long { return j == 0 ? 0ll : max(yp, go + ge * (long long)j); };        // S(0, j)
    auto col0 = [&](uint32_t i) -> long long { return i == 0 ? 0ll : max(xp, go + ge * (long long)i); };        // S(i, 0)
    long long best = max(max(xp, xs), m ? go + ge * (long long)m : 0ll) + max(max(yp, ys), n ? go + ge * (long long)n : 0ll);      // nothing aligned
    if (m == 0) best = max(best, max(max(yp, ys), n ? go + ge * (long long)n : 0ll));
    for (uint32_t j = 1; j <= n; ++j) {
        const uint32_t r0 = max(lo[j], 1u), r1 = min(hi[j], m + 1);
        if (r0 >= r1) continue;
        const uint32_t plo = lo[j - 1], phi = hi[j - 1];
        const uint8_t tj = t[j - 1];
        const int32_t* Hp = (j & 1) ? H0 : H1; int32_t* Hc = (j & 1) ? H1 : H0;
        const long long s0j = row0(j), s0jm1 = row0(j - 1);
        const long long ry = j == n ? 0ll : max(ys, go + ge * (long long)(n - j));
        long long carry = r0 == 1 ? s0j : NO_KEY;                    // row 0 opens the column's insertion chain with S(0, j): T(0) - ge * 0
        for (uint32_t base = r0; base < r1; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            const bool valid = i < r1;
            long long T = MIN_SCORE; int32_t d = MIN_SCORE; long long key = NO_KEY;
            if (valid) {
                long long hd, hl, dl;
                if (j == 1) { hd = col0(i - 1); hl = col0(i); dl = MIN_SCORE; }
                else {
                    hd = i == 1 ? s0jm1 : ((i - 1 >= plo && i - 1 < phi) ? (long long)Hp[i - 1] : (long long)MIN_SCORE);
                    const bool in = i >= plo && i < phi;
                    hl = in ? (long long)Hp[i] : (long long)MIN_SCORE; dl = in ? (long long)D[i] : (long long)MIN_SCORE;
                }
                d = floor_min(max(dl + ge, hl + go + ge));
                const int32_t s = q[i - 1] == tj ? sc.match : sc.mismatch;
                T = max((long long)floor_min(hd + s), (long long)d);
                T = max(T, (long long)floor_min(xp + s0j));                                   // the read's first i bases clipped
                T = max(T, (long long)floor_min(yp + go + ge * (long long)i));                // the target's first j bases clipped, the read's inserted
                key = T - ge * (long long)i;
            }
            long long incl = key;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) { const long long o = shfl_up_ll(incl, dd); if (lane >= dd && o > incl) incl = o; }
            long long pre = shfl_up_ll(incl, 1); if (lane == 0) pre = NO_KEY;
            if (carry > pre) pre = carry;
            if (valid) {
                const int32_t I = pre == NO_KEY ? MIN_SCORE : floor_min(pre + go + ge * (long long)i);
                const int32_t h = (int32_t)max(T, (long long)I);
                Hc[i] = h; D[i] = d;
                const long long rx = i == m ? 0ll : max(xs, go + ge * (long long)(m - i));
                if (h > MIN_SCORE) best = max(best, (long long)h + rx + ry);
            }
            const int lo32 = __shfl((int)(unsigned)(unsigned long long)incl, 63, 64), hi32 = __shfl((int)(unsigned)((unsigned long long)incl >> 32), 63, 64);
            const long long tail = (long long)(((unsigned long long)(unsigned)hi32 << 32) | (unsigned)lo32);
            if (tail > carry) carry = tail;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) { const long long o = (long long)(((unsigned long long)(unsigned)__shfl_xor((int)(unsigned)((unsigned long long)best >> 32), dd, 64) << 32) | (unsigned)__shfl_xor((int)(unsigned)(unsigned long long)best, dd, 64)); best = max(best, o); }
    if (lane == 0) scores[pid] = floor_min(best);
}

// The same recurrence with the column state on chip: H (two columns) and D live in LDS rings of BAND_RING rows indexed by
// row mod BAND_RING, which holds any band whose columns are at most BAND_RING rows tall (the host sends taller ones to
// the kernel above); the read's bases are staged in LDS once; the band ranges and target bases of 64 columns are
// fetched together and broadcast from registers.  Rows are tied to lanes by row mod 64, so a column is processed in
// 64-aligned blocks of rows and the insertion chain is a DPP prefix maximum in 32 bits (keys T - ge*i >= 0, "none" = -1;
// the host checks that they fit).  One wavefront per pair, no barriers: a wavefront's LDS operations execute in order.
constexpr uint32_t BAND_RING = 512;
constexpr size_t BAND_LDS_MAX = 64 << 10;
__device__ __forceinline__ int32_t dpp_max_shr(int32_t v, int32_t none) {            // inclusive prefix maximum over the 64 lanes
#define STITCH_DPP_STEP(CTRL, ROWMASK) { const int32_t o = __builtin_amdgcn_update_dpp(none, v, CTRL, ROWMASK, 0xF, false); v = o > v ? o : v; }
    STITCH_DPP_STEP(0x111, 0xF) STITCH_DPP_STEP(0x112, 0xF) STITCH_DPP_STEP(0x114, 0xF) STITCH_DPP_STEP(0x118, 0xF)      // row_shr 1, 2, 4, 8
    STITCH_DPP_STEP(0x142, 0xA) STITCH_DPP_STEP(0x143, 0xC)                                                          // row_bcast 15, 31
#undef STITCH_DPP_STEP
    return v;
}
__global__ __launch_bounds__(64) void banded_score_lds_kernel(const BandPair* __restrict__ pairs, const uint32_t* __restrict__ which, BandScoring sc,
                                                              const uint8_t* __restrict__ reads, const uint8_t* __restrict__ contigs,
                                                              const uint16_t* __restrict__ bands, int32_t* __restrict__ scores, const uint32_t* __restrict__ cls) {
    extern __shared__ int32_t band_lds[];
    const uint32_t pid = which[blockIdx.x];
    if (cls && cls[pid] != BAND_CLASS_RING) return;
    const BandPair P = pairs[pid];
    const int lane = threadIdx.x;
    const uint32_t m = P.m, n = P.n;
    const uint8_t* q = reads + P.q_off; const uint8_t* t = contigs + P.t_off;
    const uint16_t* lo = bands + P.band_off; const uint16_t* hi = lo + (n + 1);
    int32_t* H0 = band_lds; int32_t* H1 = H0 + BAND_RING; int32_t* D = H1 + BAND_RING;
    uint8_t* qs = (uint8_t*)(D + BAND_RING);
    for (uint32_t i = lane; i < m; i += 64) qs[i] = q[i];
    const int32_t go = sc.gap_open, ge = sc.gap_extend, NONE = -1;
    constexpr uint32_t RM = BAND_RING - 1;
    int32_t best = 0;
    uint32_t plo = lo[0], phi = hi[0];                               // raw range of column j - 1
    uint32_t vlo = 0, vhi = 0, vt = 0;
    for (uint32_t j = 1; j <= n; ++j) {
        const uint32_t jl = (j - 1) & 63u;
        if (jl == 0) {                                                // ranges and target bases of columns j .. j + 63
            const uint32_t c = j + (uint32_t)lane;
            vlo = c <= n ? lo[c] : 0u; vhi = c <= n ? hi[c] : 0u; vt = c <= n ? t[c - 1] : 0u;
        }
        const uint32_t clo = (uint32_t)__builtin_amdgcn_readlane((int)vlo, (int)jl), chi = (uint32_t)__builtin_amdgcn_readlane((int)vhi, (int)jl);
        const uint32_t tj = (uint32_t)__builtin_amdgcn_readlane((int)vt, (int)jl);
        const uint32_t r0 = max(clo, 1u), r1 = min(chi, m + 1);
        if (r0 < r1) {
            const int32_t* Hp = (j & 1) ? H0 : H1; int32_t* Hc = (j & 1) ? H1 : H0;
            int32_t carry = r0 == 1 ? 0 : NONE;                       // row 0 holds H = 0: T(0) - ge*0
            for (uint32_t b = r0 >> 6; b <= (r1 - 1) >> 6; ++b) {
                const uint32_t i = (b << 6) + (uint32_t)lane;
                const bool valid = i >= r0 && i < r1;
                int32_t T = 0, d = MIN_SCORE, key = NONE;
                if (valid) {
                    int32_t hd, hl, dl;
                    if (j == 1) { hd = 0; hl = 0; dl = MIN_SCORE; }
                    else {
                        hd = i == 1 ? 0 : ((i - 1 >= plo && i - 1 < phi) ? Hp[(i - 1) & RM] : MIN_SCORE);
                        const bool in = i >= plo && i < phi;
                        hl = in ? Hp[i & RM] : MIN_SCORE; dl = in ? D[i & RM] : MIN_SCORE;
                    }
                    d = max(max(dl + ge, hl + go + ge), MIN_SCORE);
                    const int32_t s = (uint32_t)qs[i - 1] == tj ? sc.match : sc.mismatch;
                    T = max(max(max(hd + s, MIN_SCORE), d), 0);
                    key = T - ge * (int32_t)i;
                }
                const int32_t incl = dpp_max_shr(key, NONE);
                int32_t pre = __builtin_amdgcn_update_dpp(NONE, incl, 0x138, 0xF, 0xF, false);     // wave_shr 1: the lanes before this one
                pre = carry > pre ? carry : pre;
                if (valid) {
                    const int32_t I = pre < 0 ? MIN_SCORE : max(pre + go + ge * (int32_t)i, MIN_SCORE);
                    const int32_t h = max(T, I);
                    Hc[i & RM] = h; D[i & RM] = d;
                    best = max(best, h);
                }
                const int32_t tail = __builtin_amdgcn_readlane(incl, 63);
                carry = tail > carry ? tail : carry;
            }
        }
        plo = clo; phi = chi;
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) best = max(best, __shfl_xor(best, dd, 64));
    if (lane == 0) scores[pid] = best;
}

// Pairs without a single k-mer match are scored over the FULL matrix (the band is everything): ~5 % of the pairs of a random
// 10 kb read against 5 kb targets, 50 M cells each.  One workgroup per pair, H and D of all rows in LDS, the rows dealt in
// contiguous slices to the threads: per column a thread (1) computes D and T = max(0, diagonal, D) of its rows and the
// maximum of T - ge*i over them, (2) receives the maximum over the slices above it (workgroup scan), (3) finishes I and H.
constexpr int FULL_THREADS = 256;
constexpr uint32_t FULL_MAX_ROWS = 18000;             // 8 bytes of LDS per row
__global__ __launch_bounds__(FULL_THREADS) void full_score_kernel(const BandPair* __restrict__ pairs, const uint32_t* __restrict__ which, BandScoring sc,
                                                                   const uint8_t* __restrict__ reads, const uint8_t* __restrict__ contigs,
                                                                   int32_t* __restrict__ scores) {
    extern __shared__ int32_t lds[];
    const uint32_t pid = which[blockIdx.x];
    const BandPair P = pairs[pid];
    const uint32_t m = P.m, n = P.n;
    const uint8_t* q = reads + P.q_off; const uint8_t* t = contigs + P.t_off;
    int32_t* H = lds; int32_t* D = lds + (m + 1);                       // H[i], D[i] of the previous column, rows 0..m
    __shared__ long long part[FULL_THREADS];
    __shared__ int32_t red[FULL_THREADS];
    const int tid = threadIdx.x;
    const uint32_t per = (m + FULL_THREADS - 1) / FULL_THREADS;
    const uint32_t i0 = 1 + (uint32_t)tid * per, i1 = min(i0 + per, m + 1);          // rows [i0, i1)
    const long long go = sc.gap_open, ge = sc.gap_extend;
    for (uint32_t i = tid; i <= m; i += FULL_THREADS) { H[i] = 0; D[i] = MIN_SCORE; }    // column 0: H = 0 (free clipping), no deletion yet
    __syncthreads();
    int32_t best = 0;
    for (uint32_t j = 1; j <= n; ++j) {
        const uint8_t tj = t[j - 1];
        int32_t diag = i0 <= m ? H[i0 - 1] : 0;                          // H(i0-1, j-1), read before anybody overwrites it
        __syncthreads();
        long long mx = NO_KEY;
        for (uint32_t i = i0; i < i1; ++i) {
            const int32_t hl = H[i];
            const int32_t d = floor_min(max((long long)D[i] + ge, (long long)hl + go + ge));
            const int32_t s = q[i - 1] == tj ? sc.match : sc.mismatch;
            const int32_t T = max(max(floor_min((long long)diag + s), d), 0);
            diag = hl;
            D[i] = d; H[i] = T;                                           // H holds T until step 3
            const long long key = (long long)T - ge * (long long)i;
            mx = key > mx ? key : mx;
        }
        part[tid] = mx;
        __syncthreads();
        // exclusive maximum over the slices above (row 0, H = 0, is above everything: key 0)
        long long carry = 0;
        for (int u = 0; u < tid; ++u) { const long long v = part[u]; carry = v > carry ? v : carry; }
        for (uint32_t i = i0; i < i1; ++i) {
            const int32_t T = H[i];
            const int32_t I = floor_min(carry + go + ge * (long long)i);
            const int32_t h = max(T, I);
            H[i] = h; best = max(best, h);
            const long long key = (long long)T - ge * (long long)i;
            carry = key > carry ? key : carry;
        }
        __syncthreads();
    }
    red[tid] = best;
    __syncthreads();
    for (int d = FULL_THREADS / 2; d >= 1; d >>= 1) { if (tid < d) red[tid] = max(red[tid], red[tid + d]); __syncthreads(); }
    if (tid == 0) scores[pid] = red[0];
}

// The same with the rows in registers (PER rows per thread, m <= 256 * PER) and 32-bit arithmetic: no LDS traffic in the
// row loops, the scan of the slices' maxima is a wave prefix maximum plus one LDS exchange between the four waves.  In a
// full local matrix H >= 0 everywhere, so D and I stay above go + ge*(m+1) and nothing needs clamping; the host only sends
// pairs here whose scores and keys fit 31 bits (launch_full_scores).
template <int PER>
__global__ __launch_bounds__(FULL_THREADS) void full_score_reg_kernel(const BandPair* __restrict__ pairs, const uint32_t* __restrict__ which, BandScoring sc,
                                                                       const uint8_t* __restrict__ reads, const uint8_t* __restrict__ contigs,
                                                                       int32_t* __restrict__ scores) {
    static_assert(PER % 4 == 0, "rows per thread");
    const uint32_t pid = which[blockIdx.x];
    const BandPair P = pairs[pid];
    const uint32_t m = P.m, n = P.n;
    const uint8_t* q = reads + P.q_off; const uint8_t* t = contigs + P.t_off;
    __shared__ int32_t wave_tot[FULL_THREADS / 64];
    __shared__ int32_t wave_last[2][FULL_THREADS / 64];      // H of a wave's last row, by column parity
    __shared__ int32_t red[FULL_THREADS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t i0 = 1 + (uint32_t)tid * PER;
    const int32_t ge = sc.gap_extend, goe = sc.gap_open + sc.gap_extend;
    int32_t H[PER], D[PER]; uint32_t qw[PER / 4];            // the thread's read bases, four per word (0 beyond the read: rows > m)
#pragma unroll
    for (int u = 0; u < PER; ++u) { H[u] = 0; D[u] = MIN_SCORE / 2; }
#pragma unroll
    for (int v = 0; v < PER / 4; ++v) {
        uint32_t w = 0;
        for (int b = 0; b < 4; ++b) { const uint32_t i = i0 + 4 * v + b; w |= (uint32_t)(i <= m ? q[i - 1] : 0) << (8 * b); }
        qw[v] = w;
    }
    const int32_t k0 = -ge * (int32_t)i0;                    // key(i) = T(i) - ge*i = T + k0 - ge*u
    const int32_t c0 = goe + ge * ((int32_t)i0 - 1);         // I(i) = carry + go + ge*i = carry + c0 + ge*u
    if (lane == 63) wave_last[0][wave] = 0;
    __syncthreads();
    int32_t best = 0;
    for (uint32_t j = 1; j <= n; ++j) {
        const uint32_t tj = t[j - 1];
        int32_t diag = __shfl_up(H[PER - 1], 1, 64);          // H(i0 - 1, j - 1): the last row of the thread before (row 0 for thread 0)
        if (lane == 0) diag = wave == 0 ? 0 : wave_last[(j - 1) & 1][wave - 1];
        int32_t mx = INT32_MIN;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int32_t hl = H[u];
            const int32_t d = max(D[u] + ge, hl + goe);
            const int32_t s = ((qw[u >> 2] >> (8 * (u & 3))) & 0xFFu) == tj ? sc.match : sc.mismatch;
            const int32_t T = max(max(diag + s, d), 0);
            diag = hl;
            D[u] = d; H[u] = T;                                // rows beyond m compute harmless values: their q byte is 0 and nothing reads them
            if (i0 + u <= m) mx = max(mx, T + (k0 - ge * u));
        }
        int32_t incl = mx;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const int32_t o = __shfl_up(incl, dd, 64); if (lane >= dd) incl = max(incl, o); }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int32_t carry = __shfl_up(incl, 1, 64); if (lane == 0) carry = 0;        // row 0 (key 0) is above everything
        carry = max(carry, 0);
        for (int w = 0; w < wave; ++w) carry = max(carry, wave_tot[w]);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (i0 + u <= m) {
                const int32_t T = H[u];
                const int32_t h = max(T, carry + (c0 + ge * u));
                H[u] = h; best = max(best, h);
                carry = max(carry, T + (k0 - ge * u));
            }
        }
        if (lane == 63) wave_last[j & 1][wave] = H[PER - 1];
        __syncthreads();
    }
    red[tid] = best;
    __syncthreads();
    for (int d = FULL_THREADS / 2; d >= 1; d >>= 1) { if (tid < d) red[tid] = max(red[tid], red[tid + d]); __syncthreads(); }
    if (tid == 0) scores[pid] = red[0];
}

void launch_full_scores(const BandPair* d_pairs, const uint32_t* d_which, uint32_t n_full, uint32_t max_m, const BandScoring& sc, const uint8_t* d_reads,
                        const uint8_t* d_contigs, int32_t* d_scores, hipStream_t stream) {
    if (!n_full) return;
    const uint32_t per = (max_m + FULL_THREADS - 1) / FULL_THREADS;
    // the register kernel works in 32 bits: scores up to match * m and keys up to |ge| * (m + 1) must fit
    const long long big = (long long)1 << 29;
    const bool small = (long long)std::abs(sc.match) * (max_m + 1) < big && (long long)std::abs(sc.gap_extend) * (max_m + 2) + std::abs(sc.gap_open) < big &&
                       std::abs((long long)sc.mismatch) < big;
#define STITCH_FULL_REG(PER_) hipLaunchKernelGGL(full_score_reg_kernel<PER_>, dim3(n_full), dim3(FULL_THREADS), 0, stream, d_pairs, d_which, sc, d_reads, d_contigs, d_scores)
    if (small && per <= 8) { STITCH_FULL_REG(8); return; }
    if (small && per <= 16) { STITCH_FULL_REG(16); return; }
    if (small && per <= 24) { STITCH_FULL_REG(24); return; }
    if (small && per <= 32) { STITCH_FULL_REG(32); return; }
    if (small && per <= 40) { STITCH_FULL_REG(40); return; }
#undef STITCH_FULL_REG
    const size_t lds = 8ull * (max_m + 1);
    (void)hipFuncSetAttribute((const void*)full_score_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(full_score_kernel, dim3(n_full), dim3(FULL_THREADS), lds, stream, d_pairs, d_which, sc, d_reads, d_contigs, d_scores);
}
uint32_t full_score_max_rows() { return FULL_MAX_ROWS; }

void launch_banded_scores(const BandPair* d_pairs, const uint32_t* d_which, uint32_t n_pairs, const BandScoring& sc, const uint8_t* d_reads, const uint8_t* d_contigs,
                          const uint16_t* d_bands, int32_t* d_state, int32_t* d_scores, const uint32_t* d_cls, uint32_t my_class, hipStream_t stream) {
    if (n_pairs) hipLaunchKernelGGL(banded_score_kernel, dim3(n_pairs), dim3(64), 0, stream, d_pairs, d_which, sc, d_reads, d_contigs, d_bands, d_state, d_scores, d_cls, my_class);
}
void launch_banded_scores_general(const BandPair32* d_pairs, uint32_t n_pairs, const BandScoringClip& sc, const uint8_t* d_reads, const uint8_t* d_contigs,
                                  const uint32_t* d_bands, int32_t* d_state, int32_t* d_scores, hipStream_t stream) {
    if (n_pairs) hipLaunchKernelGGL(banded_score_general_kernel, dim3(n_pairs), dim3(64), 0, stream, d_pairs, n_pairs, sc, d_reads, d_contigs, d_bands, d_state, d_scores);
}
// pairs whose band columns are at most banded_ring_rows() tall and whose reads are at most max_m long; false = not applicable
// (scores or keys do not fit 32 bits, or the read does not fit in LDS): the caller uses launch_banded_scores for them too
uint32_t banded_ring_rows() { return BAND_RING; }
bool launch_banded_scores_lds(const BandPair* d_pairs, const uint32_t* d_which, uint32_t n_pairs, uint32_t max_m, const BandScoring& sc, const uint8_t* d_reads,
                              const uint8_t* d_contigs, const uint16_t* d_bands, int32_t* d_scores, const uint32_t* d_cls, hipStream_t stream) {
    const long long big = (long long)1 << 29;
    const bool small = (long long)std::abs(sc.match) * (max_m + 1) < big && (long long)std::abs(sc.gap_extend) * (max_m + 2) + std::abs(sc.gap_open) < big &&
                       std::abs((long long)sc.mismatch) < big && sc.gap_extend <= 0 && sc.gap_open <= 0;
    const size_t lds = (size_t)BAND_RING * 12 + ((size_t)max_m + 3) / 4 * 4;
    if (!small || lds > BAND_LDS_MAX) return false;
    if (n_pairs) hipLaunchKernelGGL(banded_score_lds_kernel, dim3(n_pairs), dim3(64), lds, stream, d_pairs, d_which, sc, d_reads, d_contigs, d_bands, d_scores, d_cls);
    return true;
}

}  // namespace stitch
