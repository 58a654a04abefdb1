// The band of a (read, target strand) pair drawn ON THE DEVICE from the backbone's pieces (prealign.h BandElem: what
// prealign.cpp rasterise_band draws on the host, and what the oracle draws point by point), and the choice of the score kernel.
//
// One wavefront per pair; lo / hi of every column in LDS (32-bit words).  The pieces are drawn one after the other, the lanes
// dealing the columns a piece touches among themselves: a column's rows under one piece have a closed form, so a column is
// written once per piece and the wavefront's LDS operations, which execute in order, need no atomics.
//   diagonal piece (r, c, len): column cc is within w of the points t in [max(cc - c - w, 0), min(cc - c + w, len)]  -> rows
//       [r + t0 - w, r + t1 + w] (clipped to the matrix), exactly rasterise_band's add_diag;
//   gap piece (ai, aj, gi, gj), points s = 1 .. steps - 1 at (ai + gi s / steps, aj + gj s / steps): both coordinates grow with s,
//       so the points within w columns of cc are a range s_lo .. s_hi found by two divisions, and the rows are those of its ends.
// Then the band goes to global memory as uint16 (the score kernels' format) and the pair's class is decided: every column fits
// the register window and the first row never decreases -> BAND_CLASS_WINDOW; some column taller than the LDS ring -> _TALL;
// else _RING; the classes are counted, so that the host launches only the score kernels that have pairs.  tests/test_prealign.py compares the band drawn here with the oracle's, column by column.
#include <hip/hip_runtime.h>

#include "prealign.h"

namespace stitch {

namespace {
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, dd, 64));
    return v;
}
}  // namespace

constexpr uint32_t BAND_DEVICE_MAX_COLS = 8192;      // columns 0 .. n in 64 KB of LDS

__global__ __launch_bounds__(64) void band_draw_kernel(const BandPair* __restrict__ pairs, const uint32_t* __restrict__ which, const BandElem* __restrict__ elems,
                                                       uint32_t w_, uint32_t ring_rows, uint32_t window_rows, uint16_t* __restrict__ bands, uint32_t* __restrict__ cls,
                                                       uint32_t* __restrict__ class_counts) {
    extern __shared__ uint32_t band_lds[];
    const uint32_t pid = which[blockIdx.x];
    const BandPair P = pairs[pid];
    const int lane = threadIdx.x;
    const long m = P.m, n = P.n, w = w_;
    uint32_t* lo = band_lds; uint32_t* hi = band_lds + (P.n + 1);
    const bool whole = P.n_elem == 0;                                 // no backbone: the band is the full matrix
    for (uint32_t c = lane; c <= P.n; c += 64) { lo[c] = whole ? 0u : (uint32_t)(m + 1); hi[c] = whole ? (uint32_t)(m + 1) : 0u; }
    // (a column one lane widened for one piece is another lane's column in the next piece: the workgroup is one wave, the barrier
    // costs nothing, and the order no longer rests on how one wave's LDS operations happen to retire)
    __syncthreads();
    for (uint32_t e = 0; e < P.n_elem; ++e, __syncthreads()) {
        const BandElem E = elems[P.elem_off + e];
        if (E.d < 0) {
            const long r = E.a, c = E.b, len = E.c;
            const long c0 = max(c - w, 0l), c1 = min(c + len + w, n);
            for (long cc = c0 + lane; cc <= c1; cc += 64) {
                const long t0 = max(cc - c - w, 0l), t1 = min(cc - c + w, len);
                if (t0 > t1) continue;
                const uint32_t r0 = (uint32_t)max(r + t0 - w, 0l), r1 = (uint32_t)(min(r + t1 + w, m) + 1);
                lo[cc] = min(lo[cc], r0); hi[cc] = max(hi[cc], r1);
            }
        } else {
            const long ai = E.a, aj = E.b, gi = E.c, gj = E.d, steps = max(gi, gj);
            const long c0 = max(aj - w, 0l), c1 = min(aj + gj + w, n);
            for (long cc = c0 + lane; cc <= c1; cc += 64) {
                // points s with aj + gj s / steps in [cc - w, cc + w]
                const long d = cc - w - aj, f = cc + w - aj;
                if (f < 0) continue;
                long s_lo = 1, s_hi = steps - 1;
                if (d > 0) { if (gj == 0) continue; s_lo = max(s_lo, (d * steps + gj - 1) / gj); }
                if (gj > 0) s_hi = min(s_hi, ((f + 1) * steps + gj - 1) / gj - 1);
                if (s_lo > s_hi) continue;
                const long ra = ai + gi * s_lo / steps, rb = ai + gi * s_hi / steps;
                const uint32_t r0 = (uint32_t)max(ra - w, 0l), r1 = (uint32_t)(min(rb + w, m) + 1);
                lo[cc] = min(lo[cc], r0); hi[cc] = max(hi[cc], r1);
            }
        }
    }
    // out, and the class
    uint16_t* glo = bands + P.band_off; uint16_t* ghi = glo + (P.n + 1);
    uint32_t tall = 0, bad = 0, run_max = 0;                       // run_max: largest first row of the columns before this block
    for (uint32_t c0 = 0; c0 <= P.n; c0 += 64) {
        const uint32_t c = c0 + (uint32_t)lane;
        uint32_t l = 0, h = 0;
        if (c <= P.n) { l = lo[c]; h = hi[c]; glo[c] = (uint16_t)l; ghi[c] = (uint16_t)h; }
        if (h > l && h - l > ring_rows) tall = 1;
        const uint32_t r0 = max(l, 1u), r1 = min(h, (uint32_t)m + 1);
        const bool rows = c >= 1 && c <= P.n && r0 < r1;
        if (rows && r1 - ((r0 - 1) & ~3u) > window_rows) bad = 1;
        // the first row must not decrease: against the largest one of the columns before (earlier blocks, earlier lanes)
        uint32_t incl = rows ? r0 : 0u;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, dd, 64); if (lane >= dd) incl = max(incl, o); }
        uint32_t before = (uint32_t)__shfl_up((int)incl, 1, 64); if (lane == 0) before = 0;
        before = max(before, run_max);
        if (rows && r0 < before) bad = 1;
        run_max = max(run_max, (uint32_t)__shfl((int)incl, 63, 64));
    }
    tall = wave_max_u32(tall); bad = wave_max_u32(bad);
    if (lane == 0) {
        const uint32_t k = !bad ? BAND_CLASS_WINDOW : tall ? BAND_CLASS_TALL : BAND_CLASS_RING;
        cls[pid] = k;
        if (class_counts) atomicAdd(class_counts + k, 1u);          // (the host launches only the score kernels that have pairs)
    }
}

uint32_t band_device_max_cols() { return BAND_DEVICE_MAX_COLS; }
// `window` false: no pair is offered to the register-window kernel (its scoring range does not hold, or it is switched off)
void launch_band_draw(const BandPair* d_pairs, const uint32_t* d_which, uint32_t n_pairs, uint32_t max_n, const BandElem* d_elems, uint32_t w, uint32_t ring_rows,
                      bool window, uint16_t* d_bands, uint32_t* d_cls, uint32_t* d_class_counts, hipStream_t stream) {
    if (!n_pairs) return;
    const size_t lds = 8ull * (max_n + 1);
    (void)hipFuncSetAttribute((const void*)band_draw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(band_draw_kernel, dim3(n_pairs), dim3(64), lds, stream, d_pairs, d_which, d_elems, w, ring_rows, window ? 256u : 0u, d_bands, d_cls, d_class_counts);
}

}  // namespace stitch
