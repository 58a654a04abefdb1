// Kernel 1b — the DP fill specialised for AlignmentMode::Local (the CLI default and the benchmark mode).
//
// Same recurrence, tie-breaks and results as fill_kernel.hip (and therefore as
// fg-stitch-lib/src/align/aligners/single_contig_aligner.rs:188-451 + multi_contig_aligner.rs:264-347), with the
// simplifications Local mode allows and an HBM-leaner, VALU-leaner inner loop:
//   * move selection by ordered integer keys (dp_core.h, row_phase_a_key): the reference's chain of `>` tests
//     becomes a handful of max operations, and the winning priority is the traceback move code;
//   * row state is 8 bytes (S<<16 | S.len, D<<16 | D.len) instead of 16, read and written as 16-byte vectors;
//   * 8 rows per lane (512-row tiles), full tiles run a check-free instance, only a contig's last tile handles
//     ragged ends and row m;
//   * the next tile's state is prefetched into registers while the current one is computed;
//   * the insertion scan and the neighbour hand-offs use DPP row shifts / broadcasts instead of LDS permutes;
//   * the y-suffix trackers Sn/Ly (:431-447) are only touched for cells that reach the contig's running maximum —
//     in Local mode only rows whose Sn equals the contig's final maximum can influence the result;
//   * the per-contig jump selection (multi_contig_aligner.rs:292-331) is done once per column by one thread per
//     contig between two barriers.
// Eligibility is decided on the host (stitch_api.cpp: local16_ok): mode local, go + ge < 0, match * n <= 32767,
// n + max contig length < 65535, penalties >= -16000.  Anything else runs the generic int32 kernel.
#include <hip/hip_runtime.h>
#include "dp_core.h"
#include "walk_core.h"

namespace stitch {

struct FillShared {
    const int32_t* S0; const uint32_t* Slen0; const int32_t* Sn0; const uint8_t* SnSet0; const uint8_t* Smove0;
    const uint32_t* lx0; const JumpBase* base0;
};

namespace {

constexpr int MAXC = 256;
constexpr int R = 8;
constexpr uint32_t TILE = 64 * R;
constexpr int DPP_ROW_SHR0 = 0x110, DPP_WAVE_SHR1 = 0x138, DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143;

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_mov(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ int from_prev_lane(int v, int lane0) { return dpp_mov<DPP_WAVE_SHR1>(lane0, v); }   // lane-1's v; lane 0 gets lane0
__device__ __forceinline__ int lane_bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ void scan_step(ScanEl& inc) {
    ScanEl o; o.key = dpp_mov<CTRL, ROW_MASK>(INT32_MIN, inc.key); o.q = dpp_mov<CTRL, ROW_MASK>(0, inc.q);
    if (o.key >= inc.key) inc = o;             // the source lane holds earlier rows: it wins ties
}
__device__ __forceinline__ void wave_scan(ScanEl& inc) {   // inclusive scan with scan_combine
    scan_step<DPP_ROW_SHR0 | 1>(inc); scan_step<DPP_ROW_SHR0 | 2>(inc); scan_step<DPP_ROW_SHR0 | 4>(inc); scan_step<DPP_ROW_SHR0 | 8>(inc);
    scan_step<DPP_BCAST15, 0xA>(inc); scan_step<DPP_BCAST31, 0xC>(inc);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ uint32_t pk16(int32_t v, uint32_t len) { return ((uint32_t)v << 16) | (len & 0xFFFFu); }

struct WaveCol {                 // wave-uniform state of one contig's column
    int32_t js; uint32_t jl;     // column's best jump (score without the match term, length)
    int32_t js1; uint32_t jl1;   // same for row 1 (circular contigs may use the end-to-start jump)
    int32_t vrun;                // contig's running maximum up to column j-1
    uint32_t m, roff, j, n;
    uint8_t q;
    int32_t upS, upT; uint32_t upSl, upTl;      // carries: S[prev][i0-1], its len; S'[curr][i0-1], its len
    ScanEl carry;
};
struct LaneAcc {                 // per-lane running records over a contig's column (rows < m)
    uint32_t xw, xrow;           // best S<<16|len and its (lowest) row: the x-suffix running max (:406-429)
    uint32_t ck;                 // max of S<<16 | (0xFFFF - row): column arg-max, lowest row (:677-697)
};
struct RowM { int32_t key; uint32_t Sl, bits, dpack; int32_t dg; };   // row m's own selection, finalised after the reduction

// One 512-row tile.  PARTIAL: the contig's last tile (rows may exceed m, and row m is held back).  LASTCOL: j == n,
// the int32 arrays the fix-up kernel reads are written as well.
template <bool PARTIAL, bool LASTCOL>
__device__ __forceinline__ void tile(const JobView& V, const DpParams& P, WaveCol& wc, LaneAcc& acc, RowM& rm, const uint4 (&cur)[4], uint2 curx,
                                     uint32_t t, int lane, uint32_t* __restrict__ st, uint8_t* __restrict__ tbcol) {
    const uint32_t i0 = t * TILE + lane * R + 1;
    const uint32_t r = wc.roff + i0 - 1;
    const uint32_t m = wc.m;
    int32_t Sp[R], Dp[R]; uint32_t Slp[R], Dlp[R]; uint32_t xb[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const uint32_t w0 = (u & 1) ? cur[u >> 1].z : cur[u >> 1].x, w1 = (u & 1) ? cur[u >> 1].w : cur[u >> 1].y;
        Sp[u] = (int32_t)(w0 >> 16); Slp[u] = w0 & 0xFFFFu;
        Dp[u] = (int32_t)w1 >> 16; Dlp[u] = w1 & 0xFFFFu;
        xb[u] = ((u < 4 ? curx.x : curx.y) >> (8 * (u & 3))) & 0xFFu;
    }
    const int32_t nS = from_prev_lane(Sp[R - 1], wc.upS); const uint32_t nSl = (uint32_t)from_prev_lane((int)Slp[R - 1], (int)wc.upSl);
    wc.upS = lane_bcast(Sp[R - 1], 63); wc.upSl = (uint32_t)lane_bcast((int)Slp[R - 1], 63);

    RowK ra[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const bool row1 = (u == 0) && (i0 == 1);
        row_phase_a_key(P, (uint8_t)xb[u], wc.q, row1 ? wc.js1 : wc.js, row1 ? wc.jl1 : wc.jl, u == 0 ? nS : Sp[u - 1], u == 0 ? nSl : Slp[u - 1],
                        Sp[u], Slp[u], Dp[u], Dlp[u], ra[u]);
    }
    // phase B: insertion scan
    const int32_t lastT = ra[R - 1].Tk >> 3;
    const int32_t nT = from_prev_lane(lastT, wc.upT); const uint32_t nTl = (uint32_t)from_prev_lane((int)ra[R - 1].Tl, (int)wc.upTl);
    wc.upT = lane_bcast(lastT, 63); wc.upTl = (uint32_t)lane_bcast((int)ra[R - 1].Tl, 63);
    const int32_t kb = P.gap_open + P.gap_extend - P.gap_extend * (int32_t)i0;     // key_i = S'(i-1) + go + ge - ge*i
    const int32_t qb = 1 - (int32_t)i0;                                             // q_i   = S'.len(i-1) + 1 - i
    ScanEl el[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        el[u].key = (u == 0 ? nT : (ra[u - 1].Tk >> 3)) + kb - P.gap_extend * u;
        el[u].q = (int32_t)(u == 0 ? nTl : ra[u - 1].Tl) + qb - u;
        if (PARTIAL && i0 + u > m) el[u].key = KEY_NEG_INF;
    }
    ScanEl inc = el[0];
#pragma unroll
    for (int u = 1; u < R; ++u) inc = scan_combine(inc, el[u]);
    wave_scan(inc);
    ScanEl run; run.key = from_prev_lane(inc.key, INT32_MIN); run.q = from_prev_lane(inc.q, 0);
    run = scan_combine(wc.carry, run);                                              // lane 0: INT32_MIN never wins -> carry
    { ScanEl last; last.key = lane_bcast(inc.key, 63); last.q = lane_bcast(inc.q, 63); wc.carry = scan_combine(wc.carry, last); }
    // phase C
    const int32_t gi0 = P.gap_extend * (int32_t)i0;
    uint32_t outw[2 * R]; uint32_t code[R]; uint32_t tk = 0;
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const uint32_t i = i0 + u;
        const uint32_t ext = run.key >= el[u].key ? 1u : 0u;
        if (!ext) run = el[u];
        const int32_t bi = run.key + gi0 + P.gap_extend * u;
        const uint32_t il = (uint32_t)(run.q + (int32_t)i0 + u);
        uint32_t Slo;
        const int32_t fk = row_phase_c_key(ra[u], bi, il, Slo);
        const int32_t So = fk >> 3;
        code[u] = ((uint32_t)fk & 7u) | (ext ? TBB_IEXT : 0u) | (ra[u].dext ? TBB_DEXT : 0u);
        const uint32_t w = pk16(So, Slo);
        outw[2 * u] = w; outw[2 * u + 1] = pk16(ra[u].bd, ra[u].dlen);
        if (!PARTIAL || i < m) {
            const bool better = w > acc.xw;                                           // rows ascend within a lane: first max wins
            acc.xw = better ? w : acc.xw; acc.xrow = better ? i : acc.xrow;
            const uint32_t ck = ((uint32_t)So << 16) | (0xFFFFu - i);
            tk = ck > tk ? ck : tk;
        }
        if (PARTIAL) {
            if (i == m) { rm.key = fk; rm.Sl = Slo; rm.bits = code[u] & (TBB_IEXT | TBB_DEXT); rm.dpack = outw[2 * u + 1]; rm.dg = ra[u].dg; }
            if (i > m) { outw[2 * u] = 0; outw[2 * u + 1] = pk16(-32768, 0); code[u] = 0; }
        }
        if (LASTCOL) { if (!PARTIAL || i <= m) { V.S[r + u] = So; V.Slen[r + u] = Slo; V.Ival[r + u] = bi; V.Ilen[r + u] = il; } }
    }
    acc.ck = tk > acc.ck ? tk : acc.ck;
    // y-suffix trackers: only cells that reach the contig's running maximum can matter (:431-447)
    if ((int32_t)(tk >> 16) >= wc.vrun) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const uint32_t i = i0 + u; const int32_t So = (int32_t)(outw[2 * u] >> 16); const uint32_t Slo = outw[2 * u] & 0xFFFFu;
            if ((!PARTIAL || i < m) && So >= wc.vrun && Slo > 0u) {
                if (So >= V.Sn[r + u]) { V.Sn[r + u] = So; V.Ly[r + u] = wc.n - wc.j; V.SnLen[r + u] = Slo; }
            }
        }
    }
    uint4* __restrict__ stw = reinterpret_cast<uint4*>(st + 2 * (size_t)r);
#pragma unroll
    for (int v = 0; v < 4; ++v) { uint4 o; o.x = outw[4 * v]; o.y = outw[4 * v + 1]; o.z = outw[4 * v + 2]; o.w = outw[4 * v + 3]; stw[v] = o; }
    uint2 tbv;
    tbv.x = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
    tbv.y = code[4] | (code[5] << 8) | (code[6] << 16) | (code[7] << 24);
    *reinterpret_cast<uint2*>(tbcol + r) = tbv;
}

}  // namespace

// G workgroups cooperate on one read: workgroup `part` owns the active contigs k = part, part+G, ... and exchanges its
// contigs' column arg-max (the next column's jump sources) with the others through 8-byte {data, tag} granules in
// global memory (agent-scope relaxed atomics: the tag travels with the data, so no fence is needed; double-buffered by
// column parity).  All G workgroups of a read must be resident at once: the host keeps the grid <= the CU count.
__global__ __launch_bounds__(512) void fill_local16_kernel(const JobView* __restrict__ jobs, FillShared sh, uint32_t G) {
    const JobView& V = jobs[blockIdx.x / G];
    const uint32_t part = blockIdx.x % G;
    const DpParams P = V.P;
    const uint32_t n = V.n, nact = V.nact, Rtot = V.Rtot;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;

    __shared__ JumpBase base[MAXC];
    __shared__ JumpInfo s_jump[MAXC];
    __shared__ uint8_t s_circ[MAXC];                  // row 1 takes the circular end-to-start jump in this column
    __shared__ uint8_t rowm_xsuf[MAXC];               // cell(m, j-1).S is an x-suffix clip (:263-267)
    __shared__ int32_t rowm_S[MAXC]; __shared__ uint32_t rowm_len[MAXC];
    __shared__ int32_t s_vrun[MAXC];
    __shared__ uint32_t s_act[MAXC]; __shared__ int32_t s_opp[MAXC];
    __shared__ uint32_t s_abort;

    uint32_t* __restrict__ st = V.st16;
    unsigned long long* __restrict__ xchg = V.xchg;      // [2][C][2] granules
    if (threadIdx.x == 0) s_abort = 0;

    // ---- column 0 (init_matrices :97-186) ---------------------------------------------------------------------
    for (uint32_t k = part; k < nact; k += G) {
        const uint32_t c = V.act[k];
        const uint32_t roff = V.cd[c].roff, troff = V.cd[c].troff;
        const uint32_t mpad = (V.cd[c].m + TILE - 1) / TILE * TILE;
        for (uint32_t i = threadIdx.x; i < mpad; i += blockDim.x) {
            const uint32_t r = roff + i, tr = troff + i;
            st[2 * r] = pk16(sh.S0[tr], sh.Slen0[tr]);
            st[2 * r + 1] = pk16(-32768, 0);                          // D = "MIN": never extends, never wins
            V.Sn[r] = sh.Sn0[tr]; V.SnLen[r] = sh.Slen0[tr]; V.Ly[r] = sh.SnSet0[tr] ? n : 0u;
            V.SmoveF[r] = TB_NONE; V.ImoveF[r] = TB_NONE;
        }
    }
    for (uint32_t k = threadIdx.x; k < nact; k += blockDim.x) {
        const uint32_t c = V.act[k];
        const uint32_t trm = V.cd[c].troff + V.cd[c].m - 1;
        s_act[k] = c;
        base[c] = sh.base0[c];
        s_vrun[c] = sh.base0[c].score;
        rowm_xsuf[c] = sh.Smove0[trm] == TB_XCLIP_SUFFIX; rowm_S[c] = sh.S0[trm]; rowm_len[c] = sh.Slen0[trm];
        if (k % G == part) V.Lx[(size_t)c * (n + 1)] = sh.lx0[c];
    }
    for (uint32_t c = threadIdx.x; c < V.C; c += blockDim.x) s_opp[c] = V.opp_act[c];
    __syncthreads();

    for (uint32_t j = 1; j <= n; ++j) {
        // per-contig best jump out of column j-1 (multi_contig_aligner.rs:280-331): one thread per contig
        for (uint32_t k = part + threadIdx.x * G; k < nact; k += blockDim.x * G) {
            const uint32_t c = s_act[k];
            const JumpInfo ji = select_jump(P, base, s_act, nact, c, s_opp[c]);
            ColCtx cx; cx.jump = ji; cx.circ_ok = (P.circular && !rowm_xsuf[c]) ? 1 : 0; cx.circ_score = rowm_S[c]; cx.circ_len = rowm_len[c] + 1;
            const bool circ = local_row1_circ(cx);
            s_jump[c] = ji; s_circ[c] = circ ? 1 : 0;
            V.jt_idx[(size_t)c * (n + 1) + j] = ji.idx | (circ ? JT_CIRC_BIT : 0u);
            V.jt_from[(size_t)c * (n + 1) + j] = ji.from;
        }
        __syncthreads();

        uint8_t* __restrict__ tbcol = V.tb + (size_t)(j - 1) * Rtot;
        const uint8_t q = V.y[j - 1];

        for (uint32_t k = part + wave * G; k < nact; k += W * G) {
            const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_act[k]);
            const ContigDesc cd = V.cd[c];
            const uint32_t m = cd.m, roff = cd.roff;
            const uint8_t* __restrict__ xs = V.xseq + cd.seqoff;

            WaveCol wc;
            wc.js = s_jump[c].score; wc.jl = s_jump[c].len;
            if (s_circ[c]) { wc.js1 = rowm_S[c]; wc.jl1 = rowm_len[c] + 1; } else { wc.js1 = wc.js; wc.jl1 = wc.jl; }
            wc.vrun = s_vrun[c]; wc.m = m; wc.roff = roff; wc.j = j; wc.n = n; wc.q = q;
            wc.upS = 0; wc.upSl = 0; wc.upT = 0; wc.upTl = 0;      // row 0 of a Local-mode column: S 0, length 0
            wc.carry = scan_seed();
            LaneAcc acc; acc.xw = 0; acc.xrow = 0; acc.ck = 0;
            RowM rm; rm.key = 0; rm.Sl = 0; rm.bits = 0; rm.dpack = 0; rm.dg = 0;
            const uint32_t ntiles = (m + TILE - 1) / TILE;
            const int owner_lane = (int)(((m - 1) / R) & 63);

            const uint4* __restrict__ stv = reinterpret_cast<const uint4*>(st + 2 * (size_t)roff);
            uint4 pre[4]; uint2 prex;
            {
                const uint32_t o0 = lane * R;
#pragma unroll
                for (int v = 0; v < 4; ++v) pre[v] = stv[(o0 >> 1) + v];
                prex = *reinterpret_cast<const uint2*>(xs + o0);
            }
            for (uint32_t t = 0; t + 1 < ntiles; ++t) {
                uint4 cur[4]; const uint2 curx = prex;
#pragma unroll
                for (int v = 0; v < 4; ++v) cur[v] = pre[v];
                const uint32_t o1 = (t + 1) * TILE + lane * R;
#pragma unroll
                for (int v = 0; v < 4; ++v) pre[v] = stv[(o1 >> 1) + v];
                prex = *reinterpret_cast<const uint2*>(xs + o1);
                if (j == n) tile<false, true>(V, P, wc, acc, rm, cur, curx, t, lane, st, tbcol);
                else tile<false, false>(V, P, wc, acc, rm, cur, curx, t, lane, st, tbcol);
            }
            if (j == n) tile<true, true>(V, P, wc, acc, rm, pre, prex, ntiles - 1, lane, st, tbcol);
            else tile<true, false>(V, P, wc, acc, rm, pre, prex, ntiles - 1, lane, st, tbcol);

            // ---- wave reductions over rows < m -------------------------------------------------------------------
            // x-suffix running max: max S<<16|len, lowest row among equals; seed (MIN, 0) if there is no row below m
            const uint32_t xw = wave_max_u32(acc.xw);
            const uint32_t xrow = wave_min_u32(acc.xw == xw && acc.xrow != 0 ? acc.xrow : 0xFFFFFFFFu);
            XsRec xb_;
            if (xrow == 0xFFFFFFFFu) {
                // no lane recorded a row: either there is no row below m (seed stays MIN, len 0), or every S<<16|len is 0 and
                // the first row took the running value (0 > MIN, :408-417)
                if (m > 1) { xb_.v = 0; xb_.len = 0; xb_.row = 1; } else { xb_.v = MIN_SCORE; xb_.len = 0; xb_.row = 0; }
            }
            else { xb_.v = (int32_t)(xw >> 16); xb_.len = xw & 0xFFFFu; xb_.row = xrow; }
            // column arg-max over rows 0..m-1: row 0 holds S = 0 (key 0xFFFF); the winner's length is re-read from the state
            uint32_t ck = wave_max_u32(acc.ck); ck = ck > 0xFFFFu ? ck : 0xFFFFu;
            CmRec cb_; cb_.v = (int32_t)(ck >> 16); cb_.row = 0xFFFFu - (ck & 0xFFFFu); cb_.len = 0;
            // ---- row m (:350-351 seeded selection, :406-447 for i == m) -----------------------------------------
            {
                const uint32_t rmi = roff + m - 1;
                const int32_t ownS = rm.key >> 3; const uint32_t ownMv = (uint32_t)rm.key & 7u;
                int32_t Sm; uint32_t Slm, mvm, lx;
                lx = xb_.row == 0 ? 0u : m - xb_.row;
                bool do_x_m = false;
                if (rowm_run_wins(xb_.v, ownS, rm.dg)) { Sm = xb_.v; Slm = xb_.len; mvm = MK_XSUF; }
                else { Sm = ownS; Slm = rm.Sl; mvm = ownMv; if (rm.Sl > xb_.len) { do_x_m = true; lx = 0; } }
                if (lane == owner_lane) {
                    st[2 * rmi] = pk16(Sm, Slm); st[2 * rmi + 1] = rm.dpack;
                    tbcol[rmi] = (uint8_t)(mvm | rm.bits);
                    if (j == n) { V.S[rmi] = Sm; V.Slen[rmi] = Slm; }
                    const uint32_t rl = (j == n) ? (do_x_m ? rm.Sl : xb_.len) : 0u;
                    if (Sm >= wc.vrun) {
                        const int32_t sn = V.Sn[rmi];
                        if (Sm > sn || (Sm == sn && Slm > rl)) { V.Sn[rmi] = Sm; V.Ly[rmi] = n - j; V.SnLen[rmi] = Slm; }
                    }
                    V.Lx[(size_t)c * (n + 1) + j] = lx;
                }
                Sm = lane_bcast(Sm, owner_lane); Slm = (uint32_t)lane_bcast((int)Slm, owner_lane); mvm = (uint32_t)lane_bcast((int)mvm, owner_lane);
                if (lane == 0) {
                    if (Sm > cb_.v) { cb_.v = Sm; cb_.row = m; cb_.len = Slm; }
                    else if (cb_.row != 0) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the winner's state word was stored by this wave above
                        cb_.len = __builtin_nontemporal_load(st + 2 * (size_t)(roff + cb_.row - 1)) & 0xFFFFu;
                    }
                    JumpBase b; b.score = cb_.v; b.len = cb_.len + 1; b.from = cb_.row;
                    base[c] = b;
                    if (G > 1) {
                        unsigned long long* g = xchg + ((size_t)(j & 1) * V.C + c) * 2;
                        __hip_atomic_store(g, ((unsigned long long)j << 32) | (uint32_t)b.score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(g + 1, ((unsigned long long)j << 32) | (b.len << 16) | b.from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (cb_.v > wc.vrun) s_vrun[c] = cb_.v;
                    rowm_xsuf[c] = mvm == MK_XSUF; rowm_S[c] = Sm; rowm_len[c] = Slm;
                }
            }
        }
        __syncthreads();
        if (G > 1) {
            // gather the other workgroups' records of column j (bounded spin: a missing partner must not hang the GPU)
            for (uint32_t k = threadIdx.x; k < nact; k += blockDim.x) {
                if (k % G == part) continue;
                const uint32_t c = s_act[k];
                const unsigned long long* g = xchg + ((size_t)(j & 1) * V.C + c) * 2;
                const unsigned long long t0 = wall_clock64();
                unsigned long long a, b;
                for (;;) {
                    a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(a >> 32) == j && (uint32_t)(b >> 32) == j) break;
                    if (wall_clock64() - t0 > 400000000ull) { s_abort = 1; break; }      // 4 s at 100 MHz
                    __builtin_amdgcn_s_sleep(2);
                }
                JumpBase r; r.score = (int32_t)(uint32_t)a; r.len = ((uint32_t)b >> 16) & 0xFFFFu; r.from = (uint32_t)b & 0xFFFFu;
                base[c] = r;
            }
            __syncthreads();
            if (s_abort) { if (threadIdx.x == 0) *V.err = 1; return; }
        }
    }
}

void launch_fill_local16(const JobView* d_jobs, uint32_t n_jobs, uint32_t G, int waves, const FillShared& sh, hipStream_t stream) {
    hipLaunchKernelGGL(fill_local16_kernel, dim3(n_jobs * G), dim3(waves * 64), 0, stream, d_jobs, sh, G);
}

}  // namespace stitch
