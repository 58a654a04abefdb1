// Kernel 1b — the DP fill specialised for AlignmentMode::Local (the CLI default and the benchmark mode).
//
// Same recurrence, tie-breaks and results as fill_kernel.hip (and therefore as
// fg-stitch-lib/src/align/aligners/single_contig_aligner.rs:188-451 + multi_contig_aligner.rs:264-347), with the
// simplifications Local mode allows and an HBM-leaner, VALU-leaner inner loop:
//   * a candidate is one signed word score<<16 | len (dp_core.h, row_phase_a_word): the reference's chain of `>` tests
//     with length tie-breaks becomes compares and selects on words;
//   * row state is 8 bytes (S word, D word) instead of 16, read and written as 16-byte vectors;
//   * 4 rows per lane (256-row tiles), 12 waves per workgroup so that three waves share a SIMD; the waves share the
//     workgroup's tiles (slot table in LDS, carries handed from wave to wave through LDS), full tiles run a check-free
//     instance, only a contig's last tile handles ragged ends and row m, and the last column is an instance of its own;
//   * the next tile's state is loaded into the current tile's registers right after their last use (one buffer);
//   * the insertion scan is a lane-tagged max on DPP row shifts / broadcasts; the insertion is merged into a cell only
//     in tiles where it can change one;
//   * the y-suffix trackers Sn/Ly (:431-447) are only touched for cells that reach the contig's running maximum —
//     in Local mode only rows whose Sn equals the contig's final maximum can influence the result — and
//     written as packed records without reading Sn back;
//   * the per-contig jump selection (multi_contig_aligner.rs:292-331) is done by the wave that starts the contig, with
//     the other contigs' column arg-max spread over its lanes (one barrier per column, two with several workgroups).
// Eligibility is decided on the host (stitch_api.cpp: local16_ok): mode local, go + ge < 0, match * n <= 32767,
// n + max contig length < 65535, penalties >= -16000.  Anything else runs the generic int32 kernel.
#include <type_traits>
#include <hip/hip_runtime.h>
#include "dp_core.h"
#include "walk_core.h"
#include "fill_common.h"

namespace stitch {


namespace {

// Diagnostic build (-DSTITCH_CHECK): every offset the kernel derives is compared with the job's bounds before use; a
// violation is recorded in the error word (code << 8 | 2) and the access is skipped, so a bad offset shows up as a number
// instead of a GPU fault.
#ifdef STITCH_CHECK
#define CHK_FAIL(errp, code) atomicMax((uint32_t*)(uintptr_t)(errp), ((uint32_t)(code) << 8) | 2u)
#endif

// In-kernel stamps (diagnostic build only, -DSTITCH_PROFILE): per-wave cycle sums of the column loop's sections, written to
// the debug area behind V.err.  Never enabled in the product build.
#ifdef STITCH_PROFILE
#define PROF_DECL unsigned long long pf_t = __builtin_readcyclecounter(), pf_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PROF(k) { unsigned long long pf_n = __builtin_readcyclecounter(); pf_sum[k] += pf_n - pf_t; pf_t = pf_n; }
#else
#define PROF_DECL
#define PROF(k)
#endif


struct GPtrsCold {            // pointers of the rare paths live in LDS, not in (scarce) SGPRs
    gptr<int32_t> S, Ival; gptr<uint32_t> Slen, Ilen, Lx, jt_idx, jt_from;
    gptr<unsigned long long> xchg;
};
struct GPtrs {                // hot pointers, kept in registers
    gptr<uint32_t> st; gptr<const uint8_t> xseq; gptr<uint8_t> tb;
    gptr<u32x2> yrec;         // y-suffix tracker records {S word, n - j} per row (in the D/Dlen arrays, which this kernel does not use)
    const GPtrsCold* cold;    // in LDS
#ifdef STITCH_CHECK
    gptr<uint32_t> chk_err; uint32_t chk_Rtot;
#endif
};

// Software pipeline of the tile loop.  hipcc cannot count vmcnt across this loop (conditional memory operations in the body
// make it fall back to vmcnt(0), which also waits for the prefetch it has just issued), so the state prefetch is issued and
// waited for by hand (cdna_hip_programming.md §5.7): the loads of the NEXT slot are issued inside the current tile, right
// after phase A has consumed the registers they land in; when they are needed, at the top of the next slot, the only
// younger vector-memory operations that may still be in flight are the stores of the tile computed in between (two 16-byte
// state vectors and the traceback word at 4 rows per lane), hence vmcnt(3).  Extra (conditional) operations only make the
// wait stricter.  Rule that goes with it: every hand-issued load must be followed by a tile_wait on the same registers on
// every path, also the one after the last slot — the compiler tracks neither the loads nor the registers they land in.
// And the issue must be unconditional: under a branch (tried: skipping the reload after a wave's last slot) the outputs
// meet the old values in a phi, the compiler copies them right behind the asm, i.e. before the data has landed, and the
// next tile computes on garbage (the hand-off then never comes: caught by the bounded spins, and by the parity tests).
#ifndef STITCH_R
#define STITCH_R 4
#endif
constexpr int R = STITCH_R;                    // rows per lane: 8 (512-row tiles) or 4 (256-row tiles, half the registers)
static_assert(R == 8 || R == 4, "rows per lane");
struct TileRegs { u32x4 v0, v1, v2, v3; u32x2 x; };      // R == 4 uses v0, v1 and x.x only
__device__ __forceinline__ void tile_load_plain(TileRegs& r, gptr<const u32x4> p, gptr<const uint8_t> px) {
    r.v0 = p[0]; r.v1 = p[1];
    if constexpr (R == 8) { r.v2 = p[2]; r.v3 = p[3]; r.x = *(gptr<const u32x2>)px; } else { r.x.x = *(gptr<const uint32_t>)px; }
}
// Experiment builds (never shipped; results are garbage, only the kernel time is read — DESIGN.md "what bounds the fill"):
//   -DSTITCH_EXP_NOSTATE  the row state makes no round trip through memory: a tile's results stay in the registers and are
//                         the next tile's input (no state loads, no state stores)
//   -DSTITCH_EXP_NOTB     no traceback bytes and no y-suffix records are stored
#if defined(STITCH_EXP_NOSTATE) && defined(STITCH_EXP_NOTB)
#define EXP_VMCNT "0"
#elif defined(STITCH_EXP_NOSTATE)
#define EXP_VMCNT "1"
#elif defined(STITCH_EXP_NOTB)
#define EXP_VMCNT "2"
#else
#define EXP_VMCNT "3"
#endif
__device__ __forceinline__ void tile_load(TileRegs& r, gptr<const u32x4> p, gptr<const uint8_t> px) {
#ifdef STITCH_EXP_NOSTATE
    static_assert(R == 4, "experiment builds are R == 4");
    asm volatile("global_load_dword %0, %1, off" : "=&v"(r.x.x) : "v"(px) : "memory");
    return;
#endif
    if constexpr (R == 8) {
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                     "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                     : "=&v"(r.v0), "=&v"(r.v1), "=&v"(r.v2), "=&v"(r.v3) : "v"(p) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(r.x) : "v"(px) : "memory");
    } else {
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                     : "=&v"(r.v0), "=&v"(r.v1) : "v"(p) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=&v"(r.x.x) : "v"(px) : "memory");
    }
}
// wait until only the stores of the tile computed in between may still be in flight: R/2 state vectors + 1 traceback store
__device__ __forceinline__ void tile_wait(TileRegs& r) {
    if constexpr (R == 8) asm volatile("s_waitcnt vmcnt(5)" : "+v"(r.v0), "+v"(r.v1), "+v"(r.v2), "+v"(r.v3), "+v"(r.x) : : "memory");
    else asm volatile("s_waitcnt vmcnt(" EXP_VMCNT ")" : "+v"(r.v0), "+v"(r.v1), "+v"(r.x.x) : : "memory");
}
__device__ __forceinline__ void tile_wait_all(TileRegs& r) {
    if constexpr (R == 8) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.v0), "+v"(r.v1), "+v"(r.v2), "+v"(r.v3), "+v"(r.x) : : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.v0), "+v"(r.v1), "+v"(r.x.x) : : "memory");
}

constexpr uint32_t TILE = 64 * R;
constexpr uint32_t MAXSLOTS = 2048;

// One 512-row tile.  PARTIAL: the contig's last tile (rows may exceed m, and row m is held back).  LASTCOL (wave-uniform,
// run time): j == n, the int32 arrays the fix-up kernel reads are written as well.
template <bool PARTIAL, bool LASTCOL>
__device__ __forceinline__ void tile(const GPtrs& V, const WordConsts& K, const LaneK& LK, WaveCol& wc, LaneAcc& acc, RowM& rm, TileRegs& tr,
                                     uint32_t t, int lane, gptr<uint8_t> tbcol, gptr<const u32x4> ps_next, gptr<const uint8_t> px_next) {
    const u32x4 cur[4] = {tr.v0, tr.v1, R == 8 ? tr.v2 : tr.v0, R == 8 ? tr.v3 : tr.v1}; const u32x2 curx = tr.x;
    const uint32_t i0 = t * TILE + lane * R + 1;
    const uint32_t r = wc.roff + i0 - 1;
#ifdef STITCH_CHECK
    if (r + R > V.chk_Rtot || wc.m > 70000u || t > 300u) { CHK_FAIL(V.chk_err, 0x100000u | (t & 0xFFFu)); return; }
#endif
    const uint32_t m = wc.m;
    int32_t Sp[R], Dp[R]; uint32_t xb[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        Sp[u] = (int32_t)((u & 1) ? cur[u >> 1].z : cur[u >> 1].x); Dp[u] = (int32_t)((u & 1) ? cur[u >> 1].w : cur[u >> 1].y);
        xb[u] = ((u < 4 ? curx.x : curx.y) >> (8 * (u & 3))) & 0xFFu;
    }
    const int32_t nS = from_prev_lane(Sp[R - 1], wc.upS);
    wc.upS = lane_bcast(Sp[R - 1], 63);

    RowW ra[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const bool row1 = (u == 0) && (i0 == 1);
        row_phase_a_word(xb[u] == wc.q ? K.MW : K.XW, K.GE1, K.GO1, row1 ? wc.JSW1 : wc.JSW, u == 0 ? nS : Sp[u - 1], Sp[u], Dp[u], ra[u]);
        ROW_FENCE
    }
    // The previous column's words are dead from here on: the NEXT slot's state is loaded into the same registers and lands
    // during phases B and C (one register buffer; the wait is at the top of the next slot).
    tile_load(tr, ps_next, px_next);
    // phase B: insertion scan
    const int32_t nT = from_prev_lane(ra[R - 1].T, wc.upT);
    wc.upT = lane_bcast(ra[R - 1].T, 63);
    ScanEl el[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int32_t Tup = u == 0 ? nT : ra[u - 1].T;
        el[u].key = (word_score(Tup) - LK.giL) + (K.kb0 - K.ge * u);                // key_i = S'(i-1) + go + ge - ge*i   (i within the tile)
        el[u].q = ((int32_t)word_len(Tup) - (int32_t)LK.iL) + (1 - u);              // q_i   = S'.len(i-1) + 1 - i
        if (PARTIAL && i0 + u > m) el[u].key = SCAN_LOW;
    }
    ScanEl inc = el[0];
#pragma unroll
    for (int u = 1; u < R; ++u) inc = scan_combine(inc, el[u]);
    // wave level: max of (key << 6 | 63 - lane); lanes without a source keep INT32_MIN
    int32_t kt = (int32_t)(((uint32_t)inc.key << 6) | (uint32_t)LK.tag);
    { const int32_t o = dpp_mov<DPP_ROW_SHR0 | 1>(INT32_MIN, kt); kt = o > kt ? o : kt; } { const int32_t o = dpp_mov<DPP_ROW_SHR0 | 2>(INT32_MIN, kt); kt = o > kt ? o : kt; }
    { const int32_t o = dpp_mov<DPP_ROW_SHR0 | 4>(INT32_MIN, kt); kt = o > kt ? o : kt; } { const int32_t o = dpp_mov<DPP_ROW_SHR0 | 8>(INT32_MIN, kt); kt = o > kt ? o : kt; }
    { const int32_t o = dpp_mov<DPP_BCAST15, 0xA>(INT32_MIN, kt); kt = o > kt ? o : kt; } { const int32_t o = dpp_mov<DPP_BCAST31, 0xC>(INT32_MIN, kt); kt = o > kt ? o : kt; }
    ScanEl run;
    {
        const int32_t rt = from_prev_lane(kt, INT32_MIN);                           // exclusive: lanes before this one
        run.key = rt >> 6;                                                          // lane 0: -2^25, below the carry whatever it is
        run.q = __builtin_amdgcn_ds_bpermute((int)((63u - ((uint32_t)rt & 63u)) << 2), inc.q);
        run = scan_combine(wc.carry, run);
        const int32_t lt = lane_bcast(kt, 63);
        ScanEl last; last.key = lt >> 6; last.q = __builtin_amdgcn_readlane(inc.q, (int)(63u - ((uint32_t)lt & 63u)));
        wc.carry = scan_combine(wc.carry, last);
        wc.carry.key += K.ge * (int32_t)TILE; wc.carry.q += (int32_t)TILE;          // relative to the next tile
    }
    // phase C.  The insertion candidate changes a cell only where it beats {diagonal, deletion} (c2 of row_phase_c_word) AND
    // is not beaten by the jump (c5): with c2 and c5 both true the merged result is the jump again, exactly the (T, mvT) of
    // phase A (c5 and c2 give JW.score > BI.score > bs2h.score, hence c3, hence T = JW; c6 is c4).  That happens only where
    // the alignment really inserts: in almost every tile in no row of no lane.  A first pass therefore only carries the
    // chain's key through the lane's rows (the "I extended" traceback bit needs it anyway) and tests c2 && !c5, which are
    // comparisons of scores: word(bi, il) > bs2h <=> bi > score(bs2h) (bs2h's low half is all ones), and !c5 <=>
    // bi >= score(JW).  A negative insertion score changes nothing either: the merged cell is then clipped to (0, x-prefix), and
    // so is T (BI.score >= JW.score and > bs2.score make T negative too).  Only if some lane of the wave sees all three in
    // some row does the wave redo the rows with the full merge.
    int32_t Fo[R]; uint32_t code[R]; uint32_t tk = 0;
    bool merge = LASTCOL;                      // (the last column also stores the insertion values of every row)
    if (!LASTCOL) {
        int32_t rk = run.key; bool c2any = false;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const bool ext = rk >= el[u].key;
            code[u] = (ext ? TBB_IEXT : 0u) | (ra[u].dext ? TBB_DEXT : 0u);
            rk = ext ? rk : el[u].key;
            const int32_t bi = rk + LK.giL + K.ge * u;
            const int32_t biw = (int32_t)((uint32_t)bi << 16);                     // only looked at when 0 <= bi (<= 32767)
            c2any |= bi >= 0 && biw > ra[u].bs2h && (biw | 0xFFFF) >= ra[u].JW;
        }
        merge = __ballot(c2any) != 0ull;
#ifdef STITCH_PROFILE
        wc.n_tiles += 1; wc.n_merge += merge ? 1u : 0u; wc.n_c2 += (uint32_t)__popcll(__ballot(c2any));
#endif
    }
    // default: phase A's result; the merge, where it runs, overwrites it in place (no copies where the two paths join)
    uint32_t mvo[R];
#pragma unroll
    for (int u = 0; u < R; ++u) { Fo[u] = ra[u].T; mvo[u] = ra[u].mvT; }
    if (merge) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const uint32_t ext = run.key >= el[u].key ? 1u : 0u;
            if (!ext) run = el[u];
            const int32_t bi = run.key + LK.giL + K.ge * u;
            const uint32_t il = (uint32_t)(run.q + (int32_t)LK.iL + u);
            uint32_t mv;
            Fo[u] = row_phase_c_word(ra[u], bi, il, mv);
            mvo[u] = mv;
            if (LASTCOL) {                         // (before the last column the first pass has set these bits from the same chain)
                code[u] = (ext ? TBB_IEXT : 0u) | (ra[u].dext ? TBB_DEXT : 0u);
                if (!PARTIAL || i0 + u <= m) { const GPtrsCold& C = *V.cold; C.S[r + u] = word_score(Fo[u]); C.Slen[r + u] = word_len(Fo[u]); C.Ival[r + u] = bi; C.Ilen[r + u] = il; }
            }
            ROW_FENCE
        }
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const uint32_t i = i0 + u;
        code[u] |= mvo[u];
        if (!PARTIAL || i < m) tk = (uint32_t)Fo[u] > tk ? (uint32_t)Fo[u] : tk;       // the lane's largest S word of this tile (F >= 0)
        if (PARTIAL) {
            if (i == m) { rm.F = Fo[u]; rm.mv = mvo[u]; rm.bits = code[u] & (TBB_IEXT | TBB_DEXT); rm.BD = ra[u].BD; rm.DG = ra[u].DG; }
            if (i > m) { Fo[u] = 0; ra[u].BD = word_make(-16384, 0); code[u] = 0; }
        }
    }
    // Running records of the lane, updated per tile, not per row (rows ascend within a lane and across its tiles, so an equal
    // value never replaces an earlier one; after a contig's first tiles these branches are rarely taken):
    // x-suffix record = largest S word, first row holding it (:406-429)
    if (tk > acc.xw) {
        acc.xw = tk;
#pragma unroll
        for (int u = R - 1; u >= 0; --u) if ((!PARTIAL || i0 + u < m) && (uint32_t)Fo[u] == tk) acc.xrow = i0 + u;
    }
    // column arg-max = largest score, first row holding it, and that row's length (:677-697)
    if ((tk >> 16) > (acc.ck >> 16)) {
#pragma unroll
        for (int u = R - 1; u >= 0; --u)
            if ((!PARTIAL || i0 + u < m) && ((uint32_t)Fo[u] >> 16) == (tk >> 16)) { acc.ck = (tk & 0xFFFF0000u) | (0xFFFFu - (i0 + u)); acc.cklen = (uint32_t)Fo[u] & 0xFFFFu; }
    }
    // y-suffix trackers (:431-447): only cells that reach the contig's running maximum can matter.  Such a cell
    // has S >= vrun >= Sn (Sn is this row's maximum over earlier columns, vrun the contig's over the same columns; tests/emu
    // checks the invariant), so the reference's test `S > Sn || (S == Sn && len > 0)` is `len > 0`, i.e. the S word > 0, and
    // the record {S word, n - j} is stored without reading Sn back.  With a chimeric read every cell is reached by a jump
    // from the best column maximum, so this happens in a quarter of all rows: one masked 8-byte store per row.
#ifndef STITCH_EXP_NOTB
    if ((int32_t)(tk >> 16) >= wc.vrun) {
        const gptr<u32x2> yr = V.yrec + r;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if ((!PARTIAL || i0 + u < m) && Fo[u] >= wc.thr) { u32x2 rec; rec.x = (uint32_t)Fo[u]; rec.y = wc.n - wc.j; yr[u] = rec; }
        }
    }
#endif
#ifdef STITCH_EXP_NOSTATE
    tr.v0.x = (uint32_t)Fo[0]; tr.v0.y = (uint32_t)ra[0].BD; tr.v0.z = (uint32_t)Fo[1]; tr.v0.w = (uint32_t)ra[1].BD;
    tr.v1.x = (uint32_t)Fo[2]; tr.v1.y = (uint32_t)ra[2].BD; tr.v1.z = (uint32_t)Fo[3]; tr.v1.w = (uint32_t)ra[3].BD;
#else
    gptr<u32x4> stw = (gptr<u32x4>)(V.st + 2 * (size_t)r);
#pragma unroll
    for (int v = 0; v < R / 2; ++v) { u32x4 o; o.x = (uint32_t)Fo[2 * v]; o.y = (uint32_t)ra[2 * v].BD; o.z = (uint32_t)Fo[2 * v + 1]; o.w = (uint32_t)ra[2 * v + 1].BD; stw[v] = o; }
#endif
#ifdef STITCH_EXP_NOTB
    (void)tbcol; (void)code;
    return;
#endif
    if constexpr (R == 8) {
        u32x2 tbv;
        tbv.x = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
        tbv.y = code[4] | (code[5] << 8) | (code[6] << 16) | (code[7] << 24);
        *(gptr<u32x2>)(tbcol + r) = tbv;
    } else {
        // streamed once, read back only by the walk: non-temporal, so that it does not displace the row state in L2
        __builtin_nontemporal_store(code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24), (gptr<uint32_t>)(tbcol + r));
    }
}

}  // namespace

// G workgroups cooperate on one read: workgroup `part` owns the active contigs k = part, part+G, ... and exchanges its
// contigs' column arg-max (the next column's jump sources) with the others through 8-byte {data, tag} granules in
// global memory (agent-scope relaxed atomics: the tag travels with the data, so no fence is needed; double-buffered by
// column parity).  All G workgroups of a read must be resident at once: the host keeps the grid <= the CU count.
#ifndef STITCH_WGT0
#define STITCH_WGT0 115     // tiles given to a SIMD's first, second and third (or later) wave, relative (measured optimum, flat)
#define STITCH_WGT1 100
#define STITCH_WGT2 85
#endif
#ifndef STITCH_LB
#define STITCH_LB 768          // 12 waves per workgroup: <= 168 VGPRs, 3 waves per SIMD (MAX_WAVES in stitch_api.cpp)
#endif
__global__ __launch_bounds__(STITCH_LB) void fill_local16_kernel(const JobView* __restrict__ jobs, FillShared sh, uint32_t G, uint32_t slots_cap) {
    const JobView& V = jobs[blockIdx.x / G];
    const uint32_t part = blockIdx.x % G;
    const DpParams P = V.P;
    const uint32_t n = V.n, nact = V.nact, Rtot = V.Rtot;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;

    __shared__ JumpBase base2[2][MAXC];               // column arg-max per contig, double-buffered by column parity:
                                                      // column j reads [(j-1)&1] while its own results fill [j&1]
    __shared__ uint8_t rowm_xsuf[MAXC];               // cell(m, j-1).S is an x-suffix clip (:263-267)
    __shared__ int32_t rowm_S[MAXC]; __shared__ uint32_t rowm_len[MAXC];
    __shared__ int32_t s_vrun[MAXC];
    __shared__ uint32_t s_act[MAXC]; __shared__ int32_t s_opp[MAXC];
    __shared__ uint32_t s_abort;
    __shared__ int32_t s_carry[MAXC][12];             // carries handed from the wave that starts a contig to the one that ends it
    __shared__ uint32_t s_cflag[MAXC];                // (column << 8 | piece) for which s_carry[c] is valid
    extern __shared__ u32x4 s_slots[];                // [slots_cap] {contig | tile<<8 | flags, state row of the tile, base offset of the tile, -}:
                                                      // sized by the launch (<= MAXSLOTS), so that small jobs leave LDS for more workgroups per CU
    __shared__ uint32_t s_wbeg[16], s_wend[16];
    __shared__ uint32_t s_hwid[16], s_bnd[17];
    __shared__ uint32_t s_m[MAXC], s_roff[MAXC], s_seq[MAXC];

    __shared__ GPtrsCold s_cold;
    if (threadIdx.x == 0) {
        s_cold.S = as_global(V.S); s_cold.Ival = as_global(V.Ival);
        s_cold.Slen = as_global(V.Slen); s_cold.Ilen = as_global(V.Ilen); s_cold.Lx = as_global(V.Lx);
        s_cold.jt_idx = as_global(V.jt_idx); s_cold.jt_from = as_global(V.jt_from); s_cold.xchg = as_global(V.xchg);   // [2][C][2] granules
    }
    GPtrs GP;
    GP.st = as_global(V.st16); GP.xseq = as_global(V.xseq); GP.tb = as_global(V.tb); GP.cold = &s_cold;
#ifdef STITCH_CHECK
    GP.chk_err = as_global(V.err); GP.chk_Rtot = V.Rtot;
#endif
    GP.yrec = (gptr<u32x2>)as_global(V.D);           // [Rtot] 8-byte records: D and Dlen are contiguous (layout_job)
    const gptr<uint32_t> st = GP.st;
    const uint32_t C = V.C;
    const int32_t jump_same = P.jump_same, jump_opp = P.jump_opp, jump_inter = P.jump_inter, circular = P.circular;
    WordConsts K;
    K.MW = (int32_t)((uint32_t)P.match << 16); K.XW = (int32_t)((uint32_t)P.mismatch << 16);
    K.GE1 = (int32_t)((uint32_t)P.gap_extend << 16) + 1; K.GO1 = (int32_t)((uint32_t)(P.gap_open + P.gap_extend) << 16) + 1;
    K.ge = P.gap_extend; K.kb0 = P.gap_open + P.gap_extend;
    LaneK LK;
    LK.iL = (uint32_t)lane * R + 1u; LK.giL = K.ge * (int32_t)LK.iL; LK.tag = 63 - lane;
    if (threadIdx.x == 0) s_abort = 0;
    for (uint32_t k = threadIdx.x; k < MAXC; k += blockDim.x) s_cflag[k] = 0;

    // ---- column 0 (init_matrices :97-186) ---------------------------------------------------------------------
    for (uint32_t k = part; k < nact; k += G) {
        const uint32_t c = V.act[k];
        const uint32_t roff = V.cd[c].roff, troff = V.cd[c].troff;
        const uint32_t mpad = (V.cd[c].m + TILE - 1) / TILE * TILE;
        for (uint32_t i = threadIdx.x; i < mpad; i += blockDim.x) {
            const uint32_t r = roff + i, tr = troff + i;
            st[2 * r] = (uint32_t)word_make(sh.S0[tr], sh.Slen0[tr]);
            st[2 * r + 1] = (uint32_t)word_make(-16384, 0);            // D = "MIN": never extends, never wins, cannot wrap
            { u32x2 rec; rec.x = (uint32_t)word_make(sh.Sn0[tr], sh.Slen0[tr]); rec.y = sh.SnSet0[tr] ? n : 0u; GP.yrec[r] = rec; }
            V.SmoveF[r] = TB_NONE; V.ImoveF[r] = TB_NONE;
        }
    }
    for (uint32_t k = threadIdx.x; k < nact; k += blockDim.x) {
        const uint32_t c = V.act[k];
        const uint32_t trm = V.cd[c].troff + V.cd[c].m - 1;
        s_act[k] = c;
        base2[0][c] = sh.base0[c];
        s_vrun[c] = sh.base0[c].score;
        rowm_xsuf[c] = sh.Smove0[trm] == TB_XCLIP_SUFFIX; rowm_S[c] = sh.S0[trm]; rowm_len[c] = sh.Slen0[trm];
        if (k % G == part) V.Lx[(size_t)c * (n + 1)] = sh.lx0[c];
    }
    for (uint32_t c = threadIdx.x; c < V.C; c += blockDim.x) s_opp[c] = V.opp_act[c];
    __syncthreads();

    // ---- per-wave slot table -------------------------------------------------------------------------------------
    // The workgroup's tiles (its contigs in order, each top to bottom) are cut into W equal ranges, one per wave, so the
    // waves are balanced whatever the contig lengths are.  A wave walks, in this order: the head piece of a contig that
    // continues in the next wave (so that wave is not kept waiting), its whole contigs, and last the piece that continues the
    // previous wave's contig.
    // The SIMD serves its waves strictly in wave-slot order (s_setprio does not change it: measured), so of the three waves
    // a SIMD holds, the one in the lowest slot needs 0.6x and the one in the highest 1.0x of the time for the same tiles.
    // The tile ranges are therefore cut in proportion to a weight by a wave's rank among the waves of its SIMD.
    if (lane == 0) s_hwid[wave] = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID: wave slot [3:0], SIMD [5:4]
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t wsum = 0;
        for (int w = 0; w < W; ++w) {
            uint32_t rank = 0;
            for (int v = 0; v < W; ++v) if (((s_hwid[v] >> 4) & 3u) == ((s_hwid[w] >> 4) & 3u) && (s_hwid[v] & 15u) < (s_hwid[w] & 15u)) ++rank;
            s_bnd[w] = wsum;
            wsum += rank == 0 ? STITCH_WGT0 : rank == 1 ? STITCH_WGT1 : STITCH_WGT2;
        }
        s_bnd[W] = wsum;
        uint32_t NT = 0;
        for (uint32_t k = part; k < nact; k += G) NT += (V.cd[V.act[k]].m + TILE - 1) / TILE;
        if (NT > slots_cap) NT = slots_cap;                   // (the host sizes the table for the launch's largest workgroup)
        // weights -> tile boundaries.  No wave of the first min(W, NT) may get an empty range: a contig's pieces are numbered
        // by consecutive waves, and a wave waits for the piece number before its own (an empty range in between would never
        // publish it).
        for (int w = W; w >= 0; --w) s_bnd[w] = (uint32_t)((unsigned long long)NT * s_bnd[w] / wsum);
        if (NT >= (uint32_t)W) { for (int w = 1; w < W; ++w) { const uint32_t lo_ = s_bnd[w - 1] + 1, hi_ = NT - (uint32_t)(W - w); s_bnd[w] = s_bnd[w] < lo_ ? lo_ : s_bnd[w] > hi_ ? hi_ : s_bnd[w]; } }
        else for (int w = 0; w <= W; ++w) s_bnd[w] = (uint32_t)w < NT ? (uint32_t)w : NT;
        // tile g of the workgroup -> (contig, tile): walk once, emitting records in global order into s_slots[...] scratch order
        uint32_t ns = 0;
        for (int w = 0; w < W; ++w) {
            const uint32_t lo = s_bnd[w], hi = s_bnd[w + 1];
            s_wbeg[w] = ns;
            // locate (contig, tile) of global tile index `lo`
            uint32_t g = 0, kk = part, t0 = 0;
            for (; kk < nact; kk += G) { const uint32_t nt = (V.cd[V.act[kk]].m + TILE - 1) / TILE; if (g + nt > lo) { t0 = lo - g; break; } g += nt; }
            // first pass: find f = first index in [lo, hi) with tile 0 (or hi if none)
            uint32_t f = hi;
            { uint32_t k2 = kk, t = t0; for (uint32_t x = lo; x < hi; ++x) { if (t == 0) { f = x; break; } const uint32_t nt = (V.cd[V.act[k2]].m + TILE - 1) / TILE; if (++t == nt) { t = 0; k2 += G; } } }
            auto emit = [&](uint32_t from, uint32_t to) {      // records of global tiles [from, to), in order
                uint32_t k2 = kk, t = t0;
                for (uint32_t x = lo; x < to; ++x) {
                    const uint32_t c = V.act[k2]; const uint32_t nt = (V.cd[c].m + TILE - 1) / TILE;
                    if (x >= from && ns < slots_cap) {
                        u32x4 rec;
                        rec.x = c | (t << 8) | (t == 0 ? SLOT_FIRST : 0u) | (t + 1 == nt ? SLOT_LAST : 0u) |
                                ((x == lo && t != 0) ? SLOT_CIN : 0u) | ((x + 1 == hi && t + 1 != nt) ? SLOT_COUT : 0u);
                        rec.y = V.cd[c].roff + t * TILE; rec.z = V.cd[c].seqoff + t * TILE;
                        // piece number of this wave's part of the contig: waves between the one holding tile 0 and this one
                        const uint32_t gstart = x - t;                                   // global index of the contig's tile 0
                        uint32_t w0 = (uint32_t)w;                                       // the wave whose range holds gstart
                        while (w0 > 0 && s_bnd[w0] > gstart) --w0;
                        rec.w = (uint32_t)w - w0;
                        s_slots[ns++] = rec;
                    }
                    if (++t == nt) { t = 0; k2 += G; }
                }
            };
            // g = start of the last contig begun in this range if it does not also end here (the piece the NEXT wave waits for)
            uint32_t gpos = hi;
            { uint32_t k2 = kk, t = t0, start = lo; bool open_ = (t0 != 0);
              for (uint32_t x = lo; x < hi; ++x) { if (t == 0) { start = x; open_ = true; } const uint32_t nt = (V.cd[V.act[k2]].m + TILE - 1) / TILE; if (++t == nt) { t = 0; k2 += G; open_ = false; } }
              if (open_ && start >= f && f < hi) gpos = start; }
            emit(gpos, hi);     // 1. the head piece the next wave depends on
            emit(f, gpos);      // 2. whole contigs
            emit(lo, f);        // 3. the continuation of the previous wave's contig (its head was that wave's step 1)
            s_wend[w] = ns;
        }
    }
    for (uint32_t k = threadIdx.x; k < nact; k += blockDim.x) { const uint32_t c = V.act[k]; s_m[c] = V.cd[c].m; s_roff[c] = V.cd[c].roff; s_seq[c] = V.cd[c].seqoff; }
    __syncthreads();
    const uint32_t sbeg = s_wbeg[wave], send = s_wend[wave];

    // slot records are read from LDS one slot ahead of their use, so the LDS latency hides behind a tile's arithmetic
    auto slot_rec = [&](uint32_t s) -> u32x4 { return s_slots[s < send ? s : send - 1]; };      // clamped: the pipeline always loads
    auto rec_ptrs = [&](const u32x4& rec, gptr<const u32x4>& ps, gptr<const uint8_t>& px) {
        const uint32_t row = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.y), seq = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.z);
#ifdef STITCH_CHECK
        if (row + 64u * R > Rtot || seq > (1u << 26)) { CHK_FAIL(GP.chk_err, 0x200000u | (row >> 12)); ps = (gptr<const u32x4>)st; px = GP.xseq; return; }
#endif
        ps = (gptr<const u32x4>)(st + 2 * (size_t)(row + lane * R));
        px = GP.xseq + seq + lane * R;
    };

    PROF_DECL
    uint32_t ychunk = 0;
    auto column = [&](auto lastcol_tag, uint32_t j) __attribute__((always_inline)) {
        constexpr bool LASTCOL = decltype(lastcol_tag)::value;
        PROF(0)
        const JumpBase* base = base2[(j - 1) & 1];
        JumpBase* base_out = base2[j & 1];
        const gptr<uint8_t> tbcol = GP.tb + (size_t)(j - 1) * Rtot;
#ifdef STITCH_CHECK
        if (GP.st != as_global(V.st16) || GP.tb != as_global(V.tb) || GP.xseq != as_global(V.xseq) || (gptr<uint32_t>)GP.yrec != as_global((uint32_t*)V.D) || j > n)
            CHK_FAIL(GP.chk_err, 0x300000u);
#endif
        // the read's bases, 64 columns per (coalesced) load: lane l holds y[jb + l]
        if (((j - 1) & 63u) == 0) ychunk = (j - 1 + lane < n) ? (uint32_t)V.y[j - 1 + lane] : 0u;
        const uint8_t q = (uint8_t)__builtin_amdgcn_readlane((int)ychunk, (int)((j - 1) & 63u));

        WaveCol wc; LaneAcc acc; RowM rm;
        wc.j = j; wc.n = n; wc.q = (uint32_t)q;
        uint32_t c = 0;
        TileRegs T;
        u32x4 rn = slot_rec(sbeg);
        if (sbeg < send) { gptr<const u32x4> ps; gptr<const uint8_t> px; rec_ptrs(rn, ps, px); tile_load(T, ps, px); }
        // one slot: contig set-up on its first tile, the tile itself, and the contig's row-m / reduction epilogue on its last
        auto process = [&](uint32_t e, uint32_t piece, TileRegs& T, gptr<const u32x4> psn, gptr<const uint8_t> pxn) __attribute__((always_inline)) {
            const uint32_t t = (e >> 8) & SLOT_TILE_MASK;
            if (e & SLOT_FIRST) {
                c = e & 0xFFu;
#ifdef STITCH_CHECK
                if (c >= C) { CHK_FAIL(GP.chk_err, 0x400000u | c); c = 0; }
#endif
                // best jump out of column j-1 for contig c (multi_contig_aligner.rs:292-331), computed by this wave: lanes hold
                // the other contigs' column arg-max; inter-contig = max by (score, len), LAST aligner on full ties (max_by_key)
                const int32_t opp = s_opp[c];
                unsigned long long ik = 0;
                for (uint32_t k = lane; k < nact; k += 64) {
                    const uint32_t a = s_act[k];
                    if (a != c && (int32_t)a != opp) {
                        const JumpBase bb = base[a];
                        const unsigned long long key = ((unsigned long long)(uint32_t)bb.score << 32) | ((unsigned long long)bb.len << 16) | (k + 1);
                        ik = key > ik ? key : ik;
                    }
                }
                ik = wave_max_u64(ik);
                JumpInfo ji; { const JumpBase bs = base[c]; ji.score = bs.score + jump_same; ji.len = bs.len; ji.idx = c; ji.from = bs.from; }
                if (opp >= 0) { const JumpBase bo = base[opp]; const int32_t sc = bo.score + jump_opp; if (sc > ji.score) { ji.score = sc; ji.len = bo.len; ji.idx = (uint32_t)opp; ji.from = bo.from; } }
                if (ik != 0) {
                    const uint32_t a = s_act[(uint32_t)(ik & 0xFFFFu) - 1]; const JumpBase bi_ = base[a];
                    const int32_t sc = bi_.score + jump_inter;
                    if (sc > ji.score) { ji.score = sc; ji.len = bi_.len; ji.idx = a; ji.from = bi_.from; }
                }
                ColCtx cx; cx.jump = ji; cx.circ_ok = (circular && !rowm_xsuf[c]) ? 1 : 0; cx.circ_score = rowm_S[c]; cx.circ_len = rowm_len[c] + 1;
                const bool circ = local_row1_circ(cx);
                if (lane == 0) {
                    s_cold.jt_idx[(size_t)c * (n + 1) + j] = ji.idx | (circ ? JT_CIRC_BIT : 0u);
                    s_cold.jt_from[(size_t)c * (n + 1) + j] = ji.from;
                }
                wc.JSW = word_make(ji.score, ji.len);
                wc.JSW1 = circ ? word_make(rowm_S[c], rowm_len[c] + 1) : wc.JSW;
                wc.JSW = __builtin_amdgcn_readfirstlane(wc.JSW); wc.JSW1 = __builtin_amdgcn_readfirstlane(wc.JSW1);
                wc.vrun = __builtin_amdgcn_readfirstlane(s_vrun[c]); wc.thr = wc.vrun > 0 ? (int32_t)((uint32_t)wc.vrun << 16) : 1;
                wc.m = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_m[c]); wc.roff = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_roff[c]);
                wc.upS = 0; wc.upT = 0;                                 // row 0 of a Local-mode column: score 0, length 0
                wc.carry.key = SCAN_LOW; wc.carry.q = 0;
                acc.xw = 0; acc.xrow = 0; acc.ck = 0; acc.cklen = 0;
                rm.F = 0; rm.mv = 0; rm.bits = 0; rm.BD = 0; rm.DG = 0;
            } else if (e & SLOT_CIN) {
                // continue a contig another wave started in this column: wait for its carries (same workgroup, always resident)
                c = e & 0xFFu;
                const uint32_t want = (j << 8) | (piece & 0xFFu);      // (column, piece)
                {   // bounded: a hand-off that never comes must end the kernel with an error, not hang the GPU
                    const unsigned long long t0 = wall_clock64();
                    while (__hip_atomic_load(&s_cflag[c], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != want) {
                        if (wall_clock64() - t0 > 400000000ull || __hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { s_abort = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                wc.upS = __builtin_amdgcn_readfirstlane(s_carry[c][0]); wc.upT = __builtin_amdgcn_readfirstlane(s_carry[c][1]);
                wc.carry.key = __builtin_amdgcn_readfirstlane(s_carry[c][2]); wc.carry.q = __builtin_amdgcn_readfirstlane(s_carry[c][3]);
                wc.JSW = __builtin_amdgcn_readfirstlane(s_carry[c][4]); wc.JSW1 = __builtin_amdgcn_readfirstlane(s_carry[c][5]);
                wc.vrun = __builtin_amdgcn_readfirstlane(s_carry[c][6]); wc.thr = wc.vrun > 0 ? (int32_t)((uint32_t)wc.vrun << 16) : 1;
                wc.m = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_m[c]); wc.roff = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_roff[c]);
                // the first wave's records are already reduced; lane 0 carries them on (its rows are the lowest of this piece, and
                // rows of the earlier piece are lower still, so "lowest row wins" is preserved by the max / min reductions)
                acc.xw = lane == 0 ? (uint32_t)s_carry[c][7] : 0u; acc.xrow = lane == 0 ? (uint32_t)s_carry[c][8] : 0u;
                acc.ck = lane == 0 ? (uint32_t)s_carry[c][9] : 0u; acc.cklen = lane == 0 ? (uint32_t)s_carry[c][10] : 0u;
                rm.F = 0; rm.mv = 0; rm.bits = 0; rm.BD = 0; rm.DG = 0;
            }
            PROF(3)
            if (!(e & SLOT_LAST)) {
                tile<false, LASTCOL>(GP, K, LK, wc, acc, rm, T, t, lane, tbcol, psn, pxn);
                if (e & SLOT_COUT) {
                    // hand the contig over to the wave that owns its next tile
                    const uint32_t xw = wave_max_u32(acc.xw);
                    const uint32_t xrow = wave_min_u32(acc.xw == xw && acc.xrow != 0 ? acc.xrow : 0xFFFFFFFFu);
                    const uint32_t ck = wave_max_u32(acc.ck);
                    const uint32_t cklen = ck_len_of(acc, ck);
                    if (lane == 0) {
                        s_carry[c][0] = wc.upS; s_carry[c][1] = wc.upT; s_carry[c][2] = wc.carry.key; s_carry[c][3] = wc.carry.q;
                        s_carry[c][4] = wc.JSW; s_carry[c][5] = wc.JSW1; s_carry[c][6] = wc.vrun;
                        s_carry[c][7] = (int32_t)xw; s_carry[c][8] = (int32_t)(xrow == 0xFFFFFFFFu ? 0u : xrow); s_carry[c][9] = (int32_t)ck; s_carry[c][10] = (int32_t)cklen;
                    }
                    if (lane == 0) __hip_atomic_store(&s_cflag[c], (j << 8) | ((piece + 1u) & 0xFFu), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                PROF(4)
                return;
            }
            tile<true, LASTCOL>(GP, K, LK, wc, acc, rm, T, t, lane, tbcol, psn, pxn);
            PROF(4)
            const uint32_t m = wc.m, roff = wc.roff;
            const int owner_lane = (int)(((m - 1) / R) & 63);

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
            uint32_t ck = wave_max_u32(acc.ck);
            const uint32_t cklen = ck_len_of(acc, ck);
            ck = ck > 0xFFFFu ? ck : 0xFFFFu;
            CmRec cb_; cb_.v = (int32_t)(ck >> 16); cb_.row = 0xFFFFu - (ck & 0xFFFFu); cb_.len = cb_.row != 0 ? cklen : 0u;
            // ---- row m (:350-351 seeded selection, :406-447 for i == m) -----------------------------------------
            {
                uint32_t rmi = roff + m - 1;
#ifdef STITCH_CHECK
                if (rmi >= Rtot || m == 0) { CHK_FAIL(GP.chk_err, 0x500000u); rmi = 0; }
#endif
                const int32_t ownS = word_score(rm.F); const uint32_t ownMv = rm.mv, ownSl = word_len(rm.F);
                int32_t Sm; uint32_t Slm, mvm, lx;
                lx = xb_.row == 0 ? 0u : m - xb_.row;
                bool do_x_m = false;
                if (rowm_run_wins(xb_.v, ownS, word_score(rm.DG))) { Sm = xb_.v; Slm = xb_.len; mvm = MK_XSUF; }
                else { Sm = ownS; Slm = ownSl; mvm = ownMv; if (ownSl > xb_.len) { do_x_m = true; lx = 0; } }
                if (lane == owner_lane) {
                    st[2 * rmi] = (uint32_t)word_make(Sm, Slm); st[2 * rmi + 1] = (uint32_t)rm.BD;
                    tbcol[rmi] = (uint8_t)(mvm | rm.bits);
                    if (LASTCOL) { s_cold.S[rmi] = Sm; s_cold.Slen[rmi] = Slm; }
                    const uint32_t rl = LASTCOL ? (do_x_m ? ownSl : xb_.len) : 0u;
                    if (Sm >= wc.vrun) {
                        // before the last column (rl = 0) Sm >= vrun >= Sn, and a zero-length S is a clipped 0 that cannot exceed Sn:
                        // the reference's test reduces to Slm > 0 and Sn need not be read back (tests/emu checks both)
                        bool upd = Slm > 0u;
                        if (LASTCOL) { const int32_t sn = word_score((int32_t)GP.yrec[rmi].x); upd = Sm > sn || (Sm == sn && Slm > rl); }
                        if (upd) { u32x2 rec; rec.x = (uint32_t)word_make(Sm, Slm); rec.y = n - j; GP.yrec[rmi] = rec; }
                    }
                    s_cold.Lx[(size_t)c * (n + 1) + j] = lx;
                }
                Sm = lane_bcast(Sm, owner_lane); Slm = (uint32_t)lane_bcast((int)Slm, owner_lane); mvm = (uint32_t)lane_bcast((int)mvm, owner_lane);
                if (lane == 0) {
                    if (Sm > cb_.v) { cb_.v = Sm; cb_.row = m; cb_.len = Slm; }
                    JumpBase b; b.score = cb_.v; b.len = cb_.len + 1; b.from = cb_.row;
                    base_out[c] = b;
                    if (G > 1) {
                        gptr<unsigned long long> g = s_cold.xchg + ((size_t)(j & 1) * C + c) * 2;
                        __hip_atomic_store(g, ((unsigned long long)j << 32) | (uint32_t)b.score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(g + 1, ((unsigned long long)j << 32) | (b.len << 16) | b.from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (cb_.v > wc.vrun) s_vrun[c] = cb_.v;
                    rowm_xsuf[c] = mvm == MK_XSUF; rowm_S[c] = Sm; rowm_len[c] = Slm;
                }
            }
            PROF(5)
        };
        // one register buffer: slot s+1's state is loaded inside slot s's tile, after its last use of the registers
        for (uint32_t s = sbeg; s < send; ++s) {
            const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)rn.x), piece = (uint32_t)__builtin_amdgcn_readfirstlane((int)rn.w);
            rn = slot_rec(s + 1);                                   // LDS read, used after phase A (clamped: the last slot reloads itself)
            PROF(3)
            if (s == sbeg) tile_wait_all(T); else tile_wait(T);
            PROF(6)
            gptr<const u32x4> psn; gptr<const uint8_t> pxn; rec_ptrs(rn, psn, pxn);
            process(e, piece, T, psn, pxn);
        }
        // The last slot has reloaded itself (slot_rec clamps).  Those loads are hand-issued, so the compiler does not know they
        // are in flight: without this wait it reuses T's registers after the loop (for the exchange pointers below, or the next
        // column's slot pointers), the late data lands in them, and the kernel reads through a garbage address.
        tile_wait(T);
#ifdef STITCH_PROFILE
        pf_sum[2] += (unsigned long long)wc.n_merge + ((unsigned long long)wc.n_tiles << 32);
#endif
        PROF(3)
        __syncthreads();
        PROF(7)
        if (G == 1 && s_abort) { if (threadIdx.x == 0) *V.err = 1; return true; }
        if (G > 1) {
            // gather the other workgroups' records of column j (bounded spin: a missing partner must not hang the GPU)
            for (uint32_t k = threadIdx.x; k < nact; k += blockDim.x) {
                if (k % G == part) continue;
                const uint32_t c = s_act[k];
                gptr<unsigned long long> g = s_cold.xchg + ((size_t)(j & 1) * C + c) * 2;
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
                base_out[c] = r;
            }
            __syncthreads();
            if (s_abort) { if (threadIdx.x == 0) *V.err = 1; return true; }
        }
        return false;
    };
    // the last column also writes the int32 arrays of the fix-up kernel: a separate instance keeps those stores (and their
    // per-row branches) out of the other n-1 columns
    for (uint32_t j = 1; j < n; ++j) if (column(std::false_type{}, j)) return;
    if (column(std::true_type{}, n)) return;
    // unpack the y-suffix records of this workgroup's contigs into the arrays the fix-up kernel reads (the last column's
    // barrier has made every wave's records visible to the workgroup)
    for (uint32_t k = part; k < nact; k += G) {
        const uint32_t c = V.act[k];
        const uint32_t roff = V.cd[c].roff, m = V.cd[c].m;
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
            const u32x2 rec = GP.yrec[roff + i];
            V.Sn[roff + i] = word_score((int32_t)rec.x); V.SnLen[roff + i] = word_len((int32_t)rec.x); V.Ly[roff + i] = rec.y;
        }
    }
#ifdef STITCH_PROFILE
    if (lane == 0) { atomicAdd((unsigned long long*)(V.err + 4) + 120, pf_sum[2] >> 32); atomicAdd((unsigned long long*)(V.err + 4) + 121, pf_sum[2] & 0xFFFFFFFFull); }
    pf_sum[1] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8]
    if (lane == 0 && blockIdx.x == 0) { unsigned long long* o = (unsigned long long*)(V.err + 4) + wave * 8; for (int k = 0; k < 8; ++k) o[k] = pf_sum[k]; }
    if (threadIdx.x == 0) {
        unsigned long long* o = (unsigned long long*)(V.err + 4);
        o[122] = wall_clock64();                                                           // when this read's (last-written) workgroup ended, 100 MHz
        if (part < 32) { o[128 + 2 * part] = wall_clock64();                              // per workgroup: end time, placement (XCC_ID, HW_ID)
                         o[129 + 2 * part] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4); }
    }
#endif
}

uint32_t fill_local16_max_slots() { return MAXSLOTS; }
void launch_fill_local16(const JobView* d_jobs, uint32_t n_jobs, uint32_t G, int waves, uint32_t slots_cap, const FillShared& sh, hipStream_t stream) {
    slots_cap = slots_cap < 16u ? 16u : slots_cap > MAXSLOTS ? MAXSLOTS : slots_cap;
    hipLaunchKernelGGL(fill_local16_kernel, dim3(n_jobs * G), dim3(waves * 64), sizeof(u32x4) * (size_t)slots_cap, stream, d_jobs, sh, G, slots_cap);
}

}  // namespace stitch
