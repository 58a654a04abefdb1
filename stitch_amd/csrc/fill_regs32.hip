// Kernel 1d — the DP fill with the row state RESIDENT IN REGISTERS for EVERY clipping mode and for long reads: 32-bit scores.
//
// fill_regs.hip packs a cell's score and alignment length into one 32-bit word, which Local mode with match * n <= 32767 allows.
// Query-local, target-local and global alignments carry scores far outside 16 bits (a 10 kb global alignment starts around
// go + ge * j), and so do Local reads beyond 32 767 / match bases.  This kernel keeps the same organisation — one contig per
// wavefront, a lane holds a run of consecutive rows in registers for the whole read, the team of a read's waves exchanges the
// per-contig column arg-max through tagged granules, no workgroup barrier in the column loop, traceback bytes lane-interleaved
// (walk_core.h tb_row_offset) — with the generic recurrence (dp_core.h row_phase_a / row_phase_c, single_contig_aligner.rs:
// 292-451) on THREE registers per row: S (int32), D (int32) and the two 16-bit alignment lengths packed in one register.
// 80 rows x 3 = 240 registers do not fit the 256 a wave has at two waves per SIMD: the kernel runs ONE wave per SIMD (a 256-thread
// workgroup per CU, waves_per_eu = 1) with D (read and written once per column) and the y-suffix maxima in the wave's accumulation registers
// (gfx950: 256 VGPRs + 256 AGPRs per wave at this occupancy; v_accvgpr_read / _write move a value across: one instruction).
//
// What the clipping modes change (aligners/constants.rs:96-136, aligners/mod.rs:123-131): each of the four clip penalties is 0
// ("free") or MIN_SCORE.  Under the host's eligibility bound (stitch_api.cpp regs32_plan: every score the recurrence can
// produce stays within +-2^27 of zero) a candidate that carries a MIN_SCORE penalty never wins a comparison and a cell's score
// never reaches MIN_SCORE, so
//   x clips free (local, target-local): the x-prefix candidate xclip_score (:304-308, :384-389) and the x-suffix running
//       maximum into row m (:406-429, :350-351) exist; otherwise neither can win and both are left out;
//   y clips free (local, query-local): the y-prefix candidate go + ge * i (:391-399) and the y-suffix trackers Sn / Ly
//       (:431-447) exist; otherwise Sn stays below every cell and fill_last_column_and_end_clipping never takes it (:469-491).
// Both prefix candidates are <= a bound that the column's jump candidate beats in nearly every column (every cell can take the
// jump): they are evaluated behind one scalar test per column.
//
// The insertion chain I[i] = max(I[i-1] + ge, S'[i-1] + go + ge) uses fill_regs.hip's formulation, which does not depend on the
// mode: pass 1b runs the chain of the lane's OWN openers down the lane and merges it where it reaches a cell's score; the chains
// cross the lanes as a prefix maximum of position-normalised scores (earliest lane wins ties: the extension wins ties, :321);
// pass 2 repairs, only while the arriving chain X is alive (no opener has beaten it strictly), what pass 1b assumed.  A merge is
// the full phase C of dp_core.h (the insertion at its place in the priority order, then jump and prefix clips again).
//
// y-suffix trackers: kept EXACTLY (running maximum per row in an accumulation register, first maximum wins, ties by length as
// :431-447), but only for cells that can still matter: in local mode a cell below the best score seen so far (in any contig for
// `traceback`, in its own contig otherwise) cannot be the end of the alignment; in query-local mode (x global: the alignment
// ends in row m) a row's y-suffix clip reaches the result in two ways only: through the insertion chain of the last column into
// row m (:521-554), which needs Sn[i] + go + ge (m - i) > S(m, n); or as the SOURCE CELL of the end-of-read jump (:458-466: the
// jump is scored with S(i, n) before row i's own y-suffix clip is applied, the walk then finds the clipped cell), which
// survives row m's own y-suffix clip only if S(i, n) + jump_same >= Sn[m], i.e. Sn[i] + jump_same > Sn[m].  The final S(m, n) and
// Sn[m] are never below the running maximum RM of row m over the columns so far, so a cell matters only if
// S(i, j) + max(jump_same, go + ge (m - i)) > RM — when the read is aligned to ONE contig: with several, the walk behind an
// end-of-read jump continues in the contig the source cell came from (traceback/mod.rs:329-338), on a cell no bound of this kind
// covers, and every tracker is kept.  A cell that fails its test cannot matter later either (the bounds only rise), and a later
// cell of the row that passes is larger than every cell of that row that failed before it.
#include <hip/hip_runtime.h>
#include "dp_core.h"
#include "walk_core.h"
#include "fill_common.h"

namespace stitch {
namespace {

constexpr int RMAX = REGS_RMAX;                // rows per lane: the traceback layout is fill_regs.hip's
constexpr int NG = RMAX / 4;
constexpr uint32_t RSRC_WORD3 = 0x00020000u;
constexpr int AUX_NT = 2, AUX_SC1 = 16, AUX_VOLATILE = (int)0x80000000u;
constexpr uint32_t LDS_XW = 0, LDS_BS = NG * 64 * 4, LDS32_PER_WAVE = NG * 64 * 4 + RMAX * 64 * 4;      // bases; best{diagonal, deletion} of every row (pass 1 -> merges)
constexpr int32_t KEY_BIAS = 30000;            // records are kept on scores relative to (column j-1's best score of any contig) - KEY_BIAS, in 16 bits
constexpr int32_t CHAIN_NONE32 = MIN_SCORE;    // "no chain yet" / "no row above": loses to every real opener, cannot wrap when extended

// a value parked in an accumulation register
__device__ __forceinline__ uint32_t aget(const uint32_t& a) { uint32_t v; asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; }
__device__ __forceinline__ void aput(uint32_t& a, const uint32_t v) { asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); }

__device__ __forceinline__ uint32_t sel_lanes(const unsigned long long lanes, const uint32_t v) {      // v in the lanes of the mask, 0 elsewhere
    uint32_t r; asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(lanes)); return r;
}

// In-kernel stamps (diagnostic build only, -DSTITCH_PROFILE): per-wave cycle sums of the column loop's sections (fill_regs.hip's scheme)
#ifdef STITCH_PROFILE
#define RPROF_DECL uint32_t pf_t = (uint32_t)__builtin_readcyclecounter(), pf_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RPROF(k) { const uint32_t pf_n = (uint32_t)__builtin_readcyclecounter(); pf_sum[k] += pf_n - pf_t; pf_t = pf_n; }
#else
#define RPROF_DECL
#define RPROF(k)
#endif

__device__ __forceinline__ uint32_t byte_set_mv(const uint32_t w, const int k, const uint32_t mv) { return (w & ~(7u << (8 * k))) | (mv << (8 * k)); }      // the move code of byte k

// wave-uniform values of one contig's column and the lane's rolling values of pass 1
struct Col32 {
    int32_t match, mismatch, ge, goe, jscore; uint32_t jlen;
    uint32_t q;
    int32_t a, DG; uint32_t DGl;               // the current row's match term and diagonal candidate (score, length)
    int32_t dgm; uint32_t padreg;              // row m's diagonal candidate (its register is one of the first eight)
    int32_t jfix; uint32_t jl1, jmv1;          // circular contigs: what row 1's jump has over the column's (lane 0, consumed by its first row)
};

// ---- pass 1, one row: everything of the cell that needs column j-1 only (dp_core.h row_phase_a without the prefix clips), in
// place; best{diagonal, deletion} is parked in LDS for the merges.
// (Measured dead end, round 3: splitting this sweep into a part that runs BEFORE the team's granules are polled (deletion, diagonal)
// and the jump on top of it afterwards, to overlap the exchange with arithmetic — 15 more instructions per row, no gain: in-kernel
// stamps put the poll at 4 % of the column; a lone wave per SIMD loses its time between dependent instructions, not in the exchange.)
template <int IDX, bool CIRC>
__device__ __forceinline__ void row32_pass1(int32_t& Sreg, uint32_t& Da, uint32_t& Lreg, uint32_t& tbw, Col32& c, const uint32_t xcur, uint32_t& xnext,
                                            const uint32_t* xw_lane, int32_t* bs_lane) {
    constexpr int k = IDX & 3;
    const int32_t a = c.a, DG = c.DG; const uint32_t DGl = c.DGl;
    const int32_t Sold = Sreg, Dold = (int32_t)aget(Da);
    const uint32_t L1 = Lreg + 0x00010001u;                   // (S.len + 1) | (D.len + 1) << 16
    if (IDX > 0) {
        if (k == 3 && IDX >= 7) xnext = xw_lane[((IDX >= 7 ? IDX - 7 : 0) >> 2) * 64];
        const uint32_t xbn = ((k == 0 ? xnext : xcur) >> (8 * (k == 0 ? 3 : k - 1))) & 0xFFu;
        c.a = xbn == c.q ? c.match : c.mismatch;
        c.DG = Sold + c.a; c.DGl = L1 & 0xFFFFu;              // the NEXT row's diagonal: this row's old cell
    }
    const int32_t DE = Dold + c.ge, DO = Sold + c.goe;        // deletion (:328-338): the extension wins ties
    const bool dext = DE >= DO;
    const int32_t BD = dext ? DE : DO;
    const uint32_t BDl = dext ? (L1 >> 16) : (L1 & 0xFFFFu);
    const bool c1 = BD > DG;                                  // (:363-366)
    const int32_t bs2 = c1 ? BD : DG; const uint32_t bs2l = c1 ? BDl : DGl;
    int32_t J = c.jscore + a; uint32_t Jl = c.jlen, jmv = MV_JUMP;
    if (CIRC && k == 3) { J += c.jfix; Jl = c.jl1; jmv = c.jmv1; c.jfix = 0; c.jl1 = c.jlen; c.jmv1 = MV_JUMP; }
    // (:373-382; bit operations, not && / ||: the compiler turns the short-circuit form into a divergent branch per row)
    const bool c3 = (J > bs2) | ((J == bs2) & !c1 & (Jl > DGl));
    const int32_t T = J > bs2 ? J : bs2; const uint32_t Tl = c3 ? Jl : bs2l;      // (where c3 differs from J > bs2 the scores are equal)
    const uint32_t code = (c3 ? jmv : c1 ? (uint32_t)MV_DEL : (uint32_t)MV_DIAG) | (dext ? (uint32_t)TBB_DEXT : 0u);
    tbw = k == 3 ? code : ((tbw << 8) | code);
    bs_lane[IDX * 64] = bs2;
    if (IDX < 8) { if (c.padreg == (uint32_t)IDX) c.dgm = DG; }
    Sreg = T; aput(Da, (uint32_t)BD); Lreg = Tl | (BDl << 16);
}

// what a merge or a prefix clip needs of the column and of the lane (positions are 1-based rows of the contig)
struct Ctx32 {
    int32_t match, mismatch, go, ge, jscore; uint32_t jlen;
    int32_t j1score; uint32_t j1len, j1mv;     // row 1's jump (circular contigs: the better of the column's and the end-to-start jump)
    uint32_t q;
    bool xf, yf;
    int32_t xclip_score; uint32_t row0_len;
    int32_t pos_reg0;                          // position of the row register 0 holds (or would hold): register IDX holds pos_reg0 - IDX
    uint32_t m;
    const uint32_t* col0_len;                  // Slen0 of the contig's rows (cell(i, 0).S.len)
    const uint32_t* xw_lane;
};
__device__ __forceinline__ uint32_t byte_set(const uint32_t w, const int k, const uint32_t mv) { return (w & ~(7u << (8 * k))) | (mv << (8 * k)); }

// the x-prefix and y-prefix clip candidates of a cell whose score so far is bs (:384-399), in the reference's order
template <int IDX>
__device__ __forceinline__ void clips32(const Ctx32& X, int32_t& bs, uint32_t& ln, uint32_t& mv) {
    const int32_t pos = X.pos_reg0 - IDX;
    if (X.xf && X.xclip_score > bs) { bs = X.xclip_score; ln = X.row0_len; mv = MV_XPRE; }
    if (X.yf && pos >= 1 && pos <= (int32_t)X.m) {
        const int32_t yc = X.go + X.ge * pos;
        if (yc > bs) { bs = yc; ln = X.col0_len[pos - 1]; mv = MV_YPRE; }
    }
}
// rows of a group after pass 1, in the columns where a prefix clip can win
template <int IDX>
__device__ __forceinline__ void clip32_row(int32_t& Sreg, uint32_t& Lreg, uint32_t& tbw, const Ctx32& X) {
    constexpr int k = IDX & 3;
    int32_t bs = Sreg; const uint32_t L = Lreg; uint32_t ln = L & 0xFFFFu, mv = 0xFFu;
    clips32<IDX>(X, bs, ln, mv);
    if (mv != 0xFFu) { Sreg = bs; Lreg = (L & 0xFFFF0000u) | ln; tbw = byte_set(tbw, k, mv); }
}
// ---- phase C of dp_core.h for the lanes of `m`: the insertion (bi, il) at its place in the priority order, then the jump and
// the prefix clips again; a cell the insertion does not beat best{diagonal, deletion} in stays as it is
template <int IDX>
__device__ __forceinline__ void merge32_row(int32_t& Sreg, uint32_t& Lreg, uint32_t& tbw, const unsigned long long m, const int32_t bi, const uint32_t il,
                                            const Ctx32& X, const int32_t* bs_lane) {
    constexpr int k = IDX & 3;
    if (sel_lanes(m, 1u) == 0u) return;
    const int32_t bs2 = bs_lane[IDX * 64];
    if (!(bi > bs2)) return;
    const int32_t pos = X.pos_reg0 - IDX;
    const uint32_t xb = (X.xw_lane[(IDX >> 2) * 64] >> (8 * k)) & 0xFFu;
    const int32_t a = xb == X.q ? X.match : X.mismatch;
    int32_t bs = bi; uint32_t ln = il, mv = MV_INS;
    const int32_t J = (pos == 1 ? X.j1score : X.jscore) + a;
    if (J > bs) { bs = J; ln = pos == 1 ? X.j1len : X.jlen; mv = pos == 1 ? X.j1mv : (uint32_t)MV_JUMP; }      // (the == rule needs bs == diagonal: impossible here)
    clips32<IDX>(X, bs, ln, mv);
    Sreg = bs; Lreg = (Lreg & 0xFFFF0000u) | (ln & 0xFFFFu); tbw = byte_set(tbw, k, mv);
}

// ---- pass 1b: the chain of the lane's own openers (score, length, "arrived by an extension")
struct Chain32 { int32_t ge, goe; int32_t Is; uint32_t Il; uint32_t extn; };
template <int IDX>
__device__ __forceinline__ void chain32_row(const int32_t Sreg, const uint32_t Lreg, uint32_t& eb, Chain32& c, int32_t& Is_at, uint32_t& Il_at) {
    constexpr int k = IDX & 3;
    Is_at = c.Is; Il_at = c.Il;
    eb |= c.extn << (8 * k);
    const int32_t ext = c.Is + c.ge, open = Sreg + c.goe;
    const bool isext = ext >= open;                            // the extension wins ties (:321)
    const uint32_t ol = (Lreg & 0xFFFFu) + 1u;
    c.Il = isext ? c.Il + 1u : ol;
    c.Is = isext ? ext : open;
    c.extn = isext ? (uint32_t)TBB_IEXT : 0u;
}

// ---- pass 2, one group in which the arriving chain X is still alive in some lane (fill_regs.hip group_alive, on scores)
struct Alive32 { int32_t ge, goe; int32_t Xs; uint32_t Xl; int32_t Sup; uint32_t xext; unsigned long long alive; };
template <int G>
__device__ __forceinline__ void group32_alive(int32_t& s3, int32_t& s2, int32_t& s1, int32_t& s0, uint32_t& l3, uint32_t& l2, uint32_t& l1, uint32_t& l0,
                                              uint32_t& tbw, Alive32& c, const Ctx32& X, const int32_t* bs_lane) {
    const int32_t x3 = c.Xs, x2 = x3 + c.ge, x1 = x2 + c.ge, x0 = x1 + c.ge;
    const unsigned long long a3 = c.alive & ~__ballot(c.Sup + c.goe > x3);
    const unsigned long long a2 = a3 & ~__ballot(s3 + c.goe > x2);
    const unsigned long long a1 = a2 & ~__ballot(s2 + c.goe > x1);
    const unsigned long long a0 = a1 & ~__ballot(s1 + c.goe > x0);
    tbw |= sel_lanes(a3, c.xext << 24) | sel_lanes(a2, (uint32_t)TBB_IEXT << 16) | sel_lanes(a1, (uint32_t)TBB_IEXT << 8) | sel_lanes(a0, (uint32_t)TBB_IEXT);
    const unsigned long long m3 = a3 & __ballot(x3 >= s3), m2 = a2 & __ballot(x2 >= s2), m1 = a1 & __ballot(x1 >= s1), m0 = a0 & __ballot(x0 >= s0);
    c.Sup = s0;
    if ((m3 | m2 | m1 | m0) != 0ull) {
        merge32_row<4 * G + 3>(s3, l3, tbw, m3, x3, c.Xl, X, bs_lane); merge32_row<4 * G + 2>(s2, l2, tbw, m2, x2, c.Xl + 1u, X, bs_lane);
        merge32_row<4 * G + 1>(s1, l1, tbw, m1, x1, c.Xl + 2u, X, bs_lane); merge32_row<4 * G>(s0, l0, tbw, m0, x0, c.Xl + 3u, X, bs_lane);
    }
    c.Xs = x0 + c.ge; c.Xl += 4u; c.xext = (uint32_t)TBB_IEXT; c.alive = a0;
}

// the lane's running records over a contig's column (rows below m) on 32-bit keys (relative score << 16 | length), as fill_regs.hip
struct Recs32 { uint32_t bw, gw, g1; };
__device__ __forceinline__ void group32_records(Recs32& R, const uint32_t g4, const uint32_t g) {
    R.gw = g4 > R.bw ? g : R.gw;
    R.g1 = (g4 >> 16) > (R.bw >> 16) ? g : R.g1;
    R.bw = g4 > R.bw ? g4 : R.bw;
}
// (every row a lane processes outside row m's group is a real row, and a real cell's score lies in (kbase, kbase + 65536): it is at
// least the column's jump candidate, colmax_prev + a jump score + a mismatch, and at most colmax_prev + match — regs32_plan keeps
// the scoring within KEY_BIAS.  The clamp only serves the registers without a row, whose key nobody looks at.)
__device__ __forceinline__ uint32_t rel_key(const int32_t S, const uint32_t L, const int32_t kbase) {
    return __builtin_amdgcn_perm((uint32_t)(S - kbase), L, 0x05040100u);      // bytes 1..0 of (S - kbase) above bytes 1..0 of L
}
__device__ __forceinline__ uint32_t rel_key_clamped(const int32_t S, const uint32_t L, const int32_t kbase) {
    int32_t r = S - kbase; r = r < 0 ? 0 : (r > 65535 ? 65535 : r);
    return ((uint32_t)r << 16) | (L & 0xFFFFu);
}

template <int I> struct IC { static constexpr int v = I; };
template <int I, typename F> __device__ __forceinline__ void for_groups_down(F&& f) { if constexpr (I >= 0) { f(IC<I>()); for_groups_down<I - 1>(f); } }      // I, I - 1, ..., 0 with a compile-time index

}  // namespace

// NQ = granule registers per lane: 1 for up to 64 active contigs, 4 for up to 256
template <int NQ, bool CIRC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void fill_regs32_kernel(const JobView* __restrict__ jobs, FillShared sh, const uint2* __restrict__ wave_map, uint32_t n_waves) {
    // the launch's waves are dealt densely to the reads' contigs: wave w of the grid = entry w of `wave_map` = {read of the launch,
    // active contig} (fill_regs.hip; a team needs nothing of a workgroup)
    const uint32_t wv = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    if (wv >= n_waves) return;
    const uint2 wm = wave_map[wv];
    const uint32_t job = (uint32_t)__builtin_amdgcn_readfirstlane((int)wm.x);
    const JobView& V = jobs[job];
    const DpParams P = V.P;
    const uint32_t n = V.n, nact = V.nact, Rtot = V.Rtot, C = V.C;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn32[];
    uint8_t* const s_wave = s_dyn32 + (size_t)wave * LDS32_PER_WAVE;
    const bool xf = P.xclip_prefix == 0, yf = P.yclip_prefix == 0;      // (the four penalties go in pairs: aligners/mod.rs:123-131)

    const uint32_t kmine = (uint32_t)__builtin_amdgcn_readfirstlane((int)wm.y);
    if (kmine >= nact) return;
    const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)V.act[kmine]);
    ContigDesc cd = V.cd[c];
    cd.m = (uint32_t)__builtin_amdgcn_readfirstlane((int)cd.m); cd.roff = (uint32_t)__builtin_amdgcn_readfirstlane((int)cd.roff);
    cd.troff = (uint32_t)__builtin_amdgcn_readfirstlane((int)cd.troff); cd.seqoff = (uint32_t)__builtin_amdgcn_readfirstlane((int)cd.seqoff);
    const uint32_t m = cd.m, roff = cd.roff;
    int32_t kopp = -1;
    { const int32_t opp = V.opp_act[c]; if (opp >= 0) for (uint32_t k = 0; k < nact; ++k) if ((int32_t)V.act[k] == opp) kopp = (int32_t)k; }
    kopp = __builtin_amdgcn_readfirstlane(kopp);
    // groups of four rows -> lanes: fill_regs.hip's dealing (every lane's first row in the fullest lane's top register)
    const uint32_t ngr = (m + 3) / 4, gq = ngr / 64, grem = ngr % 64;
    const uint32_t gl = gq + ((uint32_t)lane < grem ? 1u : 0u);
    const uint32_t nrows = 4 * gl;
    const uint32_t rowbase = 4 * ((uint32_t)lane * gq + ((uint32_t)lane < grem ? (uint32_t)lane : grem));
    const uint32_t pad = 4 * ngr - m;
    const int mlane = (int)(gq > 0 ? 63u : grem - 1u);
    const uint32_t gtop = gq + (grem > 0 ? 1u : 0u);
    const uint32_t rsh = (grem > 0 && (uint32_t)lane >= grem) ? 4u : 0u;
    const bool has0 = gl > 0 && rsh == 0u;
    const uint32_t gm = (grem > 0 && gq > 0) ? 1u : 0u;
    const uint32_t padreg = 4u * gm + pad;
    const int32_t ge = P.gap_extend, goe = P.gap_open + P.gap_extend;
    const gptr<const uint8_t> yseq = as_global(V.y);
    const gptr<u32x2> yrec = (gptr<u32x2>)as_global(V.D);          // [Rtot] 8-byte records {SnLen, Ly}, lane-interleaved like the traceback
    const gptr<uint32_t> jt_idx = as_global(V.jt_idx), jt_from = as_global(V.jt_from), Lx = as_global(V.Lx);
    uint8_t* const tb0 = V.tb + roff;
    const bool ymode_global = V.yrec_global != 0;
    const bool local_mode = xf && yf;
    // query-local with several contigs: the end-of-read jump's walk continues in the contig its source cell came FROM, at the source's
    // row (the reference's TB_XJUMP quirk, traceback/mod.rs:329-338) — a cell whose y-suffix clip no score bound of its own contig
    // says anything about.  Every cell's tracker is kept then.
    const bool y_all = yf && !xf && nact > 1;
    const int32_t pos1_0 = (int32_t)(rowbase + nrows);              // 1-based row of register rsh (the lane's last row)
    const int32_t pos_reg0 = pos1_0 + (int32_t)rsh;                 // register IDX holds row pos_reg0 - IDX
    const bool mine = lane == mlane;

    // ---- column 0 (init_matrices :97-186) -----------------------------------------------------------------------------------------
    int32_t S[RMAX]; uint32_t L[RMAX];                              // vector registers: S, and S.len | D.len << 16
    uint32_t DA[RMAX], SNA[RMAX];                                   // accumulation registers: D (read and written once per column), the y-suffix maxima Sn
    {
        uint32_t* const xw0 = (uint32_t*)(s_wave + LDS_XW) + lane;
        for_groups_down<NG - 1>([&](auto gi) __attribute__((always_inline)) {
            constexpr int g = decltype(gi)::v;
            uint32_t w = 0;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                const int32_t pos = pos_reg0 - (4 * g + k);          // 1-based row, or outside 1..m
                uint32_t b = 0xFFu; int32_t s = MIN_SCORE / 2; uint32_t l = 0; int32_t sn = MIN_SCORE;
                const bool owns = (g == 0 ? has0 : (gl > 0 && (uint32_t)g < gtop)) && pos >= 1 && pos <= (int32_t)m && (uint32_t)(4 * g + k) >= rsh;
                if (owns) {
                    const uint32_t row = (uint32_t)pos - 1u, tr = cd.troff + row;
                    s = sh.S0[tr]; l = sh.Slen0[tr];
                    if (yf && sh.SnSet0[tr]) sn = sh.Sn0[tr];
                    u32x2 rec; rec.x = sh.Slen0[tr]; rec.y = (yf && sh.SnSet0[tr]) ? n : 0u;
                    yrec[roff + (4u * g + (uint32_t)k) * 64u + (uint32_t)lane] = rec;
                    V.SmoveF[roff + row] = TB_NONE; V.ImoveF[roff + row] = TB_NONE;
                    b = V.xseq[cd.seqoff + row];
                }
                S[4 * g + k] = s; aput(DA[4 * g + k], (uint32_t)MIN_SCORE); L[4 * g + k] = l & 0xFFFFu; aput(SNA[4 * g + k], (uint32_t)sn);
                w |= b << (8 * k);
            }
            xw0[g * 64] = w;
        });
    }
    if (lane == 0) V.Lx[(size_t)c * (n + 1)] = sh.lx0[c];
    int32_t vrun = sh.base0[c].score;                                // the contig's running column maximum (local mode threshold)
    const uint32_t trm = cd.troff + m - 1;
    bool rowm_xsuf = sh.Smove0[trm] == TB_XCLIP_SUFFIX; int32_t rowm_S = sh.S0[trm]; uint32_t rowm_len = sh.Slen0[trm];
    int32_t rm_run = sh.S0[trm];                                     // running maximum of row m over the columns (query-local threshold)
    const int32_t circular = P.circular;
    // the contigs' column arg-max of column j-1: lane l holds active contigs l, l + 64, ...: score, and len << 16 | from
    int32_t gsc[NQ]; uint32_t glf[NQ]; uint32_t actid[NQ];
    int32_t gmax = INT32_MIN;
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) {
        const uint32_t k = (uint32_t)lane + 64u * qq;
        gsc[qq] = INT32_MIN; glf[qq] = 0u; actid[qq] = 0u;
        if (k < nact) { actid[qq] = V.act[k]; const JumpBase b = sh.base0[V.act[k]]; gsc[qq] = b.score; glf[qq] = ((b.len & 0xFFFFu) << 16) | (b.from & 0xFFFFu); gmax = b.score > gmax ? b.score : gmax; }
    }
    gmax = (int32_t)(wave_max_u32((uint32_t)gmax ^ 0x80000000u) ^ 0x80000000u);
    int32_t colmax_prev = gmax;                                      // best score of any contig in column j-1

    const __amdgpu_buffer_rsrc_t ryr = __builtin_amdgcn_make_buffer_rsrc((uint8_t*)V.D + 8ull * roff, 0, 0x7FFFFFFF, RSRC_WORD3);
    const __amdgpu_buffer_rsrc_t rxc = __builtin_amdgcn_make_buffer_rsrc((void*)V.xchg, 0, 0x7FFFFFFF, RSRC_WORD3);

    int32_t sn0; uint32_t ly0; row0_init_sn(P, n, sn0, ly0);
    Row0 r0prev = row0_column0();

    // ---- the insertion chain across the lanes (fill_regs.hip chain_across_lanes, on 32-bit scores: a two-register scan) --------------
    auto chain_across_lanes = [&](const int32_t Es, const uint32_t El, const bool exitext, const Row0& r0, int32_t& Iin_s, uint32_t& Iin_l, uint32_t& extin) __attribute__((always_inline)) {
        const int32_t pos_exit = pos1_0 + 1;
        ScanEl el; el.key = gl == 0 ? INT32_MIN : Es - ge * pos_exit; el.q = lane;
        wave_scan(el);                                                // inclusive; the earlier lane wins ties
        const int32_t rt_key = from_prev_lane(el.key, INT32_MIN); const uint32_t w = (uint32_t)from_prev_lane(el.q, 0);
        const int32_t Ew_s = __builtin_amdgcn_ds_bpermute((int)(w << 2), Es); const uint32_t Ew_l = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(w << 2), (int)El);
        const int32_t first_pos = pos1_0 + 1 - 4 * (int32_t)gl;
        const int32_t seed_key = r0.S + goe - ge;                     // row 0's opener at position 1; the earliest opener: it wins ties
        const bool seed = lane == 0 || seed_key >= rt_key;
        const uint32_t wn = w + 1u;
        const int32_t w_exit = 4 * (int32_t)(wn * gq + (wn < grem ? wn : grem)) + 1;
        const int32_t dist = seed ? first_pos - 1 : first_pos - w_exit;
        Iin_s = (seed ? r0.S + goe : Ew_s) + ge * dist;
        Iin_l = (seed ? r0.Slen + 1u : Ew_l) + (uint32_t)dist;
        const int32_t ext_prev = from_prev_lane(exitext ? 1 : 0, 0);
        extin = (lane == 0) ? 0u : ((!seed && w + 1u == (uint32_t)lane && ext_prev == 0) ? 0u : (uint32_t)TBB_IEXT);
    };

    uint32_t ychunk = 0;
    RPROF_DECL
    for (uint32_t j = 1; j <= n; ++j) {
        const bool lastcol = j == n;
        RPROF(7)
        if (((j - 1) & 63u) == 0) ychunk = (j - 1 + lane < n) ? (uint32_t)yseq[j - 1 + lane] : 0u;
        const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)ychunk, (int)((j - 1) & 63u)) & 0xFFu;
        const Row0 r0 = row0_step(P, j, n, sn0, ly0);                // row 0 of this column (closed form, :188-239)

        const uint32_t* const xw_lane = (const uint32_t*)(s_wave + LDS_XW) + lane;
        int32_t* const bs_lane = (int32_t*)(s_wave + LDS_BS) + lane;
        uint32_t tbv[NG];
        // (the group guards are compared where they are used: kept as twenty hoisted masks per sweep they take scalar registers by the
        // dozen, and what depends only on the lane would be pinned in vector registers over the whole read)
        uint32_t gtop_x = gtop; asm volatile("" : "+s"(gtop_x));
        bool has0_x; { uint32_t h = has0 ? 1u : 0u; asm volatile("" : "+v"(h)); has0_x = h != 0u; }
#define GUARD32(g) ((g) == 0 ? has0_x : ({ asm volatile("" : "+s"(gtop_x)); (uint32_t)(g) < gtop_x; }))
        RPROF(0)
        // ---- poll the team's granules of column j-1 (two tagged 8-byte halves per contig: {column, len, from} and {column, score}) ----
        if (j > 1) {
            const uint32_t want = j - 1;
            int lane_p = lane; asm volatile("" : "+v"(lane_p));
            const uint32_t gso = (want & 1u) * C * 16u;
            const uint32_t t0 = (uint32_t)wall_clock64();
            for (uint32_t spins = 1;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int qq = 0; qq < NQ; ++qq) {
                    const uint32_t k = (uint32_t)lane_p + 64u * qq;
                    if (k < nact) {
                        const u32x4 g4 = __builtin_amdgcn_raw_buffer_load_b128(rxc, 16u * k, gso, AUX_SC1 | AUX_VOLATILE);
                        glf[qq] = g4.x; gsc[qq] = (int32_t)g4.z; ok &= (g4.y == want) && (g4.w == want);
                    }
                }
                if (__all(ok)) break;
                if ((spins & 1023u) == 0) {
                    const uint32_t e_seen = __builtin_amdgcn_raw_buffer_load_b32(rxc, 0u, 32u * C, AUX_SC1 | AUX_VOLATILE);
                    if (e_seen != 0u || (uint32_t)wall_clock64() - t0 > 400000000u) { if (lane == 0) *V.err = 1; return; }      // 4 s: a partner is not resident
                }
                __builtin_amdgcn_s_sleep(4);
            }
            int32_t best = INT32_MIN;
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) best = ((uint32_t)lane + 64u * qq < nact && gsc[qq] > best) ? gsc[qq] : best;
            colmax_prev = (int32_t)(wave_max_u32((uint32_t)best ^ 0x80000000u) ^ 0x80000000u);
            gmax = colmax_prev > gmax ? colmax_prev : gmax;
        }
        RPROF(1)
        // ---- best jump out of column j-1 for this contig (multi_contig_aligner.rs:292-331) --------------------------------------------
        auto sc_of = [&](uint32_t k) -> int32_t { int32_t v = gsc[0];
#pragma unroll
            for (int qq = 1; qq < NQ; ++qq) v = (k >> 6) == (uint32_t)qq ? gsc[qq] : v;
            return __builtin_amdgcn_readlane(v, (int)(k & 63u)); };
        auto lf_of = [&](uint32_t k) -> uint32_t { uint32_t v = glf[0];
#pragma unroll
            for (int qq = 1; qq < NQ; ++qq) v = (k >> 6) == (uint32_t)qq ? glf[qq] : v;
            return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(k & 63u)); };
        auto act_of = [&](uint32_t k) -> uint32_t { uint32_t v = actid[0];
#pragma unroll
            for (int qq = 1; qq < NQ; ++qq) v = (k >> 6) == (uint32_t)qq ? actid[qq] : v;
            return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(k & 63u)); };
        JumpInfo ji;
        {
            // inter-contig: the best OTHER contig by (score, len), the LAST aligner on full ties (max_by_key)
            unsigned long long ik = 0;
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) {
                const uint32_t k = (uint32_t)lane + 64u * qq;
                if (k < nact && k != kmine && (int32_t)k != kopp) {
                    const unsigned long long key = ((unsigned long long)((uint32_t)gsc[qq] ^ 0x80000000u) << 32) | ((unsigned long long)(glf[qq] >> 16) << 16) | (k + 1);
                    ik = key > ik ? key : ik;
                }
            }
            ik = wave_max_u64(ik);
            const uint32_t kbest1 = (uint32_t)(ik & 0xFFFFu);
            { const uint32_t lf = lf_of(kmine); ji.score = sc_of(kmine) + P.jump_same; ji.len = lf >> 16; ji.idx = c; ji.from = lf & 0xFFFFu; }
            if (kopp >= 0) { const int32_t sc = sc_of((uint32_t)kopp) + P.jump_opp; if (sc > ji.score) { const uint32_t lf = lf_of((uint32_t)kopp); ji.score = sc; ji.len = lf >> 16; ji.idx = act_of((uint32_t)kopp); ji.from = lf & 0xFFFFu; } }
            if (kbest1 != 0) { const uint32_t kw = kbest1 - 1; const int32_t sc = sc_of(kw) + P.jump_inter; if (sc > ji.score) { const uint32_t lf = lf_of(kw); ji.score = sc; ji.len = lf >> 16; ji.idx = act_of(kw); ji.from = lf & 0xFFFFu; } }
        }
        bool circ = false;
        if (CIRC) { ColCtx cc; cc.jump = ji; cc.circ_ok = (circular && !rowm_xsuf) ? 1 : 0; cc.circ_score = rowm_S; cc.circ_len = rowm_len + 1; circ = local_row1_circ(cc); }
        if (lane == 0) { jt_idx[(size_t)c * (n + 1) + j] = ji.idx; jt_from[(size_t)c * (n + 1) + j] = ji.from; }
        const int32_t jscore = __builtin_amdgcn_readfirstlane(ji.score); const uint32_t jlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)ji.len);
        const int32_t j1score = circ ? __builtin_amdgcn_readfirstlane(rowm_S) : jscore;
        const uint32_t j1len = circ ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(rowm_len + 1)) : jlen, j1mv = circ ? (uint32_t)MV_CIRC : (uint32_t)MV_JUMP;
        const int32_t kbase = colmax_prev - KEY_BIAS;

        Ctx32 X;
        X.match = P.match; X.mismatch = P.mismatch; X.go = P.gap_open; X.ge = ge; X.jscore = jscore; X.jlen = jlen; X.j1score = j1score; X.j1len = j1len; X.j1mv = j1mv;
        int32_t pos_x = pos_reg0; asm volatile("" : "+v"(pos_x));      // (opaque per column: the compiler would otherwise compute every row's position-dependent
                                                                       // term once per read and keep 80 of them per lane — in scratch)
        X.q = q; X.xf = xf; X.yf = yf; X.row0_len = r0.Slen; X.pos_reg0 = pos_x; X.m = m; X.col0_len = sh.Slen0 + cd.troff;
        { const int32_t go_j = P.gap_open + P.gap_extend * (int32_t)j; X.xclip_score = P.xclip_prefix + (P.yclip_prefix > go_j ? P.yclip_prefix : go_j); }      // :304-308
        X.xw_lane = xw_lane;

        RPROF(2)
        // ---- pass 1 -----------------------------------------------------------------------------------------------------------------
        Col32 cx;
        cx.match = P.match; cx.mismatch = P.mismatch; cx.ge = ge; cx.goe = goe; cx.jscore = jscore; cx.jlen = jlen; cx.q = q;
        cx.dgm = 0; cx.padreg = padreg;
        cx.jfix = lane == 0 ? j1score - jscore : 0; cx.jl1 = lane == 0 ? j1len : jlen; cx.jmv1 = lane == 0 ? j1mv : (uint32_t)MV_JUMP;
        uint32_t xwA = xw_lane[(gtop - 1u) * 64], xwB = xwA;
        cx.a = (xwA >> 24) == q ? cx.match : cx.mismatch;
        {   // the row above a lane's first row: the previous lane's last row (its register rsh); row 0 of column j-1 for lane 0
            const int32_t s_last = rsh ? S[4] : S[0]; const uint32_t l_last = (rsh ? L[4] : L[0]) & 0xFFFFu;
            cx.DG = from_prev_lane(s_last, r0prev.S) + cx.a; cx.DGl = (uint32_t)from_prev_lane((int)l_last, (int)r0prev.Slen) + 1u;
        }
        // the prefix clips can only win below this bound of every cell's jump candidate
        const int32_t jw_floor = (j1score < jscore ? j1score : jscore) + (P.mismatch < P.match ? P.mismatch : P.match);
        const bool may_clip = (xf && jw_floor < X.xclip_score) || (yf && jw_floor < goe);
        for_groups_down<NG - 1>([&](auto gi) __attribute__((always_inline)) {
            constexpr int g = decltype(gi)::v;
            if (GUARD32(g)) {
                uint32_t tbw = 0;
                uint32_t& xc = (g & 1) ? xwB : xwA; uint32_t& xn = (g & 1) ? xwA : xwB;
                row32_pass1<4 * g + 3, CIRC>(S[4 * g + 3], DA[4 * g + 3], L[4 * g + 3], tbw, cx, xc, xn, xw_lane, bs_lane);
                row32_pass1<4 * g + 2, CIRC>(S[4 * g + 2], DA[4 * g + 2], L[4 * g + 2], tbw, cx, xc, xn, xw_lane, bs_lane);
                row32_pass1<4 * g + 1, CIRC>(S[4 * g + 1], DA[4 * g + 1], L[4 * g + 1], tbw, cx, xc, xn, xw_lane, bs_lane);
                row32_pass1<4 * g, CIRC>(S[4 * g], DA[4 * g], L[4 * g], tbw, cx, xc, xn, xw_lane, bs_lane);
                if (__builtin_expect(may_clip, 0)) {
                    clip32_row<4 * g + 3>(S[4 * g + 3], L[4 * g + 3], tbw, X); clip32_row<4 * g + 2>(S[4 * g + 2], L[4 * g + 2], tbw, X);
                    clip32_row<4 * g + 1>(S[4 * g + 1], L[4 * g + 1], tbw, X); clip32_row<4 * g>(S[4 * g], L[4 * g], tbw, X);
                }
                tbv[g] = tbw;
            }
        });
        RPROF(3)
        // ---- pass 1b: the chain of the lane's own openers ---------------------------------------------------------------------------------
        // (as in fill_regs.hip: every cell of the column is at least its jump candidate, >= jw_floor; a chain below that is dead — it can
        // reach no cell and only decays — and a cell opens a chain worth following only if S' + go + ge >= jw_floor.  A group in which
        // no lane carries a live chain and no lane holds such a cell is left alone; its "I extended" bits are never read, the walk reads
        // them only along a live chain.)
        Chain32 cl; cl.ge = ge; cl.goe = goe; cl.Is = CHAIN_NONE32; cl.Il = 0u; cl.extn = 0u;
        const int32_t open_thr = jw_floor - goe;
        for_groups_down<NG - 1>([&](auto gi) __attribute__((always_inline)) {
            constexpr int g = decltype(gi)::v;
            if (GUARD32(g)) {
                const int32_t h32 = S[4 * g + 3] > S[4 * g + 2] ? S[4 * g + 3] : S[4 * g + 2], h10 = S[4 * g + 1] > S[4 * g] ? S[4 * g + 1] : S[4 * g];
                const bool hot = (h32 > h10 ? h32 : h10) >= open_thr || cl.Is >= jw_floor;
                if (__ballot(hot) == 0ull) return;
                uint32_t eb = 0u; int32_t i3, i2, i1, i0; uint32_t n3, n2, n1, n0;
                chain32_row<4 * g + 3>(S[4 * g + 3], L[4 * g + 3], eb, cl, i3, n3); chain32_row<4 * g + 2>(S[4 * g + 2], L[4 * g + 2], eb, cl, i2, n2);
                chain32_row<4 * g + 1>(S[4 * g + 1], L[4 * g + 1], eb, cl, i1, n1); chain32_row<4 * g>(S[4 * g], L[4 * g], eb, cl, i0, n0);
                uint32_t tbw = tbv[g] | eb;
                const unsigned long long m3 = __ballot(i3 >= S[4 * g + 3]), m2 = __ballot(i2 >= S[4 * g + 2]), m1 = __ballot(i1 >= S[4 * g + 1]), m0 = __ballot(i0 >= S[4 * g]);
                if (__builtin_expect((m3 | m2 | m1 | m0) != 0ull, 0)) {
                    merge32_row<4 * g + 3>(S[4 * g + 3], L[4 * g + 3], tbw, m3, i3, n3, X, bs_lane); merge32_row<4 * g + 2>(S[4 * g + 2], L[4 * g + 2], tbw, m2, i2, n2, X, bs_lane);
                    merge32_row<4 * g + 1>(S[4 * g + 1], L[4 * g + 1], tbw, m1, i1, n1, X, bs_lane); merge32_row<4 * g>(S[4 * g], L[4 * g], tbw, m0, i0, n0, X, bs_lane);
                }
                tbv[g] = tbw;
            }
        });
        RPROF(4)
        // (a chain that leaves its lane dead arrives dead everywhere; so does row 0's opener when r0.S + go + ge < jw_floor: then there
        // is nothing for the scan to carry and nothing for pass 2 to repair)
        int32_t Iin_s = CHAIN_NONE32; uint32_t Iin_l = 0u, extin = 0u;
        const bool chains_dead = __ballot(cl.Is >= jw_floor) == 0ull && r0.S + goe < jw_floor;
        if (!chains_dead) chain_across_lanes(cl.Is, cl.Il, cl.extn != 0u, r0, Iin_s, Iin_l, extin);

        // ---- pass 2 + tail: the arriving chain while it is alive; records, y-suffix trackers, the traceback dword -----------------------------
        const gptr<uint32_t> tbcol = (gptr<uint32_t>)as_global(tb0 + (size_t)(j - 1) * Rtot);
        Recs32 R; R.bw = 0; R.gw = 0; R.g1 = 0;
        uint32_t tbw0 = 0;
        const uint32_t ycol = n - j;
        const int32_t ybase = local_mode ? (ymode_global ? gmax : vrun) : rm_run;
        Alive32 ca; ca.ge = ge; ca.goe = goe; ca.Xs = Iin_s; ca.Xl = Iin_l; ca.Sup = CHAIN_NONE32; ca.xext = extin;
        ca.alive = chains_dead ? 0ull : __ballot(gl > 0 && Iin_s >= jw_floor);      // (a chain that arrives dead stays dead)
        const unsigned long long have0 = __ballot(has0);
        for_groups_down<NG - 1>([&](auto gi) __attribute__((always_inline)) {
            constexpr int g = decltype(gi)::v;
            if (({ asm volatile("" : "+s"(gtop_x)); (uint32_t)g < gtop_x; })) {
                if (g == 0) ca.alive &= have0;
                uint32_t tbw = tbv[g];
                if (ca.alive != 0ull) group32_alive<g>(S[4 * g + 3], S[4 * g + 2], S[4 * g + 1], S[4 * g], L[4 * g + 3], L[4 * g + 2], L[4 * g + 1], L[4 * g], tbw, ca, X, bs_lane);
                if (g == 0 ? has0_x : true) {
                    const bool rmg = g < 2 && (uint32_t)g == gm && mine;      // the group of row m, in its lane: registers <= pad hold no row below m
                    const uint32_t k3 = (rmg && 3u <= pad) ? 0u : rel_key(S[4 * g + 3], L[4 * g + 3], kbase), k2 = (rmg && 2u <= pad) ? 0u : rel_key(S[4 * g + 2], L[4 * g + 2], kbase);
                    const uint32_t k1 = (rmg && 1u <= pad) ? 0u : rel_key(S[4 * g + 1], L[4 * g + 1], kbase), k0 = rmg ? 0u : rel_key(S[4 * g], L[4 * g], kbase);
                    const uint32_t g4 = (k3 > k2 ? k3 : k2) > (k1 > k0 ? k1 : k0) ? (k3 > k2 ? k3 : k2) : (k1 > k0 ? k1 : k0);
                    group32_records(R, g4, (uint32_t)g);
                    if (yf) {
                        // may a cell of this group still matter as a y-suffix clip?  (header: local / query-local)
                        const int32_t gs = (int32_t)(g4 >> 16) + kbase;      // largest score of the group's rows below m (clamped keys only under-estimate a cell that is out of range)
                        const int32_t via_ins = P.gap_open + ge * ((int32_t)m - (pos_x - 4 * g));
                        const int32_t bound = local_mode ? gs : gs + (P.jump_same > via_ins ? P.jump_same : via_ins);
                        const bool cand = g4 != 0u && (y_all || (local_mode ? bound >= ybase : bound > ybase));
                        if (__builtin_expect(__any(cand), 0)) {
#pragma unroll
                            for (int k = 3; k >= 0; --k) {
                                const bool below_m = !(rmg && (uint32_t)k <= pad);
                                const int32_t v = S[4 * g + k]; const uint32_t ln = L[4 * g + k] & 0xFFFFu; const int32_t snv = (int32_t)aget(SNA[4 * g + k]);
                                if (cand && below_m && (v > snv || (v == snv && ln > 0u))) {      // (:431-447; cell(i, n).S.len is still 0 for i < m)
                                    aput(SNA[4 * g + k], (uint32_t)v);
                                    u32x2 rec; rec.x = ln; rec.y = ycol;
                                    __builtin_amdgcn_raw_buffer_store_b64(rec, ryr, 8u * (uint32_t)lane, (4 * g + k) * 512, 0);
                                }
                            }
                        }
                    }
                    __builtin_nontemporal_store(tbw, tbcol + (g * 64 + lane));
                    if (g < 2 && (uint32_t)g == gm) tbw0 = tbw;
                }
            }
        });

        RPROF(5)
        // ---- the contig's epilogue: wave reductions over rows < m, row m, the column arg-max granule ------------------------------------------
        {
            const uint32_t xw = wave_max_u32(R.bw);
            // S and packed lengths of group Gq of lane Lq (registers that hold row m or no row: key 0)
            auto fetch4 = [&](const uint32_t Gq, const int Lq, int32_t (&ws)[4], uint32_t (&wl)[4], uint32_t (&wk)[4]) {
                ws[0] = ws[1] = ws[2] = ws[3] = 0; wl[0] = wl[1] = wl[2] = wl[3] = 0u;
                for_groups_down<NG - 1>([&](auto gi) __attribute__((always_inline)) {
                    constexpr int g = decltype(gi)::v;
                    if (Gq == (uint32_t)g) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { ws[k] = __builtin_amdgcn_readlane(S[4 * g + k], Lq); wl[k] = (uint32_t)__builtin_amdgcn_readlane((int)L[4 * g + k], Lq) & 0xFFFFu; }
                    }
                });
#pragma unroll
                for (int k = 0; k < 4; ++k) wk[k] = rel_key_clamped(ws[k], wl[k], kbase);
                if (Gq == gm && Lq == mlane) { wk[0] = 0u; if (pad >= 1u) wk[1] = 0u; if (pad >= 2u) wk[2] = 0u; if (pad >= 3u) wk[3] = 0u; }
            };
            auto row_of = [&](const uint32_t Gq, const uint32_t k, const uint32_t Lq) -> uint32_t {      // 0-based row of register 4 Gq + k of lane Lq
                const uint32_t glL = gq + (Lq < grem ? 1u : 0u), rbL = 4u * (Lq * gq + (Lq < grem ? Lq : grem)), rshL = (grem > 0 && Lq >= grem) ? 4u : 0u;
                return rbL + 4u * glL - 1u - (4u * Gq + k - rshL);
            };
            XsRec xb_; xb_.v = MIN_SCORE; xb_.len = 0; xb_.row = 0;     // S[curr][m] starts at MIN, length 0 (:236-238)
            CmRec cb_; cb_.v = r0.S; cb_.row = 0; cb_.len = r0.Slen;      // get_jump_info starts at row 0 (:677-697)
            int32_t ws[4]; uint32_t wl[4], wk[4]; int L1 = -1; uint32_t G1 = 0;
            if (xw != 0u) {
                const uint32_t smax = xw >> 16;
                L1 = (int)__builtin_ctzll(__ballot((R.bw >> 16) == smax));
                G1 = (uint32_t)__builtin_amdgcn_readlane((int)R.g1, L1);
                fetch4(G1, L1, ws, wl, wk);
                const uint32_t k1 = (wk[3] >> 16) == smax ? 3u : (wk[2] >> 16) == smax ? 2u : (wk[1] >> 16) == smax ? 1u : 0u;
                if (ws[k1] > cb_.v) { cb_.v = ws[k1]; cb_.row = row_of(G1, k1, (uint32_t)L1) + 1u; cb_.len = wl[k1]; }
                if (xf) {
                    const int Lw = (int)__builtin_ctzll(__ballot(R.bw == xw));
                    const uint32_t Gw = (uint32_t)__builtin_amdgcn_readlane((int)R.gw, Lw);
                    if (Lw != L1 || Gw != G1) fetch4(Gw, Lw, ws, wl, wk);
                    const uint32_t kw = wk[3] == xw ? 3u : wk[2] == xw ? 2u : wk[1] == xw ? 1u : 0u;
                    xb_.v = ws[kw]; xb_.len = wl[kw]; xb_.row = row_of(Gw, kw, (uint32_t)Lw) + 1u;
                }
            }
            // ---- row m (:350-351 seeded selection, :406-447 for i == m) -------------------------------------------------------------
            int32_t wmS = 0; uint32_t wmL = 0, wmSn = 0;
            for_groups_down<7>([&](auto ri) __attribute__((always_inline)) { constexpr int r = decltype(ri)::v; if (padreg == (uint32_t)r) { wmS = S[r]; wmL = L[r]; wmSn = aget(SNA[r]); } });
            const int32_t ownS = __builtin_amdgcn_readlane(wmS, mlane); const uint32_t ownSl = (uint32_t)__builtin_amdgcn_readlane((int)wmL, mlane) & 0xFFFFu;
            const int32_t ownSn = __builtin_amdgcn_readlane((int)wmSn, mlane);
            const int32_t ownDG = __builtin_amdgcn_readlane(cx.dgm, mlane);
            const uint32_t ownByte = (uint32_t)__builtin_amdgcn_readlane((int)((tbw0 >> (8u * pad)) & 0xFFu), mlane);
            int32_t Sm; uint32_t Slm, mvm; bool do_x_m = false;
            uint32_t lx = xb_.row == 0u ? 0u : m - xb_.row;
            if (xf && rowm_run_wins(xb_.v, ownS, ownDG)) { Sm = xb_.v; Slm = xb_.len; mvm = MV_XSUF; }
            else { Sm = ownS; Slm = ownSl; mvm = ownByte & 7u; if (xf && ownSl > xb_.len) { do_x_m = true; lx = 0u; } }
            if (Sm > cb_.v) { cb_.v = Sm; cb_.row = m; cb_.len = Slm; }
            if (lane == 0) {      // the column arg-max is complete: announce it before the column's remaining work
                u32x4 g4; g4.x = (((cb_.len + 1u) & 0xFFFFu) << 16) | (cb_.row & 0xFFFFu); g4.y = j; g4.z = (uint32_t)cb_.v; g4.w = j;
                __builtin_amdgcn_raw_buffer_store_b128(g4, rxc, 0u, (uint32_t)__builtin_amdgcn_readfirstlane((int)(((j & 1u) * C + kmine) * 16u)), AUX_SC1 | AUX_VOLATILE);
            }
            if (cb_.v > vrun) vrun = cb_.v;
            if (Sm > rm_run) rm_run = Sm;
            rowm_xsuf = mvm == MV_XSUF; rowm_S = Sm; rowm_len = Slm;
            if (mine) {
                for_groups_down<7>([&](auto ri) __attribute__((always_inline)) { constexpr int r = decltype(ri)::v; if (padreg == (uint32_t)r) { S[r] = Sm; L[r] = (L[r] & 0xFFFF0000u) | (Slm & 0xFFFFu); } });
                ((gptr<uint8_t>)as_global(tb0 + (size_t)(j - 1) * Rtot))[gm * 256u + 4u * (uint32_t)lane + pad] = (uint8_t)(mvm | (ownByte & (TBB_IEXT | TBB_DEXT)));
                if (yf) {
                    const uint32_t rl = lastcol ? (do_x_m ? ownSl : xb_.len) : 0u;
                    if (Sm > ownSn || (Sm == ownSn && Slm > rl)) {
                        for_groups_down<7>([&](auto ri) __attribute__((always_inline)) { constexpr int r = decltype(ri)::v; if (padreg == (uint32_t)r) aput(SNA[r], (uint32_t)Sm); });
                        u32x2 rec; rec.x = Slm; rec.y = n - j;
                        __builtin_amdgcn_raw_buffer_store_b64(rec, ryr, 8u * (padreg * 64u + (uint32_t)lane), 0, 0);
                    }
                }
                Lx[(size_t)c * (n + 1) + j] = lx;
            }
        }
        r0prev = r0;
        RPROF(6)
#undef GUARD32
    }
#ifdef STITCH_PROFILE
    if (lane == 0) { unsigned long long* const pf = (unsigned long long*)((uint8_t*)V.err + 16); for (int k = 0; k < 8; ++k) atomicAdd(pf + k, (unsigned long long)pf_sum[k]); atomicAdd(pf + 8, 1ull); }
#endif
    // ---- column n's arrays for the fix-up kernel (single_contig_aligner.rs:453-555): S, its lengths, the insertion chain at every row
    // (recomputed from the final cells: an opener taken from a merged cell gives the same chain), the y-suffix trackers ----------------
    {
        int32_t Ls = CHAIN_NONE32; uint32_t Ll = 0u; bool lext = false;
        for_groups_down<NG - 1>([&](auto gi) __attribute__((always_inline)) {
            constexpr int g = decltype(gi)::v;
            if (g == 0 ? has0 : (uint32_t)g < gtop) {
#pragma unroll
                for (int k = 3; k >= 0; --k) {
                    const int32_t ext = Ls + ge, open = S[4 * g + k] + goe; lext = ext >= open;
                    Ll = lext ? Ll + 1u : (L[4 * g + k] & 0xFFFFu) + 1u; Ls = lext ? ext : open;
                }
            }
        });
        int32_t Is; uint32_t Il, extin;
        chain_across_lanes(Ls, Ll, lext, r0prev, Is, Il, extin);       // (r0prev: row 0 of column n as the fill's init_column left it)
        for_groups_down<NG - 1>([&](auto gi) __attribute__((always_inline)) {
            constexpr int g = decltype(gi)::v;
            if (g == 0 ? has0 : (uint32_t)g < gtop) {
#pragma unroll
                for (int k = 3; k >= 0; --k) {
                    const int32_t pos = pos_reg0 - (4 * g + k);
                    if (pos >= 1 && pos <= (int32_t)m) {
                        const uint32_t r = roff + (uint32_t)pos - 1u;
                        V.S[r] = S[4 * g + k]; V.Slen[r] = L[4 * g + k] & 0xFFFFu; V.Ival[r] = Is; V.Ilen[r] = Il;
                        V.Sn[r] = yf ? (int32_t)aget(SNA[4 * g + k]) : MIN_SCORE;
                        const u32x2 rec = yrec[roff + (4u * g + (uint32_t)k) * 64u + (uint32_t)lane];
                        V.SnLen[r] = yf ? rec.x : 0u; V.Ly[r] = yf ? rec.y : 0u;
                    }
                    const int32_t ext = Is + ge, open = S[4 * g + k] + goe; const bool e2 = ext >= open;
                    Il = e2 ? Il + 1u : (L[4 * g + k] & 0xFFFFu) + 1u; Is = e2 ? ext : open;
                }
            }
        });
    }
}

uint32_t fill_regs32_rows_per_wave() { return 64u * RMAX; }
// workgroups of four waves one CU holds at once, as the runtime's occupancy calculator sees it (one: a wave takes its SIMD's whole
// register file); 0: the kernel cannot run
int fill_regs32_workgroups_per_cu() {
    const void* kernels[4] = {(const void*)fill_regs32_kernel<1, false>, (const void*)fill_regs32_kernel<4, false>, (const void*)fill_regs32_kernel<1, true>, (const void*)fill_regs32_kernel<4, true>};
    int least = 1 << 30;
    for (const void* k : kernels) {
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { (void)hipGetLastError(); return 0; }
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, (size_t)4 * LDS32_PER_WAVE) != hipSuccess) { (void)hipGetLastError(); return 0; }
        least = nb < least ? nb : least;
    }
    return least;
}
void launch_fill_regs32(const JobView* d_jobs, const uint2* d_wave_map, uint32_t n_waves, uint32_t max_nact, bool circular, const FillShared& sh, hipStream_t stream) {
    const dim3 grid((n_waves + 3) / 4), block(256); const size_t lds = (size_t)4 * LDS32_PER_WAVE;
    if (max_nact <= 64) { if (circular) hipLaunchKernelGGL((fill_regs32_kernel<1, true>), grid, block, lds, stream, d_jobs, sh, d_wave_map, n_waves); else hipLaunchKernelGGL((fill_regs32_kernel<1, false>), grid, block, lds, stream, d_jobs, sh, d_wave_map, n_waves); }
    else { if (circular) hipLaunchKernelGGL((fill_regs32_kernel<4, true>), grid, block, lds, stream, d_jobs, sh, d_wave_map, n_waves); else hipLaunchKernelGGL((fill_regs32_kernel<4, false>), grid, block, lds, stream, d_jobs, sh, d_wave_map, n_waves); }
}

}  // namespace stitch
