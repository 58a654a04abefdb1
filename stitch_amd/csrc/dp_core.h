// Core arithmetic of the jump-aware affine-gap DP, restructured for a column-synchronous wave64 kernel.
//
// The reference fills a column row by row (fg-stitch-lib/src/align/aligners/single_contig_aligner.rs:292-451).
// Here a column is filled in three data-parallel phases per 64xR-row tile (DESIGN.md "Kernel 1"):
//   A  row-local candidates that only need column j-1 (diagonal, deletion, jump, prefix clips)      :328-399
//   B  the insertion chain I[i] = max(I[i-1]+ge, S[i-1]+go+ge) as a max-plus prefix scan            :314-326
//   C  final move selection in the reference's priority order, suffix-clip tracking, traceback byte  :340-449
// Everything in this header is plain integer code marked STITCH_HD so that the HIP kernels and the lane-serial
// emulator used by the CPU unit tests (tests/emu) execute the same statements.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define STITCH_HD __host__ __device__ __forceinline__
#else
#define STITCH_HD inline
#endif

namespace stitch {

constexpr int32_t MIN_SCORE = -858993459;          // aligners/constants.rs:7
constexpr int32_t KEY_NEG_INF = INT32_MIN / 2;     // padding rows in the insertion scan: never wins

// Reference traceback move codes (align/traceback/mod.rs:47-57); used by the walk.
enum : uint8_t { TB_START = 0, TB_INS = 1, TB_DEL = 2, TB_SUBST = 3, TB_MATCH = 4, TB_XCLIP_PREFIX = 5,
                 TB_XCLIP_SUFFIX = 6, TB_YCLIP_PREFIX = 7, TB_YCLIP_SUFFIX = 8, TB_XJUMP = 9, TB_NONE = 0xFF };

// One traceback byte per DP cell (i>=1, j>=1): bits 0-2 = S move, bit 3 = I came from an extension,
// bit 4 = D came from an extension.  MATCH vs SUBST is re-derived from the bases, the jump source from the
// per-column jump table, so 5 bits carry what the reference keeps in a 16-byte PackedLengthCell
// (align/traceback/packed_length_cell.rs:25-30).
enum : uint8_t { MV_XSUF = 0, MV_INS = 1, MV_DEL = 2, MV_DIAG = 3, MV_JUMP = 4, MV_CIRC = 5, MV_XPRE = 6, MV_YPRE = 7,
                 TBB_IEXT = 8, TBB_DEXT = 16 };

// Alignment operations on the wire (== stitch_op in include/stitch_gpu.h; aligners/constants.rs:20-29).
enum : uint8_t { OP_MATCH = 0, OP_SUBST = 1, OP_DEL = 2, OP_INS = 3, OP_XCLIP = 4, OP_YCLIP = 5, OP_XJUMP = 6, OP_YJUMP = 7 };
struct OpRec { uint8_t kind; uint8_t pad; uint16_t contig; uint32_t arg; };

struct DpParams {              // Scoring (align/scoring.rs:11-23) + Options::clipping (aligners/mod.rs:123-131)
    int32_t match, mismatch, gap_open, gap_extend;
    int32_t jump_same, jump_opp, jump_inter;
    int32_t xclip_prefix, xclip_suffix, yclip_prefix, yclip_suffix;
    int32_t circular;
};

struct JumpInfo { int32_t score; uint32_t len; uint32_t idx; uint32_t from; };   // aligners/mod.rs:56-62

// ------------------------------------------------------------------------------------------------------------
// Row 0 of column j (init_column, single_contig_aligner.rs:188-239).  Identical for every contig of a read, so
// it is evaluated in closed form instead of being stored.  `sn0`/`ly0` are the running Sn[0]/Ly[0].
// ------------------------------------------------------------------------------------------------------------
struct Row0 { int32_t S; uint32_t Slen; uint8_t Smove; uint8_t Dmove; };

STITCH_HD Row0 row0_column0() { Row0 r; r.S = 0; r.Slen = 0; r.Smove = TB_START; r.Dmove = TB_START; return r; }  // :112-118

STITCH_HD Row0 row0_step(const DpParams& P, uint32_t j, uint32_t n, int32_t& sn0, uint32_t& ly0) {
    Row0 r;
    int32_t d;
    if (j == 1) { d = P.gap_open + P.gap_extend; r.Dmove = TB_START; }
    else {
        int32_t d_score = P.gap_open + P.gap_extend * (int32_t)j;
        int32_t c_score = P.yclip_prefix + P.gap_open + P.gap_extend;
        if (d_score > c_score) { d = d_score; r.Dmove = TB_DEL; } else { d = c_score; r.Dmove = TB_YCLIP_PREFIX; }
    }
    if (d > P.yclip_prefix) { r.S = d; r.Smove = TB_DEL; r.Slen = j; }
    else { r.S = P.yclip_prefix; r.Smove = TB_YCLIP_PREFIX; r.Slen = 0; }
    if (j == n && sn0 > r.S) { r.S = sn0; r.Smove = TB_YCLIP_SUFFIX; r.Slen = 0; }
    else if (r.S + P.yclip_suffix > sn0) { sn0 = r.S + P.yclip_suffix; ly0 = n - j; }
    return r;
}
// Initial Sn[0], Ly[0] (init_matrices :125-126)
STITCH_HD void row0_init_sn(const DpParams& P, uint32_t n, int32_t& sn0, uint32_t& ly0) { sn0 = P.yclip_suffix; ly0 = n; }

// Row-0 cell of column j for the walk.  Only column n depends on the running Sn[0] (the `j == n` test above), so
// columns j < n are evaluated directly and column n replays the n-step recurrence once.
STITCH_HD Row0 row0_at(const DpParams& P, uint32_t j, uint32_t n, int32_t* sn0_out = nullptr, uint32_t* ly0_out = nullptr) {
    int32_t sn0; uint32_t ly0; row0_init_sn(P, n, sn0, ly0);
    if (j == 0) { if (sn0_out) *sn0_out = sn0; if (ly0_out) *ly0_out = ly0; return row0_column0(); }
    if (j < n && !sn0_out && !ly0_out) { int32_t t = INT32_MAX; uint32_t u = 0; return row0_step(P, j, n, t, u); }
    Row0 r = row0_column0();
    for (uint32_t jj = 1; jj <= j; ++jj) r = row0_step(P, jj, n, sn0, ly0);
    if (sn0_out) *sn0_out = sn0;
    if (ly0_out) *ly0_out = ly0;
    return r;
}

// ------------------------------------------------------------------------------------------------------------
// Column 0 (init_matrices, :97-186).  Read-independent except Ly[i] = n, so the host evaluates it once per
// contig length when a context is created and every read starts from a copy.
// ------------------------------------------------------------------------------------------------------------
struct Col0Row { int32_t S; uint32_t Slen; int32_t Sn; uint8_t Smove; uint8_t Imove; uint8_t sn_set; };
// Fills rows 1..m (out[i-1]); returns Lx[0].  S[k][m] carries the running x-suffix value exactly as :170-173.
inline uint32_t col0_init(const DpParams& P, uint32_t m, Col0Row* out) {
    uint32_t lx0 = 0;
    int32_t Sm = MIN_SCORE;   // S[k][m] running value
    for (uint32_t i = 1; i <= m; ++i) {
        Col0Row r; r.Smove = TB_START; r.Slen = 0; r.sn_set = 0;
        int32_t I;
        if (i == 1) { I = P.gap_open + P.gap_extend; r.Imove = TB_START; }
        else {
            int32_t i_score = P.gap_open + P.gap_extend * (int32_t)i;
            int32_t c_score = P.xclip_prefix + P.gap_open + P.gap_extend;
            if (i_score > c_score) { I = i_score; r.Imove = TB_INS; } else { I = c_score; r.Imove = TB_XCLIP_PREFIX; }
        }
        int32_t S;
        if (i == m) { r.Smove = TB_XCLIP_SUFFIX; r.Slen = 0; S = Sm; }
        else S = MIN_SCORE;
        if (I > S) { S = I; r.Smove = TB_INS; r.Slen = i; }
        if (P.xclip_prefix > S) { S = P.xclip_prefix; r.Smove = TB_XCLIP_PREFIX; r.Slen = 0; }
        if (i != m && S + P.xclip_suffix > Sm) { Sm = S + P.xclip_suffix; lx0 = m - i; }
        r.S = S;
        r.Sn = MIN_SCORE;
        if (S + P.yclip_suffix > MIN_SCORE) { r.Sn = S + P.yclip_suffix; r.sn_set = 1; }   // Ly[i] = n when set, else 0
        out[i - 1] = r;
    }
    return lx0;
}

// ------------------------------------------------------------------------------------------------------------
// Phase A: everything of cell (i,j) that depends on column j-1 only.
// ------------------------------------------------------------------------------------------------------------
struct ColCtx {               // uniform over a contig's column
    uint32_t j, n, m, cidx;
    uint8_t q;                // y[j-1]
    int32_t xclip_score;      // :304-308
    uint32_t row0_len;        // cell(0,j).S.len  (x-prefix clip length source, :386)
    JumpInfo jump;            // best jump for this contig in this column (multi_contig_aligner.rs:292-331)
    int32_t circ_ok;          // circular && cell(m,j-1).S.move != XCLIP_SUFFIX   (:259-267)
    int32_t circ_score;       // S[prev][m]
    uint32_t circ_len;        // cell(m,j-1).S.len + 1
};

struct RowA {
    int32_t bd; uint32_t dlen; uint32_t dext;          // deletion layer (:328-338)
    int32_t dg;                                        // diagonal candidate (:355)
    int32_t bs2;                                       // best after {diag, deletion}: what the insertion must beat
    int32_t T; uint32_t Tl; uint32_t Tm;               // best/len/move with the insertion candidate left out
    int32_t a;                                         // match_fn.score(x[i-1], y[j-1])
};

// Jump candidate of row i (get_jump_score_and_len, :242-290): the column's best jump plus the zero-cost
// end-to-start jump of a circular contig, which only row 1 may take.
STITCH_HD void row_jump(const ColCtx& cx, uint32_t i, int32_t a, int32_t& J, uint32_t& Jl, uint32_t& Jm) {
    J = cx.jump.score + a; Jl = cx.jump.len; Jm = MV_JUMP;
    if (cx.circ_ok && i == 1) {
        const int32_t z = cx.circ_score + a;
        if (!(J > z) && !(z == J && cx.circ_len <= Jl)) { J = z; Jl = cx.circ_len; Jm = MV_CIRC; }
    }
}

// `col0_len` = cell(i,0).S.len, only dereferenced when the y-prefix clip wins (:395).
// LOCAL = true is the specialisation for AlignmentMode::Local with gap_open + gap_extend < 0: all four clip penalties
// are 0 (aligners/mod.rs:125), so the x-prefix clip candidate is exactly 0 with length 0 (:304-308, row 0 is a
// zero-length y-prefix clip), every S is >= 0, the diagonal always beats the MIN seed (:357) and the y-prefix clip
// candidate go + ge*i < 0 (:391-393) can never win.
template <bool LOCAL = false>
STITCH_HD void row_phase_a(const DpParams& P, const ColCtx& cx, uint32_t i, uint8_t p, int32_t Sp_up, uint32_t Slp_up,
                           int32_t Sp, uint32_t Slp, int32_t Dp, uint32_t Dlp, const uint32_t* col0_len, RowA& r) {
    const int32_t a = (p == cx.q) ? P.match : P.mismatch;
    r.a = a;
    // deletion (:328-338)
    const int32_t de = Dp + P.gap_extend;
    const int32_t dop = Sp + P.gap_open + P.gap_extend;
    r.bd = de > dop ? de : dop;
    r.dext = (de == r.bd);
    r.dlen = (r.dext ? Dlp : Slp) + 1;
    // S without the insertion (:350-399)
    int32_t bs = MIN_SCORE; uint32_t mv = MV_XSUF; uint32_t ln = 0;
    r.dg = Sp_up + a; const uint32_t dl = Slp_up + 1;
    if (LOCAL || r.dg >= bs) { bs = r.dg; mv = MV_DIAG; ln = dl; }
    if (r.bd > bs) { bs = r.bd; mv = MV_DEL; ln = r.dlen; }
    r.bs2 = bs;
    int32_t J; uint32_t Jl, Jm; row_jump(cx, i, a, J, Jl, Jm);      // (:373-382)
    if (J > bs || (J == bs && bs == r.dg && Jl > dl)) { bs = J; mv = Jm; ln = Jl; }
    if (LOCAL) { if (0 > bs) { bs = 0; mv = MV_XPRE; ln = 0; } }
    else {
        if (cx.xclip_score > bs) { bs = cx.xclip_score; mv = MV_XPRE; ln = cx.row0_len; }
        const int32_t yc = P.yclip_prefix + P.gap_open + P.gap_extend * (int32_t)i;      // (:391-393)
        if (yc > bs) { bs = yc; mv = MV_YPRE; ln = col0_len[i - 1]; }
    }
    r.T = bs; r.Tl = ln; r.Tm = mv;
}

// Phase C: merge the insertion (bi, il) into the selection at its place in the priority order (:368-371).
template <bool LOCAL = false>
STITCH_HD void row_phase_c(const DpParams& P, const ColCtx& cx, uint32_t i, const RowA& r, int32_t bi, uint32_t il,
                           const uint32_t* col0_len, int32_t& S, uint32_t& Sl, uint32_t& mv) {
    if (bi > r.bs2) {
        int32_t bs = bi; mv = MV_INS; uint32_t ln = il;
        int32_t J; uint32_t Jl, Jm; row_jump(cx, i, r.a, J, Jl, Jm);
        if (J > bs) { bs = J; mv = Jm; ln = Jl; }             // the == rule needs bs == diag, impossible once bi > bs2 >= dg
        if (LOCAL) { if (0 > bs) { bs = 0; mv = MV_XPRE; ln = 0; } }
        else {
            if (cx.xclip_score > bs) { bs = cx.xclip_score; mv = MV_XPRE; ln = cx.row0_len; }
            const int32_t yc = P.yclip_prefix + P.gap_open + P.gap_extend * (int32_t)i;
            if (yc > bs) { bs = yc; mv = MV_YPRE; ln = col0_len[i - 1]; }
        }
        S = bs; Sl = ln;
    } else { S = r.T; Sl = r.Tl; mv = r.Tm; }
}

// ------------------------------------------------------------------------------------------------------------
// Local-mode kernel (fill_local16.hip and its CPU emulation).
//
// Move codes of the Local-mode kernel's traceback bytes (bits 0-2; IEXT/DEXT as in the generic format).
// ------------------------------------------------------------------------------------------------------------
enum : uint32_t { MK_XSUF = 0, MK_XPRE = 1, MK_JUMP = 2, MK_INS = 3, MK_DEL = 4, MK_DIAG = 5, MK_JUMPL = 6 };

// Row 1 of a circular contig may take the zero-cost jump from row m of the previous column instead of the column's
// best jump (get_jump_score_and_len :258-289).  Both candidates get the same match term added, so the choice does
// not depend on the bases and is made once per (contig, column).
STITCH_HD bool local_row1_circ(const ColCtx& cx) {
    if (!cx.circ_ok) return false;
    if (cx.jump.score > cx.circ_score) return false;
    if (cx.circ_score == cx.jump.score && cx.circ_len <= cx.jump.len) return false;
    return true;
}
// translation of a key-format traceback byte to the generic move codes the walk understands
STITCH_HD uint32_t key_code_to_generic(uint32_t code, bool row1_circ) {
    const uint32_t bits = code & (TBB_IEXT | TBB_DEXT);
    switch (code & 7u) {
        case MK_XSUF: return MV_XSUF | bits;
        case MK_XPRE: return MV_XPRE | bits;
        case MK_INS: return MV_INS | bits;
        case MK_DEL: return MV_DEL | bits;
        case MK_DIAG: return MV_DIAG | bits;
        default: return (row1_circ ? MV_CIRC : MV_JUMP) | bits;      // MK_JUMP, MK_JUMPL
    }
}

// ------------------------------------------------------------------------------------------------------------
// Local-mode selection on combined words (fill_local16.hip's inner loop).
//
// A candidate is ONE signed 32-bit word  w = score << 16 | len  (len < 65535, |score| < 32768): signed comparison
// orders by score, then by alignment length; adding (delta << 16) + 1 advances score and length together; w | 0xFFFF
// is the largest word with w's score, so `x > (w | 0xFFFF)` is "x.score > w.score".  The row state is stored in this
// form, so nothing is unpacked or packed.  The reference's rules in this form (single_contig_aligner.rs:328-399):
//   deletion   de vs open: extension wins ties (:332)           -> (DE | 0xFFFF) >= DO
//   S order    diag, then deletion / insertion / x-prefix clip only if STRICTLY better in score (:357-389)
//   jump       beats the running best if better in score, or — when the running best is still the diagonal — if it
//              ties in score with a longer alignment (:374-377): exactly `JW > DG` as words
// ------------------------------------------------------------------------------------------------------------
struct RowW { int32_t T; uint32_t mvT; int32_t BD; uint32_t dext; int32_t DG; int32_t bs2h; int32_t JW; };

STITCH_HD int32_t word_make(int32_t score, uint32_t len) { return (int32_t)(((uint32_t)score << 16) | (len & 0xFFFFu)); }
STITCH_HD int32_t word_score(int32_t w) { return w >> 16; }
STITCH_HD uint32_t word_len(int32_t w) { return (uint32_t)w & 0xFFFFu; }

// aw = match_fn.score << 16; GE1 = (ge << 16) + 1; GO1 = ((go + ge) << 16) + 1; JSW = word of the column's jump without
// the match term (for row 1 of a circular contig: of the better of it and the end-to-start jump)
STITCH_HD void row_phase_a_word(int32_t aw, int32_t GE1, int32_t GO1, int32_t JSW, int32_t Sup, int32_t Sp, int32_t Dp, RowW& r) {
    r.DG = Sup + aw + 1;
    const int32_t DE = Dp + GE1, DO = Sp + GO1;
    r.dext = (DE | 0xFFFF) >= DO;
    r.BD = r.dext ? DE : DO;
    const int32_t DGh = r.DG | 0xFFFF, BDh = r.BD | 0xFFFF;
    const bool c1 = r.BD > DGh;                       // deletion strictly better than the diagonal
    const int32_t bs2 = c1 ? r.BD : r.DG;
    r.bs2h = c1 ? BDh : DGh;
    r.JW = JSW + aw;
    const int32_t X = c1 ? BDh : r.DG;                // what the jump has to beat
    const bool c3 = r.JW > X;
    int32_t T = c3 ? r.JW : bs2;
    const bool c4 = T < 0;                            // x-prefix clip: score 0, length 0
    r.T = c4 ? 0 : T;
    r.mvT = c4 ? MK_XPRE : c3 ? MK_JUMP : c1 ? MK_DEL : MK_DIAG;
}
// merges the insertion (score bi, length il); returns the final word, `mv` its move
STITCH_HD int32_t row_phase_c_word(const RowW& r, int32_t bi, uint32_t il, uint32_t& mv) {
    const int32_t bic = bi > -16384 ? bi : -16384;    // only chains seeded with MIN get here: keep the word in range
    const int32_t BI = word_make(bic, il);
    const bool c2 = BI > r.bs2h;                      // insertion strictly better than {diag, deletion}
    const bool c5 = r.JW > (BI | 0xFFFF);
    int32_t F1 = c5 ? r.JW : BI;
    const bool c6 = F1 < 0;
    F1 = c6 ? 0 : F1;
    const uint32_t mv1 = c6 ? MK_XPRE : c5 ? MK_JUMP : MK_INS;
    mv = c2 ? mv1 : r.mvT;
    return c2 ? F1 : r.T;
}

// Phase B: the insertion chain as a prefix max.  Row i may open from row i-1 with o_i = S'(i-1)+go+ge, where S' is
// S without its own insertion candidate (an insertion-derived S can never beat the extension it came from because
// go <= 0).  With key_i = o_i - ge*i the chain is I[i] = ge*i + max_{k<=i} key_k; on equal keys the EARLIEST
// opener wins because the reference prefers the extension (`i_score == best_i_score`, :321).  The length follows
// as I.len = q_k + i with q_k = (S'.len(k-1) + 1) - k.  The chain's seed is I[curr][0] = MIN with length 0 (:192).
struct ScanEl { int32_t key; int32_t q; };
STITCH_HD ScanEl scan_seed() { ScanEl e; e.key = MIN_SCORE; e.q = 0; return e; }
STITCH_HD ScanEl scan_make(const DpParams& P, uint32_t i, int32_t Tup, uint32_t Tlup) {
    ScanEl e; e.key = Tup + P.gap_open + P.gap_extend - P.gap_extend * (int32_t)i; e.q = (int32_t)(Tlup + 1) - (int32_t)i; return e;
}
// combine(earlier, later): earlier wins ties
STITCH_HD ScanEl scan_combine(const ScanEl& earlier, const ScanEl& later) { return earlier.key >= later.key ? earlier : later; }

// Suffix-clip tracking record: lexicographic max of (value, len), first row wins (:406-429).
struct XsRec { int32_t v; uint32_t len; uint32_t row; };
STITCH_HD bool xs_better(const XsRec& cand, const XsRec& cur) {   // cand replaces cur?  (cand is a LATER row unless rows say otherwise)
    if (cand.v != cur.v) return cand.v > cur.v;
    if (cand.len != cur.len) return cand.len > cur.len;
    return cand.row < cur.row;
}
// Column arg-max for the next column's jump (get_jump_info :677-697): max value, lowest row.
struct CmRec { int32_t v; uint32_t row; uint32_t len; };
STITCH_HD bool cm_better(const CmRec& cand, const CmRec& cur) { return cand.v > cur.v || (cand.v == cur.v && cand.row < cur.row); }

// Row m: the reference seeds its selection with the running x-suffix value (:350-351).  Given the row's own
// selection (seeded with MIN like every other row) the seeded result is: the running value if it is larger, or
// if it ties and the own winner is not the diagonal (only `diag >= best` accepts a tie, :357).
STITCH_HD bool rowm_run_wins(int32_t run_v, int32_t own_v, int32_t own_dg) {
    return run_v > own_v || (run_v == own_v && own_dg != own_v);
}

// Best jump for destination contig `d` (multi_contig_aligner.rs:292-331).  `base[a]` = get_jump_info of active
// contig a without the jump score (score, len, from), `order` lists the active contig ids in aligner order,
// `opp` = contig id of the same-name opposite strand or -1.
struct JumpBase { int32_t score; uint32_t len; uint32_t from; };
STITCH_HD JumpInfo select_jump(const DpParams& P, const JumpBase* base, const uint32_t* order, uint32_t nact,
                               uint32_t d, int32_t opp) {
    JumpInfo best; best.score = base[d].score + P.jump_same; best.len = base[d].len; best.idx = d; best.from = base[d].from;
    if (opp >= 0) {
        int32_t s = base[opp].score + P.jump_opp;
        if (s > best.score) { best.score = s; best.len = base[opp].len; best.idx = (uint32_t)opp; best.from = base[opp].from; }
    }
    bool have = false; int32_t is = 0; uint32_t il = 0, ia = 0;
    for (uint32_t k = 0; k < nact; ++k) {
        uint32_t a = order[k];
        if (a == d || (int32_t)a == opp) continue;
        int32_t s = base[a].score + P.jump_inter;
        if (!have || s > is || (s == is && base[a].len >= il)) { have = true; is = s; il = base[a].len; ia = a; }   // max_by_key: last max
    }
    if (have && is > best.score) { best.score = is; best.len = il; best.idx = ia; best.from = base[ia].from; }
    return best;
}

}  // namespace stitch
