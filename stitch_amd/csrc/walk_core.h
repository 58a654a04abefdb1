// Last-column fix-ups and traceback walk over the compact traceback written by the fill kernel.
//   fixup_contig  == SingleContigAligner::fill_last_column_and_end_clipping  (single_contig_aligner.rs:453-555)
//   walk_from     == traceback_from                                           (align/traceback/mod.rs:219-373)
//   pick_primary  == traceback                                                (align/traceback/mod.rs:129-150)
// Both are serial per (read, contig) like the reference; on the GPU one lane runs each of them.
#pragma once
#include <type_traits>
#include "dp_core.h"

namespace stitch {

struct ContigDesc {            // one per (contig, strand) aligner
    uint32_t m;                // contig length
    uint32_t roff;             // offset of row 1 in the per-read row arrays (rows 1..m are contiguous; compacted
                               // over the read's active contigs)
    uint32_t troff;            // offset of row 1 in the context-wide column-0 template arrays
    uint32_t seqoff;           // offset of base 0 in the packed sequence array
    uint32_t target;           // index of the FASTA record (contig_idx % T)
    int32_t opp;               // aligner index of the same-name opposite strand, -1 if none (double_strand only)
    uint32_t inv_big, inv_small;   // fill_regs.hip's row -> lane mapping without a division: floor(2^32 / d) + 1 for d = rows of a full / a short lane (tb_row_offset)
};

struct ChainHdr {              // == Alignment (align/alignment.rs:16-51)
    int32_t score; uint32_t xstart, xend, ystart, yend, xlen, ylen;
    uint32_t start_contig_idx, end_contig_idx, length, n_ops, status;   // status: 0 ok, 1 None, 2 overflow, 3 bad move / runaway, 4 reference-undefined XJUMP
    uint32_t join_ops, join_slot;  // traceback_all only: the chain's operations are the first join_ops operations of chain `join_slot` of the same read
                                   // followed by the n_ops written for this chain (join_ops == 0: the n_ops are the whole chain)
};

// traceback_all (--suboptimal) walks from the end of EVERY active contig (align/traceback/mod.rs:152-217).  Nearly all of those walks
// run into the path of the read's best chain after a few steps (a contig the read does not align to ends its chain with a jump out of
// the best path's last columns) and then repeat that path's ten or twenty thousand steps.  A walk is a function of its state
// (contig, row, column, layer move) — plus "the first operation was a y clip", which the x-suffix branch consults and which is required
// to be false on both sides here — so: the walk of the best contig's chain (the REFERENCE walk) records its state where it first enters
// each column, and any other walk that enters a column in the recorded state stops there; its chain is the reference chain's
// operations up to that point (in final order: a prefix) followed by its own.  The host puts the two together (stitch_api.cpp
// download_chains), so what is downloaded and walked shrinks from contigs x columns to columns + contigs x a few.
struct VisitRec {              // one per column (entry j + 1; entry 0 is the reference walk's summary)
    uint16_t contig, row;      // state on entering the column ...   (entry 0: contig = slot of the reference chain, row = 1 if the records are usable)
    uint32_t layer;            // ... move of the layer the walk is in, | 0x100 if the first operation was a y clip
    uint32_t nops;             // operations the reference walk had written by then                (entry 0: all of them)
    uint32_t nonspecial;       // ... of which neither clip nor jump                                 (entry 0: all of them)
};

constexpr uint32_t ERR_CLOCK_OFF = 3072;   // behind JobView::err: {shader cycles, 100 MHz ticks} of the read's column loop (fill_regs.hip)
#ifndef STITCH_REGS_RMAX
#define STITCH_REGS_RMAX 80     // (experiment builds: 40 rows per lane at three waves per SIMD)
#endif
constexpr int REGS_RMAX = STITCH_REGS_RMAX;  // fill_regs.hip: rows a lane holds in registers (a wave owns one contig of up to 64 x REGS_RMAX rows)

struct JobView {
    DpParams P;
    uint32_t n, C, nact, Rtot;
    uint32_t tb_keyfmt;        // 0: generic move codes, rows linear; 1: the Local-mode kernels' key format; 2: key format, lane-interleaved rows
                               // (fill_regs.hip); 3: generic move codes, lane-interleaved rows (fill_regs32.hip)
    uint32_t yrec_global;      // fill_regs.hip: y-suffix records only for cells that reach the best score of ANY contig so far (mode traceback)
    const uint32_t* act;       // active aligner ids in aligner order (the reference's `self.contigs` after sub-setting)
    const int32_t* opp_act;    // [C] opposite-strand aligner id if both strands are active, else -1
    const ContigDesc* cd;      // [C]
    const uint8_t* xseq;       // contig bases (one byte per base)
    const uint8_t* y;          // read bases, upper-cased
    // per-read row state, [Rtot] each; after the fill they hold column n
    int32_t* S; uint32_t* Slen; int32_t* D; uint32_t* Dlen; int32_t* Sn; uint32_t* SnLen; uint32_t* Ly;
    unsigned long long* xchg;  // [2][C][2] inter-workgroup exchange granules of the Local-mode kernel (zeroed before launch)
    uint32_t* err;             // set to 1 if the kernel gave up waiting for a partner workgroup
    uint32_t* st16;            // [2*Rtot] packed row state of the Local-mode kernel: {S | S.len<<16, D | D.len<<16}
    uint8_t* tb;               // [n][Rtot] traceback bytes, column j at (j-1)*Rtot
    uint32_t* Lx;              // [C][n+1]
    uint32_t* jt_idx;          // [C][n+1] source contig of the column's best jump
    uint32_t* jt_from;         // [C][n+1] source row
    // column n: values the fix-ups need or change
    int32_t* Ival; uint32_t* Ilen;                    // [Rtot] written by the fill at j == n
    uint8_t* SmoveF; uint32_t* SidxF; uint32_t* SfromF; uint8_t* ImoveF;   // [Rtot] overrides, TB_NONE = untouched
    // column 0 template (shared by all reads of a context)
    const uint8_t* Smove0; const uint8_t* Imove0; const uint32_t* Slen0;   // indexed by ContigDesc::troff
    // per-contig results of the fix-ups
    int32_t* Sm; uint32_t* Lm;                        // [C] S[n%2][m] and cell(m,n).S.len
    VisitRec* visit;                                  // [n + 2] traceback_all: the reference walk's column records (nullptr: walks do not join)
    uint32_t* Wcol;                                   // [C][n+1] fill_regs.hip, jobs with per-contig y-suffix records: the column's common word (nullptr: none kept)
};

struct WalkArgs { ChainHdr* hdr; OpRec* ops; uint32_t ops_cap; int32_t mode; uint32_t from; uint32_t skip_fixup; };   // per job

constexpr uint32_t JT_CIRC_BIT = 0x80000000u;

struct SCell { uint32_t tb, len, idx, from; };

STITCH_HD uint32_t decode_move(const JobView& V, uint32_t c, uint32_t i, uint32_t j, uint32_t code) {
    switch (code & 7u) {
        case MV_XSUF: return TB_XCLIP_SUFFIX;
        case MV_INS: return TB_INS;
        case MV_DEL: return TB_DEL;
        case MV_XPRE: return TB_XCLIP_PREFIX;
        case MV_YPRE: return TB_YCLIP_PREFIX;
        default: return V.xseq[V.cd[c].seqoff + i - 1] == V.y[j - 1] ? TB_MATCH : TB_SUBST;
    }
}
// (idx, from) the reference stores next to the S move (set_s_all calls in single_contig_aligner.rs:357-399)
STITCH_HD void decode_src(const JobView& V, uint32_t c, uint32_t i, uint32_t j, uint32_t code, uint32_t& idx, uint32_t& from) {
    switch (code & 7u) {
        case MV_XSUF: idx = 0; from = 0; break;              // Cell::default() idx/from survive set_s (:350)
        case MV_INS: case MV_DIAG: idx = c; from = i - 1; break;
        case MV_DEL: case MV_YPRE: idx = c; from = i; break;
        case MV_XPRE: idx = c; from = 0; break;
        case MV_CIRC: idx = c; from = V.cd[c].m; break;
        default: idx = V.jt_idx[(size_t)c * (V.n + 1) + j] & ~JT_CIRC_BIT; from = V.jt_from[(size_t)c * (V.n + 1) + j]; break;   // MV_JUMP
    }
}
// Where the traceback byte of row i (1-based) of a contig of m rows sits within a column's bytes of that contig.  The tiled
// kernels store rows linearly.  fill_regs.hip stores them lane-interleaved: the contig's ceil(m / 4) groups of four rows are
// dealt to the 64 lanes of its wave in order (the first `grem` lanes hold one group more), a lane's rows fill its registers
// from the fullest lane's top register downwards (the lanes with a group less end in register 4, not 0), and register idx of lane l
// goes to byte ((idx >> 2) * 64 + l) * 4 + (idx & 3), so that one store instruction writes whole lines.
STITCH_HD uint32_t tb_div_magic(uint32_t d) { return d ? (uint32_t)(0x100000000ull / d) + 1u : 0u; }      // exact quotients for n * d < 2^32
STITCH_HD uint32_t tb_row_offset(uint32_t keyfmt, const ContigDesc& d, uint32_t i) {
    if (keyfmt < 2) return i - 1;
    const uint32_t ngr = (d.m + 3) / 4, gq = ngr / 64, grem = ngr % 64, row = i - 1;
    const uint32_t big = 4 * (gq + 1), small = 4 * gq;
    uint32_t lane, uu, nrows;
    // (the walk asks for a cell's byte several times per step: a multiplication by the contig's precomputed reciprocal, not a division)
    if (row < grem * big) { lane = (uint32_t)(((unsigned long long)row * d.inv_big) >> 32); uu = row - lane * big; nrows = big; }
    else { const uint32_t rr = row - grem * big; const uint32_t q = (uint32_t)(((unsigned long long)rr * d.inv_small) >> 32); lane = grem + q; uu = rr - q * small; nrows = small; }
    const uint32_t idx = nrows - 1 - uu + ((grem > 0 && lane >= grem) ? 4u : 0u);     // (the lanes that hold a group less have it at the bottom: group 0 is not theirs)
    return ((idx >> 2) * 64 + lane) * 4 + (idx & 3);
}

// Traceback byte of cell (i,j) in the generic move codes.  The Local-mode kernel writes key-format bytes (dp_core.h)
// and flags "row 1 took the circular jump" in bit 31 of the column's jump-table entry.
STITCH_HD uint32_t tb_byte(const JobView& V, uint32_t c, uint32_t i, uint32_t j) {
    const uint32_t raw = V.tb[(size_t)(j - 1) * V.Rtot + V.cd[c].roff + tb_row_offset(V.tb_keyfmt, V.cd[c], i)];
    if (V.tb_keyfmt == 0 || V.tb_keyfmt == 3) return raw;
    return key_code_to_generic(raw, i == 1 && (V.jt_idx[(size_t)c * (V.n + 1) + j] & JT_CIRC_BIT) != 0);
}

// S move of cell (i,j) of contig c as the reference's traceback matrix would hold it after the whole fill.
STITCH_HD uint32_t s_move(const JobView& V, uint32_t c, uint32_t i, uint32_t j) {
    if (j == 0) return i == 0 ? (uint32_t)TB_START : (uint32_t)V.Smove0[V.cd[c].troff + i - 1];
    if (i == 0) return row0_at(V.P, j, V.n).Smove;
    if (j == V.n) { uint8_t f = V.SmoveF[V.cd[c].roff + i - 1]; if (f != TB_NONE) return f; }
    return decode_move(V, c, i, j, tb_byte(V, c, i, j));
}
STITCH_HD void s_src(const JobView& V, uint32_t c, uint32_t i, uint32_t j, uint32_t& idx, uint32_t& from) {
    // only called for cells with i >= 1, j >= 1
    uint32_t r = V.cd[c].roff + i - 1;
    if (j == V.n && V.SmoveF[r] != TB_NONE) { idx = V.SidxF[r]; from = V.SfromF[r]; return; }
    decode_src(V, c, i, j, tb_byte(V, c, i, j), idx, from);
}
STITCH_HD uint32_t i_move(const JobView& V, uint32_t c, uint32_t i, uint32_t j) {   // i >= 1
    uint32_t r = V.cd[c].roff + i - 1;
    if (j == 0) return V.Imove0[V.cd[c].troff + i - 1];
    if (j == V.n) {
        if (V.ImoveF[r] != TB_NONE) return V.ImoveF[r];
        if (tb_byte(V, c, i, j) & TBB_IEXT) return TB_INS;
        // copy taken while column n was filled, i.e. before any fix-up touched cell (i-1,n) (:324-325)
        return i == 1 ? (uint32_t)row0_at(V.P, j, V.n).Smove : decode_move(V, c, i - 1, j, tb_byte(V, c, i - 1, j));
    }
    if (tb_byte(V, c, i, j) & TBB_IEXT) return TB_INS;
    return s_move(V, c, i - 1, j);
}
STITCH_HD uint32_t d_move(const JobView& V, uint32_t c, uint32_t i, uint32_t j) {
    if (j == 0) return TB_START;
    if (i == 0) return row0_at(V.P, j, V.n).Dmove;
    if (tb_byte(V, c, i, j) & TBB_DEXT) return TB_DEL;
    return s_move(V, c, i, j - 1);                     // copy of cell(i,j-1).S.move (:336-337), final since j-1 < n
}

// ------------------------------------------------------------------------------------------------------------
// fill_last_column_and_end_clipping for aligner c (single_contig_aligner.rs:453-555)
// ------------------------------------------------------------------------------------------------------------
STITCH_HD void fixup_contig(const JobView& V, uint32_t c) {
    const DpParams& P = V.P;
    const uint32_t m = V.cd[c].m, roff = V.cd[c].roff, n = V.n;
    int32_t sn0; uint32_t ly0;
    const Row0 r0 = row0_at(P, n, n, &sn0, &ly0);
    int32_t S0 = r0.S;                                  // S[curr][0]
    SCell c0; c0.tb = r0.Smove; c0.len = r0.Slen; c0.idx = c; c0.from = 0;   // cell(0,n).S
    uint32_t* LxN = &V.Lx[(size_t)c * (n + 1) + n];

    // S[curr][.]: row 0 and row m live in registers (row m is written through to memory), the other rows are loaded a block
    // at a time (independent loads) and handed to the row bodies below as `cur`; for i == m `cur` IS the row-m register.
    int32_t Sm_reg = V.S[roff + m - 1];
    auto Sstore = [&](uint32_t i, int32_t v) { if (i >= 1) V.S[roff + i - 1] = v; };
    // (row 0's cell lives in registers; `is0` is a compile-time constant so that no access ever selects between the register copy
    // and memory through a pointer, which would put the copy on the stack: the kernels that run beside persistent teams must not use
    // scratch memory at all, DESIGN.md 4 "Persistent teams")
    auto cell_mem = [&](uint32_t i) -> SCell {
        uint32_t r = roff + i - 1; SCell s; s.len = V.Slen[r];
        if (V.SmoveF[r] != TB_NONE) { s.tb = V.SmoveF[r]; s.idx = V.SidxF[r]; s.from = V.SfromF[r]; }
        else { uint32_t code = tb_byte(V, c, i, n); s.tb = decode_move(V, c, i, n, code); decode_src(V, c, i, n, code, s.idx, s.from); }
        return s;
    };
    auto cell_set_mem = [&](uint32_t i, uint32_t tb, uint32_t len, uint32_t idx, uint32_t from) {
        uint32_t r = roff + i - 1;
        V.SmoveF[r] = (uint8_t)tb; V.Slen[r] = len; V.SidxF[r] = idx; V.SfromF[r] = from;
    };

#if defined(__HIP_DEVICE_COMPILE__)
    constexpr uint32_t BLK = 32;      // rows whose S / Sn (pass 2: S / I) are loaded at once: the scan is a chain of memory round trips, 2 x m / BLK of them
#else
    constexpr uint32_t BLK = 8;
#endif
    // pass 1, one row (:458-517).  `cur` = S[curr][i]
    auto row1 = [&](auto is0, uint32_t i, int32_t& cur, int32_t sn) {
        constexpr bool ROW0 = decltype(is0)::value;
        // (a) jump over the remaining bases of x (:460-466)
        if (cur + P.jump_same > Sm_reg) {
            Sm_reg = cur + P.jump_same; Sstore(m, Sm_reg);
            SCell ps; if constexpr (ROW0) ps = c0; else ps = cell_mem(i);
            cell_set_mem(m, TB_XJUMP, ps.len, ps.idx, i);
        }
        // (b) y suffix clip (:469-491); the equal branch compares a cell's length with itself
        if (sn > cur) {
            cur = sn; Sstore(i, sn);
            uint32_t len;
            if constexpr (ROW0) len = n - ly0 == 0 ? 0u : row0_at(P, n - ly0, n).Slen; else len = V.SnLen[roff + i - 1];
            // idx of this cell is never consumed
            if constexpr (ROW0) { c0.tb = TB_YCLIP_SUFFIX; c0.len = len; c0.idx = c; c0.from = i; } else cell_set_mem(i, TB_YCLIP_SUFFIX, len, c, i);
        }
        // (c) x suffix clip (:494-516)
        {
            int32_t v = cur + P.xclip_suffix;
            bool do_x = false;
            if (v > Sm_reg) do_x = true;
            else if (v == Sm_reg) { uint32_t li; if constexpr (ROW0) li = c0.len; else li = cell_mem(i).len; do_x = li > cell_mem(m).len; }
            if (do_x) {
                Sm_reg = v; Sstore(m, v);
                *LxN = m - i;
                SCell ps; if constexpr (ROW0) ps = c0; else ps = cell_mem(i);
                cell_set_mem(m, TB_XCLIP_SUFFIX, ps.len, ps.idx, i);
            }
        }
    };
    row1(std::true_type(), 0, S0, sn0);
    for (uint32_t b0 = 1; b0 <= m; b0 += BLK) {
        int32_t Sb[BLK], Snb[BLK];
#pragma unroll
        for (uint32_t k = 0; k < BLK; ++k) { const uint32_t i = b0 + k; Sb[k] = i <= m ? V.S[roff + i - 1] : 0; Snb[k] = i <= m ? V.Sn[roff + i - 1] : 0; }
#pragma unroll
        for (uint32_t k = 0; k < BLK; ++k) {
            const uint32_t i = b0 + k;
            if (i < m) row1(std::false_type(), i, Sb[k], Snb[k]); else if (i == m) row1(std::false_type(), i, Sm_reg, Snb[k]);
        }
    }
    // pass 2, one row (:521-554).  `above` = S[curr][i-1] after its own update, `cur` = S[curr][i], `ival` = I[curr][i]
    auto row2 = [&](uint32_t i, int32_t above, int32_t& cur, int32_t ival) {
        uint32_t r = roff + i - 1;
        int32_t i_score = above + P.gap_open + P.gap_extend;
        if (i_score > ival) {
            V.Ival[r] = i_score;
            SCell sv; if (i == 1) sv = c0; else sv = cell_mem(i - 1);
            V.ImoveF[r] = (uint8_t)sv.tb; V.Ilen[r] = sv.len + 1;
        }
        if (i_score > cur) {
            cur = i_score; Sstore(i, i_score);
            uint32_t prev_len = V.Ilen[r];
            cell_set_mem(i, TB_INS, prev_len, c, i - 1);
            if (cur + P.xclip_suffix > Sm_reg) {
                Sm_reg = cur + P.xclip_suffix; Sstore(m, Sm_reg);
                *LxN = m - i;
                cell_set_mem(m, TB_XCLIP_SUFFIX, prev_len, c, i);
            }
        }
    };
    int32_t above = S0;
    for (uint32_t b0 = 1; b0 <= m; b0 += BLK) {
        int32_t Sb[BLK], Ib[BLK];
#pragma unroll
        for (uint32_t k = 0; k < BLK; ++k) { const uint32_t i = b0 + k; Sb[k] = i <= m ? V.S[roff + i - 1] : 0; Ib[k] = i <= m ? V.Ival[roff + i - 1] : 0; }
#pragma unroll
        for (uint32_t k = 0; k < BLK; ++k) {
            const uint32_t i = b0 + k;
            if (i < m) { row2(i, above, Sb[k], Ib[k]); above = Sb[k]; } else if (i == m) { row2(i, above, Sm_reg, Ib[k]); above = Sm_reg; }
        }
    }
    V.Sm[c] = Sm_reg;
    V.Lm[c] = cell_mem(m).len;
}

// traceback (align/traceback/mod.rs:129-150): best end contig among the active aligners, in aligner order
STITCH_HD uint32_t pick_primary(const JobView& V) {
    uint32_t best = V.act[0]; int32_t score = MIN_SCORE; uint32_t alen = 0;
    for (uint32_t k = 0; k < V.nact; ++k) {
        uint32_t c = V.act[k];
        int32_t s = V.Sm[c]; uint32_t l = V.Lm[c];
        if (s > score || (s == score && l > alen)) { best = c; score = s; alen = l; }
    }
    return best;
}

STITCH_HD bool is_active(const JobView& V, uint32_t c) {
    for (uint32_t k = 0; k < V.nact; ++k) if (V.act[k] == c) return true;
    return false;
}

// How a walk is executed.  SoloWalk: one thread, literal.  On the GPU a whole wavefront can run ONE walk with identical
// state in every lane (fill_kernel.hip, WaveWalk): lane 0 does the writes, and on a run of plain diagonal steps the 64 lanes
// fetch 64 cells of the diagonal at once, so the walk pays one memory latency per run instead of one per cell.
struct SoloWalk {
    STITCH_HD bool writer() const { return true; }
    STITCH_HD uint32_t diag_run(const JobView&, uint32_t, uint32_t, uint32_t, uint32_t, OpRec*, uint32_t, uint32_t, bool) const { return 0; }
    STITCH_HD bool enter_column(const JobView&, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, bool) const { return false; }
    STITCH_HD bool joined() const { return false; }
    STITCH_HD uint32_t join_nops() const { return 0; }
    STITCH_HD uint32_t join_nonspecial() const { return 0; }
    STITCH_HD void finish_reference(const JobView&, uint32_t, uint32_t, bool) const {}
    STITCH_HD ChainHdr reference_header() const { return ChainHdr{}; }
    STITCH_HD void reverse(OpRec* ops, uint32_t nops) const { for (uint32_t a = 0, b = nops; a + 1 < b; ++a, --b) { OpRec t = ops[a]; ops[a] = ops[b - 1]; ops[b - 1] = t; } }
};

// traceback_from (align/traceback/mod.rs:219-373).  Ops are written in reverse and flipped at the end.
template <typename Exec>
STITCH_HD void walk_from_t(const JobView& V, uint32_t contig_index, ChainHdr& H, OpRec* ops, uint32_t ops_cap, const Exec& ex) {
    const uint32_t n = V.n;
    H.status = 0; H.n_ops = 0; H.end_contig_idx = contig_index; H.join_ops = 0; H.join_slot = 0;
    if (contig_index >= V.C || !is_active(V, contig_index)) { H.status = 1; return; }
    uint32_t j = n, nops = 0;
    uint32_t xstart = 0, ystart = 0, yend = n;
    uint32_t cur = contig_index;
    uint32_t i = V.cd[cur].m, xend = V.cd[cur].m;
    const uint32_t xlen = V.cd[cur].m;
    const int32_t score = V.Sm[cur];
    const uint32_t alignment_length = V.Lm[cur];
    uint32_t first_kind = 0xFF;                          // kind of operations[0] (the first op pushed)
    uint32_t last_layer = s_move(V, cur, i, j);
    uint64_t max_steps = 2ull * ((uint64_t)n + 2) * ((uint64_t)V.Rtot + 2) + 64;   // free gaps + free jumps can emit ~n*m ops
    if (max_steps > 50000000ull) max_steps = 50000000ull;                           // a runaway walk must end in seconds, not hours
    uint64_t steps = 0;
    uint32_t nonspecial = 0;                             // operations written so far that are neither clip nor jump
    uint32_t j_entered = n;                              // the column whose entry the execution policy has seen (the start is not an entry)
    auto push = [&](uint8_t kind, uint32_t contig, uint32_t arg) {
        if (nops < ops_cap) { if (ex.writer()) { OpRec o; o.kind = kind; o.pad = 0; o.contig = (uint16_t)contig; o.arg = arg; ops[nops] = o; } }
        else H.status = 2;
        if (nops == 0) first_kind = kind;
        if (!(kind == OP_XCLIP || kind == OP_YCLIP || kind == OP_XJUMP)) ++nonspecial;
        ++nops;
    };
    for (;;) {
        if (++steps > max_steps || i > V.cd[cur].m || j > n) { H.status = 3; break; }     // (a walk that leaves the matrix ends with an error, not a fault)
        // a new column: the reference walk of traceback_all records the state, the others stop where they meet it (VisitRec above)
        if (j != j_entered) { j_entered = j; if (ex.enter_column(V, cur, i, j, last_layer, nops, nonspecial, first_kind == OP_YCLIP)) break; }
        // (cur is checked against the active set where it changes: at the start and after every jump)
        uint32_t next_layer;
        if (last_layer == TB_START) break;
        if (last_layer == TB_MATCH || last_layer == TB_SUBST) {
            // a run of L cells (i,j), (i-1,j-1), ... that are all plain diagonal steps inside the matrix (row >= 2, column < n):
            // exactly what L turns of the branch below would do, without a jump in between
            const uint32_t L = ex.diag_run(V, cur, i, j, nops, ops, ops_cap, nonspecial, first_kind == OP_YCLIP);
            if (L) {
                if (nops == 0) first_kind = last_layer == TB_MATCH ? OP_MATCH : OP_SUBST;
                if (nops + L > ops_cap) H.status = 2;
                nops += L; nonspecial += L; steps += L; i -= L; j -= L;
                if (ex.joined()) { j_entered = j; break; }      // (one of the run's cells was the reference walk's: the run was cut there)
                last_layer = s_move(V, cur, i, j);
                continue;
            }
        }
        if (last_layer == TB_INS) {
            push(OP_INS, 0, 0);
            next_layer = i_move(V, cur, i, j);
            i -= 1;
        } else if (last_layer == TB_DEL) {
            push(OP_DEL, 0, 0);
            next_layer = d_move(V, cur, i, j);
            j -= 1;
        } else if (last_layer == TB_MATCH || last_layer == TB_SUBST) {
            push(last_layer == TB_MATCH ? OP_MATCH : OP_SUBST, 0, 0);
            uint32_t idx, from; s_src(V, cur, i, j, idx, from);
            if (idx != cur || from != i - 1) {
                push(OP_XJUMP, cur, i - 1);
                cur = idx;
                if (cur >= V.C || !is_active(V, cur)) { H.status = 1; return; }
            }
            i = from; j -= 1;
            next_layer = s_move(V, cur, i, j);
        } else if (last_layer == TB_XCLIP_PREFIX) {
            next_layer = s_move(V, cur, 0, j);
            if (next_layer == TB_START || next_layer == TB_YCLIP_PREFIX) { push(OP_XCLIP, 0, i); xstart = i; }
            i = 0;
        } else if (last_layer == TB_XCLIP_SUFFIX) {
            uint32_t lx = V.Lx[(size_t)cur * (n + 1) + j];
            if (nops == 0 || first_kind == OP_YCLIP) { push(OP_XCLIP, 0, lx); xend = i - lx; }
            i -= lx;
            next_layer = s_move(V, cur, i, j);
        } else if (last_layer == TB_YCLIP_PREFIX) {
            push(OP_YCLIP, 0, j);
            ystart = j; j = 0;
            next_layer = s_move(V, cur, i, 0);
        } else if (last_layer == TB_YCLIP_SUFFIX) {
            // only reachable in column n: row 0 keeps Ly[0] in closed form, other rows in the Ly array
            uint32_t ly;
            if (i == 0) { int32_t sn0; row0_at(V.P, n, n, &sn0, &ly); } else ly = V.Ly[V.cd[cur].roff + i - 1];
            push(OP_YCLIP, 0, ly);
            uint32_t from = i == 0 ? 0u : V.SfromF[V.cd[cur].roff + i - 1];
            j -= ly;
            if (from != i) { push(OP_XJUMP, cur, i); i = from; }
            yend = j;
            next_layer = s_move(V, cur, i, j);
        } else if (last_layer == TB_XJUMP) {
            uint32_t r = V.cd[cur].roff + i - 1;
            uint32_t idx = V.SidxF[r], from = V.SfromF[r];
            push(OP_XJUMP, cur, i);
            cur = idx;
            if (cur >= V.C || !is_active(V, cur)) { H.status = 1; return; }
            i = from;
            // the source row belongs to the contig the jump came FROM; the reference indexes the other contig's
            // matrix with it (debug_assert in traceback/mod.rs:111-112) — undefined when that contig is shorter
            if (i > V.cd[cur].m) { H.status = 4; return; }
            next_layer = s_move(V, cur, i, j);
        } else { H.status = 3; break; }
        last_layer = next_layer;
    }
    H.join_ops = 0; H.join_slot = 0;
    if (H.status == 2) { H.n_ops = nops; return; }
    if (H.status == 3) return;
    ex.reverse(ops, nops);
    bool all_special = nonspecial == 0;
    if (ex.joined()) {
        // the rest of this walk is the reference walk's from the column where they met: its operations from there on (in final order the
        // first join_ops of that chain), its start coordinates (set by prefix clips at the very end of a walk), its start contig
        const VisitRec sum = V.visit[0];
        const ChainHdr R = ex.reference_header();
        all_special = all_special && sum.nonspecial == ex.join_nonspecial();
        xstart = R.xstart; ystart = R.ystart; cur = R.start_contig_idx;
        H.join_ops = sum.nops - ex.join_nops(); H.join_slot = sum.contig;
    }
    ex.finish_reference(V, nops, nonspecial, H.status == 0 && !all_special);
    if (all_special) { xstart = 0; xend = 0; ystart = 0; yend = 0; }
    H.score = score; H.xstart = xstart; H.xend = xend; H.ystart = ystart; H.yend = yend; H.xlen = xlen; H.ylen = n;
    H.start_contig_idx = cur; H.end_contig_idx = contig_index; H.length = alignment_length; H.n_ops = nops;
}
STITCH_HD void walk_from(const JobView& V, uint32_t contig_index, ChainHdr& H, OpRec* ops, uint32_t ops_cap) { walk_from_t(V, contig_index, H, ops, ops_cap, SoloWalk()); }

// What a walk of traceback_all that may join the read's reference chain keeps and decides — the same on the device (WaveWalk below: a
// wavefront per walk, lane 0 writes) and in the CPU emulation of the tests (tests/emu: one thread).
// role 0: a plain walk.  role 1: the reference walk of traceback_all, which records its state on entering every column (VisitRec).
// role 2: another walk of the same read, which stops where it enters a column in the recorded state.
struct JoinRole {
    int role = 0; uint32_t ref_slot = 0; const ChainHdr* ref_hdr = nullptr;
    mutable bool met = false; mutable uint32_t met_nops = 0, met_nonspecial = 0;
    STITCH_HD bool joined() const { return met; }
    STITCH_HD uint32_t join_nops() const { return met_nops; }
    STITCH_HD uint32_t join_nonspecial() const { return met_nonspecial; }
    STITCH_HD ChainHdr reference_header() const { return *ref_hdr; }
    STITCH_HD static uint32_t layer_word(uint32_t layer, bool yfirst) { return layer | (yfirst ? 0x100u : 0u); }
    STITCH_HD bool enter_column_as(bool writes, const JobView& V, uint32_t cur, uint32_t i, uint32_t j, uint32_t layer, uint32_t nops, uint32_t nonspecial, bool yfirst) const {
        if (role == 1) { if (writes) { VisitRec r; r.contig = (uint16_t)cur; r.row = (uint16_t)i; r.layer = layer_word(layer, yfirst); r.nops = nops; r.nonspecial = nonspecial; V.visit[j + 1] = r; } return false; }
        // (never in column 0: a walk gets there through its prefix clips, which have set ITS start coordinates already — two chains that
        // start in the same contig at different cells both arrive at (contig, row 0, column 0, start), and the one is not the other's prefix)
        if (role != 2 || yfirst || j == 0) return false;
        const VisitRec r = V.visit[j + 1];
        if (r.contig == (uint16_t)cur && r.row == (uint16_t)i && r.layer == layer_word(layer, false)) { met = true; met_nops = r.nops; met_nonspecial = r.nonspecial; return true; }
        return false;
    }
    STITCH_HD void finish_reference_as(bool writes, const JobView& V, uint32_t nops, uint32_t nonspecial, bool usable) const {
        if (role != 1 || !writes) return;
        VisitRec r; r.contig = (uint16_t)ref_slot; r.row = usable ? 1 : 0; r.layer = 0; r.nops = nops; r.nonspecial = nonspecial; V.visit[0] = r;
    }
    // the records as the reference walk finds them: nothing matches, entry 0 names the reference chain's slot and says "not usable" yet
    STITCH_HD static VisitRec cleared(uint32_t entry, uint32_t ref_slot_) { VisitRec z; z.contig = entry == 0 ? (uint16_t)ref_slot_ : (uint16_t)0xFFFFu; z.row = 0; z.layer = 0xFFFFFFFFu; z.nops = 0; z.nonspecial = 0; return z; }
};

#if defined(__HIPCC__)
// One wavefront, one walk: every lane carries the same state (walk_core.h, walk_from_t), lane 0 writes.
struct WaveWalk : JoinRole {
    int lane;
    __device__ bool writer() const { return lane == 0; }
    __device__ bool enter_column(const JobView& V, uint32_t cur, uint32_t i, uint32_t j, uint32_t layer, uint32_t nops, uint32_t nonspecial, bool yfirst) const {
        return enter_column_as(lane == 0, V, cur, i, j, layer, nops, nonspecial, yfirst);
    }
    __device__ void finish_reference(const JobView& V, uint32_t nops, uint32_t nonspecial, bool usable) const { finish_reference_as(lane == 0, V, nops, nonspecial, usable); }
    // Cells (i-l, j-l), l = 0..63, fetched by lane l.  Returns the number L of leading cells that are plain diagonal steps
    // (traceback code MV_DIAG: source = the cell up-left in the same contig) with row >= 2 and 1 <= column < n (row 1 can
    // hold the circular jump, column n the fix-up overrides: both stay on the literal path); their operations are written
    // by the lanes themselves.  Cell l >= 1 is where the walk enters column j - l: the reference walk records it there, another
    // walk compares, and the run is cut at the first cell that is the reference walk's (`met`).
    __device__ uint32_t diag_run(const JobView& V, uint32_t cur, uint32_t i, uint32_t j, uint32_t nops, OpRec* ops, uint32_t ops_cap, uint32_t nonspecial, bool yfirst) const {
        const uint32_t l = (uint32_t)lane;
        const bool inb = l + 2 <= i && l + 1 <= j && j - l < V.n;
        bool ok = false, match = false;
        if (inb) {
            const ContigDesc d = V.cd[cur];
            const uint32_t ii = i - l, jj = j - l;
            const uint32_t raw = V.tb[(size_t)(jj - 1) * V.Rtot + d.roff + tb_row_offset(V.tb_keyfmt, d, ii)];
            const uint32_t code = (V.tb_keyfmt == 1 || V.tb_keyfmt == 2) ? key_code_to_generic(raw, false) : raw;
            ok = (code & 7u) == MV_DIAG;
            match = V.xseq[d.seqoff + ii - 1] == V.y[jj - 1];
        }
        const unsigned long long bad = ~__ballot(ok);
        uint32_t L = bad ? (uint32_t)__builtin_ctzll(bad) : 64u;
        if (role == 2 && !yfirst && L > 1) {
            // lane l (1 <= l < L) holds the cell through which the walk enters column j - l, in the S layer, with l operations of this run behind it
            bool same = false; VisitRec r{};
            if (l >= 1 && l < L) { r = V.visit[j - l + 1]; same = r.contig == (uint16_t)cur && r.row == (uint16_t)(i - l) && r.layer == layer_word(match ? TB_MATCH : TB_SUBST, false); }
            const unsigned long long hit = __ballot(same);
            if (hit) {
                const int at = (int)__builtin_ctzll(hit);
                met = true; met_nops = (uint32_t)__builtin_amdgcn_readlane((int)r.nops, at); met_nonspecial = (uint32_t)__builtin_amdgcn_readlane((int)r.nonspecial, at);
                L = (uint32_t)at;                                   // the cells before it are this walk's own
            }
        }
        if (role == 1 && l >= 1 && l < L) { VisitRec r; r.contig = (uint16_t)cur; r.row = (uint16_t)(i - l); r.layer = layer_word(match ? TB_MATCH : TB_SUBST, yfirst); r.nops = nops + l; r.nonspecial = nonspecial + l; V.visit[j - l + 1] = r; }
        if (l < L && nops + l < ops_cap) { OpRec o; o.kind = match ? OP_MATCH : OP_SUBST; o.pad = 0; o.contig = 0; o.arg = 0; ops[nops + l] = o; }
        return L;
    }
    __device__ void reverse(OpRec* ops, uint32_t nops) const {
        for (uint32_t a = (uint32_t)lane; a < nops / 2; a += 64) { const OpRec t = ops[a]; ops[a] = ops[nops - 1 - a]; ops[nops - 1 - a] = t; }
    }
};
#endif

}  // namespace stitch
