// ORACLE — TEST INFRASTRUCTURE ONLY.  See stitch_oracle.hpp for scope, parity status and rules.
// Line-by-line CPU restatement of fulcrumgenomics/stitch's jump-aware affine-gap aligner.
// Citations are relative to /root/reference/fg-stitch-lib/src/.
#include "stitch_oracle.hpp"

#include <algorithm>
#include <cassert>
#include <cstring>
#include <stdexcept>

namespace orc {

// ----------------------------------------------------------------------------------------------
// PackedLengthCell — align/traceback/packed_length_cell.rs:32-182
// ----------------------------------------------------------------------------------------------
static constexpr uint32_t ALIGN_LEN_BIT_POS = 4;
static constexpr uint32_t ALIGN_LEN_BIT_MASK = 0x7FFFFFFu;   // 27 bits
static constexpr uint32_t CONTIG_IDX_POS = 31;
static constexpr uint32_t AUX_CONTIG_IDX_MASK = 0x1Fu;
static constexpr uint32_t AUX_CONTIG_FROM_POS = 5;

uint32_t Cell::set_tb(uint32_t matrix, uint16_t tb) {                     // :43-50
    if (tb > TB_MAX) throw std::runtime_error("Expected a value <= TB_MAX while setting traceback bits");
    return (matrix & ~0xFu) | (uint32_t)tb;
}
uint32_t Cell::set_len(uint32_t matrix, uint32_t len) {                   // :53-56
    const uint32_t bits = ALIGN_LEN_BIT_MASK << ALIGN_LEN_BIT_POS;
    return (matrix & ~bits) | (len << ALIGN_LEN_BIT_POS);
}
void Cell::set_idx(uint32_t idx) {                                        // :59-68
    s = (s & ~(1u << CONTIG_IDX_POS)) | (((idx >> 7) & 1u) << CONTIG_IDX_POS);
    i = (i & ~(1u << CONTIG_IDX_POS)) | (((idx >> 6) & 1u) << CONTIG_IDX_POS);
    d = (d & ~(1u << CONTIG_IDX_POS)) | (((idx >> 5) & 1u) << CONTIG_IDX_POS);
    aux = (aux & ~AUX_CONTIG_IDX_MASK) | (idx & AUX_CONTIG_IDX_MASK);
}
void Cell::set_from(uint32_t from) { aux = (aux & AUX_CONTIG_IDX_MASK) | (from << AUX_CONTIG_FROM_POS); }  // :71-74
void Cell::set_i(uint16_t tb, uint32_t len) { i = set_tb(i, tb); i = set_len(i, len); }   // :117-121
void Cell::set_d(uint16_t tb, uint32_t len) { d = set_tb(d, tb); d = set_len(d, len); }   // :124-128
void Cell::set_s(uint16_t tb, uint32_t len) { s = set_tb(s, tb); s = set_len(s, len); }   // :131-135
void Cell::set_s_all(uint16_t tb, uint32_t len, uint32_t idx, uint32_t from) {             // :138-146
    if (idx > 255) throw std::runtime_error("idx <= max_num_contigs");
    if (from > 134217727u) throw std::runtime_error("from <= max_target_len");
    s = set_tb(s, tb); s = set_len(s, len); set_idx(idx); set_from(from);
}
uint32_t Cell::get_idx() const {                                          // :88-99
    uint32_t value = 0;
    value |= (s >> 31) << 7; value |= (i >> 31) << 6; value |= (d >> 31) << 5;
    value |= aux & AUX_CONTIG_IDX_MASK;
    return value;
}

// align/traceback/mod.rs:93-100 — every cell <- (START,0) x3, idx 0, from 0
void Traceback::init(size_t m, size_t n) {
    matrix.clear();
    Cell start; start.set_all(TB_START, 0); start.set_s_all(TB_START, 0, 0, 0);
    rows = m + 1; cols = n + 1;
    matrix.resize(rows * cols, start);
}

// ----------------------------------------------------------------------------------------------
// AlignmentOperation helpers — align/aligners/constants.rs:31-85
// ----------------------------------------------------------------------------------------------
std::string Op::as_string(size_t contig_idx, size_t x_index) const {      // :37-59
    switch (kind) {
        case Match: return "=";
        case Subst: return "X";
        case Del: return "D";
        case Ins: return "I";
        case Xclip: return std::to_string(a) + "A";
        case Yclip: return std::to_string(a) + "B";
        case Xjump: {
            std::string cj;
            if (a > contig_idx) cj = std::to_string(a - contig_idx) + "C";
            else if (a < contig_idx) cj = std::to_string(contig_idx - a) + "c";
            if (b >= x_index) return cj + std::to_string(b - x_index) + "J";
            return cj + std::to_string(x_index - b) + "j";
        }
        case Yjump: return std::to_string(a) + "S";
    }
    return "";
}
int32_t Op::length_on_x(size_t x_index) const {                           // :61-71
    switch (kind) {
        case Match: case Subst: case Ins: return 1;
        case Del: case Yclip: case Yjump: return 0;
        case Xclip: return (int32_t)a;
        case Xjump: return (int32_t)b - (int32_t)x_index;
    }
    return 0;
}
size_t Op::length_on_y() const {                                          // :74-84
    switch (kind) {
        case Match: case Subst: case Del: return 1;
        case Yclip: return a;
        case Yjump: return a;
        default: return 0;
    }
}

// ----------------------------------------------------------------------------------------------
// SingleContigAligner — align/aligners/single_contig_aligner.rs
// ----------------------------------------------------------------------------------------------
void SingleContigAligner::init_matrices(size_t m, size_t n) {             // :97-186
    traceback.init(m, n);
    for (int k = 0; k < 2; ++k) {
        I[k].assign(m + 1, MIN_SCORE);
        D[k].assign(m + 1, MIN_SCORE);
        S[k].assign(m + 1, MIN_SCORE);
        S[k][0] = 0;
        if (k == 0) {
            Cell tb; tb.set_all(TB_START, 0); tb.set_s_all(TB_START, 0, contig_idx, 0);
            traceback.set(0, 0, tb);
            Lx.assign(n + 1, 0);
            Ly.assign(m + 1, 0);
            Sn.assign(m + 1, MIN_SCORE);
            Sn[0] = scoring.yclip_suffix;
            Ly[0] = n;
        }
        for (size_t i = 1; i <= m; ++i) {
            Cell tb; tb.set_all(TB_START, 0); tb.set_s_all(TB_START, 0, contig_idx, 0);
            if (i == 1) {
                I[k][i] = scoring.gap_open + scoring.gap_extend;
                tb.set_i(TB_START, 1);
            } else {
                int32_t i_score = scoring.gap_open + scoring.gap_extend * (int32_t)i;
                int32_t c_score = scoring.xclip_prefix + scoring.gap_open + scoring.gap_extend;
                if (i_score > c_score) { I[k][i] = i_score; tb.set_i(TB_INS, (uint32_t)i); }
                else { I[k][i] = c_score; tb.set_i(TB_XCLIP_PREFIX, 0); }
            }
            if (i == m) tb.set_s(TB_XCLIP_SUFFIX, 0);
            else S[k][i] = MIN_SCORE;
            if (I[k][i] > S[k][i]) { S[k][i] = I[k][i]; tb.set_s(TB_INS, (uint32_t)i); }
            if (scoring.xclip_prefix > S[k][i]) { S[k][i] = scoring.xclip_prefix; tb.set_s(TB_XCLIP_PREFIX, 0); }
            if (i != m && S[k][i] + scoring.xclip_suffix > S[k][m]) {
                S[k][m] = S[k][i] + scoring.xclip_suffix;
                Lx[0] = m - i;
            }
            if (k == 0) traceback.set(i, 0, tb);
            if (S[k][i] + scoring.yclip_suffix > Sn[i]) {
                Sn[i] = S[k][i] + scoring.yclip_suffix;
                Ly[i] = n;
            }
        }
    }
}

void SingleContigAligner::init_column(size_t j, size_t curr, size_t m, size_t n) {   // :188-239
    Cell tb; tb.set_s_all(TB_START, 0, contig_idx, 0);
    I[curr][0] = MIN_SCORE;
    if (j == 1) {
        D[curr][0] = scoring.gap_open + scoring.gap_extend;
        tb.set_d(TB_START, 1);
    } else {
        int32_t d_score = scoring.gap_open + scoring.gap_extend * (int32_t)j;
        int32_t c_score = scoring.yclip_prefix + scoring.gap_open + scoring.gap_extend;
        if (d_score > c_score) { D[curr][0] = d_score; tb.set_d(TB_DEL, (uint32_t)j); }
        else { D[curr][0] = c_score; tb.set_d(TB_YCLIP_PREFIX, 0); }
    }
    if (D[curr][0] > scoring.yclip_prefix) { S[curr][0] = D[curr][0]; tb.set_s(TB_DEL, (uint32_t)j); }
    else { S[curr][0] = scoring.yclip_prefix; tb.set_s(TB_YCLIP_PREFIX, 0); }
    if (j == n && Sn[0] > S[curr][0]) {
        S[curr][0] = Sn[0];
        tb.set_s(TB_YCLIP_SUFFIX, 0);
    } else if (S[curr][0] + scoring.yclip_suffix > Sn[0]) {
        Sn[0] = S[curr][0] + scoring.yclip_suffix;
        Ly[0] = n - j;
    }
    traceback.set(0, j, tb);
    for (size_t i = 1; i <= m; ++i) S[curr][i] = MIN_SCORE;
}

JumpInfo SingleContigAligner::get_jump_score_and_len(size_t m, size_t i, size_t j, size_t prev, int32_t addend,
                                                     JumpInfo jump_info) const {   // :242-290
    jump_info.score += addend;
    if (!circular || i != 1) return jump_info;
    uint16_t jump_from_end_tb = traceback.get(m, j - 1).get_s().tb;
    if (jump_from_end_tb == TB_XCLIP_SUFFIX) return jump_info;
    int32_t jump_from_end_score = S[prev][m] + addend;
    if (jump_info.score > jump_from_end_score) return jump_info;
    SValue jump_from_end_s = traceback.get(m, j - 1).get_s();
    uint32_t jump_from_end_len = jump_from_end_s.len + 1;
    if (jump_from_end_score == jump_info.score && jump_from_end_len <= jump_info.len) return jump_info;
    JumpInfo r; r.score = jump_from_end_score; r.len = jump_from_end_len; r.idx = contig_idx; r.from = (uint32_t)m;
    return r;
}

void SingleContigAligner::fill_column(const uint8_t* x, const uint8_t* y, size_t m, size_t n, size_t j,
                                      size_t prev, size_t curr, JumpInfo jump_info) {   // :292-451
    const uint8_t q = y[j - 1];
    const int32_t xclip_score = scoring.xclip_prefix +
        std::max(scoring.yclip_prefix, scoring.gap_open + scoring.gap_extend * (int32_t)j);
    for (size_t i = 1; i <= m; ++i) {
        const uint8_t p = x[i - 1];
        Cell tb;
        // Insertion (:314-326)
        int32_t i_score = I[curr][i - 1] + scoring.gap_extend;
        int32_t s_score = S[curr][i - 1] + scoring.gap_open + scoring.gap_extend;
        int32_t best_i_score = std::max(i_score, s_score);
        if (i_score == best_i_score) {
            tb.set_i(TB_INS, traceback.get(i - 1, j).get_i_len() + 1);
        } else {
            SValue s_value = traceback.get(i - 1, j).get_s();
            tb.set_i(s_value.tb, s_value.len + 1);
        }
        // Deletion (:328-338)
        int32_t d_score = D[prev][i] + scoring.gap_extend;
        s_score = S[prev][i] + scoring.gap_open + scoring.gap_extend;
        int32_t best_d_score = std::max(d_score, s_score);
        if (d_score == best_d_score) {
            uint32_t prev_len = traceback.get(i, j - 1).get_d_len();
            tb.set_d(TB_DEL, prev_len + 1);
        } else {
            SValue s_value = traceback.get(i, j - 1).get_s();
            tb.set_d(s_value.tb, s_value.len + 1);
        }
        // S (:350-399)
        tb.set_s(TB_XCLIP_SUFFIX, traceback.get(i, j).get_s_len());
        int32_t best_s_score = S[curr][i];
        int32_t addend = scoring.score(p, q);
        int32_t diag_score = S[prev][i - 1] + addend;
        uint32_t diag_len = traceback.get(i - 1, j - 1).get_s_len() + 1;
        if (diag_score >= best_s_score) {
            best_s_score = diag_score;
            tb.set_s_all(p == q ? TB_MATCH : TB_SUBST, diag_len, contig_idx, (uint32_t)(i - 1));
        }
        if (best_d_score > best_s_score) {
            best_s_score = best_d_score;
            tb.set_s_all(TB_DEL, tb.get_d_len(), contig_idx, (uint32_t)i);
        }
        if (best_i_score > best_s_score) {
            best_s_score = best_i_score;
            tb.set_s_all(TB_INS, tb.get_i_len(), contig_idx, (uint32_t)(i - 1));
        }
        JumpInfo x_jump_info = get_jump_score_and_len(m, i, j, prev, addend, jump_info);
        bool do_jump = x_jump_info.score > best_s_score ||
            (x_jump_info.score == best_s_score && best_s_score == diag_score && x_jump_info.len > diag_len);
        if (do_jump) {
            best_s_score = x_jump_info.score;
            tb.set_s_all(p == q ? TB_MATCH : TB_SUBST, x_jump_info.len, x_jump_info.idx, x_jump_info.from);
        }
        if (xclip_score > best_s_score) {
            best_s_score = xclip_score;
            uint32_t prev_len = traceback.get(0, j).get_s_len();
            tb.set_s_all(TB_XCLIP_PREFIX, prev_len, contig_idx, 0);
        }
        int32_t yclip_score = scoring.yclip_prefix + scoring.gap_open + scoring.gap_extend * (int32_t)i;
        if (yclip_score > best_s_score) {
            uint32_t prev_len = traceback.get(i, 0).get_s_len();
            best_s_score = yclip_score;
            tb.set_s_all(TB_YCLIP_PREFIX, prev_len, contig_idx, (uint32_t)i);
        }
        S[curr][i] = best_s_score;
        I[curr][i] = best_i_score;
        D[curr][i] = best_d_score;
        // x suffix clip tracking (:406-429)
        {
            int32_t v = S[curr][i] + scoring.xclip_suffix;
            bool do_x = false;
            if (v > S[curr][m]) do_x = true;
            else if (v == S[curr][m]) do_x = tb.get_s_len() > traceback.get(m, j).get_s_len();
            if (do_x) {
                S[curr][m] = v;
                SValue prev_s = tb.get_s();
                traceback.get_mut(m, j).set_s_all(TB_XCLIP_SUFFIX, prev_s.len, prev_s.idx, (uint32_t)i);
                Lx[j] = m - i;
            }
        }
        // y suffix clip tracking (:431-447)
        {
            int32_t v = S[curr][i] + scoring.yclip_suffix;
            bool do_y = false;
            if (v > Sn[i]) do_y = true;
            else if (v == Sn[i]) do_y = tb.get_s_len() > traceback.get(i, n).get_s_len();
            if (do_y) { Sn[i] = v; Ly[i] = n - j; }
        }
        traceback.set(i, j, tb);
    }
}

void SingleContigAligner::fill_last_column_and_end_clipping(size_t m, size_t n) {   // :453-555
    for (size_t i = 0; i <= m; ++i) {
        const size_t j = n;
        const size_t curr = j % 2;
        if (S[curr][i] + scoring.jump_score_same_contig_and_strand > S[curr][m]) {
            S[curr][m] = S[curr][i] + scoring.jump_score_same_contig_and_strand;
            SValue prev_s = traceback.get(i, j).get_s();
            traceback.get_mut(m, j).set_s_all(TB_XJUMP, prev_s.len, prev_s.idx, (uint32_t)i);
        }
        bool do_y = false;
        if (Sn[i] > S[curr][i]) do_y = true;
        else if (Sn[i] == S[curr][i]) do_y = traceback.get(i, n).get_s_len() > traceback.get(i, j).get_s_len();
        if (do_y) {
            S[curr][i] = Sn[i];
            SValue s_value = traceback.get(i, j - Ly[i]).get_s();
            traceback.get_mut(i, j).set_s_all(TB_YCLIP_SUFFIX, s_value.len, s_value.idx, (uint32_t)i);
        }
        bool do_x = false;
        {
            int32_t v = S[curr][i] + scoring.xclip_suffix;
            if (v > S[curr][m]) do_x = true;
            else if (v == S[curr][m]) do_x = traceback.get(i, j).get_s_len() > traceback.get(m, j).get_s_len();
        }
        if (do_x) {
            S[curr][m] = S[curr][i] + scoring.xclip_suffix;
            Lx[j] = m - i;
            SValue prev_s = traceback.get(i, j).get_s();
            traceback.get_mut(m, j).set_s_all(TB_XCLIP_SUFFIX, prev_s.len, prev_s.idx, (uint32_t)i);
        }
    }
    for (size_t i = 1; i <= m; ++i) {
        const size_t j = n;
        const size_t curr = j % 2;
        int32_t i_score = S[curr][i - 1] + scoring.gap_open + scoring.gap_extend;
        if (i_score > I[curr][i]) {
            I[curr][i] = i_score;
            SValue s_value = traceback.get(i - 1, j).get_s();
            traceback.get_mut(i, j).set_i(s_value.tb, s_value.len + 1);
        }
        if (i_score > S[curr][i]) {
            S[curr][i] = i_score;
            uint32_t prev_len = traceback.get(i, j).get_i_len();
            traceback.get_mut(i, j).set_s_all(TB_INS, prev_len, contig_idx, (uint32_t)(i - 1));
            if (S[curr][i] + scoring.xclip_suffix > S[curr][m]) {
                S[curr][m] = S[curr][i] + scoring.xclip_suffix;
                Lx[j] = m - i;
                traceback.get_mut(m, j).set_s_all(TB_XCLIP_SUFFIX, prev_len, contig_idx, (uint32_t)i);
            }
        }
    }
}

JumpInfo SingleContigAligner::get_jump_info(size_t m, size_t j, int32_t jump_score) const {   // :677-697
    const size_t cur = j % 2;
    int32_t best_jump_score = S[cur][0] + jump_score;
    size_t best_jump_from = 0;
    for (size_t k = 1; k <= m; ++k) {
        if (best_jump_score < S[cur][k] + jump_score) {
            best_jump_score = S[cur][k] + jump_score;
            best_jump_from = k;
        }
    }
    uint32_t best_jump_len = traceback.get(best_jump_from, j).get_s_len() + 1;
    JumpInfo r; r.score = best_jump_score; r.from = (uint32_t)best_jump_from; r.idx = contig_idx; r.len = best_jump_len;
    return r;
}

Alignment SingleContigAligner::custom(const uint8_t* x, size_t m, const uint8_t* y, size_t n) {   // :705-729
    init_matrices(m, n);
    for (size_t j = 1; j <= n; ++j) {
        size_t curr = j % 2, prev = 1 - curr;
        init_column(j, curr, m, n);
        JumpInfo ji = get_jump_info(m, j - 1, scoring.jump_score_same_contig_and_strand);
        fill_column(x, y, m, n, j, prev, curr, ji);
    }
    fill_last_column_and_end_clipping(m, n);
    std::vector<const SingleContigAligner*> aligners{this};
    return orc::traceback(aligners, n);
}

// global / querylocal / targetlocal / local — :733-872
Alignment SingleContigAligner::with_mode(Mode mode, const uint8_t* x, size_t m, const uint8_t* y, size_t n) {
    int32_t saved[4] = {scoring.xclip_prefix, scoring.xclip_suffix, scoring.yclip_prefix, scoring.yclip_suffix};
    switch (mode) {
        case Global: scoring.xclip_prefix = scoring.xclip_suffix = scoring.yclip_prefix = scoring.yclip_suffix = MIN_SCORE; break;
        case QueryLocal: scoring.xclip_prefix = scoring.xclip_suffix = MIN_SCORE; scoring.yclip_prefix = scoring.yclip_suffix = 0; break;
        case TargetLocal: scoring.xclip_prefix = scoring.xclip_suffix = 0; scoring.yclip_prefix = scoring.yclip_suffix = MIN_SCORE; break;
        case Local: scoring.xclip_prefix = scoring.xclip_suffix = scoring.yclip_prefix = scoring.yclip_suffix = 0; break;
        case Custom: break;
    }
    Alignment a = custom(x, m, y, n);
    a.mode = mode;
    auto drop = [&](bool dx, bool dy) {
        std::vector<Op> kept;
        for (const Op& op : a.operations) {
            if (dx && op.kind == Xclip) continue;
            if (dy && op.kind == Yclip) continue;
            kept.push_back(op);
        }
        a.operations.swap(kept);
    };
    if (mode == QueryLocal) drop(false, true);
    if (mode == TargetLocal) drop(true, false);
    if (mode == Local) drop(true, true);
    scoring.xclip_prefix = saved[0]; scoring.xclip_suffix = saved[1];
    scoring.yclip_prefix = saved[2]; scoring.yclip_suffix = saved[3];
    return a;
}

// ----------------------------------------------------------------------------------------------
// traceback — align/traceback/mod.rs:129-373
// ----------------------------------------------------------------------------------------------
Alignment traceback(const std::vector<const SingleContigAligner*>& aligners, size_t n) {   // :129-150
    size_t aligner_offset = 0;
    int32_t score = MIN_SCORE;
    uint32_t alignment_length = 0;
    for (size_t off = 0; off < aligners.size(); ++off) {
        const SingleContigAligner* cur = aligners[off];
        size_t m = cur->traceback.rows - 1;
        int32_t cur_score = cur->S[n % 2][m];
        uint32_t cur_len = cur->traceback.get(m, n).get_s_len();
        bool update = cur_score > score || (cur_score == score && cur_len > alignment_length);
        if (update) { aligner_offset = off; score = cur_score; alignment_length = cur_len; }
    }
    auto r = traceback_from(aligners, n, aligners[aligner_offset]->contig_idx);
    if (!r) throw std::runtime_error("traceback_from returned None");
    return *r;
}

std::vector<Alignment> traceback_all(const std::vector<const SingleContigAligner*>& aligners, size_t n,
                                     const std::set<uint32_t>& consider) {   // :152-217
    std::vector<Alignment> alignments;
    std::set<uint32_t> seen;
    size_t guard = 0;
    while (seen.size() < consider.size()) {
        if (++guard > 4 * consider.size() + 16) break;   // not in the reference: protects against its no-progress loop
        size_t aligner_offset = 0;
        int32_t score = MIN_SCORE;
        uint32_t alignment_length = 0;
        for (size_t off = 0; off < aligners.size(); ++off) {
            const SingleContigAligner* cur = aligners[off];
            if (!consider.count(cur->contig_idx)) continue;
            if (seen.count(cur->contig_idx)) continue;
            size_t m = cur->traceback.rows - 1;
            int32_t cur_score = cur->S[n % 2][m];
            uint32_t cur_len = cur->traceback.get(m, n).get_s_len();
            bool update = cur_score > score || (cur_score == score && cur_len > alignment_length);
            if (update) { aligner_offset = off; score = cur_score; alignment_length = cur_len; }
        }
        auto r = traceback_from(aligners, n, aligners[aligner_offset]->contig_idx);
        if (!r) {
            uint32_t ci = aligners[aligner_offset]->contig_idx;
            if (consider.count(ci)) seen.insert(ci);
            continue;
        }
        const Alignment& a = *r;
        if (consider.count((uint32_t)a.start_contig_idx)) seen.insert((uint32_t)a.start_contig_idx);
        if (consider.count((uint32_t)a.end_contig_idx)) seen.insert((uint32_t)a.end_contig_idx);
        for (const Op& op : a.operations)
            if (op.kind == Xjump && consider.count((uint32_t)op.a)) seen.insert((uint32_t)op.a);
        alignments.push_back(a);
    }
    return alignments;
}

std::optional<Alignment> traceback_from(const std::vector<const SingleContigAligner*>& aligners, size_t n,
                                        uint32_t contig_index) {   // :219-373
    size_t j = n;
    std::vector<Op> operations;
    size_t xstart = 0, ystart = 0, yend = n;
    assert(!aligners.empty());
    uint32_t max_contig_idx = 0;
    for (auto* a : aligners) max_contig_idx = std::max(max_contig_idx, a->contig_idx);
    std::vector<std::optional<size_t>> map(max_contig_idx + 1);
    for (size_t ai = 0; ai < aligners.size(); ++ai)
        if (!aligners[ai]->traceback.matrix.empty()) map[aligners[ai]->contig_idx] = ai;
    if (contig_index > max_contig_idx || !map[contig_index]) return std::nullopt;
    const SingleContigAligner* cur_aligner = aligners[*map[contig_index]];
    int32_t score = cur_aligner->S[n % 2][cur_aligner->traceback.rows - 1];
    uint32_t alignment_length = cur_aligner->traceback.get(cur_aligner->traceback.rows - 1, n).get_s_len();
    uint32_t contig_idx = cur_aligner->contig_idx;
    size_t xlen = cur_aligner->traceback.rows - 1;
    uint32_t cur_contig_idx = contig_idx;
    size_t i = cur_aligner->traceback.rows - 1;
    size_t xend = cur_aligner->traceback.rows - 1;
    uint16_t last_layer = cur_aligner->traceback.get(i, j).get_s().tb;
    for (;;) {
        if (cur_contig_idx > max_contig_idx || !map[cur_contig_idx]) return std::nullopt;
        cur_aligner = aligners[*map[cur_contig_idx]];
        uint16_t next_layer;
        if (last_layer == TB_START) break;
        switch (last_layer) {
            case TB_INS:
                operations.push_back(Op{Ins});
                next_layer = cur_aligner->traceback.get(i, j).get_i_tb();
                i -= 1;
                break;
            case TB_DEL:
                operations.push_back(Op{Del});
                next_layer = cur_aligner->traceback.get(i, j).get_d_tb();
                j -= 1;
                break;
            case TB_MATCH: case TB_SUBST: {
                operations.push_back(Op{last_layer == TB_MATCH ? Match : Subst});
                SValue s_value = cur_aligner->traceback.get(i, j).get_s();
                size_t s_from = s_value.from;
                if (s_value.idx != cur_contig_idx || s_from != i - 1) {
                    operations.push_back(Op{Xjump, cur_contig_idx, i - 1});
                    cur_contig_idx = s_value.idx;
                    if (cur_contig_idx > max_contig_idx || !map[cur_contig_idx]) return std::nullopt;
                    cur_aligner = aligners[*map[cur_contig_idx]];
                }
                i = s_from;
                j -= 1;
                next_layer = cur_aligner->traceback.get(s_from, j).get_s().tb;
                break;
            }
            case TB_XCLIP_PREFIX:
                next_layer = cur_aligner->traceback.get(0, j).get_s().tb;
                if (next_layer == TB_START || next_layer == TB_YCLIP_PREFIX) {
                    operations.push_back(Op{Xclip, i});
                    xstart = i;
                }
                i = 0;
                break;
            case TB_XCLIP_SUFFIX:
                if (operations.empty() || operations.front().kind == Yclip) {
                    operations.push_back(Op{Xclip, cur_aligner->Lx[j]});
                    xend = i - cur_aligner->Lx[j];
                }
                i -= cur_aligner->Lx[j];
                next_layer = cur_aligner->traceback.get(i, j).get_s().tb;
                break;
            case TB_YCLIP_PREFIX:
                operations.push_back(Op{Yclip, j});
                ystart = j;
                j = 0;
                next_layer = cur_aligner->traceback.get(i, 0).get_s().tb;
                break;
            case TB_YCLIP_SUFFIX: {
                operations.push_back(Op{Yclip, cur_aligner->Ly[i]});
                size_t s_from = cur_aligner->traceback.get(i, j).get_s().from;
                j -= cur_aligner->Ly[i];
                if (s_from != i) {
                    operations.push_back(Op{Xjump, cur_contig_idx, i});
                    i = s_from;
                }
                yend = j;
                next_layer = cur_aligner->traceback.get(i, j).get_s().tb;
                break;
            }
            case TB_XJUMP: {
                SValue s_value = cur_aligner->traceback.get(i, j).get_s();
                operations.push_back(Op{Xjump, cur_contig_idx, i});
                cur_contig_idx = s_value.idx;
                if (cur_contig_idx > max_contig_idx || !map[cur_contig_idx]) return std::nullopt;
                cur_aligner = aligners[*map[cur_contig_idx]];
                i = s_value.from;
                next_layer = cur_aligner->traceback.get(i, j).get_s().tb;
                break;
            }
            default: throw std::runtime_error("Dint expect this!");
        }
        last_layer = next_layer;
    }
    std::reverse(operations.begin(), operations.end());
    bool all_special = true;
    for (const Op& op : operations) if (!(op.kind == Xclip || op.kind == Yclip || op.kind == Xjump)) { all_special = false; break; }
    if (all_special) { xstart = 0; xend = 0; ystart = 0; yend = 0; }
    Alignment a;
    a.score = score; a.ystart = ystart; a.xstart = xstart; a.yend = yend; a.xend = xend; a.xlen = xlen; a.ylen = n;
    a.start_contig_idx = cur_contig_idx; a.end_contig_idx = contig_idx; a.operations = std::move(operations);
    a.mode = Custom; a.length = alignment_length;
    return a;
}

// ----------------------------------------------------------------------------------------------
// Alignment — align/alignment.rs
// ----------------------------------------------------------------------------------------------
std::string Alignment::cigar() const {   // :105-149
    std::string cigar;
    if (operations.empty()) return cigar;
    size_t contig_idx = start_contig_idx;
    int32_t x_index = (int32_t)xstart;
    const Op* last_op = &operations.front();
    size_t last_len = 0;
    for (const Op& op : operations) {
        if ((op.is_special() || op != *last_op) && last_len > 0)
            cigar += std::to_string(last_len) + last_op->as_string(contig_idx, (size_t)x_index);
        if (op.is_special()) {
            cigar += op.as_string(contig_idx, (size_t)x_index);
            x_index += op.length_on_x((size_t)x_index);
            last_op = &op; last_len = 0;
            if (op.kind == Xjump) contig_idx = op.a;
        } else if (op == *last_op) {
            x_index += op.length_on_x((size_t)x_index);
            last_len += 1;
        } else {
            x_index += op.length_on_x((size_t)x_index);
            last_op = &op; last_len = 1;
        }
    }
    if (last_len > 0) cigar += std::to_string(last_len) + last_op->as_string(contig_idx, (size_t)x_index);
    return cigar;
}

std::optional<size_t> Alignment::earliest_x_base_for(size_t contig_idx) const {   // :153-173
    if (operations.empty()) return std::nullopt;
    if (start_contig_idx == contig_idx) return xstart;
    size_t x_contig_idx = start_contig_idx;
    int32_t x_index = (int32_t)xstart;
    for (const Op& op : operations) {
        if (x_contig_idx == contig_idx) return (size_t)x_index;
        if (op.kind == Xjump) x_contig_idx = op.a;
        x_index += op.length_on_x((size_t)x_index);
    }
    return std::nullopt;
}

std::optional<size_t> Alignment::latest_x_base_for(size_t contig_idx) const {   // :177-200
    if (operations.empty()) return std::nullopt;
    size_t x_contig_idx = start_contig_idx;
    int32_t x_index = (int32_t)xstart;
    std::optional<size_t> latest;
    if (x_contig_idx == contig_idx) latest = xstart;
    for (const Op& op : operations) {
        if (op.kind == Xjump) x_contig_idx = op.a;
        if (x_contig_idx == contig_idx) latest = (size_t)x_index;
        x_index += op.length_on_x((size_t)x_index);
    }
    return latest;
}

bool Alignment::validate(std::string* why) const {   // :56-103
    auto fail = [&](const char* w) { if (why) *why = w; return false; };
    switch (mode) {
        case Global: if (!(xstart == 0 && xend == xlen && ystart == 0 && yend == ylen)) return fail("global bounds"); break;
        case TargetLocal: if (!(xend <= xlen && ystart == 0 && yend == ylen)) return fail("targetlocal bounds"); break;
        case QueryLocal: if (!(xstart == 0 && xend == xlen && yend <= ylen)) return fail("querylocal bounds"); break;
        case Local: if (!(xend <= xlen && yend <= ylen)) return fail("local bounds"); break;
        default: break;
    }
    int32_t xe = (int32_t)xstart; size_t ye = ystart; size_t eci = end_contig_idx; size_t len = 0;
    for (const Op& op : operations) {
        xe += op.length_on_x((size_t)xe);
        ye += op.length_on_y();
        if (op.kind == Xjump) eci = op.a;
        if (op.kind == Match || op.kind == Subst || op.kind == Del || op.kind == Ins) len += 1;
        if (!(xe <= (int32_t)xlen)) return fail("xend <= xlen");
        if (!(ye <= ylen)) return fail("yend <= ylen");
    }
    if (xend != (size_t)xe) return fail("xend");
    if (yend != ye) return fail("yend");
    if (end_contig_idx != eci) return fail("end_contig_idx");
    if (length != len) return fail("length");
    return true;
}

Alignment Alignment::split_at_y(size_t y_pivot) const {   // :207-360
    if (operations.empty()) return *this;
    assert(!(operations.front().kind == Xclip || operations.front().kind == Yclip));
    assert(!(operations.back().kind == Xclip || operations.back().kind == Yclip));
    size_t x_index = xstart, y_index = ystart, contig_index = start_contig_idx, op_index = 0;
    auto is_aln = [](const Op& op) { return op.kind == Match || op.kind == Subst || op.kind == Del || op.kind == Ins; };
    for (const Op& op : operations) {                                    // :225-237
        if (is_aln(op)) break;
        if (op.kind == Xjump) contig_index = op.a;
        y_index += op.length_on_y();
        x_index = (size_t)((int32_t)x_index + op.length_on_x(x_index));
        op_index += 1;
    }
    for (size_t k = op_index; k < operations.size(); ++k) {              // :240-250
        const Op& op = operations[k];
        if (y_index + op.length_on_y() >= y_pivot) break;
        if (op.kind == Xjump) contig_index = op.a;
        y_index += op.length_on_y();
        x_index = (size_t)((int32_t)x_index + op.length_on_x(x_index));
        op_index += 1;
    }
    Alignment pre;                                                       // :251-264
    pre.xstart = xstart; pre.xend = x_index + 1; pre.ystart = ystart; pre.yend = y_index + 1;
    pre.xlen = 0; pre.ylen = 0; pre.start_contig_idx = start_contig_idx; pre.end_contig_idx = contig_index;
    pre.operations.assign(operations.begin(), operations.begin() + (op_index + 1));
    pre.mode = mode; pre.score = 0; pre.length = 0;
    assert(y_pivot >= pre.yend);
    {                                                                    // :268-281
        size_t start = op_index;
        for (size_t k = start; k < operations.size(); ++k) {
            const Op& op = operations[k];
            if (y_index >= y_pivot && is_aln(op)) break;
            if (op.kind == Xjump) contig_index = op.a;
            y_index += op.length_on_y();
            x_index = (size_t)((int32_t)x_index + op.length_on_x(x_index));
            op_index += 1;
        }
    }
    Alignment post;                                                      // :284-297
    post.xstart = x_index; post.xend = xend; post.ystart = y_index; post.yend = yend;
    post.start_contig_idx = contig_index; post.end_contig_idx = end_contig_idx;
    post.operations.assign(operations.begin() + op_index, operations.end());
    post.mode = mode;
    Alignment aln;                                                       // :300-313
    aln.start_contig_idx = post.start_contig_idx; aln.end_contig_idx = pre.end_contig_idx;
    aln.xstart = post.xstart; aln.ystart = post.ystart - y_pivot;
    aln.xend = pre.xend; aln.yend = pre.yend + ylen - y_pivot;
    aln.ylen = ylen; aln.xlen = xlen; aln.score = score; aln.mode = mode; aln.length = length;
    bool x_clip = (aln.mode == Global || aln.mode == QueryLocal);
    bool y_clip = (aln.mode == Global || aln.mode == TargetLocal);
    if (x_clip && aln.xstart > 0) { aln.operations.push_back(Op{Xclip, aln.xstart}); aln.xstart = 0; }
    if (y_clip && aln.ystart > 0) { aln.operations.push_back(Op{Yclip, aln.ystart}); aln.ystart = 0; }
    aln.operations.insert(aln.operations.end(), post.operations.begin(), post.operations.end());
    if (pre.start_contig_idx != post.end_contig_idx || pre.xstart != post.xend)
        aln.operations.push_back(Op{Xjump, pre.start_contig_idx, pre.xstart});
    size_t yjump_len = aln.ylen + pre.ystart - post.yend;
    if (yjump_len > 0) aln.operations.push_back(Op{Yjump, yjump_len});
    aln.operations.insert(aln.operations.end(), pre.operations.begin(), pre.operations.end());
    if (x_clip && aln.xend < aln.xlen) { aln.operations.push_back(Op{Xclip, aln.xlen - aln.xend}); aln.xend = aln.xlen; }
    if (y_clip && aln.yend < aln.ylen) { aln.operations.push_back(Op{Xclip, aln.ylen - aln.yend}); aln.yend = aln.ylen; }   // sic: Xclip (:355)
    return aln;
}

// ----------------------------------------------------------------------------------------------
// MultiContigAligner — align/aligners/multi_contig_aligner.rs
// ----------------------------------------------------------------------------------------------
std::optional<size_t> MultiContigAligner::contig_index_for_strand(bool is_forward, const std::string& name) const {   // :83-90
    for (const auto& c : contigs)
        if (c.is_forward == is_forward && c.name == name) return (size_t)c.aligner.contig_idx;
    return std::nullopt;
}

void MultiContigAligner::add_contig(const std::string& name, bool is_forward, const uint8_t* seq, size_t len,
                                    bool circular, const Scoring& scoring) {   // :93-133
    if (contig_index_for_strand(is_forward, name)) throw std::runtime_error("Contig already added!");
    ContigAligner c;
    c.name = name; c.is_forward = is_forward; c.seq.assign(seq, seq + len);
    c.aligner.scoring = scoring;
    c.aligner.contig_idx = (uint32_t)contigs.size();
    c.aligner.circular = circular;
    contigs.push_back(std::move(c));
    // the struct-level to_opposite_strand map (:116-132) is never read by custom() (:294-302 use a local one)
}

Alignment MultiContigAligner::custom_with_subset(const uint8_t* y, size_t n, const std::set<uint32_t>* idx) {   // :178-223
    if (!idx) return custom(y, n);
    if (idx->empty()) throw std::runtime_error("Subsetted to an empty set of contigs");
    std::vector<ContigAligner> included, excluded;
    for (auto& c : contigs) {
        if (idx->count(c.aligner.contig_idx)) included.push_back(std::move(c));
        else excluded.push_back(std::move(c));
    }
    if (included.empty()) throw std::runtime_error("included is empty");
    contigs = std::move(included);
    Alignment aln = custom(y, n);
    std::vector<ContigAligner> all;
    for (auto& c : contigs) all.push_back(std::move(c));
    for (auto& c : excluded) all.push_back(std::move(c));
    std::stable_sort(all.begin(), all.end(), [](const ContigAligner& a, const ContigAligner& b) {
        return a.aligner.contig_idx < b.aligner.contig_idx; });
    contigs = std::move(all);
    return aln;
}

Alignment MultiContigAligner::custom(const uint8_t* y, size_t n) {   // :231-361
    size_t max_contig_index = 0;
    for (auto& c : contigs) max_contig_index = std::max(max_contig_index, (size_t)c.aligner.contig_idx);
    // contig_idx -> POSITION in this->contigs of the same-name opposite-strand aligner (:241-262)
    std::vector<std::optional<size_t>> to_opp(max_contig_index + 1);
    for (size_t i = 0; i < contigs.size(); ++i) {
        size_t li = contigs[i].aligner.contig_idx;
        if (to_opp[li]) continue;
        for (size_t j = i + 1; j < contigs.size(); ++j) {
            size_t ri = contigs[j].aligner.contig_idx;
            if (contigs[i].name == contigs[j].name && contigs[i].is_forward != contigs[j].is_forward) {
                if (to_opp[li]) throw std::runtime_error("opposite strand already set");
                to_opp[li] = j;
                to_opp[ri] = i;
            }
        }
    }
    for (auto& c : contigs) { c.aligner.init_matrices(c.len(), n); cells_filled += (uint64_t)c.len() * n; }
    for (size_t j = 1; j <= n; ++j) {
        size_t curr = j % 2, prev = 1 - curr;
        for (auto& c : contigs) c.aligner.init_column(j, curr, c.len(), n);
        std::vector<JumpInfo> inter;                                      // :280-289
        inter.reserve(contigs.size());
        for (auto& c : contigs) {
            JumpInfo info = c.aligner.get_jump_info(c.len(), j - 1, c.aligner.scoring.jump_score_inter_contig);
            info.idx = c.aligner.contig_idx;
            inter.push_back(info);
        }
        std::vector<JumpInfo> best(max_contig_index + 1);                 // :292-331
        for (auto& c : contigs) {
            const ContigAligner* opp = to_opp[c.aligner.contig_idx] ? &contigs[*to_opp[c.aligner.contig_idx]] : nullptr;
            JumpInfo same = c.aligner.get_jump_info(c.len(), j - 1, c.aligner.scoring.jump_score_same_contig_and_strand);
            std::optional<JumpInfo> flip;
            if (opp) {
                JumpInfo info = opp->aligner.get_jump_info(opp->len(), j - 1,
                                                           opp->aligner.scoring.jump_score_same_contig_opposite_strand);
                info.idx = opp->aligner.contig_idx;
                flip = info;
            }
            // jump_info_for_inter_contig (:158-169): max_by_key((score,len)) => LAST maximum wins
            uint32_t opp_idx = opp ? opp->aligner.contig_idx : c.aligner.contig_idx;
            std::optional<JumpInfo> ic;
            for (const JumpInfo& info : inter) {
                if (info.idx == c.aligner.contig_idx || info.idx == opp_idx) continue;
                if (!ic || info.score > ic->score || (info.score == ic->score && info.len >= ic->len)) ic = info;
            }
            JumpInfo bj = same;
            if (flip && flip->score > bj.score) bj = *flip;
            if (ic && ic->score > bj.score) bj = *ic;
            best[c.aligner.contig_idx] = bj;
        }
        for (auto& c : contigs)
            c.aligner.fill_column(c.seq.data(), y, c.len(), n, j, prev, curr, best[c.aligner.contig_idx]);
    }
    for (auto& c : contigs) c.aligner.fill_last_column_and_end_clipping(c.len(), n);
    std::vector<const SingleContigAligner*> aligners;
    for (auto& c : contigs) aligners.push_back(&c.aligner);
    return orc::traceback(aligners, n);
}

std::vector<Alignment> MultiContigAligner::traceback_all(size_t n, const std::set<uint32_t>* idx) {   // :363-378
    std::set<uint32_t> consider;
    if (idx && idx->size() < len()) consider = *idx;
    else for (auto& c : contigs) consider.insert(c.aligner.contig_idx);
    std::vector<const SingleContigAligner*> aligners;
    for (auto& c : contigs) aligners.push_back(&c.aligner);
    return orc::traceback_all(aligners, n, consider);
}

std::optional<Alignment> MultiContigAligner::traceback_from(size_t n, size_t contig_index) {   // :380-387
    std::vector<const SingleContigAligner*> aligners;
    for (auto& c : contigs) aligners.push_back(&c.aligner);
    return orc::traceback_from(aligners, n, (uint32_t)contig_index);
}

// ----------------------------------------------------------------------------------------------
// util/dna.rs:5-41
// ----------------------------------------------------------------------------------------------
std::vector<uint8_t> reverse_complement(const uint8_t* s, size_t n) {
    static uint8_t comp[256]; static bool init = false;
    if (!init) {
        for (int v = 0; v < 256; ++v) comp[v] = (uint8_t)v;
        const char* A = "AGCTYRWSKMDVHBN"; const char* B = "TCGARYWSMKHBDVN";
        for (int k = 0; k < 15; ++k) { comp[(uint8_t)A[k]] = (uint8_t)B[k]; comp[(uint8_t)A[k] + 32] = (uint8_t)(B[k] + 32); }
        init = true;
    }
    std::vector<uint8_t> r(n);
    for (size_t k = 0; k < n; ++k) r[k] = comp[s[n - 1 - k]];
    return r;
}

// ----------------------------------------------------------------------------------------------
// Options / Aligners — align/aligners/mod.rs
// ----------------------------------------------------------------------------------------------
void Options::clipping(int32_t& xp, int32_t& xs, int32_t& yp, int32_t& ys) const {   // :123-131
    switch (mode) {
        case Local: xp = xs = yp = ys = 0; break;
        case QueryLocal: xp = xs = MIN_SCORE; yp = ys = 0; break;
        case TargetLocal: xp = xs = 0; yp = ys = MIN_SCORE; break;
        case Global: xp = xs = yp = ys = MIN_SCORE; break;
        default: throw std::runtime_error("Custom alignment mode not supported");
    }
}
Scoring Options::contig_scoring() const {   // :143-167
    Scoring s;
    s.gap_open = gap_open; s.gap_extend = gap_extend;
    s.jump_score_same_contig_and_strand = jump_score_same_contig_and_strand.value_or(default_jump_score);
    s.jump_score_same_contig_opposite_strand = jump_score_same_contig_opposite_strand.value_or(default_jump_score);
    s.jump_score_inter_contig = jump_score_inter_contig.value_or(default_jump_score);
    s.match_score = match_score; s.mismatch_score = mismatch_score;
    clipping(s.xclip_prefix, s.xclip_suffix, s.yclip_prefix, s.yclip_suffix);
    return s;
}

Aligners Aligners::build(const Options& opts, const std::vector<TargetSeq>& target_seqs) {   // :171-211
    Aligners a; a.opts = opts;
    Scoring sc = opts.contig_scoring();
    for (const auto& t : target_seqs) a.multi_contig.add_contig(t.name, true, t.fwd.data(), t.fwd.size(), opts.circular, sc);
    if (opts.double_strand)
        for (const auto& t : target_seqs) a.multi_contig.add_contig(t.name, false, t.revcomp.data(), t.revcomp.size(), opts.circular, sc);
    return a;
}

Alignment Aligners::remove_clipping(Alignment aln) const {   // :343-353
    if (opts.mode == Local || opts.mode == QueryLocal || opts.mode == TargetLocal) {
        std::vector<Op> kept;
        for (const Op& op : aln.operations)
            if (op.kind == Match || op.kind == Subst || op.kind == Ins || op.kind == Del || op.kind == Xjump) kept.push_back(op);
        aln.operations.swap(kept);
    }
    return aln;
}

Alignment Aligners::multi_contig_align(const uint8_t* q, size_t n, const std::set<uint32_t>* idx) {   // :355-363
    return remove_clipping(multi_contig.custom_with_subset(q, n, idx));
}

std::vector<Alignment> Aligners::align(const uint8_t* read, size_t n) {   // :237-340 with pre_align == false
    std::vector<uint8_t> query(read, read + n);
    for (auto& b : query) if (b >= 'a' && b <= 'z') b = (uint8_t)(b - 32);   // seq_upper_case, io.rs:64-66
    return align_subset(query.data(), n, nullptr);
}

// :289-337 — everything after the pre-alignment decision; `q` is already upper case
std::vector<Alignment> Aligners::align_subset(const uint8_t* q, size_t n, const std::set<uint32_t>* contigs_to_align) {
    std::vector<uint8_t> query(q, q + n);
    Alignment original = multi_contig_align(query.data(), n, contigs_to_align);
    std::vector<Alignment> alignments;
    if (opts.suboptimal) {
        std::vector<Alignment> news = multi_contig.traceback_all(n, contigs_to_align);
        for (auto& a : news) {
            Alignment b = remove_clipping(a);
            alignments.push_back(realign_origin(query.data(), n, b, opts.circular_slop, false));
        }
        if (alignments.size() > 1) {
            std::stable_sort(alignments.begin(), alignments.end(), [](const Alignment& a, const Alignment& b) { return -a.score < -b.score; });
            float min_score = (float)alignments[0].score * opts.suboptimal_pct / 100.0f;
            std::vector<Alignment> kept;
            for (auto& a : alignments) if ((float)a.score >= min_score) kept.push_back(a);
            alignments.swap(kept);
        }
    } else {
        alignments.push_back(realign_origin(query.data(), n, original, opts.circular_slop, false));
    }
    return alignments;
}

// :365-410
static void start_end_for_realign(const Aligners& A, const Alignment& alignment, size_t slop,
                                  std::optional<size_t>& cs, std::optional<size_t>& ce) {
    cs.reset(); ce.reset();
    if (alignment.xstart <= slop && A.multi_contig.is_circular(alignment.start_contig_idx)) cs = alignment.start_contig_idx;
    if (alignment.xlen <= alignment.xend + slop && A.multi_contig.is_circular(alignment.end_contig_idx)) ce = alignment.end_contig_idx;
    if (cs && ce && *cs == *ce) { cs.reset(); ce.reset(); return; }
    if (!cs && !ce) return;
    if (!cs || alignment.yend == alignment.ylen) cs.reset();
    if (!ce || 0 == alignment.ystart) ce.reset();
}

Alignment Aligners::realign_origin(const uint8_t* query, size_t n, Alignment alignment, size_t slop, bool all_contigs) {   // :442-553
    std::optional<size_t> contig_at_start, contig_at_end;
    start_end_for_realign(*this, alignment, slop, contig_at_start, contig_at_end);
    if (!contig_at_start && !contig_at_end) return alignment;
    std::set<uint32_t> contig_indexes;
    if (all_contigs) { for (size_t k = 0; k < multi_contig.len(); ++k) contig_indexes.insert((uint32_t)k); }
    else {
        contig_indexes.insert((uint32_t)alignment.start_contig_idx);
        contig_indexes.insert((uint32_t)alignment.end_contig_idx);
        for (const Op& op : alignment.operations) if (op.kind == Xjump) contig_indexes.insert((uint32_t)op.a);
    }
    Alignment best = alignment;
    auto rotate = [&](size_t p) { std::vector<uint8_t> r(query + p, query + n); r.insert(r.end(), query, query + p); return r; };
    // realign_and_split_at_y (:412-431)
    auto realign = [&](const std::vector<uint8_t>& q, size_t contig_idx, size_t y_pivot) {
        multi_contig_align(q.data(), q.size(), &contig_indexes);
        std::optional<Alignment> na = multi_contig.traceback_from(q.size(), contig_idx);
        if (na && na->score > best.score && na->start_contig_idx == contig_idx && best.end_contig_idx == contig_idx)
            best = remove_clipping(*na).split_at_y(y_pivot);
    };
    if (contig_at_start) {                                                // :475-509
        size_t sc = *contig_at_start;
        std::vector<uint8_t> first = rotate(alignment.yend);
        size_t yend = alignment.ystart;
        for (const Op& op : alignment.operations) {
            if (op.kind == Xjump && op.a != sc) break;
            yend += op.length_on_y();
        }
        std::vector<uint8_t> second = rotate(yend);
        realign(first, sc, alignment.ylen - alignment.yend);
        realign(second, sc, alignment.ylen - yend);
    }
    if (contig_at_end) {                                                  // :512-550
        size_t ec = *contig_at_end;
        std::vector<uint8_t> first = rotate(alignment.ystart);
        size_t ystart = alignment.ystart, ycur = alignment.ystart, xidx = alignment.start_contig_idx;
        for (const Op& op : alignment.operations) {
            if (op.kind == Xjump) {
                if (op.a == ec && xidx != ec) ystart = ycur;
                xidx = op.a;
            }
            ycur += op.length_on_y();
        }
        std::vector<uint8_t> second = rotate(ystart);
        realign(first, ec, alignment.ylen - alignment.ystart);
        realign(second, ec, alignment.ylen - ystart);
    }
    return best;
}

// ----------------------------------------------------------------------------------------------
// SubAlignmentBuilder — align/sub_alignment.rs  (names follow the reference: "query" == x before swap)
// ----------------------------------------------------------------------------------------------
namespace {
struct SubBuilder {
    bool use_eq_and_x; char match_kind, mismatch_kind;
    std::vector<CigarOp> elements;
    size_t query_start = 0, target_start = 0, query_offset = 0, target_offset = 0;
    int32_t score = 0, num_edits = 0; size_t contig_idx = 0;
    bool cmp_op(const Op& last, const Op& cur) const {   // :37-45
        if (use_eq_and_x) return last == cur;
        return last == cur || (last.kind == Subst && cur.kind == Match) || (last.kind == Match && cur.kind == Subst);
    }
    SubAlignment snapshot() const {
        SubAlignment a; a.contig_idx = contig_idx; a.query_start = query_start; a.query_end = query_offset;
        a.target_start = target_start; a.target_end = target_offset; a.cigar = elements; a.score = score; a.num_edits = num_edits;
        return a;
    }
    std::optional<SubAlignment> add_op(const Op& op, size_t op_len, const Scoring& sc) {   // :48-131
        switch (op.kind) {
            case Match: score += sc.score('A', 'A') * (int32_t)op_len; query_offset += op_len; target_offset += op_len;
                elements.push_back({match_kind, op_len}); return std::nullopt;
            case Subst: score += sc.score('A', 'C') * (int32_t)op_len; query_offset += op_len; target_offset += op_len;
                elements.push_back({mismatch_kind, op_len}); return std::nullopt;
            case Del: score += sc.gap_open + sc.gap_extend * (int32_t)op_len; target_offset += op_len;
                elements.push_back({'D', op_len}); return std::nullopt;
            case Ins: score += sc.gap_open + sc.gap_extend * (int32_t)op_len; query_offset += op_len;
                elements.push_back({'I', op_len}); return std::nullopt;
            case Xjump: {
                SubAlignment a = snapshot();
                elements.clear(); contig_idx = op.a; target_start = target_offset;
                query_start = op.b; query_offset = op.b; score = 0; num_edits = 0;
                return a;
            }
            case Yjump: {
                SubAlignment a = snapshot();
                elements.clear(); target_offset += op.a; target_start = target_offset;
                query_start = query_offset; score = 0; num_edits = 0;
                return a;
            }
            case Xclip: case Yclip:
                if (op_len != 1) throw std::runtime_error("assert op_len == 1");
                return std::nullopt;
        }
        return std::nullopt;
    }
};
// noodles Cigar::try_from merges nothing; ops are kept as given.
}  // namespace

std::vector<SubAlignment> build_sub_alignments(const Alignment& chain, bool swap, const Scoring& scoring, bool use_eq_and_x) {   // :169-241
    SubBuilder b;
    b.use_eq_and_x = use_eq_and_x;
    b.match_kind = use_eq_and_x ? '=' : 'M';
    b.mismatch_kind = use_eq_and_x ? 'X' : 'M';
    b.query_start = chain.xstart; b.target_start = chain.ystart;
    b.query_offset = b.query_start; b.target_offset = b.target_start;
    b.contig_idx = chain.start_contig_idx;
    std::vector<SubAlignment> out;
    if (chain.operations.empty()) throw std::runtime_error("chain.operations[0]: index out of bounds");   // reference panics (:185)
    Op last = chain.operations[0];
    size_t op_len = 0;
    for (size_t i = 0; i < chain.operations.size(); ++i) {
        const Op& op = chain.operations[i];
        if (op.kind == Subst || op.kind == Ins || op.kind == Del) b.num_edits += 1;
        if (b.cmp_op(last, op)) op_len += 1;
        else {
            auto a = b.add_op(last, op_len, scoring);
            if (a && a->target_start < a->target_end) out.push_back(*a);
            op_len = 1;
        }
        last = op;
    }
    auto a = b.add_op(last, op_len, scoring);
    if (a) out.push_back(*a);
    else out.push_back(b.snapshot());
    if (swap) {
        for (auto& s : out) {
            std::swap(s.query_start, s.target_start);
            std::swap(s.query_end, s.target_end);
            for (auto& c : s.cigar) { if (c.kind == 'D') c.kind = 'I'; else if (c.kind == 'I') c.kind = 'D'; }
        }
    }
    return out;
}

// ----------------------------------------------------------------------------------------------
// SamRecordFormatter::format — align/aligners/mod.rs:622-973, rendered as SAM text
// ----------------------------------------------------------------------------------------------
static std::string cigar_to_string(const std::vector<CigarOp>& c) {
    std::string s;
    for (const auto& op : c) { s += std::to_string(op.len); s.push_back(op.kind); }
    return s;
}

std::vector<std::string> format_sam(const Options& opts, const std::vector<TargetSeq>& target_seqs,
                                    const std::string& head, const std::vector<uint8_t>& bases,
                                    const std::vector<uint8_t>* quals, const std::vector<Alignment>& chains,
                                    std::optional<int32_t> pre_alignment_score, std::string* err) {
    std::vector<std::string> records;
    // header_to_name (:612-619): first whitespace-separated token
    std::string name;
    {
        size_t p = 0;
        while (p < head.size() && isspace((unsigned char)head[p])) ++p;
        size_t e = p;
        while (e < head.size() && !isspace((unsigned char)head[e])) ++e;
        name = head.substr(p, e - p);
        if (name.empty()) { if (err) *err = "empty read name"; return records; }
    }
    auto seq_str = [](const std::vector<uint8_t>& v) { return v.empty() ? std::string("*") : std::string(v.begin(), v.end()); };
    auto qual_str = [&](const std::vector<uint8_t>* q) { return (!q || q->empty()) ? std::string("*") : std::string(q->begin(), q->end()); };
    Scoring scoring = opts.contig_scoring();
    if (chains.empty()) {                                                 // :634-667
        std::string r = name + "\t4\t*\t0\t0\t*\t*\t0\t0\t" + seq_str(bases) + "\t" + qual_str(quals);
        if (pre_alignment_score) r += "\txs:i:" + std::to_string(*pre_alignment_score);
        records.push_back(r);
        return records;
    }
    int32_t primary_alignment_score = MIN_SCORE;
    std::optional<int32_t> suboptimal_score;                              // :678-685
    {
        std::optional<int32_t> sub;
        for (size_t k = 1; k < chains.size(); ++k) if (!sub || chains[k].score > *sub) sub = chains[k].score;
        if (sub && pre_alignment_score) suboptimal_score = std::max(*sub, *pre_alignment_score);
        else if (sub) suboptimal_score = sub;
        else if (pre_alignment_score) suboptimal_score = pre_alignment_score;
    }
    const size_t T = target_seqs.size();
    for (size_t chain_idx = 0; chain_idx < chains.size(); ++chain_idx) {
        const Alignment& chain = chains[chain_idx];
        bool hard_clip = !opts.soft_clip;
        std::vector<SubAlignment> subs;
        try { subs = build_sub_alignments(chain, true, scoring, opts.use_eq_and_x); }
        catch (const std::exception& e) { if (err) *err = e.what(); return {}; }
        if (subs.empty()) { if (err) *err = "ensure!(!subs.is_empty())"; return {}; }
        // max_by_key => LAST maximum (:699-714)
        size_t primary_sub_idx = 0;
        {
            bool have = false; size_t k0 = 0, k1 = 0;
            for (size_t k = 0; k < subs.size(); ++k) {
                size_t span = subs[k].query_end - subs[k].query_start;
                long long a0, a1;
                if (opts.pick_primary == 0) { a0 = (long long)span; a1 = subs[k].score; }
                else { a0 = subs[k].score; a1 = (long long)span; }
                if (!have) { have = true; primary_sub_idx = k; k0 = (size_t)0; (void)k0; (void)k1; }
                else {
                    size_t pspan = subs[primary_sub_idx].query_end - subs[primary_sub_idx].query_start;
                    long long b0, b1;
                    if (opts.pick_primary == 0) { b0 = (long long)pspan; b1 = subs[primary_sub_idx].score; }
                    else { b0 = subs[primary_sub_idx].score; b1 = (long long)pspan; }
                    if (a0 > b0 || (a0 == b0 && a1 >= b1)) primary_sub_idx = k;
                }
            }
        }
        if (chain_idx == 0) primary_alignment_score = subs[primary_sub_idx].score;
        if (opts.filter_secondary) {                                      // :723-743
            float min_score = (float)primary_alignment_score * opts.filter_secondary_pct / 100.0f;
            std::vector<SubAlignment> kept; size_t old_idx = 0;
            size_t psi = primary_sub_idx;
            for (auto& sub : subs) {
                if (old_idx == psi) primary_sub_idx = kept.size();
                if ((float)sub.score >= min_score) kept.push_back(sub);
                old_idx += 1;
            }
            subs.swap(kept);
        }
        std::vector<std::string> chain_records, sa_strings;
        for (size_t sub_idx = 0; sub_idx < subs.size(); ++sub_idx) {
            const SubAlignment& sub = subs[sub_idx];
            bool is_supplementary = sub_idx != primary_sub_idx;
            bool is_secondary = chain_idx > 0;
            if (!(sub.contig_idx < 2 * T)) { if (err) *err = "assert sub.contig_idx < 2*targets"; return {}; }
            bool is_forward = sub.contig_idx < T;
            int flags = 0;
            if (!is_forward) flags |= 16;
            if (is_secondary) flags |= 256;
            if (is_supplementary) flags |= 2048;
            std::vector<uint8_t> bases_vec; std::vector<uint8_t> quals_vec; bool have_quals = quals != nullptr;
            std::vector<CigarOp> cigar = sub.cigar;
            bool hc = hard_clip && is_secondary;
            auto slice = [&](const std::vector<uint8_t>& v) { return std::vector<uint8_t>(v.begin() + sub.query_start, v.begin() + sub.query_end); };
            if (is_forward && !hc) { bases_vec = bases; if (quals) quals_vec = *quals; }
            else if (is_forward && hc) { bases_vec = slice(bases); if (quals) quals_vec = slice(*quals); std::reverse(cigar.begin(), cigar.end()); }
            else if (!is_forward && !hc) {
                bases_vec = reverse_complement(bases.data(), bases.size());
                if (quals) { quals_vec = *quals; std::reverse(quals_vec.begin(), quals_vec.end()); }
                std::reverse(cigar.begin(), cigar.end());
            } else {
                std::vector<uint8_t> sl = slice(bases);
                bases_vec = reverse_complement(sl.data(), sl.size());
                if (quals) { quals_vec = slice(*quals); std::reverse(quals_vec.begin(), quals_vec.end()); }
                std::reverse(cigar.begin(), cigar.end());
            }
            std::string cigar_str = cigar_to_string(cigar);
            char clip_op = hc ? 'H' : 'S';
            std::vector<CigarOp> full;
            size_t clip_prefix_len = is_forward ? sub.query_start : bases.size() - sub.query_end;
            if (clip_prefix_len > 0) full.push_back({clip_op, clip_prefix_len});
            full.insert(full.end(), cigar.begin(), cigar.end());
            size_t clip_suffix_len = is_forward ? bases.size() - sub.query_end : sub.query_start;
            if (clip_suffix_len > 0) full.push_back({clip_op, clip_suffix_len});
            std::string cigar_string = cigar_to_string(full);
            size_t reference_sequence_id = sub.contig_idx % T;
            size_t reference_start = is_forward ? sub.target_start + 1
                                                : target_seqs[reference_sequence_id].fwd.size() - sub.target_end + 1;
            int mapq = chain_idx == 0 ? 60 : 0;
            std::string r = name + "\t" + std::to_string(flags) + "\t" + target_seqs[reference_sequence_id].name + "\t" +
                std::to_string(reference_start) + "\t" + std::to_string(mapq) + "\t" + (cigar_string.empty() ? "*" : cigar_string) +
                "\t*\t0\t0\t" + seq_str(bases_vec) + "\t" + (have_quals ? qual_str(&quals_vec) : std::string("*"));
            r += "\tqs:i:" + std::to_string(sub.query_start);
            r += "\tqe:i:" + std::to_string(sub.query_end);
            r += "\tts:i:" + std::to_string(sub.target_start);
            r += "\tte:i:" + std::to_string(sub.target_end);
            r += "\tas:i:" + std::to_string(chain.score);
            if (suboptimal_score) r += "\txs:i:" + std::to_string(*suboptimal_score);
            r += "\tsi:i:" + std::to_string(sub_idx);
            r += "\tsc:Z:" + cigar_str;
            r += "\tcl:i:" + std::to_string(subs.size());
            r += "\tci:i:" + std::to_string(chain_idx);
            r += "\tcn:i:" + std::to_string(chains.size());
            r += "\tAS:i:" + std::to_string(sub.score);
            r += "\tNM:i:" + std::to_string(sub.num_edits);
            chain_records.push_back(r);
            std::string sa = target_seqs[reference_sequence_id].name + "," + std::to_string(reference_start) + "," +
                (is_forward ? "+" : "-") + "," + cigar_string + "," + std::to_string(mapq) + "," + std::to_string(sub.num_edits);
            sa_strings.push_back(sa);
        }
        // rotate_right(primary_sub_idx) (:956)
        if (!sa_strings.empty()) {
            size_t k = primary_sub_idx % sa_strings.size();
            std::rotate(sa_strings.begin(), sa_strings.begin() + (sa_strings.size() - k), sa_strings.end());
        }
        std::string sa_string;
        for (size_t k = 0; k < sa_strings.size(); ++k) { if (k) sa_string += ";"; sa_string += sa_strings[k]; }
        for (auto& r : chain_records) records.push_back(r + "\tSA:Z:" + sa_string);
    }
    return records;
}

}  // namespace orc
