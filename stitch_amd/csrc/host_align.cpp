// Host-side result handling of the `stitch align` hot path: see host_align.h for the reference map.
#include "host_align.h"

#include <algorithm>
#include <cctype>

namespace stitch {

// Aligners::remove_clipping (aligners/mod.rs:343-353): in the three local modes keep only
// Match | Subst | Ins | Del | Xjump (so Xclip, Yclip AND Yjump go); Global keeps everything.
void remove_clipping(const stitch_opts& o, HAln& a) {
    if (o.mode == 3) return;
    size_t w = 0;
    for (size_t k = 0; k < a.ops.size(); ++k) {
        uint8_t kd = a.ops[k].kind;
        if (kd <= OP_INS || kd == OP_XJUMP) a.ops[w++] = a.ops[k];
    }
    a.ops.resize(w);
}

// Alignment::split_at_y (alignment.rs:207-360).  `mode` is the alignment's own mode; chains coming out of the
// traceback carry AlignmentMode::Custom (traceback/mod.rs:369), for which neither clip branch fires.
HAln split_at_y(const HAln& a, int mode, uint32_t y_pivot) {
    if (a.ops.empty()) return a;
    uint32_t x_index = a.xstart, y_index = a.ystart, contig_index = a.start_contig_idx;
    size_t op_index = 0;
    const size_t N = a.ops.size();
    auto advance = [&](const stitch_op& op) {
        if (op.kind == OP_XJUMP) contig_index = op.contig;
        y_index += op_len_y(op);
        x_index = (uint32_t)((int32_t)x_index + op_len_x(op, x_index));
        op_index += 1;
    };
    for (size_t k = 0; k < N; ++k) {                         // leading specials (:225-237)
        if (op_is_aln(a.ops[k])) break;
        advance(a.ops[k]);
    }
    for (size_t k = op_index; k < N; ++k) {                  // up to the pivot (:240-250)
        if (y_index + op_len_y(a.ops[k]) >= y_pivot) break;
        advance(a.ops[k]);
    }
    // pre-pivot half (:251-264)
    const uint32_t pre_xend = x_index + 1, pre_yend = y_index + 1, pre_end_contig = contig_index;
    const size_t pre_ops_end = op_index + 1;                 // operations[..=op_index]
    for (size_t k = op_index; k < N; ++k) {                  // specials at the pivot (:268-281)
        if (y_index >= y_pivot && op_is_aln(a.ops[k])) break;
        advance(a.ops[k]);
    }
    // post-pivot half (:284-297)
    const uint32_t post_xstart = x_index, post_ystart = y_index, post_start_contig = contig_index;
    const size_t post_ops_begin = op_index;

    HAln r;                                                  // join (:300-313)
    r.start_contig_idx = post_start_contig; r.end_contig_idx = pre_end_contig;
    r.xstart = post_xstart; r.ystart = post_ystart - y_pivot;
    r.xend = pre_xend; r.yend = pre_yend + a.ylen - y_pivot;
    r.ylen = a.ylen; r.xlen = a.xlen; r.score = a.score; r.length = a.length;
    const bool x_clip = (mode == 3 || mode == 1), y_clip = (mode == 3 || mode == 2);
    if (x_clip && r.xstart > 0) { r.ops.push_back(mk_op(OP_XCLIP, 0, r.xstart)); r.xstart = 0; }
    if (y_clip && r.ystart > 0) { r.ops.push_back(mk_op(OP_YCLIP, 0, r.ystart)); r.ystart = 0; }
    r.ops.insert(r.ops.end(), a.ops.begin() + post_ops_begin, a.ops.end());
    // pre.start_contig_idx / pre.xstart / pre.ystart are the original start; post.end_contig_idx / xend / yend the original end
    if (a.start_contig_idx != a.end_contig_idx || a.xstart != a.xend) r.ops.push_back(mk_op(OP_XJUMP, a.start_contig_idx, a.xstart));
    const uint32_t yjump_len = r.ylen + a.ystart - a.yend;
    if (yjump_len > 0) r.ops.push_back(mk_op(OP_YJUMP, 0, yjump_len));
    r.ops.insert(r.ops.end(), a.ops.begin(), a.ops.begin() + std::min(pre_ops_end, N));
    if (x_clip && r.xend < r.xlen) { r.ops.push_back(mk_op(OP_XCLIP, 0, r.xlen - r.xend)); r.xend = r.xlen; }
    if (y_clip && r.yend < r.ylen) { r.ops.push_back(mk_op(OP_XCLIP, 0, r.ylen - r.yend)); r.yend = r.ylen; }   // Xclip, as the reference (:355)
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// SubAlignmentBuilder (align/sub_alignment.rs).  Field names below are the reference's PRE-swap names: its
// "query" is x (the contig) and its "target" is y (the read); build(.., swap = true) exchanges them at the end.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct Builder {
    bool eqx; std::vector<std::pair<char, uint32_t>> elems;
    uint32_t q_start = 0, t_start = 0, q_off = 0, t_off = 0, contig = 0; int32_t score = 0, edits = 0;
    SubAln snap() const {
        SubAln s; s.contig_idx = contig; s.query_start = q_start; s.query_end = q_off; s.target_start = t_start; s.target_end = t_off;
        s.cigar = elems; s.score = score; s.num_edits = edits; return s;
    }
};
inline bool same_run(bool eqx, const stitch_op& last, const stitch_op& cur) {     // cmp_op (:37-45)
    bool eq = last.kind == cur.kind && (last.kind <= OP_INS || (last.arg == cur.arg && (last.kind != OP_XJUMP || last.contig == cur.contig)));
    if (eqx) return eq;
    return eq || (last.kind == OP_SUBST && cur.kind == OP_MATCH) || (last.kind == OP_MATCH && cur.kind == OP_SUBST);
}
}  // namespace

bool build_subs(const HAln& chain, const stitch_opts& o, std::vector<SubAln>& out, std::string& err) {
    out.clear();
    if (chain.ops.empty()) { err = "chain has no alignment operations (the reference panics at sub_alignment.rs:185)"; return false; }
    Builder b; b.eqx = o.use_eq_and_x != 0;
    const char mk = b.eqx ? '=' : 'M', xk = b.eqx ? 'X' : 'M';
    b.q_start = b.q_off = chain.xstart; b.t_start = b.t_off = chain.ystart; b.contig = chain.start_contig_idx;
    // add_op (:48-131); returns true when a sub-alignment was closed into `closed`
    auto add_op = [&](const stitch_op& op, uint32_t len, SubAln& closed) -> int {
        switch (op.kind) {
            case OP_MATCH: b.score += o.match_score * (int32_t)len; b.q_off += len; b.t_off += len; b.elems.push_back({mk, len}); return 0;
            case OP_SUBST: b.score += o.mismatch_score * (int32_t)len; b.q_off += len; b.t_off += len; b.elems.push_back({xk, len}); return 0;
            case OP_DEL: b.score += o.gap_open + o.gap_extend * (int32_t)len; b.t_off += len; b.elems.push_back({'D', len}); return 0;
            case OP_INS: b.score += o.gap_open + o.gap_extend * (int32_t)len; b.q_off += len; b.elems.push_back({'I', len}); return 0;
            case OP_XJUMP:
                closed = b.snap(); b.elems.clear(); b.contig = op.contig; b.t_start = b.t_off; b.q_start = op.arg; b.q_off = op.arg;
                b.score = 0; b.edits = 0; return 1;
            case OP_YJUMP:
                closed = b.snap(); b.elems.clear(); b.t_off += op.arg; b.t_start = b.t_off; b.q_start = b.q_off; b.score = 0; b.edits = 0; return 1;
            default:
                if (len != 1) { err = "clip run longer than one operation (sub_alignment.rs:127)"; return -1; }
                return 0;
        }
    };
    stitch_op last = chain.ops[0];
    uint32_t run = 0;
    for (size_t k = 0; k < chain.ops.size(); ++k) {
        const stitch_op& op = chain.ops[k];
        if (op.kind == OP_SUBST || op.kind == OP_INS || op.kind == OP_DEL) b.edits += 1;   // counted before the flush (:189-194)
        if (same_run(b.eqx, last, op)) run += 1;
        else {
            SubAln closed; int rc = add_op(last, run, closed);
            if (rc < 0) return false;
            if (rc == 1 && closed.target_start < closed.target_end) out.push_back(closed);
            run = 1;
        }
        last = op;
    }
    SubAln closed; int rc = add_op(last, run, closed);
    if (rc < 0) return false;
    out.push_back(rc == 1 ? closed : b.snap());
    for (SubAln& s : out) {                                   // swap (:224-237)
        std::swap(s.query_start, s.target_start); std::swap(s.query_end, s.target_end);
        for (auto& c : s.cigar) c.first = c.first == 'D' ? 'I' : c.first == 'I' ? 'D' : c.first;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// SamRecordFormatter::format (aligners/mod.rs:622-973) rendered as SAM text.
// ---------------------------------------------------------------------------------------------------------------
static uint8_t comp_base(uint8_t a) {                        // util/dna.rs:5-29
    static const char* A = "AGCTYRWSKMDVHBN"; static const char* B = "TCGARYWSMKHBDVN";
    for (int k = 0; k < 15; ++k) { if (a == (uint8_t)A[k]) return (uint8_t)B[k]; if (a == (uint8_t)(A[k] + 32)) return (uint8_t)(B[k] + 32); }
    return a;
}
static std::string cig(const std::vector<std::pair<char, uint32_t>>& c) {
    std::string s; for (auto& e : c) { s += std::to_string(e.second); s.push_back(e.first); } return s;
}

bool format_sam_records(const stitch_opts& o, const std::vector<TargetInfo>& targets, const std::string& head,
                        const uint8_t* bases, const uint8_t* quals, size_t n, const std::vector<HAln>& chains,
                        bool has_prealign, int32_t prealign, std::vector<std::string>& out, std::string& err) {
    out.clear();
    size_t p = 0; while (p < head.size() && isspace((unsigned char)head[p])) ++p;       // header_to_name (:612-619)
    size_t e = p; while (e < head.size() && !isspace((unsigned char)head[e])) ++e;
    const std::string name = head.substr(p, e - p);
    if (name.empty()) { err = "empty read name"; return false; }
    auto text = [](const std::vector<uint8_t>& v) { return v.empty() ? std::string("*") : std::string(v.begin(), v.end()); };
    const std::vector<uint8_t> all_bases(bases, bases + n);
    const std::vector<uint8_t> all_quals = quals ? std::vector<uint8_t>(quals, quals + n) : std::vector<uint8_t>();
    if (chains.empty()) {                                      // unmapped (:634-667)
        std::string r = name + "\t4\t*\t0\t0\t*\t*\t0\t0\t" + text(all_bases) + "\t" + (quals ? text(all_quals) : std::string("*"));
        if (has_prealign) r += "\txs:i:" + std::to_string(prealign);
        out.push_back(r);
        return true;
    }
    const size_t T = targets.size();
    bool have_sub = false; int32_t sub_score = 0;              // (:678-685)
    for (size_t k = 1; k < chains.size(); ++k) if (!have_sub || chains[k].score > sub_score) { have_sub = true; sub_score = chains[k].score; }
    bool have_xs = have_sub || has_prealign;
    int32_t xs = have_sub && has_prealign ? std::max(sub_score, prealign) : have_sub ? sub_score : prealign;
    int32_t primary_alignment_score = MIN_SCORE;
    for (size_t ci = 0; ci < chains.size(); ++ci) {
        const HAln& chain = chains[ci];
        std::vector<SubAln> subs;
        if (!build_subs(chain, o, subs, err)) return false;
        if (subs.empty()) { err = "no sub-alignments"; return false; }
        size_t primary = 0;                                    // max_by_key keeps the LAST maximum (:699-714)
        for (size_t k = 1; k < subs.size(); ++k) {
            int64_t span_k = (int64_t)subs[k].query_end - subs[k].query_start, span_p = (int64_t)subs[primary].query_end - subs[primary].query_start;
            int64_t a0 = o.pick_primary == 0 ? span_k : subs[k].score, a1 = o.pick_primary == 0 ? subs[k].score : span_k;
            int64_t b0 = o.pick_primary == 0 ? span_p : subs[primary].score, b1 = o.pick_primary == 0 ? subs[primary].score : span_p;
            if (a0 > b0 || (a0 == b0 && a1 >= b1)) primary = k;
        }
        if (ci == 0) primary_alignment_score = subs[primary].score;
        if (o.filter_secondary) {                              // (:723-743)
            const float min_score = (float)primary_alignment_score * o.filter_secondary_pct / 100.0f;
            std::vector<SubAln> kept; const size_t old_primary = primary;
            for (size_t k = 0; k < subs.size(); ++k) {
                if (k == old_primary) primary = kept.size();
                if ((float)subs[k].score >= min_score) kept.push_back(subs[k]);
            }
            subs.swap(kept);
        }
        std::vector<std::string> recs, sa;
        for (size_t si = 0; si < subs.size(); ++si) {
            const SubAln& s = subs[si];
            if (!(s.contig_idx < 2 * T)) { err = "sub.contig_idx out of range"; return false; }
            const bool fwd = s.contig_idx < T, secondary = ci > 0, hc = !o.soft_clip && secondary;
            const int flags = (fwd ? 0 : 16) | (secondary ? 256 : 0) | (si != primary ? 2048 : 0);
            std::vector<uint8_t> b = hc ? std::vector<uint8_t>(all_bases.begin() + s.query_start, all_bases.begin() + s.query_end) : all_bases;
            std::vector<uint8_t> q;
            if (quals) q = hc ? std::vector<uint8_t>(all_quals.begin() + s.query_start, all_quals.begin() + s.query_end) : all_quals;
            std::vector<std::pair<char, uint32_t>> c = s.cigar;
            if (!fwd) { std::reverse(b.begin(), b.end()); for (auto& x : b) x = comp_base(x); std::reverse(q.begin(), q.end()); }
            if (!fwd || hc) std::reverse(c.begin(), c.end());  // forward hard-clipped secondaries are reversed too (:782-789)
            const std::string sub_cigar = cig(c);
            const char clip = hc ? 'H' : 'S';
            const uint32_t pre = fwd ? s.query_start : (uint32_t)n - s.query_end, post = fwd ? (uint32_t)n - s.query_end : s.query_start;
            std::vector<std::pair<char, uint32_t>> full;
            if (pre > 0) full.push_back({clip, pre});
            full.insert(full.end(), c.begin(), c.end());
            if (post > 0) full.push_back({clip, post});
            const std::string full_cigar = cig(full);
            const size_t tid = s.contig_idx % T;
            const uint32_t pos = fwd ? s.target_start + 1 : targets[tid].len - s.target_end + 1;
            const int mapq = ci == 0 ? 60 : 0;
            std::string r = name + "\t" + std::to_string(flags) + "\t" + targets[tid].name + "\t" + std::to_string(pos) + "\t" +
                std::to_string(mapq) + "\t" + (full_cigar.empty() ? std::string("*") : full_cigar) + "\t*\t0\t0\t" + text(b) + "\t" +
                (quals ? text(q) : std::string("*"));
            r += "\tqs:i:" + std::to_string(s.query_start) + "\tqe:i:" + std::to_string(s.query_end);
            r += "\tts:i:" + std::to_string(s.target_start) + "\tte:i:" + std::to_string(s.target_end);
            r += "\tas:i:" + std::to_string(chain.score);
            if (have_xs) r += "\txs:i:" + std::to_string(xs);
            r += "\tsi:i:" + std::to_string(si) + "\tsc:Z:" + sub_cigar + "\tcl:i:" + std::to_string(subs.size());
            r += "\tci:i:" + std::to_string(ci) + "\tcn:i:" + std::to_string(chains.size());
            r += "\tAS:i:" + std::to_string(s.score) + "\tNM:i:" + std::to_string(s.num_edits);
            recs.push_back(r);
            sa.push_back(targets[tid].name + "," + std::to_string(pos) + "," + (fwd ? "+" : "-") + "," + full_cigar + "," +
                         std::to_string(mapq) + "," + std::to_string(s.num_edits));
        }
        if (!sa.empty()) { size_t k = primary % sa.size(); std::rotate(sa.begin(), sa.begin() + (sa.size() - k), sa.end()); }   // rotate_right (:956)
        std::string joined; for (size_t k = 0; k < sa.size(); ++k) { if (k) joined += ";"; joined += sa[k]; }
        for (auto& r : recs) out.push_back(r + "\tSA:Z:" + joined);
    }
    return true;
}

}  // namespace stitch
