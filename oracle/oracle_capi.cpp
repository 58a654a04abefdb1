// ORACLE — TEST INFRASTRUCTURE ONLY (see stitch_oracle.hpp).  Flat C API over the restatement so
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive it through ctypes.
//
// Alignment wire format (int64 array): [score, xstart, xend, ystart, yend, xlen, ylen, start_contig_idx,
// end_contig_idx, length, mode, nops, then nops x (kind, a, b)].
#include "stitch_oracle.hpp"

#include <chrono>
#include <cstring>
#include <thread>
#include <atomic>

using namespace orc;

static thread_local std::string g_err;

static size_t put_alignment(const Alignment& a, int64_t* out, size_t cap) {
    size_t need = 12 + 3 * a.operations.size();
    if (need > cap) return need;
    out[0] = a.score; out[1] = (int64_t)a.xstart; out[2] = (int64_t)a.xend; out[3] = (int64_t)a.ystart;
    out[4] = (int64_t)a.yend; out[5] = (int64_t)a.xlen; out[6] = (int64_t)a.ylen;
    out[7] = (int64_t)a.start_contig_idx; out[8] = (int64_t)a.end_contig_idx; out[9] = (int64_t)a.length;
    out[10] = (int64_t)a.mode; out[11] = (int64_t)a.operations.size();
    for (size_t k = 0; k < a.operations.size(); ++k) {
        out[12 + 3 * k] = a.operations[k].kind; out[13 + 3 * k] = (int64_t)a.operations[k].a; out[14 + 3 * k] = (int64_t)a.operations[k].b;
    }
    return need;
}
static Alignment get_alignment(const int64_t* in) {
    Alignment a;
    a.score = (int32_t)in[0]; a.xstart = (size_t)in[1]; a.xend = (size_t)in[2]; a.ystart = (size_t)in[3]; a.yend = (size_t)in[4];
    a.xlen = (size_t)in[5]; a.ylen = (size_t)in[6]; a.start_contig_idx = (size_t)in[7]; a.end_contig_idx = (size_t)in[8];
    a.length = (size_t)in[9]; a.mode = (Mode)in[10];
    size_t nops = (size_t)in[11];
    for (size_t k = 0; k < nops; ++k) a.operations.push_back(Op{(OpKind)in[12 + 3 * k], (size_t)in[13 + 3 * k], (size_t)in[14 + 3 * k]});
    return a;
}
// scoring wire format: [match, mismatch, gap_open, gap_extend, jump_same, jump_opp, jump_inter, xclip_prefix,
// xclip_suffix, yclip_prefix, yclip_suffix]
static Scoring get_scoring(const int32_t* s) {
    Scoring sc;
    sc.match_score = s[0]; sc.mismatch_score = s[1]; sc.gap_open = s[2]; sc.gap_extend = s[3];
    sc.jump_score_same_contig_and_strand = s[4]; sc.jump_score_same_contig_opposite_strand = s[5]; sc.jump_score_inter_contig = s[6];
    sc.xclip_prefix = s[7]; sc.xclip_suffix = s[8]; sc.yclip_prefix = s[9]; sc.yclip_suffix = s[10];
    return sc;
}
// options wire format (int32[24] + float[2]): see oracle/oracle.py
static Options get_options(const int32_t* o, const float* f) {
    Options op;
    op.mode = (Mode)o[0]; op.match_score = o[1]; op.mismatch_score = o[2]; op.gap_open = o[3]; op.gap_extend = o[4];
    op.default_jump_score = o[5];
    if (o[6]) op.jump_score_same_contig_and_strand = o[7];
    if (o[8]) op.jump_score_same_contig_opposite_strand = o[9];
    if (o[10]) op.jump_score_inter_contig = o[11];
    op.double_strand = o[12]; op.circular = o[13]; op.circular_slop = (size_t)o[14];
    op.suboptimal = o[15]; op.soft_clip = o[16]; op.use_eq_and_x = o[17]; op.pick_primary = o[18]; op.filter_secondary = o[19];
    op.suboptimal_pct = f[0]; op.filter_secondary_pct = f[1];
    op.pre_align = o[20]; op.pre_align_min_score = o[20] ? o[21] : 100; op.pre_align_subset_contigs = o[20] ? o[22] != 0 : true;
    if (o[20]) { op.kmer_size = (size_t)o[23]; op.band_width = (size_t)f[2]; }
    return op;
}

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// SingleContigAligner::{local,querylocal,targetlocal,global,custom} — single_contig_aligner.rs:705-872
long orc_single(int mode, const int32_t* scoring, int circular, const uint8_t* x, size_t m, const uint8_t* y, size_t n,
                int64_t* out, size_t cap) {
    try {
        SingleContigAligner a; a.scoring = get_scoring(scoring); a.circular = circular != 0;
        Alignment r = a.with_mode((Mode)mode, x, m, y, n);
        return (long)put_alignment(r, out, cap);
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

void* orc_mc_new() { return new MultiContigAligner(); }
void orc_mc_free(void* h) { delete (MultiContigAligner*)h; }
int orc_mc_add(void* h, const char* name, int is_forward, const uint8_t* seq, size_t len, int circular, const int32_t* scoring) {
    try { ((MultiContigAligner*)h)->add_contig(name, is_forward != 0, seq, len, circular != 0, get_scoring(scoring)); return 0; }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}
void orc_mc_set_scoring(void* h, const int32_t* scoring) {
    for (auto& c : ((MultiContigAligner*)h)->contigs) c.aligner.scoring = get_scoring(scoring);
}
long orc_mc_custom(void* h, const uint8_t* y, size_t n, const uint32_t* subset, size_t nsub, int64_t* out, size_t cap) {
    try {
        std::set<uint32_t> s(subset, subset + nsub);
        Alignment r = ((MultiContigAligner*)h)->custom_with_subset(y, n, subset ? &s : nullptr);
        return (long)put_alignment(r, out, cap);
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// returns number of chains; chain k is fetched with orc_mc_chain
static thread_local std::vector<Alignment> g_chains;
long orc_mc_traceback_all(void* h, size_t n, const uint32_t* subset, size_t nsub) {
    try {
        std::set<uint32_t> s(subset, subset + nsub);
        g_chains = ((MultiContigAligner*)h)->traceback_all(n, subset ? &s : nullptr);
        return (long)g_chains.size();
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
long orc_mc_traceback_from(void* h, size_t n, size_t contig_index, int64_t* out, size_t cap) {
    try {
        auto r = ((MultiContigAligner*)h)->traceback_from(n, contig_index);
        if (!r) return 0;
        return (long)put_alignment(*r, out, cap);
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
long orc_chain(size_t k, int64_t* out, size_t cap) { return k < g_chains.size() ? (long)put_alignment(g_chains[k], out, cap) : -1; }

// Alignment methods — alignment.rs
long orc_aln_cigar(const int64_t* in, char* buf, size_t cap) {
    std::string c = get_alignment(in).cigar();
    if (c.size() + 1 <= cap) memcpy(buf, c.c_str(), c.size() + 1);
    return (long)c.size();
}
long orc_aln_split_at_y(const int64_t* in, size_t y_pivot, int64_t* out, size_t cap) {
    try { return (long)put_alignment(get_alignment(in).split_at_y(y_pivot), out, cap); }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int orc_aln_validate(const int64_t* in) { std::string why; bool ok = get_alignment(in).validate(&why); g_err = why; return ok ? 1 : 0; }
long orc_aln_earliest_x(const int64_t* in, size_t contig) { auto r = get_alignment(in).earliest_x_base_for(contig); return r ? (long)*r : -1; }
long orc_aln_latest_x(const int64_t* in, size_t contig) { auto r = get_alignment(in).latest_x_base_for(contig); return r ? (long)*r : -1; }

// PackedLengthCell — packed_length_cell.rs tests :193-259.  op: 0 set_i 1 set_d 2 set_s 3 set_s_all;
// out = [i_tb,i_len,d_tb,d_len,s_tb,s_len,s_idx,s_from]
int orc_cell_apply(uint32_t* cell4, int op, uint32_t tb, uint32_t len, uint32_t idx, uint32_t from, uint32_t* out8) {
    try {
        Cell c; c.s = cell4[0]; c.i = cell4[1]; c.d = cell4[2]; c.aux = cell4[3];
        if (op == 0) c.set_i((uint16_t)tb, len); else if (op == 1) c.set_d((uint16_t)tb, len);
        else if (op == 2) c.set_s((uint16_t)tb, len); else if (op == 3) c.set_s_all((uint16_t)tb, len, idx, from);
        cell4[0] = c.s; cell4[1] = c.i; cell4[2] = c.d; cell4[3] = c.aux;
        SValue s = c.get_s();
        out8[0] = c.get_i_tb(); out8[1] = c.get_i_len(); out8[2] = c.get_d_tb(); out8[3] = c.get_d_len();
        out8[4] = s.tb; out8[5] = s.len; out8[6] = s.idx; out8[7] = s.from;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// Aligners — aligners/mod.rs:171-553 (pre_align not restated)
struct OrcAligners { Aligners al; std::vector<TargetSeq> targets; std::vector<Alignment> chains; std::optional<int32_t> prealign; };
void* orc_al_new(const int32_t* opts, const float* fopts, size_t n_targets, const char* const* names,
                 const uint8_t* const* seqs, const size_t* lens) {
    try {
        auto* h = new OrcAligners();
        Options o = get_options(opts, fopts);
        for (size_t k = 0; k < n_targets; ++k) {
            TargetSeq t; t.name = names[k]; t.fwd.assign(seqs[k], seqs[k] + lens[k]);
            for (auto& b : t.fwd) if (b >= 'a' && b <= 'z') b = (uint8_t)(b - 32);    // target_seq.rs:111-115
            t.revcomp = reverse_complement(t.fwd.data(), t.fwd.size()); t.circular = o.circular;
            h->targets.push_back(std::move(t));
        }
        h->al = Aligners::build(o, h->targets);
        return h;
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void orc_al_free(void* h) { delete (OrcAligners*)h; }
long orc_al_align(void* hh, const uint8_t* read, size_t n) {
    auto* h = (OrcAligners*)hh;
    try {
        h->prealign.reset();
        if (h->al.opts.pre_align) h->chains = h->al.align_prealign(read, n, h->targets, &h->prealign);
        else h->chains = h->al.align(read, n);
        return (long)h->chains.size();
    }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// pre-alignment score of the last orc_al_align (the xs tag): returns 1 and sets *score, or 0 when there is none
int orc_al_prealign(void* hh, int32_t* score) { auto* h = (OrcAligners*)hh; if (!h->prealign) return 0; *score = *h->prealign; return 1; }
long orc_banded_local_score(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, size_t w, int32_t match, int32_t mismatch, int32_t go, int32_t ge) {
    try { return banded_local_score(x, m, y, n, k, w, match, mismatch, go, ge); } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
long orc_banded_score(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, size_t w, int32_t match, int32_t mismatch, int32_t go, int32_t ge,
                      int32_t xp, int32_t xs, int32_t yp, int32_t ys) {
    try { return banded_score(x, m, y, n, k, w, match, mismatch, go, ge, xp, xs, yp, ys); } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int orc_banded_band(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, size_t w, int32_t match, int32_t go, int32_t ge, uint32_t* lo, uint32_t* hi) {
    try { return banded_band(x, m, y, n, k, w, match, go, ge, lo, hi); } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
long orc_al_chain(void* hh, size_t k, int64_t* out, size_t cap) {
    auto* h = (OrcAligners*)hh;
    return k < h->chains.size() ? (long)put_alignment(h->chains[k], out, cap) : -1;
}
// Test hook: replace the chains of the last orc_al_align with caller-supplied ones (k == 0 clears first), so that
// orc_al_format_sam can be checked against hand-traced known answers (tests/golden/sam_known_answers.json).
int orc_al_set_chain(void* hh, size_t k, const int64_t* in) {
    auto* h = (OrcAligners*)hh;
    if (k == 0) h->chains.clear();
    h->chains.push_back(get_alignment(in));
    return (int)h->chains.size();
}
uint64_t orc_al_cells(void* hh) { return ((OrcAligners*)hh)->al.multi_contig.cells_filled; }
// SAM text of the chains of the last orc_al_align call; records joined with '\n'.  has_prealign: xs source.
long orc_al_format_sam(void* hh, const char* head, const uint8_t* bases, size_t n, const uint8_t* quals,
                       int has_prealign, int32_t prealign, char* buf, size_t cap) {
    auto* h = (OrcAligners*)hh;
    try {
        std::vector<uint8_t> b(bases, bases + n); std::vector<uint8_t> q; if (quals) q.assign(quals, quals + n);
        std::string err;
        auto recs = format_sam(h->al.opts, h->targets, head, b, quals ? &q : nullptr, h->chains,
                               has_prealign ? std::optional<int32_t>(prealign) : std::nullopt, &err);
        if (recs.empty()) { g_err = err; return -1; }
        std::string all;
        for (size_t k = 0; k < recs.size(); ++k) { if (k) all += "\n"; all += recs[k]; }
        if (all.size() + 1 <= cap) memcpy(buf, all.c_str(), all.size() + 1);
        return (long)all.size();
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// CPU baseline leg (bench.py only): the worker model of fg-stitch-cli/src/commands/align.rs:345-390 — `threads` workers, each
// owning its own aligner set (one Aligners per thread; its traceback matrices are allocated once and re-initialised per read,
// align/traceback/mod.rs:93-126), pulling chunks of `chunk` records (the reference: 10, align/io.rs:178-245; a bounded sample
// uses 1 so that the workers stay balanced).
//
// WARM measurement: every worker first builds its aligner set and aligns `warm` read(s) of the sample outside the clock (the
// first touch of its matrices: page faults and mm-lock contention are a property of a fresh process, not of the aligner in
// steady state); the clock starts when all workers are ready and stops when the last one has finished its last read — before
// any aligner set is freed.  Returns wall seconds; cells_out = DP cells filled inside the clock; busy_secs[t] (optional) =
// each worker's own time inside the clock; scores_out[r] = score of the first chain.  SAM text (records '\n'-terminated, header
// "read_%07zu" numbered from name_base, qualities all 'I') goes to sam_buf at sam_offsets[r]..sam_offsets[r+1] AFTER the clock
// has stopped (formatting runs on the reference's single writer thread and is not part of the aligners' time).
double orc_bench_warm(const int32_t* opts, const float* fopts, size_t n_targets, const char* const* names,
                      const uint8_t* const* seqs, const size_t* lens, const uint8_t* reads, const uint64_t* offsets,
                      size_t n_reads, int threads, int chunk, int warm, uint64_t* cells_out, int64_t* scores_out, double* busy_secs,
                      size_t name_base, char* sam_buf, size_t sam_cap, uint64_t* sam_offsets) {
    if (threads < 1) threads = 1;
    if (chunk < 1) chunk = 1;
    std::atomic<size_t> next{0};
    std::atomic<uint64_t> cells{0};
    std::atomic<int> ready{0}, done{0};
    std::atomic<bool> go{false}, release{false};
    std::vector<std::vector<Alignment>> all_chains(n_reads);
    Options o = get_options(opts, fopts);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) {
        pool.emplace_back([&, t]() {
            auto* h = (OrcAligners*)orc_al_new(opts, fopts, n_targets, names, seqs, lens);
            for (int w = 0; w < warm && n_reads > 0; ++w) {
                const size_t r = ((size_t)t + (size_t)w * (size_t)threads) % n_reads;
                (void)h->al.align(reads + offsets[r], (size_t)(offsets[r + 1] - offsets[r]));
            }
            const uint64_t cells_warm = h->al.multi_contig.cells_filled;
            ready.fetch_add(1);
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            auto b0 = std::chrono::steady_clock::now();
            for (;;) {
                const size_t r0 = next.fetch_add((size_t)chunk);
                if (r0 >= n_reads) break;
                for (size_t r = r0; r < std::min(n_reads, r0 + (size_t)chunk); ++r) {
                    h->chains = h->al.align(reads + offsets[r], (size_t)(offsets[r + 1] - offsets[r]));
                    if (scores_out) scores_out[r] = h->chains.empty() ? 0 : h->chains[0].score;
                    if (sam_buf) all_chains[r] = h->chains;
                }
            }
            if (busy_secs) busy_secs[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - b0).count();
            cells += h->al.multi_contig.cells_filled - cells_warm;
            done.fetch_add(1);
            while (!release.load(std::memory_order_acquire)) std::this_thread::yield();      // (freeing 16-byte matrices is not aligner time)
            delete h;
        });
    }
    while (ready.load() < threads) std::this_thread::yield();
    auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    while (done.load() < threads) std::this_thread::sleep_for(std::chrono::microseconds(200));
    auto t1 = std::chrono::steady_clock::now();
    release.store(true, std::memory_order_release);
    for (auto& th : pool) th.join();
    if (cells_out) *cells_out = cells.load();
    if (sam_buf && sam_offsets) {
        auto* h = (OrcAligners*)orc_al_new(opts, fopts, n_targets, names, seqs, lens);
        size_t at = 0;
        for (size_t r = 0; r < n_reads; ++r) {
            sam_offsets[r] = at;
            const size_t n = (size_t)(offsets[r + 1] - offsets[r]);
            std::vector<uint8_t> b(reads + offsets[r], reads + offsets[r] + n), q(n, (uint8_t)'I');
            char head[64]; snprintf(head, sizeof(head), "read_%07zu", name_base + r);
            std::string err;
            auto recs = format_sam(o, h->targets, head, b, &q, all_chains[r], std::nullopt, &err);
            for (size_t k = 0; k < recs.size(); ++k) {
                const std::string line = recs[k] + "\n";
                if (at + line.size() <= sam_cap) memcpy(sam_buf + at, line.data(), line.size());
                at += line.size();
            }
        }
        sam_offsets[n_reads] = at;
        delete h;
    }
    return std::chrono::duration<double>(t1 - t0).count();
}
// (kept for callers that only want a number: warm, one record per pull)
double orc_bench(const int32_t* opts, const float* fopts, size_t n_targets, const char* const* names,
                 const uint8_t* const* seqs, const size_t* lens, const uint8_t* reads, const uint64_t* offsets,
                 size_t n_reads, int threads, uint64_t* cells_out, int64_t* scores_out) {
    return orc_bench_warm(opts, fopts, n_targets, names, seqs, lens, reads, offsets, n_reads, threads, 1, 1, cells_out, scores_out, nullptr, 0, nullptr, 0, nullptr);
}

}  // extern "C"
