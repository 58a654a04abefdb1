// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the `stitch align` hot path of fulcrumgenomics/stitch (reference @ 2025-03-21),
// written to follow the reference line by line (same data structures, same loop order, same
// tie-breaks, same quirks).  It exists to CHECK the HIP implementation in stitch_amd/; nothing in the
// product path may include, link or call it.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it.
//
// Parity status: PINNED by the reference's own known-answer tests (tests/golden/*.json, transcribed
// from single_contig_aligner.rs:915-1773, multi_contig_aligner.rs:465-737, alignment.rs:530-707,
// packed_length_cell.rs:193-259, aligners/mod.rs:984-1003).  The reference is Rust and cannot be
// compiled here (no cargo/rustc; crates not vendored), so there is no oracle/_ref build.
// UNPINNED parts (no reference test exists): SubAlignmentBuilder, SamRecordFormatter::format,
// realign_origin, traceback_all filtering.  They follow the reference text only.
//
// All file:line citations are relative to /root/reference/fg-stitch-lib/src/.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <set>
#include <optional>
#include <stdexcept>

namespace orc {

// align/aligners/constants.rs:7
constexpr int32_t MIN_SCORE = -858993459;

// align/traceback/mod.rs:47-57
enum : uint16_t {
    TB_START = 0, TB_INS = 1, TB_DEL = 2, TB_SUBST = 3, TB_MATCH = 4, TB_XCLIP_PREFIX = 5,
    TB_XCLIP_SUFFIX = 6, TB_YCLIP_PREFIX = 7, TB_YCLIP_SUFFIX = 8, TB_XJUMP = 9, TB_MAX = 9
};

struct SValue { uint16_t tb; uint32_t len; uint32_t idx; uint32_t from; };

// align/traceback/packed_length_cell.rs:25-182 — bit-for-bit the same packing
struct Cell {
    uint32_t s = 0, i = 0, d = 0, aux = 0;
    static uint32_t set_tb(uint32_t matrix, uint16_t tb);
    static uint32_t set_len(uint32_t matrix, uint32_t len);
    void set_idx(uint32_t idx);
    void set_from(uint32_t from);
    void set_i(uint16_t tb, uint32_t len);
    void set_d(uint16_t tb, uint32_t len);
    void set_s(uint16_t tb, uint32_t len);
    void set_s_all(uint16_t tb, uint32_t len, uint32_t idx, uint32_t from);
    void set_all(uint16_t tb, uint32_t len) { set_i(tb, len); set_d(tb, len); set_s(tb, len); }
    uint16_t get_i_tb() const { return (uint16_t)(i & 0xF); }
    uint16_t get_d_tb() const { return (uint16_t)(d & 0xF); }
    uint32_t get_i_len() const { return (i >> 4) & 0x7FFFFFF; }
    uint32_t get_d_len() const { return (d >> 4) & 0x7FFFFFF; }
    uint32_t get_s_len() const { return (s >> 4) & 0x7FFFFFF; }
    uint32_t get_idx() const;
    uint32_t get_from() const { return aux >> 5; }
    SValue get_s() const { return SValue{(uint16_t)(s & 0xF), get_s_len(), get_idx(), get_from()}; }
};

// align/traceback/mod.rs:76-127
struct Traceback {
    size_t rows = 0, cols = 0;
    std::vector<Cell> matrix;
    void init(size_t m, size_t n);
    // debug_assert!(i < self.rows); debug_assert!(j < self.cols) (:104-105, :111-112, :117-118): a release build of the
    // reference would silently read another row's cell here (or panic past the Vec's end), e.g. when the TB_XJUMP
    // quirk (:329-338) carries a row index into a shorter contig.  The oracle reports it instead of imitating it.
    void check(size_t i, size_t j) const { if (i >= rows || j >= cols) throw std::out_of_range("traceback index out of range (reference: debug_assert)"); }
    void set(size_t i, size_t j, const Cell& v) { check(i, j); matrix[i * cols + j] = v; }
    const Cell& get(size_t i, size_t j) const { check(i, j); return matrix[i * cols + j]; }
    Cell& get_mut(size_t i, size_t j) { check(i, j); return matrix[i * cols + j]; }
};

// align/scoring.rs:11-23 (match_fn == bio MatchParams: byte equality -> match / mismatch)
struct Scoring {
    int32_t gap_open = -5, gap_extend = -1;
    int32_t jump_score_same_contig_and_strand = -10;
    int32_t jump_score_same_contig_opposite_strand = -10;
    int32_t jump_score_inter_contig = -10;
    int32_t match_score = 1, mismatch_score = -1;
    int32_t xclip_prefix = MIN_SCORE, xclip_suffix = MIN_SCORE;
    int32_t yclip_prefix = MIN_SCORE, yclip_suffix = MIN_SCORE;
    int32_t score(uint8_t a, uint8_t b) const { return a == b ? match_score : mismatch_score; }
};

// align/aligners/mod.rs:56-62
struct JumpInfo { int32_t score = 0; uint32_t len = 0, idx = 0, from = 0; };

// align/aligners/constants.rs:20-29
enum OpKind : uint8_t { Match = 0, Subst = 1, Del = 2, Ins = 3, Xclip = 4, Yclip = 5, Xjump = 6, Yjump = 7 };
struct Op {
    OpKind kind; size_t a = 0; size_t b = 0;   // Xclip(a) Yclip(a) Xjump(a=contig, b=x) Yjump(a)
    bool operator==(const Op& o) const { return kind == o.kind && a == o.a && b == o.b; }
    bool operator!=(const Op& o) const { return !(*this == o); }
    bool is_special() const { return kind == Xclip || kind == Yclip || kind == Xjump; }
    std::string as_string(size_t contig_idx, size_t x_index) const;
    int32_t length_on_x(size_t x_index) const;
    size_t length_on_y() const;
};

// constants.rs:96-136
enum Mode : int { Local = 0, QueryLocal = 1, TargetLocal = 2, Global = 3, Custom = 4 };

// align/alignment.rs:16-51
struct Alignment {
    int32_t score = 0;
    size_t ystart = 0, xstart = 0, yend = 0, xend = 0, ylen = 0, xlen = 0;
    size_t start_contig_idx = 0, end_contig_idx = 0;
    std::vector<Op> operations;
    Mode mode = Custom;
    size_t length = 0;
    std::string cigar() const;                       // alignment.rs:105-149
    Alignment split_at_y(size_t y_pivot) const;      // alignment.rs:207-360
    bool validate(std::string* why) const;           // alignment.rs:56-103 (asserts -> false)
    std::optional<size_t> earliest_x_base_for(size_t contig_idx) const;   // :153-173
    std::optional<size_t> latest_x_base_for(size_t contig_idx) const;     // :177-200
};

// align/aligners/single_contig_aligner.rs:72-83
struct SingleContigAligner {
    std::vector<int32_t> I[2], D[2], S[2];
    std::vector<size_t> Lx, Ly;
    std::vector<int32_t> Sn;
    Traceback traceback;
    Scoring scoring;
    uint32_t contig_idx = 0;
    bool circular = false;

    void init_matrices(size_t m, size_t n);                           // :97-186
    void init_column(size_t j, size_t curr, size_t m, size_t n);      // :188-239
    JumpInfo get_jump_score_and_len(size_t m, size_t i, size_t j, size_t prev, int32_t addend,
                                    JumpInfo jump_info) const;        // :242-290
    void fill_column(const uint8_t* x, const uint8_t* y, size_t m, size_t n, size_t j, size_t prev,
                     size_t curr, JumpInfo jump_info);                // :292-451
    void fill_last_column_and_end_clipping(size_t m, size_t n);       // :453-555
    JumpInfo get_jump_info(size_t m, size_t j, int32_t jump_score) const;   // :677-697
    Alignment custom(const uint8_t* x, size_t m, const uint8_t* y, size_t n);   // :705-729
    Alignment with_mode(Mode mode, const uint8_t* x, size_t m, const uint8_t* y, size_t n);  // :733-872
};

// align/traceback/mod.rs:129-373
Alignment traceback(const std::vector<const SingleContigAligner*>& aligners, size_t n);
std::vector<Alignment> traceback_all(const std::vector<const SingleContigAligner*>& aligners, size_t n,
                                     const std::set<uint32_t>& contig_indexes_to_consider);
std::optional<Alignment> traceback_from(const std::vector<const SingleContigAligner*>& aligners, size_t n,
                                        uint32_t contig_index);

// align/aligners/multi_contig_aligner.rs:18-388
struct ContigAligner {
    std::string name; bool is_forward; SingleContigAligner aligner; std::vector<uint8_t> seq;
    size_t len() const { return seq.size(); }
};
struct MultiContigAligner {
    std::vector<ContigAligner> contigs;
    size_t len() const { return contigs.size(); }
    bool is_circular(size_t contig_idx) const { return contigs[contig_idx].aligner.circular; }
    std::optional<size_t> contig_index_for_strand(bool is_forward, const std::string& name) const;
    void add_contig(const std::string& name, bool is_forward, const uint8_t* seq, size_t len, bool circular,
                    const Scoring& scoring);
    Alignment custom_with_subset(const uint8_t* y, size_t n, const std::set<uint32_t>* contig_indexes);
    Alignment custom(const uint8_t* y, size_t n);
    std::vector<Alignment> traceback_all(size_t n, const std::set<uint32_t>* contig_indexes);
    std::optional<Alignment> traceback_from(size_t n, size_t contig_index);
    uint64_t cells_filled = 0;   // instrumentation only: sum of n * sum(m_c) over every custom() call
};

// util/dna.rs:5-41
std::vector<uint8_t> reverse_complement(const uint8_t* s, size_t n);

// util/target_seq.rs:15-36
struct TargetSeq { std::string name; std::vector<uint8_t> fwd, revcomp; bool circular = false; };

// align/aligners/mod.rs:65-168
struct Options {
    Mode mode = Local;
    int32_t match_score = 1, mismatch_score = -4, gap_open = -6, gap_extend = -2, default_jump_score = -10;
    std::optional<int32_t> jump_score_same_contig_and_strand, jump_score_same_contig_opposite_strand,
        jump_score_inter_contig;
    size_t kmer_size = 12, band_width = 50;
    bool double_strand = false, circular = false;
    size_t circular_slop = 20;
    bool pre_align = false; int32_t pre_align_min_score = 100; bool pre_align_subset_contigs = true;
    bool suboptimal = false; float suboptimal_pct = 20.0f;
    bool soft_clip = false, use_eq_and_x = false;
    int pick_primary = 0;   // 0 QueryLength (default), 1 Score — align/mod.rs PrimaryPickingStrategy
    bool filter_secondary = false; float filter_secondary_pct = 10.0f;
    void clipping(int32_t& xp, int32_t& xs, int32_t& yp, int32_t& ys) const;   // :123-131
    Scoring contig_scoring() const;                                            // :143-167
};

// align/sub_alignment.rs:10-19 (cigar kept as (kind char, len) runs)
struct CigarOp { char kind; size_t len; };
struct SubAlignment {
    size_t contig_idx = 0, query_start = 0, query_end = 0, target_start = 0, target_end = 0;
    std::vector<CigarOp> cigar; int32_t score = 0; int32_t num_edits = 0;
};
std::vector<SubAlignment> build_sub_alignments(const Alignment& chain, bool swap, const Scoring& scoring,
                                               bool use_eq_and_x);   // sub_alignment.rs:169-241

// align/aligners/mod.rs:227-553
struct Aligners {
    MultiContigAligner multi_contig;
    Options opts;
    static Aligners build(const Options& opts, const std::vector<TargetSeq>& target_seqs);   // :171-211
    std::vector<Alignment> align(const uint8_t* read, size_t n);                  // :237-340, pre_align=false
    std::vector<Alignment> align_subset(const uint8_t* q, size_t n, const std::set<uint32_t>* contigs_to_align);   // :289-337
    // :237-340 with pre_align (prealign_oracle.cpp: restated from bio's published description, parity unpinned)
    std::vector<Alignment> align_prealign(const uint8_t* read, size_t n, const std::vector<TargetSeq>& target_seqs,
                                          std::optional<int32_t>* prealign_score);
    Alignment remove_clipping(Alignment aln) const;                               // :343-353
    Alignment multi_contig_align(const uint8_t* q, size_t n, const std::set<uint32_t>* idx);   // :355-363
    Alignment realign_origin(const uint8_t* q, size_t n, Alignment alignment, size_t slop, bool all_contigs);  // :442-553
};

// align/aligners/mod.rs:622-973 rendered as SAM text lines (one per record, no trailing newline).
// Integer tags print as :i:, strings as :Z:.  Header lines are not produced here.
std::vector<std::string> format_sam(const Options& opts, const std::vector<TargetSeq>& target_seqs,
                                    const std::string& head, const std::vector<uint8_t>& bases,
                                    const std::vector<uint8_t>* quals, const std::vector<Alignment>& chains,
                                    std::optional<int32_t> pre_alignment_score, std::string* err);

// prealign_oracle.cpp: bio 1.1.0 pairwise::banded restated from its published description (parity unpinned)
int32_t banded_score(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, size_t w, int32_t match, int32_t mismatch, int32_t go, int32_t ge,
                     int32_t xclip_prefix, int32_t xclip_suffix, int32_t yclip_prefix, int32_t yclip_suffix);      // every clipping mode (x = query)
int32_t banded_local_score(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, size_t w, int32_t match,
                           int32_t mismatch, int32_t go, int32_t ge);
int banded_band(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, size_t w, int32_t match, int32_t go, int32_t ge,
                uint32_t* lo, uint32_t* hi);

}  // namespace orc
