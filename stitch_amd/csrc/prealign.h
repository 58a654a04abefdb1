// Banded pre-alignment filter (`stitch align --pre-align`, aligners/mod.rs:246-295, 556-604): the reference calls
// bio 1.1.0 `pairwise::banded::Aligner::custom_with_prehash(query, target, target_kmer_hash)`, a crate that is not part
// of the reference tree and that no reference test pins (parity unpinned).  This is the crate's published
// algorithm — exact k-mer matches, a sparse-DP "backbone" chain with gap penalty gap_open + d * gap_extend, a band of
// half-width w around the backbone, Smith-Waterman inside the band — with the free choices fixed as DESIGN.md "A15" states
// them.  The host finds the k-mer matches and the backbone and rasterises the band (this file); the banded DP itself runs
// on the GPU (prealign_kernel.hip).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace stitch {

// k-mers of all target strands: (hash, strand, position) sorted, so that one lookup per read position yields the seeds of
// every strand, each strand's in ascending position
struct KmerIndex {
    uint32_t k = 0;
    std::vector<uint64_t> key;
    std::vector<uint32_t> strand, pos;
    // accelerator for k-mers of A / C / G / T only (k <= 32): their 2-bit codes in an open-addressing table, each with its run
    // [t_lo, t_hi) of (c_strand, c_pos), ordered as above; a k-mer holding any other byte goes through `key`
    std::vector<uint64_t> t_code; std::vector<uint32_t> t_lo, t_hi; uint64_t t_mask = 0;
    std::vector<uint32_t> c_strand, c_pos;
};
struct Strand { uint64_t off; uint32_t len; };          // a target strand inside the context's contig buffer
KmerIndex build_kmer_index(const uint8_t* contigs, const std::vector<Strand>& strands, uint32_t k);

// exact k-mer matches of read q against every strand, ordered by (read start, target start); a strand stops collecting
// after MAX_MATCHES + 1 seeds (such a pair is scored over the full matrix)
struct Seed { uint32_t i, j; };
constexpr size_t MAX_MATCHES = 65536;
void find_seeds(const KmerIndex& ix, const uint8_t* contigs, const std::vector<Strand>& strands, const uint8_t* q, uint32_t m,
                std::vector<std::vector<Seed>>& seeds);

// rows [lo[c], hi[c]) of the DP matrix (rows 0..m = read prefix lengths) that belong to the band in column c = 0..n
// returns true when the band is the full matrix (no seeds, or more than MAX_MATCHES)
bool make_band(const std::vector<Seed>& seeds, uint32_t m, uint32_t n, uint32_t k, uint32_t w, int32_t match, int32_t gap_open,
               int32_t gap_extend, std::vector<uint16_t>& lo, std::vector<uint16_t>& hi);

// the two halves of make_band: the backbone (indexes into `seeds`, empty = the band is the full matrix: returns true), and the band
// around it
bool backbone_chain(const std::vector<Seed>& seeds, uint32_t k, int32_t match, int32_t gap_open, int32_t gap_extend, std::vector<uint32_t>& chain);
void rasterise_band(const std::vector<Seed>& seeds, const std::vector<uint32_t>& chain, uint32_t m, uint32_t n, uint32_t k, uint32_t w,
                    std::vector<uint16_t>& lo, std::vector<uint16_t>& hi);

// The band as the device draws it (prealign_band.hip): the backbone as a short list of pieces, each the union of the squares of
// half-width w around a line of points.
//   d <  0: the diagonal points (a + t, b + t), t = 0 .. c                      (a run of seeds, the extensions to the matrix edges)
//   d >= 0: the points (a + c * s / steps, b + d * s / steps), s = 1 .. steps - 1, steps = max(c, d)   (the gap between two runs)
struct BandElem { int32_t a, b, c, d; };
void band_elements(const std::vector<Seed>& seeds, const std::vector<uint32_t>& chain, uint32_t m, uint32_t n, uint32_t k, std::vector<BandElem>& out);

// one (read, target strand) pair of a banded launch; offsets are element offsets into the launch's device buffers
struct BandPair {
    uint32_t m, n;                 // query and target lengths
    uint64_t q_off, t_off;         // query bases (launch buffer), target bases (context's contig buffer)
    uint64_t band_off;             // uint16 lo[n+1] then hi[n+1]
    uint64_t state_off;            // int32 H[2][m+1], D[m+1]
    uint32_t elem_off, n_elem;     // the pair's band pieces (when the device draws the band)
};
// what the band kernel finds: which score kernel takes the pair
enum : uint32_t { BAND_CLASS_RING = 0, BAND_CLASS_TALL = 2, BAND_CLASS_WINDOW = 3 };
struct BandScoring { int32_t match, mismatch, gap_open, gap_extend; };
// ... with the four clip penalties of the clipping mode (0 = free, MIN_SCORE = forbidden; x = the read, rows; y = the target strand,
// columns: the argument order of bio's custom_with_prehash(query, target, ..), aligners/mod.rs:556-566, and Options::clipping, :123-141)
struct BandScoringClip { int32_t match, mismatch, gap_open, gap_extend, xclip_prefix, xclip_suffix, yclip_prefix, yclip_suffix; };
// one pair of the general launch (every clipping mode, reads of any length): 32-bit band ranges, 64-bit offsets
struct BandPair32 {
    uint32_t m, n;
    uint64_t q_off, t_off;         // query bases (launch buffer), target bases (context's contig buffer)
    uint64_t band_off;             // uint32 lo[n+1] then hi[n+1]
    uint64_t state_off;            // int32 H[2][m+1], D[m+1]
};
void rasterise_band32(const std::vector<Seed>& seeds, const std::vector<uint32_t>& chain, uint32_t m, uint32_t n, uint32_t k, uint32_t w,
                      std::vector<uint32_t>& lo, std::vector<uint32_t>& hi);

}  // namespace stitch
