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

// one (read, target strand) pair of a banded launch; offsets are element offsets into the launch's device buffers
struct BandPair {
    uint32_t m, n;                 // query and target lengths
    uint64_t q_off, t_off;         // query bases (launch buffer), target bases (context's contig buffer)
    uint64_t band_off;             // uint16 lo[n+1] then hi[n+1]
    uint64_t state_off;            // int32 H[2][m+1], D[m+1]
};
struct BandScoring { int32_t match, mismatch, gap_open, gap_extend; };

}  // namespace stitch
