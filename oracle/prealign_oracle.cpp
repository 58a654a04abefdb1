// TEST INFRASTRUCTURE ONLY (see stitch_oracle.hpp).  The banded pre-alignment filter of `stitch align --pre-align`
// (fg-stitch-lib/src/align/aligners/mod.rs:246-295, 556-604).
//
// PARITY UNPINNED.  The arithmetic lives in crate `bio 1.1.0` (Cargo.lock:91-92), which is not under /root/reference:
// `bio::alignment::pairwise::banded::Aligner::custom_with_prehash(query, target, target_kmer_hash)` with
// `bio::alignment::sparse::{hash_kmers, find_kmer_matches_seq2_hashed, sdpkpp}`.  No reference test touches it
// (SURVEY.md 8c), so this file restates the crate's PUBLISHED description, not its source:
//   "Use sparse dynamic programming to find a 'backbone' alignment from exact k-mer matches, then compute the SW alignment
//    in a 'band' surrounding the backbone, with a configurable width w" (pairwise::banded), and for the backbone
//   "Sparse DP routine generalizing LCSk++ method to penalize alignment gaps.  A gap is an unknown combination of mismatch,
//    insertion and deletions, and incurs a penalty of gap_open + d * gap_extend, where d is the distance along the
//    diagonal of the gap.  ... match_score is the per-base score for each kmer match" (sparse::sdpkpp).
// Where that description leaves freedom the choices below are this repository's (DESIGN.md "A15"); the product
// (stitch_amd/csrc/prealign.cpp + the banded kernel) implements the same definition independently.
//
// Definition (x = query, rows i = 1..m; y = target, columns j = 1..n; Local clipping only, the mode cfg3 uses):
//  1. matches (i, j), 0-based starts, x[i..i+k) == y[j..j+k), sorted by (i, j).  None, or more than MAX_MATCHES:
//     the band is the full matrix.
//  2. chain: dp[q] = best of  k*match  |  dp[c] + match for c = (qi-1, qj-1) if that is a match  |
//     dp[p] + k*match - pen(d) over earlier p with pi+k <= qi and pj+k <= qj, d = max(qi-pi-k, qj-pj-k),
//     pen(0) = 0, pen(d) = -gap_open - gap_extend*d.  Candidates are tried in that order, p ascending, and replace only
//     when strictly better.  The chain ends at the first q with the largest dp.
//  3. band: add(r, c) widens columns c-w..c+w to contain rows r-w..r+w (clamped to the matrix).  add() is applied to the
//     k+1 corner points (i+t, j+t) of every chain match, to the points of the straight line between the end of a match
//     and the start of the next (integer interpolation over max(di, dj) steps), and along the first match's diagonal
//     back to the matrix edge and the last match's diagonal forward to the edge.
//  4. score = max over in-band cells of H, the affine-gap Smith-Waterman recurrence with out-of-band cells = MIN_SCORE:
//     H = max(0, Hdiag + s(x,y), D, I), D = max(Dleft + ge, Hleft + go + ge), I = max(Iup + ge, Hup + go + ge),
//     row 0 and column 0 hold H = 0 (free prefix clipping).  Values are kept >= MIN_SCORE.
#include <algorithm>
#include <cstring>
#include <map>
#include <string>

#include "stitch_oracle.hpp"

namespace orc {

static const size_t MAX_MATCHES = 65536;

struct Band { std::vector<uint32_t> lo, hi; };      // per column 0..n: rows [lo, hi)

static void band_add(Band& b, size_t m, size_t n, size_t w, long r, long c) {
    const long c0 = std::max<long>(c - (long)w, 0), c1 = std::min<long>(c + (long)w, (long)n);
    const uint32_t r0 = (uint32_t)std::max<long>(r - (long)w, 0), r1 = (uint32_t)(std::min<long>(r + (long)w, (long)m) + 1);
    for (long cc = c0; cc <= c1; ++cc) { b.lo[cc] = std::min(b.lo[cc], r0); b.hi[cc] = std::max(b.hi[cc], r1); }
}

static Band make_band(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, size_t w, int32_t match, int32_t go, int32_t ge) {
    Band b; b.lo.assign(n + 1, (uint32_t)(m + 1)); b.hi.assign(n + 1, 0);
    std::vector<std::pair<uint32_t, uint32_t>> M;
    if (k > 0 && m >= k && n >= k) {
        std::map<std::string, std::vector<uint32_t>> hash;                     // hash_kmers(target, k)
        for (size_t j = 0; j + k <= n; ++j) hash[std::string((const char*)y + j, k)].push_back((uint32_t)j);
        for (size_t i = 0; i + k <= m && M.size() <= MAX_MATCHES; ++i) {
            auto it = hash.find(std::string((const char*)x + i, k));
            if (it != hash.end()) for (uint32_t j : it->second) M.push_back({(uint32_t)i, j});
        }
    }
    if (M.empty() || M.size() > MAX_MATCHES) { for (size_t c = 0; c <= n; ++c) { b.lo[c] = 0; b.hi[c] = (uint32_t)(m + 1); } return b; }
    const size_t Q = M.size();
    std::vector<int64_t> dp(Q); std::vector<long> prev(Q, -1);
    const int64_t base = (int64_t)k * match;
    for (size_t q = 0; q < Q; ++q) {
        dp[q] = base;
        if (M[q].first > 0 && M[q].second > 0) {
            auto it = std::lower_bound(M.begin(), M.begin() + q, std::make_pair(M[q].first - 1, M[q].second - 1));
            if (it != M.begin() + q && *it == std::make_pair(M[q].first - 1, M[q].second - 1)) {
                const size_t c = (size_t)(it - M.begin());
                if (dp[c] + match > dp[q]) { dp[q] = dp[c] + match; prev[q] = (long)c; }
            }
        }
        for (size_t p = 0; p < q; ++p) {
            if (M[p].first + k > M[q].first || M[p].second + k > M[q].second) continue;
            const int64_t di = (int64_t)M[q].first - M[p].first - (int64_t)k, dj = (int64_t)M[q].second - M[p].second - (int64_t)k, d = std::max(di, dj);
            const int64_t cand = dp[p] + base - (d > 0 ? -(int64_t)go - (int64_t)ge * d : 0);
            if (cand > dp[q]) { dp[q] = cand; prev[q] = (long)p; }
        }
    }
    size_t end = 0; for (size_t q = 1; q < Q; ++q) if (dp[q] > dp[end]) end = q;
    std::vector<size_t> path; for (long q = (long)end; q >= 0; q = prev[q]) path.push_back((size_t)q);
    std::reverse(path.begin(), path.end());
    for (size_t t = 0; t < path.size(); ++t) {
        const long qi = M[path[t]].first, qj = M[path[t]].second;
        for (size_t s = 0; s <= k; ++s) band_add(b, m, n, w, qi + (long)s, qj + (long)s);
        if (t + 1 < path.size()) {
            const long ai = qi + (long)k, aj = qj + (long)k, bi = M[path[t + 1]].first, bj = M[path[t + 1]].second;
            if (bi >= ai && bj >= aj) {                                            // (a continuation overlaps: nothing between)
                const long di = bi - ai, dj = bj - aj, steps = std::max(di, dj);
                for (long s = 1; s < steps; ++s) band_add(b, m, n, w, ai + di * s / steps, aj + dj * s / steps);
            }
        }
    }
    { const long i0 = M[path.front()].first, j0 = M[path.front()].second; for (long t = 1; t <= std::min(i0, j0); ++t) band_add(b, m, n, w, i0 - t, j0 - t); }
    { const long ie = M[path.back()].first + (long)k, je = M[path.back()].second + (long)k;
      for (long t = 1; t <= std::min((long)m - ie, (long)n - je); ++t) band_add(b, m, n, w, ie + t, je + t); }
    return b;
}

// the band alone (tests compare it with the product's host code): returns 1 when it is the full matrix
int banded_band(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, size_t w, int32_t match, int32_t go, int32_t ge,
                uint32_t* lo, uint32_t* hi) {
    const Band b = make_band(x, m, y, n, k, w, match, go, ge);
    bool full = true;
    for (size_t c = 0; c <= n; ++c) { lo[c] = b.lo[c]; hi[c] = b.hi[c]; full = full && b.lo[c] == 0 && b.hi[c] == m + 1; }
    return full ? 1 : 0;
}

// The banded score under bio's four clip penalties (0 = free, MIN_SCORE = forbidden; x = query, y = target: the argument order of
// custom_with_prehash(query, target, ..), aligners/mod.rs:556-566; Options::banded_scoring hands the mode's penalties over, :133-141).
// UNPINNED like everything in this file.  What is restated is the documented meaning of the penalties ("xclip_prefix: the penalty for
// clipping a prefix of x", likewise the other three) on the band of the local filter.  Where that is silent this repository chose:
//   * row 0 and column 0 always belong to the band: S(0, j) = max(yclip_prefix, go + ge j), S(i, 0) = max(xclip_prefix, go + ge i),
//     S(0, 0) = 0 (the skipped prefix is clipped or gapped);
//   * a cell may start the alignment: xclip_prefix + S(0, j), or yclip_prefix + go + ge i;
//   * the score is the best in-band cell plus what the rest of both sequences costs there, max(xclip_suffix, go + ge (m - i)) for i < m
//     and max(yclip_suffix, go + ge (n - j)) for j < n — or the empty alignment, max(xp, xs, go + ge m) + max(yp, ys, go + ge n).
// With all four penalties 0 this is step 4 of the definition at the top of the file, cell for cell (the product's fast Local kernels
// compute that; tests/test_prealign.py compares the two forms).
int32_t banded_score(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, size_t w, int32_t match, int32_t mismatch, int32_t go, int32_t ge,
                     int32_t xp, int32_t xs, int32_t yp, int32_t ys) {
    const Band b = make_band(x, m, y, n, k, w, match, go, ge);
    auto inband = [&](size_t i, size_t j) { return i >= b.lo[j] && i < b.hi[j]; };
    auto floor_min = [](int64_t v) { return (int64_t)std::max<int64_t>(v, MIN_SCORE); };
    auto row0 = [&](size_t j) -> int64_t { return j == 0 ? 0 : std::max<int64_t>(yp, (int64_t)go + (int64_t)ge * (int64_t)j); };
    auto col0 = [&](size_t i) -> int64_t { return i == 0 ? 0 : std::max<int64_t>(xp, (int64_t)go + (int64_t)ge * (int64_t)i); };
    std::vector<int64_t> H[2], D[2], I(m + 1);
    for (int t = 0; t < 2; ++t) { H[t].assign(m + 1, MIN_SCORE); D[t].assign(m + 1, MIN_SCORE); }
    int64_t best = std::max<int64_t>(std::max(xp, xs), m ? (int64_t)go + (int64_t)ge * (int64_t)m : 0) + std::max<int64_t>(std::max(yp, ys), n ? (int64_t)go + (int64_t)ge * (int64_t)n : 0);
    for (size_t j = 1; j <= n; ++j) {
        const int cur = (int)(j & 1), prv = 1 - cur;
        const int64_t ry = j == n ? 0 : std::max<int64_t>(ys, (int64_t)go + (int64_t)ge * (int64_t)(n - j));
        for (size_t i = 1; i <= m; ++i) {
            if (!inband(i, j)) { H[cur][i] = MIN_SCORE; D[cur][i] = MIN_SCORE; I[i] = MIN_SCORE; continue; }
            const int64_t hd = j == 1 ? col0(i - 1) : i == 1 ? row0(j - 1) : (inband(i - 1, j - 1) ? H[prv][i - 1] : MIN_SCORE);
            const int64_t hl = j == 1 ? col0(i) : (inband(i, j - 1) ? H[prv][i] : MIN_SCORE), dl = (j == 1 || !inband(i, j - 1)) ? MIN_SCORE : D[prv][i];
            const int64_t hu = i == 1 ? row0(j) : (inband(i - 1, j) ? H[cur][i - 1] : MIN_SCORE), iu = (i == 1 || !inband(i - 1, j)) ? MIN_SCORE : I[i - 1];
            const int64_t d = floor_min(std::max(dl + ge, hl + go + ge));
            const int64_t in = floor_min(std::max(iu + ge, hu + go + ge));
            const int32_t s = x[i - 1] == y[j - 1] ? match : mismatch;
            int64_t h = floor_min(hd + s);
            h = std::max(h, d); h = std::max(h, in);
            h = std::max(h, floor_min((int64_t)xp + row0(j)));
            h = std::max(h, floor_min((int64_t)yp + go + (int64_t)ge * (int64_t)i));
            H[cur][i] = h; D[cur][i] = d; I[i] = in;
            const int64_t rx = i == m ? 0 : std::max<int64_t>(xs, (int64_t)go + (int64_t)ge * (int64_t)(m - i));
            if (h > MIN_SCORE) best = std::max(best, h + rx + ry);
        }
    }
    return (int32_t)std::max<int64_t>(best, MIN_SCORE);
}
int32_t banded_local_score(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, size_t w, int32_t match,
                           int32_t mismatch, int32_t go, int32_t ge) {
    return banded_score(x, m, y, n, k, w, match, mismatch, go, ge, 0, 0, 0, 0);
}

// Aligners::align with opts.pre_align (mod.rs:237-340): the banded score of every target strand decides which contigs
// take part; returns the chains and the largest kept score (None when nothing passed).
std::vector<Alignment> Aligners::align_prealign(const uint8_t* read, size_t n, const std::vector<TargetSeq>& target_seqs,
                                                std::optional<int32_t>* prealign_score) {
    // Options::clipping (mod.rs:123-131) handed to the banded scorer (:133-141), whose x is the query
    int32_t xp = 0, xs = 0, yp = 0, ys = 0;
    if (opts.mode == QueryLocal || opts.mode == Global) { xp = MIN_SCORE; xs = MIN_SCORE; }
    if (opts.mode == TargetLocal || opts.mode == Global) { yp = MIN_SCORE; ys = MIN_SCORE; }
    std::vector<uint8_t> query(read, read + n);
    for (auto& b : query) if (b >= 'a' && b <= 'z') b = (uint8_t)(b - 32);
    std::map<uint32_t, int32_t> kept;
    for (const TargetSeq& t : target_seqs) {                                     // prealign_local_banded, :575-604
        const int32_t f = banded_score(query.data(), n, t.fwd.data(), t.fwd.size(), opts.kmer_size, opts.band_width, opts.match_score,
                                       opts.mismatch_score, opts.gap_open, opts.gap_extend, xp, xs, yp, ys);
        if (f >= opts.pre_align_min_score) kept[(uint32_t)*multi_contig.contig_index_for_strand(true, t.name)] = f;
        if (opts.double_strand) {
            const int32_t r = banded_score(query.data(), n, t.revcomp.data(), t.revcomp.size(), opts.kmer_size, opts.band_width,
                                           opts.match_score, opts.mismatch_score, opts.gap_open, opts.gap_extend, xp, xs, yp, ys);
            if (r >= opts.pre_align_min_score) kept[(uint32_t)*multi_contig.contig_index_for_strand(false, t.name)] = r;
        }
        if (!opts.pre_align_subset_contigs && !kept.empty()) break;              // :276-278
    }
    *prealign_score = std::nullopt;
    if (kept.empty()) return {};
    std::set<uint32_t> subset; for (auto& kv : kept) subset.insert(kv.first);
    const std::set<uint32_t>* contigs_to_align = opts.pre_align_subset_contigs ? &subset : nullptr;
    std::vector<Alignment> alignments = align_subset(query.data(), n, contigs_to_align);
    int32_t mx = kept.begin()->second; for (auto& kv : kept) mx = std::max(mx, kv.second);
    *prealign_score = mx;
    return alignments;
}

}  // namespace orc
