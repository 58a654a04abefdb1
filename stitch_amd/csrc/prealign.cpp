// Host side of the banded pre-alignment filter: k-mer matches, backbone chain, band (see prealign.h for the provenance).
#include "prealign.h"

#include <algorithm>
#include <cstring>
#include <numeric>

namespace stitch {

namespace {
inline int base2(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
inline uint64_t mix2(uint64_t x) { x ^= x >> 29; x *= 0x9E3779B97F4A7C15ull; x ^= x >> 32; return x; }
inline uint64_t kmer_hash(const uint8_t* p, uint32_t k) {      // FNV-1a; equality is always re-checked on the bytes
    uint64_t h = 1469598103934665603ull;
    for (uint32_t a = 0; a < k; ++a) { h ^= p[a]; h *= 1099511628211ull; }
    return h;
}
}  // namespace

KmerIndex build_kmer_index(const uint8_t* contigs, const std::vector<Strand>& strands, uint32_t k) {
    KmerIndex ix; ix.k = k;
    if (k == 0) return ix;
    struct E { uint64_t h; uint32_t s, p; };
    std::vector<E> es;
    for (uint32_t s = 0; s < strands.size(); ++s)
        for (uint32_t j = 0; j + k <= strands[s].len; ++j) es.push_back(E{kmer_hash(contigs + strands[s].off + j, k), s, j});
    std::sort(es.begin(), es.end(), [](const E& a, const E& b) { return a.h != b.h ? a.h < b.h : a.s != b.s ? a.s < b.s : a.p < b.p; });
    ix.key.resize(es.size()); ix.strand.resize(es.size()); ix.pos.resize(es.size());
    for (size_t a = 0; a < es.size(); ++a) { ix.key[a] = es[a].h; ix.strand[a] = es[a].s; ix.pos[a] = es[a].p; }
    if (k <= 32) {                                         // the 2-bit accelerator
        std::vector<E> cs;
        for (uint32_t s = 0; s < strands.size(); ++s) {
            uint64_t code = 0; uint32_t good = 0;
            const uint64_t mask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
            for (uint32_t j = 0; j < strands[s].len; ++j) {
                const int b = base2(contigs[strands[s].off + j]);
                if (b < 0) { good = 0; code = 0; continue; }
                code = ((code << 2) | (uint64_t)b) & mask; ++good;
                if (good >= k) cs.push_back(E{code, s, j + 1 - k});
            }
        }
        std::sort(cs.begin(), cs.end(), [](const E& a, const E& b) { return a.h != b.h ? a.h < b.h : a.s != b.s ? a.s < b.s : a.p < b.p; });
        size_t distinct = 0;
        for (size_t a = 0; a < cs.size(); ++a) if (a == 0 || cs[a].h != cs[a - 1].h) ++distinct;
        size_t cap = 16; while (cap < 2 * distinct + 2) cap <<= 1;
        ix.t_mask = cap - 1; ix.t_code.assign(cap, 0); ix.t_lo.assign(cap, 0); ix.t_hi.assign(cap, 0);      // an empty slot has t_lo == t_hi
        ix.c_strand.resize(cs.size()); ix.c_pos.resize(cs.size());
        for (size_t a = 0; a < cs.size();) {
            size_t b = a; while (b < cs.size() && cs[b].h == cs[a].h) ++b;
            uint64_t slot = mix2(cs[a].h) & ix.t_mask;
            while (ix.t_lo[slot] != ix.t_hi[slot]) slot = (slot + 1) & ix.t_mask;
            ix.t_code[slot] = cs[a].h; ix.t_lo[slot] = (uint32_t)a; ix.t_hi[slot] = (uint32_t)b;
            for (size_t e = a; e < b; ++e) { ix.c_strand[e] = cs[e].s; ix.c_pos[e] = cs[e].p; }
            a = b;
        }
    }
    return ix;
}

void find_seeds(const KmerIndex& ix, const uint8_t* contigs, const std::vector<Strand>& strands, const uint8_t* q, uint32_t m,
                std::vector<std::vector<Seed>>& seeds) {
    seeds.assign(strands.size(), {});
    const uint32_t k = ix.k;
    if (k == 0 || m < k || ix.key.empty()) return;
    const bool fast = ix.t_mask != 0;
    const uint64_t mask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    uint64_t code = 0; uint32_t good = 0;                  // 2-bit code of the last `good` (capped at k) bases, all of them A / C / G / T
    for (uint32_t a = 0; fast && a + 1 < k && a < m; ++a) { const int b = base2(q[a]); if (b < 0) { good = 0; code = 0; } else { code = ((code << 2) | (uint64_t)b) & mask; ++good; } }
    for (uint32_t i = 0; i + k <= m; ++i) {
        if (fast) {
            const int b = base2(q[i + k - 1]);
            if (b < 0) { good = 0; code = 0; } else { code = ((code << 2) | (uint64_t)b) & mask; if (good < k) ++good; }
            if (good >= k) {                                // (a k-mer of four-letter bases only matches k-mers of four-letter bases: all of them are in the table)
                for (uint64_t slot = mix2(code) & ix.t_mask; ix.t_lo[slot] != ix.t_hi[slot]; slot = (slot + 1) & ix.t_mask) {
                    if (ix.t_code[slot] != code) continue;
                    for (uint32_t e = ix.t_lo[slot]; e < ix.t_hi[slot]; ++e) { const uint32_t s = ix.c_strand[e]; if (seeds[s].size() <= MAX_MATCHES) seeds[s].push_back(Seed{i, ix.c_pos[e]}); }
                    break;
                }
                continue;
            }
        }
        const uint64_t h = kmer_hash(q + i, k);
        for (auto a = std::lower_bound(ix.key.begin(), ix.key.end(), h); a != ix.key.end() && *a == h; ++a) {
            const size_t e = (size_t)(a - ix.key.begin());
            const uint32_t s = ix.strand[e], j = ix.pos[e];
            if (seeds[s].size() <= MAX_MATCHES && memcmp(q + i, contigs + strands[s].off + j, k) == 0) seeds[s].push_back(Seed{i, j});
        }
    }
}

bool backbone_chain(const std::vector<Seed>& seeds, uint32_t k, int32_t match, int32_t gap_open, int32_t gap_extend, std::vector<uint32_t>& chain) {
    chain.clear();
    // 1. no seeds, or too many: the band is the full matrix
    if (seeds.empty() || seeds.size() > MAX_MATCHES) return true;
    // 2. backbone: best-scoring chain of seeds (k * match per seed, + match for a seed that continues the one a step up the
    //    diagonal, gap penalty -gap_open - gap_extend * d for d = max(query gap, target gap) > 0); first best wins
    const size_t Q = seeds.size();
    std::vector<long long> best(Q), prefmax(Q); std::vector<int> from(Q, -1);
    const long long seed_score = (long long)k * match;
    size_t hi_i = 0;
    for (size_t b = 0; b < Q; ++b) {
        const Seed sb = seeds[b];
        long long v = seed_score; int f = -1;
        if (sb.i > 0 && sb.j > 0) {                        // continuation: binary search among the earlier seeds
            size_t lo_ = 0, hi_ = b;
            while (lo_ < hi_) { const size_t mid = (lo_ + hi_) / 2; const Seed s = seeds[mid]; if (s.i < sb.i - 1 || (s.i == sb.i - 1 && s.j < sb.j - 1)) lo_ = mid + 1; else hi_ = mid; }
            if (lo_ < b && seeds[lo_].i == sb.i - 1 && seeds[lo_].j == sb.j - 1 && best[lo_] + match > v) { v = best[lo_] + match; f = (int)lo_; }
        }
        // Earlier seeds a with sa.i + k <= sb.i and sa.j + k <= sb.j, tried in ascending order, a later one replacing only when strictly
        // better: the winner is the SMALLEST a among those with the largest candidate.  The literal loop is quadratic in the seeds of
        // a pair (thousands along a 5 kb segment: the critical path of cfg3's host stage); the same winner is found scanning
        // DOWNWARDS from the last seed that can precede sb and stopping at the first a whose bound
        //     prefmax[a] + k * match - pen(gi(a))     (prefmax[a] = largest best[0..a]; every a' <= a has a gap of at least gi(a)
        //                                              along the read, seeds being ordered by i, and pen grows with the gap)
        // is below the best candidate found so far: nothing at or before a can reach it, let alone tie.
        while (hi_i < b && seeds[hi_i].i + k <= sb.i) ++hi_i;          // seeds [0, hi_i) end before sb along the read (sb.i only grows)
        long long vp = -1; int fp = -1; bool have = false;
        for (size_t a = hi_i; a-- > 0;) {
            const Seed sa = seeds[a];
            const long long gi = (long long)sb.i - sa.i - k;
            const long long bound = prefmax[a] + seed_score - (gi > 0 ? -(long long)gap_open - (long long)gap_extend * gi : 0);
            if (bound < ((have && vp > v) ? vp : v + 1)) break;      // (a candidate only counts if it beats `v` strictly; once one does, ties with it still matter)
            if (sa.j + k > sb.j) continue;
            const long long gj = (long long)sb.j - sa.j - k, d = std::max(gi, gj);
            const long long cand = best[a] + seed_score - (d > 0 ? -(long long)gap_open - (long long)gap_extend * d : 0);
            if (!have || cand >= vp) { vp = cand; fp = (int)a; have = true; }      // (>=: of equal candidates the smaller index wins)
        }
        if (have && vp > v) { v = vp; f = fp; }
        best[b] = v; from[b] = f;
        prefmax[b] = b > 0 ? std::max(prefmax[b - 1], v) : v;
    }
    size_t end = 0;
    for (size_t b = 1; b < Q; ++b) if (best[b] > best[end]) end = b;
    for (int b = (int)end; b >= 0; b = from[b]) chain.push_back((uint32_t)b);
    std::reverse(chain.begin(), chain.end());

    return false;
}

bool make_band(const std::vector<Seed>& seeds, uint32_t m, uint32_t n, uint32_t k, uint32_t w, int32_t match, int32_t gap_open,
               int32_t gap_extend, std::vector<uint16_t>& lo, std::vector<uint16_t>& hi) {
    std::vector<uint32_t> chain;
    const bool full = backbone_chain(seeds, k, match, gap_open, gap_extend, chain);
    rasterise_band(seeds, chain, m, n, k, w, lo, hi);
    return full;
}

template <typename R>
static void rasterise_band_t(const std::vector<Seed>& seeds, const std::vector<uint32_t>& chain, uint32_t m, uint32_t n, uint32_t k, uint32_t w,
                             std::vector<R>& lo, std::vector<R>& hi) {
    lo.assign(n + 1, (R)(m + 1)); hi.assign(n + 1, 0);
    if (chain.empty()) { std::fill(lo.begin(), lo.end(), (R)0); std::fill(hi.begin(), hi.end(), (R)(m + 1)); return; }

    // 3. band around the backbone: the union of the squares of half-width w around its points
    auto add = [&](long r, long c) {
        const long c0 = std::max<long>(c - (long)w, 0), c1 = std::min<long>(c + (long)w, (long)n);
        const R r0 = (R)std::max<long>(r - (long)w, 0), r1 = (R)(std::min<long>(r + (long)w, (long)m) + 1);
        for (long cc = c0; cc <= c1; ++cc) { if (r0 < lo[cc]) lo[cc] = r0; if (r1 > hi[cc]) hi[cc] = r1; }
    };
    // the same for the points (r + t, c + t), t = 0..len, of a diagonal run, one update per column instead of one per point
    // and column: column cc is within w of the points t in [cc - c - w, cc - c + w]
    auto add_diag = [&](long r, long c, long len) {
        if (len < 0) return;
        const long c0 = std::max<long>(c - (long)w, 0), c1 = std::min<long>(c + len + (long)w, (long)n);
        for (long cc = c0; cc <= c1; ++cc) {
            const long t0 = std::max<long>(cc - c - (long)w, 0), t1 = std::min<long>(cc - c + (long)w, len);
            if (t0 > t1) continue;
            const R r0 = (R)std::max<long>(r + t0 - (long)w, 0), r1 = (R)(std::min<long>(r + t1 + (long)w, (long)m) + 1);
            if (r0 < lo[cc]) lo[cc] = r0;
            if (r1 > hi[cc]) hi[cc] = r1;
        }
    };
    for (size_t p = 0; p < chain.size(); ++p) {
        const Seed s = seeds[chain[p]];
        // seeds that follow each other one step down the diagonal (every position of an exact stretch is a seed) make ONE run: the
        // union of their squares is the run's
        size_t e = p;
        while (e + 1 < chain.size() && seeds[chain[e + 1]].i == seeds[chain[e]].i + 1 && seeds[chain[e + 1]].j == seeds[chain[e]].j + 1) ++e;
        add_diag((long)s.i, (long)s.j, (long)k + (long)(e - p));
        p = e;
        const Seed s_end = seeds[chain[p]];
        if (p + 1 < chain.size()) {
            const Seed nx = seeds[chain[p + 1]];
            const long ai = (long)s_end.i + k, aj = (long)s_end.j + k;
            if ((long)nx.i >= ai && (long)nx.j >= aj) {
                const long gi = (long)nx.i - ai, gj = (long)nx.j - aj, steps = std::max(gi, gj);
                for (long a = 1; a < steps; ++a) add(ai + gi * a / steps, aj + gj * a / steps);
            }
        }
    }
    { const Seed s = seeds[chain.front()]; const long back = (long)std::min(s.i, s.j); add_diag((long)s.i - back, (long)s.j - back, back - 1); }
    { const Seed s = seeds[chain.back()]; const long ie = (long)s.i + k, je = (long)s.j + k;
      add_diag(ie + 1, je + 1, std::min((long)m - ie, (long)n - je) - 1); }
}

void rasterise_band(const std::vector<Seed>& seeds, const std::vector<uint32_t>& chain, uint32_t m, uint32_t n, uint32_t k, uint32_t w,
                    std::vector<uint16_t>& lo, std::vector<uint16_t>& hi) { rasterise_band_t<uint16_t>(seeds, chain, m, n, k, w, lo, hi); }
// (reads beyond 65 534 bases: row numbers no longer fit 16 bits)
void rasterise_band32(const std::vector<Seed>& seeds, const std::vector<uint32_t>& chain, uint32_t m, uint32_t n, uint32_t k, uint32_t w,
                      std::vector<uint32_t>& lo, std::vector<uint32_t>& hi) { rasterise_band_t<uint32_t>(seeds, chain, m, n, k, w, lo, hi); }

// the calls of rasterise_band as data (same order, same arguments)
void band_elements(const std::vector<Seed>& seeds, const std::vector<uint32_t>& chain, uint32_t m, uint32_t n, uint32_t k, std::vector<BandElem>& out) {
    out.clear();
    if (chain.empty()) return;
    auto diag = [&](long r, long c, long len) { if (len >= 0) out.push_back(BandElem{(int32_t)r, (int32_t)c, (int32_t)len, -1}); };
    for (size_t p = 0; p < chain.size(); ++p) {
        const Seed s = seeds[chain[p]];
        size_t e = p;
        while (e + 1 < chain.size() && seeds[chain[e + 1]].i == seeds[chain[e]].i + 1 && seeds[chain[e + 1]].j == seeds[chain[e]].j + 1) ++e;
        diag((long)s.i, (long)s.j, (long)k + (long)(e - p));
        p = e;
        const Seed s_end = seeds[chain[p]];
        if (p + 1 < chain.size()) {
            const Seed nx = seeds[chain[p + 1]];
            const long ai = (long)s_end.i + k, aj = (long)s_end.j + k;
            if ((long)nx.i >= ai && (long)nx.j >= aj) {
                const long gi = (long)nx.i - ai, gj = (long)nx.j - aj;
                if (std::max(gi, gj) >= 2) out.push_back(BandElem{(int32_t)ai, (int32_t)aj, (int32_t)gi, (int32_t)gj});
            }
        }
    }
    { const Seed s = seeds[chain.front()]; const long back = (long)std::min(s.i, s.j); diag((long)s.i - back, (long)s.j - back, back - 1); }
    { const Seed s = seeds[chain.back()]; const long ie = (long)s.i + k, je = (long)s.j + k; diag(ie + 1, je + 1, std::min((long)m - ie, (long)n - je) - 1); }
}

}  // namespace stitch
