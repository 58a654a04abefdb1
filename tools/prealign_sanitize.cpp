// CPU sanitizer run of the pre-alignment filter's host code (seeds against brute force, backbone, band, band pieces):
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 -I stitch_amd/csrc tools/prealign_sanitize.cpp stitch_amd/csrc/prealign.cpp -o /tmp/prealign_san && /tmp/prealign_san
#include "prealign.h"
#include <cstdio>
#include <random>
#include <cstring>
using namespace stitch;
int main() {
    std::mt19937 rng(7);
    size_t checks = 0;
    for (int it = 0; it < 400; ++it) {
        const int T = 1 + rng() % 4; std::vector<uint8_t> contigs; std::vector<Strand> strands;
        const char* alpha = (it % 3 == 0) ? "ACGTN" : "ACGT"; const int na = (it % 3 == 0) ? 5 : 4;
        for (int t = 0; t < T; ++t) { uint32_t n = 1 + rng() % 700; strands.push_back({contigs.size(), n}); for (uint32_t i = 0; i < n; ++i) contigs.push_back(alpha[rng() % na]); }
        const uint32_t k = 1 + rng() % ((it % 5 == 0) ? 40 : 14);
        KmerIndex ix = build_kmer_index(contigs.data(), strands, k);
        uint32_t m = 1 + rng() % 900; std::vector<uint8_t> q;
        while (q.size() < m) { if (rng() % 3) { int t = rng() % T; uint32_t a = rng() % strands[t].len; uint32_t len = 1 + rng() % 200; for (uint32_t i = 0; i < len && a + i < strands[t].len && q.size() < m; ++i) q.push_back(rng() % 20 ? contigs[strands[t].off + a + i] : alpha[rng() % na]); } else q.push_back(alpha[rng() % na]); }
        std::vector<std::vector<Seed>> seeds; find_seeds(ix, contigs.data(), strands, q.data(), m, seeds);
        // brute-force seeds
        for (int t = 0; t < T; ++t) {
            std::vector<Seed> want;
            for (uint32_t i = 0; i + k <= m; ++i) for (uint32_t j = 0; j + k <= strands[t].len; ++j) if (memcmp(q.data() + i, contigs.data() + strands[t].off + j, k) == 0) want.push_back({i, j});
            if (want.size() > MAX_MATCHES + 1) want.resize(MAX_MATCHES + 1);
            if (want.size() != seeds[t].size()) { printf("seed count differs it=%d t=%d k=%u: %zu vs %zu\n", it, t, k, seeds[t].size(), want.size()); return 1; }
            for (size_t a = 0; a < want.size(); ++a) if (want[a].i != seeds[t][a].i || want[a].j != seeds[t][a].j) { printf("seed differs\n"); return 1; }
            std::vector<uint16_t> lo, hi; std::vector<uint32_t> chain; std::vector<BandElem> el;
            const uint32_t w = rng() % 60;
            make_band(seeds[t], m, strands[t].len, k, w, 1, -6, -2, lo, hi);
            backbone_chain(seeds[t], k, 1, -6, -2, chain); band_elements(seeds[t], chain, m, strands[t].len, k, el);
            checks += lo.size() + el.size();
        }
    }
    printf("ok %zu\n", checks);
}
