// TEST INFRASTRUCTURE — lane-serial CPU emulation of stitch_amd/csrc/fill_kernel.hip.
//
// Executes the same per-row functions (dp_core.h) and the same fix-up / walk code (walk_core.h) as the HIP
// kernels, with the wave64 cross-lane steps (neighbour shuffles, the insertion prefix scan, the two reductions)
// written as loops over a 64-entry lane array.  It exists so that the column-parallel formulation of the
// reference's row-serial recurrence can be checked against the oracle in the GPU-less build container
// (`pytest -m "not gpu"`).  It is not part of the product: libstitch_amd.so does not contain or call it.
#include <cstdio>
#include <stdexcept>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../stitch_amd/csrc/dp_core.h"
#include "../../stitch_amd/csrc/walk_core.h"

using namespace stitch;

namespace {
constexpr int R = 4;
constexpr uint32_t TILE = 64 * R;
constexpr uint32_t PAD = 512;      // row blocks are padded to the larger (Local-mode) tile

struct EmuCtx {
    DpParams P; uint32_t C = 0;
    std::vector<ContigDesc> cd; std::vector<std::string> names; std::vector<uint8_t> fwd; std::vector<uint8_t> xseq;
    std::vector<int32_t> S0, Sn0; std::vector<uint32_t> Slen0, lx0; std::vector<uint8_t> SnSet0, Smove0, Imove0; std::vector<JumpBase> base0;
};

struct EmuJob {
    std::vector<uint8_t> y; std::vector<uint32_t> act; std::vector<int32_t> opp; std::vector<ContigDesc> cd;
    std::vector<int32_t> S, D, Sn, Ival, Sm; std::vector<uint32_t> Slen, Dlen, SnLen, Ly, Ilen, SidxF, SfromF, Lx, jti, jtf, Lm;
    std::vector<uint8_t> tb, SmoveF, ImoveF;
    JobView V;
};

void emu_fill(const EmuCtx& X, EmuJob& J) {
    JobView& V = J.V; const DpParams P = V.P; const uint32_t n = V.n, nact = V.nact, Rtot = V.Rtot;
    std::vector<JumpBase> base[2] = {std::vector<JumpBase>(X.C), std::vector<JumpBase>(X.C)};
    std::vector<uint8_t> rowm_mv(X.C); std::vector<int32_t> rowm_S(X.C); std::vector<uint32_t> rowm_len(X.C);
    for (uint32_t k = 0; k < nact; ++k) {
        uint32_t c = V.act[k]; const ContigDesc& cd = V.cd[c];
        uint32_t mpad = (cd.m + PAD - 1) / PAD * PAD;
        for (uint32_t i = 0; i < mpad; ++i) {
            uint32_t r = cd.roff + i, tr = cd.troff + i;
            V.S[r] = X.S0[tr]; V.Slen[r] = X.Slen0[tr]; V.D[r] = MIN_SCORE; V.Dlen[r] = 0;
            V.Sn[r] = X.Sn0[tr]; V.SnLen[r] = X.Slen0[tr]; V.Ly[r] = X.SnSet0[tr] ? n : 0u; V.SmoveF[r] = TB_NONE; V.ImoveF[r] = TB_NONE;
        }
        uint32_t trm = cd.troff + cd.m - 1;
        base[0][c] = X.base0[c]; rowm_mv[c] = X.Smove0[trm]; rowm_S[c] = X.S0[trm]; rowm_len[c] = X.Slen0[trm];
        V.Lx[(size_t)c * (n + 1)] = X.lx0[c];
    }
    int32_t sn0; uint32_t ly0; row0_init_sn(P, n, sn0, ly0);
    Row0 r0prev = row0_column0();
    for (uint32_t j = 1; j <= n; ++j) {
        const int cur = j & 1, prv = cur ^ 1;
        const Row0 r0 = row0_step(P, j, n, sn0, ly0);
        const uint8_t q = V.y[j - 1];
        int32_t go_j = P.gap_open + P.gap_extend * (int32_t)j;
        const int32_t xclip_score = P.xclip_prefix + (P.yclip_prefix > go_j ? P.yclip_prefix : go_j);
        uint8_t* tbcol = V.tb + (size_t)(j - 1) * Rtot;
        for (uint32_t k = 0; k < nact; ++k) {
            const uint32_t c = V.act[k]; const ContigDesc cd = V.cd[c]; const uint32_t m = cd.m, roff = cd.roff;
            const uint8_t* xs = V.xseq + cd.seqoff;
            ColCtx cx; cx.j = j; cx.n = n; cx.m = m; cx.cidx = c; cx.q = q; cx.xclip_score = xclip_score; cx.row0_len = r0.Slen;
            cx.jump = select_jump(P, base[prv].data(), V.act, nact, c, V.opp_act[c]);
            cx.circ_ok = (P.circular && rowm_mv[c] != TB_XCLIP_SUFFIX) ? 1 : 0; cx.circ_score = rowm_S[c]; cx.circ_len = rowm_len[c] + 1;
            V.jt_idx[(size_t)c * (n + 1) + j] = cx.jump.idx; V.jt_from[(size_t)c * (n + 1) + j] = cx.jump.from;
            int32_t upS = r0prev.S; uint32_t upSl = r0prev.Slen; int32_t upT = r0.S; uint32_t upTl = r0.Slen;
            ScanEl carry = scan_seed();
            XsRec xbest[64]; CmRec cbest[64];
            for (int l = 0; l < 64; ++l) { xbest[l].v = MIN_SCORE; xbest[l].len = 0; xbest[l].row = 0; cbest[l].v = r0.S; cbest[l].row = 0; cbest[l].len = r0.Slen; }
            int32_t ownS = 0, ownDg = 0, ownSn = 0; uint32_t ownSl = 0, ownMv = 0, ownBits = 0;
            const uint32_t ntiles = (m + TILE - 1) / TILE;
            for (uint32_t t = 0; t < ntiles; ++t) {
                int32_t Sp[64][R], Dp[64][R], Snv[64][R]; uint32_t Slp[64][R], Dlp[64][R]; uint8_t xb[64][R];
                RowA ra[64][R]; ScanEl el[64][R], agg[64], inc[64], run[64];
                for (int l = 0; l < 64; ++l) {
                    uint32_t i0 = t * TILE + l * R + 1, r = roff + i0 - 1;
                    for (int u = 0; u < R; ++u) { Sp[l][u] = V.S[r + u]; Slp[l][u] = V.Slen[r + u]; Dp[l][u] = V.D[r + u]; Dlp[l][u] = V.Dlen[r + u]; Snv[l][u] = V.Sn[r + u]; xb[l][u] = xs[i0 - 1 + u]; }
                }
                for (int l = 0; l < 64; ++l) {
                    uint32_t i0 = t * TILE + l * R + 1;
                    int32_t nS = l == 0 ? upS : Sp[l - 1][R - 1]; uint32_t nSl = l == 0 ? upSl : Slp[l - 1][R - 1];
                    for (int u = 0; u < R; ++u) {
                        uint32_t i = i0 + u;
                        row_phase_a(P, cx, i <= m ? i : 1u, xb[l][u], u == 0 ? nS : Sp[l][u - 1], u == 0 ? nSl : Slp[l][u - 1], Sp[l][u], Slp[l][u],
                                    Dp[l][u], Dlp[l][u], X.Slen0.data() + cd.troff, ra[l][u]);
                    }
                }
                upS = Sp[63][R - 1]; upSl = Slp[63][R - 1];
                for (int l = 0; l < 64; ++l) {
                    uint32_t i0 = t * TILE + l * R + 1;
                    int32_t nT = l == 0 ? upT : ra[l - 1][R - 1].T; uint32_t nTl = l == 0 ? upTl : ra[l - 1][R - 1].Tl;
                    for (int u = 0; u < R; ++u) {
                        uint32_t i = i0 + u;
                        el[l][u] = scan_make(P, i, u == 0 ? nT : ra[l][u - 1].T, u == 0 ? nTl : ra[l][u - 1].Tl);
                        if (i > m) el[l][u].key = KEY_NEG_INF;
                    }
                    agg[l] = el[l][0];
                    for (int u = 1; u < R; ++u) agg[l] = scan_combine(agg[l], el[l][u]);
                }
                upT = ra[63][R - 1].T; upTl = ra[63][R - 1].Tl;
                // Hillis-Steele inclusive scan, exactly the kernel's shuffle schedule
                for (int l = 0; l < 64; ++l) inc[l] = agg[l];
                for (int d = 1; d < 64; d <<= 1) {
                    ScanEl nx[64];
                    for (int l = 0; l < 64; ++l) nx[l] = l >= d ? scan_combine(inc[l - d], inc[l]) : inc[l];
                    for (int l = 0; l < 64; ++l) inc[l] = nx[l];
                }
                for (int l = 0; l < 64; ++l) run[l] = l == 0 ? carry : scan_combine(carry, inc[l - 1]);
                carry = scan_combine(carry, inc[63]);
                for (int l = 0; l < 64; ++l) {
                    uint32_t i0 = t * TILE + l * R + 1, r = roff + i0 - 1;
                    for (int u = 0; u < R; ++u) {
                        uint32_t i = i0 + u; bool valid = i <= m;
                        uint32_t ext = run[l].key >= el[l][u].key ? 1u : 0u;
                        if (!ext) run[l] = el[l][u];
                        int32_t bi = run[l].key + P.gap_extend * (int32_t)i; uint32_t il = (uint32_t)(run[l].q + (int32_t)i);
                        int32_t So; uint32_t Slo, mv;
                        row_phase_c(P, cx, valid ? i : 1u, ra[l][u], bi, il, X.Slen0.data() + cd.troff, So, Slo, mv);
                        uint32_t code = mv | (ext ? TBB_IEXT : 0u) | (ra[l][u].dext ? TBB_DEXT : 0u);
                        if (!valid) continue;
                        V.D[r + u] = ra[l][u].bd; V.Dlen[r + u] = ra[l][u].dlen;
                        if (j == n) { V.Ival[r + u] = bi; V.Ilen[r + u] = il; }
                        if (i < m) {
                            XsRec xc; xc.v = So + P.xclip_suffix; xc.len = Slo; xc.row = i; if (xs_better(xc, xbest[l])) xbest[l] = xc;
                            CmRec cc; cc.v = So; cc.row = i; cc.len = Slo; if (cm_better(cc, cbest[l])) cbest[l] = cc;
                            int32_t v = So + P.yclip_suffix;
                            if (v > Snv[l][u] || (v == Snv[l][u] && Slo > 0u)) { V.Sn[r + u] = v; V.Ly[r + u] = n - j; V.SnLen[r + u] = Slo; }
                            V.S[r + u] = So; V.Slen[r + u] = Slo; tbcol[r + u] = (uint8_t)code;
                        } else {
                            ownS = So; ownSl = Slo; ownMv = mv; ownBits = code & (TBB_IEXT | TBB_DEXT); ownDg = ra[l][u].dg; ownSn = Snv[l][u];
                        }
                    }
                }
            }
            XsRec xb_ = xbest[0]; CmRec cb_ = cbest[0];
            for (int l = 1; l < 64; ++l) { if (xs_better(xbest[l], xb_)) xb_ = xbest[l]; if (cm_better(cbest[l], cb_)) cb_ = cbest[l]; }
            {
                const uint32_t rm = roff + m - 1;
                int32_t Sm; uint32_t Slm, mvm, lx = xb_.row == 0 ? 0u : m - xb_.row; bool do_x_m = false;
                if (rowm_run_wins(xb_.v, ownS, ownDg)) { Sm = xb_.v; Slm = xb_.len; mvm = MV_XSUF; }
                else { Sm = ownS; Slm = ownSl; mvm = ownMv; if (P.xclip_suffix == 0 && ownSl > xb_.len) { do_x_m = true; lx = 0; } }
                V.S[rm] = Sm; V.Slen[rm] = Slm; tbcol[rm] = (uint8_t)(mvm | ownBits);
                uint32_t rl = (j == n) ? (do_x_m ? ownSl : xb_.len) : 0u;
                int32_t v = Sm + P.yclip_suffix;
                if (v > ownSn || (v == ownSn && Slm > rl)) { V.Sn[rm] = v; V.Ly[rm] = n - j; V.SnLen[rm] = Slm; }
                V.Lx[(size_t)c * (n + 1) + j] = lx;
                CmRec cc; cc.v = Sm; cc.row = m; cc.len = Slm; if (cm_better(cc, cb_)) cb_ = cc;
                JumpBase b; b.score = cb_.v; b.len = cb_.len + 1; b.from = cb_.row; base[cur][c] = b;
                uint32_t refmv = (mvm == MV_XSUF) ? TB_XCLIP_SUFFIX : (mvm == MV_INS) ? TB_INS : (mvm == MV_DEL) ? TB_DEL
                               : (mvm == MV_XPRE) ? TB_XCLIP_PREFIX : (mvm == MV_YPRE) ? TB_YCLIP_PREFIX : TB_MATCH;
                rowm_mv[c] = (uint8_t)refmv; rowm_S[c] = Sm; rowm_len[c] = Slm;
            }
        }
        r0prev = r0;
    }
}

// ---- emulation of fill_local16.hip: combined-word selection, 16-bit packed state, 8 rows per lane, filtered Sn ------
void emu_fill_local(const EmuCtx& X, EmuJob& J) {
    constexpr int R8 = 8; constexpr uint32_t T8 = 64 * R8;
    JobView& V = J.V; const DpParams P = V.P; const uint32_t n = V.n, nact = V.nact, Rtot = V.Rtot;
    std::vector<int32_t> st(2 * (size_t)Rtot);
    std::vector<JumpBase> base(X.C); std::vector<int32_t> vrun(X.C), rowm_S(X.C); std::vector<uint8_t> rowm_mv(X.C); std::vector<uint32_t> rowm_len(X.C);
    const int32_t GE1 = (int32_t)((uint32_t)P.gap_extend << 16) + 1, GO1 = (int32_t)((uint32_t)(P.gap_open + P.gap_extend) << 16) + 1;
    const int32_t MW = (int32_t)((uint32_t)P.match << 16), XW = (int32_t)((uint32_t)P.mismatch << 16);
    for (uint32_t k = 0; k < nact; ++k) {
        uint32_t c = V.act[k]; const ContigDesc& cd = V.cd[c];
        uint32_t mpad = (cd.m + PAD - 1) / PAD * PAD;
        for (uint32_t i = 0; i < mpad; ++i) {
            uint32_t r = cd.roff + i, tr = cd.troff + i;
            st[2 * r] = word_make(X.S0[tr], X.Slen0[tr]); st[2 * r + 1] = word_make(-16384, 0);
            V.Sn[r] = X.Sn0[tr]; V.SnLen[r] = X.Slen0[tr]; V.Ly[r] = X.SnSet0[tr] ? n : 0u; V.SmoveF[r] = TB_NONE; V.ImoveF[r] = TB_NONE;
        }
        uint32_t trm = cd.troff + cd.m - 1;
        base[c] = X.base0[c]; vrun[c] = X.base0[c].score; rowm_mv[c] = X.Smove0[trm]; rowm_S[c] = X.S0[trm]; rowm_len[c] = X.Slen0[trm];
        V.Lx[(size_t)c * (n + 1)] = X.lx0[c];
    }
    for (uint32_t j = 1; j <= n; ++j) {
        std::vector<JumpInfo> sj(X.C); std::vector<uint8_t> scirc(X.C);
        for (uint32_t k = 0; k < nact; ++k) {
            uint32_t c = V.act[k];
            JumpInfo ji = select_jump(P, base.data(), V.act, nact, c, V.opp_act[c]);
            ColCtx cx{}; cx.jump = ji; cx.circ_ok = (P.circular && rowm_mv[c] != TB_XCLIP_SUFFIX) ? 1 : 0; cx.circ_score = rowm_S[c]; cx.circ_len = rowm_len[c] + 1;
            scirc[c] = local_row1_circ(cx) ? 1 : 0; sj[c] = ji;
            V.jt_idx[(size_t)c * (n + 1) + j] = ji.idx | (scirc[c] ? JT_CIRC_BIT : 0u); V.jt_from[(size_t)c * (n + 1) + j] = ji.from;
        }
        const uint8_t q = V.y[j - 1];
        uint8_t* tbcol = V.tb + (size_t)(j - 1) * Rtot;
        for (uint32_t k = 0; k < nact; ++k) {
            const uint32_t c = V.act[k]; const ContigDesc cd = V.cd[c]; const uint32_t m = cd.m, roff = cd.roff;
            const uint8_t* xs = V.xseq + cd.seqoff;
            const int32_t JSW = word_make(sj[c].score, sj[c].len);
            const int32_t JSW1 = scirc[c] ? word_make(rowm_S[c], rowm_len[c] + 1) : JSW;
            const int32_t vr = vrun[c];
            int32_t upS = 0, upT = 0;                 // words of row 0: score 0, length 0
            ScanEl carry = scan_seed();
            XsRec xb_; xb_.v = MIN_SCORE; xb_.len = 0; xb_.row = 0; CmRec cb_; cb_.v = 0; cb_.row = 0; cb_.len = 0;
            int32_t ownF = 0, ownDG = 0, ownD = 0; uint32_t ownMv = 0, ownBits = 0;
            const uint32_t ntiles = (m + T8 - 1) / T8;
            for (uint32_t t = 0; t < ntiles; ++t) {
                static int32_t Sp[64][R8], Dp[64][R8]; static RowW ra[64][R8]; static ScanEl el[64][R8], agg[64], inc[64], run[64];
                for (int l = 0; l < 64; ++l) for (int u = 0; u < R8; ++u) { uint32_t r = roff + t * T8 + l * R8 + u; Sp[l][u] = st[2 * r]; Dp[l][u] = st[2 * r + 1]; }
                for (int l = 0; l < 64; ++l) for (int u = 0; u < R8; ++u) {
                    uint32_t i = t * T8 + l * R8 + u + 1;
                    int32_t nS = u ? Sp[l][u - 1] : (l ? Sp[l - 1][R8 - 1] : upS);
                    row_phase_a_word(xs[i - 1] == q ? MW : XW, GE1, GO1, i == 1 ? JSW1 : JSW, nS, Sp[l][u], Dp[l][u], ra[l][u]);
                }
                upS = Sp[63][R8 - 1];
                for (int l = 0; l < 64; ++l) {
                    for (int u = 0; u < R8; ++u) {
                        uint32_t i = t * T8 + l * R8 + u + 1;
                        int32_t nT = u ? ra[l][u - 1].T : (l ? ra[l - 1][R8 - 1].T : upT);
                        el[l][u] = scan_make(P, i, word_score(nT), word_len(nT));
                        if (i > m) el[l][u].key = KEY_NEG_INF;
                    }
                    agg[l] = el[l][0]; for (int u = 1; u < R8; ++u) agg[l] = scan_combine(agg[l], el[l][u]);
                }
                upT = ra[63][R8 - 1].T;
                inc[0] = agg[0]; for (int l = 1; l < 64; ++l) inc[l] = scan_combine(inc[l - 1], agg[l]);
                for (int l = 0; l < 64; ++l) run[l] = l == 0 ? carry : scan_combine(carry, inc[l - 1]);
                carry = scan_combine(carry, inc[63]);
                for (int l = 0; l < 64; ++l) for (int u = 0; u < R8; ++u) {
                    uint32_t i = t * T8 + l * R8 + u + 1, r = roff + i - 1; bool valid = i <= m;
                    uint32_t ext = run[l].key >= el[l][u].key ? 1u : 0u; if (!ext) run[l] = el[l][u];
                    int32_t bi = run[l].key + P.gap_extend * (int32_t)i; uint32_t il = (uint32_t)(run[l].q + (int32_t)i);
                    uint32_t mv; int32_t F = row_phase_c_word(ra[l][u], bi, il, mv);
                    int32_t So = word_score(F); uint32_t Slo = word_len(F);
                    uint32_t code = mv | (ext ? TBB_IEXT : 0u) | (ra[l][u].dext ? TBB_DEXT : 0u);
                    if (!valid) { st[2 * r] = 0; st[2 * r + 1] = word_make(-16384, 0); continue; }
                    st[2 * r + 1] = ra[l][u].BD;
                    if (j == n) { V.Ival[r] = bi; V.Ilen[r] = il; }
                    if (i < m) {
                        st[2 * r] = F; tbcol[r] = (uint8_t)code;
                        if (j == n) { V.S[r] = So; V.Slen[r] = Slo; }
                        XsRec xc; xc.v = So; xc.len = Slo; xc.row = i; if (xs_better(xc, xb_)) xb_ = xc;
                        CmRec cc; cc.v = So; cc.row = i; cc.len = Slo; if (cm_better(cc, cb_)) cb_ = cc;
                        // So >= vr implies So >= Sn[r]: Sn[r] is a maximum over earlier columns of this row and vr the contig's running
                        // maximum over the same columns, so the kernel stores without reading Sn (checked here)
                        if (So >= vr && So < V.Sn[r]) throw std::runtime_error("y-suffix tracker invariant: Sn above the contig's running maximum");
                        if (So >= vr && Slo > 0u) { V.Sn[r] = So; V.Ly[r] = n - j; V.SnLen[r] = Slo; }
                    } else { ownF = F; ownMv = mv; ownBits = code & (TBB_IEXT | TBB_DEXT); ownDG = ra[l][u].DG; ownD = ra[l][u].BD; }
                }
            }
            const uint32_t rm = roff + m - 1;
            int32_t ownS = word_score(ownF); uint32_t ownSl = word_len(ownF);
            int32_t Sm; uint32_t Slm, mvm, lx = xb_.row == 0 ? 0u : m - xb_.row; bool do_x_m = false;
            if (rowm_run_wins(xb_.v, ownS, word_score(ownDG))) { Sm = xb_.v; Slm = xb_.len; mvm = MK_XSUF; }
            else { Sm = ownS; Slm = ownSl; mvm = ownMv; if (ownSl > xb_.len) { do_x_m = true; lx = 0; } }
            st[2 * rm] = word_make(Sm, Slm); st[2 * rm + 1] = ownD; tbcol[rm] = (uint8_t)(mvm | ownBits);
            if (j == n) { V.S[rm] = Sm; V.Slen[rm] = Slm; }
            uint32_t rl = (j == n) ? (do_x_m ? ownSl : xb_.len) : 0u;
            if (Sm >= vr) {
                int32_t sn = V.Sn[rm];
                if (Sm < sn) throw std::runtime_error("y-suffix tracker invariant (row m)");
                // before the last column rl = 0 and a zero-length S is a clipped 0, which cannot exceed Sn >= 0: the kernel's test is Slm > 0
                if (j < n && ((Sm > sn || (Sm == sn && Slm > rl)) != (Slm > 0u))) throw std::runtime_error("y-suffix tracker invariant (row m, length)");
                if (Sm > sn || (Sm == sn && Slm > rl)) { V.Sn[rm] = Sm; V.Ly[rm] = n - j; V.SnLen[rm] = Slm; }
            }
            V.Lx[(size_t)c * (n + 1) + j] = lx;
            CmRec cc; cc.v = Sm; cc.row = m; cc.len = Slm; if (cm_better(cc, cb_)) cb_ = cc;
            JumpBase b; b.score = cb_.v; b.len = cb_.len + 1; b.from = cb_.row; base[c] = b;
            if (cb_.v > vr) vrun[c] = cb_.v;
            rowm_mv[c] = (uint8_t)(mvm == MK_XSUF ? TB_XCLIP_SUFFIX : TB_MATCH); rowm_S[c] = Sm; rowm_len[c] = Slm;
        }
    }
}

size_t put_chain(const ChainHdr& H, const OpRec* ops, int64_t* out, size_t cap) {
    size_t need = 12 + 3 * (size_t)H.n_ops;
    if (need > cap) return need;
    out[0] = H.score; out[1] = H.xstart; out[2] = H.xend; out[3] = H.ystart; out[4] = H.yend; out[5] = H.xlen; out[6] = H.ylen;
    out[7] = H.start_contig_idx; out[8] = H.end_contig_idx; out[9] = H.length; out[10] = 4; out[11] = H.n_ops;
    for (uint32_t k = 0; k < H.n_ops; ++k) {
        out[12 + 3 * k] = ops[k].kind;
        bool xj = ops[k].kind == OP_XJUMP;
        out[13 + 3 * k] = xj ? ops[k].contig : ops[k].arg; out[14 + 3 * k] = xj ? ops[k].arg : 0;
    }
    return need;
}
}  // namespace

extern "C" {

// params: [match, mismatch, go, ge, jump_same, jump_opp, jump_inter, xp, xs, yp, ys, circular]
void* emu_ctx_new(const int32_t* params, uint32_t C, const char* const* names, const int32_t* is_fwd, const uint8_t* const* seqs,
                  const uint32_t* lens) {
    auto* X = new EmuCtx();
    DpParams& P = X->P;
    P.match = params[0]; P.mismatch = params[1]; P.gap_open = params[2]; P.gap_extend = params[3]; P.jump_same = params[4];
    P.jump_opp = params[5]; P.jump_inter = params[6]; P.xclip_prefix = params[7]; P.xclip_suffix = params[8]; P.yclip_prefix = params[9];
    P.yclip_suffix = params[10]; P.circular = params[11];
    X->C = C;
    uint32_t troff = 0;
    for (uint32_t a = 0; a < C; ++a) {
        ContigDesc d{}; d.m = lens[a]; d.troff = troff; d.roff = 0; d.seqoff = (uint32_t)X->xseq.size(); d.target = a; d.opp = -1;
        X->xseq.insert(X->xseq.end(), seqs[a], seqs[a] + lens[a]);
        while (X->xseq.size() % PAD) X->xseq.push_back(0);
        troff += (d.m + PAD - 1) / PAD * PAD;
        X->cd.push_back(d); X->names.emplace_back(names[a]); X->fwd.push_back((uint8_t)is_fwd[a]);
    }
    for (uint32_t a = 0; a < C; ++a) {
        if (X->cd[a].opp >= 0) continue;
        for (uint32_t b = a + 1; b < C; ++b)
            if (X->names[a] == X->names[b] && X->fwd[a] != X->fwd[b]) { X->cd[a].opp = (int32_t)b; X->cd[b].opp = (int32_t)a; }
    }
    X->S0.assign(troff, MIN_SCORE); X->Sn0.assign(troff, MIN_SCORE); X->Slen0.assign(troff, 0); X->lx0.assign(C, 0);
    X->SnSet0.assign(troff, 0); X->Smove0.assign(troff, 0); X->Imove0.assign(troff, 0); X->base0.resize(C);
    std::vector<Col0Row> rows;
    for (uint32_t a = 0; a < C; ++a) {
        rows.resize(X->cd[a].m);
        X->lx0[a] = col0_init(P, X->cd[a].m, rows.data());
        JumpBase b; b.score = 0; b.from = 0; b.len = 1;
        for (uint32_t i = 1; i <= X->cd[a].m; ++i) {
            const Col0Row& r = rows[i - 1]; uint32_t x = X->cd[a].troff + i - 1;
            X->S0[x] = r.S; X->Slen0[x] = r.Slen; X->Sn0[x] = r.Sn; X->SnSet0[x] = r.sn_set; X->Smove0[x] = r.Smove; X->Imove0[x] = r.Imove;
            if (b.score < r.S) { b.score = r.S; b.from = i; b.len = r.Slen + 1; }
        }
        X->base0[a] = b;
    }
    return X;
}
void emu_ctx_free(void* h) { delete (EmuCtx*)h; }

// Runs one DP job and writes chains in the oracle's wire format, concatenated; returns the number of chains or <0.
// mode 0: traceback, 1: one chain per active contig (status None => n_ops = -1 marker), 2: traceback_from(from), 3: as 1 with the walks JOINED to the
// reference chain the way the device does it (walk_core.h JoinRole)
long emu_job(void* h, const uint8_t* y, uint32_t n, const uint32_t* act, uint32_t nact, int mode, uint32_t from, int64_t* out, size_t cap,
             size_t* used, int local16) {
    const EmuCtx& X = *(EmuCtx*)h;
    EmuJob J; J.y.assign(y, y + n); J.act.assign(act, act + nact); J.cd = X.cd; J.opp.assign(X.C, -1);
    std::vector<uint8_t> isact(X.C, 0); for (uint32_t a : J.act) isact[a] = 1;
    uint32_t roff = 0;
    for (uint32_t a = 0; a < X.C; ++a) {
        if (isact[a]) { J.cd[a].roff = roff; roff += (J.cd[a].m + PAD - 1) / PAD * PAD; }
        if (isact[a] && X.cd[a].opp >= 0 && isact[X.cd[a].opp]) J.opp[a] = X.cd[a].opp;
    }
    const uint32_t Rj = roff;
    J.S.assign(Rj, 0); J.D.assign(Rj, 0); J.Sn.assign(Rj, 0); J.Ival.assign(Rj, 0); J.Sm.assign(X.C, 0);
    J.Slen.assign(Rj, 0); J.Dlen.assign(Rj, 0); J.SnLen.assign(Rj, 0); J.Ly.assign(Rj, 0); J.Ilen.assign(Rj, 0); J.SidxF.assign(Rj, 0); J.SfromF.assign(Rj, 0);
    J.Lx.assign((size_t)X.C * (n + 1), 0); J.jti.assign((size_t)X.C * (n + 1), 0); J.jtf.assign((size_t)X.C * (n + 1), 0); J.Lm.assign(X.C, 0);
    J.tb.assign((size_t)n * Rj, 0); J.SmoveF.assign(Rj, TB_NONE); J.ImoveF.assign(Rj, TB_NONE);
    JobView& V = J.V;
    V.P = X.P; V.n = n; V.C = X.C; V.nact = nact; V.Rtot = Rj; V.act = J.act.data(); V.opp_act = J.opp.data(); V.cd = J.cd.data();
    V.xseq = X.xseq.data(); V.y = J.y.data(); V.S = J.S.data(); V.Slen = J.Slen.data(); V.D = J.D.data(); V.Dlen = J.Dlen.data();
    V.Sn = J.Sn.data(); V.SnLen = J.SnLen.data(); V.Ly = J.Ly.data(); V.tb = J.tb.data(); V.Lx = J.Lx.data(); V.jt_idx = J.jti.data();
    V.jt_from = J.jtf.data(); V.Ival = J.Ival.data(); V.Ilen = J.Ilen.data(); V.SmoveF = J.SmoveF.data(); V.SidxF = J.SidxF.data();
    V.SfromF = J.SfromF.data(); V.ImoveF = J.ImoveF.data(); V.Smove0 = X.Smove0.data(); V.Imove0 = X.Imove0.data(); V.Slen0 = X.Slen0.data();
    V.Sm = J.Sm.data(); V.Lm = J.Lm.data();
    V.tb_keyfmt = local16 ? 1u : 0u; V.visit = nullptr; V.Wcol = nullptr;
    try { if (local16) emu_fill_local(X, J); else emu_fill(X, J); }
    catch (const std::exception& e) { fprintf(stderr, "emu: %s\n", e.what()); return -3; }
    for (uint32_t k = 0; k < nact; ++k) fixup_contig(V, act[k]);
    uint32_t max_m = 0; for (auto& d : X.cd) max_m = d.m > max_m ? d.m : max_m;
    const uint32_t ops_cap = (n + 1) * (max_m + 2) + 64;   // degenerate scorings (free gaps and jumps) can emit ~n*m ops
    std::vector<OpRec> ops(ops_cap);
    size_t o = 0; long nch = 0; bool undefined = false;
    auto emit = [&](uint32_t c) -> bool {
        ChainHdr H{}; walk_from(V, c, H, ops.data(), ops_cap);
        if (H.status == 4) { undefined = true; H.status = 1; }
        if (H.status >= 2) return false;
        if (H.status == 1) { H.n_ops = 0; H.score = MIN_SCORE; }
        size_t need = put_chain(H, ops.data(), out + o, cap > o ? cap - o : 0);
        if (o + need > cap) return false;
        if (H.status == 1) out[o + 10] = -1;       // mode slot marks None
        o += need; ++nch; return true;
    };
    if (mode == 3) {
        // traceback_all with JOINED walks, as fixup_walk_kernel + walk_all_kernel + download_chains do it (walk_core.h JoinRole): the chain from
        // the best end contig is walked first and records its state on entering every column; every other walk stops where it enters a column
        // in the recorded state, and its chain is the reference chain's first join_ops operations followed by its own
        struct JoinSolo : JoinRole {
            bool writer() const { return true; }
            uint32_t diag_run(const JobView&, uint32_t, uint32_t, uint32_t, uint32_t, OpRec*, uint32_t, uint32_t, bool) const { return 0; }
            bool enter_column(const JobView& V_, uint32_t cur, uint32_t i, uint32_t j, uint32_t layer, uint32_t nops, uint32_t nonspecial, bool yfirst) const {
                return enter_column_as(true, V_, cur, i, j, layer, nops, nonspecial, yfirst);
            }
            void finish_reference(const JobView& V_, uint32_t nops, uint32_t nonspecial, bool usable) const { finish_reference_as(true, V_, nops, nonspecial, usable); }
            void reverse(OpRec* o_, uint32_t nops) const { for (uint32_t a = 0, b = nops; a + 1 < b; ++a, --b) { OpRec t = o_[a]; o_[a] = o_[b - 1]; o_[b - 1] = t; } }
        };
        const uint32_t r = pick_primary(V);
        uint32_t kr = 0; for (uint32_t k = 0; k < nact; ++k) if (act[k] == r) kr = k;
        std::vector<VisitRec> visit(n + 2);
        for (uint32_t e = 0; e < n + 2; ++e) visit[e] = JoinRole::cleared(e, kr);
        V.visit = visit.data();
        std::vector<ChainHdr> hdr(nact); std::vector<std::vector<OpRec>> opsv(nact, std::vector<OpRec>(ops_cap));
        { JoinSolo ex; ex.role = 1; ex.ref_slot = kr; walk_from_t(V, r, hdr[kr], opsv[kr].data(), ops_cap, ex); }
        for (uint32_t k = 0; k < nact; ++k) {
            if (k == kr) continue;
            JoinSolo ex;
            if (visit[0].row == 1) { ex.role = 2; ex.ref_hdr = &hdr[kr]; }
            walk_from_t(V, act[k], hdr[k], opsv[k].data(), ops_cap, ex);
        }
        for (uint32_t k = 0; k < nact; ++k) {
            ChainHdr H = hdr[k];
            if (H.status == 4) H.status = 1;
            if (H.status >= 2) return -2;
            std::vector<OpRec> whole;
            if (H.status == 1) { H.n_ops = 0; H.score = MIN_SCORE; }
            else if (H.join_ops) {
                if (H.join_slot != kr || H.join_ops > hdr[kr].n_ops) return -4;
                whole.assign(opsv[kr].begin(), opsv[kr].begin() + H.join_ops);
                whole.insert(whole.end(), opsv[k].begin(), opsv[k].begin() + H.n_ops);
                H.n_ops = (uint32_t)whole.size();
            }
            else whole.assign(opsv[k].begin(), opsv[k].begin() + H.n_ops);
            size_t need = put_chain(H, whole.data(), out + o, cap > o ? cap - o : 0);
            if (o + need > cap) return -2;
            if (H.status == 1) out[o + 10] = -1;
            o += need; ++nch;
        }
        V.visit = nullptr;
    }
    else if (mode == 1) { for (uint32_t k = 0; k < nact; ++k) if (!emit(act[k])) return -2; }
    else if (!emit(mode == 0 ? pick_primary(V) : from)) return -2;
    (void)undefined;   // chains the reference cannot define are reported as None; the tests skip them when the oracle throws
    if (used) *used = o;
    return nch;
}

// fill_regs.hip's row -> byte mapping of the lane-interleaved traceback layout, as the walk computes it (walk_core.h)
uint32_t emu_tb_row_offset(uint32_t m, uint32_t i) {
    ContigDesc d{}; d.m = m;
    const uint32_t ngr = (m + 3) / 4, gq = ngr / 64;
    d.inv_big = tb_div_magic(4 * (gq + 1)); d.inv_small = tb_div_magic(4 * gq);
    return tb_row_offset(2, d, i);
}

}  // extern "C"
