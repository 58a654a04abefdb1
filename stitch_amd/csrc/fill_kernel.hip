// Kernel 1 — column-synchronous fill of the jump-aware affine-gap DP (gfx950, wave64).
//
// Replaces, for a batch of reads, MultiContigAligner::custom's column loop
// (fg-stitch-lib/src/align/aligners/multi_contig_aligner.rs:264-347) together with
// SingleContigAligner::{init_column, get_jump_info, fill_column} (single_contig_aligner.rs:188-451, 677-697).
//
// Mapping: one workgroup per read ("job"), one wavefront per contig (contigs are dealt round-robin over the
// waves), 64 lanes x R consecutive rows per tile, tiles walked top to bottom with the insertion-scan carry kept
// in registers.  Column j-1's per-contig arg-max (the jump source) lives in LDS; one __syncthreads per column is
// the only workgroup-wide synchronisation, as the recurrence demands (a jump may come from any row of any
// contig of the previous column).  Row state (S, D, their alignment lengths, the y-suffix trackers) is kept in
// global memory in place, 16-byte vector loads/stores per lane; the traceback is one byte per cell, written
// once, coalesced, column-major.
#include <hip/hip_runtime.h>
#include "dp_core.h"
#include "walk_core.h"

namespace stitch {

struct FillShared {            // context-level arrays shared by every job of a launch
    const int32_t* S0; const uint32_t* Slen0; const int32_t* Sn0; const uint8_t* SnSet0; const uint8_t* Smove0;   // by troff
    const uint32_t* lx0;       // [C]   Lx[0]
    const JumpBase* base0;     // [C]   get_jump_info over column 0
};

__device__ __forceinline__ int32_t shfl_up_i(int32_t v, int d) { return __shfl_up(v, d, 64); }
__device__ __forceinline__ uint32_t shfl_up_u(uint32_t v, int d) { return (uint32_t)__shfl_up((int)v, d, 64); }
__device__ __forceinline__ int32_t bcast_i(int32_t v, int lane) { return __shfl(v, lane, 64); }
__device__ __forceinline__ uint32_t bcast_u(uint32_t v, int lane) { return (uint32_t)__shfl((int)v, lane, 64); }

constexpr int MAX_CONTIGS = 256;

template <int R>
__global__ __launch_bounds__(512) void fill_kernel(const JobView* __restrict__ jobs, FillShared sh) {
    const JobView& V = jobs[blockIdx.x];
    const DpParams P = V.P;
    const uint32_t n = V.n, nact = V.nact, Rtot = V.Rtot;   // Rtot: this read's (compacted) row count
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int W = blockDim.x >> 6;

    __shared__ JumpBase base[2][MAX_CONTIGS];
    __shared__ uint8_t rowm_mv[MAX_CONTIGS];          // reference move of cell(m, j-1).S, for the circular rule
    __shared__ int32_t rowm_S[MAX_CONTIGS];           // S[prev][m]
    __shared__ uint32_t rowm_len[MAX_CONTIGS];        // cell(m, j-1).S.len
    __shared__ uint32_t s_act[MAX_CONTIGS];
    __shared__ int32_t s_opp[MAX_CONTIGS];

    // ---- column 0: copy the template (init_matrices, single_contig_aligner.rs:97-186) -----------------------
    for (uint32_t k = 0; k < nact; ++k) {
        const uint32_t c = V.act[k];
        const uint32_t roff = V.cd[c].roff, troff = V.cd[c].troff;
        const uint32_t mpad = (V.cd[c].m + 64 * R - 1) / (64 * R) * (64 * R);
        for (uint32_t i = threadIdx.x; i < mpad; i += blockDim.x) {
            const uint32_t r = roff + i, tr = troff + i;
            V.S[r] = sh.S0[tr]; V.Slen[r] = sh.Slen0[tr];
            V.D[r] = MIN_SCORE; V.Dlen[r] = 0;
            V.Sn[r] = sh.Sn0[tr]; V.SnLen[r] = sh.Slen0[tr]; V.Ly[r] = sh.SnSet0[tr] ? n : 0u;
            V.SmoveF[r] = TB_NONE; V.ImoveF[r] = TB_NONE;
        }
    }
    for (uint32_t k = threadIdx.x; k < nact; k += blockDim.x) {
        uint32_t c = V.act[k];
        const uint32_t trm = V.cd[c].troff + V.cd[c].m - 1;
        s_act[k] = c;
        base[0][c] = sh.base0[c];
        rowm_mv[c] = sh.Smove0[trm]; rowm_S[c] = sh.S0[trm]; rowm_len[c] = sh.Slen0[trm];
        V.Lx[(size_t)c * (n + 1)] = sh.lx0[c];
    }
    for (uint32_t c = threadIdx.x; c < V.C; c += blockDim.x) s_opp[c] = V.opp_act[c];
    __syncthreads();

    int32_t sn0; uint32_t ly0; row0_init_sn(P, n, sn0, ly0);
    Row0 r0prev = row0_column0();

    for (uint32_t j = 1; j <= n; ++j) {
        const int cur = j & 1, prv = cur ^ 1;
        const Row0 r0 = row0_step(P, j, n, sn0, ly0);
        const uint8_t q = V.y[j - 1];
        int32_t go_j = P.gap_open + P.gap_extend * (int32_t)j;
        const int32_t xclip_score = P.xclip_prefix + (P.yclip_prefix > go_j ? P.yclip_prefix : go_j);   // :304-308
        uint8_t* tbcol = V.tb + (size_t)(j - 1) * Rtot;

        for (uint32_t k = wave; k < nact; k += W) {
            const uint32_t c = s_act[k];
            const ContigDesc cd = V.cd[c];
            const uint32_t m = cd.m, roff = cd.roff;
            const uint8_t* xs = V.xseq + cd.seqoff;

            ColCtx cx;
            cx.j = j; cx.n = n; cx.m = m; cx.cidx = c; cx.q = q; cx.xclip_score = xclip_score; cx.row0_len = r0.Slen;
            cx.jump = select_jump(P, base[prv], s_act, nact, c, s_opp[c]);
            cx.circ_ok = (P.circular && rowm_mv[c] != TB_XCLIP_SUFFIX) ? 1 : 0;
            cx.circ_score = rowm_S[c];
            cx.circ_len = rowm_len[c] + 1;
            if (lane == 0) {
                V.jt_idx[(size_t)c * (n + 1) + j] = cx.jump.idx;
                V.jt_from[(size_t)c * (n + 1) + j] = cx.jump.from;
            }

            // carries between tiles (wave-uniform)
            int32_t upS = r0prev.S; uint32_t upSl = r0prev.Slen;       // S[prev][i0-1], cell(i0-1,j-1).S.len
            int32_t upT = r0.S; uint32_t upTl = r0.Slen;               // S[curr][i0-1] w/o insertion, its length
            ScanEl carry = scan_seed();
            XsRec xbest; xbest.v = MIN_SCORE; xbest.len = 0; xbest.row = 0;          // S[curr][m] starts at MIN, len 0 (:236-238)
            CmRec cbest; cbest.v = r0.S; cbest.row = 0; cbest.len = r0.Slen;          // get_jump_info starts at row 0
            // row m is finalised after the x-suffix reduction
            int32_t ownS = 0, ownDg = 0, ownI = 0; uint32_t ownSl = 0, ownMv = 0, ownBits = 0, ownIl = 0; int32_t ownSn = 0;
            const uint32_t ntiles = (m + 64 * R - 1) / (64 * R);
            const int owner_lane = (int)(((m - 1) / R) & 63);

            for (uint32_t t = 0; t < ntiles; ++t) {
                const uint32_t i0 = t * 64 * R + lane * R + 1;
                const uint32_t r = roff + i0 - 1;
                int32_t Sp[R], Dp[R], Snv[R]; uint32_t Slp[R], Dlp[R]; uint8_t xb[R];
#pragma unroll
                for (int u = 0; u < R; ++u) { Sp[u] = V.S[r + u]; Slp[u] = V.Slen[r + u]; Dp[u] = V.D[r + u]; Dlp[u] = V.Dlen[r + u]; Snv[u] = V.Sn[r + u]; xb[u] = xs[i0 - 1 + u]; }
                int32_t nS = shfl_up_i(Sp[R - 1], 1); uint32_t nSl = shfl_up_u(Slp[R - 1], 1);
                if (lane == 0) { nS = upS; nSl = upSl; }
                upS = bcast_i(Sp[R - 1], 63); upSl = bcast_u(Slp[R - 1], 63);

                RowA ra[R];
                int32_t Dn[R]; uint32_t Dln[R];
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const uint32_t i = i0 + u;
                    row_phase_a(P, cx, i <= m ? i : 1u, xb[u], u == 0 ? nS : Sp[u - 1], u == 0 ? nSl : Slp[u - 1], Sp[u], Slp[u], Dp[u], Dlp[u],
                                sh.Slen0 + cd.troff, ra[u]);
                    Dn[u] = ra[u].bd; Dln[u] = ra[u].dlen;
                }
                // insertion scan (phase B)
                int32_t nT = shfl_up_i(ra[R - 1].T, 1); uint32_t nTl = shfl_up_u(ra[R - 1].Tl, 1);
                if (lane == 0) { nT = upT; nTl = upTl; }
                upT = bcast_i(ra[R - 1].T, 63); upTl = bcast_u(ra[R - 1].Tl, 63);
                ScanEl el[R];
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const uint32_t i = i0 + u;
                    el[u] = scan_make(P, i, u == 0 ? nT : ra[u - 1].T, u == 0 ? nTl : ra[u - 1].Tl);
                    if (i > m) el[u].key = KEY_NEG_INF;
                }
                ScanEl agg = el[0];
#pragma unroll
                for (int u = 1; u < R; ++u) agg = scan_combine(agg, el[u]);
                ScanEl inc = agg;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    ScanEl o; o.key = shfl_up_i(inc.key, d); o.q = shfl_up_i(inc.q, d);
                    if (lane >= d) inc = scan_combine(o, inc);
                }
                ScanEl run; run.key = shfl_up_i(inc.key, 1); run.q = shfl_up_i(inc.q, 1);
                run = (lane == 0) ? carry : scan_combine(carry, run);
                {
                    ScanEl last; last.key = bcast_i(inc.key, 63); last.q = bcast_i(inc.q, 63);
                    carry = scan_combine(carry, last);
                }
                // phase C
                int32_t So[R]; uint32_t Slo[R]; uint32_t code[R]; int32_t Io[R]; uint32_t Ilo[R];
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const uint32_t i = i0 + u;
                    const bool valid = i <= m;
                    const uint32_t ext = run.key >= el[u].key ? 1u : 0u;
                    if (!ext) run = el[u];
                    const int32_t bi = run.key + P.gap_extend * (int32_t)i;
                    const uint32_t il = (uint32_t)(run.q + (int32_t)i);
                    uint32_t mv;
                    row_phase_c(P, cx, valid ? i : 1u, ra[u], bi, il, sh.Slen0 + cd.troff, So[u], Slo[u], mv);
                    code[u] = mv | (ext ? TBB_IEXT : 0u) | (ra[u].dext ? TBB_DEXT : 0u);
                    Io[u] = bi; Ilo[u] = il;
                    if (valid && i < m) {
                        XsRec xc; xc.v = So[u] + P.xclip_suffix; xc.len = Slo[u]; xc.row = i;
                        if (xs_better(xc, xbest)) xbest = xc;
                        CmRec cc; cc.v = So[u]; cc.row = i; cc.len = Slo[u];
                        if (cm_better(cc, cbest)) cbest = cc;
                        // y suffix tracking (:431-447); cell(i,n) still holds its initial length 0 for i < m
                        const int32_t v = So[u] + P.yclip_suffix;
                        if (v > Snv[u] || (v == Snv[u] && Slo[u] > 0u)) { V.Sn[r + u] = v; V.Ly[r + u] = n - j; V.SnLen[r + u] = Slo[u]; }
                    }
                    if (valid && i == m) {
                        ownS = So[u]; ownSl = Slo[u]; ownMv = mv; ownBits = code[u] & (TBB_IEXT | TBB_DEXT); ownDg = ra[u].dg;
                        ownI = bi; ownIl = il; ownSn = Snv[u];
                    }
                    if (!valid) { So[u] = MIN_SCORE; Slo[u] = 0; code[u] = 0; Dn[u] = MIN_SCORE; Dln[u] = 0; Io[u] = MIN_SCORE; Ilo[u] = 0; }
                }
                // stores (row m's S/len/byte are patched after the reduction)
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const uint32_t i = i0 + u;
                    V.D[r + u] = Dn[u]; V.Dlen[r + u] = Dln[u];
                    if (i != m) { V.S[r + u] = So[u]; V.Slen[r + u] = Slo[u]; tbcol[r + u] = (uint8_t)code[u]; }
                    if (j == n) { V.Ival[r + u] = Io[u]; V.Ilen[r + u] = Ilo[u]; }
                }
            }

            // ---- reductions over the wave: x-suffix running max and column arg-max (rows < m) ---------------
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                // total order (value, len, lower row); the initial record has row 0 and therefore survives full ties,
                // which is the reference's "no update on equality" (:408-417)
                XsRec o; o.v = __shfl_xor(xbest.v, d, 64); o.len = (uint32_t)__shfl_xor((int)xbest.len, d, 64); o.row = (uint32_t)__shfl_xor((int)xbest.row, d, 64);
                if (xs_better(o, xbest)) xbest = o;
                CmRec p; p.v = __shfl_xor(cbest.v, d, 64); p.row = (uint32_t)__shfl_xor((int)cbest.row, d, 64); p.len = (uint32_t)__shfl_xor((int)cbest.len, d, 64);
                if (cm_better(p, cbest)) cbest = p;
            }
            // ---- row m (:350-351 seeded selection, :406-447 for i == m) ---------------------------------------
            {
                const uint32_t rm = roff + m - 1;
                int32_t Sm; uint32_t Slm, mvm, lx;
                lx = xbest.row == 0 ? 0u : m - xbest.row;
                bool do_x_m = false;
                if (rowm_run_wins(xbest.v, ownS, ownDg)) { Sm = xbest.v; Slm = xbest.len; mvm = MV_XSUF; }
                else {
                    Sm = ownS; Slm = ownSl; mvm = ownMv;
                    if (P.xclip_suffix == 0 && ownSl > xbest.len) { do_x_m = true; lx = 0; }
                }
                if (lane == owner_lane) {
                    V.S[rm] = Sm; V.Slen[rm] = Slm; tbcol[rm] = (uint8_t)(mvm | ownBits);
                    const uint32_t rl = (j == n) ? (do_x_m ? ownSl : xbest.len) : 0u;
                    const int32_t v = Sm + P.yclip_suffix;
                    if (v > ownSn || (v == ownSn && Slm > rl)) { V.Sn[rm] = v; V.Ly[rm] = n - j; V.SnLen[rm] = Slm; }
                    V.Lx[(size_t)c * (n + 1) + j] = lx;
                }
                Sm = bcast_i(Sm, owner_lane); Slm = bcast_u(Slm, owner_lane); mvm = bcast_u(mvm, owner_lane);
                CmRec cc; cc.v = Sm; cc.row = m; cc.len = Slm;
                if (cm_better(cc, cbest)) cbest = cc;
                if (lane == 0) {
                    JumpBase b; b.score = cbest.v; b.len = cbest.len + 1; b.from = cbest.row;
                    base[cur][c] = b;
                    uint32_t refmv = (mvm == MV_XSUF) ? TB_XCLIP_SUFFIX : (mvm == MV_INS) ? TB_INS : (mvm == MV_DEL) ? TB_DEL
                                   : (mvm == MV_XPRE) ? TB_XCLIP_PREFIX : (mvm == MV_YPRE) ? TB_YCLIP_PREFIX : TB_MATCH;
                    rowm_mv[c] = (uint8_t)refmv; rowm_S[c] = Sm; rowm_len[c] = Slm;
                }
            }
        }
        r0prev = r0;
        __syncthreads();
    }
}

void launch_fill(const JobView* d_jobs, uint32_t n_jobs, int waves, const FillShared& sh, hipStream_t stream) {
    hipLaunchKernelGGL(fill_kernel<4>, dim3(n_jobs), dim3(waves * 64), 0, stream, d_jobs, sh);
}

// Kernel 2 — last-column fix-ups, then the traceback walk(s).  One wavefront per read: lanes run the serial
// fix-up of one contig each (single_contig_aligner.rs:453-555), then lanes walk (traceback/mod.rs:219-373).
// mode: 0 = traceback (best end contig), 1 = one chain per active contig (traceback_all candidates, chosen among
// on the host in the reference's order), 2 = traceback_from(from).
// (WaveWalk, the wavefront-wide execution of one walk, lives in walk_core.h: fill_regs.hip's persistent teams run it too)

__global__ __launch_bounds__(64) void fixup_walk_kernel(const JobView* __restrict__ jobs, const WalkArgs* __restrict__ args, uint32_t fixups_done) {
    const JobView& V = jobs[blockIdx.x];
    const WalkArgs A = args[blockIdx.x];
    const int lane = threadIdx.x;
    if (!A.skip_fixup && !fixups_done) for (uint32_t k = lane; k < V.nact; k += 64) fixup_contig(V, V.act[k]);
    __syncthreads();
    if (A.mode == 1) {
        // traceback_all candidates: one chain per active contig, walked by walk_all_kernel (one wavefront each).  This kernel walks the
        // REFERENCE chain — the one from the best end contig — and records where it enters every column; the others stop where they meet
        // it (walk_core.h, VisitRec)
        if (V.visit != nullptr) {
            const uint32_t r = pick_primary(V);
            uint32_t kr = 0;
            for (uint32_t k = 0; k < V.nact; ++k) if (V.act[k] == r) kr = k;
            for (uint32_t e = (uint32_t)lane; e < V.n + 2; e += 64) V.visit[e] = JoinRole::cleared(e, kr);
            __syncthreads();
            ChainHdr H;
            WaveWalk ex; ex.lane = lane; ex.role = 1; ex.ref_slot = kr;
            walk_from_t(V, r, H, A.ops + (size_t)kr * A.ops_cap, A.ops_cap, ex);
            if (lane == 0) A.hdr[kr] = H;
        }
    } else {
        const uint32_t c = A.mode == 0 ? pick_primary(V) : A.from;
        ChainHdr H;
        WaveWalk ex; ex.lane = lane;
        walk_from_t(V, c, H, A.ops, A.ops_cap, ex);
        if (lane == 0) A.hdr[0] = H;
    }
}

// the last-column fix-ups alone, 64 contigs per workgroup (one lane each): `chunks` workgroups per job
__global__ __launch_bounds__(64) void fixup_only_kernel(const JobView* __restrict__ jobs, const WalkArgs* __restrict__ args, uint32_t chunks) {
    const uint32_t job = blockIdx.x / chunks, k = (blockIdx.x % chunks) * 64u + threadIdx.x;
    const JobView& V = jobs[job];
    if (!args[job].skip_fixup && k < V.nact) fixup_contig(V, V.act[k]);
}

// mode 1 (--suboptimal): the walk from every active contig's end cell, one wavefront per (job, contig).  Launched with one wavefront per
// workgroup and one workgroup per walk, or (persistent teams resident: the walks only get the wave slots the fill left free, a few
// half-empty CUs) with a handful of workgroups of WALK_WAVES wavefronts that stride over the walks.
constexpr uint32_t WALK_WAVES = 8;         // two per SIMD beside one resident fill wave (96 registers each, no LDS)
__global__ __launch_bounds__(64 * WALK_WAVES) void walk_all_kernel(const JobView* __restrict__ jobs, const WalkArgs* __restrict__ args, uint32_t stride, uint32_t total) {
    const uint32_t waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
    for (uint32_t item = blockIdx.x * waves + wave; item < total; item += gridDim.x * waves) {
        const uint32_t job = item / stride, k = item % stride;
        const JobView& V = jobs[job];
        const WalkArgs A = args[job];
        if (A.mode != 1 || k >= V.nact) continue;
        ChainHdr H;
        WaveWalk ex; ex.lane = threadIdx.x & 63;
        if (V.visit != nullptr) {
            const VisitRec sum = V.visit[0];
            if (k == sum.contig) continue;                 // the reference chain: walked by fixup_walk_kernel
            if (sum.row == 1) { ex.role = 2; ex.ref_hdr = A.hdr + sum.contig; }
        }
        walk_from_t(V, V.act[k], H, A.ops + (size_t)k * A.ops_cap, A.ops_cap, ex);
        if ((threadIdx.x & 63) == 0) A.hdr[k] = H;
    }
}

// max_wgs == 0: as many workgroups as there are jobs / walks.  max_wgs > 0 (persistent teams are resident: stitch_api.cpp
// run_jobs_streaming): no launch has more than max_wgs workgroups — the fix-ups run 64 contigs per workgroup in a launch of their own
// (the caller keeps n_jobs x ceil(max_nact / 64) within max_wgs), the walks of --suboptimal stride over their (job, contig) pairs.
void launch_fixup_walk(const JobView* d_jobs, const WalkArgs* d_args, uint32_t n_jobs, uint32_t max_nact_mode1, hipStream_t stream, uint32_t max_wgs, uint32_t max_nact) {
    const uint32_t chunks = (max_nact + 63u) / 64u;
    if (max_wgs == 0) {
        // (more than 64 contigs: the fix-ups, one lane per contig and serial down its rows, in workgroups of 64 contigs side by side rather
        // than in rounds of 64 within the read's one workgroup: cfg5's 200 contigs took four rounds, 75 ms per launch)
        if (chunks > 1) hipLaunchKernelGGL(fixup_only_kernel, dim3(n_jobs * chunks), dim3(64), 0, stream, d_jobs, d_args, chunks);
        hipLaunchKernelGGL(fixup_walk_kernel, dim3(n_jobs), dim3(64), 0, stream, d_jobs, d_args, chunks > 1 ? 1u : 0u);
        if (max_nact_mode1) hipLaunchKernelGGL(walk_all_kernel, dim3(n_jobs * max_nact_mode1), dim3(64), 0, stream, d_jobs, d_args, max_nact_mode1, n_jobs * max_nact_mode1);
        return;
    }
    if (chunks > 1) {
        hipLaunchKernelGGL(fixup_only_kernel, dim3(n_jobs * chunks), dim3(64), 0, stream, d_jobs, d_args, chunks);
        hipLaunchKernelGGL(fixup_walk_kernel, dim3(n_jobs), dim3(64), 0, stream, d_jobs, d_args, 1u);
    }
    else hipLaunchKernelGGL(fixup_walk_kernel, dim3(n_jobs), dim3(64), 0, stream, d_jobs, d_args, 0u);
    if (max_nact_mode1) { const uint32_t total = n_jobs * max_nact_mode1; const uint32_t wgs = (total + WALK_WAVES - 1) / WALK_WAVES; hipLaunchKernelGGL(walk_all_kernel, dim3(wgs < max_wgs ? wgs : max_wgs), dim3(64 * WALK_WAVES), 0, stream, d_jobs, d_args, max_nact_mode1, total); }
}

}  // namespace stitch
