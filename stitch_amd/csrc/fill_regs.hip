// Kernel 1c — the Local-mode DP fill with the row state RESIDENT IN REGISTERS, one contig per wavefront.
//
// Same recurrence, tie-breaks and outputs as fill_local16.hip (fg-stitch-lib/src/align/aligners/single_contig_aligner.rs:188-451
// + multi_contig_aligner.rs:264-347; dp_core.h holds the word arithmetic both kernels execute).  What differs:
//   * ROWS -> LANES.  A wave owns one contig of up to 64 x RMAX rows; lane l holds a run of CONSECUTIVE rows, a multiple of four
//     (S word and D word of every row in two VGPRs: 2 x RMAX = 160 of the wave's 256 registers) for the whole read.  The row
//     state never touches memory; per column a wave writes its traceback bytes and y-suffix records and nothing else
//     (fill_local16 streams 16 B of state per cell through L2 and the fabric).
//   * One insertion scan per CONTIG and column instead of one per 256-row tile: the chain I[i] = max(I[i-1]+ge, S'[i-1]+go+ge)
//     runs serially down a lane's rows (a compare and two selects per row) and crosses lanes once, as a lane-tagged DPP prefix
//     maximum.  The per-tile overheads of the tiled kernels (scan, reductions, slot bookkeeping) go.
//   * Registers 4g+3 .. 4g of a lane = four consecutive rows ("group" g), top to bottom.  The contig's ceil(m / 4) groups are
//     dealt to the lanes as evenly as possible: the first lanes hold one group more than the others, and EVERY lane's first row
//     is in the fullest lane's top register — a lane with a group less ends in register 4, group 0 is not its own.  So the
//     unrolled code of the groups g >= 1 runs under a scalar condition and with every lane enabled; only group 0 is per lane.
//   * A read's contigs are dealt to the waves of G workgroups (a team); per column the team exchanges the per-contig column
//     arg-max (the next column's jump sources, get_jump_info :677-697) through one 8-byte granule {column, score, len, from} per
//     contig and column parity, written by the wave that owns the contig and polled by every wave on its own: there is no
//     workgroup barrier in the column loop, and the polled records stay in registers.
//   * y-suffix records (:431-447) are kept for cells that reach the best score seen so far in ANY contig (mode traceback) or in
//     their own contig (traceback_all / traceback_from): DESIGN.md "y-suffix records".
// Eligibility (stitch_api.cpp regs_plan): what fill_local16 admits, and: every contig <= 64 x RMAX rows, gap penalties small
// enough for 16-bit insertion-chain words and lane-tagged scan keys.  Everything else runs fill_local16 / the generic kernel.
// Traceback bytes and y-suffix records are stored lane-interleaved (walk_core.h tb_row_offset, V.tb_keyfmt == 2) so that every
// store instruction writes whole 256-byte lines.  All spins are bounded and end the kernel with an error word, never a hang.
#include <type_traits>
#include <hip/hip_runtime.h>
#include "dp_core.h"
#include "walk_core.h"
#include "fill_common.h"

namespace stitch {
namespace {

constexpr int RMAX = REGS_RMAX;                // rows per lane (walk_core.h: the traceback layout depends on it)
constexpr int NG = RMAX / 4;                   // groups of four rows: one traceback dword, one word of bases
static_assert(RMAX == 80 || RMAX == 40, "the group blocks below are written out for 20 (or, in experiment builds, 10) groups");
constexpr uint32_t RSRC_WORD3 = 0x00020000u;   // raw buffer descriptor, gfx94x / gfx950
constexpr int AUX_NT = 2;                      // streamed once: non-temporal
constexpr int AUX_SC1 = 16;                    // agent-scope coherence (what an agent-scope atomic load carries on gfx94x / gfx950)
constexpr int AUX_VOLATILE = (int)0x80000000u; // compiler-side: never merged, hoisted or dropped
#ifndef STITCH_POLL_SLEEP
#define STITCH_POLL_SLEEP 4                    // x 64 clocks between two looks at the granules
#endif

// In-kernel stamps (diagnostic build only, -DSTITCH_PROFILE): per-wave cycle sums of the column loop's sections, added up per read
// in the debug area behind V.err.  Never enabled in the product build.
#ifdef STITCH_PROFILE
#define RPROF_DECL uint32_t pf_t = (uint32_t)__builtin_readcyclecounter(), pf_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pf_cnt[2] = {0, 0}, pf_cls[6] = {0, 0, 0, 0, 0, 0}, pf_c0 = 0, pf_p0 = 0, pf_p1 = 0;
#define RPROF(k) { const uint32_t pf_n = (uint32_t)__builtin_readcyclecounter(); pf_sum[k] += pf_n - pf_t; pf_t = pf_n; }
#define RCOUNT(k) pf_cnt[k] += 1u;
#else
#define RPROF_DECL
#define RPROF(k)
#define RCOUNT(k)
#endif

// per wave in LDS: NG x 64 words of bases (constant), NG x 64 words of traceback codes (pass 1 -> pass 2), RMAX x 64 scores of
// best{diagonal, deletion} (pass 1 -> the insertion merge of pass 2)
constexpr uint32_t LDS_XW = 0, LDS_TB = NG * 64 * 4, LDS_BS = 2 * NG * 64 * 4, LDS_PER_WAVE = 2 * NG * 64 * 4 + RMAX * 64 * 2;

// The insertion chain I[i] = max(I[i-1] + ge, S'[i-1] + go + ge) (single_contig_aligner.rs:314-326; S' = S without its own
// insertion candidate, dp_core.h phase B) is carried as a WORD like S and D (score << 16 | length): one step down a lane is
//   ext = I + GE1, open = S'(row above) + GO1, I = score(ext) >= score(open) ? ext : open     (the extension wins ties, :321)
// i.e. two adds, a compare of the high halves and a select.  Every row has an opener with score >= go + ge (S' >= 0), so the
// chain's score stays >= go + ge + ge and the 16-bit field cannot wrap (regs_plan: ge >= -1024, go + ge >= -8000).
//
// Two passes per column.  Pass 1 runs the chain L of the lane's OWN openers (what arrives from the lanes above is not known
// yet) and already merges it into the cells.  L's value behind the lane's last row crosses the lanes as a position-normalised
// key in a lane-tagged DPP prefix maximum, which gives every lane X = the chain arriving at its first row.  The true chain is
//   X carried down (score + ge, length + 1 per row) as long as no opener has beaten it STRICTLY, and L from that row on:
// the extension wins ties, so X survives until an opener is strictly better; at that row L is that very opener (it beats
// L's own extension, which is <= X's), and from there both recurrences are the same.  Pass 2 therefore visits a lane's rows
// from the top only WHILE X IS ALIVE in some lane of the wave — in most contigs and columns a few rows; only below a
// high-scoring path the chain lives on for score / |ge| rows — and repairs there what pass 1 assumed: the "extended" bit (an
// alive X arrived by extension) and the merge (with X instead of L; when X changes a cell, the result does not depend on
// what L did to it: row_alive).  An opener taken from a cell that the insertion itself produced never beats the extension
// (go <= 0: regs_plan), so both passes may open from the row's final word.
constexpr int32_t CHAIN_NONE = (int32_t)0x92A00000u;      // word(-28000, 0): "no chain yet", cannot wrap when extended
constexpr int32_t SUP_NONE = (int32_t)0xA2400000u;        // word(-24000, 0): "no row above": its opener (>= -32000) is below every X (>= -28000)

__device__ __forceinline__ int32_t chain_step(const int32_t I, const int32_t Tabove, const int32_t GE1, const int32_t GO1) {
    const int32_t ext = I + GE1, open = Tabove + GO1;
    return word_score(ext) >= word_score(open) ? ext : open;
}

// wave-uniform values of one contig's column, and the lane's rolling values of the row loop of pass 1
struct Col {
    int32_t MW1, XW1, GE1, GO1, JSWm1;         // (match, mismatch) << 16 | 1; gap words; the column's jump word minus one length unit
    uint32_t q;                                // y[j-1]
    int32_t aw1, DG;                           // the current row's (match | mismatch) word and diagonal candidate: old S word of the row above + aw1
    int32_t dgm; uint32_t pad;                 // row m (register `pad` of its lane): its diagonal candidate, needed for its finalisation
    int32_t jfix;                              // circular contigs: what row 1's jump word has over the column's (lane 0, consumed by its first row)
};

// ---- pass 1, one row (register IDX): everything of the cell that needs column j-1 only (dp_core.h row_phase_a_word), written
// in place; the score of best{diagonal, deletion} is parked in LDS for the insertion merges
// xcur = the four bases of the row's own group, xnext = those of the group below it, read from LDS at the group's first row.  The
// two change roles from group to group (the caller passes them by the group's parity): no copy when a group ends.
// The x-prefix clip (:383-389: a negative cell becomes score 0, length 0, move MK_XPRE) is NOT applied here: when the column's
// jump word plus a mismatch is not negative, no cell of the column is (every cell can take the jump) — nearly every column of a
// read that aligns at all.  In the other columns clip_row repairs a group's four cells behind its rows (one scalar branch per group).
template <int IDX, bool CIRC>
__device__ __forceinline__ void row_pass1(uint32_t& Sreg, uint32_t& Dreg, uint32_t& tbw, Col& c, const uint32_t xcur, uint32_t& xnext, const uint32_t* xw_lane, uint16_t* bs_lane) {
    constexpr int k = IDX & 3;
    const int32_t aw1 = c.aw1, DG = c.DG;                    // prepared by the row above (or the column's prologue)
    const int32_t Sold = (int32_t)Sreg, Dold = (int32_t)Dreg;
    if (IDX > 0) {
        // the NEXT row's diagonal candidate takes this row's old S word: computed here, so that the old word is dead before the
        // new one is written and the row's S register is updated in place
        if (k == 3 && IDX >= 7) xnext = xw_lane[((IDX >= 7 ? IDX - 7 : 0) >> 2) * 64];      // (used by the group's last row: no LDS latency there)
        const uint32_t xbn = ((k == 0 ? xnext : xcur) >> (8 * (k == 0 ? 3 : k - 1))) & 0xFFu;
        c.aw1 = xbn == c.q ? c.MW1 : c.XW1;
        c.DG = Sold + c.aw1;                                 // diagonal: score + a, length + 1
    }
    const int32_t DE = Dold + c.GE1, DO = Sold + c.GO1;
    const bool dext = word_score(DE) >= word_score(DO);      // the extension wins ties (:332)
    const int32_t BD = dext ? DE : DO;
    const bool c1 = word_score(BD) > word_score(DG);         // deletion strictly better than the diagonal
    const int32_t bs2 = c1 ? BD : DG;
    const int32_t X = c1 ? (BD | 0xFFFF) : DG;               // what the jump has to beat (:373-382)
    int32_t JW = aw1 + c.JSWm1;
    if (CIRC && k == 3) { JW += c.jfix; c.jfix = 0; }       // (a lane's first row is byte 3 of its top group; only row 1 of the contig has a non-zero term)
    const bool c3 = JW > X;
    const int32_t T = c3 ? JW : bs2;
    const uint32_t code = (c3 ? MK_JUMP : c1 ? MK_DEL : MK_DIAG) | (dext ? (uint32_t)TBB_DEXT : 0u);
    tbw = k == 3 ? code : ((tbw << 8) | code);               // byte k of the group's dword: register 4g+3 first, 4g last
    bs_lane[IDX * 64] = (uint16_t)((uint32_t)bs2 >> 16);
    if (IDX < 8) { if (c.pad == (uint32_t)IDX) c.dgm = DG; }           // (c.pad: the register of row m, one of the first eight)
    Sreg = (uint32_t)T; Dreg = (uint32_t)BD;
}


template <int IDX>
__device__ __forceinline__ void clip_row(uint32_t& Sreg, uint32_t& tbw) {
    constexpr int k = IDX & 3;
    const unsigned long long c4 = __ballot((int32_t)Sreg < 0);
    const uint32_t ntb = (tbw & ~(7u << (8 * k))) | ((uint32_t)MK_XPRE << (8 * k));
    // (IN PLACE, as in merge_row)
    asm("v_max_i32 %0, 0, %0" : "+v"(Sreg));
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(tbw) : "v"(ntb), "s"(c4));
}

// ---- pass 1b: the chain L of the lane's own openers.  chain_row steps it over one row (the chain arriving at the row is kept
// for the merge, its "extended" bit goes into the group's traceback dword); merge_row merges a chain word into the cell where it
// changes it (dp_core.h row_phase_c_word: beats best{diagonal, deletion}, is not beaten by the jump).  The chain opens from S
// WITHOUT its own insertion candidate (dp_core.h, phase B), i.e. from the words as pass 1 left them: a group's four steps are
// taken first, then ONE test whether any of its cells has to be merged (a branch per row costs more than the row's arithmetic).
// (The chain's values at column n are an output, the fix-up kernel reads them: the kernel recomputes them from the final words
// after the column loop rather than carry a test for the last column through every row of every column.)
struct Col2 {
    int32_t GE1, GO1;
    int32_t Iw;                                // the insertion chain's word AT the current row ...
    uint32_t extn;                             // ... and TBB_IEXT if it got there by an extension (else 0)
};
template <int IDX>
__device__ __forceinline__ int32_t chain_row(const uint32_t Sreg, uint32_t& tbw, Col2& c) {
    constexpr int k = IDX & 3;
    const int32_t Iw = c.Iw;
    tbw |= c.extn << (8 * k);
    const int32_t ext = Iw + c.GE1, open = (int32_t)Sreg + c.GO1;
    const bool isext = word_score(ext) >= word_score(open);  // the extension wins ties (:321)
    c.Iw = isext ? ext : open;
    c.extn = isext ? (uint32_t)TBB_IEXT : 0u;                // "I extended" is a property of the NEXT row's cell
    return Iw;
}
// The merge of a chain word I into a cell whose score it reaches (m: score(I) >= score(S), S >= 0) is short.  Written out,
// row_phase_c_word with bi = score(I) >= score(S): the clamp does nothing (bi >= 0), BI = I; the jump cannot beat it (c5 false:
// the jump's score is <= the cell's, which took the jump if it was better, <= bi); the result is not negative (c6 false).  So
//   c2 = score(I) > score(best{diagonal, deletion})  ?  (I, MK_INS)  :  the cell as it is
// and for a cell the test does not admit, row_phase_c_word changes nothing (c2 false, or the jump / the clip win again).
// The score of best{diagonal, deletion} comes from LDS, where pass 1 parked it.
template <int IDX>
__device__ __forceinline__ void merge_row(uint32_t& Sreg, uint32_t& tbw, const unsigned long long m, const int32_t Iw, const uint16_t* bs_lane) {
    constexpr int k = IDX & 3;
    const int32_t bs = (int32_t)(int16_t)bs_lane[IDX * 64];
    const unsigned long long c2 = m & __ballot(word_score(Iw) > bs);
    const uint32_t ntb = (tbw & ~(7u << (8 * k))) | ((uint32_t)MK_INS << (8 * k));
    // (selected IN PLACE: the register allocator otherwise gives the merged value a register of its own and copies every row's
    // word there and back on the path that does not merge)
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(Sreg) : "v"(Iw), "s"(c2));
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(tbw) : "v"(ntb), "s"(c2));
}

// ---- pass 2, one group in which the chain X that arrived at the lane's first row is still alive in some lane.  The lanes'
// states are kept as masks in scalar registers (ballots), the selects take them as they are: no branch but the one for a merge,
// and that once per group — the openers may be taken from the words as pass 1b left them: a cell that X changes, and a cell
// whose score X reaches without changing it, both open below X's own extension.
struct ColA {
    int32_t GE1, GO1;
    int32_t X;                                 // the chain carried down to the group's first row
    int32_t Sup;                               // word of the row above (SUP_NONE above the lane's first row)
    uint32_t xext;                             // TBB_IEXT if X got to the group's first row by an extension (a lane's first row: from the scan)
    unsigned long long alive;                  // lanes in which no opener has strictly beaten X so far
};
__device__ __forceinline__ uint32_t select_lanes(const unsigned long long lanes, const uint32_t v) {      // v in the lanes of the mask, 0 elsewhere
    uint32_t r; asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(lanes)); return r;
}
template <int G>
__device__ __forceinline__ void group_alive(uint32_t& s3, uint32_t& s2, uint32_t& s1, uint32_t& s0, uint32_t& tbw, ColA& c, const uint16_t* bs_lane) {
    const int32_t x3 = c.X, x2 = x3 + c.GE1, x1 = x2 + c.GE1, x0 = x1 + c.GE1;
    const unsigned long long a3 = c.alive & ~__ballot(word_score(c.Sup + c.GO1) > word_score(x3));
    const unsigned long long a2 = a3 & ~__ballot(word_score((int32_t)s3 + c.GO1) > word_score(x2));
    const unsigned long long a1 = a2 & ~__ballot(word_score((int32_t)s2 + c.GO1) > word_score(x1));
    const unsigned long long a0 = a1 & ~__ballot(word_score((int32_t)s1 + c.GO1) > word_score(x0));
    // an alive X arrived by an extension (a lane's first row, byte 3 of its top group: as the scan says); pass 1b left the bit of
    // the lane's own chain there: 0 in a first row
    tbw |= select_lanes(a3, c.xext << 24) | select_lanes(a2, (uint32_t)TBB_IEXT << 16) | select_lanes(a1, (uint32_t)TBB_IEXT << 8) | select_lanes(a0, (uint32_t)TBB_IEXT);
    // X reaches a cell's score: if it beats best{diagonal, deletion} the cell becomes X whatever pass 1b made of it; if it does
    // not, the lane's own chain (score <= X's) did not either, and the cell is as pass 1 left it (merge_row)
    const unsigned long long m3 = a3 & __ballot(word_score(x3) >= word_score((int32_t)s3)), m2 = a2 & __ballot(word_score(x2) >= word_score((int32_t)s2));
    const unsigned long long m1 = a1 & __ballot(word_score(x1) >= word_score((int32_t)s1)), m0 = a0 & __ballot(word_score(x0) >= word_score((int32_t)s0));
    c.Sup = (int32_t)s0;
    if (__builtin_expect((m3 | m2 | m1 | m0) != 0ull, 0)) {
        merge_row<4 * G + 3>(s3, tbw, m3, x3, bs_lane); merge_row<4 * G + 2>(s2, tbw, m2, x2, bs_lane);
        merge_row<4 * G + 1>(s1, tbw, m1, x1, bs_lane); merge_row<4 * G>(s0, tbw, m0, x0, bs_lane);
    }
    c.X = x0 + c.GE1; c.xext = (uint32_t)TBB_IEXT; c.alive = a0;
}

// the lane's running records over a contig's column (rows below m): the largest S word and the group of four rows that held it
// first; the group that first held the largest score.  Which row of which lane that is, the epilogue works out for the one
// lane that matters (x-suffix running max :406-429: largest word, topmost row; column arg-max :677-697: largest score, topmost row)
struct Recs { uint32_t bw, gw, g1; };

__device__ __forceinline__ void group_records(Recs& R, const uint32_t g4, const uint32_t g) {
    R.gw = g4 > R.bw ? g : R.gw;
    R.g1 = (g4 >> 16) > (R.bw >> 16) ? g : R.g1;
    R.bw = g4 > R.bw ? g4 : R.bw;
}

#if STITCH_REGS_RMAX == 80
#define REP20(X) X(19) X(18) X(17) X(16) X(15) X(14) X(13) X(12) X(11) X(10) X(9) X(8) X(7) X(6) X(5) X(4) X(3) X(2) X(1) X(0)
#else
#define REP20(X) X(9) X(8) X(7) X(6) X(5) X(4) X(3) X(2) X(1) X(0)
#endif
#ifndef STITCH_REGS_WAVES_PER_EU
#define STITCH_REGS_WAVES_PER_EU 2
#endif

}  // namespace

// NQ = granule registers per lane: 1 for up to 64 active contigs, 4 for up to 256
// YB: every job of the launch keeps per-contig y-suffix records (--suboptimal, the re-alignments of --circular): a cell whose word is the
// column's common word W = jump word + match is recorded as bit 7 of its traceback byte instead of an 8-byte record (see P2TAIL)
template <int NQ, bool CIRC, bool YB>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(STITCH_REGS_WAVES_PER_EU, STITCH_REGS_WAVES_PER_EU))) void fill_regs_kernel(const JobView* __restrict__ jobs, FillShared sh, const uint2* __restrict__ wave_map, uint32_t n_waves, const StreamCtl* __restrict__ qp) {
    // The launch's waves are dealt to the reads' contigs DENSELY: wave w of the grid is entry w of `wave_map` = {read of the launch,
    // active contig of that read}.  A team (the waves of one read) needs nothing of a workgroup - no barrier, no shared LDS, the
    // exchange goes through memory - so its waves may sit in any workgroups; the host keeps the launch within the wave slots of the
    // chip, so all of them are resident.  (Before: whole workgroups per read, whose spare waves idled: a read cut down to two contigs
    // by the pre-alignment filter used half a workgroup, 50 contigs 52 waves.)
    const uint32_t wv = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    if (wv >= n_waves) return;
    const uint2 wm = wave_map[wv];
    // (tell the compiler these are uniform, so that everything read through V is scalar)
    const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)wm.x);       // classic launch: the job; persistent teams: the team
    const uint32_t kmine = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wm.y & 0x1FFFFFFFu));      // this wave's contig: active contig number
    // (experiment, STITCH_REGS_MAP=2) bit 30: the host has dealt this team's waves to workgroups of one XCD and asks for PLAIN granule stores:
    // they stay in that XCD's L2, where the team's sc1 polls (L1-bypassing, L2-served) find them without the trip through the fabric
    const uint32_t gran_aux = ((uint32_t)__builtin_amdgcn_readfirstlane((int)wm.y) & 0x40000000u) ? (uint32_t)AUX_VOLATILE : (uint32_t)(AUX_SC1 | AUX_VOLATILE);
    // bit 29: the waves of this workgroup all belong to ONE team (the host sets it where every read's contigs fill whole workgroups): the
    // workgroup's first wave polls the team's granules for all of them and hands them over through LDS.  Every wave of a team polling every
    // granule on its own is (waves x contigs) sc1 loads per round — at cfg5's 200 contigs 3.2 MB per round over ten teams, rounds of a
    // microsecond or two: terabytes per second at the L2s and the fabric for 1.6 KB of news per team and column.
    const bool wg_poll = ((uint32_t)__builtin_amdgcn_readfirstlane((int)wm.y) & 0x20000000u) != 0u;
    if (kmine == 0x1FFFFFFFu) return;               // a padding entry of the wave map
    const bool streaming = qp != nullptr;           // (the queue's words are read through qp where they are needed: two scalar registers across the column loop, not fifteen)
    if (wg_poll) {      // (LDS comes as the last workgroup left it: no tag of this launch there before anybody looks)
        extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn0[];
        if ((threadIdx.x >> 6) == 0 && (threadIdx.x & 63) == 0) *(volatile uint32_t*)(s_dyn0 + LDS_TB + 4096) = 0u;
        __syncthreads();
    }
    // ---- persistent teams: one round of this loop per read the team aligns (a classic launch leaves after the first) -----------------
    for (uint32_t seq = 1;; ++seq) {
    uint32_t job = slot;
    if (streaming) {
        const StreamCtl q = *qp;
        // The team's first wave takes the next job off the queue and announces it in the team's mailbox {round, job}; the others wait for
        // this round's announcement.  (A team cannot lap itself: a read has at least two columns, and no wave gets past column 2 of a read
        // without the granules all its team mates wrote AFTER reading this round's mailbox.)
        const gptr<unsigned long long> mb = as_global(q.mbox) + slot;
        const int lane0 = threadIdx.x & 63;
        uint32_t idx = 0xFFFFFFFFu;
        if (kmine == 0u) {
            if (lane0 == 0) {
                // The job's arena block is the one job - blocks used: the team waits until the host has walked that read and taken its chains, and
                // it takes the job off the queue only THEN — a team that waits holds no job, so it can leave when the host asks teams to.  ONE lane
                // of the team looks at the host's words, and not often: they live in pinned host memory, every look is a trip over PCIe (a whole
                // team looking — 200 waves every 3 us at cfg5, where a run has few spare blocks and teams wait as a rule — slowed every fill
                // wave and every copy of the process five-fold: gpurun_out/r4j).  Bounded: 40 s; the host ends a run that makes no progress
                // earlier.  The team's other waves wait for the mailbox, i.e. in device memory.
                // The host's other word (h_abort): 0 = carry on; 1 .. 2^31 - 1 = that many teams are asked to LEAVE (a launch beside the teams
                // waits for a wave slot where the dispatcher put it; while the host waits for that launch it recycles no blocks, so the teams are
                // all here: the first to look take a ticket each and go, the others carry on); above = hand out no more jobs.
                {
                    const uint32_t t0 = (uint32_t)wall_clock64();
                    uint32_t tried = 0u;
                    for (uint32_t spins = 1;; ++spins) {
                        const uint32_t ab = __hip_atomic_load(q.h_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (ab >= 0x80000000u) break;
                        if (ab > tried) { tried = ab; if (atomicAdd(q.next + 16, 1u) < ab) break; atomicSub(q.next + 16, 1u); }
                        const uint32_t nx = __hip_atomic_load(q.next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (nx >= q.n_jobs) break;
                        const uint32_t ready = __hip_atomic_load(q.h_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (ready > nx) {
                            idx = atomicAdd(q.next, 1u);
                            if (idx >= q.n_jobs) { idx = 0xFFFFFFFFu; break; }
                            // (another team took job nx in between: this one's block may not be free yet — wait for it, holding the job)
                            for (uint32_t sp2 = 1; __hip_atomic_load(q.h_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) <= idx; ++sp2) {
                                if ((sp2 & 7u) == 0 && (__hip_atomic_load(q.h_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= 0x80000000u || (uint32_t)wall_clock64() - t0 > 4000000000u)) { idx = 0xFFFFFFFFu; break; }
                                for (int k = 0; k < 6; ++k) __builtin_amdgcn_s_sleep(127);
                            }
                            break;
                        }
                        if ((uint32_t)wall_clock64() - t0 > 4000000000u) break;
                        for (int k = 0; k < 6; ++k) __builtin_amdgcn_s_sleep(127);      // ~20 us between two looks
                    }
                }
                __hip_atomic_store(mb, ((unsigned long long)seq << 32) | idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
        }
        else {
            const uint32_t t0 = (uint32_t)wall_clock64();
            for (uint32_t spins = 1;; ++spins) {
                const unsigned long long v = __hip_atomic_load(mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(v >> 32) == seq) { idx = (uint32_t)v; break; }
                if ((spins & 255u) == 0 && (uint32_t)wall_clock64() - t0 > 4200000000u) {      // (the first wave's own wait for a block is bounded by 40 s: beyond that it is gone)
                    if (lane0 == 0) __hip_atomic_store(q.h_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    return;
                }
                if (spins < 64u) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(127);
            }
            idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
        }
        if (idx == 0xFFFFFFFFu) return;                      // the queue is empty (or the host has called the run off)
        job = idx;
    }
    const JobView& V = jobs[job];
#ifdef STITCH_EXP_PRIO
    if (job & 1u) __builtin_amdgcn_s_setprio(2);       // experiment: every other read's team has priority on the SIMDs it shares
#endif
    const DpParams P = V.P;
    const uint32_t n = V.n, nact = V.nact, Rtot = V.Rtot, C = V.C;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (wave-uniform, provably)
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    uint8_t* const s_wave = s_dyn + (size_t)wave * LDS_PER_WAVE;

    if (kmine >= nact) return;                     // (a padding entry of the wave map)
    const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)V.act[kmine]);
    ContigDesc cd = V.cd[c];
    cd.m = (uint32_t)__builtin_amdgcn_readfirstlane((int)cd.m); cd.roff = (uint32_t)__builtin_amdgcn_readfirstlane((int)cd.roff);
    cd.troff = (uint32_t)__builtin_amdgcn_readfirstlane((int)cd.troff); cd.seqoff = (uint32_t)__builtin_amdgcn_readfirstlane((int)cd.seqoff);
    const uint32_t m = cd.m, roff = cd.roff;
    // position of the same-name opposite strand in the active list (-1: none in this subset)
    int32_t kopp = -1;
    { const int32_t opp = V.opp_act[c]; if (opp >= 0) for (uint32_t k = 0; k < nact; ++k) if ((int32_t)V.act[k] == opp) kopp = (int32_t)k; }
    kopp = __builtin_amdgcn_readfirstlane(kopp);
    // groups of four rows -> lanes
    const uint32_t ngr = (m + 3) / 4, gq = ngr / 64, grem = ngr % 64;
    const uint32_t gl = gq + ((uint32_t)lane < grem ? 1u : 0u);                     // groups of this lane
    const uint32_t nrows = 4 * gl;
    const uint32_t rowbase = 4 * ((uint32_t)lane * gq + ((uint32_t)lane < grem ? (uint32_t)lane : grem));   // 0-based row of the lane's first row
    const uint32_t pad = 4 * ngr - m;                 // 0..3 registers without a row, at the bottom of the last lane that has rows
    const int mlane = (int)(gq > 0 ? 63u : grem - 1u);            // row m: register rsh + `pad` of that lane
    const uint32_t gtop = gq + (grem > 0 ? 1u : 0u);  // groups of the fullest lane: the unrolled blocks g >= gtop are skipped by every lane
    // EVERY lane's first row sits in register 4 gtop - 1: the lanes that hold one group less than the fullest have it at the
    // BOTTOM — their rows end in register 4 (rsh), group 0 is not theirs.  So the group blocks g >= 1 run under a scalar condition
    // (g < gtop) and only group 0 is per lane.
    const uint32_t rsh = (grem > 0 && (uint32_t)lane >= grem) ? 4u : 0u;          // register of the lane's last row
    const bool has0 = gl > 0 && rsh == 0u;                                        // group 0 is this lane's
    const uint32_t gm = (grem > 0 && gq > 0) ? 1u : 0u;                           // group of row m (of lane mlane): its register is 4 gm + pad
    const uint32_t padreg = 4u * gm + pad;
    const int32_t jump_same = P.jump_same, jump_opp = P.jump_opp, jump_inter = P.jump_inter;
    const int32_t MW = (int32_t)((uint32_t)P.match << 16), XW = (int32_t)((uint32_t)P.mismatch << 16);
    const int32_t GE1 = (int32_t)((uint32_t)P.gap_extend << 16) + 1, GO1 = (int32_t)((uint32_t)(P.gap_open + P.gap_extend) << 16) + 1;
    const int32_t ge = P.gap_extend, kb0 = P.gap_open + P.gap_extend;
    const gptr<const uint8_t> yseq = as_global(V.y);
    const gptr<u32x2> yrec = (gptr<u32x2>)as_global(V.D);      // [Rtot] 8-byte records: D and Dlen are contiguous (layout_job)
    const gptr<unsigned long long> xchg = as_global(V.xchg);
    const gptr<uint32_t> jt_idx = as_global(V.jt_idx), jt_from = as_global(V.jt_from), Lx = as_global(V.Lx);
    uint8_t* const tb0 = V.tb + roff;
    const bool ymode_global = V.yrec_global != 0;
    // register IDX of this lane holds the contig's row (1-based) pos1 = rowbase + nrows - (IDX - rsh)
    const int32_t pos1_0 = (int32_t)(rowbase + nrows);                  // ... of the lane's LAST row (register rsh)
    // scan terms of register IDX: key = S'.score + kbase + ge * IDX, q = S'.len + qbase + IDX with kbase = go + ge - ge * pos1_0, qbase = 1 - pos1_0;
    // I score = run.key + bbase - ge * IDX, I length = run.q + lbase - IDX with bbase = ge * pos1_0, lbase = pos1_0

    // ---- column 0 (init_matrices :97-186): registers, bases, per-row arrays of this wave's rows -----------------------------------
    uint32_t S[RMAX], D[RMAX];
#pragma unroll
    for (int i = 0; i < RMAX; ++i) { S[i] = 0u; D[i] = (uint32_t)word_make(-16384, 0); }     // D = "MIN": never extends, never wins, cannot wrap
    {
        uint32_t* const xw0 = (uint32_t*)(s_wave + LDS_XW) + lane;
#define INIT(g) if ((g) == 0 ? has0 : (gl > 0 && (uint32_t)(g) < gtop)) { \
            uint32_t w = 0; \
            _Pragma("unroll") for (int k = 3; k >= 0; --k) { \
                const uint32_t row = rowbase + (nrows - 1 - (4u * (g) + (uint32_t)k - rsh));              /* 0-based row of the contig */ \
                uint32_t b = 0xFFu;                                                                         /* no row: equals no base */ \
                if (row < m) { \
                    const uint32_t tr = cd.troff + row; \
                    S[4 * (g) + k] = (uint32_t)word_make(sh.S0[tr], sh.Slen0[tr]); \
                    u32x2 rec; rec.x = (uint32_t)word_make(sh.Sn0[tr], sh.Slen0[tr]); rec.y = sh.SnSet0[tr] ? n : (YB ? 0xFFFFFFFFu : 0u);      /* (YB: "never set" must read as older than every column, see the scan behind the column loop) */ \
                    yrec[roff + (4u * (g) + (uint32_t)k) * 64u + (uint32_t)lane] = rec; \
                    V.SmoveF[roff + row] = TB_NONE; V.ImoveF[roff + row] = TB_NONE; \
                    b = V.xseq[cd.seqoff + row]; \
                } \
                w |= b << (8 * k); \
            } \
            xw0[(g) * 64] = w; \
        }
        REP20(INIT)
#undef INIT
    }
    if (lane == 0) V.Lx[(size_t)c * (n + 1)] = sh.lx0[c];
    int32_t vrun = sh.base0[c].score;                                     // the contig's running maximum over columns < j
    int32_t cmax_prev = sh.base0[c].score;                                // ... and its maximum in column j-1 alone (what this wave announced)
    // cell (m, j-1), for the zero-cost end-to-start jump of a circular contig (get_jump_score_and_len :258-289)
    const uint32_t trm = cd.troff + m - 1;
    bool rowm_xsuf = sh.Smove0[trm] == TB_XCLIP_SUFFIX; int32_t rowm_S = sh.S0[trm]; uint32_t rowm_len = sh.Slen0[trm];
    const int32_t circular = P.circular;
    int32_t gmax = 0;                                                     // best score of any contig in columns < j (row 0 holds 0)
    // the contigs' column arg-max of column j-1, lane l holding active contigs l, l + 64, ...: {column, score, len, from} granules
    unsigned long long gv[NQ];
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) gv[qq] = 0ull;
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) {
        const uint32_t k = (uint32_t)lane + 64u * qq;
        if (k < nact) { const JumpBase b = sh.base0[V.act[k]]; gv[qq] = ((unsigned long long)(uint32_t)(b.score & 0xFFFF) << 32) | ((unsigned long long)(b.len & 0xFFFFu) << 16) | (b.from & 0xFFFFu); gmax = b.score > gmax ? b.score : gmax; }
    }
    gmax = (int32_t)wave_max_u32((uint32_t)gmax);
    // More than one granule register per lane (NQ > 1: up to 256 contigs): the records are only needed from the poll to the end of the jump
    // selection, so they must not stay in registers across the sweeps (eight registers the sweeps do not have: the kernel went to scratch
    // memory, which a persistent launch must not use).  Column 1 takes its records (column 0's, from the host) out of LDS.
    unsigned long long* const g_stash = (unsigned long long*)(s_wave + LDS_TB);
    if (NQ > 1) {
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) g_stash[lane + 64 * qq] = gv[qq];
    }

    // descriptors of the arrays written with buffer stores (scalar base + lane offset: no address arithmetic in vector registers)
    const __amdgpu_buffer_rsrc_t ryr = __builtin_amdgcn_make_buffer_rsrc((uint8_t*)V.D + 8ull * roff, 0, 0x7FFFFFFF, RSRC_WORD3);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(V.S, 0, 0x7FFFFFFF, RSRC_WORD3);
    const __amdgpu_buffer_rsrc_t rxc = __builtin_amdgcn_make_buffer_rsrc((void*)V.xchg, 0, 0x7FFFFFFF, RSRC_WORD3);
    const uint32_t oSlen = (uint32_t)((const uint8_t*)V.Slen - (const uint8_t*)V.S), oIval = (uint32_t)((const uint8_t*)V.Ival - (const uint8_t*)V.S), oIlen = (uint32_t)((const uint8_t*)V.Ilen - (const uint8_t*)V.S);
    // ---- the insertion chain across the lanes: E = the chain of a lane's own openers as it arrives BEHIND the lane's last row,
    // exitext = whether it got there by an extension; Iin / extin = what arrives at this lane's first row
    auto chain_across_lanes = [&](const int32_t E, const bool exitext, const int32_t pos_x, const uint32_t gl_x, int32_t& Iin, uint32_t& extin) __attribute__((always_inline)) {
        const int32_t pos_exit = pos_x + 1;                           // 1-based position of the row behind this lane's last row
        int32_t nk = word_score(E) - ge * pos_exit;
        if (gl == 0) nk = SCAN_LOW;                                   // lanes without rows
        int32_t kt = (int32_t)(((uint32_t)nk << 6) | (uint32_t)(63 - lane));
        { const int32_t o = dpp_mov<DPP_ROW_SHR0 | 1>(INT32_MIN, kt); kt = o > kt ? o : kt; } { const int32_t o = dpp_mov<DPP_ROW_SHR0 | 2>(INT32_MIN, kt); kt = o > kt ? o : kt; }
        { const int32_t o = dpp_mov<DPP_ROW_SHR0 | 4>(INT32_MIN, kt); kt = o > kt ? o : kt; } { const int32_t o = dpp_mov<DPP_ROW_SHR0 | 8>(INT32_MIN, kt); kt = o > kt ? o : kt; }
        { const int32_t o = dpp_mov<DPP_BCAST15, 0xA>(INT32_MIN, kt); kt = o > kt ? o : kt; } { const int32_t o = dpp_mov<DPP_BCAST31, 0xC>(INT32_MIN, kt); kt = o > kt ? o : kt; }
        const int32_t rt = from_prev_lane(kt, INT32_MIN);             // exclusive: lanes before this one
        const uint32_t w = 63u - ((uint32_t)rt & 63u);                // the lane whose chain arrives (if any lane is above)
        const int32_t Ew = __builtin_amdgcn_ds_bpermute((int)(w << 2), E);
        const int32_t first_pos = pos_x + 1 - 4 * (int32_t)gl_x;      // 1-based position of this lane's first row
        // the row-0 opener: word GO1 at position 1, key go + ge - ge; it is the earliest opener: it wins ties
        const bool seed = lane == 0 || (kb0 - ge) >= (rt >> 6);
        const uint32_t wn = w + 1u;                                   // the winner's exit position = first row of the lane after it
        const int32_t w_exit = 4 * (int32_t)(wn * gq + (wn < grem ? wn : grem)) + 1;
        const int32_t dist = seed ? first_pos - 1 : first_pos - w_exit;
        const int32_t base = seed ? GO1 : Ew;
        int32_t sc = word_score(base) + ge * dist; sc = sc > -28000 ? sc : -28000;      // (a chain that low loses to the first opener below)
        Iin = word_make(sc, word_len(base) + (uint32_t)dist);
        // did it arrive by an extension?  Not if it is the opener of the row right above: row 0 for lane 0, or the previous
        // lane's last row when that lane's own last step chose the opener
        const int32_t ext_prev = from_prev_lane(exitext ? 1 : 0, 0);
        extin = (lane == 0) ? 0u : ((!seed && w + 1u == (uint32_t)lane && ext_prev == 0) ? 0u : (uint32_t)TBB_IEXT);
    };
    // (the two words a row selects between by the base comparison, kept in vector registers: a select cannot take both from scalar
    // registers, and re-materialising them costs two instructions per group of rows)
    int32_t MW1v = MW + 1, XW1v = XW + 1; asm volatile("" : "+v"(MW1v), "+v"(XW1v));
    uint32_t ychunk = 0;
    unsigned long long own_gran = 0ull;                                  // what this wave announced for column j-1 (wave-uniform)
    RPROF_DECL
    // the shader clock this kernel actually gets: s_memtime (shader cycles) against s_memrealtime (100 MHz) over the column loop of
    // each read's first wave, left behind the error word for the host (stitch_timing.clk_*; profiles/clock_probe.sh)
    const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_w0 = wall_clock64();
    for (uint32_t j = 1; j <= n; ++j) {
        const bool lastcol = j == n;
        RPROF(7)
#ifdef STITCH_PROFILE
        pf_c0 = pf_t;
#endif
        // ---- poll the team's granules of column j-1 (column 0 came from the host) --------------------------------------------------
        if (j > 1) {
            const uint32_t want = j - 1;
            int lane_p = lane; asm volatile("" : "+v"(lane_p));
            // (scalar base + lane offset through the buffer descriptor: no 64-bit address per lane; sc1 = agent scope, as the
            // atomic load it stands for, and "volatile": every round of the loop reads memory)
            const uint32_t gso = (want & 1u) * C * 8u;
            // (a spinning wave takes issue slots from the wave it shares its SIMD with, which is computing another read's column: the
            // loop is kept to a load, a compare and a sleep; the clock is looked at once in 1024 rounds)
            // the workgroup's shared copy of the team's granules: two columns' worth (the poller fills one while its mates may still read the
            // other: it cannot be two columns ahead of a mate, whose granule of the column between it needs), then the tag of what is there
            unsigned long long* const wg_gran = (unsigned long long*)(s_dyn + LDS_TB) + (want & 1u) * 256u;
            volatile uint32_t* const wg_tag = (volatile uint32_t*)(s_dyn + LDS_TB + 4096);
            const uint32_t tag_now = (seq << 16) | (want & 0xFFFFu);
            const uint32_t t0 = (uint32_t)wall_clock64();
            if (wg_poll && wave != 0) {
                for (uint32_t spins = 1;; ++spins) {
                    if (*wg_tag == tag_now) break;
                    if ((spins & 4095u) == 0 && (uint32_t)wall_clock64() - t0 > 420000000u) {
                        if (lane == 0) { *V.err = 1; if (streaming) __hip_atomic_store(qp->h_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
                        return;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
#pragma unroll
                for (int qq = 0; qq < NQ; ++qq) gv[qq] = wg_gran[lane_p + 64 * qq];
            }
            else
            for (uint32_t spins = 1;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int qq = 0; qq < NQ; ++qq) {
                    const uint32_t k = (uint32_t)lane_p + 64u * qq;
                    if (NQ > 1) gv[qq] = 0ull;
                    if (k < nact) {
                        const u32x2 g2 = __builtin_amdgcn_raw_buffer_load_b64(rxc, 8u * k, gso, AUX_SC1 | AUX_VOLATILE);
                        // (the wave's OWN granule is not waited for: it knows what it wrote, and the store's round trip through the
                        // memory system — microseconds — would sit on the critical path of the one wave the whole team waits for)
                        gv[qq] = ((unsigned long long)g2.y << 32) | g2.x; ok &= (g2.y >> 16) == want || k == kmine;
                    }
                }
                if (__all(ok)) break;
                if ((spins & 1023u) == 0) {
                    // 4 s at 100 MHz: a partner is not resident — or another wave of the read has said so already (its error word, read
                    // at agent scope: a workgroup that becomes resident late does not wait its own four seconds)
                    const uint32_t e_seen = __builtin_amdgcn_raw_buffer_load_b32(rxc, 0u, 32u * C, AUX_SC1 | AUX_VOLATILE);
                    if (e_seen != 0u || (uint32_t)wall_clock64() - t0 > 400000000u) {
                        if (lane == 0) { *V.err = 1; if (streaming) __hip_atomic_store(qp->h_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
                        return;
                    }
                }
                __builtin_amdgcn_s_sleep(STITCH_POLL_SLEEP);
            }
            if (wg_poll && wave == 0) {      // (the poller's own granule as it wrote it, then the news for the workgroup's other waves)
#pragma unroll
                for (int qq = 0; qq < NQ; ++qq) { if ((uint32_t)lane_p + 64u * qq == kmine) gv[qq] = own_gran; wg_gran[lane_p + 64 * qq] = gv[qq]; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the records are in LDS before their tag
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) *wg_tag = tag_now;
            }
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) if ((uint32_t)lane_p + 64u * qq == kmine) gv[qq] = own_gran;
            uint32_t best = 0;
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) { const uint32_t sc = (uint32_t)(gv[qq] >> 32) & 0xFFFFu; best = ((uint32_t)lane + 64u * qq < nact && sc > best) ? sc : best; }
            const int32_t colmax = (int32_t)wave_max_u32(best);
            gmax = colmax > gmax ? colmax : gmax;
        }
        else if (NQ > 1) {
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) gv[qq] = g_stash[lane + 64 * qq];
        }
        RPROF(0)
#ifdef STITCH_PROFILE
        pf_p1 = pf_t; pf_p0 = pf_c0;
#endif
        // the read's bases, 64 columns per (coalesced) load: lane l holds y[jb + l]
        if (((j - 1) & 63u) == 0) ychunk = (j - 1 + lane < n) ? (uint32_t)yseq[j - 1 + lane] : 0u;
        const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)ychunk, (int)((j - 1) & 63u)) & 0xFFu;

        // ---- best jump out of column j-1 for this contig (multi_contig_aligner.rs:292-331): inter-contig = max by (score, len),
        // LAST aligner on full ties (max_by_key); the records of active contig k sit in lane k % 64, register k / 64
        if (NQ > 1 && j > 1) {      // (this column's records where rec_of finds them; column 1's are there already)
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) g_stash[lane + 64 * qq] = gv[qq];
        }
        auto rec_of = [&](uint32_t k) -> unsigned long long {
            if (NQ > 1) {           // from LDS (every lane reads the one address): picking among NQ registers by a run-time index put the records in scratch memory
                const unsigned long long v = g_stash[k];
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
                return ((unsigned long long)hi << 32) | lo;
            }
            const unsigned long long v = gv[0];
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)(k & 63u)), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)(k & 63u));
            return ((unsigned long long)hi << 32) | lo;
        };
        auto act_of = [&](uint32_t k) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)V.act[k]); };      // (a scalar load: its result only goes into the jump table)
        JumpInfo ji;
        {
            // the best OTHER contig (not this one, not its opposite strand): 0 = none, else its position in the active list + 1
            uint32_t kbest1 = 0;
            if (NQ == 1) {
                // one record per lane: a 32-bit maximum of score << 16 | len, and the LAST lane that holds it
                const bool elig = (uint32_t)lane < nact && (uint32_t)lane != kmine && lane != kopp;
                const uint32_t key = elig ? (uint32_t)(gv[0] >> 16) : 0u;
                const uint32_t mx = wave_max_u32(key);
                const unsigned long long who = __ballot(elig && key == mx);
                kbest1 = who ? 64u - (uint32_t)__builtin_clzll(who) : 0u;
            }
            else {
                unsigned long long ik = 0;
#pragma unroll
                for (int qq = 0; qq < NQ; ++qq) {
                    const uint32_t k = (uint32_t)lane + 64u * qq;
                    if (k < nact && k != kmine && (int32_t)k != kopp) {
                        const unsigned long long key = ((gv[qq] & 0x0000FFFFFFFF0000ull) << 0) | (k + 1);      // score << 32 | len << 16 | k + 1
                        ik = key > ik ? key : ik;
                    }
                }
                ik = wave_max_u64(ik);
                kbest1 = (uint32_t)(ik & 0xFFFFu);
            }
            { const unsigned long long b = rec_of(kmine); ji.score = (int32_t)((b >> 32) & 0xFFFFu) + jump_same; ji.len = (uint32_t)(b >> 16) & 0xFFFFu; ji.idx = c; ji.from = (uint32_t)b & 0xFFFFu; }
            if (kopp >= 0) { const unsigned long long b = rec_of((uint32_t)kopp); const int32_t sc = (int32_t)((b >> 32) & 0xFFFFu) + jump_opp; if (sc > ji.score) { ji.score = sc; ji.len = (uint32_t)(b >> 16) & 0xFFFFu; ji.idx = act_of((uint32_t)kopp); ji.from = (uint32_t)b & 0xFFFFu; } }
            if (kbest1 != 0) {
                const uint32_t kw = kbest1 - 1; const unsigned long long b = rec_of(kw);
                const int32_t sc = (int32_t)((b >> 32) & 0xFFFFu) + jump_inter;
                if (sc > ji.score) { ji.score = sc; ji.len = (uint32_t)(b >> 16) & 0xFFFFu; ji.idx = act_of(kw); ji.from = (uint32_t)b & 0xFFFFu; }
            }
        }
        // Row 1 of a circular contig may take the zero-cost jump from row m of the previous column instead (dp_core.h
        // local_row1_circ); the walk learns it from bit 31 of the column's jump-table entry
        bool circ = false;
        if (CIRC) { ColCtx cc; cc.jump = ji; cc.circ_ok = (circular && !rowm_xsuf) ? 1 : 0; cc.circ_score = rowm_S; cc.circ_len = rowm_len + 1; circ = local_row1_circ(cc); }
        const int32_t JSW = __builtin_amdgcn_readfirstlane(word_make(ji.score, ji.len));
        const int32_t JSW1 = circ ? __builtin_amdgcn_readfirstlane(word_make(rowm_S, rowm_len + 1)) : JSW;
        // y-suffix records are kept for cells whose score reaches ybase (never for a zero word)
        const int32_t ybase = ymode_global ? gmax : vrun;
        const int32_t ythr = ybase > 0 ? (int32_t)((uint32_t)ybase << 16) : 1;
        // (YB) the column's common word: what a cell gets that matches the read's base and takes the jump — in the contigs the read is not
        // on (199 of cfg5's 200) that is every cell that reaches the contig's running best, a quarter of all rows in every column, and 8 bytes
        // of record each were 4.3 of the 5.3 bytes per cell the kernel wrote there.  Such a cell's record would be {W, n - j}: a BIT says as
        // much (bit 7 of its traceback byte, which no reader of the byte looks at), and W goes to Wcol[j] once per column.  The scan behind the
        // column loop writes each row's LAST such record where the stores would have left it.  Off where W is below the threshold (then no
        // cell's bit is set: the word of a row that takes no part is 0, the threshold at least 1).
        const uint32_t Wc = YB ? (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(MW1v + JSW - 1) >= ythr ? (int32_t)(MW1v + JSW - 1) : -1) : 0xFFFFFFFFu;
        if (lane == 0) {
            jt_idx[(size_t)c * (n + 1) + j] = ji.idx | (circ ? JT_CIRC_BIT : 0u); jt_from[(size_t)c * (n + 1) + j] = ji.from;
            if (YB) V.Wcol[(size_t)c * (n + 1) + j] = Wc;
        }

        // (what depends only on the lane is constant over the read: the compiler would hoist the base words and the group guards out
        // of the column loop and pin registers for them; values it cannot see through keep them one LDS read / one compare each)
        const int lane_x = lane; uint32_t gl_x = gl; int32_t pos_x = pos1_0; bool has0_x = has0; asm volatile("" : "+v"(gl_x), "+v"(pos_x)); { uint32_t h = has0 ? 1u : 0u; asm volatile("" : "+v"(h)); has0_x = h != 0u; }
        uint32_t gtop_x = gtop; asm volatile("" : "+s"(gtop_x));        // (opaque: the compiler would keep twenty compare results per sweep in scalar registers)
        const uint32_t* const xw_lane = (const uint32_t*)(s_wave + LDS_XW) + lane_x;
        uint32_t tbv[NG];                              // the groups' traceback dwords, from sweep to sweep in registers (the kernel has them to spare)
        uint16_t* const bs_lane = (uint16_t*)(s_wave + LDS_BS) + lane_x;

        RPROF(1)
        // ---- pass 1: rows top to bottom = registers 4 gl - 1 .. 0 ------------------------------------------------------------------
        Col cx;
        cx.MW1 = MW1v; cx.XW1 = XW1v; cx.GE1 = GE1; cx.GO1 = GO1; cx.JSWm1 = JSW - 1; cx.q = q;
        cx.dgm = 0; cx.pad = padreg;
        cx.jfix = lane == 0 ? JSW1 - JSW : 0;
        uint32_t xwA = xw_lane[(gtop - 1u) * 64], xwB = xwA;             // every lane's first row is byte 3 of group gtop - 1 (either parity: both copies)
        cx.aw1 = (xwA >> 24) == q ? cx.MW1 : cx.XW1;
        // the row above a lane's first row is the previous lane's last row: its register rsh; row 0 for lane 0 (score 0, length 0 in Local mode)
        cx.DG = from_prev_lane((int)(rsh ? S[4] : S[0]), 0) + cx.aw1;
#define XC(g) (((g) & 1) ? xwB : xwA)
#define XN(g) (((g) & 1) ? xwA : xwB)
        // (the smaller of the column's jump words, plus the worse of match and mismatch: a lower bound of every cell's jump candidate)
        const int32_t jw_floor = (word_score(JSW1) < word_score(JSW) ? word_score(JSW1) : word_score(JSW)) + (P.mismatch < P.match ? P.mismatch : P.match);
        const bool may_clip = __builtin_amdgcn_readfirstlane(jw_floor) < 0;
        // CAN AN INSERTION MATTER IN THIS COLUMN AT ALL?  Every cell of the column is at least its jump candidate (and 0), i.e. >=
        // tfloor = max(jw_floor, 0).  No cell without its insertion exceeds U = max(this contig's maximum in column j-1, the column's
        // jump score) + the better of match and mismatch: a diagonal comes from a cell of column j-1, a deletion is never above the
        // S of its own cell, the jump is the jump.  An insertion chain is an opener S' + go + ge carried down at ge <= 0 per row, so no
        // chain exceeds U + go + ge; when that is below tfloor no insertion reaches a cell (the merge needs score(I) >= score(S)), there
        // is no INS move in the column, and the walk reads a column's "I extended" bits only behind one.  Then pass 1b, the scan across
        // the lanes and pass 2's repair are skipped — for every contig whose column maximum is more than |go + ge| - |mismatch| + match
        // below the best jump source, i.e. all but the one or two the alignment currently runs in.  (Wave-uniform, exact.)
        const int32_t jw_top = (word_score(JSW1) > word_score(JSW) ? word_score(JSW1) : word_score(JSW));
        const int32_t u_top = (cmax_prev > jw_top ? cmax_prev : jw_top) + (P.mismatch > P.match ? P.mismatch : P.match);
        const bool no_ins = __builtin_amdgcn_readfirstlane((u_top > 0 ? u_top : 0) + kb0 < (jw_floor > 0 ? jw_floor : 0) ? 1 : 0) != 0;
/* per lane for group 0 only; the other groups under a scalar condition, compared where it is used (kept as twenty boolean masks the
   conditions of a sweep take forty scalar registers) */
#define GUARD(g) ((g) == 0 ? has0_x : ({ asm volatile("" : "+s"(gtop_x)); (uint32_t)(g) < gtop_x; }))
#define P1(g) if (GUARD(g)) { uint32_t tbw; \
            row_pass1<4 * (g) + 3, CIRC>(S[4 * (g) + 3], D[4 * (g) + 3], tbw, cx, XC(g), XN(g), xw_lane, bs_lane); row_pass1<4 * (g) + 2, CIRC>(S[4 * (g) + 2], D[4 * (g) + 2], tbw, cx, XC(g), XN(g), xw_lane, bs_lane); \
            row_pass1<4 * (g) + 1, CIRC>(S[4 * (g) + 1], D[4 * (g) + 1], tbw, cx, XC(g), XN(g), xw_lane, bs_lane); row_pass1<4 * (g), CIRC>(S[4 * (g)], D[4 * (g)], tbw, cx, XC(g), XN(g), xw_lane, bs_lane); \
            if (__builtin_expect(may_clip, 0)) { clip_row<4 * (g) + 3>(S[4 * (g) + 3], tbw); clip_row<4 * (g) + 2>(S[4 * (g) + 2], tbw); clip_row<4 * (g) + 1>(S[4 * (g) + 1], tbw); clip_row<4 * (g)>(S[4 * (g)], tbw); } \
            tbv[g] = tbw; }
        REP20(P1)
#undef P1
#undef XC
#undef XN
        RPROF(2)
        // ---- pass 1b: the chain of the lane's own openers, merged into the lane's cells; its "extended" bits join the codes in LDS ---
        int32_t Iin = CHAIN_NONE; uint32_t extin = 0u;
        bool chains_dead = true;
        const int32_t tfl = jw_floor > 0 ? jw_floor : 0;                  // every cell of this column is at least this
        const int32_t live_thr = __builtin_amdgcn_readfirstlane(word_make(tfl, 0)), open_thr = __builtin_amdgcn_readfirstlane(word_make(tfl - kb0, 0));
        if (!no_ins) {
        Col2 cl;
        cl.GE1 = GE1; cl.GO1 = GO1;
        cl.Iw = CHAIN_NONE; cl.extn = 0u;                                 // nothing arrives at the lane's first row from the lane itself
        /* ONE test per group whether the chain can matter there at all: a chain is dead once its score is below tfloor (every cell of \
           the column is >= tfloor, a merge needs score(I) >= score(S), and a chain only decays from row to row unless an opener renews \
           it), and a cell opens a chain worth following only if score(S') + go + ge >= tfloor.  A group in which no lane carries a live \
           chain and no lane holds such a cell is left alone: its chain stays dead (the stale word is below tfloor, which is all anybody \
           asks of it), its "I extended" bits are never read (the walk reads them only behind an INS move, i.e. along a live chain). */ \
#define P1B(g) if (GUARD(g)) { \
            const uint32_t h32 = S[4 * (g) + 3] > S[4 * (g) + 2] ? S[4 * (g) + 3] : S[4 * (g) + 2], h10 = S[4 * (g) + 1] > S[4 * (g)] ? S[4 * (g) + 1] : S[4 * (g)]; \
            const bool hot = (int32_t)(h32 > h10 ? h32 : h10) >= open_thr || cl.Iw >= live_thr; \
            if (__ballot(hot) != 0ull) { \
            const uint32_t tbl = tbv[g]; \
            uint32_t eb = 0u; \
            const int32_t i3 = chain_row<4 * (g) + 3>(S[4 * (g) + 3], eb, cl), i2 = chain_row<4 * (g) + 2>(S[4 * (g) + 2], eb, cl); \
            const int32_t i1 = chain_row<4 * (g) + 1>(S[4 * (g) + 1], eb, cl), i0 = chain_row<4 * (g)>(S[4 * (g)], eb, cl); \
            uint32_t tbw = tbl | eb; \
            /* the insertion can only change a cell if its score reaches the cell's (S >= 0, so a negative insertion never does) */ \
            const bool m3 = word_score(i3) >= word_score((int32_t)S[4 * (g) + 3]), m2 = word_score(i2) >= word_score((int32_t)S[4 * (g) + 2]); \
            const bool m1 = word_score(i1) >= word_score((int32_t)S[4 * (g) + 1]), m0 = word_score(i0) >= word_score((int32_t)S[4 * (g)]); \
            if (__builtin_expect((__ballot(m3) | __ballot(m2) | __ballot(m1) | __ballot(m0)) != 0ull, 0)) { RCOUNT(0) \
                merge_row<4 * (g) + 3>(S[4 * (g) + 3], tbw, __ballot(m3), i3, bs_lane); merge_row<4 * (g) + 2>(S[4 * (g) + 2], tbw, __ballot(m2), i2, bs_lane); \
                merge_row<4 * (g) + 1>(S[4 * (g) + 1], tbw, __ballot(m1), i1, bs_lane); merge_row<4 * (g)>(S[4 * (g)], tbw, __ballot(m0), i0, bs_lane); \
            } \
            tbv[g] = tbw; } }
        REP20(P1B)
#undef P1B
#undef GUARD
        RPROF(3)
        // ---- the insertion chain across the lanes ------------------------------------------------------------------------------------
        // E = the chain of the lane's own openers as it arrives BEHIND the lane's last row (pass 1b's last step ended there).
        // What arrives at lane l's first row is the best E of the
        // lanes above — or the opener of row 0 (S' = 0: word GO1 at row 1) — carried down: score + ge per row, length + 1 per row.
        // "Best" = largest score at a common position, earliest lane on ties (the extension wins ties, :321): a prefix maximum
        // of position-normalised keys with the lane as a tag, as in fill_local16.hip.
        // (a chain that leaves its lane dead arrives dead everywhere, and so does row 0's opener, go + ge < 0 <= tfloor: with no live
        // exit in the wave there is nothing for the scan to carry and nothing for pass 2 to repair)
        chains_dead = __ballot(cl.Iw >= live_thr) == 0ull;
        if (!chains_dead) chain_across_lanes(cl.Iw, cl.extn != 0u, pos_x, gl_x, Iin, extin);
        }

        RPROF(4)
        // ---- pass 2: where the arriving chain is alive, its extended bits and its merge (row_alive); per group the lane's running
        // records, y-suffix records and the traceback dword; in the last column also the int32 arrays the fix-up kernel reads -----------
        const __amdgpu_buffer_rsrc_t rtb = __builtin_amdgcn_make_buffer_rsrc(tb0 + (size_t)(j - 1) * Rtot, 0, 0x7FFFFFFF, RSRC_WORD3);
        const gptr<uint32_t> tbcol = (gptr<uint32_t>)as_global(tb0 + (size_t)(j - 1) * Rtot), tbcol_hi = tbcol + 12 * 64;
        Recs R; R.bw = 0; R.gw = 0; R.g1 = 0;
        const uint32_t ycol = n - j;
        uint32_t tbw0 = 0;                                               // the traceback dword of row m's group (gm: 0 or 1)
        const bool mine = lane == mlane;
        // register k of group gm holds a row below m (a row that takes part in the records) unless this is the lane of row m and
        // k <= pad (k < pad: no row at all; k == pad: row m itself)
#define ROWM_GROUP(g) ((g) < 2 && (uint32_t)(g) == gm && mine)
// (A/B, -DSTITCH_YREC_B64: one 64-bit store per record instead of two dword stores — half the store instructions, the same bytes: cfg5 22.6
// against 22.9-23.3 reads/s, gpurun_out/r4yb: the records cost what their bytes cost the write path, not their issue slots)
#ifdef STITCH_YREC_B64
#define YREC_STORE(t, off) { u32x2 rec_; rec_.x = (t); rec_.y = ycol; __builtin_amdgcn_raw_buffer_store_b64(rec_, ryr, vo, (off), 0); }
#else
#define YREC_STORE(t, off) { __builtin_amdgcn_raw_buffer_store_b32((t), ryr, vo, (off), 0); __builtin_amdgcn_raw_buffer_store_b32(ycol, ryr, vo, (off) + 4, 0); }
#endif
#define P2TAIL(g) \
            const uint32_t t3 = (ROWM_GROUP(g) && 3u <= pad) ? 0u : S[4 * (g) + 3], t2 = (ROWM_GROUP(g) && 2u <= pad) ? 0u : S[4 * (g) + 2]; \
            const uint32_t t1 = (ROWM_GROUP(g) && 1u <= pad) ? 0u : S[4 * (g) + 1], t0 = ROWM_GROUP(g) ? 0u : S[4 * (g)]; \
            const uint32_t g4 = (t3 > t2 ? t3 : t2) > (t1 > t0 ? t1 : t0) ? (t3 > t2 ? t3 : t2) : (t1 > t0 ? t1 : t0); \
            group_records(R, g4, (uint32_t)(g)); \
            if (YB) {                                        /* the cells that hold the column's common word: a bit each; the others as below */ \
                uint32_t g4u = 0u; \
                { const bool w = t3 == Wc; tbw |= w ? 0x80000000u : 0u; const uint32_t u = w ? 0u : t3; g4u = u; } \
                { const bool w = t2 == Wc; tbw |= w ? 0x00800000u : 0u; const uint32_t u = w ? 0u : t2; g4u = u > g4u ? u : g4u; } \
                { const bool w = t1 == Wc; tbw |= w ? 0x00008000u : 0u; const uint32_t u = w ? 0u : t1; g4u = u > g4u ? u : g4u; } \
                { const bool w = t0 == Wc; tbw |= w ? 0x00000080u : 0u; const uint32_t u = w ? 0u : t0; g4u = u > g4u ? u : g4u; } \
                if (__builtin_expect((int32_t)g4u >= ythr, 0)) { \
                    const uint32_t vo = 8u * (uint32_t)lane_x; \
                    if ((int32_t)t3 >= ythr && t3 != Wc) { YREC_STORE(t3, (4 * (g) + 3) * 512) } \
                    if ((int32_t)t2 >= ythr && t2 != Wc) { YREC_STORE(t2, (4 * (g) + 2) * 512) } \
                    if ((int32_t)t1 >= ythr && t1 != Wc) { YREC_STORE(t1, (4 * (g) + 1) * 512) } \
                    if ((int32_t)t0 >= ythr && t0 != Wc) { YREC_STORE(t0, (4 * (g)) * 512) } \
                } \
            } else \
            if ((int32_t)g4 >= ythr) {                       /* (two dword stores: a 64-bit one wants a register PAIR, i.e. a neighbour of the row's register saved and restored) */ \
                const uint32_t vo = 8u * (uint32_t)lane_x; \
                if ((int32_t)t3 >= ythr) { YREC_STORE(t3, (4 * (g) + 3) * 512) } \
                if ((int32_t)t2 >= ythr) { YREC_STORE(t2, (4 * (g) + 2) * 512) } \
                if ((int32_t)t1 >= ythr) { YREC_STORE(t1, (4 * (g) + 1) * 512) } \
                if ((int32_t)t0 >= ythr) { YREC_STORE(t0, (4 * (g)) * 512) } \
            } \
            __builtin_nontemporal_store(tbw, ((g) < 12 ? tbcol : tbcol_hi) + (((g) < 12 ? (g) : (g) - 12) * 64 + lane_x));      /* scalar base + lane offset + immediate (< 4 KB) */ \
            if ((g) < 2 && (uint32_t)(g) == gm) tbw0 = tbw;
        {
            ColA ca;
            ca.GE1 = GE1; ca.GO1 = GO1;
            ca.X = Iin; ca.Sup = SUP_NONE; ca.xext = extin;
            // every lane starts in group gtop - 1: the lanes' alive state is scalar from the start, and group_alive runs under scalar
            // conditions with every lane enabled (what it does to a lane that is not alive, or has no group 0, nobody looks at)
            ca.alive = (no_ins || chains_dead) ? 0ull : __ballot(gl > 0 && Iin >= live_thr);      // (a chain that arrives dead stays dead)
            const unsigned long long have0 = __ballot(has0);
#define P2(g) if (({ asm volatile("" : "+s"(gtop_x)); (uint32_t)(g) < gtop_x; })) { \
            if ((g) == 0) ca.alive &= have0; \
            uint32_t tbw = tbv[g]; \
            if (ca.alive != 0ull) { RCOUNT(1) group_alive<(g)>(S[4 * (g) + 3], S[4 * (g) + 2], S[4 * (g) + 1], S[4 * (g)], tbw, ca, bs_lane); } \
            if ((g) == 0 ? has0_x : true) { \
                P2TAIL(g) \
            } \
        }
            REP20(P2)
#undef P2
        }
#undef P2TAIL
#undef ROWM_GROUP

        RPROF(5)
        // ---- the contig's epilogue: wave reductions over rows < m, row m, the column arg-max granule --------------------------------
        {
            const uint32_t xw = wave_max_u32(R.bw);
            // the four words of group G of lane L as the records saw them (the registers of row m and of no row count as 0)
            auto fetch4 = [&](const uint32_t G, const int L, uint32_t (&w)[4]) {
                w[0] = w[1] = w[2] = w[3] = 0u;
#define FETCH(g) case (g): w[3] = (uint32_t)__builtin_amdgcn_readlane((int)S[4 * (g) + 3], L); w[2] = (uint32_t)__builtin_amdgcn_readlane((int)S[4 * (g) + 2], L); \
                           w[1] = (uint32_t)__builtin_amdgcn_readlane((int)S[4 * (g) + 1], L); w[0] = (uint32_t)__builtin_amdgcn_readlane((int)S[4 * (g)], L); break;
                switch (G) { REP20(FETCH) default: break; }      // (a search tree of scalar compares, not twenty in a row)
#undef FETCH
                if (G == gm && L == mlane) { w[0] = 0u; if (pad >= 1u) w[1] = 0u; if (pad >= 2u) w[2] = 0u; if (pad >= 3u) w[3] = 0u; }
            };
            // 0-based row of register 4 G + k of lane L
            auto row_of = [&](const uint32_t G, const uint32_t k, const uint32_t L) -> uint32_t {
                const uint32_t glL = gq + (L < grem ? 1u : 0u), rbL = 4u * (L * gq + (L < grem ? L : grem)), rshL = (grem > 0 && L >= grem) ? 4u : 0u;
                return rbL + 4u * glL - 1u - (4u * G + k - rshL);
            };
            // The granule the other waves wait for needs the column arg-max (largest score, topmost row, its length) and row m; the
            // x-suffix running maximum's ROW is only stored (Lx): it is looked up after the granule has gone out.
            // rows ascend with the lane and, inside a lane, from register 4 gl - 1 down to 0: the topmost holder of a value is in the
            // lowest lane that has it, in the group that lane met it in first, in the highest register
            XsRec xb_;
            CmRec cb_;
            uint32_t w[4]; int L1 = -1; uint32_t G1 = 0;
            const uint32_t smax = xw >> 16;
            if (xw == 0u) {
                // every S word below row m is 0: the first row takes the running value (0 > MIN, :408-417) — unless there is no row
                if (m > 1) { xb_.v = 0; xb_.len = 0; } else { xb_.v = MIN_SCORE; xb_.len = 0; }
            }
            else { xb_.v = (int32_t)smax; xb_.len = xw & 0xFFFFu; }
            // column arg-max over rows 0..m-1: the first row holding the largest score; row 0 holds S = 0
            if (smax == 0u) { cb_.v = 0; cb_.row = 0; cb_.len = 0; }
            else {
                L1 = (int)__builtin_ctzll(__ballot((R.bw >> 16) == smax));
                G1 = (uint32_t)__builtin_amdgcn_readlane((int)R.g1, L1);
                fetch4(G1, L1, w);
                const uint32_t k1 = (w[3] >> 16) == smax ? 3u : (w[2] >> 16) == smax ? 2u : (w[1] >> 16) == smax ? 1u : 0u;
                cb_.v = (int32_t)smax; cb_.row = row_of(G1, k1, (uint32_t)L1) + 1u; cb_.len = w[k1] & 0xFFFFu;
            }
            // ---- row m (:350-351 seeded selection, :406-447 for i == m): register 4 gm + pad of lane mlane ---------------------------
            const uint32_t wm = gm == 0 ? (pad == 0 ? S[0] : pad == 1 ? S[1] : pad == 2 ? S[2] : S[3]) : (pad == 0 ? S[4] : pad == 1 ? S[5] : pad == 2 ? S[6] : S[7]);
            const int32_t dgm = cx.dgm;
            const uint32_t bytem = (tbw0 >> (8u * pad)) & 0xFFu;
            const int32_t ownW = (int32_t)__builtin_amdgcn_readlane((int)wm, mlane);
            const int32_t ownDG = __builtin_amdgcn_readlane(dgm, mlane);
            const uint32_t ownByte = (uint32_t)__builtin_amdgcn_readlane((int)bytem, mlane);
            const int32_t ownS = word_score(ownW); const uint32_t ownMv = ownByte & 7u, ownSl = word_len(ownW);
            int32_t Sm; uint32_t Slm, mvm;
            bool do_x_m = false;
            if (rowm_run_wins(xb_.v, ownS, word_score(ownDG))) { Sm = xb_.v; Slm = xb_.len; mvm = MK_XSUF; }
            else { Sm = ownS; Slm = ownSl; mvm = ownMv; if (ownSl > xb_.len) do_x_m = true; }
            const uint32_t smw = (uint32_t)word_make(Sm, Slm);
            // the column arg-max is complete: announce it before the column's remaining work (the other waves wait for nothing else)
            if (Sm > cb_.v) { cb_.v = Sm; cb_.row = m; cb_.len = Slm; }
            own_gran = ((unsigned long long)j << 48) | ((unsigned long long)(uint32_t)(cb_.v & 0xFFFF) << 32) | ((unsigned long long)((cb_.len + 1u) & 0xFFFFu) << 16) | (cb_.row & 0xFFFFu);
            if (lane == 0) {
                const unsigned long long gran = own_gran;
                u32x2 g2; g2.x = (uint32_t)gran; g2.y = (uint32_t)(gran >> 32);
                // one aligned 8-byte write, agent scope (the aux bits must be an immediate)
                if (gran_aux == (uint32_t)AUX_VOLATILE) __builtin_amdgcn_raw_buffer_store_b64(g2, rxc, 0u, ((j & 1u) * C + kmine) * 8u, AUX_VOLATILE);
                else __builtin_amdgcn_raw_buffer_store_b64(g2, rxc, 0u, ((j & 1u) * C + kmine) * 8u, AUX_SC1 | AUX_VOLATILE);
            }
            // the x-suffix running maximum's row (1-based; 0: none): the topmost row holding the largest WORD
            if (xw == 0u) xb_.row = m > 1 ? 1u : 0u;
            else {
                const int Lw = (int)__builtin_ctzll(__ballot(R.bw == xw));
                const uint32_t Gw = (uint32_t)__builtin_amdgcn_readlane((int)R.gw, Lw);
                if (Lw != L1 || Gw != G1) fetch4(Gw, Lw, w);
                const uint32_t kw = w[3] == xw ? 3u : w[2] == xw ? 2u : w[1] == xw ? 1u : 0u;
                xb_.row = row_of(Gw, kw, (uint32_t)Lw) + 1u;
            }
            const uint32_t lx = (do_x_m || xb_.row == 0u) ? 0u : m - xb_.row;
            if (cb_.v > vrun) vrun = cb_.v;
            cmax_prev = cb_.v;
            rowm_xsuf = mvm == MK_XSUF; rowm_S = Sm; rowm_len = Slm;
            if (mine) {                                     // the register of row m takes the seeded result
                if (gm == 0) { if (pad == 0) S[0] = smw; else if (pad == 1) S[1] = smw; else if (pad == 2) S[2] = smw; else S[3] = smw; }
                else { if (pad == 0) S[4] = smw; else if (pad == 1) S[5] = smw; else if (pad == 2) S[6] = smw; else S[7] = smw; }
            }
            if (mine) {
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(mvm | (ownByte & (TBB_IEXT | TBB_DEXT))), rtb, 4u * (uint32_t)lane + pad, gm * 256u, 0);
                const uint32_t rl = lastcol ? (do_x_m ? ownSl : xb_.len) : 0u;
                if (Sm >= ybase) {
                    const uint32_t yi = padreg * 64u + (uint32_t)lane;
                    bool upd = Slm > 0u;
                    if (lastcol) { const u32x2 old = __builtin_amdgcn_raw_buffer_load_b64(ryr, 8u * yi, 0, 0); const int32_t sn_ = word_score((int32_t)old.x); upd = Sm > sn_ || (Sm == sn_ && Slm > rl); }
                    if (upd) { u32x2 rec; rec.x = smw; rec.y = n - j; __builtin_amdgcn_raw_buffer_store_b64(rec, ryr, 8u * yi, 0, 0); }
                }
                Lx[(size_t)c * (n + 1) + j] = lx;
            }
        }
#ifdef STITCH_PROFILE
        { const uint32_t pf_e = (uint32_t)__builtin_readcyclecounter(); const int o = no_ins ? 3 : 0; pf_cls[o] += 1u; pf_cls[o + 1] += pf_e - pf_c0; pf_cls[o + 2] += pf_p1 - pf_p0; }
#endif
    }
    if (kmine == 0 && lane == 0) {
        unsigned long long* const ck = (unsigned long long*)((uint8_t*)V.err + ERR_CLOCK_OFF);
        ck[0] = __builtin_readcyclecounter() - clk_c0; ck[1] = wall_clock64() - clk_w0;
    }
    // ---- column n's arrays for the fix-up kernel (single_contig_aligner.rs:453-555): the final words, and the insertion chain at
    // every row — recomputed here from the final words (an opener taken from a merged cell gives the same chain, see above)
    {
        // (the lane's geometry afresh from an untraceable lane number, as in the unpack loop below: nothing of it carried through the column loop)
        uint32_t lane_o = threadIdx.x & 63u; asm volatile("" : "+v"(lane_o));
        const uint32_t gl_o = gq + (lane_o < grem ? 1u : 0u), rsh_o = (grem > 0 && lane_o >= grem) ? 4u : 0u;
        const bool has0_o = gl_o > 0 && rsh_o == 0u;
        uint32_t gl_x = gl_o; int32_t pos_x = (int32_t)(4u * (lane_o * gq + (lane_o < grem ? lane_o : grem)) + 4u * gl_o);
        const uint32_t rg4_x = 4u * (roff + (uint32_t)pos_x - 1u + rsh_o);       // 4 x (linear row index of the row register 0 holds, or would hold)
        const bool mine = (int)lane_o == mlane;
        int32_t L = CHAIN_NONE; bool lext = false;
#define GUARD(g) ((g) == 0 ? has0_o : (uint32_t)(g) < gtop)
#define LASTA(g) if (GUARD(g)) { _Pragma("unroll") for (int k = 3; k >= 0; --k) { \
            const int32_t ext = L + GE1, open = (int32_t)S[4 * (g) + k] + GO1; lext = word_score(ext) >= word_score(open); L = lext ? ext : open; } }
        REP20(LASTA)
#undef LASTA
        int32_t I; uint32_t extin;
        chain_across_lanes(L, lext, pos_x, gl_x, I, extin);
#define LASTB(g) if (GUARD(g)) { _Pragma("unroll") for (int k = 3; k >= 0; --k) { \
            if (!((g) < 2 && (uint32_t)(g) == gm && mine && (uint32_t)k < pad)) { \
                const uint32_t vo = rg4_x - 4u * (4 * (g) + k); \
                __builtin_amdgcn_raw_buffer_store_b32((uint32_t)word_score((int32_t)S[4 * (g) + k]), rS, vo, 0, 0); __builtin_amdgcn_raw_buffer_store_b32(word_len((int32_t)S[4 * (g) + k]), rS, vo, oSlen, 0); \
                __builtin_amdgcn_raw_buffer_store_b32((uint32_t)word_score(I), rS, vo, oIval, 0); __builtin_amdgcn_raw_buffer_store_b32(word_len(I), rS, vo, oIlen, 0); \
            } \
            const int32_t ext = I + GE1, open = (int32_t)S[4 * (g) + k] + GO1; I = word_score(ext) >= word_score(open) ? ext : open; } }
        REP20(LASTB)
#undef LASTB
#undef GUARD
    }
#ifdef STITCH_PROFILE
    RPROF(6)
    if (lane == 0) { unsigned long long* const pf = (unsigned long long*)((uint8_t*)V.err + 16); for (int k = 0; k < 8; ++k) atomicAdd(pf + k, (unsigned long long)pf_sum[k]); atomicAdd(pf + 8, 1ull); atomicAdd(pf + 9, (unsigned long long)pf_cnt[0]); atomicAdd(pf + 10, (unsigned long long)pf_cnt[1]); for (int k = 0; k < 6; ++k) atomicAdd(pf + 11 + k, (unsigned long long)pf_cls[k]); }
#endif
    // ---- (YB) the records that went out as bits: for every row the LAST column whose bit is set, written where the stores would have left
    // it.  A row's records are written in column order and each replaces the one before, so the slot must end up with the later of {the
    // 8-byte record it holds, the row's last bit}: n - j of the later is the smaller.  Columns are scanned from the last one back, a row is
    // done at its first bit; in a contig the read is not on a quarter of the rows have their bit in any column, so the wave is through
    // after a few dozen columns — and where W stayed below the threshold for long (the contig the read ends in) the columns without bits are
    // skipped 64 at a time.
    if (YB) {
        uint32_t lane_o = threadIdx.x & 63u; asm volatile("" : "+v"(lane_o));
        const uint32_t gl_o = gq + (lane_o < grem ? 1u : 0u), rsh_o = (grem > 0 && lane_o >= grem) ? 4u : 0u;
        const bool has0_o = gl_o > 0 && rsh_o == 0u, mine_o = (int)lane_o == mlane;
        const gptr<u32x2> yrec_o = (gptr<u32x2>)as_global(V.D);
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((uint8_t*)(V.Wcol + (size_t)c * (n + 1)), 0, 0x7FFFFFFF, RSRC_WORD3);
        // bit 7 of byte k of done[g]: register 4 g + k of this lane has nothing (more) to find — no row, row m, or found
        uint32_t done[NG]; uint32_t pending = 0;
#define DINIT(g) { uint32_t v = ((g) == 0 ? has0_o : (gl_o > 0 && (uint32_t)(g) < gtop)) ? 0u : 0x80808080u; \
            if ((g) < 2 && (uint32_t)(g) == gm && mine_o) v |= 0x80808080u >> (8u * (3u - pad));      /* registers 0..pad of the group: no row, and row m (its records are the epilogue's) */ \
            done[g] = v; pending += 4u - (uint32_t)__builtin_popcount(v); }
        REP20(DINIT)
#undef DINIT
        for (uint32_t jb = n; jb >= 1u && __ballot(pending != 0u) != 0ull; jb = jb > 64u ? jb - 64u : 0u) {
            // 64 columns' common words at once, lane l holding column jb - l
            const uint32_t wv_ = lane_o < jb ? __builtin_amdgcn_raw_buffer_load_b32(rW, 4u * (jb - lane_o), 0, 0) : 0xFFFFFFFFu;
            unsigned long long cols = __ballot(wv_ != 0xFFFFFFFFu);
            while (cols != 0ull && __ballot(pending != 0u) != 0ull) {
                const int l = (int)__builtin_ctzll(cols); cols &= cols - 1ull;
                const uint32_t j = jb - (uint32_t)l, Wj = (uint32_t)__builtin_amdgcn_readlane((int)wv_, l), yc = n - j;
                const __amdgpu_buffer_rsrc_t rtbj = __builtin_amdgcn_make_buffer_rsrc(tb0 + (size_t)(j - 1) * Rtot, 0, 0x7FFFFFFF, RSRC_WORD3);
#define SCAN(g) if ((uint32_t)(g) < gtop) { \
                    const uint32_t nb = __builtin_amdgcn_raw_buffer_load_b32(rtbj, 4u * lane_o, (g) * 256, 0) & 0x80808080u & ~done[g]; \
                    if (nb != 0u) { \
                        _Pragma("unroll") for (int k = 0; k < 4; ++k) if (nb & (0x80u << (8 * k))) { \
                            const uint32_t at = roff + (4u * (g) + (uint32_t)k) * 64u + lane_o; \
                            if (yrec_o[at].y > yc) { u32x2 rec; rec.x = Wj; rec.y = yc; yrec_o[at] = rec; } \
                        } \
                        done[g] |= nb; pending -= (uint32_t)__builtin_popcount(nb); \
                    } }
                REP20(SCAN)
#undef SCAN
            }
        }
    }
    // ---- unpack the y-suffix records of this wave's rows into the arrays the fix-up kernel reads (its own stores: no barrier) -------
    // (the lane's geometry worked out afresh from a lane number the compiler cannot trace back: what it would otherwise carry through the
    // column loop for this epilogue — addresses, row counts — it had to spill to scratch memory, which a persistent launch must not use)
    {
        uint32_t lane_o = threadIdx.x & 63u; asm volatile("" : "+v"(lane_o));
        const uint32_t gl_o = gq + (lane_o < grem ? 1u : 0u), nrows_o = 4u * gl_o;
        const uint32_t rowbase_o = 4u * (lane_o * gq + (lane_o < grem ? lane_o : grem)), rsh_o = (grem > 0 && lane_o >= grem) ? 4u : 0u;
        const gptr<u32x2> yrec_o = (gptr<u32x2>)as_global(V.D);
#pragma unroll 1
        for (uint32_t i = 0; i < nrows_o; ++i) {
            const uint32_t row = rowbase_o + (nrows_o - 1 - i);
            if (row < m) {
                const u32x2 rec = yrec_o[roff + (i + rsh_o) * 64u + lane_o];
                V.Sn[roff + row] = word_score((int32_t)rec.x); V.SnLen[roff + row] = word_len((int32_t)rec.x); V.Ly[roff + row] = (YB && rec.y == 0xFFFFFFFFu) ? 0u : rec.y;
            }
        }
    }
    if (!streaming) return;
    // ---- persistent teams: this wave's results are complete.  They are written back beyond this XCD's L2 (the fix-up / walk kernel the
    // host starts for the read runs on any CU), THEN the wave counts itself in; the wave that completes the count tells the host.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the compiler may drop the wait behind the write-back when it thinks nothing is outstanding)
    if (lane == 0) {
        const uint32_t before = atomicAdd(qp->cnt + job, 1u);
        if (before + 1u == nact) __hip_atomic_store(qp->h_done + job, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    }
}

uint32_t fill_regs_rows_per_wave() { return 64u * RMAX; }
// workgroups of `waves` waves one CU holds at once, as the runtime's occupancy calculator sees it (the host never launches more
// workgroups than CUs x this: all workgroups of a read must be resident, they wait for each other every column)
int fill_regs_workgroups_per_cu(uint32_t waves) {
    const void* kernels[8] = {(const void*)fill_regs_kernel<1, false, false>, (const void*)fill_regs_kernel<4, false, false>, (const void*)fill_regs_kernel<1, true, false>, (const void*)fill_regs_kernel<4, true, false>,
                              (const void*)fill_regs_kernel<1, false, true>, (const void*)fill_regs_kernel<4, false, true>, (const void*)fill_regs_kernel<1, true, true>, (const void*)fill_regs_kernel<4, true, true>};
    int least = 1 << 30;
    for (const void* k : kernels) {
        // (more than 64 KiB of dynamic LDS has to be allowed explicitly)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { (void)hipGetLastError(); return 0; }
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, (int)waves * 64, (size_t)waves * LDS_PER_WAVE) != hipSuccess) { (void)hipGetLastError(); return 0; }
        least = nb < least ? nb : least;
    }
    return least;
}
// max_nact: the largest number of active contigs of any job of the launch; circular: opts.circular
// q == nullptr: a classic launch (wave_map[w].x = the job); else persistent teams that pull jobs off the queue (wave_map[w].x = the team)
void launch_fill_regs(const JobView* d_jobs, const uint2* d_wave_map, uint32_t n_waves, uint32_t waves, uint32_t max_nact, bool circular, bool ybits, const FillShared& sh, const StreamCtl* q, hipStream_t stream) {
    const dim3 grid((n_waves + waves - 1) / waves), block(waves * 64); const size_t lds = (size_t)waves * LDS_PER_WAVE;
#define GO(NQ_, CIRC_, YB_) hipLaunchKernelGGL((fill_regs_kernel<NQ_, CIRC_, YB_>), grid, block, lds, stream, d_jobs, sh, d_wave_map, n_waves, q)
    if (max_nact <= 64) { if (circular) { if (ybits) GO(1, true, true); else GO(1, true, false); } else { if (ybits) GO(1, false, true); else GO(1, false, false); } }
    else { if (circular) { if (ybits) GO(4, true, true); else GO(4, true, false); } else { if (ybits) GO(4, false, true); else GO(4, false, false); } }
#undef GO
}

}  // namespace stitch
