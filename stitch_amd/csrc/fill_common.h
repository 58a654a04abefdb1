// Device helpers shared by the Local-mode fill kernels (fill_local16.hip: row state streamed through memory; fill_regs.hip:
// row state resident in registers): address-space-1 pointers, DPP wave scans and reductions, and the per-column records of a
// contig (DESIGN.md "Kernel 1").  Everything here sits in an anonymous namespace of the including file.
#pragma once
#include <hip/hip_runtime.h>
#include "dp_core.h"
#include "walk_core.h"

namespace stitch {

struct FillShared {              // column-0 state of every aligner, evaluated once per context on the host (col0_init)
    const int32_t* S0; const uint32_t* Slen0; const int32_t* Sn0; const uint8_t* SnSet0; const uint8_t* Smove0;
    const uint32_t* lx0; const JumpBase* base0;
};

// Persistent teams (fill_regs.hip, stitch_api.cpp run_jobs_streaming): a launch's teams — the waves of one read, one per active contig —
// pull the next read from a queue when theirs ends, instead of a launch ending with its slowest read.  `next` == nullptr: the
// classic launch, wave_map[w].x IS the job.  The words the HOST writes or reads while the kernel runs live in pinned host memory
// (system-scope accesses, never cached on the device); the queue head, the per-job wave counters and the teams' mailboxes are
// device memory touched with agent-scope accesses only, like the column granules.
struct StreamCtl {
    uint32_t* next;                     // device: the next job to hand out ([16]: teams that have left on the host's request)
    uint32_t* cnt;                      // device [n_jobs]: waves of the job that have stored all their results
    unsigned long long* mbox;           // device [teams]: {sequence number, job} announced by a team's first wave
    uint32_t* h_ready;                  // pinned host: jobs below this number may start (the arena block they use is free again)
    uint32_t* h_abort;                  // pinned host: 0 = carry on, 1 .. 2^31 - 1 = so many teams leave at their next read, above = hand out no more jobs
    uint32_t* h_done;                   // pinned host [n_jobs]: set by the last wave of a job once every wave's results are written back
    uint32_t* h_err;                    // pinned host: set by a wave that gave up waiting (partner, mailbox or block)
    uint32_t n_jobs;
};

namespace {

// Pointers loaded from the JobView are generic ("flat") to the compiler; flat loads cannot be waited for with a counted
// vmcnt (cdna_hip_programming.md: flat_* return out of order), which would serialise the software pipeline.  Casting them
// to address space 1 once makes every access a global_load/global_store.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // native vectors: HIP's u32x4 class has no address-space-1 overloads
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <typename T> using gptr = T __attribute__((address_space(1)))*;
template <typename T> __device__ __forceinline__ gptr<T> as_global(T* p) { return (gptr<T>)(uintptr_t)p; }

constexpr int MAXC = 256;
// slot.x = contig | tile << 8 | flags.  A wave's slots are a contiguous range of the workgroup's (contig, tile) list, so
// a contig may start in one wave and end in the next: CIN = first slot of a segment that continues another wave's work
// (wait for its carries in LDS), COUT = last slot of a segment that stops before the contig's last tile (publish them).
constexpr uint32_t SLOT_FIRST = 0x40000000u, SLOT_LAST = 0x80000000u, SLOT_CIN = 0x20000000u, SLOT_COUT = 0x10000000u, SLOT_TILE_MASK = 0xFFFFFu;
constexpr int DPP_ROW_SHR0 = 0x110, DPP_WAVE_SHR1 = 0x138, DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143;

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_mov(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ int from_prev_lane(int v, int lane0) { return dpp_mov<DPP_WAVE_SHR1>(lane0, v); }   // lane-1's v; lane 0 gets lane0
__device__ __forceinline__ int lane_bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ void scan_step(ScanEl& inc) {
    ScanEl o; o.key = dpp_mov<CTRL, ROW_MASK>(INT32_MIN, inc.key); o.q = dpp_mov<CTRL, ROW_MASK>(0, inc.q);
    if (o.key >= inc.key) inc = o;             // the source lane holds earlier rows: it wins ties
}
__device__ __forceinline__ void wave_scan(ScanEl& inc) {   // inclusive scan with scan_combine
    scan_step<DPP_ROW_SHR0 | 1>(inc); scan_step<DPP_ROW_SHR0 | 2>(inc); scan_step<DPP_ROW_SHR0 | 4>(inc); scan_step<DPP_ROW_SHR0 | 8>(inc);
    scan_step<DPP_BCAST15, 0xA>(inc); scan_step<DPP_BCAST31, 0xC>(inc);
}
// Wave reductions on DPP (VALU rate; a ds_bpermute butterfly is six dependent LDS round trips): the inclusive-scan steps of
// wave_scan leave the total in lane 63.  Lanes a step does not reach keep `identity`.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t identity, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    { const uint32_t o = dpp_u32<DPP_ROW_SHR0 | 1>(0u, v); v = o > v ? o : v; } { const uint32_t o = dpp_u32<DPP_ROW_SHR0 | 2>(0u, v); v = o > v ? o : v; }
    { const uint32_t o = dpp_u32<DPP_ROW_SHR0 | 4>(0u, v); v = o > v ? o : v; } { const uint32_t o = dpp_u32<DPP_ROW_SHR0 | 8>(0u, v); v = o > v ? o : v; }
    { const uint32_t o = dpp_u32<DPP_BCAST15, 0xA>(0u, v); v = o > v ? o : v; } { const uint32_t o = dpp_u32<DPP_BCAST31, 0xC>(0u, v); v = o > v ? o : v; }
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return ~wave_max_u32(~v); }
// max of 64-bit keys hi:lo as two 32-bit reductions: the largest hi, then the largest lo among the lanes that hold it
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    const uint32_t mh = wave_max_u32(hi);
    const uint32_t ml = wave_max_u32(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 32) | ml;
}

// Rows of a tile are independent, and the scheduler would interleave all eight of them: dozens of compare masks (SGPR
// pairs) live at once, far beyond the 102 SGPRs of a wave, i.e. v_writelane/v_readlane spill traffic in the inner loop.
// With three waves per SIMD nothing is lost by finishing one row before the next (measured: no difference without the
// fence, with pairs of rows or row by row): the fence keeps the schedule row-serial.
#ifndef STITCH_NO_ROW_FENCE
#define ROW_FENCE __builtin_amdgcn_sched_barrier(0);
#else
#define ROW_FENCE
#endif

struct WaveCol {                 // wave-uniform state of one contig's column
    int32_t JSW, JSW1;           // word of the column's best jump without the match term; same for row 1 (circular contigs)
    int32_t vrun;                // contig's running maximum up to column j-1
    int32_t thr;                 // max(vrun << 16, 1): an S word >= thr has score >= vrun and a non-zero length
    uint32_t m, roff, j, n;
    uint32_t q;
#ifdef STITCH_PROFILE
    uint32_t n_tiles = 0, n_merge = 0, n_c2 = 0;
#endif
    int32_t upS, upT;            // carries (words): S[prev][i0-1] and S'[curr][i0-1]
    ScanEl carry;
};
struct LaneAcc {                 // per-lane running records over a contig's column (rows < m)
    uint32_t xw, xrow;           // best S word and its (lowest) row: the x-suffix running max (:406-429)
    uint32_t ck;                 // max of S<<16 | (0xFFFF - row): column arg-max, lowest row (:677-697)
    uint32_t cklen;              // S.len of the row that holds ck
};
// S.len of the column arg-max: the one lane whose record equals the reduced maximum holds it (rows are part of the key)
__device__ __forceinline__ uint32_t ck_len_of(const LaneAcc& acc, uint32_t ck) {
    const unsigned long long who = __ballot(acc.ck == ck && ck != 0u);
    return who ? (uint32_t)__builtin_amdgcn_readlane((int)acc.cklen, (int)__builtin_ctzll(who)) : 0u;
}
struct RowM { int32_t F; uint32_t mv, bits; int32_t BD, DG; };   // row m's own selection, finalised after the reduction
struct WordConsts { int32_t MW, XW, GE1, GO1, ge, kb0; };        // match/mismatch << 16, gap words, ge, go + ge
// Per-lane constants of the insertion scan.  Keys and lengths are taken relative to the tile (row index within the tile,
// iL = lane * R + 1 for a lane's first row), so they do not depend on the tile; the wave's carry is rebased by one tile
// (ge * TILE, TILE) when it moves on.  A tile-relative key lies in (-2^24, 2^23) for every scoring local16_ok admits
// (|ge| * 256 + 32767 + |go + ge| < 2^23), which leaves the low 6 bits of a word for a lane tag: the wave-level scan is then
// a plain max (earlier lanes carry the larger tag, so they win ties like the reference's extension, :321) and the winner's
// length term is fetched from the lane the tag names.
struct LaneK { int32_t giL; uint32_t iL; int32_t tag; };      // ge * iL, iL, 63 - lane
constexpr int32_t SCAN_LOW = -(1 << 24);       // below every real tile-relative key: the chain's seed and the padding rows

}  // namespace
}  // namespace stitch
