// C ABI of the MI355X-native `stitch align` hot path (include/stitch_gpu.h).
//
// Per batch this file does what Aligners::align does per read (fg-stitch-lib/src/align/aligners/mod.rs:237-340):
//   pass 1  one DP job per distinct read  -> fill kernel -> fix-up + walk kernel (primary chain, or one candidate
//           chain per end contig when --suboptimal)
//   host    remove_clipping, traceback_all's end-contig selection, realign_origin planning (mod.rs:442-553)
//   pass 2  the rotated-read re-alignments of circular contigs as further DP jobs on the contig subset of the chain
//   host    accept / split_at_y, suboptimal filter, result arena
// There is no CPU DP in this library: without a working HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <vector>
#include <thread>
#include <atomic>

#include "../../include/stitch_gpu.h"
#include "dp_core.h"
#include "host_align.h"
#include "walk_core.h"
#include "prealign.h"

namespace stitch {
void launch_banded_scores(const BandPair* d_pairs, const uint32_t* d_which, uint32_t n_pairs, const BandScoring& sc, const uint8_t* d_reads, const uint8_t* d_contigs,
                          const uint16_t* d_bands, int32_t* d_state, int32_t* d_scores, const uint32_t* d_cls, uint32_t my_class, hipStream_t stream);
uint32_t banded_ring_rows();
bool launch_banded_scores_lds(const BandPair* d_pairs, const uint32_t* d_which, uint32_t n_pairs, uint32_t max_m, const BandScoring& sc, const uint8_t* d_reads,
                              const uint8_t* d_contigs, const uint16_t* d_bands, int32_t* d_scores, const uint32_t* d_cls, hipStream_t stream);
void launch_full_scores(const BandPair* d_pairs, const uint32_t* d_which, uint32_t n_full, uint32_t max_m, const BandScoring& sc, const uint8_t* d_reads,
                        const uint8_t* d_contigs, int32_t* d_scores, hipStream_t stream);
uint32_t full_score_max_rows();
bool launch_full_scores_skew16(const BandPair* d_pairs, const uint32_t* d_which, uint32_t n_full, uint32_t max_m, uint32_t max_n, const BandScoring& sc,
                               const uint8_t* d_reads, const uint8_t* d_contigs, int32_t* d_scores, hipStream_t stream);
uint32_t band_device_max_cols();
void launch_banded_scores_general(const BandPair32* d_pairs, uint32_t n_pairs, const BandScoringClip& sc, const uint8_t* d_reads, const uint8_t* d_contigs,
                                  const uint32_t* d_bands, int32_t* d_state, int32_t* d_scores, hipStream_t stream);
void launch_band_draw(const BandPair* d_pairs, const uint32_t* d_which, uint32_t n_pairs, uint32_t max_n, const BandElem* d_elems, uint32_t w, uint32_t ring_rows,
                      bool window, uint16_t* d_bands, uint32_t* d_cls, uint32_t* d_class_counts, hipStream_t stream);
bool band_fits_window(const uint16_t* lo, const uint16_t* hi, uint32_t m, uint32_t n);
bool window_scoring_ok(const BandScoring& sc, uint32_t max_m);
void launch_banded_scores_window(const BandPair* d_pairs, const uint32_t* d_which, uint32_t n_pairs, const BandScoring& sc, const uint8_t* d_reads,
                                 const uint8_t* d_contigs, const uint16_t* d_bands, int32_t* d_scores, const uint32_t* d_cls, hipStream_t stream);
struct FillShared {
    const int32_t* S0; const uint32_t* Slen0; const int32_t* Sn0; const uint8_t* SnSet0; const uint8_t* Smove0;
    const uint32_t* lx0; const JumpBase* base0;
};
struct StreamCtl {                      // == fill_common.h (persistent teams of fill_regs.hip)
    uint32_t* next; uint32_t* cnt; unsigned long long* mbox; uint32_t* h_ready; uint32_t* h_abort; uint32_t* h_done; uint32_t* h_err; uint32_t n_jobs;
};
void launch_fill(const JobView* d_jobs, uint32_t n_jobs, int waves, const FillShared& sh, hipStream_t stream);
void launch_fixup_walk(const JobView* d_jobs, const WalkArgs* d_args, uint32_t n_jobs, uint32_t max_nact_mode1, hipStream_t stream, uint32_t max_wgs = 0, uint32_t max_nact = 0);
constexpr size_t PIN_BYTES = (size_t)64 << 20;   // pinned staging buffer for result downloads
constexpr uint32_t TILE_ROWS = 512;   // 64 lanes x 8 rows (Local-mode kernel; the generic one uses 256): contig row blocks are padded to this
void launch_fill_local16(const JobView* d_jobs, uint32_t n_jobs, uint32_t G, int waves, uint32_t slots_cap, const FillShared& sh, hipStream_t stream);
uint32_t fill_local16_max_slots();
void launch_fill_regs(const JobView* d_jobs, const uint2* d_wave_map, uint32_t n_waves, uint32_t waves, uint32_t max_nact, bool circular, bool ybits, const FillShared& sh, const StreamCtl* q /* device copy; nullptr = classic launch */, hipStream_t stream);
uint32_t fill_regs_rows_per_wave();
int fill_regs_workgroups_per_cu(uint32_t waves);
void launch_fill_regs32(const JobView* d_jobs, const uint2* d_wave_map, uint32_t n_waves, uint32_t max_nact, bool circular, const FillShared& sh, hipStream_t stream);
uint32_t fill_regs32_rows_per_wave();
int fill_regs32_workgroups_per_cu();
}  // namespace stitch

using namespace stitch;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(STITCH_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

struct stitch_index {
    std::vector<std::string> names;
    std::vector<std::vector<uint8_t>> seqs;     // upper-cased forward strands
};

struct Aligner { uint32_t target; bool fwd; uint32_t m, troff, seqoff; int32_t opp; };

struct Job {                                     // one full jump DP
    std::vector<uint8_t> y;                      // upper-cased query
    std::vector<uint32_t> act;                   // active aligner ids, ascending (== aligner order)
    int mode = 0;                                // 0 traceback, 1 traceback_all candidates, 2 traceback_from(from)
    uint32_t from = 0;
    // results
    std::vector<HAln> chains;                    // mode 1: one per active contig (status per chain below)
    std::vector<uint32_t> status;
};

// Diagnostic / experiment knobs, read from the environment ONCE when a context is created (never per launch).
struct Knobs {
    bool no_ybits = false;
    bool fail_first_attempt = false;             // test hook: treat the first attempt of every cooperative launch as timed out
    bool fail_keeps_pipeline = false;            // test hook ("pipelined"): ... and keep two fills in flight, so that a repeat runs beside the other window's fill
    bool debug = false, force_generic = false, profile_dump = false, banded_global = false, fill_only = false, no_regs = false, no_regs32 = false, force_regs32 = false, no_pipeline = false, no_fill_overlap = false, prealign_v1 = false, host_bands = false;
    size_t array_align = 0, job_align = 0;
    std::string dump_dir;                        // (debugging) column-n arrays of every job as the fill left them, one file per job
    int max_waves = 0, wg_per_read = 0, tiles_per_wave = 0; long regs_min_rows = -1;
    int regs_waves = 0, regs_map = 0; bool trace = false, no_stream = false, no_join = false, no_wg_poll = false, prealign_general = false; int stream_blocks = 0, stream_teams = 0, stream_range = 0; bool test_stream_retire = false; bool test_stream_stall = false;      // (experiments) waves per workgroup of fill_regs.hip, order of its wave map; launch timeline on stderr
    static Knobs from_env() {
        Knobs k;
        auto num = [](const char* name) -> unsigned long long { const char* e = getenv(name); return e ? strtoull(e, nullptr, 10) : 0ull; };
        k.debug = getenv("STITCH_DEBUG") != nullptr; k.force_generic = getenv("STITCH_FORCE_GENERIC") != nullptr;
        k.profile_dump = getenv("STITCH_PROFILE_DUMP") != nullptr; k.banded_global = getenv("STITCH_BANDED_GLOBAL") != nullptr;
        k.fail_first_attempt = getenv("STITCH_TEST_FAIL_FIRST_ATTEMPT") != nullptr;
        { const char* e = getenv("STITCH_TEST_FAIL_FIRST_ATTEMPT"); k.fail_keeps_pipeline = e && !strcmp(e, "pipelined"); }
        k.fill_only = getenv("STITCH_EXP_FILL_ONLY") != nullptr;      // experiment builds whose results are garbage: time the fill, skip the walk
        k.no_regs = getenv("STITCH_NO_REGS") != nullptr;             // keep the state-streaming kernel even where the register-resident one applies
        k.no_fill_overlap = getenv("STITCH_NO_FILL_OVERLAP") != nullptr;      // two windows, but a fill starts only when the one before it has ended
        k.no_pipeline = getenv("STITCH_NO_PIPELINE") != nullptr;     // one arena window: a launch is finished before the next fill starts
        k.host_bands = getenv("STITCH_PREALIGN_HOST_BANDS") != nullptr;     // the pre-alignment's bands drawn and classified on the host (the path of targets beyond the band kernel's LDS)
        k.prealign_v1 = getenv("STITCH_PREALIGN_V1") != nullptr;     // the pre-alignment's first-generation score kernels (A/B runs, tests)
        k.no_regs32 = getenv("STITCH_NO_REGS32") != nullptr;         // keep the generic kernel where the 32-bit register-resident one applies
        k.force_regs32 = getenv("STITCH_FORCE_REGS32") != nullptr;   // (tests) the 32-bit register-resident kernel also where a 16-bit Local-mode kernel applies
        if (const char* e = getenv("STITCH_DUMP_DIR")) k.dump_dir = e;
        k.array_align = (size_t)num("STITCH_ARRAY_ALIGN"); k.job_align = (size_t)num("STITCH_JOB_ALIGN");
        k.max_waves = (int)num("STITCH_MAX_WAVES"); k.wg_per_read = (int)num("STITCH_WG_PER_READ"); k.tiles_per_wave = (int)num("STITCH_TILES_PER_WAVE");
        k.regs_waves = (int)num("STITCH_REGS_WAVES"); k.regs_map = (int)num("STITCH_REGS_MAP"); k.trace = getenv("STITCH_TRACE") != nullptr;
        k.no_wg_poll = getenv("STITCH_NO_WG_POLL") != nullptr;    // every wave polls its team's granules itself even where a workgroup's waves are one team's (A/B runs)
        k.prealign_general = getenv("STITCH_PREALIGN_GENERAL") != nullptr;      // (tests) the filter's general path (every mode, 32-bit band ranges) also where the fast Local path applies
        k.no_ybits = getenv("STITCH_NO_YBITS") != nullptr;      // per-contig y-suffix records all as 8-byte records, none as a bit of the traceback byte (A/B runs, tests)
        k.no_join = getenv("STITCH_NO_JOIN") != nullptr;          // traceback_all: every chain walked to its start (A/B runs, tests), none joined to the reference chain
        k.no_stream = getenv("STITCH_NO_STREAM") != nullptr;
        // Under rocprofv3 (kernel trace or counters) a launch beside resident teams is not reported complete until the teams' own dispatch
        // is: the tool handles completions in submission order (measured: the first walk of every persistent run "stalls", the run is called
        // off after its bound and continues launch by launch, gpurun_out/collect_r04_a/trace).  A profiled process therefore goes launch by
        // launch from the start, and says so (stitch_timing.stream_runs stays 0); bench.py prints which path ran.
        { const char* pre = getenv("LD_PRELOAD"); if (getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || (pre && strstr(pre, "rocprofiler"))) k.no_stream = true; }      // launch by launch even where persistent teams apply (A/B runs, tests)
        k.stream_blocks = (int)num("STITCH_STREAM_BLOCKS");       // (tests) cap on the arena blocks of a persistent-team run
        k.stream_range = (int)num("STITCH_STREAM_RANGE");         // (experiments) most jobs walked by one fix-up + walk launch
        k.test_stream_retire = getenv("STITCH_TEST_STREAM_RETIRE") != nullptr;      // (tests) the first bounded wait of a run asks one team to leave
        k.test_stream_stall = getenv("STITCH_TEST_STREAM_STALL") != nullptr;      // (tests) the first bounded wait beside resident teams counts as a stall: the run is called off
        k.stream_teams = (int)num("STITCH_STREAM_TEAMS");         // (tests) ... and on its teams, so that small batches queue up behind few teams
        if (const char* e = getenv("STITCH_REGS_MIN_ROWS")) k.regs_min_rows = atol(e);     // (tests: 0 sends every eligible read to fill_regs.hip)
        return k;
    }
};

constexpr uint32_t REGS_WAVES_DEFAULT = 4;
struct stitch_ctx {
    int device = 0;
    Knobs knobs;
    stitch_opts opts{};
    DpParams P{};
    std::vector<TargetInfo> targets;
    std::vector<Aligner> al;
    uint32_t C = 0, T = 0, RtotT = 0, max_m = 0;
    hipStream_t stream = nullptr, stream2 = nullptr;   // stream2: the banded kernel, concurrent with the full-matrix kernel
    hipStream_t stream3 = nullptr;                     // the fills of the second arena window (two fills in flight: run_jobs_in_order)
    hipEvent_t ev2[2] = {nullptr, nullptr};
    uint32_t* pin_cnt = nullptr;                       // pre-alignment: the band kernel's class counts of the two chunks in flight (pinned)
    hipStream_t pstream[3] = {nullptr, nullptr, nullptr};      // the pre-alignment filter's streams
    hipEvent_t evu[2] = {nullptr, nullptr};     // pre-alignment: a chunk's uploads are done
    hipEvent_t evc[2] = {nullptr, nullptr};     // pre-alignment: end of the device work of the chunk in each of the two chunk regions
    // device, context lifetime
    uint8_t* d_xseq = nullptr; int32_t* d_S0 = nullptr; uint32_t* d_Slen0 = nullptr; int32_t* d_Sn0 = nullptr;
    uint8_t* d_SnSet0 = nullptr; uint8_t* d_Smove0 = nullptr; uint8_t* d_Imove0 = nullptr; uint32_t* d_lx0 = nullptr;
    JumpBase* d_base0 = nullptr;
    // device arena reused across launches
    uint8_t* arena = nullptr; size_t arena_bytes = 0;     // arena = arena_raw rounded up to a multiple of 1 GiB
    uint8_t* arena_raw = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t evp[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};      // per arena window: fill start / end, walk start / end
    // results of the last batch
    std::vector<stitch_read_result> rr; std::vector<stitch_chain> chains; std::vector<stitch_op> ops;
    uint8_t* pin = nullptr;                       // pinned staging buffer for result downloads (PIN_BYTES)
    uint8_t* pin_h2d = nullptr; size_t pin_h2d_bytes = 0;   // pinned staging of a launch's per-job inputs
    uint32_t* pin_q = nullptr; size_t pin_q_words = 0;      // persistent teams: the words host and kernel exchange while it runs (pinned, mapped, coherent)
    uint16_t* pin_bands[2] = {nullptr, nullptr}; size_t pin_bands_elems = 0;   // pinned band staging of the pre-alignment pipeline (two chunks)
    std::vector<std::vector<HAln>> job_chains;   // final chains per job (a run of identical reads shares one job)
    std::vector<long> per_read;                  // read -> index into job_chains, -1 = none; for stitch_format_sam
    stitch_timing tm{};
    int n_cus = 256; uint32_t tm_wg_per_read = 1;
    uint32_t regs_waves = 4;                     // waves per workgroup of fill_regs.hip (REGS_WAVES_DEFAULT; STITCH_REGS_WAVES for experiments)
    int regs_wg_per_cu = 0;                      // workgroups of fill_regs.hip one CU holds at once (occupancy query; 0: kernel unusable)
    int regs32_wg_per_cu = 0;                    // ... of fill_regs32.hip (one: a wave takes a SIMD's whole register file)
    bool tm_fast = false;                        // last run_jobs used the Local-mode 16-bit kernel
    bool in_stream_fallback = false;             // run_jobs_streaming is on the stack (its fallback goes launch by launch)
    uint32_t* stream_abort_word = nullptr;        // persistent teams running: the pinned word that calls the run off (set by bounded_sync when a launch beside them does not end)
    bool stream_stalled = false;                 // ... and that it happened
    uint32_t stream_retired = 0, stream_may_retire = 0;      // teams of the current run asked to leave early (bounded_sync), and how many may be (never the last one)
    double stream_bound_s = 0.25;                // how long a launch beside the teams may take before one is asked to
    bool warmed_up = false;                      // a classic launch of the register-resident fill with its walk and downloads has completed in this context
    size_t mem_limit = 0;                        // optional cap on arena bytes (STITCH_ARENA_BYTES), for tests
    // banded pre-alignment filter (prealign.h): host copies of the contig strands, their k-mer indexes, device scratch
    std::vector<uint8_t> h_xseq; std::vector<Strand> strands; KmerIndex kidx;
    uint8_t* pre_buf = nullptr; size_t pre_bytes = 0;
};

static std::vector<uint8_t> revcomp(const std::vector<uint8_t>& s) {             // util/dna.rs:5-41
    static const char* A = "AGCTYRWSKMDVHBN"; static const char* B = "TCGARYWSMKHBDVN";
    uint8_t comp[256]; for (int v = 0; v < 256; ++v) comp[v] = (uint8_t)v;
    for (int k = 0; k < 15; ++k) { comp[(uint8_t)A[k]] = (uint8_t)B[k]; comp[(uint8_t)A[k] + 32] = (uint8_t)(B[k] + 32); }
    std::vector<uint8_t> r(s.size());
    for (size_t k = 0; k < s.size(); ++k) r[k] = comp[s[s.size() - 1 - k]];
    return r;
}

template <typename T> static int upload(T** dst, const std::vector<T>& src) {
    HIP_TRY(hipMalloc((void**)dst, std::max<size_t>(16, src.size() * sizeof(T))));
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return STITCH_OK;
}

extern "C" {

const char* stitch_last_error(void) { return g_err.c_str(); }
const char* stitch_version(void) { return "stitch_amd 0.1 (gfx950)"; }

void stitch_opts_default(stitch_opts* o) {       // Builder defaults, aligners/mod.rs:65-116
    memset(o, 0, sizeof(*o));
    o->mode = 0; o->match_score = 1; o->mismatch_score = -4; o->gap_open = -6; o->gap_extend = -2;
    o->jump_same = o->jump_opposite = o->jump_inter = -10;
    o->circular_slop = 20; o->pre_align_min_score = 100; o->pre_align_subset_contigs = 1; o->kmer_size = 12; o->band_width = 50;
    o->suboptimal_pct = 20.0f; o->filter_secondary_pct = 10.0f;
}

int stitch_index_build(const char* const* names, const uint8_t* const* seqs, const uint32_t* lens, uint32_t n_contigs,
                       stitch_index** out) {
    if (!out || !names || !seqs || !lens) return fail(STITCH_EINVAL, "null argument");
    if (n_contigs == 0) return fail(STITCH_EINVAL, "Found no sequences in the FASTA");            // target_seq.rs:107
    auto idx = std::make_unique<stitch_index>();
    for (uint32_t k = 0; k < n_contigs; ++k) {
        if (lens[k] == 0) return fail(STITCH_EINVAL, "empty contig");
        if (lens[k] > 134217727u) return fail(STITCH_EINVAL, "contig longer than 2^27-1 bp (packed_length_cell.rs:108-110)");
        idx->names.emplace_back(names[k]);
        std::vector<uint8_t> s(seqs[k], seqs[k] + lens[k]);
        for (auto& b : s) if (b >= 'a' && b <= 'z') b = (uint8_t)(b - 32);                         // target_seq.rs:111-115
        idx->seqs.push_back(std::move(s));
    }
    *out = idx.release();
    return STITCH_OK;
}

uint32_t stitch_index_n_contigs(const stitch_index* i) { return i ? (uint32_t)i->names.size() : 0; }
void stitch_index_destroy(stitch_index* i) { delete i; }

// blob: "STIX" u32 version u32 T, then per contig: u32 name_len, name, u32 seq_len, seq
int stitch_index_serialize(const stitch_index* idx, void* buf, size_t* len) {
    if (!idx || !len) return fail(STITCH_EINVAL, "null argument");
    size_t need = 12;
    for (size_t k = 0; k < idx->names.size(); ++k) need += 8 + idx->names[k].size() + idx->seqs[k].size();
    if (!buf) { *len = need; return STITCH_OK; }
    if (*len < need) { *len = need; return fail(STITCH_EINVAL, "buffer too small"); }
    uint8_t* p = (uint8_t*)buf;
    auto put32 = [&](uint32_t v) { memcpy(p, &v, 4); p += 4; };
    memcpy(p, "STIX", 4); p += 4; put32(1); put32((uint32_t)idx->names.size());
    for (size_t k = 0; k < idx->names.size(); ++k) {
        put32((uint32_t)idx->names[k].size()); memcpy(p, idx->names[k].data(), idx->names[k].size()); p += idx->names[k].size();
        put32((uint32_t)idx->seqs[k].size()); memcpy(p, idx->seqs[k].data(), idx->seqs[k].size()); p += idx->seqs[k].size();
    }
    *len = need;
    return STITCH_OK;
}

int stitch_index_deserialize(const void* buf, size_t len, stitch_index** out) {
    if (!buf || !out || len < 12 || memcmp(buf, "STIX", 4) != 0) return fail(STITCH_EINVAL, "not a stitch index blob");
    const uint8_t* p = (const uint8_t*)buf; const uint8_t* end = p + len; p += 4;
    auto get32 = [&](uint32_t& v) { if (p + 4 > end) return false; memcpy(&v, p, 4); p += 4; return true; };
    uint32_t ver, T;
    if (!get32(ver) || ver != 1 || !get32(T) || T == 0) return fail(STITCH_EINVAL, "bad index blob header");
    auto idx = std::make_unique<stitch_index>();
    for (uint32_t k = 0; k < T; ++k) {
        uint32_t nl, sl;
        if (!get32(nl) || p + nl > end) return fail(STITCH_EINVAL, "truncated index blob");
        idx->names.emplace_back((const char*)p, nl); p += nl;
        if (!get32(sl) || p + sl > end) return fail(STITCH_EINVAL, "truncated index blob");
        idx->seqs.emplace_back(p, p + sl); p += sl;
    }
    *out = idx.release();
    return STITCH_OK;
}

void stitch_ctx_destroy(stitch_ctx* c) {
    if (c && c->pin_q) { (void)hipHostFree(c->pin_q); c->pin_q = nullptr; }
    if (!c) return;
    (void)hipSetDevice(c->device);
    void* ptrs[] = {c->d_xseq, c->d_S0, c->d_Slen0, c->d_Sn0, c->d_SnSet0, c->d_Smove0, c->d_Imove0, c->d_lx0, c->d_base0, c->arena_raw, c->pre_buf};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->pin_h2d) (void)hipHostFree(c->pin_h2d);
    for (auto* b : c->pin_bands) if (b) (void)hipHostFree(b);
    if (c->pin_cnt) (void)hipHostFree(c->pin_cnt);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& w : c->evp) for (auto& e : w) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev2) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->evc) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->evu) if (e) (void)hipEventDestroy(e);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    for (auto& st : c->pstream) if (st) (void)hipStreamDestroy(st);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int stitch_ctx_create(int device_ordinal, const stitch_index* idx, const stitch_opts* o, stitch_ctx** out) {
    if (!idx || !o || !out) return fail(STITCH_EINVAL, "null argument");
    if (o->mode < 0 || o->mode > 3) return fail(STITCH_EINVAL, "Custom alignment mode not supported");      // mod.rs:129
    // the reference's constructor asserts (single_contig_aligner.rs:638-655, scoring.rs:36-73)
    if (o->gap_open > 0) return fail(STITCH_EINVAL, "gap_open can't be positive");
    if (o->gap_extend > 0) return fail(STITCH_EINVAL, "gap_extend can't be positive");
    if (o->jump_same > 0 || o->jump_opposite > 0 || o->jump_inter > 0) return fail(STITCH_EINVAL, "jump scores can't be positive");
    if (o->pre_align && (o->kmer_size < 1 || o->band_width < 0)) return fail(STITCH_EINVAL, "bad k-mer size or band width");
    const uint32_t T = (uint32_t)idx->names.size();
    const uint32_t C = T * (o->double_strand ? 2u : 1u);
    if (C > 256) return fail(STITCH_EINVAL, "more than 256 contig-strands: the reference's traceback cell holds contig indexes 0..255 (packed_length_cell.rs:112-114, 139)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(STITCH_EDEVICE, "no HIP device: this library has no CPU path");
    if (device_ordinal < 0 || device_ordinal >= ndev) return fail(STITCH_EDEVICE, "bad device ordinal");
    HIP_TRY(hipSetDevice(device_ordinal));

    std::unique_ptr<stitch_ctx, void (*)(stitch_ctx*)> c(new stitch_ctx(), stitch_ctx_destroy);
    c->device = device_ordinal; c->opts = *o; c->T = T; c->C = C;
    DpParams& P = c->P;
    P.match = o->match_score; P.mismatch = o->mismatch_score; P.gap_open = o->gap_open; P.gap_extend = o->gap_extend;
    P.jump_same = o->jump_same; P.jump_opp = o->jump_opposite; P.jump_inter = o->jump_inter; P.circular = o->circular ? 1 : 0;
    switch (o->mode) {                                        // Options::clipping, mod.rs:123-131
        case 0: P.xclip_prefix = P.xclip_suffix = P.yclip_prefix = P.yclip_suffix = 0; break;
        case 1: P.xclip_prefix = P.xclip_suffix = MIN_SCORE; P.yclip_prefix = P.yclip_suffix = 0; break;
        case 2: P.xclip_prefix = P.xclip_suffix = 0; P.yclip_prefix = P.yclip_suffix = MIN_SCORE; break;
        default: P.xclip_prefix = P.xclip_suffix = P.yclip_prefix = P.yclip_suffix = MIN_SCORE; break;
    }
    for (uint32_t t = 0; t < T; ++t) c->targets.push_back(TargetInfo{idx->names[t], (uint32_t)idx->seqs[t].size()});

    // aligners: forward strands 0..T-1, then reverse complements T..2T-1 (build_aligners, mod.rs:182-205)
    std::vector<uint8_t> xseq;
    uint32_t troff = 0;
    for (uint32_t a = 0; a < C; ++a) {
        const uint32_t t = a % T; const bool fwd = a < T;
        Aligner al; al.target = t; al.fwd = fwd; al.m = (uint32_t)idx->seqs[t].size(); al.troff = troff; al.seqoff = (uint32_t)xseq.size();
        al.opp = -1;
        std::vector<uint8_t> s = fwd ? idx->seqs[t] : revcomp(idx->seqs[t]);
        xseq.insert(xseq.end(), s.begin(), s.end());
        while (xseq.size() % TILE_ROWS) xseq.push_back(0);     // tile loads may run past m
        troff += (al.m + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
        c->max_m = std::max(c->max_m, al.m);
        c->al.push_back(al);
    }
    c->RtotT = troff;
    // opposite strands: same NAME and different strand, first match wins (multi_contig_aligner.rs:241-262)
    for (uint32_t a = 0; a < C; ++a) {
        if (c->al[a].opp >= 0) continue;
        for (uint32_t b = a + 1; b < C; ++b)
            if (c->targets[c->al[a].target].name == c->targets[c->al[b].target].name && c->al[a].fwd != c->al[b].fwd) {
                c->al[a].opp = (int32_t)b; c->al[b].opp = (int32_t)a;
            }
    }
    // column 0 (init_matrices) per aligner + get_jump_info over column 0
    std::vector<int32_t> S0(troff, MIN_SCORE), Sn0(troff, MIN_SCORE); std::vector<uint32_t> Slen0(troff, 0), lx0(C, 0);
    std::vector<uint8_t> SnSet0(troff, 0), Smove0(troff, 0), Imove0(troff, 0); std::vector<JumpBase> base0(C);
    std::vector<Col0Row> rows;
    for (uint32_t a = 0; a < C; ++a) {
        const Aligner& al = c->al[a];
        rows.resize(al.m);
        lx0[a] = col0_init(P, al.m, rows.data());
        JumpBase b; b.score = 0; b.from = 0; b.len = 0 + 1;                    // row 0: S[0][0] = 0, cell(0,0).S.len = 0
        for (uint32_t i = 1; i <= al.m; ++i) {
            const Col0Row& r = rows[i - 1]; const uint32_t x = al.troff + i - 1;
            S0[x] = r.S; Slen0[x] = r.Slen; Sn0[x] = r.Sn; SnSet0[x] = r.sn_set; Smove0[x] = r.Smove; Imove0[x] = r.Imove;
            if (b.score < r.S) { b.score = r.S; b.from = i; b.len = r.Slen + 1; }   // strict <: lowest row wins (:683)
        }
        base0[a] = b;
    }
    int rc;
    if ((rc = upload(&c->d_xseq, xseq)) || (rc = upload(&c->d_S0, S0)) || (rc = upload(&c->d_Slen0, Slen0)) || (rc = upload(&c->d_Sn0, Sn0)) ||
        (rc = upload(&c->d_SnSet0, SnSet0)) || (rc = upload(&c->d_Smove0, Smove0)) || (rc = upload(&c->d_Imove0, Imove0)) ||
        (rc = upload(&c->d_lx0, lx0)) || (rc = upload(&c->d_base0, base0))) return rc;
    { hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal)); c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; }
    HIP_TRY(hipStreamCreate(&c->stream));
    HIP_TRY(hipStreamCreate(&c->stream2));
    HIP_TRY(hipStreamCreate(&c->stream3));
    for (auto& st : c->pstream) HIP_TRY(hipStreamCreate(&st));
    for (auto& e : c->ev2) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : c->evc) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : c->evu) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : c->ev) HIP_TRY(hipEventCreate(&e));
    for (auto& w : c->evp) for (auto& e : w) HIP_TRY(hipEventCreate(&e));
    if (o->pre_align) {
        c->h_xseq = xseq;
        for (uint32_t a = 0; a < C; ++a) c->strands.push_back(Strand{c->al[a].seqoff, c->al[a].m});
        c->kidx = build_kmer_index(c->h_xseq.data(), c->strands, (uint32_t)o->kmer_size);
        c->pre_bytes = (size_t)4 << 30;                       // 256 reads x 100 target strands of cfg3 per launch
        if (const char* e = getenv("STITCH_PREALIGN_BYTES")) c->pre_bytes = std::max<size_t>((size_t)1 << 20, (size_t)strtoull(e, nullptr, 10));
        HIP_TRY(hipMalloc((void**)&c->pre_buf, c->pre_bytes));
    }
    if (const char* lim = getenv("STITCH_ARENA_BYTES")) c->mem_limit = (size_t)strtoull(lim, nullptr, 10);
    c->knobs = Knobs::from_env();
    c->regs_waves = (c->knobs.regs_waves == 2 || c->knobs.regs_waves == 4 || c->knobs.regs_waves == 8) ? (uint32_t)c->knobs.regs_waves : REGS_WAVES_DEFAULT;
    c->regs_wg_per_cu = fill_regs_workgroups_per_cu(c->regs_waves);
    c->regs32_wg_per_cu = fill_regs32_workgroups_per_cu();
    *out = c.release();
    return STITCH_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Running a list of DP jobs on the device
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct JobLayout {
    uint32_t n, nact, Rj, slots, ops_cap;
    size_t off_S, off_Slen, off_D, off_Dlen, off_Sn, off_SnLen, off_Ly, off_Ival, off_Ilen, off_SidxF, off_SfromF, off_SmoveF, off_ImoveF,
        off_st16, off_xchg, off_tb, off_Lx, off_jti, off_jtf, off_Sm, off_Lm, off_visit, off_Wcol, off_y, off_act, off_opp, off_cd, off_hdr, off_ops, bytes,
        stride;              // distance to the next job's block in a launch (bytes rounded up to the launch's block alignment)
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

JobLayout layout_job(const stitch_ctx& c, const Job& jb) {
    JobLayout L{};
    L.n = (uint32_t)jb.y.size(); L.nact = (uint32_t)jb.act.size();
    uint32_t R = 0; for (uint32_t a : jb.act) R += (c.al[a].m + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
    L.Rj = R;
    L.slots = jb.mode == 1 ? L.nact : 1;
    L.ops_cap = 2 * L.n + 2 * c.max_m + 64;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    // the arrays the fill kernel streams every column (row state, y-suffix records, traceback) start at multiples of 2 MiB in
    // jobs that are large enough not to notice (measured on cfg2: 3-4 % over 256-byte packing; 4 KiB: nothing, 64 MiB: the same)
    const size_t big_align = c.knobs.array_align ? c.knobs.array_align : (8ull * R >= ((size_t)1 << 20) ? (size_t)2 << 20 : (size_t)256);
    auto take_big = [&](size_t bytes) { if (big_align > 256) o = align_up(o, big_align); return take(bytes); };
    L.off_S = take(4ull * R); L.off_Slen = take(4ull * R); L.off_D = take_big(4ull * R); L.off_Dlen = take(4ull * R);
    if (L.off_Dlen != L.off_D + 4ull * R) abort();      // fill_local16.hip keeps its 8-byte y-suffix records in D..Dlen
    L.off_Sn = take(4ull * R); L.off_SnLen = take(4ull * R); L.off_Ly = take(4ull * R);
    L.off_Ival = take(4ull * R); L.off_Ilen = take(4ull * R); L.off_SidxF = take(4ull * R); L.off_SfromF = take(4ull * R);
    L.off_SmoveF = take(R); L.off_ImoveF = take(R);
    L.off_st16 = take_big(8ull * R);
    L.off_xchg = take(32ull * c.C + 4096);      // 2 parities x C contigs x two 8-byte granules, then the error word
    L.off_tb = take_big((size_t)L.n * R);
    L.off_Lx = take(4ull * c.C * (L.n + 1)); L.off_jti = take(4ull * c.C * (L.n + 1)); L.off_jtf = take(4ull * c.C * (L.n + 1));
    L.off_Sm = take(4ull * c.C); L.off_Lm = take(4ull * c.C);
    L.off_Wcol = take(jb.mode != 0 ? 4ull * c.C * (L.n + 1) : 0);      // the column's common word of every contig (fill_regs.hip: y-suffix records as one bit per cell)
    L.off_visit = take(jb.mode == 1 ? sizeof(VisitRec) * ((size_t)L.n + 2) : 0);      // traceback_all: the reference walk's column records (walk_core.h)
    L.off_y = take(L.n); L.off_act = take(4ull * L.nact); L.off_opp = take(4ull * c.C); L.off_cd = take(sizeof(ContigDesc) * (size_t)c.C);
    L.off_hdr = take(sizeof(ChainHdr) * (size_t)L.slots); L.off_ops = take(sizeof(OpRec) * (size_t)L.slots * L.ops_cap);
    L.bytes = o;
    return L;
}

// The Local-mode kernel keeps scores and alignment lengths in 16 bits (fill_local16.hip).
// fill_local16.hip deals a read's active contigs round-robin to its G workgroups, and a workgroup's slot table holds
// fill_local16_max_slots() 256-row tiles: the tiles of the fullest workgroup at G workgroups per read ...
uint32_t local16_wg_tiles(const stitch_ctx& c, const Job& jb, uint32_t G) {
    std::vector<uint32_t> per(G, 0);
    for (size_t k = 0; k < jb.act.size(); ++k) per[k % G] += (c.al[jb.act[k]].m + 255) / 256;
    return *std::max_element(per.begin(), per.end());
}
// ... and the fewest workgroups per read whose fullest one still fits its table (0: none up to 64 does)
uint32_t local16_min_g(const stitch_ctx& c, const Job& jb) {
    for (uint32_t G = 1; G <= 64 && G <= std::max<size_t>(1, jb.act.size()); ++G) if (local16_wg_tiles(c, jb, G) <= fill_local16_max_slots()) return G;
    return 0;
}

bool local16_ok(const stitch_ctx& c, const Job& jb) {
    const stitch_opts& o = c.opts;
    if (c.knobs.force_generic) return false;
    const long long n = (long long)jb.y.size();
    const int32_t lo = std::min({o.mismatch_score, o.gap_open + o.gap_extend, o.jump_same, o.jump_opposite, o.jump_inter, o.match_score});
    if (!(o.mode == 0 && o.gap_open + o.gap_extend < 0 && (long long)std::max(o.match_score, 0) * n <= 32767 &&
          n + (long long)c.max_m + 2 <= 65535 && lo >= -16000 && o.match_score <= 16000)) return false;
    // Alignment lengths are 16-bit too.  A chain's length = its read bases (<= n) + the contig bases it inserts; every inserted base
    // costs |gap_extend| out of a score budget of match * n, so with gap_extend < 0 the length stays below n + match * n / |ge|;
    // a free gap extension has no such bound.
    if (o.gap_extend >= 0 || n + (long long)std::max(o.match_score, 0) * n / (long long)(-o.gap_extend) + 2 > 65535) return false;
    return local16_min_g(c, jb) != 0;
}

// The register-resident Local-mode kernel (fill_regs.hip): one wave per active contig, REGS_WAVES waves per workgroup.  Returns the
// workgroups a read needs (0: not applicable — a contig longer than a wave holds, a gap-extension penalty
// too large for lane-tagged scan keys, more workgroups than the device holds at once, or a read too small to be worth a team).
constexpr uint32_t REGS_WAVES = 4;      // fill_regs32.hip: four waves per workgroup, one workgroup per CU
uint32_t regs_plan(const stitch_ctx& c, const Job& jb) {
    if (c.knobs.no_regs || c.regs_wg_per_cu <= 0 || !local16_ok(c, jb)) return 0;
    if (c.opts.gap_extend < -1024 || c.opts.gap_open + c.opts.gap_extend < -8000) return 0;      // (16-bit insertion-chain words, fill_regs.hip)
    if (c.opts.gap_open > 0) return 0;                 // (an opener from an insertion-derived cell must not beat the extension: fill_regs.hip)
    uint64_t rows = 0;
    for (uint32_t a : jb.act) { if (c.al[a].m > fill_regs_rows_per_wave()) return 0; rows += c.al[a].m; }
    const uint64_t min_rows = c.knobs.regs_min_rows >= 0 ? (uint64_t)c.knobs.regs_min_rows : 2048u;
    if (rows < min_rows) return 0;
    const uint32_t G = ((uint32_t)jb.act.size() + c.regs_waves - 1) / c.regs_waves;
    if (G > (uint32_t)c.n_cus * (uint32_t)c.regs_wg_per_cu) return 0;
    return G;
}

// The 32-bit register-resident kernel (fill_regs32.hip): every clipping mode, reads beyond the 16-bit kernels' range.  One wave per
// active contig, four waves per workgroup, one workgroup per CU.  Returns the workgroups a read needs (0: not applicable).
// Its arithmetic rests on every score the recurrence can produce staying a "real" number: |score| < 2^27, so that a candidate
// that carries a MIN_SCORE clip penalty never wins and no cell reaches MIN_SCORE (fill_regs32.hip header).  Alignment lengths are
// 16-bit: a chain's length = its read bases (<= n) + the contig bases it inserts — forced ones at the ends of an x-global
// alignment (<= 2 max_m) and voluntary ones, each of which costs |gap_extend| out of a score budget of match * n.
uint32_t regs32_plan(const stitch_ctx& c, const Job& jb) {
    const stitch_opts& o = c.opts;
    if (c.knobs.no_regs32 || c.knobs.force_generic || c.regs32_wg_per_cu <= 0) return 0;
    const long long n = (long long)jb.y.size(), mm = (long long)c.max_m;
    const long long big = std::max<long long>({std::llabs((long long)o.match_score), std::llabs((long long)o.mismatch_score), std::llabs((long long)o.gap_open) + std::llabs((long long)o.gap_extend),
                                               std::llabs((long long)o.jump_same), std::llabs((long long)o.jump_opposite), std::llabs((long long)o.jump_inter)});
    if (big > 4000 || big * (n + mm + 4) >= (1ll << 27)) return 0;      // (4000: the records' 16-bit relative scores, KEY_BIAS in fill_regs32.hip)
    if (o.gap_extend >= 0 || o.gap_open > 0) return 0;
    const long long forced = o.mode == 0 ? 0 : 2 * mm;
    if (n + forced + (long long)std::max(o.match_score, 0) * n / (long long)(-o.gap_extend) + 2 > 65535) return 0;
    if (n + mm + 2 > 65535) return 0;
    uint64_t rows = 0;
    for (uint32_t a : jb.act) { if (c.al[a].m > fill_regs32_rows_per_wave()) return 0; rows += c.al[a].m; }
    const uint64_t min_rows = c.knobs.regs_min_rows >= 0 ? (uint64_t)c.knobs.regs_min_rows : 2048u;
    if (rows < min_rows) return 0;
    const uint32_t G = ((uint32_t)jb.act.size() + REGS_WAVES - 1) / REGS_WAVES;
    if (G > (uint32_t)c.n_cus * (uint32_t)c.regs32_wg_per_cu) return 0;
    return G;
}

constexpr int MAX_WAVES_GENERIC = 8;               // fill_kernel.hip: __launch_bounds__(512)
#ifndef STITCH_LB
#define STITCH_LB 768
#endif
constexpr int MAX_WAVES_LOCAL = STITCH_LB / 64;       // fill_local16.hip: __launch_bounds__(STITCH_LB), 12 waves

int pick_waves(const stitch_ctx& c, uint32_t nact, int maxw) {          // fewest rounds of contigs per column, then fewest waves
    if (c.knobs.max_waves) maxw = std::max(1, std::min(maxw, c.knobs.max_waves));
    int best_w = 1, best_rounds = (int)nact;
    for (int w = 1; w <= maxw; ++w) { int r = ((int)nact + w - 1) / w; if (r < best_rounds) { best_rounds = r; best_w = w; } }
    return best_w;
}

// Waiting for the second stream while persistent teams are resident.  The fix-up / walk kernels and the runtime's copy kernels of that
// stream need one of the few wave slots the teams leave free, and the dispatcher deals workgroups to the shader engines in a fixed
// rotation: now and then (seen about once in a few thousand launches) a workgroup's turn falls on an engine the teams fill completely and
// it waits there for as long as they stay.  They stay until the queue is empty — so a wait that lasts beyond a bound calls the run off
// (the teams leave after the read they are on, the stuck launch then runs) and what is left goes launch by launch.
static int bounded_sync(stitch_ctx& c, hipStream_t s) {
    if (!c.stream_abort_word) { HIP_TRY(hipStreamSynchronize(s)); return STITCH_OK; }
    // The stuck workgroup waits for a slot in ONE shader engine; a team's 50 waves sit in six or seven of the chip's 32.  After a quarter of
    // a second ONE team is asked to leave (while the host waits here it recycles no arena blocks, so after the few spare ones are used up
    // every team is waiting for a block and looks at the host's words every 20 us: the first to look takes the ticket and returns) — that
    // costs a fortieth of the call's rate and frees the right engine about one time in five (seen: the walk ran 6 ms after the ticket).
    // More tickets cost more than they are worth (eight teams gone, and the launch still stuck, has been seen too): 30 ms after the ticket —
    // a waiting team takes it within microseconds, the freed slots are taken within milliseconds or not at all —
    // the run is called off — every team leaves, waiting ones at once — and what is left goes launch by launch, 5 % below the teams' rate.
    auto t0 = std::chrono::steady_clock::now();
    bool ticket_here = false;                    // (this wait has asked a team to leave: the clock runs from the ticket)
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return STITCH_OK;
        if (e != hipErrorNotReady) return fail(STITCH_EDEVICE, std::string("hipStreamQuery: ") + hipGetErrorString(e));
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const uint32_t next_tickets = std::min<uint32_t>(c.stream_may_retire, 1u);
        if (!c.stream_stalled && (c.knobs.test_stream_stall || (next_tickets <= c.stream_retired && waited > (ticket_here ? 0.03 : c.stream_bound_s)))) {
            c.stream_stalled = true;
            *(volatile uint32_t*)c.stream_abort_word = 0xFFFFFFFFu;
            std::atomic_thread_fence(std::memory_order_seq_cst);
        }
        else if (!c.stream_stalled && next_tickets > c.stream_retired && (waited > (c.stream_retired == 0 ? c.stream_bound_s : 0.1) || (c.knobs.test_stream_retire && c.stream_retired == 0))) {
            c.tm.teams_retired += next_tickets - c.stream_retired; c.stream_retired = next_tickets; ticket_here = true;
            *(volatile uint32_t*)c.stream_abort_word = c.stream_retired;
            std::atomic_thread_fence(std::memory_order_seq_cst);
            if (c.knobs.trace) fprintf(stderr, "[trace] a launch beside the teams has waited %.0f ms: %u team(s) in all asked to leave\n", waited * 1e3, c.stream_retired);
            t0 = std::chrono::steady_clock::now();
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// Chains of the jobs [k0, k0 + nj) after their fix-up + walk: headers first, then the operation lists, both batched through a pinned
// staging buffer (one synchronous pageable copy per chain costs ~0.15 ms each; --suboptimal yields hundreds of chains per read).
// blocks[q] = the arena block of job k0 + q, d_views = the launch's job table (entry q = job k0 + q).  A chain whose operations did not
// fit its buffer (only possible with free gaps / jumps) is walked again into an exact-size buffer; that path allocates and frees device
// memory, which waits for every kernel on the device: with persistent teams running (allow_rewalk == false) the caller is told
// instead (return value 1) and repeats the read on the classic path.
int download_chains(stitch_ctx& c, hipStream_t sB, std::vector<Job>& jobs, const std::vector<JobLayout>& lay, size_t k0, uint32_t nj,
                    const std::vector<uint8_t*>& blocks, JobView* d_views, bool allow_rewalk) {
        auto t_d2h0 = std::chrono::steady_clock::now();
        if (!c.pin) { HIP_TRY(hipHostMalloc((void**)&c.pin, PIN_BYTES, hipHostMallocDefault)); }
        struct Pending { void* dst; const uint8_t* src; size_t bytes; };
        std::vector<Pending> pend;
        auto flush = [&]() -> int {
            size_t i = 0;
            while (i < pend.size()) {
                if (pend[i].bytes > PIN_BYTES) {
                    if (!allow_rewalk) { HIP_TRY(hipMemcpyAsync(pend[i].dst, pend[i].src, pend[i].bytes, hipMemcpyDeviceToHost, sB)); if (int e2 = bounded_sync(c, sB)) return e2; }
                    else HIP_TRY(hipMemcpy(pend[i].dst, pend[i].src, pend[i].bytes, hipMemcpyDeviceToHost));
                    ++i; continue;
                }
                size_t used = 0, j = i;
                while (j < pend.size() && used + pend[j].bytes <= PIN_BYTES) {
                    HIP_TRY(hipMemcpyAsync(c.pin + used, pend[j].src, pend[j].bytes, hipMemcpyDeviceToHost, sB));
                    used += align_up(pend[j].bytes, 64); ++j;
                }
                if (int e2 = bounded_sync(c, sB)) return e2;
                used = 0;
                for (size_t k = i; k < j; ++k) { memcpy(pend[k].dst, c.pin + used, pend[k].bytes); used += align_up(pend[k].bytes, 64); }
                i = j;
            }
            pend.clear();
            return STITCH_OK;
        };
        struct Join { uint32_t q, s, ref, n; };
        std::vector<Join> joins;
        std::vector<std::vector<ChainHdr>> hdrs(nj);
        for (uint32_t q = 0; q < nj; ++q) {
            const JobLayout& L = lay[k0 + q];
            hdrs[q].resize(L.slots);
            pend.push_back({hdrs[q].data(), blocks[q] + L.off_hdr, sizeof(ChainHdr) * (size_t)L.slots});
        }
        if (int e = flush()) return e;
        for (uint32_t q = 0; q < nj; ++q) {
            Job& jb = jobs[k0 + q]; const JobLayout& L = lay[k0 + q]; uint8_t* B = blocks[q];
            jb.chains.assign(L.slots, HAln()); jb.status.assign(L.slots, 0);
            for (uint32_t s = 0; s < L.slots; ++s) {
                ChainHdr H = hdrs[q][s];
                const uint8_t* ops_src = B + L.off_ops + sizeof(OpRec) * (size_t)s * L.ops_cap;
                uint8_t* big = nullptr;
                if (H.status == 2) {
                    if (!allow_rewalk) return 1;
                    // more operations than the default buffer holds (only with free gaps/jumps): walk this chain again
                    // into a buffer of the size the first walk counted; the fix-ups must not run twice
                    struct Retry { ChainHdr h; WalkArgs a; };
                    const size_t ops_bytes = sizeof(OpRec) * (size_t)H.n_ops;
                    HIP_TRY(hipMalloc((void**)&big, align_up(sizeof(Retry), 256) + ops_bytes));
                    Retry rt{}; rt.a.hdr = (ChainHdr*)big; rt.a.ops = (OpRec*)(big + align_up(sizeof(Retry), 256)); rt.a.ops_cap = H.n_ops;
                    rt.a.mode = 2; rt.a.from = H.end_contig_idx; rt.a.skip_fixup = 1;
                    HIP_TRY(hipMemcpy(big, &rt, sizeof(Retry), hipMemcpyHostToDevice));
                    launch_fixup_walk(d_views + q, (const WalkArgs*)(big + offsetof(Retry, a)), 1, 0, sB);
                    HIP_TRY(hipStreamSynchronize(sB));
                    HIP_TRY(hipMemcpy(&H, big, sizeof(ChainHdr), hipMemcpyDeviceToHost));
                    ops_src = big + align_up(sizeof(Retry), 256);
                }
                jb.status[s] = H.status;
                if (H.status == 4) { if (big) (void)hipFree(big); return fail(STITCH_EINVAL, "end-of-read jump into a shorter contig: the reference indexes its traceback matrix out of range here (traceback/mod.rs:329-338); result undefined"); }
                if (H.status >= 2) { if (big) (void)hipFree(big); return fail(STITCH_EINTERNAL, "traceback failed on the device (status " + std::to_string(H.status) + ")"); }
                if (H.status == 1) { if (big) (void)hipFree(big); continue; }
                if ((size_t)H.n_ops > (big ? (size_t)H.n_ops : (size_t)L.ops_cap)) return fail(STITCH_EINTERNAL, "chain header reports more operations than its buffer holds");
                HAln& a = jb.chains[s];
                a.score = H.score; a.xstart = H.xstart; a.xend = H.xend; a.ystart = H.ystart; a.yend = H.yend; a.xlen = H.xlen; a.ylen = H.ylen;
                a.start_contig_idx = H.start_contig_idx; a.end_contig_idx = H.end_contig_idx; a.length = H.length;
                a.ops.resize(H.n_ops);
                if (H.join_ops) { if (big || H.join_slot >= L.slots || H.join_slot == s) return fail(STITCH_EINTERNAL, "bad chain join"); joins.push_back({q, s, H.join_slot, H.join_ops}); }
                static_assert(sizeof(OpRec) == sizeof(stitch_op), "op layout");
                if (H.n_ops) {
                    if (big) { HIP_TRY(hipMemcpy(a.ops.data(), ops_src, sizeof(OpRec) * (size_t)H.n_ops, hipMemcpyDeviceToHost)); }
                    else pend.push_back({a.ops.data(), ops_src, sizeof(OpRec) * (size_t)H.n_ops});
                }
                if (big) (void)hipFree(big);
            }
        }
        if (int e = flush()) return e;
        // chains that met the reference chain of their read: its first join_ops operations, then their own (walk_core.h, VisitRec)
        for (const Join& jn : joins) {
            Job& jb = jobs[k0 + jn.q];
            const HAln& ref = jb.chains[jn.ref];
            if (jn.ref >= jb.chains.size() || jb.status[jn.ref] != 0 || jn.n > ref.ops.size()) return fail(STITCH_EINTERNAL, "a chain joins a reference chain that does not hold that many operations");
            HAln& a = jb.chains[jn.s];
            std::vector<stitch_op> all; all.reserve((size_t)jn.n + a.ops.size());
            all.insert(all.end(), ref.ops.begin(), ref.ops.begin() + jn.n);
            all.insert(all.end(), a.ops.begin(), a.ops.end());
            a.ops.swap(all);
        }
        c.tm.d2h_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_d2h0).count();
        return STITCH_OK;
}

static int run_jobs_in_order(stitch_ctx& c, std::vector<Job>& jobs);

// Jobs of one launch run side by side until the longest is done, so launches are formed from jobs of similar size: the list
// is processed in descending order of work (read length x active rows) and handed back in the caller's order.
int run_jobs(stitch_ctx& c, std::vector<Job>& jobs) {
    if (jobs.size() < 2) return run_jobs_in_order(c, jobs);
    std::vector<size_t> perm(jobs.size());
    for (size_t k = 0; k < perm.size(); ++k) perm[k] = k;
    auto work = [&](const Job& jb) { unsigned long long rows = 0; for (uint32_t a : jb.act) rows += c.al[a].m; return rows * (unsigned long long)jb.y.size(); };
    std::vector<unsigned long long> w(jobs.size());
    for (size_t k = 0; k < jobs.size(); ++k) w[k] = work(jobs[k]);
    std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) { return w[a] > w[b]; });
    std::vector<Job> sorted; sorted.reserve(jobs.size());
    for (size_t k : perm) sorted.push_back(std::move(jobs[k]));
    const int rc = run_jobs_in_order(c, sorted);
    for (size_t k = 0; k < perm.size(); ++k) jobs[perm[k]] = std::move(sorted[k]);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent teams (round 4).  A classic launch of fill_regs.hip holds as many reads as fill the chip's wave slots and ends with its
// SLOWEST read (reads differ by 20-30 %: DESIGN.md 4); with two launches in flight the next launch takes the slots finished reads
// free, but launch k + 2 still waits for all of launch k.  Here ONE launch per call keeps T teams (T x W waves = the chip's wave
// slots, W = active contigs per read) resident for the whole batch: a team that ends a read pulls the next one off a queue in device
// memory.  The arena is cut into B blocks, job i uses block i mod B; the host watches the jobs' completion words (pinned host memory),
// runs fix-up + walk + downloads for finished jobs on the second stream and then lets job i + B start (h_ready).  Nothing the
// running kernel READS is written by the host after the launch except those pinned words: every job's read, contig tables and
// exchange granules are uploaded / cleared up front in a region of their own, and the block arrays are all written by the fill before
// it reads them.  Applies when every job takes the register-resident kernel with the same number of active contigs (cfg2, cfg4, cfg5;
// not the filtered reads of cfg3, whose teams differ in size) and there are more jobs than teams.
// Returns STITCH_OK with *handled = true when the jobs were run here; *handled = false (and STITCH_OK) when the classic path should.
static int run_jobs_streaming(stitch_ctx& c, std::vector<Job>& jobs, bool* handled) {
    *handled = false;
    const size_t N = jobs.size();
    const Knobs& kn = c.knobs;
    if (kn.no_stream || kn.debug || kn.profile_dump || kn.fill_only || !kn.dump_dir.empty() || kn.no_pipeline || kn.fail_first_attempt || kn.force_regs32 || kn.regs_map == 1) return STITCH_OK;
    if (N < 2 || c.regs_wg_per_cu <= 0) return STITCH_OK;
    const uint32_t W = (uint32_t)jobs[0].act.size();
    for (const Job& jb : jobs) if (jb.act.size() != W || jb.y.size() < 2 || regs_plan(c, jb) == 0) return STITCH_OK;
    const size_t slots = (size_t)c.n_cus * (size_t)c.regs_wg_per_cu * c.regs_waves;
    size_t T = slots / W;
    if (kn.stream_teams > 0) T = std::min<size_t>(T, (size_t)kn.stream_teams);
    if (T < 1 || N <= T) return STITCH_OK;            // (one classic launch holds them all)
    HIP_TRY(hipSetDevice(c.device));
    std::vector<JobLayout> lay(N);
    size_t max_job = 0, in_total = 0;
    std::vector<size_t> in_at(N + 1, 0);
    for (size_t k = 0; k < N; ++k) { lay[k] = layout_job(c, jobs[k]); max_job = std::max(max_job, lay[k].bytes); in_at[k + 1] = in_at[k] + align_up(lay[k].off_hdr - lay[k].off_y, 256); }
    in_total = in_at[N];
    const size_t xbytes = align_up(32ull * c.C + 4096, 256);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    size_t budget = (size_t)(std::min(free_b + c.arena_bytes, total_b) * 0.90);
    if (c.mem_limit) budget = std::min(budget, c.mem_limit);
    // behind the blocks: job table, walk arguments, the jobs' inputs, their exchange granules + error words, wave counters, mailboxes, queue head, wave map, the control block
    const size_t fixed = align_up(sizeof(JobView) * N, 256) + align_up(sizeof(WalkArgs) * N, 256) + in_total + xbytes * N + align_up(4 * N, 256) + align_up(8 * T, 256) + 256 +
                         align_up(sizeof(uint2) * (T * W + 1024) + 64, 256) + 256 + ((size_t)1 << 20);
    // blocks at multiples of a large power of two, like the classic windows (the jobs' streams want to be congruent)
    size_t a_pick = 0, stride = 0, B = 0;
    for (size_t a = (size_t)1 << 30; a >= 256; a >>= 1) {
        if (a > max_job && a > 256) continue;
        const size_t st = align_up(max_job, a);
        if (a > 256 && st % (4 * a) == 0) continue;
        if (budget < fixed + a + st) { if (a == 256) break; continue; }
        size_t b = std::min(N, (budget - fixed - a) / st);
        if (kn.stream_blocks > 0) b = std::min<size_t>(b, (size_t)kn.stream_blocks);
        if (b >= std::min(N, T + 2) || a == 256) { a_pick = a; stride = st; B = b; break; }
    }
    if (B < 2) return STITCH_OK;
    if (B < T) T = B;                                  // (fewer blocks than the chip holds teams: as many teams as blocks)
    if (N <= T) return STITCH_OK;
    const size_t need = B * stride + fixed + a_pick;
    auto arena_fits = [&]() { return c.arena && align_up((size_t)(uintptr_t)c.arena, a_pick) + B * stride + fixed <= (size_t)(uintptr_t)c.arena + c.arena_bytes; };
    if (!arena_fits()) {
        if (c.arena) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c.arena_raw)); c.arena_raw = nullptr; c.arena = nullptr; c.arena_bytes = 0; }
        if (hipMalloc((void**)&c.arena_raw, need + a_pick) != hipSuccess) { (void)hipGetLastError(); c.arena_raw = nullptr; return STITCH_OK; }      // (the classic path sizes its own arena)
        c.arena = (uint8_t*)align_up((size_t)(uintptr_t)c.arena_raw, std::max<size_t>(a_pick, 256)); c.arena_bytes = need + a_pick - (size_t)(c.arena - c.arena_raw);
    }
    uint8_t* const blocks0 = (uint8_t*)align_up((size_t)(uintptr_t)c.arena, a_pick);
    uint8_t* tail = blocks0 + B * stride;
    auto carve = [&](size_t bytes) { uint8_t* at = tail; tail += align_up(bytes, 256); return at; };
    JobView* const d_views = (JobView*)carve(sizeof(JobView) * N);
    WalkArgs* const d_wargs = (WalkArgs*)carve(sizeof(WalkArgs) * N);
    uint8_t* const d_in = carve(in_total);
    uint8_t* const d_zero = tail;                                   // one clear: granules + error words, wave counters, mailboxes, queue head
    uint8_t* const d_x = carve(xbytes * N);
    uint32_t* const d_cnt = (uint32_t*)carve(4 * N);
    unsigned long long* const d_mbox = (unsigned long long*)carve(8 * T);
    uint32_t* const d_next = (uint32_t*)carve(256);
    const size_t zero_bytes = (size_t)(tail - d_zero);
    uint2* const d_wave_map = (uint2*)carve(sizeof(uint2) * (T * W + 1024) + 64);
    StreamCtl* const d_ctl = (StreamCtl*)carve(256);
    if ((size_t)(tail - c.arena) > c.arena_bytes) return fail(STITCH_EINTERNAL, "arena overflow (persistent teams)");
    // pinned words: [0] ready, [16] abort, [32] err (a cache line each), [64 ..] done[N]
    if (c.pin_q_words < 64 + N) {
        if (c.pin_q) { (void)hipHostFree(c.pin_q); c.pin_q = nullptr; c.pin_q_words = 0; }
        const size_t words = std::max<size_t>(4096, (64 + N) * 2);
        HIP_TRY(hipHostMalloc((void**)&c.pin_q, 4 * words, hipHostMallocMapped | hipHostMallocCoherent));
        c.pin_q_words = words;
    }
    volatile uint32_t* const hq = c.pin_q;
    for (size_t k = 0; k < 64 + N; ++k) hq[k] = 0u;
    if (!c.pin) { HIP_TRY(hipHostMalloc((void**)&c.pin, PIN_BYTES, hipHostMallocDefault)); }
    if (in_total > c.pin_h2d_bytes) {
        if (c.pin_h2d) { (void)hipHostFree(c.pin_h2d); c.pin_h2d = nullptr; c.pin_h2d_bytes = 0; }
        const size_t want_b = std::max<size_t>(in_total * 3 / 2, (size_t)1 << 20);
        HIP_TRY(hipHostMalloc((void**)&c.pin_h2d, want_b, hipHostMallocDefault));
        c.pin_h2d_bytes = want_b;
    }
    hipStream_t const sA = c.stream, sB = c.stream2;
    FillShared sh{c.d_S0, c.d_Slen0, c.d_Sn0, c.d_SnSet0, c.d_Smove0, c.d_lx0, c.d_base0};
    c.tm_fast = true;
    const auto t_h2d0 = std::chrono::steady_clock::now();
    std::vector<JobView> views(N); std::vector<WalkArgs> wargs(N);
    auto block_of = [&](size_t k) { return blocks0 + (k % B) * stride; };
    for (size_t k = 0; k < N; ++k) {
        const Job& jb = jobs[k]; const JobLayout& L = lay[k];
        uint8_t* const Bk = block_of(k); uint8_t* const In = d_in + in_at[k]; uint8_t* const X = d_x + xbytes * k;
        std::vector<ContigDesc> cd(c.C); std::vector<int32_t> opp(c.C, -1); std::vector<uint8_t> isact(c.C, 0);
        for (uint32_t a : jb.act) isact[a] = 1;
        uint32_t roff = 0;
        for (uint32_t a = 0; a < c.C; ++a) {
            ContigDesc d{}; d.m = c.al[a].m; d.troff = c.al[a].troff; d.seqoff = c.al[a].seqoff; d.target = c.al[a].target; d.opp = c.al[a].opp;
            d.roff = 0;
            { const uint32_t ngr = (d.m + 3) / 4, gq = ngr / 64; d.inv_big = tb_div_magic(4 * (gq + 1)); d.inv_small = tb_div_magic(4 * gq); }
            if (isact[a]) { d.roff = roff; roff += (d.m + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS; }
            cd[a] = d;
            if (isact[a] && c.al[a].opp >= 0 && isact[c.al[a].opp]) opp[a] = c.al[a].opp;      // (only within the current subset: multi_contig_aligner.rs:241-262)
        }
        uint8_t* stg = c.pin_h2d + in_at[k];
        memcpy(stg, jb.y.data(), L.n);
        memcpy(stg + (L.off_act - L.off_y), jb.act.data(), 4ull * L.nact);
        memcpy(stg + (L.off_opp - L.off_y), opp.data(), 4ull * c.C);
        memcpy(stg + (L.off_cd - L.off_y), cd.data(), sizeof(ContigDesc) * (size_t)c.C);
        JobView& V = views[k];
        V.tb_keyfmt = 2u; V.yrec_global = jb.mode == 0 ? 1u : 0u;
        V.P = c.P; V.n = L.n; V.C = c.C; V.nact = L.nact; V.Rtot = L.Rj;
        V.y = In; V.act = (const uint32_t*)(In + (L.off_act - L.off_y)); V.opp_act = (const int32_t*)(In + (L.off_opp - L.off_y)); V.cd = (const ContigDesc*)(In + (L.off_cd - L.off_y));
        V.xseq = c.d_xseq;
        V.S = (int32_t*)(Bk + L.off_S); V.Slen = (uint32_t*)(Bk + L.off_Slen); V.D = (int32_t*)(Bk + L.off_D); V.Dlen = (uint32_t*)(Bk + L.off_Dlen);
        V.st16 = (uint32_t*)(Bk + L.off_st16);
        V.xchg = (unsigned long long*)X; V.err = (uint32_t*)(X + 32ull * c.C);
        V.Sn = (int32_t*)(Bk + L.off_Sn); V.SnLen = (uint32_t*)(Bk + L.off_SnLen); V.Ly = (uint32_t*)(Bk + L.off_Ly);
        V.tb = Bk + L.off_tb; V.Lx = (uint32_t*)(Bk + L.off_Lx); V.jt_idx = (uint32_t*)(Bk + L.off_jti); V.jt_from = (uint32_t*)(Bk + L.off_jtf);
        V.Ival = (int32_t*)(Bk + L.off_Ival); V.Ilen = (uint32_t*)(Bk + L.off_Ilen); V.SmoveF = Bk + L.off_SmoveF;
        V.SidxF = (uint32_t*)(Bk + L.off_SidxF); V.SfromF = (uint32_t*)(Bk + L.off_SfromF); V.ImoveF = Bk + L.off_ImoveF;
        V.Smove0 = c.d_Smove0; V.Imove0 = c.d_Imove0; V.Slen0 = c.d_Slen0;
        V.Sm = (int32_t*)(Bk + L.off_Sm); V.Lm = (uint32_t*)(Bk + L.off_Lm);
        V.visit = (jb.mode == 1 && !c.knobs.no_join) ? (VisitRec*)(Bk + L.off_visit) : nullptr;
        V.Wcol = jb.mode != 0 ? (uint32_t*)(Bk + L.off_Wcol) : nullptr;
        WalkArgs& A = wargs[k]; A.hdr = (ChainHdr*)(Bk + L.off_hdr); A.ops = (OpRec*)(Bk + L.off_ops); A.ops_cap = L.ops_cap; A.mode = jb.mode; A.from = jb.from; A.skip_fixup = 0;
        c.tm.cells += (uint64_t)L.n * [&] { uint64_t s2 = 0; for (uint32_t a : jb.act) s2 += c.al[a].m; return s2; }();
    }
    std::vector<uint2> wave_map; wave_map.reserve(T * W);
    const size_t per_xcd = (T + 7) / 8;
    if (kn.regs_map == 2 && per_xcd * W <= (size_t)(c.n_cus / 8) * c.regs_wg_per_cu * c.regs_waves) {
        // (experiment) a team's waves in workgroups of ONE XCD: workgroup b goes to XCD b mod 8 (observed dispatch order, no guarantee), so
        // XCD x's workgroups x, x + 8, ... hold the teams x, x + 8, ... one after the other; bit 30 asks for plain granule stores
        const size_t wg_per_xcd = (per_xcd * W + c.regs_waves - 1) / c.regs_waves;
        wave_map.assign(8 * wg_per_xcd * c.regs_waves, make_uint2(0u, 0xFFFFFFFFu));
        for (size_t x = 0; x < 8; ++x) for (size_t sl = 0; sl < per_xcd * W; ++sl) {
            const size_t t = x + 8 * (sl / W);
            if (t >= T) continue;
            const size_t wg = 8 * (sl / c.regs_waves) + x;
            wave_map[wg * c.regs_waves + sl % c.regs_waves] = make_uint2((uint32_t)t, (uint32_t)(sl % W) | (getenv("STITCH_EXP_PLAIN_GRANULES") ? 0x40000000u : 0u));
        }
    }
    else {
        // (where a team's waves fill whole workgroups, a workgroup's first wave polls the granules for all of them: bit 29, fill_regs.hip)
        const uint32_t wgp = (W % c.regs_waves == 0 && !kn.no_wg_poll) ? 0x20000000u : 0u;
        for (uint32_t t = 0; t < (uint32_t)T; ++t) for (uint32_t k = 0; k < W; ++k) wave_map.push_back(make_uint2(t, k | wgp));
    }
    StreamCtl ctl{}; ctl.next = d_next; ctl.cnt = d_cnt; ctl.mbox = d_mbox; ctl.n_jobs = (uint32_t)N;
    ctl.h_ready = c.pin_q + 0; ctl.h_abort = c.pin_q + 16; ctl.h_err = c.pin_q + 32; ctl.h_done = c.pin_q + 64;
    HIP_TRY(hipMemcpyAsync(d_in, c.pin_h2d, in_total, hipMemcpyHostToDevice, sB));
    HIP_TRY(hipMemsetAsync(d_zero, 0, zero_bytes, sB));
    HIP_TRY(hipMemcpyAsync(d_views, views.data(), sizeof(JobView) * N, hipMemcpyHostToDevice, sB));
    HIP_TRY(hipMemcpyAsync(d_wargs, wargs.data(), sizeof(WalkArgs) * N, hipMemcpyHostToDevice, sB));
    HIP_TRY(hipMemcpyAsync(d_wave_map, wave_map.data(), sizeof(uint2) * wave_map.size(), hipMemcpyHostToDevice, sB));
    HIP_TRY(hipMemcpyAsync(d_ctl, &ctl, sizeof(ctl), hipMemcpyHostToDevice, sB));
    HIP_TRY(hipStreamSynchronize(sB));
    c.tm.h2d_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_h2d0).count();
    if (const char* e = getenv("STITCH_STREAM_SETTLE_MS")) { HIP_TRY(hipDeviceSynchronize()); std::this_thread::sleep_for(std::chrono::milliseconds(atoi(e))); }      // (experiment)
    c.stream_abort_word = c.pin_q + 16; c.stream_stalled = false; c.stream_retired = 0; c.stream_may_retire = (uint32_t)std::min<size_t>(1, T > 0 ? T - 1 : 0);
    if (const char* e = getenv("STITCH_STREAM_BOUND_MS")) c.stream_bound_s = std::max(1, atoi(e)) * 1e-3;
    hq[0] = (uint32_t)std::min(N, B);                  // the first B jobs find their blocks free
    std::atomic_thread_fence(std::memory_order_seq_cst);
    hipEvent_t* const ev = c.evp[0];
    HIP_TRY(hipEventRecord(ev[0], sA));
    bool ybits = !c.knobs.no_ybits;                    // (every job keeps per-contig y-suffix records: one bit per cell for the column's common word)
    for (size_t k = 0; k < N; ++k) if (jobs[k].mode == 0) ybits = false;
    launch_fill_regs(d_views, d_wave_map, (uint32_t)wave_map.size(), c.regs_waves, W, c.opts.circular != 0, ybits, sh, d_ctl, sA);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev[1], sA));
    c.tm.fill_kind = 2u; c.tm.wg_per_read = (W + c.regs_waves - 1) / c.regs_waves; c.tm_wg_per_read = c.tm.wg_per_read;
    if (kn.trace) fprintf(stderr, "[trace] persistent teams: %zu jobs, %zu teams of %u waves, %zu blocks of %zu bytes\n", N, T, W, B, stride);

    // ---- the host's loop: finished jobs, in order, get their fix-up + walk and downloads on sB; then the job B further on may start ----
    size_t fin = 0;
    bool broken = false; int rc_fatal = STITCH_OK;
    auto t_last = std::chrono::steady_clock::now();
    const auto t_run0 = t_last;
    while (fin < N) {
        size_t b = fin;
        // At most eight jobs per fix-up + walk launch.  The fill's teams never leave, so a walk's workgroups only ever get the few wave slots
        // the fill left free, and the dispatcher deals a grid's workgroups to the XCDs (and their shader engines) in a fixed rotation: a
        // workgroup whose turn falls on a full engine waits there although another has room.  Measured (gpurun_out/r4b, r4c): walks of up
        // to six workgroups beside 2000 resident fill waves take their 7-9 ms, walks of 14 or more never start until fill waves leave.
        const uint32_t walk_wgs = 8;
        const size_t range_cap = kn.stream_range > 0 ? (size_t)kn.stream_range : std::max<size_t>(1, walk_wgs / ((W + 63) / 64));      // (fix-ups: 64 contigs per workgroup)
        while (b < N && hq[64 + b] != 0u && b - fin < range_cap) ++b;
        if (b == fin) {
            if (hq[32] != 0u) { broken = true; break; }                            // a wave gave up waiting (partner not resident, lost mailbox)
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_last).count() > 5.0 + 1e-3 * (double)lay[fin].n) { broken = true; break; }      // (no read takes that long: 10 us per column is the rule)
            std::this_thread::sleep_for(std::chrono::microseconds(40));
            continue;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        const uint32_t nj = (uint32_t)(b - fin);
        uint32_t max_nact_mode1 = 0;
        for (size_t k = fin; k < b; ++k) if (jobs[k].mode == 1) max_nact_mode1 = std::max(max_nact_mode1, lay[k].nact);
        HIP_TRY(hipEventRecord(ev[2], sB));
        launch_fixup_walk(d_views + fin, d_wargs + fin, nj, max_nact_mode1, sB, walk_wgs, W);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev[3], sB));
        const double tr0 = kn.trace ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run0).count() : 0.0;
        if (kn.trace) { if (int e2 = bounded_sync(c, sB)) return e2; }
        const double tr1 = kn.trace ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run0).count() : 0.0;
        for (uint32_t q2 = 0; q2 < nj; ++q2) HIP_TRY(hipMemcpyAsync(c.pin + 4ull * q2, views[fin + q2].err, 4, hipMemcpyDeviceToHost, sB));
        if (int e2 = bounded_sync(c, sB)) return e2;
        const double tr2 = kn.trace ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run0).count() : 0.0;
        { float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, ev[2], ev[3])); c.tm.walk_ms += ms; if (kn.trace) fprintf(stderr, "[trace] jobs %zu-%zu: walk submitted %.1f, done %.1f (kernel %.2f ms), error words %.1f\n", fin, b, tr0, tr1, ms, tr2); }
        for (uint32_t q2 = 0; q2 < nj; ++q2) {
            uint32_t e = 0; memcpy(&e, c.pin + 4ull * q2, 4);
            if ((e & 0xFFu) == 2u) { rc_fatal = fail(STITCH_EINTERNAL, "fill kernel bounds check failed, code " + std::to_string(e >> 8)); broken = true; }
            else if (e) broken = true;
        }
        if (broken) break;
        std::vector<uint8_t*> blk(nj);
        for (uint32_t q2 = 0; q2 < nj; ++q2) blk[q2] = block_of(fin + q2);
        const int rd = download_chains(c, sB, jobs, lay, fin, nj, blk, d_views + fin, false);
        if (rd == 1) { broken = true; break; }           // a chain wants the exact-size re-walk (device-synchronising): the classic path does it
        if (rd) { rc_fatal = rd; broken = true; break; }
        if (c.stream_stalled) { fin = b; broken = true; if (kn.trace) fprintf(stderr, "[trace] a launch beside the teams did not end within its bound: the run is called off after job %zu\n", b); break; }
        if (kn.trace) fprintf(stderr, "[trace] jobs %zu-%zu walked and downloaded at %.1f ms\n", fin, b, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run0).count());
        fin = b;
        std::atomic_thread_fence(std::memory_order_release);
        hq[0] = (uint32_t)std::min(N, fin + B);
        t_last = std::chrono::steady_clock::now();
    }
    if (c.stream_stalled) broken = true;
    if (broken) { hq[16] = 0xFFFFFFFFu; std::atomic_thread_fence(std::memory_order_seq_cst); }
    c.stream_abort_word = nullptr;
    HIP_TRY(hipStreamSynchronize(sA));                  // (every team leaves once the queue is empty or called off; every wait in the kernel is bounded)
    { float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1])); c.tm.fill_ms += ms; c.tm.fill_kernel_ms += ms; }
    c.tm.launches += 1; c.tm.jobs += (uint32_t)fin; c.tm.stream_runs += 1;
    {
        HIP_TRY(hipMemcpyAsync(c.pin, (const uint8_t*)views[0].err + ERR_CLOCK_OFF, 16, hipMemcpyDeviceToHost, sB));
        HIP_TRY(hipStreamSynchronize(sB));
        unsigned long long ck[2]; memcpy(ck, c.pin, 16);
        c.tm.clk_shader_cycles += ck[0]; c.tm.clk_ref_ticks += ck[1];
    }
    if (rc_fatal) return rc_fatal;
    if (broken) {
        // the run was called off (a partner that was not resident, a chain beyond its buffer): the jobs that are not finished run launch
        // by launch on the classic path, which has the fallback kernels
        c.tm.fallbacks += 1;
        std::vector<Job> rest; rest.reserve(N - fin);
        for (size_t k = fin; k < N; ++k) rest.push_back(std::move(jobs[k]));
        for (const Job& jb : rest) { uint64_t rows = 0; for (uint32_t a : jb.act) rows += c.al[a].m; c.tm.cells -= (uint64_t)jb.y.size() * rows; }      // (counted again below)
        const int rc = run_jobs_in_order(c, rest);
        for (size_t k = fin; k < N; ++k) jobs[k] = std::move(rest[k - fin]);
        if (rc) return rc;
    }
    *handled = true;
    return STITCH_OK;
}

static int run_jobs_in_order(stitch_ctx& c, std::vector<Job>& jobs) {
    if (jobs.empty()) return STITCH_OK;
    if (!c.in_stream_fallback) {
        // A context's first jobs always go launch by launch.  Whatever the runtime sets up on first use — code objects, its copy kernels, pools
        // that grow — must not happen for the first time beside resident teams: measured, the first call of a process as a persistent-team
        // run stalls for the kernel's whole bounded waits (86 s against 0.8 s for the same call made second, gpurun_out/r4t, r4v), and one
        // small classic call before it is enough.  So the first jobs of a context's first eligible call (a quarter of them, at most 40) run as one classic launch.
        if (!c.warmed_up && !c.knobs.no_stream && jobs.size() >= 4) {
            bool uniform = true;
            for (const Job& jb : jobs) if (jb.act.size() != jobs[0].act.size() || regs_plan(c, jb) == 0) { uniform = false; break; }
            if (uniform) {
                const size_t H = std::min<size_t>(40, std::max<size_t>(1, jobs.size() / 4));
                std::vector<Job> head; head.reserve(H);
                for (size_t k = 0; k < H; ++k) head.push_back(std::move(jobs[k]));
                c.in_stream_fallback = true;
                int rc = run_jobs_in_order(c, head);
                c.in_stream_fallback = false;
                for (size_t k = 0; k < H; ++k) jobs[k] = std::move(head[k]);
                if (rc) return rc;
                std::vector<Job> rest; rest.reserve(jobs.size() - H);
                for (size_t k = H; k < jobs.size(); ++k) rest.push_back(std::move(jobs[k]));
                rc = run_jobs_in_order(c, rest);
                for (size_t k = H; k < jobs.size(); ++k) jobs[k] = std::move(rest[k - H]);
                return rc;
            }
        }
        bool handled = false;
        c.in_stream_fallback = true;                  // (the persistent-team run hands what it could not finish to this function)
        const int rc = c.warmed_up ? run_jobs_streaming(c, jobs, &handled) : STITCH_OK;
        c.in_stream_fallback = false;
        if (rc || handled) return rc;
    }
    HIP_TRY(hipSetDevice(c.device));
    std::vector<JobLayout> lay(jobs.size());
    size_t max_job = 0;
    size_t max_in = 0;
    for (size_t k = 0; k < jobs.size(); ++k) { lay[k] = layout_job(c, jobs[k]); max_job = std::max(max_job, lay[k].bytes); max_in = std::max(max_in, lay[k].off_hdr - lay[k].off_y); }
    // a launch keeps its jobs' small inputs (read, active contigs, opposite strands, contig table) and their exchange granules + error words
    // in ONE region behind the job blocks: one upload and one clear per launch instead of one of each per job (cfg3: 8033 copies and 2702
    // clears per 2048 reads, 6.7 % of the GPU time and as many host submissions)
    const size_t xbytes = align_up(32ull * c.C + 4096, 256), in_room = align_up(max_in, 256) + xbytes;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    size_t budget = (size_t)(std::min(free_b + c.arena_bytes, total_b) * 0.90);
    if (c.mem_limit) budget = std::min(budget, c.mem_limit);
    if (max_job + (1 << 20) > budget) return fail(STITCH_ENOMEM, "one read needs " + std::to_string(max_job >> 20) + " MiB of device memory; only " + std::to_string(budget >> 20) + " MiB usable");
    // arena: as much of the free memory as useful, at least one job
    // (a launch never holds more jobs than there are compute units, so more than the largest such window is never used)
    // Job blocks start at multiples of a large power of two.  Measured (MI355X, cfg2, same box): with the blocks packed back
    // to back the reads of a launch run at very different speeds depending on where their block starts (the slowest ends
    // 170 ms after the fastest in a 585 ms launch, the same slots every time, whatever read or CUs they get); blocks at
    // multiples of 1 GiB: 505 ms, 128 or 512 MiB: 541, 16 MiB: 695, 1 GiB + 256 KiB: 594.  So: the largest power of two
    // (<= 1 GiB, <= the job size) that still lets the launch window fit the memory.
    size_t want = 0, block_align = 256, win_jobs = jobs.size(), regs_wave_cap = 0;
    for (size_t a = (size_t)1 << 30; a >= 256; a >>= 1) {
        if (a > max_job && a > 256) continue;
        if (c.knobs.job_align) a = std::max<size_t>(256, c.knobs.job_align);      // (experiments)
        for (size_t k = 0; k < jobs.size(); ++k) lay[k].stride = align_up(lay[k].bytes, a);
        // a spacing that is a multiple of 4 a leaves the two address bits above the alignment equal in all jobs: measured as bad
        // as no alignment at all (cfg2 at 4 GiB: 616 ms against 445 at 3 GiB); half the alignment then gives an odd multiple
        // (14 kb reads: 3.5 GiB spacing 613-688 ms, 4 GiB 845-866)
        if (a > 256 && !c.knobs.job_align && align_up(max_job, a) % (4 * a) == 0) continue;
        block_align = a; want = 0;
        // (the Local-mode kernel gives a read of T tiles min(4, ceil(T / 250)) workgroups, see the launch loop below)
        bool all_fast = true; uint32_t gd_min = 4;
        for (const Job& jb : jobs) {
            if (!local16_ok(c, jb)) { all_fast = false; break; }
            uint32_t tiles = 0; for (uint32_t a : jb.act) tiles += (c.al[a].m + 255) / 256;
            gd_min = std::min(gd_min, std::min(4u, std::max(1u, (tiles + 249u) / 250u)));
        }
        // (reads that get ONE workgroup each do not wait for anybody, so any number of them shares a launch)
        if (c.knobs.wg_per_read) gd_min = (uint32_t)c.knobs.wg_per_read;          // (experiments)
        // (the register-resident kernel gives a read regs_plan() workgroups, all resident at once)
        uint32_t rg_min = 0xFFFFFFFFu;
        if (all_fast) for (const Job& jb : jobs) rg_min = std::min(rg_min, regs_plan(c, jb));
        size_t win = (all_fast && gd_min > 1) ? std::min<size_t>(jobs.size(), (size_t)std::max(1, c.n_cus) / gd_min) : jobs.size();
        if (all_fast && rg_min > 0 && rg_min != 0xFFFFFFFFu && !c.knobs.force_regs32) {
            // the register kernel: a launch holds the reads whose contigs fill the chip's wave slots - or an equal share of the batch
            // where that leaves a remainder: a launch lasts its reads' columns however few they are, and two launches of half the
            // chip each run side by side (two windows, two streams) in the time one of them takes alone
            const size_t slots = (size_t)c.n_cus * (size_t)c.regs_wg_per_cu * c.regs_waves;
            size_t W = 0, act_max = 1; for (const Job& jb : jobs) { W += jb.act.size(); act_max = std::max(act_max, jb.act.size()); }
            regs_wave_cap = slots;
            if (W > slots) { const size_t nl = (W + slots - 1) / slots; regs_wave_cap = std::min(slots, (W + nl - 1) / nl + act_max); }
            size_t most = 1, cur_jobs = 0, cur_waves = 0;
            for (const Job& jb : jobs) {
                if (cur_jobs && cur_waves + jb.act.size() > regs_wave_cap) { cur_jobs = 0; cur_waves = 0; }
                ++cur_jobs; cur_waves += jb.act.size(); most = std::max(most, cur_jobs);
            }
            win = most;
        }
        if (!all_fast || c.knobs.force_regs32) {      // (the 32-bit register-resident kernel: one workgroup per CU, all workgroups of a launch resident)
            uint32_t r32_min = 0xFFFFFFFFu;
            for (const Job& jb : jobs) r32_min = std::min(r32_min, regs32_plan(c, jb));
            if (r32_min > 0 && r32_min != 0xFFFFFFFFu) {      // (as many reads as its wave slots hold: the waves are dealt densely there too)
                size_t act_min = ~(size_t)0; for (const Job& jb : jobs) act_min = std::min(act_min, jb.act.size());
                win = std::min<size_t>(jobs.size(), std::max<size_t>(1, (size_t)c.n_cus * (size_t)c.regs32_wg_per_cu * REGS_WAVES / std::max<size_t>(1, act_min)));
            }
        }
        win_jobs = win;
        size_t cur = 0;
        for (size_t k = 0; k < jobs.size(); ++k) {
            cur += lay[k].stride + sizeof(JobView) + sizeof(WalkArgs) + 512 + 8ull * c.C + in_room;
            if (k >= win) cur -= lay[k - win].stride + sizeof(JobView) + sizeof(WalkArgs) + 512 + 8ull * c.C + in_room;
            want = std::max(want, cur);
        }
        want += (size_t)2 << 20;
        if (want + (1 << 20) <= budget || a == 256 || c.knobs.job_align) break;      // fits (or nothing smaller to try)
    }
    if (c.knobs.debug) fprintf(stderr, "[stitch] job blocks at multiples of %zu bytes, window %zu bytes\n", block_align, want);
    // Two windows where more than one launch is expected and both fit: the fill of launch k + 1 then runs beside fix-up, walk,
    // downloads and host parsing of launch k (6 ms of 139 per launch at cfg2, more with --suboptimal's hundreds of chains).
    const bool quiet = !c.knobs.debug && !c.knobs.profile_dump && !c.knobs.fill_only && c.knobs.dump_dir.empty() && !c.knobs.no_pipeline && (!c.knobs.fail_first_attempt || c.knobs.fail_keeps_pipeline);
    const size_t win_align = std::max<size_t>(block_align, 256);
    const bool want_two = quiet && jobs.size() > win_jobs && 2 * align_up(want + (1 << 20), win_align) + win_align <= budget;
    // Where two full windows do not fit (cfg5: ten 20 kb reads against 200 contigs are 200 GB of traceback) but half the memory still
    // holds several reads, the launches are made half as large instead and two of them are in flight: the same number of
    // workgroups on the chip, and fix-up, walk and downloads of one launch beside the fill of the next.
    size_t max_stride = 0; for (size_t k = 0; k < jobs.size(); ++k) max_stride = std::max(max_stride, lay[k].stride);
    const size_t per_job_room = sizeof(JobView) + sizeof(WalkArgs) + 512 + 8ull * c.C + in_room;
    const bool want_halves = quiet && !want_two && win_jobs >= 4 && jobs.size() >= 4 && want + (1 << 20) > budget / 2 &&
                             2 * (max_stride + per_job_room) + ((size_t)4 << 20) <= budget / 2 / win_align * win_align;
    size_t arena_need = std::min(want_two ? 2 * align_up(want + (1 << 20), win_align) : want_halves ? budget : want + (1 << 20), budget);
    if (arena_need > c.arena_bytes) {
        // growing costs seconds (free + allocate: ~5 s for 250 GB), so a context that has to grow takes half as much again
        if (c.arena) arena_need = std::min(budget, arena_need + arena_need / 2);
        if (c.arena) {
            HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c.arena_raw)); c.arena_raw = nullptr; c.arena = nullptr; c.arena_bytes = 0;
            // (what hipMemGetInfo reports right after the free lags behind, in both directions on different runs: the request
            // stays what the budget above allows, and the loop below shrinks it if the allocation fails)
        }
        // a smaller arena only means more launches: shrink until the allocation succeeds or one read no longer fits
        for (;;) {
            if (c.knobs.debug) fprintf(stderr, "[stitch] arena -> %zu bytes (%zu jobs, %zu free), previous base %p\n", arena_need, jobs.size(), free_b, (void*)c.arena_raw);
            // (the base too is a multiple of the block alignment: the blocks' addresses, not only their distances, are then the
            // same in every process)
            const size_t slack = block_align > 256 ? block_align : 0;
            if (hipMalloc((void**)&c.arena_raw, arena_need + slack) == hipSuccess) { c.arena = (uint8_t*)align_up((size_t)(uintptr_t)c.arena_raw, std::max<size_t>(block_align, 256)); break; }
            (void)hipGetLastError(); c.arena_raw = nullptr; c.arena = nullptr;
            const size_t smaller = std::max(max_job + ((size_t)1 << 20), (size_t)(arena_need * 0.85));
            if (smaller >= arena_need) return fail(STITCH_ENOMEM, "cannot allocate " + std::to_string(arena_need >> 20) + " MiB of device memory for one read");
            arena_need = smaller;
        }
        c.arena_bytes = arena_need;
        if (c.knobs.debug) fprintf(stderr, "[stitch] arena at %p (allocation at %p)\n", (void*)c.arena, (void*)c.arena_raw);
    }
    // (an arena that was large enough already may hold two windows too)
    const size_t half = (c.arena_bytes / 2) / win_align * win_align;
    const bool pipeline = quiet && ((jobs.size() > win_jobs && half >= want + (1 << 20)) ||
                                    (want_halves && half >= 2 * (max_stride + per_job_room) + ((size_t)4 << 20)));
    const size_t win_bytes = pipeline ? half : c.arena_bytes;
    if (c.knobs.debug) fprintf(stderr, "[stitch] %s window(s) of %zu bytes\n", pipeline ? "two" : "one", win_bytes);
    FillShared sh{c.d_S0, c.d_Slen0, c.d_Sn0, c.d_SnSet0, c.d_Smove0, c.d_lx0, c.d_base0};
    bool fast = true;
    for (const Job& jb : jobs) if (!local16_ok(c, jb)) { fast = false; break; }
    c.tm_fast = fast;

    // One launch: the jobs [k0, k1) in one window of the arena.  start() packs them, uploads their inputs and launches the fill;
    // finish() runs fix-up + walk, checks the kernels' error words (and repeats the launch on a fallback kernel once), downloads
    // the chains.  With two windows the fill of launch k + 1 is started BEFORE launch k is finished: fix-up, walk, downloads and
    // the host's parsing of launch k then run beside the next fill instead of between two fills.
    struct Launch {
        size_t k0 = 0, k1 = 0, win_base = 0; uint32_t nj = 0, regs_G = 0, regs32_G = 0, g_min = 1, G = 1, kind = 0, slots_cap = 0; int waves = 1, slot = 0;
        std::vector<JobView> views; std::vector<WalkArgs> wargs; std::vector<size_t> base; JobView* d_views = nullptr; WalkArgs* d_wargs = nullptr;
        uint8_t* d_x = nullptr; size_t x_bytes = 0;       // the launch's exchange granules + error words (one clear)
        double h_start = 0, h_submit = 0;                // (STITCH_TRACE) host clock at start() and at the fill's submission, ms since the call's base
    };
    const auto t_trace0 = std::chrono::steady_clock::now();
    auto host_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_trace0).count(); };
    // Streams.  sB: uploads, fix-up + walk, downloads.  The fills of window 0 run on c.stream, those of window 1 on c.stream3: the fill of
    // launch k + 1 is NOT ordered behind the fill of launch k.  The reads of a launch end at different times (an unalignable read keeps
    // every insertion chain alive, a clean one skips most of that work: 20-30 % apart), and a launch sized to fill the chip holds it
    // until its slowest read is done; with the next launch already queued, the dispatcher hands every set of workgroup slots a
    // finished read frees to the next launch's workgroups, in block order — whole teams, since a read's workgroups are consecutive
    // blocks.  A team that is only partly resident spins (bounded) until the slots it lacks come free, which the older launch
    // guarantees by running to its end on its own; launch k + 2 shares a stream with launch k and therefore starts only after it, by
    // which time launch k + 1 has long been fully resident: never more than one launch is left waiting for slots.
    hipStream_t const sB = pipeline ? c.stream2 : c.stream;
    const bool overlap_fills = pipeline && !c.knobs.no_fill_overlap;
    HIP_TRY(hipEventRecord(c.ev[0], c.stream));      // time base of this call's launches
    double covered_until = 0.0;                      // ms since the base up to which some fill was running
    auto start = [&](const size_t k0, const int slot, Launch& Ln) -> int {
        // greedy pack of consecutive jobs into the window
        size_t k1 = k0, used = 0;
        Ln.h_start = host_ms();
        const size_t win_base = (size_t)slot * win_bytes;
        hipEvent_t* const ev = c.evp[slot];
        hipStream_t const sA = (overlap_fills && slot == 1) ? c.stream3 : c.stream;
        const size_t view_room = 1 << 20;
        // Local-mode kernel: all G workgroups of all reads of a launch must be resident at once (one workgroup per CU), and a
        // workgroup's slot table holds 2048 tiles.
        size_t max_jobs = 4096;
        uint32_t g_min = 1;
        const uint32_t regs_G = (fast && !c.knobs.force_regs32) ? regs_plan(c, jobs[k0]) : 0u;       // > 0: this launch runs the register-resident kernel
        // the 32-bit register-resident kernel: where no 16-bit Local-mode kernel applies (other clipping modes, long reads)
        const uint32_t regs32_G = (regs_G == 0 && (!fast || c.knobs.force_regs32)) ? regs32_plan(c, jobs[k0]) : 0u;
        // every workgroup of a team resident at once: a launch holds no more teams than the chip does - unless a team is ONE workgroup, whose
        // waves are together by construction (reads cut down to a few contigs by the filter: a fill launch lasts its reads' columns
        // whatever their number, so the more the better)
        // fill_regs.hip deals the launch's waves to the reads' contigs densely (a team's waves sit in any workgroups): the launch holds
        // what the chip's wave slots hold
        const size_t regs_wave_slots = regs_G ? (regs_wave_cap ? regs_wave_cap : (size_t)c.n_cus * (size_t)std::max(c.regs_wg_per_cu, 0) * c.regs_waves)
                                              : (size_t)c.n_cus * (size_t)std::max(c.regs32_wg_per_cu, 0) * REGS_WAVES;
        size_t regs_waves_used = 0;
        if (regs_G || regs32_G) max_jobs = 4096;
        else if (fast) {
            uint32_t tiles = 0; for (uint32_t a : jobs[k0].act) tiles += (c.al[a].m + 255) / 256;
            g_min = local16_min_g(c, jobs[k0]);
            // at least 4 workgroups per read: measured best on cfg2 (64 reads x 4 beats 85 x 3 by 15 %: shorter columns per
            // workgroup, and 50 contigs still split evenly)
            // (reads aligned to a few contigs only -- pre-alignment subsets, origin re-alignments -- get fewer workgroups each,
            // about 250 tiles per workgroup, and more of them share a launch)
            const uint32_t g_des = std::min(4u, std::max(1u, (tiles + 249u) / 250u));
            max_jobs = std::max<size_t>(1, (size_t)c.n_cus / std::max(g_min, g_des));
            // one workgroup per read: nothing waits across workgroups, so the launch may hold more reads than there are CUs
            // (two or three such workgroups of a few waves share a CU and fill each other's per-column stalls)
            if (std::max(g_min, g_des) == 1) max_jobs = 4096;
            if (c.knobs.wg_per_read) max_jobs = std::max<size_t>(1, (size_t)c.n_cus / std::max(g_min, (uint32_t)c.knobs.wg_per_read));
        }
        const size_t per_job = sizeof(JobView) + sizeof(WalkArgs) + 512 + 8ull * c.C + in_room;     // the launch's job table, wave map, inputs and granules, after the jobs' own buffers
        while (k1 < jobs.size() && used + lay[k1].stride + per_job + view_room <= win_bytes && (k1 - k0) < max_jobs &&
               (k1 == k0 || (regs32_G ? regs32_plan(c, jobs[k1]) > 0 : regs_G ? regs_plan(c, jobs[k1]) > 0 : (regs_plan(c, jobs[k1]) == 0 || !fast))) &&
               ((!regs_G && !regs32_G) || k1 == k0 || regs_waves_used + lay[k1].nact <= regs_wave_slots)) {
            if (regs_G || regs32_G) regs_waves_used += lay[k1].nact;
            if (fast && !regs_G && !regs32_G && k1 > k0) {
                // a later job may need MORE workgroups than the first (shorter read, more contigs): all workgroups of the launch
                // must still be resident at once
                const uint32_t gk = local16_min_g(c, jobs[k1]);
                if (gk > g_min) { if ((k1 - k0 + 1) * (size_t)gk > (size_t)c.n_cus) break; g_min = gk; max_jobs = std::min(max_jobs, std::max<size_t>(1, (size_t)c.n_cus / gk)); }
            }
            used += lay[k1].stride + per_job; ++k1;
        }
        if (k1 == k0) return fail(STITCH_ENOMEM, "arena too small for one job");
        const uint32_t nj = (uint32_t)(k1 - k0);
        Ln.k0 = k0; Ln.k1 = k1; Ln.nj = nj; Ln.win_base = win_base; Ln.slot = slot; Ln.regs_G = regs_G; Ln.regs32_G = regs32_G;
        Ln.views.assign(nj, JobView{}); Ln.wargs.assign(nj, WalkArgs{}); Ln.base.assign(nj, 0);
        std::vector<JobView>& views = Ln.views; std::vector<WalkArgs>& wargs = Ln.wargs; std::vector<size_t>& base = Ln.base;
        int waves = 1;
        auto t_h2d0 = std::chrono::steady_clock::now();
        // the small per-job inputs (read, active contigs, opposite strands, contig table) are contiguous in a job's block:
        // they are assembled in one pinned buffer and go up with one copy per job, without a synchronisation in between
        std::vector<size_t> stage_at(nj + 1, 0);
        for (uint32_t q = 0; q < nj; ++q) stage_at[q + 1] = stage_at[q] + align_up(lay[k0 + q].off_hdr - lay[k0 + q].off_y, 256);
        // (where the region will be: behind the blocks and the launch's tables, laid out below)
        size_t blocks_end = 0; for (uint32_t q = 0; q < nj; ++q) blocks_end += lay[k0 + q].stride;
        uint8_t* tail0 = c.arena + win_base + align_up(blocks_end, 256);
        size_t n_map = 0; if (regs_G || regs32_G) { for (uint32_t q = 0; q < nj; ++q) n_map += lay[k0 + q].nact; n_map += 16; }
        uint8_t* const d_in = tail0 + align_up(sizeof(JobView) * nj, 256) + align_up(sizeof(WalkArgs) * nj, 256) + align_up(sizeof(uint2) * n_map, 256);
        uint8_t* const d_x = d_in + align_up(stage_at[nj], 256);
        Ln.d_x = d_x; Ln.x_bytes = xbytes * nj;
        if (stage_at[nj] > c.pin_h2d_bytes) {
            if (c.pin_h2d) { (void)hipHostFree(c.pin_h2d); c.pin_h2d = nullptr; c.pin_h2d_bytes = 0; }
            const size_t want_b = std::max<size_t>(stage_at[nj] * 3 / 2, (size_t)1 << 20);
            HIP_TRY(hipHostMalloc((void**)&c.pin_h2d, want_b, hipHostMallocDefault));
            c.pin_h2d_bytes = want_b;
        }
        size_t o = 0;
        for (uint32_t q = 0; q < nj; ++q) {
            const Job& jb = jobs[k0 + q]; const JobLayout& L = lay[k0 + q];
            base[q] = win_base + o; uint8_t* B = c.arena + win_base + o; o += L.stride;
            waves = std::max(waves, pick_waves(c, L.nact, MAX_WAVES_GENERIC));
            // per-job tables
            std::vector<ContigDesc> cd(c.C); std::vector<int32_t> opp(c.C, -1); std::vector<uint8_t> isact(c.C, 0);
            for (uint32_t a : jb.act) isact[a] = 1;
            uint32_t roff = 0;
            for (uint32_t a = 0; a < c.C; ++a) {
                ContigDesc d{}; d.m = c.al[a].m; d.troff = c.al[a].troff; d.seqoff = c.al[a].seqoff; d.target = c.al[a].target; d.opp = c.al[a].opp;
                d.roff = 0;
                { const uint32_t ngr = (d.m + 3) / 4, gq = ngr / 64; d.inv_big = tb_div_magic(4 * (gq + 1)); d.inv_small = tb_div_magic(4 * gq); }
                if (isact[a]) { d.roff = roff; roff += (d.m + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS; }
                cd[a] = d;
                // the opposite strand only counts when it is part of the current subset (multi_contig_aligner.rs:241-262)
                if (isact[a] && c.al[a].opp >= 0 && isact[c.al[a].opp]) opp[a] = c.al[a].opp;
            }
            uint8_t* stg = c.pin_h2d + stage_at[q];
            memcpy(stg, jb.y.data(), L.n);
            memcpy(stg + (L.off_act - L.off_y), jb.act.data(), 4ull * L.nact);
            memcpy(stg + (L.off_opp - L.off_y), opp.data(), 4ull * c.C);
            memcpy(stg + (L.off_cd - L.off_y), cd.data(), sizeof(ContigDesc) * (size_t)c.C);
            uint8_t* const In = d_in + stage_at[q]; uint8_t* const X = d_x + xbytes * q;
            JobView& V = views[q];
            V.tb_keyfmt = regs_G ? 2u : regs32_G ? 3u : fast ? 1u : 0u; V.yrec_global = jb.mode == 0 ? 1u : 0u;
            V.P = c.P; V.n = L.n; V.C = c.C; V.nact = L.nact; V.Rtot = L.Rj;
            V.act = (const uint32_t*)(In + (L.off_act - L.off_y)); V.opp_act = (const int32_t*)(In + (L.off_opp - L.off_y)); V.cd = (const ContigDesc*)(In + (L.off_cd - L.off_y));
            V.xseq = c.d_xseq; V.y = In;
            V.S = (int32_t*)(B + L.off_S); V.Slen = (uint32_t*)(B + L.off_Slen); V.D = (int32_t*)(B + L.off_D); V.Dlen = (uint32_t*)(B + L.off_Dlen);
            V.st16 = (uint32_t*)(B + L.off_st16);
            V.xchg = (unsigned long long*)X; V.err = (uint32_t*)(X + 32ull * c.C);
            V.Sn = (int32_t*)(B + L.off_Sn); V.SnLen = (uint32_t*)(B + L.off_SnLen); V.Ly = (uint32_t*)(B + L.off_Ly);
            V.tb = B + L.off_tb; V.Lx = (uint32_t*)(B + L.off_Lx); V.jt_idx = (uint32_t*)(B + L.off_jti); V.jt_from = (uint32_t*)(B + L.off_jtf);
            V.Ival = (int32_t*)(B + L.off_Ival); V.Ilen = (uint32_t*)(B + L.off_Ilen); V.SmoveF = B + L.off_SmoveF;
            V.SidxF = (uint32_t*)(B + L.off_SidxF); V.SfromF = (uint32_t*)(B + L.off_SfromF); V.ImoveF = B + L.off_ImoveF;
            V.Smove0 = c.d_Smove0; V.Imove0 = c.d_Imove0; V.Slen0 = c.d_Slen0;
            V.Sm = (int32_t*)(B + L.off_Sm); V.Lm = (uint32_t*)(B + L.off_Lm);
            V.visit = (jb.mode == 1 && !c.knobs.no_join) ? (VisitRec*)(B + L.off_visit) : nullptr;
            V.Wcol = jb.mode != 0 ? (uint32_t*)(B + L.off_Wcol) : nullptr;
            WalkArgs& A = wargs[q]; A.hdr = (ChainHdr*)(B + L.off_hdr); A.ops = (OpRec*)(B + L.off_ops); A.ops_cap = L.ops_cap; A.mode = jb.mode; A.from = jb.from; A.skip_fixup = 0;
            c.tm.cells += (uint64_t)L.n * [&] { uint64_t s = 0; for (uint32_t a : jb.act) s += c.al[a].m; return s; }();
        }
        // launch-level tables live after the jobs
        uint8_t* tail = c.arena + win_base + align_up(o, 256);
        JobView* d_views = (JobView*)tail; tail += align_up(sizeof(JobView) * nj, 256);
        WalkArgs* d_wargs = (WalkArgs*)tail; tail += align_up(sizeof(WalkArgs) * nj, 256);
        std::vector<uint2> wave_map;                     // fill_regs.hip: wave of the grid -> (read of the launch, active contig)
        uint32_t wgp = (regs_G && !c.knobs.no_wg_poll) ? 0x20000000u : 0u;      // fill_regs.hip: one poller per workgroup where every read's contigs fill whole workgroups
        for (uint32_t q = 0; q < nj; ++q) if (lay[k0 + q].nact % c.regs_waves != 0) wgp = 0u;
        if (regs_G || regs32_G) for (uint32_t q = 0; q < nj; ++q) for (uint32_t k = 0; k < lay[k0 + q].nact; ++k) wave_map.push_back(make_uint2(q, k | wgp));
        if (regs_G && c.knobs.regs_map == 1 && c.regs_waves == 8 && wave_map.size() >= 16) {
            // (experiment) eight-wave workgroups whose two waves per SIMD (waves t and t + 4) come from reads half a launch apart, as
            // two four-wave workgroups of one CU do; entries without a wave carry a contig number no read has
            const size_t W = wave_map.size(), half = (W / 2 + 3) / 4 * 4;
            std::vector<uint2> m2;
            for (size_t i = 0; 4 * i < half; ++i) for (uint32_t t = 0; t < 8; ++t) {
                const size_t src = t < 4 ? 4 * i + t : half + 4 * i + (t - 4);
                m2.push_back((t < 4 ? src < half : src < W) && src < W ? wave_map[src] : make_uint2(0u, 0xFFFFFFFFu));
            }
            wave_map.swap(m2);
        }
        uint2* d_wave_map = (uint2*)tail; tail += align_up(sizeof(uint2) * std::max(wave_map.size(), n_map), 256);
        if (wave_map.size() > n_map && (regs_G || regs32_G)) return fail(STITCH_EINTERNAL, "wave map larger than planned");
        if (tail > d_in) return fail(STITCH_EINTERNAL, "launch tables overlap the input region");
        tail = d_x + xbytes * nj;
        if ((size_t)(tail - c.arena) > win_base + win_bytes) return fail(STITCH_EINTERNAL, "arena overflow");
        HIP_TRY(hipMemcpyAsync(d_in, c.pin_h2d, stage_at[nj], hipMemcpyHostToDevice, sB));      // every job's inputs: one copy
        HIP_TRY(hipMemsetAsync(d_x, 0, xbytes * nj, sB));                                       // ... granules and error words: one clear
        Ln.d_views = d_views; Ln.d_wargs = d_wargs;
        if (!wave_map.empty()) HIP_TRY(hipMemcpyAsync(d_wave_map, wave_map.data(), sizeof(uint2) * wave_map.size(), hipMemcpyHostToDevice, sB));
        HIP_TRY(hipMemcpyAsync(d_views, views.data(), sizeof(JobView) * nj, hipMemcpyHostToDevice, sB));
        HIP_TRY(hipMemcpyAsync(d_wargs, wargs.data(), sizeof(WalkArgs) * nj, hipMemcpyHostToDevice, sB));
        HIP_TRY(hipStreamSynchronize(sB));              // (the staging buffer is free again; with two windows the fill before this one is still running on sA)
        c.tm.h2d_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_h2d0).count();

        // Kernel 1 and 2, timed with events on the stream they run on
        HIP_TRY(hipEventRecord(ev[0], sA));
        // workgroups per read for the Local-mode kernel: fill the CUs, but keep every workgroup resident at once (they
        // wait for each other every column) and leave each at least a couple of contigs
        uint32_t G = 1;
        if (regs_G) G = regs_G;
        else if (regs32_G) G = regs32_G;
        else if (fast) {
            uint32_t min_act = 0xFFFFFFFFu;
            for (uint32_t q = 0; q < nj; ++q) min_act = std::min(min_act, lay[k0 + q].nact);
            const uint32_t cus = (uint32_t)c.n_cus;
            // most workgroups per read that fit, preferring counts that split the contigs evenly (all workgroups of a read
            // meet every column, so the one with the most contigs sets the pace)
            const uint32_t g_cap = std::max(1u, std::min({cus / nj, min_act, 32u}));
            double best = 0;
            for (uint32_t g = 1; g <= g_cap; ++g) {
                const double eff = (double)min_act / g / (double)((min_act + g - 1) / g);
                if (g * eff > best + 1e-9) { best = g * eff; G = g; }
            }
            if (c.knobs.wg_per_read) G = (uint32_t)c.knobs.wg_per_read;
            G = std::max(G, g_min);
            if (nj * G > cus) G = std::max(1u, cus / nj);
            // the slot table of the fullest workgroup must hold its tiles (round-robin dealing is not monotone in G)
            for (;;) {
                uint32_t worst = 0; for (uint32_t q = 0; q < nj; ++q) worst = std::max(worst, local16_wg_tiles(c, jobs[k0 + q], G));
                if (worst <= fill_local16_max_slots()) break;
                if ((G + 1) * nj > cus) return fail(STITCH_EINTERNAL, "no workgroup count fits the Local-mode kernel's slot table for this launch");
                ++G;
            }
            // the waves of a workgroup share its tiles evenly (fill_local16.hip), so use all 12 unless there are fewer tiles
            uint32_t min_tiles = 0xFFFFFFFFu;
            for (uint32_t q = 0; q < nj; ++q) { uint32_t t = 0; for (uint32_t a : jobs[k0 + q].act) t += (c.al[a].m + 255) / 256; min_tiles = std::min(min_tiles, t / G); }
            // ... of a few tiles each: a contig cut across waves is a serial chain of hand-offs within a column (about 1 us
            // each), which dominates when a workgroup holds only a contig or two (re-alignment jobs on a chain's contigs)
            const uint32_t tpw = c.knobs.tiles_per_wave ? (uint32_t)c.knobs.tiles_per_wave : 5u;
            waves = (int)std::max(1u, std::min<uint32_t>(MAX_WAVES_LOCAL, min_tiles / tpw));
            if (c.knobs.max_waves) waves = std::max(1, std::min(waves, c.knobs.max_waves));
        }
        c.tm_wg_per_read = G; c.tm.wg_per_read = G; c.tm.fill_kind = regs_G ? 2u : regs32_G ? 3u : fast ? 1u : 0u;
        uint32_t slots_cap = 0;                          // tiles of the launch's largest workgroup (contigs are dealt round-robin to a read's G workgroups)
        if (fast && !regs_G && !regs32_G) for (uint32_t q = 0; q < nj; ++q) {
            std::vector<uint32_t> per(G, 0);
            const std::vector<uint32_t>& act = jobs[k0 + q].act;
            for (size_t k = 0; k < act.size(); ++k) per[k % G] += (c.al[act[k]].m + 255) / 256;
            for (uint32_t v : per) slots_cap = std::max(slots_cap, v);
        }
        const uint32_t kind = regs_G ? 2u : regs32_G ? 3u : fast ? 1u : 0u;
        Ln.G = G; Ln.kind = kind; Ln.waves = waves; Ln.slots_cap = slots_cap; Ln.g_min = g_min;
        if (kind == 3u) { uint32_t mx = 0; for (uint32_t q = 0; q < nj; ++q) mx = std::max(mx, lay[k0 + q].nact); launch_fill_regs32(d_views, d_wave_map, (uint32_t)wave_map.size(), mx, c.opts.circular != 0, sh, sA); }
        else if (kind == 2u) { uint32_t mx = 0; bool ybits = !c.knobs.no_ybits; for (uint32_t q = 0; q < nj; ++q) { mx = std::max(mx, lay[k0 + q].nact); if (jobs[k0 + q].mode == 0) ybits = false; } launch_fill_regs(d_views, d_wave_map, (uint32_t)wave_map.size(), c.regs_waves, mx, c.opts.circular != 0, ybits, sh, nullptr, sA); }
        else if (kind == 1u) launch_fill_local16(d_views, nj, G, waves, slots_cap, sh, sA);
        else launch_fill(d_views, nj, waves, sh, sA);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev[1], sA));
        Ln.h_submit = host_ms();
        return STITCH_OK;
    };
    auto finish = [&](Launch& Ln) -> int {
        const size_t k0 = Ln.k0, k1 = Ln.k1; const uint32_t nj = Ln.nj; (void)k1;
        const double h_fin0 = host_ms(); double h_walked = 0; float tr_fill0 = 0, tr_fill1 = 0, tr_walk0 = 0, tr_walk1 = 0;
        std::vector<JobView>& views = Ln.views; std::vector<WalkArgs>& wargs = Ln.wargs; std::vector<size_t>& base = Ln.base;
        JobView* const d_views = Ln.d_views; WalkArgs* const d_wargs = Ln.d_wargs;
        uint32_t G = Ln.G, slots_cap = Ln.slots_cap, kind = Ln.kind; int waves = Ln.waves;
        hipEvent_t* const ev = c.evp[Ln.slot];
        hipStream_t const sA = (overlap_fills && Ln.slot == 1) ? c.stream3 : c.stream;
        // The kernels whose workgroups wait for each other (several workgroups per read) need every workgroup of the launch resident
        // at once.  The grid is sized for that, but another process on the device, CU masking or reserved CUs can break it: the
        // kernels then give up after a bounded wait (error word 1) and the launch is run ONCE more on the streaming kernel with one
        // workgroup per read (nothing waits across workgroups there), or on the generic kernel if a read's tiles exceed one slot table.
        for (int attempt = 0;; ++attempt) {
        if (attempt > 0) {      // (the first attempt was launched by start())
            if (kind == 1u) launch_fill_local16(d_views, nj, G, waves, slots_cap, sh, sA);
            else launch_fill(d_views, nj, waves, sh, sA);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(ev[1], sA));
        }
        HIP_TRY(hipStreamWaitEvent(sB, ev[1], 0));      // fix-up + walk of this launch: behind its fill, beside the next one
        if (c.knobs.debug) { HIP_TRY(hipStreamSynchronize(sA)); fprintf(stderr, "[stitch] fill done (%u jobs, fast=%d)\n", nj, (int)fast); }
        if (c.knobs.fill_only) {        // experiment builds (garbage results): the fill's time is all that is wanted
            HIP_TRY(hipEventSynchronize(ev[1]));
            float ms_f = 0; HIP_TRY(hipEventElapsedTime(&ms_f, ev[0], ev[1])); c.tm.fill_ms += ms_f; c.tm.fill_kernel_ms += ms_f; c.tm.launches += 1; c.tm.jobs += nj;
            if (c.knobs.debug) fprintf(stderr, "[stitch] fill-only launch: %u jobs, %u workgroups per read, %d waves, fill %.1f ms\n", nj, G, waves, ms_f);
            break;
        }
        if (!c.knobs.dump_dir.empty()) {
            // debugging aid: what the fill kernel hands to the fix-up kernel (rows of the active contigs in order, roff-indexed), so
            // that two kernels can be compared array by array on the same job
            HIP_TRY(hipEventSynchronize(ev[1]));
            static int dump_no = 0;
            for (uint32_t q = 0; q < nj; ++q) {
                const JobLayout& L = lay[k0 + q]; const uint8_t* B = c.arena + base[q];
                const std::string path = c.knobs.dump_dir + "/fill_" + std::to_string(dump_no++) + "_kind" + std::to_string(kind) + ".bin";
                FILE* f = fopen(path.c_str(), "wb");
                if (!f) continue;
                const uint32_t hdr[4] = {L.n, L.nact, L.Rj, c.C};
                fwrite(hdr, 4, 4, f);
                std::vector<uint8_t> buf(4ull * L.Rj);
                for (size_t off : {L.off_S, L.off_Slen, L.off_Ival, L.off_Ilen, L.off_Sn, L.off_SnLen, L.off_Ly}) { HIP_TRY(hipMemcpy(buf.data(), B + off, 4ull * L.Rj, hipMemcpyDeviceToHost)); fwrite(buf.data(), 1, buf.size(), f); }
                std::vector<uint8_t> big(4ull * c.C * (L.n + 1));
                for (size_t off : {L.off_Lx, L.off_jti, L.off_jtf}) { HIP_TRY(hipMemcpy(big.data(), B + off, big.size(), hipMemcpyDeviceToHost)); fwrite(big.data(), 1, big.size(), f); }
                std::vector<ContigDesc> cds(c.C); HIP_TRY(hipMemcpy(cds.data(), views[q].cd, sizeof(ContigDesc) * c.C, hipMemcpyDeviceToHost));
                for (const ContigDesc& d : cds) { const uint32_t v[2] = {d.m, d.roff}; fwrite(v, 4, 2, f); }
                fclose(f);
            }
        }
        uint32_t max_nact_mode1 = 0;
        for (uint32_t q = 0; q < nj; ++q) if (jobs[k0 + q].mode == 1) max_nact_mode1 = std::max(max_nact_mode1, lay[k0 + q].nact);
        HIP_TRY(hipEventRecord(ev[2], sB));
        { uint32_t max_nact = 0; for (uint32_t q = 0; q < nj; ++q) max_nact = std::max(max_nact, lay[k0 + q].nact); launch_fixup_walk(d_views, d_wargs, nj, max_nact_mode1, sB, 0, max_nact); }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev[3], sB));
        HIP_TRY(hipStreamSynchronize(sB));
        float ms = 0;
        float ms_fill = 0, t_start = 0, t_end = 0;
        HIP_TRY(hipEventElapsedTime(&ms_fill, ev[0], ev[1])); c.tm.fill_kernel_ms += ms_fill;      // this kernel's own duration (what a profiler lists)
        // fill_ms = the time during which a fill kernel was running: two fills in flight are not counted twice
        HIP_TRY(hipEventElapsedTime(&t_start, c.ev[0], ev[0])); HIP_TRY(hipEventElapsedTime(&t_end, c.ev[0], ev[1]));
        { const double a = std::max((double)t_start, covered_until); if ((double)t_end > a) c.tm.fill_ms += (double)t_end - a; covered_until = std::max(covered_until, (double)t_end); }
        HIP_TRY(hipEventElapsedTime(&ms, ev[2], ev[3])); c.tm.walk_ms += ms;
        if (c.knobs.trace) { h_walked = host_ms(); tr_fill0 = t_start; tr_fill1 = t_end; HIP_TRY(hipEventElapsedTime(&tr_walk0, c.ev[0], ev[2])); HIP_TRY(hipEventElapsedTime(&tr_walk1, c.ev[0], ev[3])); }
        if (c.knobs.debug) fprintf(stderr, "[stitch] launch: %u jobs, %u workgroups per read, %d waves, fill %.1f ms, fix-up + walk %.1f ms\n", nj, G, waves, ms_fill, ms);
        if (attempt == 0) { c.tm.launches += 1; c.tm.jobs += nj; }
        if (c.knobs.profile_dump && (kind == 2u || kind == 3u)) {
            // fill_regs.hip / fill_regs32.hip, -DSTITCH_PROFILE: cycle sums per section over the waves of a read
            static const char* nm2[8] = {"poll", "jump", "pass1", "pass1b", "scan", "pass2+tail", "epilogue(last: +loop end)", "loop-top"};
            static const char* nm3[8] = {"row0+bases", "poll", "jump-select", "pass1", "pass1b", "scan+pass2+tail", "epilogue", "loop-top"};
            const char* const* nm = kind == 2u ? nm2 : nm3;
            for (uint32_t q = 0; q < std::min(nj, 4u); ++q) {
                unsigned long long pf[17] = {0}; HIP_TRY(hipMemcpy(pf, (const uint8_t*)views[q].err + 16, kind == 2u ? sizeof(pf) : 11 * sizeof(pf[0]), hipMemcpyDeviceToHost));
                if (kind == 2u) fprintf(stderr, "[prof] read %u: columns in which an insertion can matter: %llu wave-columns, %.0f ticks each (poll %.0f); others: %llu, %.0f ticks each (poll %.0f)\n", q,
                                        pf[11], (double)pf[12] / (double)std::max<unsigned long long>(pf[11], 1), (double)pf[13] / (double)std::max<unsigned long long>(pf[11], 1),
                                        pf[14], (double)pf[15] / (double)std::max<unsigned long long>(pf[14], 1), (double)pf[16] / (double)std::max<unsigned long long>(pf[14], 1));
                const double cols = (double)std::max<unsigned long long>(pf[8], 1) * (double)views[q].n;
                fprintf(stderr, "[prof] read %u: %llu waves;", q, pf[8]);
                for (int k = 0; k < 8; ++k) fprintf(stderr, " %s=%.0f", nm[k], (double)pf[k] / cols);
                fprintf(stderr, " (counter ticks per column and wave); groups with a merge in pass 1b %.2f, groups visited by pass 2 %.2f per column and wave\n", (double)pf[9] / cols, (double)pf[10] / cols);
            }
        }
        if (c.knobs.profile_dump && fast && kind != 2u) {
            unsigned long long pf[128]; HIP_TRY(hipMemcpy(pf, (const uint8_t*)views[0].err + 16, sizeof(pf), hipMemcpyDeviceToHost));
            unsigned long long t_first = ~0ull;
            std::vector<unsigned long long> t_end(nj);
            for (uint32_t q = 0; q < nj; ++q) { unsigned long long tm2[3]; HIP_TRY(hipMemcpy(tm2, (const uint8_t*)views[q].err + 16 + 120 * 8, 24, hipMemcpyDeviceToHost)); t_end[q] = tm2[2]; t_first = std::min(t_first, tm2[2]);
                fprintf(stderr, "[prof] job %u: tiles=%llu merged=%llu (%.1f%%)\n", q, tm2[0], tm2[1], 100.0 * tm2[1] / (tm2[0] ? tm2[0] : 1)); }
            for (uint32_t q = 0; q < nj; ++q) {
                unsigned long long w[64]; HIP_TRY(hipMemcpy(w, (const uint8_t*)views[q].err + 16 + 128 * 8, 16 * std::min(G, 32u), hipMemcpyDeviceToHost));
                fprintf(stderr, "[prof] place job %u:", q);
                for (uint32_t p = 0; p < std::min(G, 32u); ++p) fprintf(stderr, " [end %.1f xcc %u se %u sh %u cu %u]", (double)(w[2 * p] - t_first) / 1e5, (unsigned)(w[2 * p + 1] >> 32) & 15u,
                                                                         (unsigned)(w[2 * p + 1] >> 13) & 7u, (unsigned)(w[2 * p + 1] >> 12) & 1u, (unsigned)(w[2 * p + 1] >> 8) & 15u);
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "[prof] end of each read after the first to end, ms:");
            for (uint32_t q = 0; q < nj; ++q) fprintf(stderr, " %.1f", (double)(t_end[q] - t_first) / 1e5);
            fprintf(stderr, "\n");
            static const char* nm[8] = {"gather/loop", "select", "barrier1", "slot-setup", "tile", "finalize", "tile_wait", "barrier2"};
            for (int w = 0; w < waves; ++w) { fprintf(stderr, "[prof] wave %d:", w); for (int k = 0; k < 8; ++k) { if (k == 2) fprintf(stderr, " tiles=%llu merged=%llu", pf[w * 8 + 2] >> 32, pf[w * 8 + 2] & 0xFFFFFFFFull); else if (k == 1) fprintf(stderr, " simd=%u slot=%u cu=%u", (unsigned)((pf[w * 8 + 1] >> 4) & 3), (unsigned)(pf[w * 8 + 1] & 15), (unsigned)((pf[w * 8 + 1] >> 8) & 15)); else fprintf(stderr, " %s=%.1fM", nm[k], pf[w * 8 + k] / 1e6); } fprintf(stderr, "\n"); }
        }
        if (kind == 2u) {      // the shader clock the fill got (fill_regs.hip leaves it behind the first read's error word)
            // (copies on sB, never on the null stream: a blocking copy there would wait for the NEXT launch's fill on sA)
            if (!c.pin) { HIP_TRY(hipHostMalloc((void**)&c.pin, PIN_BYTES, hipHostMallocDefault)); }
            HIP_TRY(hipMemcpyAsync(c.pin, (const uint8_t*)views[0].err + ERR_CLOCK_OFF, 16, hipMemcpyDeviceToHost, sB));
            HIP_TRY(hipStreamSynchronize(sB));
            unsigned long long ck[2]; memcpy(ck, c.pin, 16);
            c.tm.clk_shader_cycles += ck[0]; c.tm.clk_ref_ticks += ck[1];
        }
        bool timed_out = false;
        if (kind != 0u) {
            if (!c.pin) { HIP_TRY(hipHostMalloc((void**)&c.pin, PIN_BYTES, hipHostMallocDefault)); }
            for (uint32_t q = 0; q < nj; ++q) HIP_TRY(hipMemcpyAsync(c.pin + 4ull * q, views[q].err, 4, hipMemcpyDeviceToHost, sB));
            HIP_TRY(hipStreamSynchronize(sB));
            for (uint32_t q = 0; q < nj; ++q) {
                uint32_t e = 0; memcpy(&e, c.pin + 4ull * q, 4);
                if ((e & 0xFFu) == 2u) return fail(STITCH_EINTERNAL, "fill kernel bounds check failed, code " + std::to_string(e >> 8));
                if (e) timed_out = true;
            }
        }
        if (attempt == 0 && c.knobs.fail_first_attempt && (kind >= 2u || G > 1)) timed_out = true;      // (test hook: exercises the relaunch)
        if (!timed_out) break;
        if (attempt > 0 || (kind == 1u && G == 1)) return fail(STITCH_EINTERNAL, "the fill kernel timed out waiting for a partner (workgroups of one read not co-resident, or a lost hand-off between waves)");
        {
            bool one_table = kind != 3u;      // (a launch of the 32-bit kernel holds reads no 16-bit kernel takes: the generic kernel repeats it)
            for (uint32_t q = 0; q < nj; ++q) if (one_table && (!local16_ok(c, jobs[k0 + q]) || local16_wg_tiles(c, jobs[k0 + q], 1) > fill_local16_max_slots())) one_table = false;
            kind = one_table ? 1u : 0u; G = 1; slots_cap = 0;
            // (the wave count is worked out afresh for the kernel that runs now: what the first attempt chose may exceed the generic
            // kernel's launch bounds)
            waves = 1;
            for (uint32_t q = 0; q < nj; ++q) {
                slots_cap = std::max(slots_cap, local16_wg_tiles(c, jobs[k0 + q], 1));
                views[q].tb_keyfmt = kind; waves = std::max(waves, pick_waves(c, lay[k0 + q].nact, MAX_WAVES_GENERIC));
            }
            HIP_TRY(hipMemsetAsync(Ln.d_x, 0, Ln.x_bytes, sA));
            waves = std::min(waves, MAX_WAVES_GENERIC);
            if (kind == 1u) waves = MAX_WAVES_LOCAL;
            HIP_TRY(hipMemcpyAsync(d_views, views.data(), sizeof(JobView) * nj, hipMemcpyHostToDevice, sA));
            c.tm.fill_kind = kind; c.tm.wg_per_read = 1; c.tm.fallbacks += 1;
            if (c.knobs.debug) fprintf(stderr, "[stitch] partner timeout: launch of %u jobs repeated with one workgroup per read (kernel kind %u)\n", nj, kind);
            HIP_TRY(hipEventRecord(ev[0], sA));
        }
        }


        if (c.knobs.fill_only) return STITCH_OK;
        {
            std::vector<uint8_t*> blocks(nj);
            for (uint32_t q = 0; q < nj; ++q) blocks[q] = c.arena + base[q];
            if (int e = download_chains(c, sB, jobs, lay, k0, nj, blocks, d_views, true)) return e;
        }
        if (c.knobs.trace) fprintf(stderr, "[trace] launch jobs %zu-%zu slot %d: host start %.1f submit %.1f finish-enter %.1f walked %.1f done %.1f | device fill %.1f-%.1f walk %.1f-%.1f (ms)\n",
                                   k0, k1, Ln.slot, Ln.h_start, Ln.h_submit, h_fin0, h_walked, host_ms(), tr_fill0, tr_fill1, tr_walk0, tr_walk1);
        return STITCH_OK;
    };
    // the launches, in order; with two windows the next fill is started before the last one is finished
    Launch ring[2]; int pend = -1, no = 0;
    for (size_t k0 = 0; k0 < jobs.size(); ++no) {
        const int slot = pipeline ? (no & 1) : 0;
        if (!pipeline && pend >= 0) { if (int e = finish(ring[pend])) return e; pend = -1; }
        ring[slot] = Launch();
        if (int e = start(k0, slot, ring[slot])) return e;
        if (pend >= 0) { if (int e = finish(ring[pend])) return e; }
        pend = slot; k0 = ring[slot].k1;
    }
    if (pend >= 0) { if (int e = finish(ring[pend])) return e; }
    if (c.tm.fill_kind == 2u) c.warmed_up = true;
    return STITCH_OK;
}

// The pre-alignment filter of Aligners::align (mod.rs:246-295): banded local score of every (read, target strand) pair on the
// device, then the reference's keep / early-break / subset logic on the host.  Sets jobs[k].act and the xs score.

// the reference's keep / early-break / subset rules on the banded scores of every (read, contig-strand) pair (mod.rs:246-295)
static void prealign_apply_rules(stitch_ctx& c, std::vector<Job>& jobs, const std::vector<int32_t>& sco, std::vector<uint8_t>& has, std::vector<int32_t>& score) {
    const uint32_t C = c.C, T = c.T; const size_t NJ = jobs.size();
    for (size_t q = 0; q < NJ; ++q) {
        Job& jb = jobs[q];
        std::vector<uint32_t> kept; int32_t best = 0; bool any = false;
        for (uint32_t t = 0; t < T; ++t) {                                     // targets in order, forward then reverse complement (:249-279)
            const int32_t f = sco[q * C + t];
            if (f >= c.opts.pre_align_min_score) { kept.push_back(t); best = any ? std::max(best, f) : f; any = true; }
            if (c.opts.double_strand) {
                const int32_t r = sco[q * C + T + t];
                if (r >= c.opts.pre_align_min_score) { kept.push_back(T + t); best = any ? std::max(best, r) : r; any = true; }
            }
            if (!c.opts.pre_align_subset_contigs && any) break;                // :276-278
        }
        has[q] = any ? 1 : 0; score[q] = best;
        if (any && c.opts.pre_align_subset_contigs) { std::sort(kept.begin(), kept.end()); jb.act = kept; }
    }
}

// The filter where the fast path does not apply: a clipping mode other than Local (the reference's banded scorer takes the mode's
// clip penalties, Options::banded_scoring, mod.rs:133-141) or a read beyond 65 534 bases (the fast path's 16-bit band ranges).  Seeds,
// backbone and band on host threads as there, the band as 32-bit ranges, every pair — also those without a seed, whose band is the whole
// matrix — through banded_score_general_kernel (prealign_kernel.hip), a chunk of reads at a time.  Plain and unhurried: cfg3 is Local.
static int run_prealign_general(stitch_ctx& c, std::vector<Job>& jobs, std::vector<uint8_t>& has, std::vector<int32_t>& score) {
    HIP_TRY(hipSetDevice(c.device));
    hipStream_t const PS = c.pstream[0];
    const uint32_t C = c.C;
    const size_t NJ = jobs.size(), NP = NJ * C;
    int32_t xp = 0, xs = 0, yp = 0, ys = 0;                                    // Options::clipping (mod.rs:123-131) in bio's argument order: x = the read
    if (c.opts.mode == 1 || c.opts.mode == 3) { xp = MIN_SCORE; xs = MIN_SCORE; }
    if (c.opts.mode == 2 || c.opts.mode == 3) { yp = MIN_SCORE; ys = MIN_SCORE; }
    const BandScoringClip sc{c.opts.match_score, c.opts.mismatch_score, c.opts.gap_open, c.opts.gap_extend, xp, xs, yp, ys};
    has.assign(NJ, 0); score.assign(NJ, 0);
    if (jobs.empty()) return STITCH_OK;
    auto al256 = [](size_t v) { return (v + 255) / 256 * 256; };
    std::vector<int32_t> sco(NP, 0);
    auto t_dev0 = std::chrono::steady_clock::now();
    size_t per_read_bands = 0; for (uint32_t a = 0; a < C; ++a) per_read_bands += 2ull * (c.al[a].m + 1);
    for (size_t k0 = 0; k0 < NJ;) {
        // as many reads as the scratch holds: bases, pairs, scores, bands, three int32 of state per row and pair
        size_t k1 = k0, bytes = 4096;
        while (k1 < NJ) {
            const size_t m = jobs[k1].y.size();
            const size_t need = al256(m + 8) + (size_t)C * (sizeof(BandPair32) + 4) + 4ull * per_read_bands + (size_t)C * 12ull * (m + 1) + 1024;
            if (bytes + need > c.pre_bytes) break;
            bytes += need; ++k1;
        }
        if (k1 == k0) return fail(STITCH_ENOMEM, "pre_align: one read does not fit in the pre-alignment scratch (STITCH_PREALIGN_BYTES)");
        const size_t nj = k1 - k0, np = nj * C;
        std::vector<uint8_t> h_reads; std::vector<uint64_t> q_at(nj);
        for (size_t q = 0; q < nj; ++q) { q_at[q] = h_reads.size(); h_reads.insert(h_reads.end(), jobs[k0 + q].y.begin(), jobs[k0 + q].y.end()); }
        h_reads.resize(h_reads.size() + 8, 0);
        std::vector<BandPair32> pairs(np);
        std::vector<uint32_t> bands(nj * per_read_bands);
        size_t state_elems = 0;
        std::vector<uint64_t> a_off(C + 1, 0);
        for (uint32_t a = 0; a < C; ++a) a_off[a + 1] = a_off[a] + 2ull * (c.al[a].m + 1);
        for (size_t q = 0; q < nj; ++q) for (uint32_t a = 0; a < C; ++a) {
            BandPair32& P = pairs[q * C + a]; const uint32_t m = (uint32_t)jobs[k0 + q].y.size();
            P.m = m; P.n = c.al[a].m; P.q_off = q_at[q]; P.t_off = c.al[a].seqoff; P.band_off = q * per_read_bands + a_off[a]; P.state_off = state_elems;
            state_elems += 3ull * (m + 1);
        }
        {
            auto t_h0 = std::chrono::steady_clock::now();
            const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>({nj, (size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)16}));
            std::atomic<size_t> next{0}; std::atomic<bool> failed{false};
            auto work = [&]() {
                try {
                    std::vector<uint32_t> lo_, hi_, chain_; std::vector<std::vector<Seed>> seeds_;
                    for (;;) {
                        const size_t q = next.fetch_add(1); if (q >= nj) break;
                        const Job& jb = jobs[k0 + q]; const uint32_t m = (uint32_t)jb.y.size();
                        find_seeds(c.kidx, c.h_xseq.data(), c.strands, jb.y.data(), m, seeds_);
                        for (uint32_t a = 0; a < C; ++a) {
                            backbone_chain(seeds_[a], (uint32_t)c.opts.kmer_size, c.opts.match_score, c.opts.gap_open, c.opts.gap_extend, chain_);      // (empty: the whole matrix)
                            rasterise_band32(seeds_[a], chain_, m, c.al[a].m, (uint32_t)c.opts.kmer_size, (uint32_t)c.opts.band_width, lo_, hi_);
                            const BandPair32& P = pairs[q * C + a];
                            memcpy(bands.data() + P.band_off, lo_.data(), 4ull * (P.n + 1)); memcpy(bands.data() + P.band_off + P.n + 1, hi_.data(), 4ull * (P.n + 1));
                        }
                    }
                } catch (...) { failed.store(true); next.store(nj); }
            };
            struct Joiner { std::vector<std::thread> pool; ~Joiner() { for (auto& th : pool) if (th.joinable()) th.join(); } } J;
            for (unsigned t = 1; t < nt; ++t) J.pool.emplace_back(work);
            work();
            for (auto& th : J.pool) th.join();
            if (failed.load()) return fail(STITCH_ENOMEM, "pre_align: out of host memory while building the bands");
            c.tm.prealign_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_h0).count();
        }
        uint8_t* p = c.pre_buf;
        uint8_t* d_reads = p; p += al256(h_reads.size());
        BandPair32* d_pairs = (BandPair32*)p; p += al256(np * sizeof(BandPair32));
        int32_t* d_scores = (int32_t*)p; p += al256(np * 4);
        uint32_t* d_bands = (uint32_t*)p; p += al256(bands.size() * 4);
        int32_t* d_state = (int32_t*)p; p += al256(state_elems * 4);
        if ((size_t)(p - c.pre_buf) > c.pre_bytes) return fail(STITCH_EINTERNAL, "pre-alignment scratch overflow");
        HIP_TRY(hipMemcpyAsync(d_reads, h_reads.data(), h_reads.size(), hipMemcpyHostToDevice, PS));
        HIP_TRY(hipMemcpyAsync(d_pairs, pairs.data(), np * sizeof(BandPair32), hipMemcpyHostToDevice, PS));
        HIP_TRY(hipMemcpyAsync(d_bands, bands.data(), bands.size() * 4, hipMemcpyHostToDevice, PS));
        launch_banded_scores_general(d_pairs, (uint32_t)np, sc, d_reads, c.d_xseq, d_bands, d_state, d_scores, PS);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(sco.data() + k0 * C, d_scores, np * 4, hipMemcpyDeviceToHost, PS));
        HIP_TRY(hipStreamSynchronize(PS));
        k0 = k1;
    }
    c.tm.prealign_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dev0).count();
    prealign_apply_rules(c, jobs, sco, has, score);
    return STITCH_OK;
}

int run_prealign(stitch_ctx& c, std::vector<Job>& jobs, std::vector<uint8_t>& has, std::vector<int32_t>& score) {
    {   // (other clipping modes, reads beyond the 16-bit band ranges: the general path)
        bool general = c.opts.mode != 0;
        for (const Job& jb : jobs) if (jb.y.size() > 65534) general = true;
        if (general || c.knobs.prealign_general) return run_prealign_general(c, jobs, has, score);
    }
    HIP_TRY(hipSetDevice(c.device));
    // the filter's own streams: a call's later groups of reads are filtered while the jump DP of the earlier ones runs (stitch_align_batch)
    hipStream_t const PS0 = c.pstream[0], PS1 = c.pstream[1], PS2 = c.pstream[2];      // reads + full-matrix kernels; band and banded score kernels; uploads
    const uint32_t C = c.C, T = c.T;
    const BandScoring sc{c.opts.match_score, c.opts.mismatch_score, c.opts.gap_open, c.opts.gap_extend};
    has.assign(jobs.size(), 0); score.assign(jobs.size(), 0);
    if (jobs.empty()) return STITCH_OK;
    auto al256 = [](size_t v) { return (v + 255) / 256 * 256; };

    // Device scratch of a call: [reads of every job | pairs | banded scores | full-matrix pairs, ids, scores] for the whole call, then
    // two chunk regions (bands, id lists, the global-state kernel's state) that alternate.  A read's first base sits at an offset
    // congruent to 1 modulo 4, so that the word holding the bases of rows 4k .. 4k + 3 (row i compares base i - 1) is aligned.
    const size_t NJ = jobs.size(), NP = NJ * C;
    std::vector<uint64_t> q_at(NJ);
    std::vector<uint8_t> h_reads;
    uint32_t max_n = 0;
    for (uint32_t a = 0; a < C; ++a) max_n = std::max(max_n, c.al[a].m);
    for (size_t k = 0; k < NJ; ++k) {
        const size_t m = jobs[k].y.size();
        if (m > 65534) return fail(STITCH_EINVAL, "pre_align: reads longer than 65534 bases are not supported");
        h_reads.resize((h_reads.size() + 3) / 4 * 4 + 1, 0);
        q_at[k] = h_reads.size();
        h_reads.insert(h_reads.end(), jobs[k].y.begin(), jobs[k].y.end());
    }
    h_reads.resize(h_reads.size() + 8, 0);
    uint8_t* p0 = c.pre_buf;
    uint8_t* d_reads = p0; p0 += al256(h_reads.size());
    BandPair* d_pairs = (BandPair*)p0; p0 += al256(NP * sizeof(BandPair));
    int32_t* d_scores = (int32_t*)p0; p0 += al256(NP * 4);
    BandPair* d_fpairs = (BandPair*)p0; p0 += al256(NP * sizeof(BandPair));
    uint32_t* d_fids = (uint32_t*)p0; p0 += al256(NP * 4);
    int32_t* d_fscores = (int32_t*)p0; p0 += al256(NP * 4);
    uint32_t* d_cls = (uint32_t*)p0; p0 += al256(NP * 4);
    uint32_t* d_cnt = (uint32_t*)p0; p0 += 256;                        // class counts of the two chunks in flight (4 words each)
    if (!c.pin_cnt) HIP_TRY(hipHostMalloc((void**)&c.pin_cnt, 64, hipHostMallocDefault));
    // the bands: drawn and classified on the device from the backbone's pieces (prealign_band.hip), or, for targets beyond that
    // kernel's LDS and on request, on the host
    const bool dev_bands = !c.knobs.prealign_v1 && !c.knobs.host_bands && !c.knobs.banded_global && max_n + 1 <= band_device_max_cols();
    const bool win_scoring = window_scoring_ok(sc, 65535);
    const size_t call_bytes = (size_t)(p0 - c.pre_buf);
    if (call_bytes + 8192 > c.pre_bytes) return fail(STITCH_ENOMEM, "pre_align: the batch does not fit in the pre-alignment scratch (STITCH_PREALIGN_BYTES)");
    const size_t region_bytes = (c.pre_bytes - call_bytes) / 2 / 256 * 256;

    // chunks of reads: as many as a chunk region holds, and at most PRE_CHUNK, so that a batch makes several chunks and the host
    // stage of one (seeds, backbone, band: threads) overlaps the device stage of the ones before
    constexpr size_t PRE_CHUNK = 64;
    std::vector<std::pair<size_t, size_t>> chunks;
    for (size_t k0 = 0; k0 < NJ;) {
        size_t k1 = k0, bytes = 0;
        while (k1 < NJ && k1 - k0 < PRE_CHUNK) {
            const size_t m = jobs[k1].y.size();
            size_t need = 0;
            // (a band's pieces: a diagonal and a gap per run of the backbone, runs at least k apart unless seeds overlap — a noisy multi-kb read
            // on its true target has hundreds; what a chunk really needs is checked against its region when its pieces are known)
            for (uint32_t a = 0; a < C; ++a) need += al256(4ull * (c.al[a].m + 1)) + al256(12ull * (m + 1)) + 16 + (2 * (std::min<size_t>(m, c.al[a].m) / (size_t)std::max(1, (int)c.opts.kmer_size)) + 4) * sizeof(BandElem);
            if (bytes + need + 4096 > region_bytes) break;
            bytes += need; ++k1;
        }
        if (k1 == k0) return fail(STITCH_ENOMEM, "pre_align: one read does not fit in the pre-alignment scratch (STITCH_PREALIGN_BYTES)");
        chunks.push_back({k0, k1}); k0 = k1;
    }

    // Band layout of a chunk, fixed before any band is known so that the host threads write straight into the pinned
    // staging buffer: pair (q, a) owns 2 (n_a + 1) uint16 at q * per_read + a_off[a] (a pair scored over the full matrix
    // leaves its slot unused), and 3 (m_q + 1) int32 of device state at state_at[q] + a * 3 (m_q + 1).
    std::vector<uint64_t> a_off(C + 1, 0);
    for (uint32_t a = 0; a < C; ++a) a_off[a + 1] = a_off[a] + 2ull * (c.al[a].m + 1);
    const uint64_t per_read = a_off[C];
    {
        size_t most = 0; for (auto& ch : chunks) most = std::max(most, ch.second - ch.first);
        const size_t want = dev_bands ? 0 : most * per_read;
        if (want > c.pin_bands_elems) {
            for (auto*& b : c.pin_bands) { if (b) { (void)hipHostFree(b); b = nullptr; } }
            c.pin_bands_elems = 0;
            for (auto*& b : c.pin_bands) HIP_TRY(hipHostMalloc((void**)&b, want * sizeof(uint16_t), hipHostMallocDefault));
            c.pin_bands_elems = want;
        }
    }
    struct Staged {                                   // what the host stage hands to the device stage
        size_t k0 = 0, k1 = 0;
        const uint16_t* bands = nullptr; size_t band_elems = 0; std::vector<BandPair> pairs; std::vector<BandElem> elems;      // (elems: the bands' pieces, when the device draws them)
        std::vector<uint32_t> full_ids, win_ids, banded_ids, tall_ids; uint32_t banded_max_m = 0; size_t state_elems = 0;   // ids = pair index in the CALL; win = register-window kernel, banded = LDS-ring kernel, tall = global-state kernel
        double host_ms = 0;
    };
    auto host_stage = [&](size_t k0, size_t k1, uint16_t* bands, Staged& S) {
        auto t_h0 = std::chrono::steady_clock::now();
        const size_t nj = k1 - k0, np = nj * C;
        S.k0 = k0; S.k1 = k1; S.pairs.resize(np); S.bands = bands; S.band_elems = nj * per_read;
        std::vector<uint64_t> state_at(nj);
        for (size_t q = 0; q < nj; ++q) { state_at[q] = S.state_elems; S.state_elems += (size_t)C * 3ull * (jobs[k0 + q].y.size() + 1); }
        std::vector<uint8_t> full(np, 0);                     // 1 = full matrix, 2 = a band column taller than the LDS ring, 3 = fits the register window
        const bool win_ok = !c.knobs.prealign_v1 && !c.knobs.banded_global && win_scoring;
        std::vector<std::vector<BandElem>> elems_of(dev_bands ? nj : 0);
        const uint32_t ring = banded_ring_rows();
        // seeds, backbone and band of every pair: independent per read, so the reads are dealt to host threads
        {
            const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>({nj, (size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)16}));
            std::atomic<size_t> next{0};
            std::atomic<bool> failed{false};
            auto work = [&]() {
                try {
                std::vector<uint16_t> lo_, hi_; std::vector<std::vector<Seed>> seeds_; std::vector<uint32_t> chain_; std::vector<BandElem> el_;
                for (;;) {
                    const size_t q = next.fetch_add(1); if (q >= nj) break;
                    const Job& jb = jobs[k0 + q];
                    const uint32_t m = (uint32_t)jb.y.size();
                    find_seeds(c.kidx, c.h_xseq.data(), c.strands, jb.y.data(), m, seeds_);
                    for (uint32_t a = 0; a < C; ++a) {
                        const Aligner& A = c.al[a];
                        BandPair& P = S.pairs[q * C + a];
                        P.m = m; P.n = A.m; P.q_off = q_at[k0 + q]; P.t_off = A.seqoff; P.band_off = 0; P.state_off = 0;
                        P.elem_off = 0; P.n_elem = 0;
                        const bool is_full = backbone_chain(seeds_[a], (uint32_t)c.opts.kmer_size, c.opts.match_score, c.opts.gap_open, c.opts.gap_extend, chain_);
                        if (is_full && m <= full_score_max_rows()) { full[q * C + a] = 1; continue; }                   // full-matrix kernels: no band needed
                        P.band_off = q * per_read + a_off[a];
                        P.state_off = state_at[q] + (uint64_t)a * 3ull * (m + 1);
                        if (dev_bands) {                                                // the backbone's pieces; the device draws the band and picks the kernel
                            band_elements(seeds_[a], chain_, m, A.m, (uint32_t)c.opts.kmer_size, el_);      // (none: the full matrix of a read too long for the full-matrix kernels)
                            P.elem_off = (uint32_t)elems_of[q].size(); P.n_elem = (uint32_t)el_.size();
                            elems_of[q].insert(elems_of[q].end(), el_.begin(), el_.end());
                            continue;
                        }
                        rasterise_band(seeds_[a], chain_, m, A.m, (uint32_t)c.opts.kmer_size, (uint32_t)c.opts.band_width, lo_, hi_);
                        if (win_ok && band_fits_window(lo_.data(), hi_.data(), m, A.m)) full[q * C + a] = 3;
                        else for (uint32_t col = 0; col <= A.m; ++col) if (hi_[col] > lo_[col] && (uint32_t)(hi_[col] - lo_[col]) > ring) { full[q * C + a] = 2; break; }
                        memcpy(bands + P.band_off, lo_.data(), sizeof(uint16_t) * (A.m + 1));
                        memcpy(bands + P.band_off + A.m + 1, hi_.data(), sizeof(uint16_t) * (A.m + 1));
                    }
                }
                } catch (...) { failed.store(true); next.store(nj); }      // (out of host memory in a worker: reported by the caller, never std::terminate)
            };
            struct Joiner { std::vector<std::thread> pool; ~Joiner() { for (auto& th : pool) if (th.joinable()) th.join(); } } J;
            for (unsigned t = 1; t < nt; ++t) J.pool.emplace_back(work);
            work();
            for (auto& th : J.pool) th.join();
            if (failed.load()) throw std::bad_alloc();
        }
        if (dev_bands) {
            for (size_t q = 0; q < nj; ++q) {
                const uint32_t at = (uint32_t)S.elems.size();
                for (uint32_t a = 0; a < C; ++a) S.pairs[q * C + a].elem_off += at;
                S.elems.insert(S.elems.end(), elems_of[q].begin(), elems_of[q].end());
            }
        }
        const uint32_t g0 = (uint32_t)(k0 * C);
        for (size_t k = 0; k < np; ++k) {
            if (full[k] == 1) S.full_ids.push_back(g0 + (uint32_t)k);
            else if (full[k] == 2) S.tall_ids.push_back(g0 + (uint32_t)k);
            else if (full[k] == 3) S.win_ids.push_back(g0 + (uint32_t)k);
            else { S.banded_ids.push_back(g0 + (uint32_t)k); S.banded_max_m = std::max(S.banded_max_m, S.pairs[k].m); }
        }
        S.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_h0).count();
    };

    // Full-matrix pairs collect over the chunks and go out in batches of thousands (one wavefront per pair: a chunk's few
    // hundred would leave most of the GPU idle), on the context's stream, beside the banded kernels on the second stream.
    constexpr size_t FULL_BATCH = 8192;
    std::vector<BandPair> f_pairs; f_pairs.reserve(NP);      // compact copies, in launch order (kept until the end: the uploads are asynchronous)
    std::vector<uint32_t> f_gid, f_ids;                               // pair index in the call; identity list for the kernels' `which`
    size_t f_sent = 0;
    auto flush_full = [&]() -> int {
        const size_t n = f_pairs.size() - f_sent;
        if (!n) return STITCH_OK;
        uint32_t mm = 0, mn = 0;
        for (size_t k = f_sent; k < f_pairs.size(); ++k) { mm = std::max(mm, f_pairs[k].m); mn = std::max(mn, f_pairs[k].n); }
        HIP_TRY(hipMemcpyAsync(d_fpairs + f_sent, f_pairs.data() + f_sent, n * sizeof(BandPair), hipMemcpyHostToDevice, PS0));
        HIP_TRY(hipMemcpyAsync(d_fids + f_sent, f_ids.data() + f_sent, n * 4, hipMemcpyHostToDevice, PS0));
        if (c.knobs.prealign_v1 || !launch_full_scores_skew16(d_fpairs, d_fids + f_sent, (uint32_t)n, mm, mn, sc, d_reads, c.d_xseq, d_fscores, PS0))
            launch_full_scores(d_fpairs, d_fids + f_sent, (uint32_t)n, mm, sc, d_reads, c.d_xseq, d_fscores, PS0);
        HIP_TRY(hipGetLastError());
        f_sent = f_pairs.size();
        return STITCH_OK;
    };
    f_ids.resize(NP); std::iota(f_ids.begin(), f_ids.end(), 0u);

    auto t_dev0 = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpyAsync(d_reads, h_reads.data(), h_reads.size(), hipMemcpyHostToDevice, PS0));
    HIP_TRY(hipEventRecord(c.ev2[0], PS0));
    HIP_TRY(hipStreamWaitEvent(PS1, c.ev2[0], 0));
    std::vector<Staged> staged(chunks.size());
    uint32_t cnt_seen[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    HIP_TRY(hipMemsetAsync(d_cnt, 0, 256, PS2));                     // the band kernel's class counts: cleared once per call
    auto device_stage = [&](size_t i) -> int {                       // asynchronous: everything of chunk i on the second stream, then its event
        Staged& S = staged[i];
        const size_t np = (S.k1 - S.k0) * C, g0 = S.k0 * C;
        uint8_t* p = p0 + (i & 1) * region_bytes;
        uint8_t* const p_end = p + region_bytes;
        uint16_t* d_bands = (uint16_t*)p; p += al256(S.band_elems * 2);
        uint32_t* d_banded = (uint32_t*)p; p += al256(S.banded_ids.size() * 4);
        uint32_t* d_tall = (uint32_t*)p; p += al256(S.tall_ids.size() * 4);
        uint32_t* d_win = (uint32_t*)p; p += al256(S.win_ids.size() * 4);
        BandElem* d_elems = (BandElem*)p; p += al256(S.elems.size() * sizeof(BandElem));
        int32_t* d_state = (int32_t*)p; p += al256(S.state_elems * 4);
        if (p > p_end) return fail(STITCH_ENOMEM, "pre_align: a chunk's bands have more pieces than its share of the pre-alignment scratch holds (STITCH_PREALIGN_BYTES)");
        // the uploads (128 MB of band ranges per 64 reads at cfg3) go on a stream of their own, beside the kernels of the chunk before
        hipStream_t up = PS2;
        if (!dev_bands) HIP_TRY(hipMemcpyAsync(d_bands, S.bands, S.band_elems * 2, hipMemcpyHostToDevice, up));
        else if (!S.elems.empty()) HIP_TRY(hipMemcpyAsync(d_elems, S.elems.data(), S.elems.size() * sizeof(BandElem), hipMemcpyHostToDevice, up));
        HIP_TRY(hipMemcpyAsync(d_pairs + g0, S.pairs.data(), np * sizeof(BandPair), hipMemcpyHostToDevice, up));
        if (!S.banded_ids.empty()) HIP_TRY(hipMemcpyAsync(d_banded, S.banded_ids.data(), S.banded_ids.size() * 4, hipMemcpyHostToDevice, up));
        if (!S.tall_ids.empty()) HIP_TRY(hipMemcpyAsync(d_tall, S.tall_ids.data(), S.tall_ids.size() * 4, hipMemcpyHostToDevice, up));
        if (!S.win_ids.empty()) HIP_TRY(hipMemcpyAsync(d_win, S.win_ids.data(), S.win_ids.size() * 4, hipMemcpyHostToDevice, up));
        if (dev_bands) {
            // every pair with a band is in `banded_ids`; the band kernel (on the upload stream: beside the score kernels of the chunk
            // before) draws the bands, names each pair's score kernel and counts the classes; the host waits for the counts and launches
            // only the score kernels that have pairs (an empty one still costs its launch and 5000 workgroups that come and go: 3 ms
            // of the 13 a chunk took), each of which skips the others' pairs
            const uint32_t nb = (uint32_t)S.banded_ids.size();
            // (the counts are CUMULATIVE over the call's chunks, cleared once per call: a 16-byte clear per chunk is a kernel of its own that
            // queues for a wave slot behind the score kernels' thousands of workgroups — 10-35 ms of the pipeline's latency each in
            // profiles/r04_a_kernel_stats_cfg3.csv's trace, `__amd_rocclr_fillBufferAligned` 6.5 % of the GPU time)
            uint32_t* const cnt_d = d_cnt + 4 * (i & 1); uint32_t* const cnt_h = c.pin_cnt + 4 * (i & 1);
            launch_band_draw(d_pairs, d_banded, nb, max_n, d_elems, (uint32_t)c.opts.band_width, banded_ring_rows(), win_scoring, d_bands, d_cls, cnt_d, up);
            HIP_TRY(hipMemcpyAsync(cnt_h, cnt_d, 16, hipMemcpyDeviceToHost, up));
            HIP_TRY(hipEventRecord(c.evu[i & 1], up));
            HIP_TRY(hipEventSynchronize(c.evu[i & 1]));
            HIP_TRY(hipStreamWaitEvent(PS1, c.evu[i & 1], 0));
            uint32_t cnt_now[4];
            for (int k4 = 0; k4 < 4; ++k4) { cnt_now[k4] = cnt_h[k4] - cnt_seen[i & 1][k4]; cnt_seen[i & 1][k4] = cnt_h[k4]; }
            if (cnt_now[BAND_CLASS_WINDOW]) launch_banded_scores_window(d_pairs, d_banded, nb, sc, d_reads, c.d_xseq, d_bands, d_scores, d_cls, PS1);
            if (cnt_now[BAND_CLASS_RING] && !launch_banded_scores_lds(d_pairs, d_banded, nb, S.banded_max_m, sc, d_reads, c.d_xseq, d_bands, d_scores, d_cls, PS1))
                launch_banded_scores(d_pairs, d_banded, nb, sc, d_reads, c.d_xseq, d_bands, d_state, d_scores, d_cls, BAND_CLASS_RING, PS1);
            if (cnt_now[BAND_CLASS_TALL]) launch_banded_scores(d_pairs, d_banded, nb, sc, d_reads, c.d_xseq, d_bands, d_state, d_scores, d_cls, BAND_CLASS_TALL, PS1);
        } else {
            HIP_TRY(hipEventRecord(c.evu[i & 1], up));
            HIP_TRY(hipStreamWaitEvent(PS1, c.evu[i & 1], 0));
            if (!S.win_ids.empty()) launch_banded_scores_window(d_pairs, d_win, (uint32_t)S.win_ids.size(), sc, d_reads, c.d_xseq, d_bands, d_scores, nullptr, PS1);
            if (c.knobs.banded_global || !launch_banded_scores_lds(d_pairs, d_banded, (uint32_t)S.banded_ids.size(), S.banded_max_m, sc, d_reads, c.d_xseq, d_bands, d_scores, nullptr, PS1))
                launch_banded_scores(d_pairs, d_banded, (uint32_t)S.banded_ids.size(), sc, d_reads, c.d_xseq, d_bands, d_state, d_scores, nullptr, 0, PS1);
            launch_banded_scores(d_pairs, d_tall, (uint32_t)S.tall_ids.size(), sc, d_reads, c.d_xseq, d_bands, d_state, d_scores, nullptr, 0, PS1);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c.evc[i & 1], PS1));
        for (uint32_t g : S.full_ids) { f_gid.push_back(g); f_pairs.push_back(S.pairs[g - g0]); }
        if (f_pairs.size() - f_sent >= FULL_BATCH) { const int rc = flush_full(); if (rc) return rc; }
        return STITCH_OK;
    };

    // pipeline: a producer thread stages chunk i + 1 on the host while chunk i runs on the device and chunk i - 1 finishes there;
    // a staging buffer and a chunk region are reused two chunks later, after the event of the chunk that used them
    // (the host stage reads jobs[].y only)
    std::vector<std::atomic<int>> ready(chunks.size());
    for (auto& r : ready) r.store(0);
    std::atomic<size_t> consumed{0};
    std::atomic<bool> stop{false};
    std::thread producer([&]() {
        for (size_t i = 0; i < chunks.size() && !stop.load(); ++i) {
            while (i > consumed.load() + 1 && !stop.load()) std::this_thread::sleep_for(std::chrono::microseconds(200));   // two staging buffers: chunk i reuses the one of chunk i - 2
            try { host_stage(chunks[i].first, chunks[i].second, c.pin_bands[i & 1], staged[i]); ready[i].store(1, std::memory_order_release); }
            catch (...) { ready[i].store(-1, std::memory_order_release); return; }              // out of host memory
        }
    });
    int rc = STITCH_OK;
    for (size_t i = 0; i < chunks.size(); ++i) {
        while (!ready[i].load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(100));
        if (ready[i].load() < 0) { rc = fail(STITCH_ENOMEM, "pre_align: out of host memory while building the bands"); break; }
        c.tm.prealign_host_ms += staged[i].host_ms;
        rc = device_stage(i);
        if (rc) break;
        if (i >= 1) {                                                    // chunk i - 1 is done: its staging buffer and region are free for chunk i + 1
            if (hipEventSynchronize(c.evc[(i - 1) & 1]) != hipSuccess) { rc = fail(STITCH_EINTERNAL, "pre_align: device stage failed"); break; }
            staged[i - 1].pairs = std::vector<BandPair>();
        }
        consumed.store(i);
    }
    stop.store(true);
    producer.join();
    if (!rc) rc = flush_full();
    const hipError_t e1 = hipStreamSynchronize(PS1), e2 = hipStreamSynchronize(PS0), e3 = hipStreamSynchronize(PS2);      // (also on the error paths: nothing of this call stays in flight)
    if (rc) return rc;
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(STITCH_EINTERNAL, "pre_align: device stage failed");
    std::vector<int32_t> sco(NP), fsco(f_pairs.size());
    HIP_TRY(hipMemcpyAsync(sco.data(), d_scores, NP * 4, hipMemcpyDeviceToHost, PS0));      // (not on the null stream: that would wait for the fills of other reads in flight)
    if (!fsco.empty()) HIP_TRY(hipMemcpyAsync(fsco.data(), d_fscores, fsco.size() * 4, hipMemcpyDeviceToHost, PS0));
    HIP_TRY(hipStreamSynchronize(PS0));
    for (size_t k = 0; k < f_gid.size(); ++k) sco[f_gid[k]] = fsco[k];
    c.tm.prealign_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dev0).count();

    prealign_apply_rules(c, jobs, sco, has, score);
    return STITCH_OK;
}

// traceback_all's selection loop over per-end-contig candidate chains (traceback/mod.rs:152-217)
std::vector<HAln> select_all(const stitch_ctx& c, Job& jb) {           // (moves the selected chains out of the job)
    std::vector<HAln> out;
    std::vector<uint8_t> consider(c.C, 0), seen(c.C, 0);
    for (uint32_t a : jb.act) consider[a] = 1;
    size_t n_consider = jb.act.size(), n_seen = 0, guard = 0;
    auto mark = [&](uint32_t a) { if (a < c.C && consider[a] && !seen[a]) { seen[a] = 1; ++n_seen; } };
    while (n_seen < n_consider) {
        if (++guard > 4 * n_consider + 16) break;
        size_t pick = 0; int32_t score = MIN_SCORE; uint32_t alen = 0;
        for (size_t k = 0; k < jb.act.size(); ++k) {
            if (seen[jb.act[k]]) continue;
            // S[n%2][m] and cell(m,n).S.len of aligner k: carried in its candidate chain
            int32_t s = jb.chains[k].score; uint32_t l = jb.chains[k].length;
            if (s > score || (s == score && l > alen)) { pick = k; score = s; alen = l; }
        }
        if (jb.status[pick] == 1) { mark(jb.act[pick]); continue; }
        HAln& a = jb.chains[pick];
        mark(a.start_contig_idx); mark(a.end_contig_idx);
        for (const stitch_op& op : a.ops) if (op.kind == OP_XJUMP) mark(op.contig);
        out.push_back(std::move(a));
    }
    return out;
}

struct RealignPlan {                                // realign_origin (mod.rs:442-553) for one chain
    bool needed = false;
    HAln original;
    std::vector<uint32_t> subset;
    struct Step { size_t job; uint32_t contig; uint32_t y_pivot; };
    std::vector<Step> steps;                        // in the reference's order
};

void plan_realign(const stitch_ctx& c, const std::vector<uint8_t>& query, const HAln& aln, std::vector<Job>& jobs, RealignPlan& plan) {
    const uint32_t slop = (uint32_t)c.opts.circular_slop;
    const bool circ = c.opts.circular != 0;          // is_circular(): every aligner carries opts.circular (mod.rs:191,201)
    bool at_start = aln.xstart <= slop && circ, at_end = aln.xlen <= aln.xend + slop && circ;      // :365-385
    uint32_t cs = aln.start_contig_idx, ce = aln.end_contig_idx;
    if (at_start && at_end && cs == ce) return;      // :389-395
    if (!at_start && !at_end) return;
    if (at_start && aln.yend == aln.ylen) at_start = false;                                          // :398-407
    if (at_end && aln.ystart == 0) at_end = false;
    if (!at_start && !at_end) return;
    plan.needed = true;
    plan.original = aln;
    std::vector<uint8_t> in(c.C, 0);
    in[aln.start_contig_idx] = 1; in[aln.end_contig_idx] = 1;
    for (const stitch_op& op : aln.ops) if (op.kind == OP_XJUMP) in[op.contig] = 1;
    for (uint32_t a = 0; a < c.C; ++a) if (in[a]) plan.subset.push_back(a);
    const uint32_t n = (uint32_t)query.size();
    auto add = [&](uint32_t pivot_y, uint32_t contig) {
        Job jb; jb.y.assign(query.begin() + pivot_y, query.end()); jb.y.insert(jb.y.end(), query.begin(), query.begin() + pivot_y);
        jb.act = plan.subset; jb.mode = 2; jb.from = contig;
        plan.steps.push_back({jobs.size(), contig, aln.ylen - pivot_y});
        jobs.push_back(std::move(jb));
    };
    (void)n;
    if (at_start) {                                   // :475-509
        uint32_t yend = aln.ystart;
        for (const stitch_op& op : aln.ops) { if (op.kind == OP_XJUMP && op.contig != cs) break; yend += op_len_y(op); }
        add(aln.yend, cs); add(yend, cs);
    }
    if (at_end) {                                     // :512-550
        uint32_t ystart = aln.ystart, ycur = aln.ystart, xidx = aln.start_contig_idx;
        for (const stitch_op& op : aln.ops) {
            if (op.kind == OP_XJUMP) { if (op.contig == ce && xidx != ce) ystart = ycur; xidx = op.contig; }
            ycur += op_len_y(op);
        }
        add(aln.ystart, ce); add(ystart, ce);
    }
}

HAln finish_realign(const stitch_ctx& c, const RealignPlan& plan, const std::vector<Job>& jobs) {
    HAln best = plan.original;
    for (const auto& st : plan.steps) {               // realign_and_split_at_y (mod.rs:412-431)
        const Job& jb = jobs[st.job];
        if (jb.status[0] == 1) continue;
        HAln na = jb.chains[0];
        if (na.score > best.score && na.start_contig_idx == st.contig && best.end_contig_idx == st.contig) {
            if (!c.opts.keep_clipping) remove_clipping(c.opts, na);
            best = split_at_y(na, 4, st.y_pivot);
        }
    }
    return best;
}

}  // namespace

extern "C" {

int stitch_align_batch(stitch_ctx* c, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                       const stitch_read_result** per_read, const stitch_chain** chains, const stitch_op** ops,
                       uint64_t* cells_filled) {
    if (!c || (!bases && n_reads) || !offsets) return fail(STITCH_EINVAL, "null argument");
    auto t_host0 = std::chrono::steady_clock::now();
    c->tm = stitch_timing{};
    c->rr.clear(); c->chains.clear(); c->ops.clear(); c->per_read.assign(n_reads, -1); c->job_chains.clear();
    std::vector<uint32_t> all(c->C); for (uint32_t a = 0; a < c->C; ++a) all[a] = a;

    // pass 1: one job per run of identical consecutive reads (FastxGroupingIterator, align/io.rs:118-146)
    std::vector<Job> jobs; std::vector<uint32_t> job_of(n_reads);
    for (uint32_t r = 0; r < n_reads; ++r) {
        const uint8_t* s = bases + offsets[r]; const size_t n = (size_t)(offsets[r + 1] - offsets[r]);
        if (n == 0) return fail(STITCH_EINVAL, "empty read");
        if (r > 0) {
            const size_t pn = (size_t)(offsets[r] - offsets[r - 1]);
            if (pn == n && memcmp(bases + offsets[r - 1], s, n) == 0) { job_of[r] = job_of[r - 1]; continue; }
        }
        Job jb; jb.y.assign(s, s + n);
        for (auto& b : jb.y) if (b >= 'a' && b <= 'z') b = (uint8_t)(b - 32);          // seq_upper_case (io.rs:64-66)
        jb.act = all; jb.mode = c->opts.suboptimal ? 1 : 0;
        job_of[r] = (uint32_t)jobs.size();
        jobs.push_back(std::move(jb));
    }
    int rc;
    // pre-alignment filter (mod.rs:246-287): reads without a passing target are not aligned at all; with
    // pre_align_subset_contigs the others are aligned to the passing contig-strands only
    const size_t n_jobs_all = jobs.size();
    std::vector<uint8_t> pre_has; std::vector<int32_t> pre_score; std::vector<size_t> live_of;
    const bool dbg = c->knobs.debug;
    auto stamp = [&](const char* what) { if (dbg) fprintf(stderr, "[stitch] %-28s at %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count()); };
    // (Measured dead end, round 3: filtering the next group of a call's reads on a second host thread and the filter's own streams while
    // the DP of the group before runs.  Both stages slow down by what the overlap would save and the smaller launches cost more: a
    // fill launch lasts its reads' 10 000 columns whatever their number, so 1024 reads in groups of 256 ran at 1599 reads/s against
    // 2152 in one piece, groups of 128 at 956.)
    if (c->opts.pre_align) {
        rc = run_prealign(*c, jobs, pre_has, pre_score); if (rc) return rc;
        std::vector<Job> live;
        for (size_t k = 0; k < jobs.size(); ++k) if (pre_has[k]) { live_of.push_back(k); live.push_back(std::move(jobs[k])); }
        jobs.swap(live);
    } else { live_of.resize(jobs.size()); for (size_t k = 0; k < jobs.size(); ++k) live_of[k] = k; }
    stamp("jobs built / pre-aligned");
    rc = run_jobs(*c, jobs);
    if (rc) return rc;
    stamp("pass 1 done");

    if (c->knobs.fill_only) {           // experiment: no chains exist
        c->rr.assign(n_reads, stitch_read_result{});
        if (per_read) *per_read = c->rr.data(); if (chains) *chains = c->chains.data(); if (ops) *ops = c->ops.data();
        if (cells_filled) *cells_filled = c->tm.cells;
        return STITCH_OK;
    }

    // host: chains of pass 1, realign planning
    struct PerJob { std::vector<HAln> chains; std::vector<RealignPlan> plans; };
    std::vector<PerJob> pj(jobs.size());
    std::vector<Job> jobs2;
    for (size_t k = 0; k < jobs.size(); ++k) {
        std::vector<HAln> cand;
        if (c->opts.suboptimal) cand = select_all(*c, jobs[k]);
        else { if (jobs[k].status[0] != 0) return fail(STITCH_EINTERNAL, "primary traceback returned None"); cand.push_back(std::move(jobs[k].chains[0])); }
        for (HAln& a : cand) { if (!c->opts.keep_clipping) remove_clipping(c->opts, a); }
        pj[k].chains = std::move(cand);
        pj[k].plans.resize(pj[k].chains.size());
        if (c->opts.circular) for (size_t q = 0; q < pj[k].chains.size(); ++q) plan_realign(*c, jobs[k].y, pj[k].chains[q], jobs2, pj[k].plans[q]);
    }
    stamp("chains selected, realign planned");
    if (!jobs2.empty()) { rc = run_jobs(*c, jobs2); if (rc) return rc; }
    stamp("pass 2 done");
    for (size_t k = 0; k < jobs.size(); ++k) {
        for (size_t q = 0; q < pj[k].chains.size(); ++q)
            if (pj[k].plans[q].needed) pj[k].chains[q] = finish_realign(*c, pj[k].plans[q], jobs2);
        std::vector<HAln>& al = pj[k].chains;
        if (c->opts.suboptimal && al.size() > 1) {                                        // mod.rs:318-329
            std::stable_sort(al.begin(), al.end(), [](const HAln& a, const HAln& b) { return -a.score < -b.score; });
            const float min_score = (float)al[0].score * c->opts.suboptimal_pct / 100.0f;
            std::vector<HAln> kept; for (HAln& a : al) if ((float)a.score >= min_score) kept.push_back(std::move(a));
            al.swap(kept);
        }
    }
    stamp("realign finished, sorted");
    // result arena, input order
    c->rr.resize(n_reads);
    std::vector<long> live_idx(n_jobs_all, -1);
    for (size_t l = 0; l < live_of.size(); ++l) live_idx[live_of[l]] = (long)l;
    static const std::vector<HAln> none;
    c->job_chains.resize(pj.size());
    for (size_t k = 0; k < pj.size(); ++k) c->job_chains[k] = std::move(pj[k].chains);
    { size_t n_ops = 0; for (uint32_t r = 0; r < n_reads; ++r) { const long l = live_idx[job_of[r]]; if (l >= 0) for (const HAln& a : c->job_chains[(size_t)l]) n_ops += a.ops.size(); }
      c->ops.reserve(n_ops); }
    for (uint32_t r = 0; r < n_reads; ++r) {
        const long l = live_idx[job_of[r]];
        const std::vector<HAln>& al = l >= 0 ? c->job_chains[(size_t)l] : none;
        stitch_read_result& R = c->rr[r]; memset(&R, 0, sizeof(R));
        if (c->opts.pre_align && pre_has[job_of[r]]) { R.has_prealign = 1; R.prealign_score = pre_score[job_of[r]]; }
        R.chains_begin = c->chains.size(); R.n_chains = (uint32_t)al.size();
        for (const HAln& a : al) {
            stitch_chain ch{}; ch.score = a.score; ch.xstart = a.xstart; ch.xend = a.xend; ch.ystart = a.ystart; ch.yend = a.yend; ch.xlen = a.xlen; ch.ylen = a.ylen;
            ch.start_contig_idx = a.start_contig_idx; ch.end_contig_idx = a.end_contig_idx; ch.length = a.length;
            ch.ops_begin = c->ops.size(); ch.ops_len = (uint32_t)a.ops.size();
            c->ops.insert(c->ops.end(), a.ops.begin(), a.ops.end());
            c->chains.push_back(ch);
        }
        c->per_read[r] = l;
    }
    if (per_read) *per_read = c->rr.data();
    if (chains) *chains = c->chains.data();
    if (ops) *ops = c->ops.data();
    if (cells_filled) *cells_filled = c->tm.cells;
    c->tm.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    stamp("result arena built");
    return STITCH_OK;
}

long stitch_format_sam(stitch_ctx* c, uint32_t read_idx, const char* head, const uint8_t* bases, const uint8_t* quals,
                       size_t n, char* buf, size_t cap) {
    if (!c || !head || !bases) return fail(STITCH_EINVAL, "null argument");
    if (read_idx >= c->per_read.size()) return fail(STITCH_EINVAL, "read_idx outside the last batch");
    static const std::vector<HAln> none_chains;
    std::vector<std::string> recs; std::string err;
    if (!format_sam_records(c->opts, c->targets, head, bases, quals, n, c->per_read[read_idx] >= 0 ? c->job_chains[(size_t)c->per_read[read_idx]] : none_chains, c->rr[read_idx].has_prealign != 0,
                            c->rr[read_idx].prealign_score, recs, err)) return fail(STITCH_EINVAL, err);
    std::string all;
    for (size_t k = 0; k < recs.size(); ++k) { if (k) all += "\n"; all += recs[k]; }
    if (buf && all.size() < cap) memcpy(buf, all.c_str(), all.size() + 1);
    return (long)all.size();
}

long stitch_format_sam_chains(const stitch_opts* o, const char* const* target_names, const uint32_t* target_lens, uint32_t n_targets, const char* head,
                             const uint8_t* bases, const uint8_t* quals, size_t n, const stitch_chain* chains, uint32_t n_chains, const stitch_op* ops,
                             int has_prealign, int32_t prealign, char* buf, size_t cap) {
    if (!o || !target_names || !target_lens || !head || !bases || (!chains && n_chains)) return fail(STITCH_EINVAL, "null argument");
    std::vector<TargetInfo> targets;
    for (uint32_t t = 0; t < n_targets; ++t) targets.push_back(TargetInfo{target_names[t], target_lens[t]});
    std::vector<HAln> al(n_chains);
    for (uint32_t k = 0; k < n_chains; ++k) {
        const stitch_chain& ch = chains[k]; HAln& a = al[k];
        a.score = ch.score; a.xstart = ch.xstart; a.xend = ch.xend; a.ystart = ch.ystart; a.yend = ch.yend; a.xlen = ch.xlen; a.ylen = ch.ylen;
        a.start_contig_idx = ch.start_contig_idx; a.end_contig_idx = ch.end_contig_idx; a.length = ch.length;
        if (ch.ops_len) a.ops.assign(ops + ch.ops_begin, ops + ch.ops_begin + ch.ops_len);
    }
    std::vector<std::string> recs; std::string err;
    if (!format_sam_records(*o, targets, head, bases, quals, n, al, has_prealign != 0, prealign, recs, err)) return fail(STITCH_EINVAL, err);
    std::string all;
    for (size_t k = 0; k < recs.size(); ++k) { if (k) all += "\n"; all += recs[k]; }
    if (buf && all.size() < cap) memcpy(buf, all.c_str(), all.size() + 1);
    return (long)all.size();
}

int stitch_prealign_band(const uint8_t* read, uint32_t read_len, const uint8_t* target, uint32_t target_len, uint32_t k, uint32_t w,
                         int32_t match, int32_t gap_open, int32_t gap_extend, uint16_t* lo, uint16_t* hi) {
    if (!read || !target || !lo || !hi) return fail(STITCH_EINVAL, "null argument");
    if (read_len > 65534) return fail(STITCH_EINVAL, "pre_align: reads longer than 65534 bases are not supported");
    // (the backbone's pruned search is exact only for a gap penalty that grows with the gap; the context constructor rejects the same)
    if (gap_open > 0 || gap_extend > 0) return fail(STITCH_EINVAL, "gap_open and gap_extend can't be positive");
    std::vector<Strand> st{Strand{0, target_len}};
    const KmerIndex ix = build_kmer_index(target, st, k);
    std::vector<std::vector<Seed>> seeds; std::vector<uint16_t> l, h;
    find_seeds(ix, target, st, read, read_len, seeds);
    const bool full = make_band(seeds[0], read_len, target_len, k, w, match, gap_open, gap_extend, l, h);
    memcpy(lo, l.data(), 2ull * (target_len + 1)); memcpy(hi, h.data(), 2ull * (target_len + 1));
    return full ? 1 : 0;
}

int stitch_prealign_band_device(int device, const uint8_t* read, uint32_t read_len, const uint8_t* target, uint32_t target_len, uint32_t k, uint32_t w,
                                int32_t match, int32_t gap_open, int32_t gap_extend, uint16_t* lo, uint16_t* hi, uint32_t* kernel_class) {
    if (!read || !target || !lo || !hi) return fail(STITCH_EINVAL, "null argument");
    if (read_len > 65534) return fail(STITCH_EINVAL, "pre_align: reads longer than 65534 bases are not supported");
    // (the backbone's pruned search is exact only for a gap penalty that grows with the gap; the context constructor rejects the same)
    if (gap_open > 0 || gap_extend > 0) return fail(STITCH_EINVAL, "gap_open and gap_extend can't be positive");
    if (target_len + 1 > band_device_max_cols()) return fail(STITCH_EINVAL, "the device draws bands of up to 8191 columns");
    std::vector<Strand> st{Strand{0, target_len}};
    const KmerIndex ix = build_kmer_index(target, st, k);
    std::vector<std::vector<Seed>> seeds; std::vector<uint32_t> chain; std::vector<BandElem> el;
    find_seeds(ix, target, st, read, read_len, seeds);
    const bool full = backbone_chain(seeds[0], k, match, gap_open, gap_extend, chain);
    band_elements(seeds[0], chain, read_len, target_len, k, el);
    HIP_TRY(hipSetDevice(device));
    BandPair P{}; P.m = read_len; P.n = target_len; P.band_off = 0; P.elem_off = 0; P.n_elem = (uint32_t)el.size();
    const uint32_t zero = 0;
    struct Bufs { BandPair* p = nullptr; uint32_t* w = nullptr; BandElem* e = nullptr; uint16_t* b = nullptr; uint32_t* c = nullptr;
                  ~Bufs() { (void)hipFree(p); (void)hipFree(w); (void)hipFree(e); (void)hipFree(b); (void)hipFree(c); } } B;
    HIP_TRY(hipMalloc((void**)&B.p, sizeof(P))); HIP_TRY(hipMalloc((void**)&B.w, 4)); HIP_TRY(hipMalloc((void**)&B.e, std::max<size_t>(1, el.size()) * sizeof(BandElem)));
    HIP_TRY(hipMalloc((void**)&B.b, 4ull * (target_len + 1))); HIP_TRY(hipMalloc((void**)&B.c, 4));
    HIP_TRY(hipMemcpy(B.p, &P, sizeof(P), hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(B.w, &zero, 4, hipMemcpyHostToDevice));
    if (!el.empty()) HIP_TRY(hipMemcpy(B.e, el.data(), el.size() * sizeof(BandElem), hipMemcpyHostToDevice));
    launch_band_draw(B.p, B.w, 1, target_len, B.e, w, banded_ring_rows(), true, B.b, B.c, nullptr, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(lo, B.b, 2ull * (target_len + 1), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hi, B.b + (target_len + 1), 2ull * (target_len + 1), hipMemcpyDeviceToHost));
    uint32_t cls = 0; HIP_TRY(hipMemcpy(&cls, B.c, 4, hipMemcpyDeviceToHost));
    if (kernel_class) *kernel_class = cls;
    return full ? 1 : 0;
}

int stitch_shard_range(const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads, uint32_t world, uint32_t rank, uint32_t* lo, uint32_t* hi) {
    if (!offsets || (!bases && n_reads) || !lo || !hi || world == 0 || rank >= world) return fail(STITCH_EINVAL, "bad argument");
    auto same = [&](uint32_t a, uint32_t b) {
        const uint64_t la = offsets[a + 1] - offsets[a], lb = offsets[b + 1] - offsets[b];
        return la == lb && memcmp(bases + offsets[a], bases + offsets[b], (size_t)la) == 0;
    };
    auto cut = [&](uint32_t r) -> uint32_t {          // first read of rank r's block
        if (r == 0) return 0;
        if (r >= world) return n_reads;
        uint32_t k = (uint32_t)((uint64_t)n_reads * r / world);
        while (k > 0 && k < n_reads && same(k, k - 1)) ++k;
        return k;
    };
    uint32_t prev = 0, a = 0, b = 0;
    for (uint32_t r = 0; r <= rank + 1; ++r) { const uint32_t c = std::max(cut(r), prev); if (r == rank) a = c; if (r == rank + 1) b = c; prev = c; }
    *lo = a; *hi = b;
    return STITCH_OK;
}

long stitch_split_at_y(const stitch_chain* in, const stitch_op* in_ops, int32_t mode, uint32_t y_pivot, stitch_chain* out, stitch_op* out_ops, uint32_t cap) {
    if (!in || !out || (!in_ops && in->ops_len) || (!out_ops && cap)) return fail(STITCH_EINVAL, "null argument");
    HAln a; a.score = in->score; a.xstart = in->xstart; a.xend = in->xend; a.ystart = in->ystart; a.yend = in->yend; a.xlen = in->xlen; a.ylen = in->ylen;
    a.start_contig_idx = in->start_contig_idx; a.end_contig_idx = in->end_contig_idx; a.length = in->length;
    a.ops.assign(in_ops, in_ops + in->ops_len);
    const HAln r = split_at_y(a, mode, y_pivot);
    *out = stitch_chain{}; out->score = r.score; out->xstart = r.xstart; out->xend = r.xend; out->ystart = r.ystart; out->yend = r.yend; out->xlen = r.xlen; out->ylen = r.ylen;
    out->start_contig_idx = r.start_contig_idx; out->end_contig_idx = r.end_contig_idx; out->length = r.length; out->ops_begin = 0; out->ops_len = (uint32_t)r.ops.size();
    for (size_t k = 0; k < r.ops.size() && k < cap; ++k) out_ops[k] = r.ops[k];
    return (long)r.ops.size();
}

int stitch_last_timing(const stitch_ctx* c, stitch_timing* out, size_t out_size) {
    if (!c || !out) return fail(STITCH_EINVAL, "null argument");
    // the struct only ever grows at its end: a caller built against an older header gets the fields it knows, never more bytes than it has
    std::memcpy(out, &c->tm, out_size < sizeof(stitch_timing) ? out_size : sizeof(stitch_timing));
    return STITCH_OK;
}

}  // extern "C"
