/* stitch_gpu.h — C ABI of the MI355X-native `stitch align` hot path.
 *
 * The reference (fulcrumgenomics/stitch, Rust) has no FFI; the seam this library replaces is
 *   Builder::build_aligners(&[TargetSeq]) -> Aligners            fg-stitch-lib/src/align/aligners/mod.rs:171-211
 *   Aligners::align(&mut self, record, target_seqs, target_hashes) -> (Vec<Alignment>, Option<i32>)      :237-340
 *   SamRecordFormatter::format(record, chains, pre_alignment_score) -> Vec<SamRecord>                     :622-973
 * called once per read group from each worker thread (fg-stitch-cli/src/commands/align.rs:356-375).
 * A GPU needs many reads per call, so the entry points are batch-oriented; everything else (names, argument
 * meaning, result layout, ordering) follows the reference.  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions: all integers little-endian; return 0 on success, a negative STITCH_E* otherwise (the reference's
 * panics become error codes); stitch_last_error() gives the thread-local message.  A stitch_ctx is NOT
 * thread-safe (same as `&mut Aligners`); use one ctx per device and host thread.  Output order == input order.
 */
#ifndef STITCH_GPU_H
#define STITCH_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STITCH_OK 0
#define STITCH_EINVAL (-1)      /* bad argument (positive penalties, empty FASTA, > 255 contig-strands, ...) */
#define STITCH_EDEVICE (-2)     /* HIP runtime error / no device */
#define STITCH_ENOMEM (-3)      /* a single read does not fit in device memory */
#define STITCH_EINTERNAL (-4)   /* traceback overflow or inconsistent state (a bug) */

/* == Options, aligners/mod.rs:65-116 (defaults are the CLI's, fg-stitch-cli/src/commands/align.rs:94-275) */
typedef struct stitch_opts {
  int32_t mode;                 /* 0 local, 1 query-local, 2 target-local, 3 global        constants.rs:96-136 */
  int32_t match_score, mismatch_score, gap_open, gap_extend;
  int32_t jump_same, jump_opposite, jump_inter;      /* already defaulted from --jump-score  mod.rs:143-152 */
  int32_t double_strand, circular, circular_slop;
  int32_t pre_align, pre_align_min_score, pre_align_subset_contigs, kmer_size, band_width;   /* mod.rs:246-295; local mode, DESIGN.md A15 */
  int32_t suboptimal; float suboptimal_pct;
  int32_t soft_clip, use_eq_and_x, pick_primary /* 0 query-length, 1 score */, filter_secondary;
  float filter_secondary_pct;
  int32_t keep_clipping;        /* test hook: 1 = skip Aligners::remove_clipping (mod.rs:343-353) so results can be
                                   compared with the reference's MultiContigAligner::custom tests */
} stitch_opts;

void stitch_opts_default(stitch_opts* o);   /* == Builder::default() */

typedef struct stitch_index stitch_index;   /* the reference index: == &[TargetSeq], util/target_seq.rs:15-36 */
typedef struct stitch_ctx stitch_ctx;       /* == Aligners + SamRecordFormatter for one device */

/* == target_seq::from_fasta minus file IO (util/target_seq.rs:69-123): names + sequences, upper-cased here. */
int stitch_index_build(const char* const* names, const uint8_t* const* seqs, const uint32_t* lens, uint32_t n_contigs,
                       stitch_index** out);
/* Flat blob for the one-time RCCL broadcast of the index (rank 0 -> all).  Call with buf == NULL to get *len. */
int stitch_index_serialize(const stitch_index*, void* buf, size_t* len);
int stitch_index_deserialize(const void* buf, size_t len, stitch_index** out);
uint32_t stitch_index_n_contigs(const stitch_index*);
void stitch_index_destroy(stitch_index*);

/* == Builder::build_aligners + build_sam_record_formatter (mod.rs:171-225).  Uploads the contigs (both strands when
 * opts->double_strand) and the column-0 state to `device_ordinal`. */
int stitch_ctx_create(int device_ordinal, const stitch_index*, const stitch_opts*, stitch_ctx** out);
void stitch_ctx_destroy(stitch_ctx*);

/* == Alignment, align/alignment.rs:16-51 (after remove_clipping / realign_origin, as Aligners::align returns it) */
typedef struct stitch_chain {
  int32_t score;
  uint32_t xstart, xend, ystart, yend, xlen, ylen;      /* x = contig, y = read (multi_contig_aligner.rs:334-345) */
  uint32_t start_contig_idx, end_contig_idx, length;
  uint64_t ops_begin; uint32_t ops_len;                 /* into the ops array */
  uint32_t pad;
} stitch_chain;

/* == AlignmentOperation, aligners/constants.rs:20-29.  kind: 0 Match 1 Subst 2 Del 3 Ins 4 Xclip(arg)
 * 5 Yclip(arg) 6 Xjump(contig,arg=x) 7 Yjump(arg) */
typedef struct stitch_op { uint8_t kind; uint8_t pad; uint16_t contig; uint32_t arg; } stitch_op;

typedef struct stitch_read_result {
  uint64_t chains_begin; uint32_t n_chains;             /* n_chains == 0 => unmapped (pre-align filter) */
  int32_t prealign_score; uint8_t has_prealign; uint8_t pad[3];
} stitch_read_result;

/* == Aligners::align for a batch.  `bases` = concatenated reads (any case; upper-cased like
 * FastxOwnedRecord::seq_upper_case, align/io.rs:64-66), offsets[n_reads+1].  Runs of identical consecutive reads
 * are aligned once (FastxGroupingIterator, align/io.rs:118-146).  Result arrays are owned by the ctx and stay valid
 * until the next stitch_align_batch / stitch_ctx_destroy on it.  cells_filled (optional) = sum over every full jump
 * DP executed (incl. origin re-alignments) of n * sum of filled contig lengths — the Gcells/s numerator. */
int stitch_align_batch(stitch_ctx*, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads,
                       const stitch_read_result** per_read, const stitch_chain** chains, const stitch_op** ops,
                       uint64_t* cells_filled);

/* == SamRecordFormatter::format (mod.rs:622-973) for read `read_idx` of the last batch, as SAM text: records are
 * '\n'-separated, no trailing newline.  `head` = FASTQ header line without '@', `quals` may be NULL (FASTA input).
 * Returns the text length (excluding NUL) or a negative error; if the length >= cap nothing is copied. */
long stitch_format_sam(stitch_ctx*, uint32_t read_idx, const char* head, const uint8_t* bases, const uint8_t* quals,
                       size_t n, char* buf, size_t cap);

/* Timing of the last stitch_align_batch on this ctx, measured with HIP events on the stream the kernels ran on:
 * fill_ms = time during which the DP fill kernel was running, walk_ms = fix-up + traceback kernel, launches = number of
 * fill launches, cells = DP cells filled by them.  Used by bench.py for the roofline line. */
typedef struct stitch_timing { double fill_ms, walk_ms, h2d_ms, d2h_ms, host_ms; uint64_t cells; uint32_t launches; uint32_t jobs;
                               double prealign_ms /* banded kernels incl. their transfers */, prealign_host_ms /* seeds, backbone, band: runs on host
                                  threads concurrently with prealign_ms of the chunk before, so the two overlap */;
                               uint32_t fill_kind /* kernel of the last fill launch: 0 generic int32, 1 Local-mode streaming, 2 Local-mode register-resident, 3 register-resident 32-bit (every mode, long reads) */,
                                        wg_per_read /* workgroups that shared one read in that launch */,
                                        fallbacks /* launches repeated with one workgroup per read after a partner timeout */,
                                        stream_runs /* fill launches whose teams were persistent: each pulled its next read off a queue when one ended (round 4) */;
                               uint64_t clk_shader_cycles, clk_ref_ticks /* register-resident fill only: shader cycles (s_memtime) and 100 MHz ticks
                                  (s_memrealtime) over the column loop of the first read of every launch, summed: cycles / ticks x 100 = MHz */;
                               double fill_kernel_ms /* sum of the fill kernels' own durations (what a profiler lists per dispatch).  fill_ms is the time during
                                  which a fill kernel was RUNNING: with two launches in flight (the next one takes the slots finished reads free) the
                                  two differ */;
                               uint32_t teams_retired /* persistent teams asked to leave early because a launch beside them waited beyond its bound (each frees
                                  its wave slots for the rest of the call; 0 in nearly every call) */, reserved_; } stitch_timing;
/* `out_size` = sizeof(stitch_timing) as the CALLER was compiled: the library copies min(out_size, its own size) bytes, so the struct can
 * grow at its end without overrunning a binding built against an older header (fields are only ever appended). */
int stitch_last_timing(const stitch_ctx*, stitch_timing* out, size_t out_size);

/* Test hook (host only, no device needed): the band of the pre-alignment filter for one (read, target strand) pair as the
 * library computes it — rows [lo[c], hi[c]) for the columns c = 0..target_len.  Returns 1 when the band is the full matrix
 * (no seed), 0 otherwise, negative on error.  tests/test_prealign.py compares it with the oracle's restatement. */
int stitch_prealign_band(const uint8_t* read, uint32_t read_len, const uint8_t* target, uint32_t target_len, uint32_t k, uint32_t w,
                         int32_t match, int32_t gap_open, int32_t gap_extend, uint16_t* lo, uint16_t* hi);

/* The same band as the DEVICE draws it in production (seeds and backbone on the host, the band from the backbone's pieces and the
 * choice of the score kernel in prealign_band.hip), for the same comparison.  *kernel_class: 0 = LDS-ring kernel, 2 = global-state
 * kernel, 3 = register-window kernel (only when the band is not the full matrix).  Needs the GPU `device`. */
int stitch_prealign_band_device(int device, const uint8_t* read, uint32_t read_len, const uint8_t* target, uint32_t target_len, uint32_t k, uint32_t w,
                                int32_t match, int32_t gap_open, int32_t gap_extend, uint16_t* lo, uint16_t* hi, uint32_t* kernel_class);

/* Multi-GPU sharding of a read stream (host only): the contiguous block [*lo, *hi) of the batch for `rank` of `world` ranks.
 * Blocks are equal shares cut only where two consecutive reads differ, so a run of identical reads — which the reference
 * aligns once (FastxGroupingIterator, align/io.rs:118-146) — stays on one rank; concatenated in rank order the blocks are the
 * stream, so per-rank outputs concatenated in rank order equal the single-GPU output (fg-stitch-cli/src/commands/align.rs:
 * 338-441 keeps input order through its channels).  Reads are compared as given (case-sensitively, like the reference). */
int stitch_shard_range(const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads, uint32_t world, uint32_t rank,
                       uint32_t* lo, uint32_t* hi);

/* Test hook (host only, no device needed): the library's Alignment::split_at_y (align/alignment.rs:207-360; used by
 * realign_origin to un-rotate a re-aligned read) on a caller-supplied chain.  `mode` is the alignment's own mode (0 local,
 * 1 query-local, 2 target-local, 3 global, 4 custom = what the traceback produces).  Writes the result chain and at most `cap`
 * operations; returns the number of operations of the result (nothing is written beyond cap) or a negative error.
 * tests/test_split_at_y_golden.py runs the reference's own test vectors through it. */
long stitch_split_at_y(const stitch_chain* in, const stitch_op* in_ops, int32_t mode, uint32_t y_pivot,
                       stitch_chain* out, stitch_op* out_ops, uint32_t cap);

/* Test hook (host only, no device needed): SamRecordFormatter::format (mod.rs:622-973, with SubAlignmentBuilder::build,
 * sub_alignment.rs:36-241) on CALLER-SUPPLIED chains, so that hand-traced known answers for the formatter's quirks
 * (tests/golden/sam_known_answers.json) can be checked against the library's host code without an alignment run.  Targets are
 * given by name and length (the formatter reads nothing else of them); chain k's operations are ops[ops_begin .. ops_begin +
 * ops_len).  Returns like stitch_format_sam. */
long stitch_format_sam_chains(const stitch_opts*, const char* const* target_names, const uint32_t* target_lens, uint32_t n_targets,
                             const char* head, const uint8_t* bases, const uint8_t* quals, size_t n,
                             const stitch_chain* chains, uint32_t n_chains, const stitch_op* ops,
                             int has_prealign, int32_t prealign, char* buf, size_t cap);

const char* stitch_last_error(void);
const char* stitch_version(void);

#ifdef __cplusplus
}
#endif
#endif
