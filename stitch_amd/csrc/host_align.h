// Host-side mirror of the reference's result types and per-read orchestration:
//   Alignment + split_at_y                 fg-stitch-lib/src/align/alignment.rs:16-51, 207-360
//   Aligners::{remove_clipping, get_start_and_end_contig_indexes_for_realignment, realign_origin}
//                                          fg-stitch-lib/src/align/aligners/mod.rs:343-553
//   traceback_all's end-contig selection   fg-stitch-lib/src/align/traceback/mod.rs:152-217
//   SubAlignmentBuilder / SamRecordFormatter::format   align/sub_alignment.rs, aligners/mod.rs:622-973
// The DP itself never runs here: every alignment these functions consume comes out of the HIP kernels.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/stitch_gpu.h"
#include "dp_core.h"

namespace stitch {

struct HAln {                                  // == Alignment
    int32_t score = 0;
    uint32_t xstart = 0, xend = 0, ystart = 0, yend = 0, xlen = 0, ylen = 0;
    uint32_t start_contig_idx = 0, end_contig_idx = 0, length = 0;
    std::vector<stitch_op> ops;
};

inline bool op_is_aln(const stitch_op& o) { return o.kind <= OP_INS; }
inline int32_t op_len_x(const stitch_op& o, uint32_t x_index) {          // constants.rs:61-71
    switch (o.kind) {
        case OP_MATCH: case OP_SUBST: case OP_INS: return 1;
        case OP_XCLIP: return (int32_t)o.arg;
        case OP_XJUMP: return (int32_t)o.arg - (int32_t)x_index;
        default: return 0;
    }
}
inline uint32_t op_len_y(const stitch_op& o) {                            // constants.rs:74-84
    switch (o.kind) {
        case OP_MATCH: case OP_SUBST: case OP_DEL: return 1;
        case OP_YCLIP: case OP_YJUMP: return o.arg;
        default: return 0;
    }
}
inline stitch_op mk_op(uint8_t kind, uint32_t contig, uint32_t arg) { stitch_op o; o.kind = kind; o.pad = 0; o.contig = (uint16_t)contig; o.arg = arg; return o; }

void remove_clipping(const stitch_opts& o, HAln& a);                      // mod.rs:343-353
HAln split_at_y(const HAln& a, int mode, uint32_t y_pivot);               // alignment.rs:207-360

struct SubAln {                                // == SubAlignment after the swap (sub_alignment.rs:10-19, 224-237)
    uint32_t contig_idx = 0, query_start = 0, query_end = 0, target_start = 0, target_end = 0;
    std::vector<std::pair<char, uint32_t>> cigar;
    int32_t score = 0, num_edits = 0;
};
// SubAlignmentBuilder::build(chain, swap = true) (sub_alignment.rs:169-241); false + err on the reference's panic
bool build_subs(const HAln& chain, const stitch_opts& o, std::vector<SubAln>& out, std::string& err);

struct TargetInfo { std::string name; uint32_t len; };
// SamRecordFormatter::format as SAM text lines (mod.rs:622-973)
bool format_sam_records(const stitch_opts& o, const std::vector<TargetInfo>& targets, const std::string& head,
                        const uint8_t* bases, const uint8_t* quals, size_t n, const std::vector<HAln>& chains,
                        bool has_prealign, int32_t prealign, std::vector<std::string>& out, std::string& err);

}  // namespace stitch
