#!/bin/bash
# Kernel statistics and PMC counters of ANY command of this repository, per kernel (counters for every kernel that
# carries a claim), on a GPU box, from the repository root:
#   bash profiles/collect_cmd.sh r04 cfg5 python3 tests/config_runs.py --config cfg5
# -> profiles/<tag>_kernel_stats_<name>.csv   rocprofv3 --kernel-trace --stats of the command as it runs in production
#    profiles/<tag>_pmc_<name>.json           one --pmc pass per counter group (never combined with a trace domain; the program follows
#                                             `--` directly), per kernel: dispatches seen, average per dispatch, summed over instances
# The PMC passes run with STITCH_NO_STREAM=1 STITCH_NO_FILL_OVERLAP=1: a counter pass serialises the process's kernels, and the
# persistent teams (and two fills in flight) wait for launches beside them — the kernels' code and their per-cell counts are the same.
set -o pipefail
tag=$1; name=$2; shift 2
out=gpurun_out/collect_${tag}_${name}
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -o run --output-format csv -- "$@" > "$out/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$out/trace.log"; exit 1; }
export STITCH_NO_STREAM=1 STITCH_NO_FILL_OVERLAP=1
for grp in "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  d="$out/pmc_$(echo $grp | tr ' ' '_' | cut -c1-40)"
  rocprofv3 --pmc $grp -d "$d" -o run --output-format csv -- "$@" > "$d.log" 2>&1 || { echo "pmc $grp failed"; tail -5 "$d.log"; exit 1; }
  echo "pmc $grp done" >> "$out/progress.txt"
done
python3 profiles/summarize_cmd.py "$out" "$tag" "$name"
cp profiles/${tag}_*_${name}.* "$out/" 2>/dev/null
