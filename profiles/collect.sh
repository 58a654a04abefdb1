#!/bin/bash
# Collects the numbers profiles/ holds, on a GPU box (run from the repository root):
#   bash profiles/collect.sh r01_g
# 1. bench.py as the driver runs it            -> profiles/<tag>_bench_fill.json
# 2. rocprofv3 --kernel-trace --stats of that   -> profiles/<tag>_kernel_stats_fill.csv
# 3. PMC counters, one pass per group (TCC FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md), never
#    combined with a trace domain                                                        -> profiles/<tag>_pmc_fill.json
# The program follows `--` directly (no env/bash wrapper: the profiler has initialised the GPU by then).
set -o pipefail
tag=${1:-r01_x}
out=gpurun_out/collect_$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -o run --output-format csv -- python3 bench.py --cpu-reads 0 > "$out/trace.log" 2>&1 || { echo "trace failed"; exit 1; }
for grp in "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  d="$out/pmc_$(echo $grp | tr ' ' '_' | cut -c1-40)"
  rocprofv3 --pmc $grp -d "$d" -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-reads 0 > "$d.log" 2>&1 || { echo "pmc $grp failed"; exit 1; }
  echo "pmc $grp done" >> "$out/progress.txt"
done
python3 profiles/summarize.py "$out" "$tag"
# the bench line last: its roofline.traffic / roofline.valu figures read the PMC summary just written
python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; exit 1; }
cp "$out/bench.json" "profiles/${tag}_bench_fill.json"
cp profiles/${tag}_* "$out/"        # gpurun merges only gpurun_out/ back: copy from there into profiles/ and commit
