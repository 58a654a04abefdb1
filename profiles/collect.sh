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
# (under rocprofv3 the library goes launch by launch: the tool reports a launch beside resident teams complete only when the teams' own
# dispatch is, DESIGN.md 4 "Persistent teams"; the bench line of this pass says launch_mode "launch by launch" and its avg_launch_ms is what
# the kernel statistics must agree with.  The production line, persistent teams, comes last, un-profiled.)
rocprofv3 --kernel-trace --stats -d "$out/trace" -o run --output-format csv -- python3 bench.py --cpu-reads 0 > "$out/trace.log" 2>&1 || { echo "trace failed"; exit 1; }
grep '^{"metric"' "$out/trace.log" | tail -1 > "profiles/${tag}_bench_fill_under_rocprofv3.json"
# (a counter pass serialises the process's kernels: the persistent teams of round 4 wait for the walks the host launches beside them, and two
# fills in flight for each other, so the counted passes run launch by launch; same kernel, same counts per cell)
export STITCH_NO_STREAM=1 STITCH_NO_FILL_OVERLAP=1
for grp in "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  d="$out/pmc_$(echo $grp | tr ' ' '_' | cut -c1-40)"
  rocprofv3 --pmc $grp -d "$d" -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-reads 0 --reads-per-step 320 > "$d.log" 2>&1 || { echo "pmc $grp failed"; exit 1; }
  echo "pmc $grp done" >> "$out/progress.txt"
done
python3 profiles/summarize.py "$out" "$tag"
unset STITCH_NO_STREAM STITCH_NO_FILL_OVERLAP
# the bench line last: its roofline.traffic / roofline.valu figures read the PMC summary just written
python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; exit 1; }
cp "$out/bench.json" "profiles/${tag}_bench_fill.json"
cp profiles/${tag}_* "$out/"        # gpurun merges only gpurun_out/ back: copy from there into profiles/ and commit
