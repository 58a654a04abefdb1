#!/bin/bash
# What bounds the fill?  Times the Local-mode fill of the bench workload in four builds (run on a GPU box from the repository
# root; results are garbage in the experiment builds, only the kernel time is read — STITCH_EXP_FILL_ONLY skips the walk):
#   base      the shipped kernel
#   nostate   the row state never leaves the registers (no 8 B/row load + store per column)
#   notb      no traceback bytes and no y-suffix records are stored
#   neither   both removed: arithmetic, per-column synchronisation and the exchange only
#   bash profiles/exp_bound.sh <tag> [extra env assignments for the run, e.g. STITCH_NO_REGS=1]
set -o pipefail
tag=${1:-exp}; shift
out=gpurun_out/exp_bound_$tag
mkdir -p "$out"
for v in base nostate notb neither; do
  case $v in
    base) defs="" ;; nostate) defs="STITCH_EXP_NOSTATE" ;; notb) defs="STITCH_EXP_NOTB" ;; neither) defs="STITCH_EXP_NOSTATE STITCH_EXP_NOTB" ;;
  esac
  STITCH_DEFINES="$defs" python3 -c "from stitch_amd import build; build.build(force=True)" > "$out/build_$v.log" 2>&1 || { echo "build $v failed"; exit 1; }
  env "$@" STITCH_EXP_FILL_ONLY=1 python3 bench.py --cpu-reads 0 > "$out/$v.json" 2> "$out/$v.err" || { echo "bench $v failed"; tail -5 "$out/$v.err"; exit 1; }
  python3 - "$out/$v.json" "$v" <<'PY' | tee -a "$out/summary.txt"
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[2]:8s} fill {r['avg_launch_ms']:8.1f} ms per launch of {r['cells_per_launch']:.4g} cells = {r['fill_gcells_per_sec']:7.1f} Gcells/s  ({r['kernel']})")
PY
done
python3 -c "from stitch_amd import build; build.build(force=True)" > "$out/build_restore.log" 2>&1
cp "$out/summary.txt" "profiles/${tag}_exp_bound.txt"; cp "profiles/${tag}_exp_bound.txt" "$out/"
