#!/bin/bash
# The shader clock the fill kernel actually runs at, three ways, in ONE file (profiles/<tag>_clock.txt):
#   1. inside the kernel: s_memtime (shader cycles) against s_memrealtime (100 MHz) over each launch's column loop
#      (fill_regs.hip -> stitch_timing.clk_*; bench.py prints it as roofline.clock.in_kernel_mhz)
#   2. rocm-smi sclk / socket power, sampled every 0.5 s while bench.py runs (the run's steady-state rows are the ones above 500 W)
#   3. PMC with one launch at a time: GRBM_GUI_ACTIVE / 8 XCDs / t, SQ_BUSY_CYCLES / 32 / t, SQ_WAVE_CYCLES x 4 / waves / t
#   bash profiles/clock_probe.sh r03_a
set -o pipefail
tag=${1:-r03_x}
out=gpurun_out/clock_$tag
mkdir -p "$out"
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics Package Power" | tr '\n' ' '; echo; sleep 0.5; done ) > "$out/smi.txt" &
smi_pid=$!
python3 bench.py --steps 6 --warmup 2 --cpu-reads 0 > "$out/bench.json" 2> "$out/bench.err"
rc=$?
kill $smi_pid 2>/dev/null; wait $smi_pid 2>/dev/null
[ $rc -eq 0 ] || { echo "bench failed"; tail -5 "$out/bench.err"; exit 1; }
# leg 3 needs ONE launch at a time (with two fills in flight a dispatch's duration says nothing about its cycles): a kernel trace
# and one PMC pass of the same command with the fills ordered behind each other
export TMPDIR=/tmp
export STITCH_NO_FILL_OVERLAP=1
rocprofv3 --kernel-trace --stats -d "$out/trace1" -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-reads 0 > "$out/trace1.log" 2>&1 || echo "serial trace failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$out/pmc1" -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-reads 0 > "$out/pmc1.log" 2>&1 || echo "serial pmc failed"
unset STITCH_NO_FILL_OVERLAP
python3 - "$out" "$tag" <<'PY'
import json, os, re, sys
out, tag = sys.argv[1], sys.argv[2]
here = "profiles"
b = json.loads([l for l in open(os.path.join(out, "bench.json")) if l.startswith("{")][0])
rf = b["roofline"]
lines = [f"# shader clock of {rf['kernel']} on {b['device']['name']} ({tag}); bench: {b['value']:.1f} reads/s, {rf['avg_launch_ms']:.2f} ms per fill launch",
         "", "1. in-kernel (s_memtime / s_memrealtime over the column loop, every launch of the timed steps):",
         f"   {rf.get('clock', {}).get('in_kernel_mhz', float('nan')):.0f} MHz", "",
         "2. rocm-smi while the bench ran (0.5 s samples; sclk MHz, socket power W):"]
rows = []
for l in open(os.path.join(out, "smi.txt")):
    m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\).*Power \(W\): ([\d.]+)", l)
    if m:
        rows.append((int(m.group(1)), float(m.group(2))))
busy = [r for r in rows if r[1] >= 500.0]
lines.append("   all samples: " + " ".join(f"{c}/{int(p)}" for c, p in rows))
if busy:
    lines.append(f"   under load (>= 500 W, {len(busy)} samples): sclk {min(c for c, _ in busy)}-{max(c for c, _ in busy)} MHz, mean {sum(c for c, _ in busy) / len(busy):.0f}; "
                 f"power {min(p for _, p in busy):.0f}-{max(p for _, p in busy):.0f} W, mean {sum(p for _, p in busy) / len(busy):.0f}")
lines += ["", "3. PMC, one launch at a time (STITCH_NO_FILL_OVERLAP=1; profiled runs clock lower than plain ones: MI355X_MICROARCH.md, DVFS give-back):"]
import csv, glob
t = None
for pth in glob.glob(os.path.join(out, "trace1", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(pth)):
        if "fill_regs_kernel" in r["Name"]:
            t = float(r["AverageNs"]) * 1e-9
cnt = {}
for pth in glob.glob(os.path.join(out, "pmc1", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(pth)):
        if "fill_regs_kernel" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, d in per.items():
        cnt[k] = sum(d.values()) / len(d)
waves = rf["cells_per_launch"] / b["config"]["cells_per_read"] * 50
if t and cnt:
    lines.append(f"   launch duration (kernel trace, fills one after the other): {t * 1e3:.2f} ms; waves per launch: {waves:.0f}")
    if "GRBM_GUI_ACTIVE" in cnt: lines.append(f"   GRBM_GUI_ACTIVE / 8 / t        = {cnt['GRBM_GUI_ACTIVE'] / 8 / t / 1e6:.0f} MHz")
    if "SQ_BUSY_CYCLES" in cnt: lines.append(f"   SQ_BUSY_CYCLES / 32 / t         = {cnt['SQ_BUSY_CYCLES'] / 32 / t / 1e6:.0f} MHz   (one counter per shader engine: 8 XCDs x 4)")
    if "SQ_WAVE_CYCLES" in cnt: lines.append(f"   SQ_WAVE_CYCLES x 4 / waves / t  = {cnt['SQ_WAVE_CYCLES'] * 4 / waves / t / 1e6:.0f} MHz   (quad-cycles of resident waves: a lower bound, not every quad-cycle of a wave is counted)")
else:
    lines.append("   (no serial trace / PMC pass)")
open(os.path.join(here, f"{tag}_clock.txt"), "w").write("\n".join(lines) + "\n")
open(os.path.join(out, f"{tag}_clock.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
