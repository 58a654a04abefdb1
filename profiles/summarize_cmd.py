"""profiles/collect_cmd.sh's summaries: <tag>_kernel_stats_<name>.csv and <tag>_pmc_<name>.json — per kernel (template arguments kept:
fill_regs_kernel<1, false> and <4, true> are different programs) the dispatches seen and the per-dispatch average of every counter,
summed over the counter's instances, with the derived figures the roofline tables of DESIGN.md quote:
  bytes      = (2 x FETCH_SIZE + WRITE_SIZE) x 1024   (KB units; FETCH_SIZE counts half of a wide read on gfx950: MI355X_MICROARCH.md)
  wait_any   = SQ_WAIT_ANY / SQ_WAVE_CYCLES,  active = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  avg_ms     = average dispatch duration of the kernel-trace pass."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out, tag, name = sys.argv[1], sys.argv[2], sys.argv[3]
here = os.path.dirname(os.path.abspath(__file__))


def short(n):
    n = re.sub(r"^void\s+", "", n)
    return re.sub(r"\(.*", "", n)


dur = {}
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(here, f"{tag}_kernel_stats_{name}.csv"), "w") as f:
        f.write("Name,Calls,TotalDurationUs,AverageUs,Percentage\n")
        for r in rows:
            f.write(f"\"{r['Name']}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e3:.3f},{float(r['AverageNs']) / 1e3:.3f},{r['Percentage']}\n")
            dur[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "total_ms": float(r["TotalDurationNs"]) / 1e6}

pmc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
for path in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "stitch::" not in r["Kernel_Name"]:
            continue
        pmc[short(r["Kernel_Name"])][r["Counter_Name"]][(path, r["Dispatch_Id"])] += float(r["Counter_Value"])
summary = {}
for k, cs in pmc.items():
    e = {c: {"launches": len(d), "avg_per_launch_raw": sum(d.values()) / len(d)} for c, d in sorted(cs.items())}
    g = lambda c: e[c]["avg_per_launch_raw"] if c in e else None
    der = {}
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        der["bytes_per_dispatch"] = (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0
    if g("SQ_WAVE_CYCLES"):
        for nm, c in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("active_inst_any", "SQ_ACTIVE_INST_ANY"), ("active_inst_valu", "SQ_ACTIVE_INST_VALU")):
            if g(c) is not None:
                der[nm] = g(c) / g("SQ_WAVE_CYCLES")
    for nm, c in (("valu_wave_insts", "SQ_INSTS_VALU"), ("salu_wave_insts", "SQ_INSTS_SALU"), ("lds_wave_insts", "SQ_INSTS_LDS"), ("vmem_rd_wave_insts", "SQ_INSTS_VMEM_RD"), ("vmem_wr_wave_insts", "SQ_INSTS_VMEM_WR")):
        if g(c) is not None:
            der[nm] = g(c)
    if k in dur:
        der.update({"trace_calls": dur[k]["calls"], "trace_avg_ms": dur[k]["avg_ms"]})
        if "bytes_per_dispatch" in der:
            # (the PMC passes run the kernels one at a time: their dispatches are the classic launches; the trace pass may hold persistent ones)
            der["note"] = "bytes_per_dispatch and instruction counts are per dispatch of the PMC passes (classic launches, STITCH_NO_STREAM=1)"
    e["derived"] = der
    summary[k] = e
summary["command"] = " ".join(sys.argv[4:]) if len(sys.argv) > 4 else None
summary["units"] = "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (FETCH_SIZE counts half of a wide read on gfx950)"
for log in glob.glob(os.path.join(out, "pmc_*.log")) + glob.glob(os.path.join(out, "trace.log")):
    for line in open(log, errors="replace"):
        if line.startswith('{"config"'):
            summary.setdefault("runs", {})[os.path.basename(log)] = json.loads(line)
with open(os.path.join(here, f"{tag}_pmc_{name}.json"), "w") as f:
    json.dump(summary, f, indent=1)
print("kernels:", [k for k in summary if k.startswith("stitch")])
