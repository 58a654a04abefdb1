"""Turns the rocprofv3 output of profiles/collect.sh into the two committed summaries:
<tag>_kernel_stats_fill.csv (per-kernel calls / total / average duration) and <tag>_pmc_fill.json (per kernel and
counter: launches seen and the raw per-launch average, summed over the counter's instances)."""
import csv
import re
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
from bench import kernel_src_hash          # the hash bench.py compares before it quotes these counters

stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(here, f"{tag}_kernel_stats_fill.csv"), "w") as f:
        f.write("Name,Calls,TotalDurationUs,AverageUs,Percentage\n")
        for r in rows:
            f.write(f"\"{r['Name']}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e3:.3f},{float(r['AverageNs']) / 1e3:.3f},{r['Percentage']}\n")

pmc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))          # kernel -> counter -> dispatch -> value
for path in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "stitch::" not in r["Kernel_Name"]:
            continue
        name = re.sub(r"<.*", "", re.sub(r"^void\s+", "", r["Kernel_Name"].split("(")[0]))      # "void stitch::fill_regs_kernel<1>(...)" -> "stitch::fill_regs_kernel"
        pmc[name][r["Counter_Name"]][(path, r["Dispatch_Id"])] += float(r["Counter_Value"])
summary = {k: {c: {"launches": len(d), "avg_per_launch_raw": sum(d.values()) / len(d)} for c, d in sorted(cs.items())} for k, cs in pmc.items()}
cells = None
for log in glob.glob(os.path.join(out, "pmc_*.log")):                       # the bench line of a PMC pass: cells per launch
    for line in open(log, errors="replace"):
        if line.startswith('{"metric"'):
            cells = json.loads(line)["roofline"]["cells_per_launch"]
if summary:
    summary["cells_per_launch"] = cells
    summary["kernel_src_sha"] = kernel_src_hash()
    summary["units"] = "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (FETCH_SIZE counts half of a wide read on gfx950)"
    with open(os.path.join(here, f"{tag}_pmc_fill.json"), "w") as f:
        json.dump(summary, f, indent=1)
print("kernel stats:", bool(stats), "pmc kernels:", list(summary))
