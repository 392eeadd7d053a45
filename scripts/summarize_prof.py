#!/usr/bin/env python3
"""Summarise a scripts/profile.sh output directory: per-kernel stats and per-kernel average PMC values."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
csv.field_size_limit(1 << 30)


def short(name: str) -> str:
    name = name.replace("psk::", "").replace("void ", "")
    return name if len(name) < 120 else name[:117] + "..."


for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    print("== rocprofv3 --kernel-trace --stats: per-kernel durations")
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print(f"{short(row['Name']):120s} calls={row['Calls']:>6s} avg_us={float(row['AverageNs']) / 1e3:10.1f} "
                  f"min_us={float(row['MinNs']) / 1e3:10.1f} max_us={float(row['MaxNs']) / 1e3:10.1f} pct={row['Percentage']}")
for d in sorted(glob.glob(out + "/pmc_*")):
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                a = agg[short(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"] or 0)
                a[1] += 1
        print(f"== rocprofv3 --pmc ({d.split('/')[-1]}): average counter value per dispatch")
        for k, cs in sorted(agg.items()):
            if "at::native" in k or "rocclr" in k:
                continue
            print(f"{k:120s} " + "  ".join(f"{c}={a[0] / max(a[1], 1):.6g} (n={a[1]})" for c, a in sorted(cs.items())))
