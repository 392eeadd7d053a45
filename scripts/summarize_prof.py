#!/usr/bin/env python3
"""Summarise a scripts/profile.sh output directory: per-kernel stats and per-kernel average PMC values."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

out = Path(sys.argv[1])
csv.field_size_limit(1 << 30)


def short(name: str) -> str:
    name = name.replace("psk::", "")
    return name if len(name) < 110 else name[:107] + "..."


for f in sorted(out.glob("trace/**/*kernel_stats.csv")):
    print(f"== kernel stats ({f.relative_to(out)})")
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print(f"{short(row.get('Name', '')):112s} calls={row.get('Calls'):>6s} avg_ns={float(row.get('AverageNs', 0)):>12.0f} "
                  f"total_ns={row.get('TotalDurationNs'):>14s} pct={row.get('Percentage')}")
for d in sorted(out.glob("pmc_*")):
    for f in sorted(d.glob("**/*counter_collection.csv")):
        agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get("Kernel_Name", ""))
                c = row.get("Counter_Name", "")
                v = float(row.get("Counter_Value", 0) or 0)
                a = agg[k][c]
                a[0] += v
                a[1] += 1
        print(f"== pmc ({f.relative_to(out)}): average per dispatch")
        for k, cs in sorted(agg.items()):
            print(f"{k:112s} " + "  ".join(f"{c}={a[0] / max(a[1], 1):.4g} (n={a[1]})" for c, a in sorted(cs.items())))
