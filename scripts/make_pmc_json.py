#!/usr/bin/env python3
"""gpurun_out/r02/pmc_<op>.json (scripts/pmc_op.sh) -> profiles/r02_pmc_traffic.json, the file bench.py reads `roofline.traffic` from"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "r02"
names = {"bloom_add": "bloom_insert", "bloom_check": "bloom_check", "bloom_check_fresh": "bloom_check_all_fresh", "cms_add": "cms_add_weighted",
         "cms_check": "cms_check", "cbf_add": "cbf_add", "cbf_check": "cbf_check", "cbf_remove": "cbf_remove"}
out = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over scripts/prof_ops.py <op> (scripts/pmc_op.sh, "
              "scripts/profile_r02.sh); counters in KiB; FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section: on gfx950 it reports half of a wide "
              "coalesced streaming read; calibrated in round 1 on k_part_scatter reading exactly the 160 MB of keys -> FETCH_SIZE 78 226 KiB). "
              "k_lookup_collect mixes 4- and 16-byte loads, for which the factor lies between 1 and 2: its doubled figure is an upper bound. "
              "L2 -> fabric requests include Infinity Cache hits: upper bound of the HBM bytes.",
    "keys": None,
}
def factor(kernel: str) -> int:
    """FETCH_SIZE correction: 2 for the streaming kernels (16 B per lane, coalesced), 1 for the direct kernels, whose random
    4-byte gathers are one 64-byte request each (round 1: TCC_EA0_RDREQ == probe count)"""
    return 1 if kernel.startswith(("k_apply", "k_cbf_remove", "k_cbf_to_remove")) else 2


for f in sorted(src.glob("pmc_*.json")):
    d = json.loads(f.read_text())
    out["keys"] = d["keys"]
    hbm = int(sum(factor(k) * v["fetch_KiB_per_launch"] + v["write_KiB_per_launch"] for k, v in d["kernels"].items()) * 1024)
    out[names.get(d["op"], d["op"])] = {"kernels": d["kernels"], "hbm_bytes_per_launch": hbm, "bytes_per_key": round(hbm / d["keys"], 1)}
(ROOT / "profiles" / "r02_pmc_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
print("wrote profiles/r02_pmc_traffic.json:", {k: v["bytes_per_key"] for k, v in out.items() if isinstance(v, dict)})
