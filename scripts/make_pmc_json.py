#!/usr/bin/env python3
"""gpurun_out/<round>/pmc_<op>.json (scripts/pmc_op.sh) -> profiles/<round>_pmc_traffic.json and profiles/<round>_l2_hit.json, the files
bench.py reads `roofline.traffic` and `roofline.l2_hit` from.   make_pmc_json.py [dir] [round tag, default r03]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
TAG = sys.argv[2] if len(sys.argv) > 2 else "r03"
src = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / TAG
names = {"bloom_add": "bloom_insert", "bloom_check": "bloom_check", "bloom_check_fresh": "bloom_check_all_fresh", "bloom_check_half": "bloom_check_half_fresh", "cms_add": "cms_add_weighted",
         "cms_check": "cms_check", "cbf_add": "cbf_add", "cbf_check": "cbf_check", "cbf_check_kept": "cbf_check_unchanged_table", "cbf_remove": "cbf_remove", "cfg4_stream": "cfg4_stream",
         "bloom31_add": "bloom31_insert", "bloom31_check": "bloom31_check"}
SCALABLE = {"bloom31_add", "bloom31_check"}  # measured on one 2^25-key call; cfg 5 makes ceil(n / 2^25) such calls per step
out = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over scripts/prof_ops.py <op> (scripts/pmc_op.sh, "
              "scripts/profile_r02.sh); counters in KiB; FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section: on gfx950 it reports half of a wide "
              "coalesced streaming read; calibrated in round 1 on k_part_scatter reading exactly the 160 MB of keys -> FETCH_SIZE 78 226 KiB). "
              "k_lookup_collect mixes 4- and 16-byte loads, for which the factor lies between 1 and 2: its doubled figure is an upper bound. "
              "L2 -> fabric requests include Infinity Cache hits: upper bound of the HBM bytes.",
    "keys": None,
}
l2 = {"source": "rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum over scripts/prof_ops.py <op> (scripts/pmc_op.sh): L2 hit rate = "
                "TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) over the kernels of one launch (MI355X_MICROARCH.md, L2 section)"}


def factor(kernel: str) -> int:
    """FETCH_SIZE correction: 2 for the streaming kernels (16 B per lane, coalesced), 1 for the direct kernels, whose random
    4-byte gathers are one 64-byte request each (round 1: TCC_EA0_RDREQ == probe count)"""
    return 1 if kernel.startswith(("k_apply", "k_cbf_remove", "k_cbf_to_remove")) else 2


for f in sorted(src.glob("pmc_*.json")):
    d = json.loads(f.read_text())
    if f.stem == "pmc_cfg4_stream_borrow":
        d["op"] = "cfg4_stream_borrow_keys"
    if d["keys"] == 10_000_000:
        out["keys"] = d["keys"]
    # kernels of the SET-UP (prof_ops.py fills the filter once before the profiled lookups: 1 dispatch against `launches` profiled
    # calls) are not part of the operation: listed aside, not counted
    setup = {k: v for k, v in d["kernels"].items() if v["dispatches_per_launch"] < 0.3 and not k.startswith("__amd_rocclr")}
    d["kernels"] = {k: v for k, v in d["kernels"].items() if k not in setup}
    # set-up launches that were COUNTED as launches of the operation but ran the set-up's kernels (cms_add: the first weighted adds go out in
    # the PayWeight format, until the tally that selects PayWeightSmall is published) leave every kept kernel at < 1 dispatch per launch:
    # per-launch figures are per launch that RAN them
    dmax = max([v["dispatches_per_launch"] for k, v in d["kernels"].items() if not k.startswith("__amd_rocclr")] or [1.0])
    if setup and 0 < dmax < 0.999:
        for v in d["kernels"].values():
            for f in ("fetch_KiB_per_launch", "write_KiB_per_launch", "l2_requests_per_launch", "dispatches_per_launch"):
                if v.get(f) is not None:
                    v[f] = round(v[f] / dmax, 2)
    hbm = int(sum(factor(k) * v["fetch_KiB_per_launch"] + v["write_KiB_per_launch"] for k, v in d["kernels"].items()) * 1024)
    rec = {"keys": d["keys"], "kernels": d["kernels"], "hbm_bytes_per_launch": hbm, "bytes_per_key": round(hbm / d["keys"], 1), "l2_hit": d.get("l2_hit")}
    if setup:
        rec["setup_kernels_not_counted"] = sorted(setup)
    if d["op"] in SCALABLE:
        rec["per_key_scalable"] = True
    out[names.get(d["op"], d["op"])] = rec
    l2[names.get(d["op"], d["op"])] = {"l2_hit": d.get("l2_hit"), "keys": d["keys"],
                                        "kernels": {k: {"l2_hit": v.get("l2_hit"), "l2_requests_per_launch": v.get("l2_requests_per_launch")} for k, v in d["kernels"].items()}}
# prof_ops.py's "cbf_remove" puts the keys back before every remove (else there is nothing to remove): the remove alone = that minus the add
if "cbf_remove" in out and "cbf_add" in out and out["cbf_remove"]["keys"] == out["cbf_add"]["keys"]:
    both = out.pop("cbf_remove")
    both["note"] = "add_many + remove_many of the same keys per launch (scripts/prof_ops.py)"
    out["cbf_add_plus_remove"] = both
    hbm = both["hbm_bytes_per_launch"] - out["cbf_add"]["hbm_bytes_per_launch"]
    out["cbf_remove"] = {"keys": both["keys"], "hbm_bytes_per_launch": hbm, "bytes_per_key": round(hbm / both["keys"], 1),
                         "kernels": {k: v for k, v in both["kernels"].items() if "<3," in k or "SpillRaiseFlagCounter" in k or "PayUnitMasked" in k or "k_nib_gather" in k or "k_nib_collect" in k},
                         "l2_hit": both.get("l2_hit"), "note": "cbf_add_plus_remove minus cbf_add (the validated remove's optimistic path: pass 1 + k_nib_apply<3>)"}
    l2["cbf_add_plus_remove"] = l2.pop("cbf_remove")
    l2["cbf_remove"] = {"l2_hit": next((v.get("l2_hit") for k, v in both["kernels"].items() if "<3," in k), None), "keys": both["keys"],
                        "note": "L2 hit rate of k_nib_apply<3> (the optimistic decrement), the dominant kernel of the validated remove"}
(ROOT / "profiles" / f"{TAG}_pmc_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
(ROOT / "profiles" / f"{TAG}_l2_hit.json").write_text(json.dumps(l2, indent=1) + "\n")
print(f"wrote profiles/{TAG}_pmc_traffic.json:", {k: v["bytes_per_key"] for k, v in out.items() if isinstance(v, dict)})
print(f"wrote profiles/{TAG}_l2_hit.json:", {k: v["l2_hit"] for k, v in l2.items() if isinstance(v, dict)})
