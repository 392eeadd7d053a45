#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sketch engine on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): BloomFilter(est_elements=28005615, fpr=0.01) -> m = 2^28 bits, k = 7,
default_fnv_1a; one STEP = clear the filter, batch-insert 10M synthetic 16-byte keys, (N > 1: merge the
per-GPU replicas with allreduce(OR) over RCCL), batch-check the same 10M keys.  Keys are generated on the
device (counter-based splitmix64) before the timed region, i.e. inputs are resident in HBM.
`value` = all ranks' key operations (insert + lookup) per second, in million keys/s.  Weak scaling: every
rank owns 10M keys of the stream.

The JSON line also carries
  roofline     -- the dominant kernel (Bloom insert): algorithmic bytes (72 B/key = 16 B key + 7 x (4 B read +
                  4 B write)) / average kernel time from HIP events on the launch stream, vs 8 TB/s HBM peak.
  cpu_baseline -- the plain-C oracle (oracle/, a port of the reference semantics) timed on this box's host
                  cores on a bounded sample of the same workload (rank 0, N = 1 only).
  detail       -- per-phase rates, CMS (2^20 x 5) and CBF rates, and the measured random-access ceilings
                  (GUPS-style atomics / gathers into a table of the same size).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SEED = 0x5EED
BYTES = {"bloom_insert": 72, "bloom_check": 45, "cms_add": 60, "cms_check": 40, "cbf_add": 76, "cbf_check": 48}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", "--keys-per-rank", dest="n", type=int, default=10_000_000, help="keys per rank per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-detail", action="store_true", help="skip the CMS / CBF / GUPS side measurements")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: merge, then look up (no lookup pass 1 under the merge)")
    return ap.parse_args()


def gen_keys(n, start, device):
    from pyprobables_amd import _native as N

    t = torch.empty((n, 16), dtype=torch.uint8, device=f"cuda:{device}")
    N.check(N.lib().psk_gen_keys16(t.data_ptr(), start, n, SEED, device, torch.cuda.current_stream(device).cuda_stream or None))
    return t


def gen_weights(n, start, device):
    from pyprobables_amd import _native as N

    t = torch.empty(n, dtype=torch.int32, device=f"cuda:{device}")
    N.check(N.lib().psk_gen_weights(t.data_ptr(), start, n, SEED, device, torch.cuda.current_stream(device).cuda_stream or None))
    return t


class EventTimer:
    """HIP events on torch's current stream == the stream the engine launches on"""

    def __init__(self):
        self.pairs = {}

    def time(self, name, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.pairs.setdefault(name, []).append((a, b))
        return r

    def mean_ms(self, name):
        p = self.pairs.get(name, [])
        return sum(a.elapsed_time(b) for a, b in p) / len(p) if p else float("nan")


def timed_loop(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def side_measurements(device, n):
    """CMS / CBF rates and the GUPS-style random-access ceilings (not part of `value`)"""
    import pyprobables_amd as pa
    from pyprobables_amd import _native as N

    out = {}
    keys = gen_keys(n, 0, device)
    w = gen_weights(n, 0, device)
    cms = pa.CountMinSketch(width=2**20, depth=5, device=device)
    ms = timed_loop(lambda: cms.add_many(keys, w), 5)
    out["cms_add_Mupd_s"] = n / ms / 1e3
    out["cms_add_GBs"] = n * BYTES["cms_add"] / ms / 1e6
    ms = timed_loop(lambda: cms.check_many(keys), 5)
    out["cms_check_Mkeys_s"] = n / ms / 1e3
    del cms
    ncbf = min(n, 10_000_000)
    cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01, device=device)  # 2^28 x u32 = 1 GiB
    ms = timed_loop(lambda: cbf.add_many(keys[:ncbf]), 3, warm=1)
    out["cbf_add_Mops_s"] = ncbf / ms / 1e3
    ms = timed_loop(lambda: cbf.remove_many(keys[:ncbf]), 1, warm=0)
    out["cbf_remove_Mops_s"] = ncbf / ms / 1e3
    ms = timed_loop(lambda: cbf.check_many(keys[:ncbf]), 3, warm=1)
    out["cbf_check_Mkeys_s"] = ncbf / ms / 1e3
    del cbf
    # random-access ceilings at the headline table size (2^23 words = 32 MiB) and at 1 GiB
    st = lambda: torch.cuda.current_stream(device).cuda_stream or None  # noqa: E731
    sink = torch.zeros(1, dtype=torch.int64, device=f"cuda:{device}")
    nprobe = 7 * n
    for label, words in (("32MiB", 2**23), ("1GiB", 2**28)):
        tab = torch.zeros(words, dtype=torch.int32, device=f"cuda:{device}")
        for op, name in ((0, "atomic_or"), (1, "atomic_add"), (2, "gather")):
            ms = timed_loop(lambda: N.check(N.lib().psk_gups(tab.data_ptr(), words, nprobe, op, 12345, sink.data_ptr(), device, st())), 3, warm=1)
            out[f"gups_{name}_{label}_Gprobes_s"] = nprobe / ms / 1e6
        del tab
    return out


def cpu_baseline(n_sample, reps=6):
    """the plain-C oracle (kind 'port') on this box's host cores, single thread"""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle

    ob = oracle.OracleBloom(2**28, 7)
    t_total, ops = 0.0, 0
    for r in range(reps):
        keys = oracle.gen_keys16(r * n_sample, n_sample)
        t0 = time.perf_counter()
        ob.add_keys(keys)
        res = ob.check_keys(keys)
        t_total += time.perf_counter() - t0
        ops += 2 * n_sample
        assert bool(res.all())
    return {
        "value": ops / t_total / 1e6,
        "unit": "Mkeys/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{reps} x (insert {n_sample} + check {n_sample}) 16-byte keys into m=2^28 k=7, oracle/psk_oracle.c (gcc -O2)",
        "seconds": t_total,
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if rank != 0:  # only rank 0 reports; keep library banners of the other ranks off the job's stdout
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N > 1 through torch.distributed.run (one process per GPU)")
    # PSK_BENCH_SINGLE_DEVICE=1 (test hook): every rank uses cuda:0 and the gloo backend, so that the N > 1 logic (shard
    # offsets, two real replicas, merge, overlap) can be driven on a one-GPU box; RCCL refuses two ranks on one device
    single_device = bool(os.environ.get("PSK_BENCH_SINGLE_DEVICE"))
    if single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # PSK_BENCH_FORCE_DIST=1 drives the whole N > 1 code path (RCCL init, merge, barriers) with a single rank
    distributed = world > 1 or bool(os.environ.get("PSK_BENCH_FORCE_DIST"))
    if distributed:
        import torch.distributed as dist

        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ["PSK_FORCE_MERGE_PATH"] = "1"

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    import pyprobables_amd as pa
    from pyprobables_amd import parallel

    dev, n = local_rank, args.n
    keys = gen_keys(n, rank * n, dev)  # rank r owns keys [r*n, (r+1)*n) of the stream: resident before timing
    blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=dev)
    assert blm.number_bits == 2**28 and blm.number_hashes == 7
    timer = EventTimer()
    state = {}

    # A timing-enabled HIP event is a barrier packet: the next kernel cannot be dispatched ahead of it, ~7 us of bubble
    # each.  The timed steps therefore carry only the ONE event pair the roofline needs (around the insert launch), and
    # only on a sample of the steps; the other phases are timed in a separate instrumented pass after the timed region.
    def step(record, every_phase=False):
        plain = lambda _n, f: f()  # noqa: E731
        t = timer.time if (record and every_phase) else plain
        ti = timer.time if record else plain
        t("clear", blm.clear)
        ti("insert" if not every_phase else "insert_detail", lambda: blm.add_many(keys))
        if distributed and not args.no_overlap:
            # merge on a side stream; pass 1 of the lookup (hash + partition: it never reads the table) runs under it
            def merge_and_check():
                h = parallel.merge_bloom_async(blm)
                blm.check_many_begin(keys)
                h.wait()
                return blm.check_many_finish()

            state["res"] = t("merge+check", merge_and_check)
        else:
            if distributed:
                t("merge", lambda: parallel.merge_bloom(blm, sync_elements=False))  # table merge only: no host sync
            state["res"] = t("check", lambda: blm.check_many(keys))

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    sample_every = max(1, min(20, args.steps // 3))  # the insert launch is event-timed on every sample_every-th step (>= 3 samples)
    for it in range(args.steps):
        step(it % sample_every == 0)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{dev}")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    for _ in range(min(args.steps, 10)):  # instrumented pass (not part of `value`): per-phase HIP-event times
        step(True, every_phase=True)
    fence()
    ok = bool(state["res"].all().item())  # every inserted key must be found (size-independent parity property)
    bits_set = blm._cnt_number_bits_set()
    merged_ok = None
    if distributed:
        # multi-GPU parity, outside the timed region: the merged replica must equal ONE filter fed every rank's keys
        ref = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=dev)
        for r in range(world):
            ref.add_many(gen_keys(n, r * n, dev))
        merged_ok = bool(torch.equal(ref.table_tensor, blm.table_tensor))
        del ref

    ms_step = elapsed / args.steps * 1e3
    total_ops = 2 * n * world
    ins_ms, chk_ms = timer.mean_ms("insert"), timer.mean_ms("check")
    overlapped = distributed and not args.no_overlap
    mc_ms = timer.mean_ms("merge+check") if overlapped else None
    ach = n * BYTES["bloom_insert"] / (ins_ms * 1e-3) / 1e9
    traffic = None  # HBM bytes per insert launch from the committed PMC profile (same workload, same kernels)
    try:
        pmc = json.loads((ROOT / "profiles" / "r01_pmc_traffic.json").read_text())
        if pmc["keys"] == n:
            traffic = pmc["bloom_insert"]["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    line = {
        "metric": "million keys/sec insert+lookup (Bloom m=2^28 k=7)",
        "value": total_ops / (elapsed / args.steps) / 1e6,
        "unit": "Mkeys/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": "BloomFilter(est_elements=28005615, fpr=0.01): m=2^28 bits, k=7, default_fnv_1a; per rank per step: "
                        "clear + insert 10M x 16B keys + (merge) + check the same 10M",
            "keys_per_rank": n, "key_bytes": 16, "m_bits": 2**28, "k": 7,
            "parallelism": f"key-range x{world}, replica per GPU, allreduce(OR)" if world > 1 else "single GPU",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "Bloom insert = k_part_scatter<KeysFixed16,IdxBloom<pow2>,PayNone,SpillBloomOr,7> + k_bloom_apply "
                      "(one insert launch = both kernels; avg_kernel_ms is their sum between two HIP events)",
            "achieved": ach,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; bytes per insert launch)" if traffic else None,
            "algorithmic_bytes_per_launch": n * BYTES["bloom_insert"],
            "algorithmic_bytes_per_key": BYTES["bloom_insert"],
            "avg_kernel_ms": ins_ms,
            "limiter": "pass 1 is co-limited by VALU (the k FNV-1a chains alone are 75 us of its 165 us: scripts/ablate.py dbg=2) "
                       "and the LDS counting sort, not by HBM (430 MB in 165 us); pass 2 streams at ~5.5 TB/s",
        },
        "detail": {
            "insert_Mkeys_s": n / ins_ms / 1e3,
            "check_Mkeys_s": None if overlapped else n / chk_ms / 1e3,
            "check_GBs": None if overlapped else n * BYTES["bloom_check"] / chk_ms / 1e6,
            "merge_plus_check_ms": mc_ms,  # N > 1: allreduce(OR) with the lookup's pass 1 running under it, then pass 2
            "clear_ms": timer.mean_ms("clear"),
            "merge_ms": timer.mean_ms("merge") if distributed and not overlapped else None,
            "all_inserted_found": ok,
            "merged_table_equals_single_stream": merged_ok,
            "bits_set": bits_set,
        },
    }
    if rank == 0 and world == 1 and not args.no_detail:
        line["detail"].update(side_measurements(dev, n))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(min(n, 10_000_000))
    elif rank == 0:
        line["cpu_baseline"] = None
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line must be the LAST thing on stdout: RCCL's NCCL_DEBUG=VERSION banner sits in the C stdio buffer
        # until exit, so drain C stdio first (the other ranks' stdout was sent to /dev/null at start)
        import ctypes  # noqa: PLC0415

        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
    if not ok:
        raise SystemExit("parity property violated: an inserted key was not found")
    if merged_ok is False:
        raise SystemExit("parity property violated: the merged table differs from the single-stream filter")


if __name__ == "__main__":
    main()
