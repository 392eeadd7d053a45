#!/usr/bin/env python3
"""bench.py -- benchmarks of the sketch engine on MI355X (one JSON line on stdout, printed by rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg5]

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts its own ranks (one process per GPU,
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...``); launched under
torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.

Configurations (BASELINE.json `configs`); inputs are generated on the device (counter-based splitmix64
stream) BEFORE the timed region, i.e. they are resident in HBM when the clock starts:

  cfg2 (default, the headline)  BloomFilter(28005615, 0.01) -> m = 2^28 bits, k = 7.  One STEP per rank = clear the filter,
        batch-insert 10M 16-byte keys, (N > 1: allreduce(OR) over RCCL), batch-check the same 10M keys.
        `value` = all ranks' key operations (insert + lookup) per second.  Weak scaling: every rank owns 10M keys.
  cfg3  CountMinSketch(width=2^20, depth=5).  STEP = clear + 100M weighted updates as 10 passes over 10M keys
        (+ SUM merge for N > 1).  `value` = updates per second.
  cfg4  CountingBloomFilter(28005615, 0.01) -> 2^28 x uint32 = 1 GiB.  STEP = clear + the 50-batch mixed stream: batch b
        adds the 1M keys [bB, (b+1)B) and (b >= 1) removes the first B/2 keys of batch b-1 -- 74.5M operations in
        1M-key batches.  `value` = operations per second.
  cfg5  BloomFilter(224044920, 0.01) -> m = 2^31 bits (256 MiB per replica).  N_total keys (default 10^9) are split by
        key range over the ranks; STEP = clear + insert the shard + allreduce(OR) + check the shard.  Strong scaling.

The default run (--gpus 1, cfg2) ALSO executes short steps of cfg3 (as written: 100M updates in 10 passes), cfg4 and cfg5
(N = 1) after the headline's timed region and reports them under `configs.cfg3|cfg4|cfg5` (own ms_per_step, roofline,
parity flags), so that the driver's one line covers the whole of BASELINE.json's metric (--no-extra-configs skips them).

Failure handling: every rank arms a watchdog (--timeout seconds); a rank that fails or hangs still makes rank 0 (or the
self-launching parent) print ONE JSON line with `"rc" != 0` and `"error"`, never a silent hang.

Per-rank memory, `--config cfg5 --gpus 8`: 125M keys x 16 B = 2.0 GB of resident keys + the 256 MiB replica + ~2.6 GB of
partition scratch per 2^25-key chunk (bucket buffer ~1.1 GB insert / ~1.9 GB lookup, segment counts) + 0.5 GB of merge
buffers: < 6 GB of the 288 GB.  At N = 1 the 10^9 keys are 16 GB.

Every line carries
  roofline      the dominant kernel of the configuration (cfg2: the Bloom insert launch = k_part_bins + k_bloom_apply):
                ALGORITHMIC bytes (72 B per inserted key, 45 per Bloom lookup, 60 per CMS update, ...) divided
                by the launch time measured with HIP events on the launch stream, against the 8 TB/s HBM peak.
  rooflines     the same object for the other operations of the metric (Bloom check, CMS add, CMS / CBF lookups ...).
  cpu_baseline  the plain-C oracle (a port of the reference semantics) on this box's host cores: 1 thread, all cores
                (per-thread replica + OR merge), and a pure-Python mirror of the reference's per-key loop on a 100k-key
                sample (N = 1, rank 0, bounded to ~20 s of host work).
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time
import traceback
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# dmabuf IPC: RCCL across processes needs it on this driver.  Set here, before torch / the HSA runtime start, so that BOTH launch
# forms (python bench.py --gpus N, and python -m torch.distributed.run ... bench.py) run their ranks with the same environment.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SEED = 0x5EED
# algorithmic bytes per unit of work (L = 16-byte key, k = 7, d = 5)
BYTES = {"bloom_insert": 72, "bloom_check": 45, "cms_add": 60, "cms_check": 40, "cbf_add": 76, "cbf_remove": 76, "cbf_check": 48,
         "bloom_step": 72 + 45}  # one key inserted AND looked up: what one unit of cfg 2 / cfg 5's `value` moves
DEFAULT_STEPS = {"cfg2": (200, 20), "cfg3": (20, 3), "cfg4": (10, 2), "cfg5": (5, 1)}
PMC_FILE = next((f for f in (ROOT / "profiles" / f"r0{r}_pmc_traffic.json" for r in (6, 5, 4, 3)) if f.exists()), ROOT / "profiles" / "r06_pmc_traffic.json")
L2_FILE = next((f for f in (ROOT / "profiles" / f"r0{r}_l2_hit.json" for r in (6, 5, 4, 3)) if f.exists()), ROOT / "profiles" / "r06_l2_hit.json")
METRIC_CFG2 = "million keys/sec insert+lookup (Bloom m=2^28 k=7, CMS 2^20x5)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(DEFAULT_STEPS), default="cfg2")
    ap.add_argument("--n", "--keys-per-rank", dest="n", type=int, default=10_000_000, help="cfg2/cfg3: keys per rank per step")
    ap.add_argument("--n-total", type=int, default=1_000_000_000, help="cfg5: keys in the whole stream (all ranks)")
    ap.add_argument("--batch", type=int, default=1_000_000, help="cfg4: keys per batch")
    ap.add_argument("--batches", type=int, default=50, help="cfg4: batches per step")
    ap.add_argument("--spinup", type=float, default=1.0, help="seconds of untimed steps before warm-up (clock ramp)")
    ap.add_argument("--repeats", type=int, default=5, help="the block of --steps timed steps is run this many times, each between its own fences; "
                    "`ms_per_step` / `value` are the MEDIAN block's, min / max are reported next to it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--print-detail", action="store_true", help="also print the FULL result object (the one written to gpurun_out/bench_detail.json) on stderr")
    ap.add_argument("--no-detail", action="store_true", help="skip the CMS / CBF / GUPS side measurements")
    ap.add_argument("--no-combine", action="store_true", help="cfg4: apply every 1M-key batch at once (update windows off: psk_set_option update_window=0)")
    ap.add_argument("--legacy-combine", action="store_true", help="cfg4: the round-2/3 opt-in (combine_updates=True: removes are plain decrements applied after "
                    "the window's adds -- exact for well-formed streams only); default is the plain API, whose update windows are exact for any stream")
    ap.add_argument("--borrow-keys", action="store_true", help="cfg4: the resident, never overwritten key batches are BORROWED (PSK_DEVICE_BORROWED): the "
                    "engine keeps pointers and hashes them where they lie at the flush, instead of copying every batch into its key lists "
                    "(same throughput, no 2 x 1.25 GiB of lists)")
    ap.add_argument("--borrow-window", action="store_true", help="cfg4: the default API with CountingBloomFilter(borrow_keys=True): the update window keeps the "
                    "resident, never overwritten key batches where they are instead of copying them (exact for any stream, like the default)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="psk_set_option before the run (A/B of engine tunables)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: merge, then look up (no lookup pass 1 under the merge)")
    ap.add_argument("--no-extra-configs", action="store_true", help="default run: skip the short cfg3 / cfg4 / cfg5 steps reported under `configs`")
    ap.add_argument("--timeout", type=float, default=900.0, help="watchdog: seconds after which a rank gives up (JSON error line, rc 124)")
    ap.add_argument("--init-timeout", type=float, default=180.0, help="N > 1: seconds allowed for the process-group rendezvous / RCCL init")
    args = ap.parse_args()
    ds, dw = DEFAULT_STEPS[args.config]
    args.steps = ds if args.steps is None else args.steps
    args.warmup = dw if args.warmup is None else args.warmup
    return args


# Which part of the run this rank is in -- a failure names it: {"rc": ..., "error": "rank 3 failed in phase 'merge (allreduce OR)': ..."}.
# Every rank that fails also leaves a small file next to the rendezvous (one node: /tmp is shared), so that whoever prints the JSON line
# -- rank 0, or the self-launching parent when rank 0 is the one that died -- can say WHICH rank failed WHERE, not just "a collective hung".
PHASE = {"name": "start"}


def phase(name: str) -> None:
    PHASE["name"] = name


def _err_dir() -> Path:
    return Path(os.environ.get("TMPDIR", "/tmp")) / f"psk_bench_{os.environ.get('MASTER_PORT', 'single')}"


def leave_error_note(rank: int, text: str) -> None:
    try:
        d = _err_dir()
        d.mkdir(parents=True, exist_ok=True)
        (d / f"rank{rank}.json").write_text(json.dumps({"rank": rank, "phase": PHASE["name"], "error": text[-1500:]}))
    except OSError:
        pass


def collect_error_notes(skip_rank=None) -> str:
    notes = []
    try:
        for f in sorted(_err_dir().glob("rank*.json")):
            n = json.loads(f.read_text())
            if n.get("rank") != skip_rank:
                notes.append(f"rank {n['rank']} failed in phase '{n['phase']}': {n['error']}")
            f.unlink()
    except (OSError, ValueError):
        pass
    return " || ".join(notes)


def error_line(args, rc: int, error: str) -> str:
    """the JSON line of a run that failed: same keys as a good line, value null, rc != 0"""
    return json.dumps({"metric": METRIC_CFG2 if args.config == "cfg2" else f"bench {args.config}", "value": None, "unit": None, "n_gpus": args.gpus,
                       "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "rc": rc, "error": error[-2000:]})


def self_launch(args) -> None:
    """`python bench.py --gpus N` with N > 1 outside torch.distributed.run: start the N ranks ourselves.  The ranks' stdout
    is passed through; if they fail, hang past --timeout or end without a JSON line, ONE JSON error line is printed."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    timer = threading.Timer(args.timeout + 60.0, lambda: os.killpg(proc.pid, 9))  # (the ranks' own watchdogs fire first)
    timer.daemon = True
    timer.start()
    out = proc.stdout.read()
    rc = proc.wait()
    timer.cancel()
    sys.stdout.write(out)
    last = out.strip().splitlines()[-1] if out.strip() else ""
    has_json = last.startswith("{") and last.endswith("}")
    if rc != 0 and not has_json:
        os.environ["MASTER_PORT"] = str(port)  # (where the ranks left their notes)
        notes = collect_error_notes()
        print(error_line(args, rc, f"the ranks ended with rc {rc} and no result line" + (" (killed by the launcher's watchdog)" if rc in (-9, 137) else "")
                         + (f" | {notes}" if notes else "")), flush=True)
    sys.exit(rc if rc >= 0 else 128 - rc)


# ----------------------------------------------------------------------------------------------- plumbing
class Ctx:
    """rank / device / process group of this process"""

    def __init__(self, args):
        import torch

        self.torch = torch
        self.rank = int(os.environ.get("RANK", 0))
        self.world = int(os.environ.get("WORLD_SIZE", 1))
        local_rank = int(os.environ.get("LOCAL_RANK", 0))
        if self.rank != 0:  # only rank 0 reports; keep library banners of the other ranks off the job's stdout
            os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
        if args.gpus != self.world and self.world > 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        # PSK_BENCH_SINGLE_DEVICE=1 (test hook): every rank uses cuda:0 and the gloo backend, so that the N > 1 logic (shard
        # offsets, real replicas, merge, overlap) can be driven on a one-GPU box; RCCL refuses two ranks on one device
        self.single_device = bool(os.environ.get("PSK_BENCH_SINGLE_DEVICE"))
        if self.single_device:
            local_rank = 0
        self.dev = local_rank
        torch.cuda.set_device(local_rank)
        self.dist = None
        # PSK_BENCH_FORCE_DIST=1 drives the whole N > 1 code path (RCCL init, merge, barriers) with a single rank
        self.distributed = self.world > 1 or bool(os.environ.get("PSK_BENCH_FORCE_DIST"))
        if self.distributed:
            import torch.distributed as dist

            if self.world == 1:
                os.environ.setdefault("MASTER_PORT", "29533")
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
                os.environ["PSK_FORCE_MERGE_PATH"] = "1"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            from datetime import timedelta

            phase("rendezvous (init_process_group)")
            tmo = timedelta(seconds=args.init_timeout)  # rendezvous + every collective: a dead rank raises instead of hanging
            if self.single_device:
                dist.init_process_group("gloo", timeout=tmo)
            else:
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), timeout=tmo)
            self.dist = dist

    def fence(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds: float) -> float:
        if self.dist is None:
            return seconds
        te = self.torch.tensor([seconds], dtype=self.torch.float64, device=f"cuda:{self.dev}")
        self.dist.all_reduce(te, op=self.dist.ReduceOp.MAX)
        return float(te.item())

    def stream(self):
        return self.torch.cuda.current_stream(self.dev).cuda_stream or None

    def gen_keys(self, n, start):
        from pyprobables_amd import _native as N

        t = self.torch.empty((n, 16), dtype=self.torch.uint8, device=f"cuda:{self.dev}")
        N.check(N.lib().psk_gen_keys16(t.data_ptr(), start, n, SEED, self.dev, self.stream()))
        return t

    def gen_weights(self, n, start):
        from pyprobables_amd import _native as N

        t = self.torch.empty(n, dtype=self.torch.int32, device=f"cuda:{self.dev}")
        N.check(N.lib().psk_gen_weights(t.data_ptr(), start, n, SEED, self.dev, self.stream()))
        return t


class EventTimer:
    """HIP events on torch's current stream == the stream the engine launches on"""

    def __init__(self, torch):
        self.torch, self.pairs = torch, {}

    def time(self, name, fn):
        a, b = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.pairs.setdefault(name, []).append((a, b))
        return r

    def mean_ms(self, name):
        p = self.pairs.get(name, [])
        return sum(a.elapsed_time(b) for a, b in p) / len(p) if p else float("nan")

    def median_ms(self, name):
        """the median pair: one sample that caught a host hiccup (the GPU idling inside the event window) does not move it"""
        t = sorted(a.elapsed_time(b) for a, b in self.pairs.get(name, []))
        if not t:
            return float("nan")
        return t[len(t) // 2] if len(t) % 2 else 0.5 * (t[len(t) // 2 - 1] + t[len(t) // 2])


def timed_loop(torch, fn, iters, warm=2):
    for _ in range(warm):  # synchronised warm-up calls: scratch buffers exist and the engine's adaptive choices (Bloom lookup
        fn()               # scheme follows the previous call's miss tally) have settled before the clock starts
        torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def pmc_traffic(op: str, n: int):
    """HBM-side bytes per launch of `op` from the committed PMC profile of the same workload and kernels
    (scripts/profile_r06_pmc.sh -> profiles/r06_pmc_traffic.json); None when there is no matching record.  A record measured
    on chunks of the same kernels (`per_key_scalable`: cfg 5 inserts its shard in 2^25-key calls) is scaled by the key count."""
    try:
        pmc = json.loads(PMC_FILE.read_text())
        rec = pmc[op]
        if rec.get("keys", pmc.get("keys")) == n:
            return rec["hbm_bytes_per_launch"]
        if rec.get("per_key_scalable"):
            return int(rec["bytes_per_key"] * n)
    except Exception:
        pass
    return None


def l2_hit(op: str):
    """L2 hit rate TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) of the launch's kernels (profiles/r05_l2_hit.json,
    scripts/profile_r05.sh: its own rocprofv3 --pmc pass); None when there is no record"""
    try:
        return json.loads(L2_FILE.read_text())[op]["l2_hit"]
    except Exception:
        return None


def roofline(op: str, kernel: str, units: int, ms: float, limiter: str, traffic_op: str | None = None):
    """roofline object of one launch: `units` units of work of BYTES[op] algorithmic bytes each in `ms` milliseconds"""
    ach = units * BYTES[op] / (ms * 1e-3) / 1e9 if ms == ms and ms > 0 else None
    traffic = pmc_traffic(traffic_op or op, units)
    return {
        "bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": None if ach is None else ach / HBM_PEAK_GBS, "traffic": traffic,
        "traffic_source": f"{PMC_FILE.relative_to(ROOT)} (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes; bytes per launch)" if traffic else None,
        "l2_hit": l2_hit(traffic_op or op),
        "algorithmic_bytes_per_launch": units * BYTES[op], "algorithmic_bytes_per_unit": BYTES[op], "avg_kernel_ms": ms,
        "limiter": limiter,
    }


def step_roofline(kernel: str, units: int, step_ms: float, launches: dict, limiter: str):
    """roofline object of a whole timed STEP (what `value` measures): `units` keys each inserted and looked up (BYTES["bloom_step"] algorithmic
    bytes) in `step_ms` -- HIP events around whole steps of the timed region, everything the step enqueues included (clear, merge) --
    with the objects of its launches nested under `launches` (each timed between its own events on other steps of the timed region)"""
    r = roofline("bloom_step", kernel, units, step_ms, limiter)
    tr = [v.get("traffic") for v in launches.values()]
    r["traffic"] = sum(tr) if tr and all(t is not None for t in tr) else None
    r["traffic_source"] = next((v["traffic_source"] for v in launches.values() if v.get("traffic_source")), None) if r["traffic"] else None
    r["l2_hit"] = {k: v.get("l2_hit") for k, v in launches.items()}
    r["launch_ms_sum"] = sum(v["avg_kernel_ms"] for v in launches.values())
    r["launches"] = launches
    return r


# ----------------------------------------------------------------------------------------------- CPU baseline


def host_cores():
    """threads the all-cores CPU legs use: the CPUs this process may run on, capped by the container's CPU-time quota (cgroup
    cpu.max).  On the GPU boxes the affinity mask shows 256 CPUs but the quota is 16 CPUs' worth of time: 256 threads then
    time-share 16 cores (measured: 44 Mkeys/s with 256 threads, 60-75 with 16-64; scripts/cpu_scaling.py)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"{aff} CPUs in the affinity mask"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            note += f", cgroup cpu.max = {q} CPUs"
            aff = min(aff, q)
    except (OSError, ValueError):
        pass
    return aff, note


def cpu_baseline(n_sample: int, reps: int = 2):
    """(i) a pure-Python mirror of the reference's per-key loop on a 100k-key sample, (ii) the plain-C
    oracle (kind 'port') on one core and on all host cores (per-thread replica + OR merge).  ~20 s of host work."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle
    import pymirror

    m, k = 2**28, 7
    # -- C port, one thread: `value`
    ob = oracle.OracleBloom(m, k)
    t_total, ops = 0.0, 0
    for r in range(reps):
        keys = oracle.gen_keys16(r * n_sample, n_sample)
        t0 = time.perf_counter()
        ob.add_keys(keys)
        res = ob.check_keys(keys)
        t_total += time.perf_counter() - t0
        ops += 2 * n_sample
        assert bool(res.all())
    one = {"value": ops / t_total / 1e6, "unit": "Mkeys/s", "cores": 1, "kind": "port", "seconds": t_total,
           "sample": f"{reps} x (insert {n_sample} + check {n_sample}) 16-byte keys into m=2^28 k=7, oracle/psk_oracle.c (gcc -O2), 1 thread"}
    # -- C port, all cores
    cores, quota_note = host_cores()
    n_mt = max(n_sample * 4, cores * 6_000_000)  # enough keys per thread for the merge of T replicas to amortise (a few seconds per leg)
    obm = oracle.OracleBloom(m, k)
    found = obm.insert_check_mt(0, n_mt, cores)
    t_mt = obm.mt_seconds                          # the three phases; replica allocation / first touch is outside the clock
    allc = {"value": 2 * n_mt / t_mt / 1e6, "unit": "Mkeys/s", "cores": cores, "kind": "port", "seconds": t_mt, "all_found": found == n_mt,
            "sample": f"insert {n_mt} + check {n_mt} keys, {cores} threads ({quota_note}): per-thread 32 MiB replica, OR merge, lookups (oracle/psk_oracle.c, key generation included)"}
    # -- C port, all cores, ONE shared table with relaxed atomic ORs (no replicas, no merge): the honest "all cores" figure
    obs = oracle.OracleBloom(m, k)
    found_s = obs.insert_check_mt_shared(0, n_mt, cores)
    t_sh = obs.mt_seconds
    shared = {"value": 2 * n_mt / t_sh / 1e6, "unit": "Mkeys/s", "cores": cores, "kind": "port", "seconds": t_sh, "all_found": found_s == n_mt,
              "table_equals_replica_variant": bool((obs.bloom == obm.bloom).all()),
              "sample": f"insert {n_mt} + check {n_mt} keys, {cores} threads ({quota_note}) on ONE shared 32 MiB table (__atomic_fetch_or, relaxed), lookups (oracle/psk_oracle.c, key generation included)"}
    # -- pure-Python mirror (what the reference's interpreted loop costs here; never the reference itself)
    n_py = 100_000
    mb = pymirror.MirrorBloom(m, k)
    pkeys = [pymirror.key16(i) for i in range(n_py)]
    t0 = time.perf_counter()
    for kx in pkeys:
        mb.add(kx)
    ok = all(mb.check(kx) for kx in pkeys)
    t_py = time.perf_counter() - t0
    py = {"value": 2 * n_py / t_py / 1e6, "unit": "Mkeys/s", "cores": 1, "kind": "python-mirror", "seconds": t_py, "all_found": ok,
          "sample": f"insert {n_py} + check {n_py} keys through oracle/pymirror.py (interpreted per-key FNV-1a + add_alt/check_alt, bigint arithmetic like the reference)"}
    return {**one, "legs": {"port_1core": dict(one), "port_allcores": allc, "port_allcores_shared_table": shared, "python_mirror": py}}


def cpu_baseline_cfg3(n_sample: int = 4_000_000):
    """cfg 3 on the host: the plain-C oracle's CountMinSketch (countminsketch.py:257-288 restated) on a sample of the
    same weighted stream -- one thread, and all cores as per-thread replicas summed at the end (exact while no bin reaches a rail:
    cfg 3's weights are 1 .. 7).  Bounded to a few seconds."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import numpy as np
    import oracle

    keys, w = oracle.gen_keys16(0, n_sample), oracle.gen_weights(0, n_sample)
    oc = oracle.OracleCMS(2**20, 5)
    t0 = time.perf_counter()
    oc.add_keys(keys, w)
    t1 = time.perf_counter() - t0
    one = {"value": n_sample / t1 / 1e6, "unit": "Mupdates/s", "cores": 1, "kind": "port", "seconds": t1,
           "sample": f"{n_sample} weighted updates (16-byte keys, weights 1..7) into width=2^20 depth=5, oracle/psk_oracle.c (gcc -O2), 1 thread"}
    t0 = time.perf_counter()
    oc.check_keys(keys[: n_sample // 2])
    t2 = time.perf_counter() - t0
    one["check_Mkeys_s"] = n_sample // 2 / t2 / 1e6
    cores, quota_note = host_cores()
    per = max(1, (n_sample * 4) // cores)
    reps = [oracle.OracleCMS(2**20, 5) for _ in range(cores)]
    parts = [(oracle.gen_keys16(t * per, per), oracle.gen_weights(t * per, per)) for t in range(cores)]  # (generation outside the clock)
    t0 = time.perf_counter()
    ths = [threading.Thread(target=r.add_keys, args=p) for r, p in zip(reps, parts)]  # (ctypes drops the GIL inside the C call)
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    total = np.zeros(reps[0].bins.size, dtype=np.int64)
    for r in reps:
        total += r.bins  # (the merge of the replicas is part of the leg)
    t3 = time.perf_counter() - t0
    allc = {"value": per * cores / t3 / 1e6, "unit": "Mupdates/s", "cores": cores, "kind": "port", "seconds": t3,
            "sum_of_bins_ok": int(total.sum()) == 5 * int(sum(int(p[1].sum()) for p in parts)),
            "sample": f"{per * cores} weighted updates, {cores} threads ({quota_note}): per-thread 20 MiB replica + int64 sum of the replicas (exact below the rails)"}
    return {**one, "legs": {"port_1core": dict(one), "port_allcores": allc}}


def cpu_baseline_cfg4(batch: int = 1_000_000, nbatches: int = 3):
    """cfg 4 on the host: the plain-C oracle's CountingBloomFilter (countingbloom.py:135-208 restated) on the first batches of the same
    mixed stream into the same 2^28 x uint32 table (1 GiB), one thread.  The stream does not parallelise as it stands -- a remove must see
    the adds before it -- so there is no all-cores leg."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import numpy as np
    import oracle

    keys = oracle.gen_keys16(0, batch * nbatches)
    oc = oracle.OracleCBF(2**28, 7)
    oc.bloom[:] = 0  # (first touch of the 1 GiB outside the clock)
    ops = 0
    t0 = time.perf_counter()
    for b in range(nbatches):
        oc.update_keys(keys[b * batch:(b + 1) * batch])
        ops += batch
        if b >= 1:
            oc.update_keys(keys[(b - 1) * batch:(b - 1) * batch + batch // 2], -np.ones(batch // 2, dtype=np.int64))
            ops += batch // 2
    t1 = time.perf_counter() - t0
    return {"value": ops / t1 / 1e6, "unit": "Mops/s", "cores": 1, "kind": "port", "seconds": t1, "elements_added": oc.els_added,
            "sample": f"the first {nbatches} batches of the stream ({ops} operations: add {batch} keys, remove the first half of the previous batch) into the "
                      "2^28 x uint32 table, oracle/psk_oracle.c (gcc -O2), 1 thread; the stream is order-dependent: no all-cores leg"}


# ----------------------------------------------------------------------------------------------- cfg2
class Cfg2:
    """headline: Bloom m = 2^28, k = 7; clear + insert 10M + (merge) + check 10M per rank per step"""

    def __init__(self, ctx: Ctx, args):
        import pyprobables_amd as pa
        from pyprobables_amd import parallel

        self.ctx, self.args, self.pa, self.parallel = ctx, args, pa, parallel
        self.n = args.n
        self.keys = ctx.gen_keys(self.n, ctx.rank * self.n)  # rank r owns keys [r*n, (r+1)*n): resident before timing
        self.blm = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=ctx.dev)
        assert self.blm.number_bits == 2**28 and self.blm.number_hashes == 7
        self.timer = EventTimer(ctx.torch)
        self.state = {}
        self.ops_per_step = 2 * self.n * ctx.world

    # A timing-enabled HIP event is a barrier packet: the next kernel cannot be dispatched ahead of it, ~7 us of bubble
    # each.  The timed steps therefore carry only the ONE event pair the roofline needs (around the insert launch), and
    # only on a sample of the steps; the other phases are timed in a separate instrumented pass after the timed region.
    def step(self, record=0, every_phase=False):
        """record: 0 no events; 1 one event pair around the WHOLE step (-> roofline of the step); 2 events between the launches (-> their
        nested objects) -- on different steps, so that the whole-step sample carries no event bubbles inside"""
        plain = lambda _n, f: f()  # noqa: E731
        t = self.timer.time if (record and every_phase) else plain
        ti = self.timer.time if (record == 2 or (record and every_phase)) else plain
        blm, keys, ctx = self.blm, self.keys, self.ctx
        if record == 1 and not every_phase:
            return self.timer.time("step", lambda: self.step(0))
        t("clear", blm.clear)
        ti("insert" if not every_phase else "insert_detail", lambda: blm.add_many(keys))
        if ctx.distributed and not self.args.no_overlap:
            # merge on a side stream; pass 1 of the lookup (hash + partition: it never reads the table) runs under it
            def merge_and_check():
                h = self.parallel.merge_bloom_async(blm)
                blm.check_many_begin(keys)
                h.wait()
                return blm.check_many_finish()

            self.state["res"] = ti("merge+check", merge_and_check)
        else:
            if ctx.distributed:
                ti("merge", lambda: self.parallel.merge_bloom(blm, sync_elements=False))  # table merge only: no host sync
            self.state["res"] = ti("check" if not every_phase else "check_detail", lambda: blm.check_many(keys))

    def instrumented(self):
        for _ in range(min(self.args.steps, 10)):
            self.step(True, every_phase=True)

    def rank_times(self):
        """(insert ms, merge ms, lookup ms, table bytes) of ONE un-overlapped step on this rank (multi_gpu_diagnostics)"""
        torch, blm, keys = self.ctx.torch, self.blm, self.keys
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        blm.clear()
        ev[0].record()
        blm.add_many(keys)
        ev[1].record()
        self.parallel.merge_bloom(blm, sync_elements=False)
        ev[2].record()
        blm.check_many(keys)
        ev[3].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3]), 2**28 // 8

    def predicted(self, body):
        """the weak-scaling curve of cfg 2 from this run's per-key times (per-rank figures of the un-overlapped diagnostic step when N > 1)"""
        mg, d = body.get("multi_gpu"), body["detail"]
        if mg:
            ins = max(p["insert_ms"] for p in mg["per_rank"])
            chk = max(p["check_ms"] for p in mg["per_rank"])
        else:
            ins, chk = self.n / d["insert_Mkeys_s"] / 1e3, self.n / d["check_Mkeys_s"] / 1e3
        return predicted_scaling("weak", 2**28 // 8, ins / (self.n / 1e6), chk / (self.n / 1e6), self.n, d.get("clear_ms") or 0.0, 0.68, _worst_merge(body))

    def finish(self, ms_step):
        ctx, args, n, torch, timer, blm = self.ctx, self.args, self.n, self.ctx.torch, self.timer, self.blm
        ok = bool(self.state["res"].all().item())  # every inserted key must be found (size-independent parity property)
        bits_set = blm._cnt_number_bits_set()
        merged_ok = None
        if ctx.distributed:
            # multi-GPU parity, outside the timed region: the merged replica must equal ONE filter fed every rank's keys
            ref = self.pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=ctx.dev)
            for r in range(ctx.world):
                ref.add_many(ctx.gen_keys(n, r * n))
            merged_ok = bool(torch.equal(ref.table_tensor, blm.table_tensor))
            del ref
        overlapped = ctx.distributed and not args.no_overlap
        ins_ms, chk_ms = timer.mean_ms("insert"), timer.mean_ms("merge+check" if overlapped else "check")
        detail = {
            "insert_Mkeys_s": n / ins_ms / 1e3,
            "check_Mkeys_s": None if overlapped else n / chk_ms / 1e3,
            "merge_plus_check_ms": timer.mean_ms("merge+check") if overlapped else None,  # allreduce(OR) with lookup pass 1 under it, then pass 2
            "clear_ms": timer.mean_ms("clear"),
            "merge_ms": timer.mean_ms("merge") if ctx.distributed and not overlapped else None,
            "all_inserted_found": ok, "merged_table_equals_single_stream": merged_ok, "bits_set": bits_set,
        }
        launches = {"bloom_insert": roofline(
            "bloom_insert", "Bloom insert = k_part_bins<KeysFixed16,IdxBloom<pow2>,PayNone,SpillBloomOr,7> + k_bloom_apply "
            "(one insert launch = both kernels; avg_kernel_ms is their sum between two HIP events)", n, ins_ms,
            "pass 1 is VALU bound (SQ counters: a VALU instruction issued in ~96 % of the SIMD cycles, 2 of 3 of them the k FNV-1a chains), not HBM bound; pass 2 streams at ~5 TB/s")}
        if not overlapped:
            launches["bloom_check"] = roofline(
                "bloom_check", "Bloom lookup of present keys = k_part_bins<KeysFixed16,IdxBloom<pow2>,PayTileTag,SpillBloomFlag,7> + k_bloom_test_flag + "
                "k_bloom_flag_resolve (tile flags; keyed probes / return trip for batches with absent keys: detail.check_*_fresh)",
                n, chk_ms, "pass 1 is the insert's (hash + LDS counting sort, 2.67-byte probes); pass 2 streams the probes back "
                "from the Infinity Cache against an LDS-resident slice")
        rooflines = {k: v for k, v in launches.items() if k != "bloom_insert"}
        if ctx.rank == 0 and ctx.world == 1 and not args.no_detail:
            side, rl = side_measurements(ctx, n, blm, self.keys)
            detail.update(side)
            rooflines.update(rl)
        step_ms = timer.median_ms("step")
        detail["step_ms_events"] = step_ms
        detail["step_ms_events_mean"] = timer.mean_ms("step")
        detail["step_event_samples"] = len(timer.pairs.get("step", []))
        line = {
            "metric": METRIC_CFG2,
            "config": {
                "workload": "cfg2: BloomFilter(est_elements=28005615, fpr=0.01): m=2^28 bits, k=7, default_fnv_1a; per rank per step: "
                            "clear + insert 10M x 16B keys + (merge) + check the same 10M",
                "keys_per_rank": n, "key_bytes": 16, "m_bits": 2**28, "k": 7,
                "parallelism": f"key-range x{ctx.world}, replica per GPU, allreduce(OR)" if ctx.world > 1 else "single GPU",
            },
            "roofline": step_roofline(
                "one timed STEP = clear + Bloom insert (k_part_bins + k_bloom_apply) + " + ("allreduce(OR) + " if ctx.distributed else "") +
                "Bloom lookup (k_part_bins + k_bloom_test_flag + k_bloom_flag_resolve); avg_kernel_ms = HIP events around whole steps of the timed region (median pair; detail.step_ms_events_mean)",
                n, step_ms, launches,
                "both pass 1s are VALU bound (the k FNV-1a chains are 2 of 3 of their instructions: profiles/r06_sq_pass1.txt), not HBM bound; the pass 2s "
                "stream the probes back at 4-5 TB/s"),
            "rooflines": rooflines, "detail": detail,
        }
        fails = []
        if not ok:
            fails.append("parity property violated: an inserted key was not found")
        if merged_ok is False:
            fails.append("parity property violated: the merged table differs from the single-stream filter")
        return line, fails


def _worst_merge(body):
    mg = body.get("multi_gpu")
    return {"world": mg["ranks_seen_by_rccl"], "merge_ms": max(p["merge_ms"] for p in mg["per_rank"])} if mg else None


def side_measurements(ctx: Ctx, n, blm, keys):
    """lookups with misses, CMS / CBF rates and the GUPS-style random-access ceilings (not part of `value`)"""
    import pyprobables_amd as pa
    from pyprobables_amd import _native as N

    torch, dev = ctx.torch, ctx.dev
    out, rl = {}, {}
    # Host buffers (SURVEY.md 8d / BASELINE.md 4: the copy reported separately, never `value`): the same 10 M keys handed over as a numpy
    # array -- H2D copy over PCIe + insert / lookup, the call returns when the result is back
    hk = keys.cpu().numpy()
    hb = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=dev)
    hb.add_many(hk[: n // 10])  # (staging buffers exist)
    hb.check_many(hk[: n // 10])
    t0 = time.perf_counter()
    hb.add_many(hk)
    t_in = time.perf_counter() - t0
    t0 = time.perf_counter()
    hres = hb.check_many(hk)
    t_ck = time.perf_counter() - t0
    out["bloom_host_buffers_Mkeys_s"] = {"insert": n / t_in / 1e6, "check": n / t_ck / 1e6, "all_found": bool(hres.all()),
                                         "note": "PCIe inclusive: numpy keys in, results back on the host; not part of `value`"}
    del hb, hk, hres
    # Bloom lookups that MISS (the timed step only looks up inserted keys: every probe hits).  blm holds keys [0, n).
    fresh = ctx.gen_keys(n, 10 * n)
    mixed = torch.cat([keys[: n // 2], fresh[: n - n // 2]])
    # (all absent: the automatic choice goes tile flags -> return trip -> lazy gathers within three calls; half absent: the return trip)
    ms = timed_loop(torch, lambda: blm.check_many(fresh), 5, warm=4)
    out["check_all_fresh_Mkeys_s"] = n / ms / 1e3
    ms = timed_loop(torch, lambda: blm.check_many(mixed), 5, warm=4)
    out["check_half_fresh_Mkeys_s"] = n / ms / 1e3
    res = blm.check_many(mixed)
    out["check_half_fresh_hits"] = int(res.sum().item())          # n/2 true members + the false positives among the fresh half
    out["check_half_fresh_members_found"] = bool(res[: n // 2].all().item())
    del fresh, mixed, res
    # membership as a ballot bitmap + hit count (psk_bloom_check_bits): the partitioned lookup's bytes packed by one streaming pass
    bits, hits = blm.check_many_bits(keys)
    out["check_bits_all_found"] = bool(int(hits.item()) == n)
    ms = timed_loop(torch, lambda: blm.check_many_bits(keys), 5, warm=4)
    out["check_bits_Mkeys_s"] = n / ms / 1e3
    del bits, hits
    # The reference's NATIVE key type: variable-length byte / str keys (hashes.py:98 walks the key element by element).  A ragged batch
    # handed over as (blob, offsets) on the device: lengths 4 + min(36, floor(Exp(12.6))) bytes (4 .. 40, mean ~15.4).
    import numpy as np

    rng = np.random.default_rng(7)
    lens = 4 + np.minimum(36, np.floor(rng.exponential(12.6, n))).astype(np.int64)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    pair = (torch.from_numpy(rng.integers(0, 256, int(offs[-1]), dtype=np.uint8)).to(f"cuda:{dev}"), torch.from_numpy(offs).to(f"cuda:{dev}"))
    rb = pa.BloomFilter(est_elements=28005615, false_positive_rate=0.01, device=dev)
    for _ in range(40):  # (the batch was made on the host: the clocks have dropped meanwhile)
        rb.add_many(pair)
    ms_a = timed_loop(torch, lambda: rb.add_many(pair), 5)
    ms_c = timed_loop(torch, lambda: rb.check_many(pair), 5)
    out["bloom_ragged_keys_Mkeys_s"] = {"insert": n / ms_a / 1e3, "check": n / ms_c / 1e3, "all_found": bool(rb.check_many(pair).all().item()),
                                        "mean_key_bytes": float(lens.mean()), "max_key_bytes": int(lens.max()),
                                        "note": "ragged byte keys (the reference's native key type) as a device (blob, offsets) pair into the cfg-2 "
                                                "filter: 16-byte windows per lane + a per-tile length sort in pass 1; not part of `value`"}
    del rb, pair, lens, offs
    w = ctx.gen_weights(n, 0)
    cms = pa.CountMinSketch(width=2**20, depth=5, device=dev)
    ms = timed_loop(torch, lambda: cms.add_many(keys, w), 5)
    out["cms_add_Mupd_s"] = n / ms / 1e3
    rl["cms_add"] = roofline("cms_add", "CMS weighted add = k_part_bins<...,IdxCms<pow2>,PayWeightSmall,...,5> (weights 0..15 as 20-bit fields; k_part_scatter<PayWeight> otherwise) (sums the weights) + k_tally_fold + k_counter_apply",
                             n, ms, "pass 1 (5 FNV-1a chains + LDS counting sort), pass 2 folds 2^15-cell slices", "cms_add_weighted")
    ms = timed_loop(torch, lambda: cms.check_many(keys), 5)
    out["cms_check_Mkeys_s"] = n / ms / 1e3
    rl["cms_check"] = roofline("cms_check", "CMS lookup (min over 5 rows)", n, ms, "see DESIGN.md 3.2")
    del cms
    ncbf = min(n, 10_000_000)
    cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01, device=dev)  # 2^28 x u32 = 1 GiB
    ms = timed_loop(torch, lambda: cbf.add_many(keys[:ncbf]), 3, warm=1)
    out["cbf_add_Mops_s"] = ncbf / ms / 1e3
    rl["cbf_add"] = roofline("cbf_add", "CBF unit add into the 1 GiB table (one level of 2^18-counter nibble-delta slices: k_part_scatter + k_nib_apply_pipe<0>, the pipelined pass over the table)",
                             ncbf, ms, "the fold read-modify-writes the whole 1 GiB table")
    from pyprobables_amd import _native as _N
    _N.set_option("cbf_lookup_shadow", 0)  # `cbf_check` = a lookup of a table that has just changed: the whole 32-bit table is read
    try:
        ms = timed_loop(torch, lambda: cbf.check_many(keys[:ncbf]), 3, warm=1)
    finally:
        _N.set_option("cbf_lookup_shadow", 1)
    out["cbf_check_Mkeys_s"] = ncbf / ms / 1e3
    rl["cbf_check"] = roofline("cbf_check", "CBF lookup (min over 7 counters, 1 GiB table): k_part_scatter<PayBloomLookup> + k_nib_gather + k_nib_collect",
                               ncbf, ms, "pass 2 streams the whole 1 GiB table (107 B per key at 10 M keys) into 4-bit slice images")
    # read-mostly: from the third lookup in a row of an unchanged table on, pass 2 loads the 4-bit images the second one left behind (128 MiB)
    ms = timed_loop(torch, lambda: cbf.check_many(keys[:ncbf]), 5, warm=3)
    out["cbf_check_unchanged_table_Mkeys_s"] = ncbf / ms / 1e3
    rl["cbf_check_unchanged_table"] = roofline("cbf_check", "CBF lookup of an unchanged 1 GiB table: k_part_scatter<PayBloomLookup> + k_nib_gather (kept 4-bit images) + k_nib_collect",
                                               ncbf, ms, "pass 2 loads the kept images (cells / 2 bytes) instead of the table", "cbf_check_unchanged_table")
    rm = EventTimer(torch)  # every remove needs its keys back in first: add (untimed), remove (timed), the first pair is warm-up
    for it in range(4):
        cbf.add_many(keys[:ncbf])
        (rm.time if it else (lambda _n, f: f()))("remove", lambda: cbf.remove_many(keys[:ncbf]))
    torch.cuda.synchronize()
    ms = rm.mean_ms("remove")
    out["cbf_remove_Mops_s"] = ncbf / ms / 1e3
    rl["cbf_remove"] = roofline("cbf_remove", "validated CBF remove of present keys, 1 GiB table: k_part_scatter + k_nib_apply_pipe<3> (optimistic decrement, one pipelined pass over the table)",
                                ncbf, ms, "the pass over the table (2 GiB read + written) whatever the batch brings")
    del cbf
    # The reference's own interface -- one key per call, the value back at once (bloom.py:252, countminsketch.py:257, countingbloom.py:125):
    # microseconds per call in a python loop, one launch + a polled completion mailbox each (NOTES.md 3.5); latency, not part of `value`
    def per_call_us(fn, calls=2000):
        for _ in range(200):
            fn()
        t0 = time.perf_counter()
        for _ in range(calls):
            fn()
        return (time.perf_counter() - t0) / calls * 1e6

    key1 = "0123456789abcdef"
    cms1 = pa.CountMinSketch(width=2**20, depth=5, device=dev)
    cbf1 = pa.CountingBloomFilter(est_elements=1_000_000, false_positive_rate=0.01, device=dev)
    out["per_key_call_us"] = {"bloom_check": per_call_us(lambda: key1 in blm), "cms_add": per_call_us(lambda: cms1.add(key1)),
                              "cms_check": per_call_us(lambda: cms1.check(key1)), "cbf_add": per_call_us(lambda: cbf1.add(key1)),
                              "cbf_check": per_call_us(lambda: cbf1.check(key1)), "key": "16 characters"}
    del cms1, cbf1
    # random-access ceilings at the headline table size (2^23 words = 32 MiB) and at 1 GiB
    sink = torch.zeros(1, dtype=torch.int64, device=f"cuda:{dev}")
    nprobe = 7 * n
    for label, words in (("32MiB", 2**23), ("1GiB", 2**28)):
        tab = torch.zeros(words, dtype=torch.int32, device=f"cuda:{dev}")
        for op, name in ((0, "atomic_or"), (1, "atomic_add"), (2, "gather")):
            ms = timed_loop(torch, lambda: N.check(N.lib().psk_gups(tab.data_ptr(), words, nprobe, op, 12345, sink.data_ptr(), dev, ctx.stream())), 3, warm=1)
            out[f"gups_{name}_{label}_Gprobes_s"] = nprobe / ms / 1e6
        del tab
    return out, rl


# ----------------------------------------------------------------------------------------------- cfg3
class Cfg3:
    """CountMinSketch 2^20 x 5: 100M weighted updates per step as 10 passes over 10M keys"""

    PASSES = 10

    def __init__(self, ctx: Ctx, args):
        import pyprobables_amd as pa
        from pyprobables_amd import parallel

        self.ctx, self.args, self.pa, self.parallel = ctx, args, pa, parallel
        self.n = args.n
        base = ctx.rank * self.n * self.PASSES
        self.keys = ctx.gen_keys(self.n, ctx.rank * self.n)                       # update i uses key (i mod n) of the rank's range
        self.w = [ctx.gen_weights(self.n, base + p * self.n) for p in range(self.PASSES)]  # and weight w(i): 400 MB resident
        self.cms = pa.CountMinSketch(width=2**20, depth=5, device=ctx.dev)
        self.timer = EventTimer(ctx.torch)
        self.ops_per_step = self.n * self.PASSES * ctx.world

    def step(self, record=0, every_phase=False):
        record = record == 1
        t = self.timer.time if record else (lambda _n, f: f())
        self.cms.clear()
        for p in range(self.PASSES):
            if record and p == 1:
                t("add", lambda: self.cms.add_many(self.keys, self.w[p]))
            else:
                self.cms.add_many(self.keys, self.w[p])
        if self.ctx.distributed:
            self.parallel.merge_counters(self.cms)

    def instrumented(self):
        pass

    def finish(self, ms_step):
        ctx, torch, n, cms = self.ctx, self.ctx.torch, self.n, self.cms
        # parity properties over the whole stream (the cell-by-cell compare with the oracle is tests/test_gpu_fullsize.py):
        # every update lands once per row; elements_added is the sum of the weights
        wsum = sum(int(w.to(torch.int64).sum().item()) for w in self.w)
        if ctx.dist is not None:
            tt = torch.tensor([wsum], dtype=torch.int64, device=f"cuda:{ctx.dev}")
            ctx.dist.all_reduce(tt)
            wsum = int(tt.item())
        els = cms.elements_added
        total = int(cms.table_tensor[: 5 * 2**20].to(torch.int64).sum().item())
        ok = els == wsum and total == 5 * wsum and cms.batch_diagnostics()["saturated"] == 0
        # and a prefix against the oracle: one pass of 1M keys into a fresh sketch
        prefix_ok = None
        if ctx.rank == 0:
            sys.path.insert(0, str(ROOT / "oracle"))
            import numpy as np
            import oracle

            m = 1_000_000
            c2 = self.pa.CountMinSketch(width=2**20, depth=5, device=ctx.dev)
            c2.add_many(ctx.gen_keys(m, 0), ctx.gen_weights(m, 0))
            oc = oracle.OracleCMS(2**20, 5)
            oc.add_keys(oracle.gen_keys16(0, m), oracle.gen_weights(0, m))
            prefix_ok = bool(np.array_equal(c2.table_tensor.cpu().numpy()[: oc.bins.size], oc.bins)) and c2.elements_added == oc.els_added
        add_ms = self.timer.mean_ms("add")
        chk_ms = timed_loop(torch, lambda: cms.check_many(self.keys), 5)
        line = {
            "metric": "million updates/sec (CMS 2^20 x 5, 100M weighted adds)",
            "config": {"workload": "cfg3: CountMinSketch(width=2^20, depth=5): per rank per step clear + 100M weighted updates "
                                   "(10 passes over 10M 16-byte keys, weights 1..7) + (SUM merge)",
                       "keys_per_rank": n, "passes": self.PASSES, "width": 2**20, "depth": 5,
                       "parallelism": f"key-range x{ctx.world}, replica per GPU, allreduce(SUM)" if ctx.world > 1 else "single GPU"},
            "roofline": roofline("cms_add", "CMS weighted add = k_part_bins<...,IdxCms<pow2>,PayWeightSmall,...,5> (weights 0..15 as 20-bit fields; k_part_scatter<PayWeight> otherwise) (sums the weights) + k_tally_fold + k_counter_apply "
                                 "(one pass of 10M updates between two HIP events)", n, add_ms,
                                 "pass 1 (5 FNV-1a chains + LDS counting sort), pass 2 folds 2^15-cell slices", "cms_add_weighted"),
            "rooflines": {"cms_check": roofline("cms_check", "CMS lookup (min over 5 rows)", n, chk_ms, "see DESIGN.md 3.2")},
            "detail": {"add_Mupd_s": n / add_ms / 1e3, "check_Mkeys_s": n / chk_ms / 1e3, "elements_added": els,
                       "sum_bins_equals_depth_x_weights": ok, "prefix_1M_equals_oracle": prefix_ok},
        }
        fails = [] if ok and prefix_ok is not False else ["parity property violated (cfg3)"]
        return line, fails


# ----------------------------------------------------------------------------------------------- cfg4
class Cfg4:
    """CountingBloomFilter 2^28 x u32 (1 GiB): the 50-batch add / remove stream in 1M-key batches"""

    def __init__(self, ctx: Ctx, args):
        import pyprobables_amd as pa

        if ctx.world > 1:
            raise SystemExit("cfg4 is a single-GPU configuration (removes must follow their adds: SURVEY.md 8e)")
        self.ctx, self.args, self.pa = ctx, args, pa
        self.B, self.nb = args.batch, args.batches
        self.keys = ctx.gen_keys(self.B * self.nb, 0)   # 50M keys = 800 MB resident
        # default: the plain API (add_many / remove_many, no opt-in) -- the engine's update windows (psk_window.hpp) make the small
        # batches share passes over the table and keep the reference's semantics for any stream
        self.mode = "off" if args.no_combine else ("borrow" if args.borrow_keys else (True if args.legacy_combine else ("window_borrow" if args.borrow_window else "window")))
        if self.mode == "off":
            from pyprobables_amd import _native as N

            N.set_option("update_window", 0)
        self.cbf = pa.CountingBloomFilter(est_elements=28005615, false_positive_rate=0.01, device=ctx.dev,
                                          combine_updates=self.mode if self.mode in (True, "borrow") else False, borrow_keys=self.mode == "window_borrow")
        assert self.cbf.number_bits == 2**28
        self.adds, self.removes = self.B * self.nb, (self.nb - 1) * (self.B // 2)
        self.ops_per_step = self.adds + self.removes
        self.timer = EventTimer(ctx.torch)

    def step(self, record=0, every_phase=False):
        cbf, keys, B = self.cbf, self.keys, self.B
        run = lambda: self._stream(cbf, keys, B)  # noqa: E731
        cbf.clear()
        if record == 1:
            self.timer.time("stream", run)
        else:
            run()

    def _stream(self, cbf, keys, B):
        for b in range(self.nb):
            cbf.add_many(keys[b * B:(b + 1) * B])
            if b >= 1:
                cbf.remove_many(keys[(b - 1) * B:(b - 1) * B + B // 2])  # each key removed at most once: well-formed
        cbf.synchronize()  # (deferred updates, if any, are part of the step)

    def instrumented(self):
        pass

    @staticmethod
    def _opt(name):
        from pyprobables_amd import _native as N

        return N.get_option(name)

    def finish(self, ms_step):
        ctx, torch, cbf = self.ctx, self.ctx.torch, self.cbf
        expect = self.adds - self.removes
        els = cbf.elements_added
        total = int(cbf.table_tensor.view(torch.int32).to(torch.int64).sum().item())
        diag = cbf.batch_diagnostics()
        ok = els == expect and total == 7 * expect and diag == {"violations": 0, "saturated": 0}
        ms = self.timer.mean_ms("stream")
        line = {
            "metric": "million ops/sec (CBF m=2^28 k=7, mixed add/remove stream in 1M-key batches)",
            "config": {"workload": f"cfg4: CountingBloomFilter(28005615, 0.01): 2^28 x uint32 = 1 GiB; per step clear + {self.nb} batches: "
                                   f"add {self.B} keys, remove the first {self.B // 2} keys of the previous batch",
                       "batch_keys": self.B, "batches": self.nb, "ops_per_step": self.ops_per_step, "parallelism": "single GPU",
                       "api": {"window": "default (add_many / remove_many, no opt-in)", "off": "default API, update windows off",
                               "window_borrow": "default API + borrow_keys=True (the update window hashes the caller's key tensors where they lie: no key copies; exact for any stream)",
                               True: "opt-in combine_updates=True", "borrow": "opt-in combine_updates='borrow'"}[self.mode],
                       "combine_updates": self.mode if self.mode in (True, "borrow") else False,
                       "note": "default API: the engine lets the 1M-key batches wait, in arrival order, in an update window and applies them in one "
                               "pass over the table that PROVES every remove (psk_window.hpp; exact for any stream: a window it cannot prove is "
                               "undone and replayed batch by batch), all inside the timed step (the stream ends with a flush).  --legacy-combine / "
                               "--borrow-keys: the round-3 opt-ins (removes as plain decrements after the window's adds: well-formed streams only); "
                               "--no-combine: update windows off, every batch at once"},
            "roofline": roofline("cbf_add", "CBF stream = per window: key copies + k_part_scatter<PayNonePhased> + k_win_fold over the 1 GiB table",
                                 self.ops_per_step, ms, "the fold read-modify-writes the whole 1 GiB table; batches are combined before it",
                                 # (profiles/r06_pmc_traffic.json: "cfg4_stream" = the default API's update windows -- key copies + pass 1 + k_win_fold --, "..._borrow_keys" = the same without copies)
                                 {True: "cfg4_stream_combine", "borrow": "cfg4_stream_combine_borrow", "off": "cfg4_stream_nocombine", "window": "cfg4_stream", "window_borrow": "cfg4_stream_borrow_keys"}[self.mode]),
            "rooflines": {},
            "detail": {"elements_added": els, "expected_elements": expect, "sum_counters_equals_k_x_live": total == 7 * expect, "diagnostics": diag,
                       "update_window_folds": self._opt("update_window_folds"), "update_window_replays": self._opt("update_window_replays")},
        }
        return line, ([] if ok else ["parity property violated (cfg4)"])


# ----------------------------------------------------------------------------------------------- cfg5
class Cfg5:
    """Bloom m = 2^31 sharded by key range: clear + insert shard + allreduce(OR) + check shard"""

    def __init__(self, ctx: Ctx, args):
        import pyprobables_amd as pa
        from pyprobables_amd import parallel

        self.ctx, self.args, self.pa, self.parallel = ctx, args, pa, parallel
        self.lo, self.hi = parallel.shard_range(args.n_total, ctx.rank, ctx.world)
        self.n = self.hi - self.lo
        # the shard's keys, generated on the device in rounds of 64M keys (1 GiB: one call = one partition round, option partition_max_keys)
        # and kept resident -- every call sweeps the whole 256 MiB table once per round, so the calls are as large as a round
        self.chunks, R = [], 1 << 26
        for s in range(self.lo, self.hi, R):
            self.chunks.append(ctx.gen_keys(min(R, self.hi - s), s))
        self.blm = pa.BloomFilter(est_elements=224044920, false_positive_rate=0.01, device=ctx.dev)
        assert self.blm.number_bits == 2**31 and self.blm.number_hashes == 7
        self.timer = EventTimer(ctx.torch)
        self.ops_per_step = 2 * args.n_total
        self.found = None

    def step(self, record=0, every_phase=False):
        if record == 1:  # one event pair around the whole step; the launches are timed on other steps (record == 2)
            return self.timer.time("step", lambda: self.step(0))
        t = self.timer.time if record == 2 else (lambda _n, f: f())
        blm = self.blm
        blm.clear()
        t("insert", lambda: [blm.add_many(c) for c in self.chunks])
        if self.ctx.distributed:
            t("merge", lambda: self.parallel.merge_bloom(blm, sync_elements=False))
        self.found = t("check", lambda: [blm.check_many(c) for c in self.chunks])

    def instrumented(self):
        pass

    def rank_times(self):
        """(insert ms, merge ms, lookup ms, table bytes) of one step on this rank (multi_gpu_diagnostics)"""
        torch, blm = self.ctx.torch, self.blm
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        blm.clear()
        ev[0].record()
        for c in self.chunks:
            blm.add_many(c)
        ev[1].record()
        self.parallel.merge_bloom(blm, sync_elements=False)
        ev[2].record()
        for c in self.chunks:
            blm.check_many(c)
        ev[3].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3]), 2**31 // 8

    def predicted(self, body):
        """the strong-scaling curve of cfg 5: per-key insert / lookup times of this run's shard, applied to n_total / N keys per rank"""
        d, mg = body["detail"], body.get("multi_gpu")
        ins = max(p["insert_ms"] for p in mg["per_rank"]) if mg else d["insert_ms"]
        chk = max(p["check_ms"] for p in mg["per_rank"]) if mg else d["check_ms"]
        return predicted_scaling("strong", 2**31 // 8, ins / (self.n / 1e6), chk / (self.n / 1e6), self.args.n_total, 0.05, 0.0, _worst_merge(body))

    def finish(self, ms_step):
        ctx, torch = self.ctx, self.ctx.torch
        ok = all(bool(r.all().item()) for r in self.found)
        # merged == single stream on a prefix (fresh filters; the full-size compare with the oracle is in tests/)
        merged_ok = None
        if ctx.distributed:
            npre = 4_000_000
            lo, hi = self.parallel.shard_range(npre, ctx.rank, ctx.world)
            a = self.pa.BloomFilter(est_elements=224044920, false_positive_rate=0.01, device=ctx.dev)
            a.add_many(ctx.gen_keys(hi - lo, lo))
            self.parallel.merge_bloom(a)
            b = self.pa.BloomFilter(est_elements=224044920, false_positive_rate=0.01, device=ctx.dev)
            b.add_many(ctx.gen_keys(npre, 0))
            merged_ok = bool(torch.equal(a.table_tensor, b.table_tensor)) and a.elements_added == npre
            del a, b
        ins, chk, mrg = self.timer.mean_ms("insert"), self.timer.mean_ms("check"), self.timer.mean_ms("merge")
        line = {
            "metric": "million keys/sec insert+lookup (Bloom m=2^31 k=7, key stream sharded by rank, allreduce(OR))",
            "scaling": "strong",
            "config": {"workload": "cfg5: BloomFilter(224044920, 0.01): m=2^31 bits (256 MiB replica), k=7; per step clear + insert the rank's "
                                   "key range + allreduce(OR) + check the rank's key range",
                       "keys_total": self.args.n_total, "keys_per_rank": self.n, "m_bits": 2**31, "k": 7,
                       "parallelism": f"key-range x{ctx.world}, replica per GPU, allreduce(OR) = all_to_all + OR kernel + all_gather"},
            "roofline": step_roofline(
                "one timed STEP = clear + insert the shard (k_part_scatter + k_bloom_apply per 32M-key chunk) + " + ("allreduce(OR) + " if ctx.distributed else "") +
                "check the shard; avg_kernel_ms = HIP events around whole steps of the timed region (median pair)", self.n, self.timer.median_ms("step"),
                {"bloom_insert": roofline("bloom_insert", "Bloom insert, m=2^31 (2048 slices): k_part_scatter + k_bloom_apply per 32M-key chunk", self.n, ins,
                                          "short (tile, slice) runs at 2048 slices: pass 1 is write-out bound", "bloom31_insert"),
                 "bloom_check": roofline("bloom_check", "Bloom lookup, m=2^31", self.n, chk, "as the insert", "bloom31_check")},
                "short (tile, slice) runs at 2048 slices: both pass 1s are write-out bound"),
            "rooflines": {},
            "detail": {"insert_ms": ins, "merge_ms": None if mrg != mrg else mrg, "check_ms": chk,
                       "insert_Mkeys_s_per_gpu": self.n / ins / 1e3, "check_Mkeys_s_per_gpu": self.n / chk / 1e3,
                       "merge_GBs_per_gpu": None if mrg != mrg else 2**28 / mrg / 1e6,
                       "all_inserted_found": ok, "merged_prefix_equals_single_stream": merged_ok},
        }
        fails = []
        if not ok:
            fails.append("parity property violated: an inserted key was not found")
        if merged_ok is False:
            fails.append("parity property violated: merged prefix differs from the single-stream filter")
        return line, fails


# ----------------------------------------------------------------------------------------------- the scaling curve this line predicts
XGMI_LINK_GBS = 153.0      # per link and direction; an MI355X node is fully connected, 7 links per GPU
RCCL_CALL_US = 25.0        # fixed cost assumed per RCCL call of the merge (all_to_all_single, all_gather_into_tensor): launch + handshake
OR_KERNEL_GBS = 4000.0     # psk_or_reduce_slices streams its R input slices and writes one (measured 3.6-4.9 TB/s on the streaming kernels)


def predicted_merge_ms(table_bytes: int, R: int) -> float:
    """allreduce(OR) over R fully connected GPUs: slice exchange (every GPU sends slice j straight to owner j, all R - 1 links busy at once:
    table / R bytes per link), the OR kernel over R slices of table / R bytes, all_gather (table / R per link again)"""
    if R <= 1:
        return 0.0
    per_link = table_bytes / R
    return 2.0 * (per_link / (XGMI_LINK_GBS * 1e9) * 1e3 + RCCL_CALL_US * 1e-3) + (table_bytes + per_link) / (OR_KERNEL_GBS * 1e9) * 1e3


def predicted_scaling(kind: str, table_bytes: int, insert_ms_per_Mkey: float, check_ms_per_Mkey: float, keys_1gpu: int, clear_ms: float,
                      scatter_share: float, measured=None):
    """What the 1/2/4/8-GPU curve of this configuration should look like, from THIS run's per-key insert / lookup times and the link model
    above -- so that the driver's SCALE_rNN.json can be read against a number: `value_Mkeys_s` per N, `efficiency` against N x the N = 1 value.
    kind "weak": every rank owns keys_1gpu keys (cfg 2); "strong": keys_1gpu keys are split over the ranks (cfg 5).
    scatter_share: the part of a lookup that is its pass 1 (never reads the table: it runs under the merge when bench.py overlaps them).
    measured = {"world": R, "merge_ms": ...}: this run's own merge next to the model's."""
    out = {"model": f"step(N) = clear + insert + allreduce(OR) + lookup; allreduce(OR) = 2 x (table/N bytes per xGMI link at {XGMI_LINK_GBS:g} GB/s + "
                    f"{RCCL_CALL_US:g} us per RCCL call) + OR kernel over (N + 1) x table/N bytes at {OR_KERNEL_GBS:g} GB/s; the lookup's pass 1 "
                    f"({scatter_share:.2f} of a lookup) hides under the merge; insert / lookup per key as measured in this run",
           "table_bytes": table_bytes, "per_N": {}}
    base = None
    for R in (1, 2, 4, 8):
        keys_rank = keys_1gpu if kind == "weak" else -(-keys_1gpu // R)
        ins, chk = insert_ms_per_Mkey * keys_rank / 1e6, check_ms_per_Mkey * keys_rank / 1e6
        mrg = predicted_merge_ms(table_bytes, R)
        hidden = min(mrg, scatter_share * chk) if R > 1 else 0.0
        step = clear_ms + ins + mrg + chk - hidden
        total_keys = keys_rank * R if kind == "weak" else keys_1gpu
        val = 2 * total_keys / step / 1e3
        base = val if R == 1 else base
        out["per_N"][str(R)] = {"ms_per_step": step, "merge_ms": mrg, "value_Mkeys_s": val, "efficiency": val / (R * base)}
    if measured and measured.get("world", 1) > 1 and measured.get("merge_ms"):
        R = measured["world"]
        out["measured_vs_model"] = {"world": R, "merge_ms_measured": measured["merge_ms"], "merge_ms_model": predicted_merge_ms(table_bytes, R),
                                    "ratio": measured["merge_ms"] / predicted_merge_ms(table_bytes, R)}
    return out


WORKLOADS = {"cfg2": Cfg2, "cfg3": Cfg3, "cfg4": Cfg4, "cfg5": Cfg5}
DTYPES = {"cfg2": "u64", "cfg3": "i32", "cfg4": "u32", "cfg5": "u64"}  # arithmetic type of the path (hash chains / counters)


UNITS = {"cfg2": "Mkeys/s", "cfg3": "Mupdates/s", "cfg4": "Mops/s", "cfg5": "Mkeys/s"}
EXTRA_STEPS = {"cfg3": (5, 1), "cfg4": (3, 1), "cfg5": (2, 1)}  # (steps, warmup) of the extra configurations on the default line


def multi_gpu_diagnostics(ctx: Ctx, args, wl):
    """N > 1: what every rank measured for itself -- insert / merge / lookup of one un-overlapped step, HIP events on its own stream --
    gathered on every rank, plus the merge's bus rate and the group size RCCL reports.  The first real 8-GPU run cannot be rehearsed on
    the one-GPU boxes of this pool, so the line says where the time of each rank went."""
    torch, dist = ctx.torch, ctx.dist
    ins, mrg, chk, table_bytes = wl.rank_times()
    mine = torch.tensor([ins, mrg, chk], dtype=torch.float64, device=f"cuda:{ctx.dev}")
    everyone = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(everyone, mine)
    seen = dist.get_world_size()
    if seen != max(args.gpus, 1):
        raise RuntimeError(f"the process group has {seen} ranks, --gpus says {args.gpus}")
    per_rank = [{"rank": r, "insert_ms": float(t[0]), "merge_ms": float(t[1]), "check_ms": float(t[2])} for r, t in enumerate(everyone)]
    worst = max(p["merge_ms"] for p in per_rank)
    R = seen
    # allreduce(OR) = slice exchange + all_gather: every GPU sends and receives (R - 1) / R of the table in each of the two steps
    moved = 2.0 * (R - 1) / R * table_bytes
    return {"ranks_seen_by_rccl": seen, "backend": dist.get_backend(), "per_rank": per_rank,
            "merge_GBs_per_gpu": (moved / (worst * 1e-3) / 1e9) if worst > 0 and R > 1 else None,
            "merge_bytes_per_gpu_each_way": moved, "table_bytes": table_bytes,
            "note": "un-overlapped step after the timed region (clear, insert, merge, lookup one after the other); merge_GBs_per_gpu = bytes every "
                    "GPU sends (= receives) in the slice exchange + all_gather, divided by the slowest rank's merge time"}


def run_workload(ctx: Ctx, args, name: str, steps: int, warmup: int, spinup: float, repeats: int = 1):
    """spin-up, `warmup` untimed steps, `repeats` blocks of EXACTLY `steps` timed steps between two fences each, per-phase pass, parity.
    -> (workload, body, fails, seconds per step of the median block (max over ranks), untimed spin-up steps)"""
    torch = ctx.torch
    phase(f"{name}: set-up (keys, sketch)")
    wl = WORKLOADS[name](ctx, args)
    phase(f"{name}: spin-up and warm-up steps")
    # clock ramp: a fresh process starts at idle clocks and the first tens of milliseconds after a fence run slow; spin
    # the same step untimed first so that short runs (the driver's --steps 20) measure the steady state
    t_spin = time.perf_counter()
    spun = 0
    while time.perf_counter() - t_spin < spinup:
        wl.step(False)
        spun += 1
        if spun % 8 == 0:
            torch.cuda.synchronize()
    for _ in range(warmup):
        wl.step(False)
    ctx.fence()
    phase(f"{name}: timed steps")
    # The block of EXACTLY `steps` steps, each block between its own fences, `repeats` times: the median block is the result, min / max say
    # how far one block may lie from it.  Events: every sample_every-th step carries ONE pair around the whole step (the step's roofline),
    # the steps half-way between them carry the events between the launches (their nested objects) -- the whole-step samples hold no bubbles.
    sample_every = max(2, min(20, steps // 3))  # (>= 3 samples of either kind per block when steps >= 6)
    blocks = []
    for _rep in range(max(1, repeats)):
        ctx.fence()
        t0 = time.perf_counter()
        for it in range(steps):
            # (never the block's first step: it starts on an idle GPU right behind the fence, and an event pair around it times the host's
            # enqueueing of the step -- tens of microseconds of Python -- on top of the kernels)
            # (blocks of fewer than three steps have no other step to offer)
            wl.step(0 if (it == 0 and steps >= 3) else (1 if it % sample_every == 0 else (2 if it % sample_every == sample_every // 2 else 0)))
        ctx.fence()
        blocks.append(ctx.max_over_ranks(time.perf_counter() - t0))
    srt = sorted(blocks)
    elapsed = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    phase(f"{name}: per-phase instrumented pass")
    wl.instrumented()  # per-phase HIP-event times (not part of `value`)
    ctx.fence()
    phase(f"{name}: parity checks and result")
    body, fails = wl.finish(elapsed / steps * 1e3)
    if ctx.distributed and hasattr(wl, "rank_times"):
        phase(f"{name}: per-rank diagnostics")
        body["multi_gpu"] = multi_gpu_diagnostics(ctx, args, wl)
    if name in ("cfg2", "cfg5") and ctx.rank == 0:
        body["predicted"] = wl.predicted(body)
    body["config"]["timed_blocks"] = {
        "repeats": len(blocks), "steps_per_block": steps, "ms_per_step_median": elapsed / steps * 1e3, "ms_per_step_min": srt[0] / steps * 1e3,
        "ms_per_step_max": srt[-1] / steps * 1e3, "ms_per_step_blocks": [b / steps * 1e3 for b in blocks],
        "note": "every block is `steps` steps between a barrier + synchronize on both sides (max over ranks); ms_per_step / value are the median block's"}
    return wl, body, fails, elapsed / steps, spun


def extra_config(ctx: Ctx, args, name: str):
    """one of cfg3 / cfg4 / cfg5 (N = 1) as written, a few steps: the object reported under `configs.<name>`"""
    steps, warmup = EXTRA_STEPS[name]
    wl, body, fails, sec, spun = run_workload(ctx, args, name, steps, warmup, min(args.spinup, 0.3), repeats=min(args.repeats, 3))
    obj = {"metric": body.pop("metric"), "value": wl.ops_per_step / sec / 1e6, "unit": UNITS[name], "ms_per_step": sec * 1e3, "steps": steps,
           "warmup": warmup, "scaling": body.pop("scaling", "weak"), "dtype": DTYPES[name], "parity_ok": not fails, "parity_failures": fails}
    obj.update(body)
    del wl
    ctx.torch.cuda.empty_cache()
    return obj, fails


# ----------------------------------------------------------------------------------------------- the result line
# The driver keeps the last ~8 KB of stdout: the LAST line is a compact object (numbers and short names only, < 6 KB); everything
# else the run measured (limiter notes, per-launch objects, per-rank diagnostics, the scaling model, the host legs) is the FULL object,
# written to gpurun_out/bench_detail.json (PSK_BENCH_DETAIL overrides the path; --print-detail also sends it to stderr).
LINE_LIMIT = 6000


def _r(x, sig=6):
    """floats to `sig` significant digits (the line is read by a parser with a length cap, not by a bit-exact consumer)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _short(text, n=80):
    text = str(text)
    return text if len(text) <= n else text[: n - 3] + "..."


def compact_roofline(r, kernel=None):
    if not r:
        return None
    l2 = r.get("l2_hit")
    if isinstance(l2, dict):  # a whole step: the launches' rates, in launch order
        l2 = [v for v in l2.values()]
    return _r({"bound": r["bound"], "kernel": _short(kernel or r["kernel"]), "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
               "frac": r["frac"], "traffic": r.get("traffic"), "l2_hit": l2, "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"],
               "avg_kernel_ms": r["avg_kernel_ms"]})


def compact_cpu(c):
    if not c:
        return None
    out = {k: c.get(k) for k in ("value", "unit", "cores", "kind", "seconds")}
    out["sample"] = _short(c.get("sample", ""), 110)
    for name, leg in (c.get("legs") or {}).items():
        if name != "port_1core":
            out[name] = {"value": leg["value"], "cores": leg["cores"]}
    return _r(out)


SHORT_KERNEL = {
    "cfg2": "step = clear + 2 x k_part_bins + k_bloom_apply + k_bloom_test_flag",
    "cfg3": "CMS add: k_part_bins<IdxCms,PayWeightSmall,5> + k_counter_apply",
    "cfg4": "window: key copies + k_part_scatter<PayNonePhased> + k_win_fold (1 GiB)",
    "cfg5": "step = clear + chunks of (k_part_scatter + k_bloom_apply | k_bloom_test_flag)",
}
SHORT_WORKLOAD = {
    "cfg2": "cfg2 Bloom m=2^28 k=7: per rank per step clear + insert 10M 16B keys + (merge) + check them",
    "cfg3": "cfg3 CMS 2^20x5: per step clear + 100M weighted updates (10 passes x 10M keys) + (merge)",
    "cfg4": "cfg4 CBF m=2^28 (1 GiB) k=7: per step clear + 50 batches (add 1M, remove 0.5M of the previous)",
    "cfg5": "cfg5 Bloom m=2^31 k=7: per step clear + insert the rank's key range + allreduce(OR) + check it",
}
DETAIL_KEEP = {
    "cfg2": ("insert_Mkeys_s", "check_Mkeys_s", "check_all_fresh_Mkeys_s", "check_half_fresh_Mkeys_s", "check_bits_Mkeys_s", "clear_ms", "merge_ms",
             "merge_plus_check_ms", "all_inserted_found", "merged_table_equals_single_stream", "bits_set", "step_ms_events", "step_event_samples",
             "cms_add_Mupd_s", "cms_check_Mkeys_s", "cbf_add_Mops_s", "cbf_check_Mkeys_s", "cbf_check_unchanged_table_Mkeys_s", "cbf_remove_Mops_s",
             "gups_atomic_or_32MiB_Gprobes_s", "gups_gather_32MiB_Gprobes_s", "gups_atomic_add_1GiB_Gprobes_s", "gups_gather_1GiB_Gprobes_s"),
    "cfg3": ("add_Mupd_s", "check_Mkeys_s", "elements_added", "sum_bins_equals_depth_x_weights", "prefix_1M_equals_oracle", "merge_ms"),
    "cfg4": ("elements_added", "expected_elements", "sum_counters_equals_k_x_live"),
    "cfg5": ("insert_ms", "merge_ms", "check_ms", "insert_Mkeys_s_per_gpu", "check_Mkeys_s_per_gpu", "merge_GBs_per_gpu", "all_inserted_found",
             "merged_prefix_equals_single_stream"),
}


def compact_config(name, cfg):
    keep = ("keys_per_rank", "keys_total", "key_bytes", "m_bits", "k", "width", "depth", "updates_per_step", "ops_per_step", "batch", "batches", "parallelism", "api")
    out = {"workload": SHORT_WORKLOAD[name]}
    out.update({k: (_short(v, 70) if isinstance(v, str) else v) for k, v in cfg.items() if k in keep})
    return out


def compact_line(full: dict, name: str, detail_path) -> dict:
    """the driver's line: the contract's fields, the dominant launch's roofline, one short object per other operation / configuration"""
    head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "repeats", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "rc")
    out = {k: full.get(k) for k in head}
    out["config"] = compact_config(name, full["config"])
    out["roofline"] = compact_roofline(full.get("roofline"), SHORT_KERNEL[name])
    lau = (full.get("roofline") or {}).get("launches") or {}
    ops = {**lau, **(full.get("rooflines") or {})}
    if ops:  # per operation: fraction of the 8 TB/s roofline, launch time, counter traffic over algorithmic bytes
        out["rooflines"] = {k: {"frac": v["frac"], "ms": v["avg_kernel_ms"], "traffic_x": (v["traffic"] / v["algorithmic_bytes_per_launch"]) if v.get("traffic") else None,
                                "l2_hit": v.get("l2_hit")} for k, v in ops.items()}
    det = full.get("detail") or {}
    out["detail"] = {k: det[k] for k in DETAIL_KEEP[name] if det.get(k) is not None}
    rag = det.get("bloom_ragged_keys_Mkeys_s")
    if rag:
        out["detail"]["ragged_insert_Mkeys_s"], out["detail"]["ragged_check_Mkeys_s"] = rag["insert"], rag["check"]
    pk = det.get("per_key_call_us")
    if pk:
        out["detail"]["per_key_call_us"] = {k: _r(v, 3) for k, v in pk.items() if k != "key"}
    hb = det.get("bloom_host_buffers_Mkeys_s")
    if hb:
        out["detail"]["pcie_inclusive_insert_Mkeys_s"], out["detail"]["pcie_inclusive_check_Mkeys_s"] = hb["insert"], hb["check"]
    if full.get("multi_gpu"):
        mg = full["multi_gpu"]
        out["multi_gpu"] = {"ranks_seen_by_rccl": mg["ranks_seen_by_rccl"], "backend": mg["backend"], "merge_GBs_per_gpu": mg.get("merge_GBs_per_gpu"),
                            "insert_ms_max": max(p["insert_ms"] for p in mg["per_rank"]), "merge_ms_max": max(p["merge_ms"] for p in mg["per_rank"]),
                            "check_ms_max": max(p["check_ms"] for p in mg["per_rank"])}
    if full.get("predicted"):
        out["predicted_value_by_n"] = {n: v["value_Mkeys_s"] for n, v in full["predicted"]["per_N"].items()}
    if full.get("configs"):
        out["configs"] = {}
        for cname, c in full["configs"].items():
            o = {"value": c["value"], "unit": c["unit"], "ms_per_step": c["ms_per_step"], "steps": c["steps"], "parity_ok": c["parity_ok"],
                 "frac": (c.get("roofline") or {}).get("frac"), "traffic": (c.get("roofline") or {}).get("traffic"),
                 "avg_kernel_ms": (c.get("roofline") or {}).get("avg_kernel_ms")}
            if c.get("cpu_baseline"):
                o["cpu_baseline"] = {k: c["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind", "seconds")}
            if c.get("with_borrow_keys"):
                o["with_borrow_keys"] = {k: c["with_borrow_keys"][k] for k in ("value", "ms_per_step", "parity_ok", "roofline_frac")}
            for k in ("check_Mkeys_s", "insert_Mkeys_s_per_gpu", "check_Mkeys_s_per_gpu"):
                if (c.get("detail") or {}).get(k) is not None:
                    o[k] = c["detail"][k]
            out["configs"][cname] = o
    if full.get("cms"):
        out["cms"] = {k: full["cms"][k] for k in ("insert_Mupdates_s", "lookup_Mkeys_s")}
    out["cpu_baseline"] = compact_cpu(full.get("cpu_baseline"))
    if full.get("error"):
        out["error"] = _short(full["error"], 600)
    out["detail_file"] = str(detail_path) if detail_path else None
    out = _r(out)
    text = json.dumps(out, separators=(",", ":"))
    for drop in ("predicted_value_by_n", "rooflines", "detail"):  # never over the cap, whatever a future field adds
        if len(text) <= LINE_LIMIT:
            break
        out.pop(drop, None)
        text = json.dumps(out, separators=(",", ":"))
    return out


def write_detail(full: dict):
    """the full object -> gpurun_out/bench_detail.json (scratch on the GPU box; copies judged live under profiles/); returns the path or None"""
    path = Path(os.environ.get("PSK_BENCH_DETAIL", ROOT / "gpurun_out" / "bench_detail.json"))
    try:
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(json.dumps(full) + "\n")
        try:
            return path.relative_to(ROOT)
        except ValueError:
            return path
    except OSError:
        return None


def main():
    args = parse()
    self_launch(args)
    rank = int(os.environ.get("RANK", 0))

    def give_up():  # a hang (a rank that died inside a collective, a wedged rendezvous): say so in the result line, then leave
        leave_error_note(rank, f"watchdog: no result after {args.timeout:g} s")
        if rank == 0:
            notes = collect_error_notes(skip_rank=0)
            os.write(1, ("\n" + error_line(args, 124, f"watchdog: rank 0 had no result after {args.timeout:g} s, phase '{PHASE['name']}' (a rank failed or a collective hangs)"
                                           + (f" | {notes}" if notes else "")) + "\n").encode())
        os._exit(124)

    dog = threading.Timer(args.timeout, give_up)
    dog.daemon = True
    dog.start()
    try:
        rc = run(args)
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001  -- the line must exist whatever happened (rank 0 reports; the others leave a note)
        text = f"{type(e).__name__}: {e} | " + traceback.format_exc(limit=4)
        if rank == 0:
            time.sleep(0.5)  # (a rank that failed first has written its note by now)
            notes = collect_error_notes(skip_rank=0)
            sys.stdout.flush()
            print(error_line(args, 1, f"rank 0 failed in phase '{PHASE['name']}': {text}" + (f" | {notes}" if notes else "")), flush=True)
        else:
            leave_error_note(rank, text)
        os._exit(1)  # (not sys.exit: a failed rank must not wait in atexit handlers for collectives that will never complete)
    dog.cancel()
    if rc:
        raise SystemExit(rc)


def run(args):
    inject = os.environ.get("PSK_BENCH_INJECT")  # test hook for the failure paths: "raise" / "hang" (optionally ":<rank>")
    if inject:
        kind, _, who = inject.partition(":")
        if not who or int(who) == int(os.environ.get("RANK", 0)):
            if kind == "hang":
                time.sleep(1e6)
            raise RuntimeError("injected failure (PSK_BENCH_INJECT)")
    ctx = Ctx(args)
    torch = ctx.torch
    for opt in args.option:
        from pyprobables_amd import _native as _n
        name, _, value = opt.partition("=")
        _n.set_option(name, int(value, 0))
    wl, body, fails, sec, spun = run_workload(ctx, args, args.config, args.steps, args.warmup, args.spinup, repeats=args.repeats)

    line = {
        "metric": body.pop("metric"),
        "value": wl.ops_per_step / sec / 1e6,
        "unit": UNITS[args.config],
        "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec * 1e3,
        "higher_is_better": True, "scaling": body.pop("scaling", "weak"), "vs_baseline": None, "dtype": DTYPES[args.config],
        "data": "synthetic", "rc": 0,
    }
    tb = body["config"]["timed_blocks"]
    line.update({"repeats": tb["repeats"], "ms_per_step_min": tb["ms_per_step_min"], "ms_per_step_max": tb["ms_per_step_max"]})
    body["config"]["clock_spinup"] = f"{spun} untimed steps (~{args.spinup:g} s) before the {args.warmup} warm-up steps"
    line.update(body)
    if args.config == "cfg2" and ctx.world == 1 and not ctx.distributed and not args.no_extra_configs and not args.no_detail:
        # the rest of BASELINE.json's metric on the same line: cfg3 as written (CMS 2^20 x 5, 100M weighted updates), and short
        # cfg4 / cfg5 (N = 1) steps -- after, and outside, the headline's timed region
        del wl
        torch.cuda.empty_cache()
        line["configs"] = {}
        for name in ("cfg3", "cfg4", "cfg5"):
            obj, f2 = extra_config(ctx, args, name)
            line["configs"][name] = obj
            fails += [f"{name}: {x}" for x in f2]
        # cfg 4 once more with CountingBloomFilter(borrow_keys=True): same API calls, same exactness; the update window hashes the caller's
        # (resident, never overwritten) key tensors where they lie instead of copying every batch into its key list
        import copy

        a2 = copy.copy(args)
        a2.borrow_window = True
        ob, f2 = extra_config(ctx, a2, "cfg4")
        line["configs"]["cfg4"]["with_borrow_keys"] = {"value": ob["value"], "unit": ob["unit"], "ms_per_step": ob["ms_per_step"], "parity_ok": ob["parity_ok"],
                                                       "api": ob["config"]["api"], "roofline_frac": ob["roofline"]["frac"]}
        fails += [f"cfg4 (borrow_keys): {x}" for x in f2]
        c3 = line["configs"]["cfg3"]
        line["cms"] = {"insert_Mupdates_s": c3["value"], "lookup_Mkeys_s": c3["detail"]["check_Mkeys_s"], "source": "configs.cfg3 (100M weighted updates in 10 passes; lookups of the 10M keys)"}
    if ctx.rank == 0 and ctx.world == 1 and not args.no_cpu_baseline:
        if args.config == "cfg2":
            line["cpu_baseline"] = cpu_baseline(min(args.n, 10_000_000))
            if "configs" in line:  # the other halves of the metric: their own host legs (bounded to ~3 s each)
                line["configs"]["cfg3"]["cpu_baseline"] = cpu_baseline_cfg3()
                line["configs"]["cfg4"]["cpu_baseline"] = cpu_baseline_cfg4()
        elif args.config == "cfg3":
            line["cpu_baseline"] = cpu_baseline_cfg3()
        elif args.config == "cfg4":
            line["cpu_baseline"] = cpu_baseline_cfg4()
        else:
            line["cpu_baseline"] = cpu_baseline(10_000_000)
    elif ctx.rank == 0:
        line["cpu_baseline"] = None
    if fails:
        line["rc"] = 2
        line["error"] = "; ".join(fails)
    if ctx.dist is not None:
        ctx.dist.destroy_process_group()
    if ctx.rank == 0:
        # the JSON line must be the LAST thing on stdout: RCCL's NCCL_DEBUG=VERSION banner sits in the C stdio buffer
        # until exit, so drain C stdio first (the other ranks' stdout was sent to /dev/null at start)
        import ctypes  # noqa: PLC0415

        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        where = write_detail(line)
        if args.print_detail:
            print(json.dumps(line), file=sys.stderr, flush=True)
        print(json.dumps(compact_line(line, args.config, where), separators=(",", ":")), flush=True)
    if fails:
        print("; ".join(fails), file=sys.stderr)
        return 2
    return 0


if __name__ == "__main__":
    main()
