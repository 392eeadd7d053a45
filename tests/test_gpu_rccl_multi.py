"""The N > 1 path over REAL RCCL: one process per visible GPU, backend "nccl", world = device_count().
Every rank inserts its key range into its own device replica, merges (parallel.merge_bloom / merge_bloom_async /
merge_counters, incl. the widen-and-clamp branch) and compares its merged table with ONE oracle sketch fed the whole stream
(bloom.py:401-428: union is a bytewise OR; countminsketch.py:380-391: join clamps at the rails).  On a one-GPU box the same
workers run as a one-rank RCCL group with PSK_FORCE_MERGE_PATH=1, so the collective code (all_to_all -> HIP OR kernel ->
all_gather; widen -> all_reduce -> clamp) still executes and the test itself is exercised before it meets an 8-GPU node."""

import os
import socket
import sys
from datetime import timedelta
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rccl_worker(rank, world, port, n_total):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world == 1:
        os.environ["PSK_FORCE_MERGE_PATH"] = "1"
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"), timeout=timedelta(seconds=180))
    import oracle

    import pyprobables_amd as pa
    from pyprobables_amd import parallel

    dev = f"cuda:{rank}"
    lo, hi = parallel.shard_range(n_total, rank, world)
    keys = oracle.gen_keys16(0, n_total)
    mine = torch.from_numpy(keys[lo:hi]).to(dev)
    # ---- Bloom: m = 958506 (not a multiple of 16 * world bytes: the zero-padded staging copy) and m = 2^22; sync and async merge
    for est, use_async in ((100000, False), (100000, True), (437_000, False)):
        blm = pa.BloomFilter(est_elements=est, false_positive_rate=0.01, device=rank)
        blm.add_many(mine)
        if use_async:
            h = parallel.merge_bloom_async(blm)
            h.wait()
            blm.elements_added = parallel._sum_int(blm.elements_added, blm.table_tensor.device)
        else:
            parallel.merge_bloom(blm)
        ob = oracle.OracleBloom(blm.number_bits, blm.number_hashes)
        ob.add_keys(keys)
        assert np.array_equal(np.frombuffer(bytes(blm.bloom), dtype=np.uint8), ob.bloom), f"rank {rank}: merged Bloom table (est {est}, async {use_async})"
        assert blm.elements_added == n_total
        assert bool(blm.check_many(torch.from_numpy(keys).to(dev)).all().item())  # keys of the OTHER ranks are found through the merge
    # ---- CMS: wrap-free SUM
    w = oracle.gen_weights(0, n_total)
    cms = pa.CountMinSketch(width=4099, depth=5, device=rank)
    cms.add_many(mine, torch.from_numpy(w[lo:hi]).to(dev))
    parallel.merge_counters(cms)
    oc = oracle.OracleCMS(4099, 5)
    oc.add_keys(keys, w)
    assert np.array_equal(cms.table_tensor.cpu().numpy()[: oc.bins.size], oc.bins), f"rank {rank}: merged CMS bins"
    assert cms.elements_added == oc.els_added
    # ---- CMS: every rank holds 2^30 in the bins of key 0 and a bound of 3 * 2^30 -> the widened reduction; clamps at INT32_MAX from 2 ranks on
    big = pa.CountMinSketch(width=4096, depth=3, device=rank)
    k0 = torch.from_numpy(keys[:1]).to(dev)
    w30 = torch.tensor([1 << 30], dtype=torch.int32, device=dev)
    big.add_many(k0, w30)
    big.remove_many(k0, w30)
    big.add_many(k0, w30)
    parallel.merge_counters(big)
    want = min(world << 30, 2**31 - 1)
    assert int(big.check_many(k0).item()) == want, f"rank {rank}: clamped merge"
    assert big.elements_added == (world << 30)
    # ---- CBF: SUM of uint32 counters
    cbf = pa.CountingBloomFilter(est_elements=50000, false_positive_rate=0.01, device=rank)
    cbf.add_many(mine)
    parallel.merge_counters(cbf)
    ocb = oracle.OracleCBF(cbf.number_bits, cbf.number_hashes)
    ocb.update_keys(keys)
    assert np.array_equal(cbf.table_tensor.cpu().numpy().view(np.uint32)[: ocb.bloom.size], ocb.bloom), f"rank {rank}: merged CBF counters"
    assert cbf.elements_added == n_total
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_insert_and_merge_over_rccl_equals_single_stream():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    world = torch.cuda.device_count()
    mp.spawn(_rccl_worker, args=(world, _free_port(), 400_003), nprocs=world, join=True)
