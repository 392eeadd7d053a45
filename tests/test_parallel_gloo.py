"""Multi-process (world_size 2 and 3, gloo, CPU) tests of the sharded-insert + merge composition in
pyprobables_amd/parallel.py: key-range partition, allreduce(OR) = all_to_all + OR-reduce + all_gather,
SUM all-reduce for counters.  The per-rank replicas are produced by the plain-C oracle (there is no GPU
here); the OR-reduce step is injected as a torch op because the product's reduce step is a HIP kernel."""

import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torch_or_reduce(dst, src, nslices, slice_words):
    acc = src[:slice_words].clone()
    for j in range(1, nslices):
        acc |= src[j * slice_words:(j + 1) * slice_words]
    dst.copy_(acc)


class _FakeSketch:
    """what merge_* touch on a sketch, backed by CPU tensors"""

    def __init__(self, table, els):
        self.table_tensor = table
        self._els_added = els
        self._tab = None

    @property
    def elements_added(self):
        return self._els_added

    @elements_added.setter
    def elements_added(self, v):
        self._els_added = v


def _worker(rank, world, port, n_total, m_bits, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle

    from pyprobables_amd import parallel

    lo, hi = parallel.shard_range(n_total, rank, world)
    keys = oracle.gen_keys16(lo, hi - lo)

    # ---- Bloom: per-rank replica -> allreduce(OR)
    ob = oracle.OracleBloom(m_bits, 7)
    ob.add_keys(keys)
    padded = (ob.bloom.size + 15) & ~15
    raw = np.zeros(padded, dtype=np.uint8)
    raw[: ob.bloom.size] = ob.bloom
    blm = _FakeSketch(torch.from_numpy(raw.view(np.int32).copy()), hi - lo)
    if rank % 2 == 0:
        parallel.merge_bloom(blm, or_reduce=_torch_or_reduce)
    else:  # the async form must be the same collective (on CPU tensors it simply runs inline)
        h = parallel.merge_bloom_async(blm, or_reduce=_torch_or_reduce)
        h.wait()
        blm.elements_added = parallel._sum_int(blm.elements_added, blm.table_tensor.device)

    # ---- CMS: per-rank replica -> allreduce(SUM)
    w = oracle.gen_weights(lo, hi - lo)
    oc = oracle.OracleCMS(1009, 4)
    oc.add_keys(keys, w)
    cms = _FakeSketch(torch.from_numpy(oc.bins.copy()), oc.els_added)
    parallel.merge_counters(cms)

    np.save(Path(out_dir) / f"bloom_{rank}.npy", blm.table_tensor.numpy().view(np.uint8)[: ob.bloom.size])
    np.save(Path(out_dir) / f"cms_{rank}.npy", cms.table_tensor.numpy())
    (Path(out_dir) / f"els_{rank}.txt").write_text(f"{blm.elements_added} {cms.elements_added}")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,m_bits", [(2, 958506), (3, 958506), (2, 2**20)])
def test_sharded_insert_and_merge_equals_single_stream(tmp_path, oracle, world, m_bits):
    n_total = 20001  # not divisible by the world size
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, m_bits, str(tmp_path)), nprocs=world, join=True)
    keys = oracle.gen_keys16(0, n_total)
    ob = oracle.OracleBloom(m_bits, 7)
    ob.add_keys(keys)
    oc = oracle.OracleCMS(1009, 4)
    oc.add_keys(keys, oracle.gen_weights(0, n_total))
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"bloom_{r}.npy"), ob.bloom), f"rank {r} bloom"
        assert np.array_equal(np.load(tmp_path / f"cms_{r}.npy"), oc.bins), f"rank {r} cms"
        els_b, els_c = map(int, (tmp_path / f"els_{r}.txt").read_text().split())
        assert els_b == n_total and els_c == oc.els_added


def test_shard_range_covers_stream():
    from pyprobables_amd.parallel import shard_range

    for n, w in [(10, 1), (10, 3), (7, 8), (1_000_000_000, 8), (0, 4)]:
        pieces = [shard_range(n, r, w) for r in range(w)]
        assert pieces[0][0] == 0 and pieces[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
        sizes = [hi - lo for lo, hi in pieces]
        assert max(sizes) - min(sizes) <= 1


def test_hip_or_reduce_refuses_cpu_tensors():
    from pyprobables_amd.parallel import hip_or_reduce

    with pytest.raises(RuntimeError):
        hip_or_reduce(torch.zeros(4, dtype=torch.int32), torch.zeros(8, dtype=torch.int32), 2, 4)


def _wrap_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyprobables_amd import parallel

    # CMS (int32, both rails): bin 0 would wrap past INT32_MAX, bin 1 past INT32_MIN, bin 2 stays exact, bin 3 mixed signs
    per_rank = [[2**31 - 10, -(2**31) + 5, 1000, 2**31 - 1], [100, -50, 2345, -(2**31)]]
    cms = _FakeSketch(torch.tensor(per_rank[rank], dtype=torch.int32), 2**62 + 7)
    parallel.merge_counters(cms, unsigned=False)
    # CBF (uint32 bit patterns in an int32 tensor): cell 0 would wrap past 2^32-1, cell 1 is exact above 2^31
    u = np.array([[3_000_000_000, 2_000_000_000, 7], [2_000_000_000, 2_000_000_000, 8]][rank], dtype=np.uint32)
    cbf = _FakeSketch(torch.from_numpy(u.view(np.int32).copy()), 2**63 + 11)
    parallel.merge_counters(cbf, unsigned=True)
    # small tables keep the plain 32-bit path (bound sum below the rail)
    small = _FakeSketch(torch.tensor([5, -7, 2**29 - 1], dtype=torch.int32), 3)
    parallel.merge_counters(small, unsigned=False)
    np.save(Path(out_dir) / f"wrap_cms_{rank}.npy", cms.table_tensor.numpy())
    np.save(Path(out_dir) / f"wrap_cbf_{rank}.npy", cbf.table_tensor.numpy().view(np.uint32))
    np.save(Path(out_dir) / f"wrap_small_{rank}.npy", small.table_tensor.numpy())
    (Path(out_dir) / f"wrap_els_{rank}.txt").write_text(f"{cms.elements_added} {cbf.elements_added} {small.elements_added}")
    dist.destroy_process_group()


def test_counter_merge_that_would_wrap_is_clamped_like_join(tmp_path):
    """a 32-bit all_reduce(SUM) wraps silently; the merge agrees on the summed bounds first and, above the rail, sums in
    64 bits and clamps exactly like countminsketch.py:380-391 (join) / countingbloom.py:149-151"""
    port = _free_port()
    mp.spawn(_wrap_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(tmp_path / f"wrap_cms_{r}.npy").tolist() == [2**31 - 1, -(2**31), 3345, -1]
        assert np.load(tmp_path / f"wrap_cbf_{r}.npy").tolist() == [2**32 - 1, 4_000_000_000, 15]
        assert np.load(tmp_path / f"wrap_small_{r}.npy").tolist() == [10, -14, 2**30 - 2]
        e_cms, e_cbf, e_small = map(int, (tmp_path / f"wrap_els_{r}.txt").read_text().split())
        assert e_cms == 2**63 - 1          # 2 x (2^62 + 7) clamps at the int64 rail (countminsketch.py:285-287)
        assert e_cbf == 2**64 - 1          # 2 x (2^63 + 11) clamps at the uint64 rail (countingbloom.py:154)
        assert e_small == 6
