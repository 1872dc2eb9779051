"""N > 1 path on CPU: world_size-2 gloo process group, block partition of the samples and the
single variable-length gather of guide tables (crisprhawk_hip.parallel).  No GPU, no HIP."""
import os
import socket

import numpy as np
import pytest

from crisprhawk_hip.parallel import COLUMNS, gather_tables, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 2504, 5008):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def _fake_table(rank, n):
    rng = np.random.default_rng(100 + rank)
    cols = {k: rng.integers(0, 200, size=n).astype(dt) for k, dt in COLUMNS}
    cols["hap"][: n // 3] = 0  # REF rows
    cols["cfdon"] = rng.random(n)
    cols["win"] = rng.integers(0, 2**62, size=(n, 5)).astype(np.uint64)
    return cols


def _worker(rank, world, port, sizes, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, _ = shard_range(1000, rank, world)
    merged = gather_tables(_fake_table(rank, sizes[rank]), hap_offset=lo)
    if rank == 0:
        np.savez(os.path.join(out_dir, "merged.npz"), **merged)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_tables_world2_gloo(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    sizes = [37, 0 + 52]
    mp.spawn(_worker, args=(2, port, sizes, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "merged.npz")
    parts = [_fake_table(r, sizes[r]) for r in range(2)]
    off = [shard_range(1000, r, 2)[0] for r in range(2)]
    for k in got.files:
        want = np.concatenate([p[k] for p in parts])
        if k == "hap":
            want = np.concatenate([np.where(p["hap"] == 0, 0, p["hap"].astype(np.int64) + o).astype(np.uint32)
                                   for p, o in zip(parts, off)])
        assert np.array_equal(got[k], want), k
    assert len(got["hap"]) == sum(sizes)
