"""N > 1 path on CPU: world_size 2, block partition of the samples and the single variable-length gather of guide
tables (crisprhawk_hip.parallel) - over a gloo process group (tests/util.GlooComm adapts it to the communicator
interface) and over the package's own TcpComm, the stand-ins for RcclComm.  No GPU, no HIP."""
import os
import socket
import sys

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
    sys.path[:0] = [os.path.dirname(os.path.abspath(__file__))]
    from util import GlooComm
    lo, _ = shard_range(1000, rank, world)
    merged = gather_tables(_fake_table(rank, sizes[rank]), hap_offset=lo, comm=GlooComm())
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


def _rank_rows(rank, world):
    """Oracle guide rows of this rank's sample block (REF + its samples' haplotypes), as table columns."""
    from crisprhawk_hip import synth
    from oracle import oracle as ora
    from util import oracle_haplotypes
    reg = synth.make_region(8801, "chrM", 20_000, 1_000, 19_000)
    synth.add_phased_variants(reg, 8802, 150, 8, af_min=0.3, af_max=0.8)
    lo, hi = shard_range(len(reg.samples), rank, world) if world else (0, len(reg.samples))
    fx = dict(region_seq=reg.sequence, startp=reg.startp, samples=reg.samples[lo:hi],
              variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt[lo:hi]]] for v in reg.variants])
    haps = oracle_haplotypes(fx)
    scan = [ora.scan_bounds(h["posmap"], reg.startp, reg.stopp, 3) for h in haps]
    hs = ora.HapSet([h["seq"] for h in haps], [h["posmap"] for h in haps], [h["samples"] == ["REF"] for h in haps], scan)
    res = ora.search(hs, "NGG", 20, False)
    g = res.guides
    n = len(g["start"])
    win = np.zeros((n, 5), dtype=np.uint64)
    code = {c: i for i, c in enumerate("ACGT")}
    for i, w in enumerate(res.windows):  # planes A, C, G, T, V of the 43-nt window, bit j = base j
        for j, ch in enumerate(w):
            win[i, code[ch.upper()]] |= np.uint64(1) << np.uint64(j)
            if ch.islower():
                win[i, 4] |= np.uint64(1) << np.uint64(j)
    cols = dict(hap=g["hap"].astype(np.uint32), pos=g["pos"].astype(np.uint32), strand=g["strand"].astype(np.uint8),
                start=g["start"].astype(np.int64), stop=g["stop"].astype(np.int64), flags=np.zeros(n, np.uint8),
                cfdon=np.full(n, np.nan), win=win)
    isref_row = np.asarray(hs.is_ref)[g["hap"]]
    groups, _ = ora.collapse_rows(g["start"], g["stop"], g["strand"], isref_row, res.windows, 20, 3, False)
    perm, off = [], [0]
    for rows in groups.values():
        perm += rows
        off.append(len(perm))
    labels = [h["samples"] for h in haps]
    return cols, np.array(perm), np.array(off), isref_row, labels, res.windows


def _worker_collapsed(rank, world, port, out_dir):
    import sys
    sys.path[:0] = [os.path.dirname(os.path.abspath(__file__))]
    import torch.distributed as dist
    from crisprhawk_hip.parallel import gather_collapsed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from util import GlooComm
    cols, perm, off, isref_row, labels, _ = _rank_rows(rank, world)
    res = gather_collapsed(cols, perm, off, isref_row, hap_offset=1000 * rank, guidelen=20, pamlen=3, comm=GlooComm())
    if rank == 0:
        rep, moff, members = res
        np.savez(os.path.join(out_dir, "collapsed.npz"), moff=moff, members=members, **rep)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_collapsed_world2_gloo(tmp_path):
    """Two ranks with half of the samples each: representatives + member lists gathered to rank 0 and merged there
    must equal the grouping of the two tables laid side by side."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_collapsed, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "collapsed.npz")
    want = {}
    for r in range(2):
        cols, perm, off, isref_row, labels, wins = _rank_rows(r, 2)
        for i in range(len(cols["hap"])):
            key = (int(cols["start"][i]), int(cols["stop"][i]), int(cols["strand"][i]), bool(isref_row[i]), wins[i][10:-10])
            h = int(cols["hap"][i])
            want.setdefault(key, set()).add(0 if h == 0 else h + 1000 * r)
    moff, members = got["moff"], got["members"]
    assert len(moff) - 1 == len(want) and (np.diff(got["start"]) >= 0).all()
    mask = np.uint64((1 << 23) - 1)
    letters = np.array(list("ACGT"))
    seen = set()
    for g in range(len(moff) - 1):
        core = (got["win"][g] >> np.uint64(10)) & mask
        seq = ""
        for j in range(23):
            bits = [(int(core[p]) >> j) & 1 for p in range(5)]
            ch = letters[bits[:4].index(1)]
            seq += ch.lower() if bits[4] else ch
        mem = set(int(x) for x in members[moff[g]:moff[g + 1]])
        origin = mem == {0}
        key = (int(got["start"][g]), int(got["stop"][g]), int(got["strand"][g]), origin, seq)
        assert key in want and want[key] == mem and key not in seen, key
        assert sorted(mem) == [int(x) for x in members[moff[g]:moff[g + 1]]]
        seen.add(key)


def _worker_tcp(rank, world, rdzv, sizes, out_dir):
    from crisprhawk_hip.parallel import TcpComm
    comm = TcpComm(rank, world, rdzv=rdzv, timeout=60)
    assert comm.bcast_obj(b"id" * 64 if rank == 0 else None) == b"id" * 64  # how the RCCL unique id travels
    got = comm.allgather_i64([rank, 10 * rank + 1])
    assert got.tolist() == [[r, 10 * r + 1] for r in range(world)]
    lo, _ = shard_range(1000, rank, world)
    merged = gather_tables(_fake_table(rank, sizes[rank]), hap_offset=lo, comm=comm)
    if rank == 0:
        np.savez(os.path.join(out_dir, "merged_tcp.npz"), **merged)
    comm.barrier()
    comm.close()


def test_gather_tables_world3_tcp(tmp_path):
    """The package's own socket communicator (bench.py's control plane): rendezvous through a port file, id
    broadcast, count all-gather, variable-length gather with an empty rank."""
    import multiprocessing as mp
    sizes = [11, 0, 23]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker_tcp, args=(r, 3, str(tmp_path / "rdzv"), sizes, str(tmp_path))) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = np.load(tmp_path / "merged_tcp.npz")
    parts = [_fake_table(r, sizes[r]) for r in range(3)]
    off = [shard_range(1000, r, 3)[0] for r in range(3)]
    for k in got.files:
        want = np.concatenate([p[k] for p in parts])
        if k == "hap":
            want = np.concatenate([np.where(p["hap"] == 0, 0, p["hap"].astype(np.int64) + o).astype(np.uint32)
                                   for p, o in zip(parts, off)])
        assert np.array_equal(got[k], want), k


def test_table_gather_plan_world3_with_a_fake_transport():
    """hawk_table_gather's directory / offset arithmetic (hawk_host_gather_plan, the function the RCCL path itself runs)
    for a world of three with one empty rank, driven on the CPU: the transfers every rank is told to post are carried out
    by a byte-copying stand-in for ncclSend / ncclRecv and must reproduce the concatenated table, haplotype indices moved
    into the global numbering as k_hap_shift does."""
    import ctypes as C
    from crisprhawk_hip import _lib
    L = _lib.lib()

    class Op(C.Structure):
        _fields_ = [("col", C.c_uint32), ("peer", C.c_uint32), ("offset", C.c_uint64), ("bytes", C.c_uint64)]
    widths = [4, 4, 1, 8, 8, 1, 8, 8, 8, 8, 8, 8]
    rng = np.random.default_rng(11)
    world, dst = 3, 1
    rows = [5, 0, 7]
    hap_off = [0, 3, 3]
    tabs = [[rng.integers(0, 255, size=n * w, dtype=np.uint8) for w in widths] for n in rows]
    for r in range(world):  # column 0 = haplotype index (u32): make it small numbers, 0 = REF
        tabs[r][0] = rng.integers(0, 3, size=rows[r]).astype(np.uint32).view(np.uint8)
    dir4 = np.array([[rows[r], hap_off[r], 100 + r, 200 + r] for r in range(world)], dtype=np.uint64).reshape(-1)
    plans = []
    for r in range(world):
        off = np.zeros(world + 1, dtype=np.uint64)
        tot = np.zeros(3, dtype=np.uint64)
        n_ops = C.c_uint32(0)
        ops = (Op * (12 * world))()
        rc = L.hawk_host_gather_plan(world, r, dst, dir4.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                     tot.ctypes.data_as(C.c_void_p), C.byref(ops), 12 * world, C.byref(n_ops))
        assert rc == 0
        assert off.tolist() == [0, 5, 5, 12] and tot.tolist() == [12, 303, 603]
        plans.append([(o.col, o.peer, o.offset, o.bytes) for o in ops[:n_ops.value]])
        # counting only
        rc = L.hawk_host_gather_plan(world, r, dst, dir4.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                     tot.ctypes.data_as(C.c_void_p), None, 0, C.byref(n_ops))
        assert rc == 0 and n_ops.value == len(plans[-1])
    assert plans[1 if dst != 1 else 0] and all(p == dst and o == 0 for _, p, o, _ in plans[0])  # a sender: one op per column, to dst
    assert len(plans[0]) == 12 and len(plans[2]) == 12 and len(plans[dst]) == 24  # the empty rank posts nothing, two slices arrive
    merged = [np.zeros(12 * w, dtype=np.uint8) for w in widths]
    sends = {r: {col: nb for col, _, _, nb in plans[r]} for r in range(world) if r != dst}
    for col, peer, offset, nb in plans[dst]:
        src = tabs[peer][col]
        assert len(src) == nb and (peer == dst or sends[peer][col] == nb)  # every recv meets a send of the same size
        merged[col][offset:offset + nb] = src
    want = [np.concatenate([tabs[r][k] for r in range(world)]) for k in range(12)]
    for k in range(12):
        assert np.array_equal(merged[k], want[k])
    hap = merged[0].view(np.uint32).copy()
    for r in range(world):  # k_hap_shift
        sl = slice(int(sum(rows[:r])), int(sum(rows[:r + 1])))
        hap[sl] = np.where(hap[sl] == 0, 0, hap[sl] + hap_off[r])
    local = np.concatenate([tabs[r][0].view(np.uint32) for r in range(world)])
    assert np.array_equal(hap == 0, local == 0) and hap.max() <= 2 + 3
    assert L.hawk_host_gather_plan(3, 3, 0, dir4.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                   tot.ctypes.data_as(C.c_void_p), None, 0, C.byref(n_ops)) == _lib.HAWK_E_INVALID


# ---- region split: one stretch of the interval x all samples per rank, collapsed groups exchanged and merged at the seam ----
import functools


@functools.lru_cache(maxsize=None)
def _region_rows():
    """The UNTILED oracle's guide rows of a small region (all samples) + every haplotype's position map."""
    from crisprhawk_hip import synth
    from oracle import oracle as ora
    from util import oracle_haplotypes
    reg = synth.make_region(8811, "chrM", 24_000, 1_000, 23_000)
    synth.add_phased_variants(reg, 8812, 260, 6, frac_snv=0.7, frac_del=0.15, max_indel=6, af_min=0.3, af_max=0.8)
    fx = dict(region_seq=reg.sequence, startp=reg.startp, samples=reg.samples,
              variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants])
    haps = oracle_haplotypes(fx)
    scan = [ora.scan_bounds(h["posmap"], reg.startp, reg.stopp, 3) for h in haps]
    hs = ora.HapSet([h["seq"] for h in haps], [h["posmap"] for h in haps], [h["samples"] == ["REF"] for h in haps], scan)
    res = ora.search(hs, "NGG", 20, False)
    return reg, hs, res


def _window_planes(windows):
    """[n, 5] plane slices (A, C, G, T, V; bit j = base j) of equally long cased windows"""
    if not windows:
        return np.zeros((0, 5), dtype=np.uint64)
    a = np.frombuffer("".join(windows).encode("ascii"), dtype=np.uint8).reshape(len(windows), -1)
    sh = np.arange(a.shape[1], dtype=np.uint64)
    win = np.zeros((len(windows), 5), dtype=np.uint64)
    for p, ch in enumerate(b"ACGT"):
        win[:, p] = ((((a & 0xDF) == ch).astype(np.uint64)) << sh).sum(axis=1, dtype=np.uint64)
    win[:, 4] = ((((a & 0x20) != 0).astype(np.uint64)) << sh).sum(axis=1, dtype=np.uint64)
    return win


@functools.lru_cache(maxsize=None)
def _stretch_part(rank, world):
    """What rank `rank` of a region-sharded search holds after its own collapse: the report groups of the rows its stretch OWNS
    (a PAM hit belongs to the stretch whose scan range holds its relative position; a seam maps into every haplotype by the
    posmap_rev rule, workload._seam_rel - the rule tests/test_gpu_tiling.py holds the device tiles to), as tiling._groups_of_tile
    lays them out.  Rows come from the untiled oracle run, so haplotype ids are region-wide."""
    from crisprhawk_hip.hapset import PosSegments, segments_from_posmap
    from crisprhawk_hip.workload import _seam_rel
    from oracle import oracle as ora
    reg, hs, res = _region_rows()
    lo, hi = reg.startp + 100, reg.stopp - 100
    seams = [lo + (hi - lo) * r // world for r in range(world + 1)]
    g = res.guides
    rel_lo = np.zeros(len(hs.seqs), dtype=np.int64)
    rel_hi = np.zeros(len(hs.seqs), dtype=np.int64)
    for h, pm in enumerate(hs.posmaps):
        r_, g_ = segments_from_posmap(pm)
        seg = PosSegments(r_, g_, len(pm))
        rel_lo[h] = 0 if rank == 0 else _seam_rel(seg, seams[rank])
        rel_hi[h] = len(pm) + 1 if rank == world - 1 else _seam_rel(seg, seams[rank + 1])
    own = (g["pos"] >= rel_lo[g["hap"]]) & (g["pos"] < rel_hi[g["hap"]])
    idx = np.flatnonzero(own)
    all_w = res.windows
    wins = [all_w[i] for i in idx]
    isref_row = np.asarray(hs.is_ref)[g["hap"][idx]]
    groups, _ = ora.collapse_rows(g["start"][idx], g["stop"][idx], g["strand"][idx], isref_row, wins, 20, 3, False)
    first = np.array([rows[0] for rows in groups.values()], dtype=np.int64)
    order = np.lexsort((g["strand"][idx][first], g["start"][idx][first]))  # a collapsed table is ordered by (start, strand)
    glist = [list(groups.values())[k] for k in order]
    first = first[order]
    n = len(glist)
    win = _window_planes(wins)
    part = {"pos": g["pos"][idx][first].astype(np.uint32), "strand": g["strand"][idx][first].astype(np.uint8),
            "start": g["start"][idx][first].astype(np.int64), "stop": g["stop"][idx][first].astype(np.int64),
            "flags": np.zeros(n, np.uint8), "cfdon": np.full(n, np.nan), "gc_num": np.zeros(n, np.uint8), "gc_den": np.zeros(n, np.uint8),
            "win": win[first], "origin": isref_row[first].astype(np.uint8), "sizes": np.array([len(r) for r in glist], dtype=np.int64),
            "members": np.concatenate([np.sort(g["hap"][idx][np.array(r)]) for r in glist]).astype(np.int64) if n else np.zeros(0, np.int64)}
    return part, (None if rank == 0 else seams[rank]), int(own.sum())


def _worker_region(rank, world, port, out_dir):
    import sys
    sys.path[:0] = [os.path.dirname(os.path.abspath(__file__))]
    import torch.distributed as dist
    from crisprhawk_hip.tiling import gather_tile_groups
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from util import GlooComm
    part, seam, n_own = _stretch_part(rank, world)
    acc, nbytes = gather_tile_groups(GlooComm(), part, seam, 23, 20, 3)
    if rank == 0:
        np.savez(os.path.join(out_dir, "region.npz"), nbytes=nbytes, **acc)
    dist.barrier()
    dist.destroy_process_group()


def test_region_split_world2_gloo_against_the_untiled_oracle(tmp_path):
    """Two ranks, each a stretch of the region x all samples: per-rank report groups -> gather_tile_groups -> merged at the seam
    on rank 0 must be the grouping of the untiled oracle's rows (groups whose members lie on both sides of the seam - an indel
    upstream shifts a haplotype's PAM position across it - are united; no row is lost or counted twice)."""
    import torch.multiprocessing as mp
    from oracle import oracle as ora
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_region, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "region.npz")
    reg, hs, res = _region_rows()
    g = res.guides
    isref_row = np.asarray(hs.is_ref)[g["hap"]]
    want = {}
    all_w = res.windows
    for i in range(len(g)):
        key = (int(g["start"][i]), int(g["stop"][i]), int(g["strand"][i]), bool(isref_row[i]), all_w[i][10:-10])
        want.setdefault(key, set()).add(int(g["hap"][i]))
    # every row is owned by exactly one stretch
    owned = [_stretch_part(r, 2)[2] for r in range(2)]
    assert sum(owned) == len(g) and min(owned) > 0
    sizes, members = got["sizes"], got["members"]
    assert len(sizes) == len(want) and int(sizes.sum()) == len(members)
    assert (np.diff(got["start"]) >= 0).all()
    moff = np.concatenate(([0], np.cumsum(sizes)))
    mask = np.uint64((1 << 23) - 1)
    seen = set()
    for k in range(len(sizes)):
        core = (got["win"][k] >> np.uint64(10)) & mask
        seq = ""
        for j in range(23):
            bits = [(int(core[p]) >> j) & 1 for p in range(5)]
            ch = "ACGT"[bits[:4].index(1)]
            seq += ch.lower() if bits[4] else ch
        mem = set(int(x) for x in members[moff[k]:moff[k + 1]])
        key = (int(got["start"][k]), int(got["stop"][k]), int(got["strand"][k]), bool(got["origin"][k]), seq)
        assert key in want and want[key] == mem and key not in seen, key
        seen.add(key)
    # both stretches have groups inside the seam's merge window (tiling._merge_at_seam's `near` path ran on real groups)
    seam = _stretch_part(1, 2)[1]
    parts = [_stretch_part(r, 2)[0] for r in range(2)]
    assert (parts[0]["start"] > seam - 87).any() and (parts[1]["start"] < seam + 87).any()
    assert int(got["nbytes"]) > 0
