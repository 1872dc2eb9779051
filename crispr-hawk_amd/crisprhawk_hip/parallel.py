"""One process per GPU: sample sharding and the guide-table exchange.

Haplotypes are independent units for the scan / filter / scoring kernels
(search_guides.py:111-131, 530-547); the only cross-haplotype inputs are the REF haplotype's
windows (redundancy filter, CFDon wild type), so REF is replicated on every rank and the
samples are block-partitioned.  No collective sits on the data path.  After the search each
rank holds its own guide table; ``gather_tables`` is the single exchange north_star asks for:
an all-gather of the row counts (8 bytes per rank) followed by ONE variable-length gather of the
columns to rank 0 (RCCL over xGMI when the process group is "nccl": every peer has its own link
to rank 0, so a direct all-to-one gather is bound by rank 0's 7-link ingress, not by a ring).
``torch.distributed`` is plumbing only: the columns are filled by the HIP kernels.
"""
from typing import Dict, List, Optional, Tuple

import numpy as np

COLUMNS = (("hap", np.uint32), ("pos", np.uint32), ("strand", np.uint8), ("start", np.int64), ("stop", np.int64),
           ("flags", np.uint8), ("cfdon", np.float64))


def shard_range(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of units [lo, hi) owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def table_columns(tab) -> Dict[str, np.ndarray]:
    tab.download()
    cols = {k: np.ascontiguousarray(getattr(tab, k)) for k, _ in COLUMNS}
    cols["win"] = np.ascontiguousarray(tab.win.T)  # [rows, 5]
    return cols


def gather_tables(cols: Dict[str, np.ndarray], hap_offset: int, group=None, device: Optional[str] = None,
                  dst: int = 0) -> Optional[Dict[str, np.ndarray]]:
    """Concatenate every rank's table on rank `dst`.  ``hap_offset`` is added to this rank's
    local haplotype indices (REF, index 0 on every rank, stays 0) so the merged table indexes the
    global haplotype list.  Returns the merged columns on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.device(device) if device else torch.device("cpu")
    n = len(cols["hap"])
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([n], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group) if dev.type == "cuda" else dist.all_gather(
        list(counts.split(1)), mine, group=group)
    counts = counts.cpu().numpy()
    nmax = int(counts.max())
    hap = cols["hap"].astype(np.int64)
    hap = np.where(hap == 0, 0, hap + hap_offset).astype(np.uint32)  # local haplotype 0 is REF on every rank
    send = dict(cols, hap=hap)
    merged: Dict[str, List[np.ndarray]] = {k: [] for k in send}
    for k, a in send.items():
        a2 = a.reshape(n, -1).view(np.uint8).reshape(n, -1)  # rows of raw bytes
        buf = torch.zeros((nmax, a2.shape[1]), dtype=torch.uint8, device=dev)
        if n:
            buf[:n] = torch.from_numpy(a2).to(dev)
        recv = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, recv, dst=dst, group=group)
        if rank == dst:
            for r in range(world):
                merged[k].append(recv[r][: int(counts[r])].cpu().numpy())
    if rank != dst:
        return None
    out = {}
    for k, parts in merged.items():
        raw = np.concatenate(parts, axis=0)
        dt = send[k].dtype
        out[k] = raw.view(dt).reshape((-1,) + send[k].shape[1:])
    return out


def _gather_var(arr: np.ndarray, group, dev, dst: int):
    """Variable-length gather of a [n, ...] array to `dst`: list of per-rank arrays there, None elsewhere."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = len(arr)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    a2 = np.ascontiguousarray(arr).reshape(n, -1).view(np.uint8).reshape(n, -1)
    width = a2.shape[1] if n else int(np.prod(arr.shape[1:], dtype=np.int64)) * arr.dtype.itemsize
    buf = torch.zeros((nmax, max(width, 1)), dtype=torch.uint8, device=dev)
    if n:
        buf[:n, :width] = torch.from_numpy(a2).to(dev)
    recv = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return [recv[r][: counts[r], :width].cpu().numpy().view(arr.dtype).reshape((-1,) + arr.shape[1:]) for r in range(world)]


def gather_collapsed(cols: Dict[str, np.ndarray], group_perm: np.ndarray, group_off: np.ndarray, is_ref_row: np.ndarray,
                     hap_offset: int, guidelen: int, pamlen: int, group=None, device: Optional[str] = None, dst: int = 0):
    """The exchange after a per-rank `GuideTable.collapse()` (SURVEY §8 f2 x §8e): instead of every row, each rank
    sends one representative row per group of report-identical rows plus the group's member haplotype ids
    (4 B per row instead of 74 B: C3 moves ~130 MB per rank instead of 2.1 GB).  Rank `dst` merges groups that
    occur on several ranks (the same guide carried by samples of different ranks; REF rows, present on every rank)
    by their full key - start, stop, strand, origin, the five core slices - and returns
    ({column: representative rows}, member_off[n_groups + 1], members[global haplotype ids, ascending per group]);
    other ranks return None.  `cols`: table_columns(tab); `is_ref_row`: whether each row is from the REF haplotype."""
    import torch
    dev = torch.device(device) if device else torch.device("cpu")
    perm = np.asarray(group_perm, dtype=np.int64)
    off = np.asarray(group_off, dtype=np.int64)
    reps = perm[off[:-1]]
    hap = cols["hap"].astype(np.int64)
    ghap = np.where(hap == 0, 0, hap + hap_offset)  # local haplotype 0 is REF on every rank
    parts = {k: _gather_var(np.ascontiguousarray(cols[k][reps]), group, dev, dst) for k in cols}
    origin = _gather_var(np.ascontiguousarray(is_ref_row[reps].astype(np.uint8)), group, dev, dst)
    sizes = _gather_var(np.diff(off), group, dev, dst)
    members = _gather_var(ghap[perm].astype(np.uint32), group, dev, dst)
    if parts["hap"] is None:
        return None
    rep = {k: np.concatenate(v) for k, v in parts.items()}
    origin, sizes, members = np.concatenate(origin), np.concatenate(sizes), np.concatenate(members)
    # second-level merge on the full key
    L = guidelen + pamlen
    mask = np.uint64((1 << L) - 1) if L < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    core = (rep["win"] >> np.uint64(10)) & mask
    key = np.zeros(len(sizes), dtype=[("start", np.int64), ("strand", np.uint8), ("stop", np.int64), ("origin", np.uint8),
                                      ("c0", np.uint64), ("c1", np.uint64), ("c2", np.uint64), ("c3", np.uint64), ("c4", np.uint64)])
    key["start"], key["strand"], key["stop"], key["origin"] = rep["start"], rep["strand"], rep["stop"], origin
    for p in range(5):
        key[f"c{p}"] = core[:, p]
    _, first, inv = np.unique(key, return_index=True, return_inverse=True)  # sorted by (start, strand, ...)
    inv = inv.reshape(-1)
    n_groups = len(first)
    grp_of_member = np.repeat(inv, sizes)
    order = np.lexsort((members, grp_of_member))
    gm, mm = grp_of_member[order], members[order]
    keep = np.ones(len(mm), dtype=bool)
    keep[1:] = (gm[1:] != gm[:-1]) | (mm[1:] != mm[:-1])  # REF (0) arrives once per rank
    gm, mm = gm[keep], mm[keep]
    member_off = np.zeros(n_groups + 1, dtype=np.int64)
    np.add.at(member_off, gm + 1, 1)
    member_off = np.cumsum(member_off)
    out = {k: v[first] for k, v in rep.items()}
    out["hap"] = mm[member_off[:-1]].astype(np.uint32)  # a representative's haplotype: the group's first member
    return out, member_off, mm.astype(np.uint32)
