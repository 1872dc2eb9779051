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
