"""Shared helpers for the parity tests: golden-fixture loading and the haplotype
bookkeeping (sample -> chromosome copy -> variant list, collapse of identical sequences)
needed to drive the oracle from a fixture's raw inputs."""
import gzip
import json
import os
from collections import OrderedDict

import numpy as np

from oracle import oracle as ora

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G3_CASES = ["c1", "phased4", "phased16", "cpf1", "iupac", "indel_dense", "tiny", "ngn"]


def load_golden(name):
    with gzip.open(os.path.join(GOLDEN, name), "rb") as f:
        return json.loads(f.read().decode())


def posmap_from_breaks(breaks, n):
    pm = np.empty(n, dtype=np.int64)
    for k, (i, g) in enumerate(breaks):
        j = breaks[k + 1][0] if k + 1 < len(breaks) else n
        pm[i:j] = g + np.arange(j - i)
    return pm


def copy_variants(fx):
    """[(sample, copy) -> [(pos, ref, alt)]] in the reference's iteration order
    (haplotypes.py:132-159: samples in VCF order, only samples with >= 1 variant)."""
    samples = fx["samples"]
    per = OrderedDict((s, ([], [])) for s in samples)
    for pos, ref, alt, _af, gt in fx["variants"]:
        for si, row in enumerate(gt):
            for c in (0, 1):
                if row[c] == "1":
                    per[samples[si]][c].append((pos, ref, alt))
    return OrderedDict((s, v) for s, v in per.items() if v[0] or v[1])


def oracle_haplotypes(fx):
    """Rebuild the haplotype list through the ORACLE from the fixture's raw inputs, in the
    reference's order: REF, then per sample copy 0 / copy 1 (one entry when both copies are
    identical, haplotypes.py:326-333), collapsed by identical cased sequence keeping the
    first member's position map (haplotypes.py:232-294)."""
    ref = fx["region_seq"]
    startp = fx["startp"]
    entries = [(ref, np.arange(startp, startp + len(ref), dtype=np.int64), "REF")]
    for s, (v0, v1) in copy_variants(fx).items():
        h0 = ora.hap_build(ref, startp, v0)
        h1 = ora.hap_build(ref, startp, v1)
        if h0[0] == h1[0]:
            entries.append((h0[0], h0[1], f"{s}:1|1"))
        else:
            entries.append((h0[0], h0[1], f"{s}:1|0"))
            entries.append((h1[0], h1[1], f"{s}:0|1"))
    groups = OrderedDict()
    for seq, pm, smp in entries:
        groups.setdefault(seq, []).append((pm, smp))
    out = []
    for seq, members in groups.items():
        samples = ["REF"] if seq.isupper() else sorted({m[1] for m in members})
        out.append(dict(seq=seq, posmap=members[0][0], samples=samples))
    return out


def hapset_from_golden(fx):
    """Oracle HapSet straight from the reference-built haplotypes stored in the fixture."""
    seqs = [h["seq"] for h in fx["haplotypes"]]
    pms = [posmap_from_breaks(h["posmap_breaks"], h["posmap_len"]) for h in fx["haplotypes"]]
    is_ref = [h["samples"] == ["REF"] for h in fx["haplotypes"]]
    scan = [tuple(s) for s in fx["scan"]]
    return ora.HapSet(seqs, pms, is_ref, scan)


def synth_region_from_fixture(fx):
    """Rebuild the SynthRegion a g3 fixture was generated from (only the padded region's
    bases are stored; the contig prefix is irrelevant to every consumer)."""
    from crisprhawk_hip import synth
    reg = synth.SynthRegion(fx["contig"], "N" * (fx["startp"] - 1) + fx["region_seq"], fx["bed_start"], fx["bed_stop"])
    reg.samples = fx["samples"]
    reg.variants = [synth.VariantSite(p, r, a, af, np.array([[int(c) for c in row] for row in gt], dtype=np.uint8))
                    for p, r, a, af, gt in fx["variants"]]
    return reg


class GlooComm:
    """The communicator interface of crisprhawk_hip.parallel (rank, world, barrier, allgather_i64, gatherv_bytes) over a
    torch.distributed gloo process group: the CPU stand-in the world_size-2 tests run the exchange logic on."""

    def __init__(self):
        import torch.distributed as dist
        self._dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def barrier(self):
        self._dist.barrier()

    def allgather_i64(self, vec):
        import torch
        mine = torch.tensor([int(v) for v in vec], dtype=torch.int64)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self._dist.all_gather(out, mine)
        return torch.stack(out).numpy()

    def gatherv_bytes(self, arr, dst=0):
        import numpy as np
        import torch
        a = np.ascontiguousarray(arr)
        n = len(a)
        counts = self.allgather_i64([n])[:, 0]
        width = int(np.prod(a.shape[1:], dtype=np.int64)) * a.dtype.itemsize if a.ndim > 1 else a.dtype.itemsize
        buf = torch.zeros((int(counts.max()), max(width, 1)), dtype=torch.uint8)
        if n:
            buf[:n, :width] = torch.from_numpy(a.reshape(n, -1).view(np.uint8).reshape(n, -1).copy())
        recv = [torch.empty_like(buf) for _ in range(self.world)] if self.rank == dst else None
        self._dist.gather(buf, recv, dst=dst)
        if self.rank != dst:
            return None
        return [recv[r][: int(counts[r]), :width].numpy().copy().view(a.dtype).reshape((-1,) + a.shape[1:]) for r in range(self.world)]
