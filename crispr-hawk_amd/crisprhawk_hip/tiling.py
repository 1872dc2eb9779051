"""Region tiling: one large region (a whole contig) searched tile by tile inside a fixed HBM budget.

The reference treats a BED interval as ONE region however long it is (`search_guides.search`,
search_guides.py:510-548, over haplotypes built for the whole interval, haplotypes.py:337-368): chr22 x 5009
haplotypes would be 159 GB of planes.  Here the interval is cut at genomic seams; each tile is expanded
(hawk_xplan_run), searched (hawk_search) and collapsed (hawk_table_collapse) on its own and leaves only its report
groups behind.  What makes the result identical to the untiled search:

* ownership - a PAM hit belongs to the tile whose scan range holds its relative position; a seam maps into every
  haplotype by the reference's own posmap_rev rule (last relative position of the genomic position, walking forward over
  deletions), the same rule on both sides, so each haplotype position is scanned exactly once (workload.ScanOwnership);
* flanks - a tile's string reaches `flank` bases past its seams, enough for the guidelen+pamlen+20-nt window of every
  owned hit (checked per haplotype when the tile is prepared); the first and last tile end where the region ends, so
  is_pamhit_in_range (search_guides.py:395-420) sees the true region ends;
* REF partners - remove_redundant_guides and CFDon pair a haplotype guide with the REF guide at its (start, strand)
  (search_guides.py:340-369, scoring.py:368-381) even when that REF guide's PAM lies across the seam: every tile tells the
  kernels the REGION's REF scan range (hawk_hapset_set_ref_partner_range);
* groups - report groups (reports.py:958-1008) of neighbouring tiles are merged on their full key around each seam.

Haplotype identity is per tile (two copies that differ only outside a tile are one row there); the report merges
identical rows and unites their samples, so the collapsed report does not depend on it.
"""
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib, parallel
from .hapset import GroupTable
from .workload import HapInfo, RowLabel, ScanOwnership, _expand_rows_gt, _ref_only_set, _variant_table, invert_on_device

PADDING = 100  # region_constructor.py:21


class DenseGenotypes:
    """Genotypes as a [variant, chromosome copy] 0/1 matrix (column = 2 * sample + copy)."""

    def __init__(self, G: np.ndarray):
        self.G = np.asarray(G, dtype=np.uint8)
        self.n_cols = self.G.shape[1]

    def dense(self, var_lo: int, var_hi: int, col_lo: int, col_hi: int) -> np.ndarray:
        return self.G[var_lo:var_hi, col_lo:col_hi]

    def carried(self, var_lo: int, var_hi: int, col_lo: int, col_hi: int):
        """-> (counts[col_hi - col_lo], variant indices relative to var_lo, ordered by column then variant)"""
        sub = self.G[var_lo:var_hi, col_lo:col_hi]
        cols, sites = np.nonzero(sub.T)
        return np.bincount(cols, minlength=col_hi - col_lo), sites.astype(np.uint32)


@dataclass
class VariantPanel:
    """Position-sorted biallelic variant table of one contig + who carries what."""
    pos: np.ndarray            # int64, 1-based, ascending
    ref: List[str]
    alt: List[str]
    vid: List[str]
    af: np.ndarray             # float64 (NaN = missing)
    samples: List[str]
    genotypes: object          # .carried(var_lo, var_hi, col_lo, col_hi)
    ref_len: np.ndarray = field(default=None)

    def __post_init__(self):
        self.pos = np.asarray(self.pos, dtype=np.int64)
        if self.ref_len is None:
            self.ref_len = np.array([len(r) for r in self.ref], dtype=np.int64)

    @classmethod
    def from_region(cls, reg) -> "VariantPanel":
        """A synth.SynthRegion's variant list as a panel."""
        vs = reg.variants
        G = np.stack([v.gt.reshape(-1) for v in vs]) if vs else np.zeros((0, 2 * len(reg.samples)), np.uint8)
        return cls(np.array([v.pos for v in vs], dtype=np.int64), [v.ref for v in vs], [v.alt for v in vs],
                   [f"{reg.contig}-{v.pos}-{v.ref}/{v.alt}" for v in vs], np.array([v.af for v in vs], dtype=np.float64),
                   list(reg.samples), DenseGenotypes(G))


@dataclass
class Tile:
    index: int
    seq_lo: int                 # genomic position of the tile string's first base
    seq_hi: int                 # genomic position of its last base
    own_lo: Optional[int]       # seam where its scan range starts (None: the region's start rule)
    own_hi: Optional[int]       # seam where it stops (None: the region's stop rule)


def plan_tiles(startp: int, stopp: int, tile_nt: int, flank: int) -> List[Tile]:
    """Cut the BED interval [startp + 100, stopp - 100) at multiples of tile_nt; every tile string carries `flank`
    bases beyond its seams, the outer tiles end with the (padded) region."""
    lo, hi = startp + PADDING, stopp - PADDING
    seams = list(range(lo, hi, max(1, tile_nt))) + [hi]
    if len(seams) > 2 and seams[-1] - seams[-2] < max(2 * flank, tile_nt // 4):  # no sliver at the end
        seams.pop(-2)
    tiles = []
    for t in range(len(seams) - 1):
        first, last = t == 0, t == len(seams) - 2
        tiles.append(Tile(t, startp if first else max(startp, seams[t] - flank), stopp if last else min(stopp, seams[t + 1] + flank),
                          None if first else seams[t], None if last else seams[t + 1]))
    return tiles


class PreparedTile:
    """One tile ready to run: its expansion plan (inputs + row metadata in HBM), row labels and ownership."""

    def __init__(self, tile: Tile, ds, plan, labels: List[Optional[RowLabel]], n_hap: int):
        self.tile, self.plan, self.labels, self.n_hap = tile, plan, labels, n_hap
        self._first = ds  # the set the preparation run produced; handed out once

    def expand(self):
        """A DeviceHapSet of the tile's rows (device work only after the first call)."""
        if self._first is not None:
            ds, self._first = self._first, None
            return ds
        if self.plan is None:
            raise RuntimeError("REF-only tile has no plan: keep the prepared set")
        ds, _, _ = self.plan.run()
        return ds

    def close(self):
        if self._first is not None:
            self._first.close()
            self._first = None
        if self.plan is not None:
            self.plan.close()
            self.plan = None


class TiledRegionSearch:
    """Whole-region guide search, tile by tile.

    `fetch(lo, hi)` returns the contig bases of genomic positions [lo, hi] (1-based, inclusive) as str / bytes / uint8
    array; `startp` / `stopp` are the padded region coordinates (Coordinate.start / .stop, coordinate.py:118-136).
    `sample_range`: this rank's block of the panel (haplotypes shard across GPUs, REF on every rank)."""

    def __init__(self, fetch: Callable[[int, int], object], contig: str, startp: int, stopp: int, panel: Optional[VariantPanel], pam,
                 guidelen: int, right: bool, tile_nt: int = 4_000_000, flank: Optional[int] = None, device: Optional[int] = None,
                 sample_range: Optional[Tuple[int, int]] = None):
        self.fetch, self.contig, self.startp, self.stopp, self.panel = fetch, contig, startp, stopp, panel
        self.pam, self.guidelen, self.right, self.device = pam, guidelen, bool(right), device
        self.fused = True  # False: materialise every tile's planes (hawk_xplan_run) and search those - round 2's step, for A/B
        self.L = guidelen + len(pam)
        self.guard = self.L + 2 * 10 + 8
        self.flank = flank if flank is not None else max(1024, 8 * self.guard)
        if self.flank < self.guard + 8:
            raise ValueError("flank smaller than a padded guide window")
        self.tiles = plan_tiles(startp, stopp, tile_nt, self.flank)
        n_s = len(panel.samples) if panel is not None else 0
        self.sample_range = sample_range if sample_range is not None else (0, n_s)
        self.prepared: List[Optional[PreparedTile]] = [None] * len(self.tiles)

    # ---- preparation (host index work + one expansion per tile) ---------------------------------
    def prepare_tile(self, t: int, keep_plan: bool = True) -> PreparedTile:
        tile = self.tiles[t]
        pamlen = len(self.pam)
        seq = self.fetch(tile.seq_lo, tile.seq_hi)
        if isinstance(seq, (bytes, bytearray)):
            seq = bytes(seq).decode("ascii")
        elif isinstance(seq, np.ndarray):
            seq = seq.tobytes().decode("ascii")
        if len(seq) != tile.seq_hi - tile.seq_lo + 1:
            raise ValueError("fetch() returned a string of the wrong length")
        seq = seq.upper()  # soft-masked genomes: the reference upper-cases every region (sequence.py:49); lower case means "variant" here
        # the region's scan range in this tile's REF coordinates (REF's position map is the identity)
        reg_lo = self.startp + PADDING - tile.seq_lo
        reg_hi = (self.stopp - PADDING) - pamlen + 1 - tile.seq_lo
        p_lo = min(len(seq), max(0, reg_lo))
        own = ScanOwnership(tile.own_lo, tile.own_hi, (p_lo, max(p_lo, min(len(seq), reg_hi))), self.guard)
        ref_set = _ref_only_set(seq, tile.seq_lo, tile.seq_hi, pamlen, self.device, own)
        ref_label = RowLabel("REF", "NA", {}, f"hap_t{t}_ref", ref_set.host_meta[0].seg)
        slo, shi = self.sample_range
        p = self.panel
        if p is None or shi <= slo or len(p.pos) == 0:
            return self._keep(t, PreparedTile(tile, ref_set, None, [ref_label], 1))
        # variants that lie wholly inside the tile string
        v_lo = int(np.searchsorted(p.pos, tile.seq_lo, side="left"))
        v_hi = int(np.searchsorted(p.pos, tile.seq_hi, side="right"))
        while v_hi > v_lo and p.pos[v_hi - 1] + p.ref_len[v_hi - 1] - 1 > tile.seq_hi:
            v_hi -= 1
        # A record that only partly lies in the tile string is left out: harmless while it stays clear of everything the
        # tile owns (its neighbour carries it whole).  One that reaches to within flank + guard of the owned range - an
        # indel or SV longer than the flank - would map the seam differently on its two sides: refuse, the tiles need a
        # larger flank (TiledRegionSearch(flank=...)).
        reach = self.guard
        k = v_lo - 1  # the last record starting before the tile string: does its REF span reach in?  (interior seams only:
        #               at the region's own ends a straddling record is outside the region, as for the one-piece search)
        if tile.own_lo is not None and k >= 0 and p.pos[k] + p.ref_len[k] - 1 >= tile.seq_lo and p.pos[k] + p.ref_len[k] - 1 >= tile.own_lo - reach:
            raise ValueError(f"variant at {int(p.pos[k])} (REF span {int(p.ref_len[k])}) crosses the start of tile {t}'s string into what the "
                             "tile scans: the flank is too small for this record")
        if tile.own_hi is not None and v_hi < len(p.pos) and p.pos[v_hi] <= tile.seq_hi and p.pos[v_hi] <= tile.own_hi + reach:
            raise ValueError(f"variant at {int(p.pos[v_hi])} (REF span {int(p.ref_len[v_hi])}) crosses the end of tile {t}'s string inside what "
                             "the tile scans: the flank is too small for this record")
        if v_hi <= v_lo:
            return self._keep(t, PreparedTile(tile, ref_set, None, [ref_label], 1))
        tab = _variant_table(p.pos[v_lo:v_hi], p.ref[v_lo:v_hi], p.alt[v_lo:v_hi], seq, tile.seq_lo)
        r0, chain = tab[0], tab[2]
        # the tile's genotype rectangle is inverted into per-copy carried lists on the device (hawk_gt_lists)
        # ... and stays there: plan, checks, segments and the seams' reverse look-ups are built from the lists in place
        g, col_off = invert_on_device(ref_set._ctx, p.genotypes.dense(v_lo, v_hi, 2 * slo, 2 * shi), r0, chain)
        if int(col_off[-1]) == 0:
            _lib.lib().hawk_gt_destroy(g)
            return self._keep(t, PreparedTile(tile, ref_set, None, [ref_label], 1))
        ds, info, _ms, kept = _expand_rows_gt(ref_set, seq, tile.seq_lo, tile.seq_hi, pamlen, p.samples[slo:shi], tab, g, col_off,
                                              self.device, keep_plan=keep_plan, own=own)
        ref_set.close()
        labels: List[Optional[RowLabel]] = [None] * ds.n_hap
        vid, af, ref, alt = p.vid, p.af, p.ref, p.alt
        for k, (r, inf) in enumerate(zip(kept, info)):
            if r == 0:
                labels[0] = RowLabel("REF", "NA", {}, f"hap_t{t}_ref", ds.host_meta[0].seg)
                continue
            idx = [v_lo + int(i) for i in inf.variant_idx]
            snv = [i for i in idx if len(ref[i]) == len(alt[i])]
            indel = [i for i in idx if len(ref[i]) != len(alt[i])]
            ids = [vid[i] for i in snv + indel]  # haplotype.py:234-242: SNVs first, then indels
            labels[r] = RowLabel(",".join(inf.samples), ",".join(ids) if ids else "NA", {vid[i]: float(af[i]) for i in idx},
                                 f"hap_t{t}_{k:08d}", ds.host_meta[r].seg)
        plan = getattr(ds, "plan", None)
        return self._keep(t, PreparedTile(tile, ds, plan, labels, ds.n_hap))

    def _keep(self, t: int, pt: PreparedTile) -> PreparedTile:
        self.prepared[t] = pt
        return pt

    # ---- the per-tile device step -----------------------------------------------------------------
    def run_tile(self, t: int, cfd=None, flank_key: Tuple[int, int] = (0, 0), cfd_na_on_ambiguous: bool = True, export: bool = True):
        """search (+ CFDon) straight from the tile's plan -> collapse (-> group export) of tile t: no haplotype plane is
        written (hawk_xplan_view); a tile without variants searches its REF set.  Returns (GroupTable or None, stats)."""
        pt = self.prepared[t] or self.prepare_tile(t)
        fused = pt.plan is not None and self.fused
        if fused and pt._first is not None:  # the preparation run's planes (content hashes) are not needed again
            pt._first.close()
            pt._first = None
        ds = pt.plan.view() if fused else pt.expand()
        try:
            mm, ptab = cfd if cfd is not None else (None, None)
            tab = ds.search(self.pam.bits, self.pam.bitsrc, len(self.pam), self.guidelen, self.right, mm, ptab, download=False,
                            cfd_na_on_ambiguous=cfd_na_on_ambiguous)
            stats = {"tile": t, "rows": tab.n_rows, "candidates": tab.n_candidates, "hits": tab.n_hits, "n_hap": ds.n_hap,
                     "search_ms": tab.timing["total_ms"], "scanned_positions": tab.timing["scanned_positions"],
                     "v_count_ms": tab.timing.get("v_count_ms", 0.0), "v_emit_ms": tab.timing.get("v_emit_ms", 0.0),
                     "v_emit_rows_ms": tab.timing.get("v_emit_rows_ms", 0.0),
                     "records": int(getattr(pt.plan, "n_records", 0)) if pt.plan is not None else 0,
                     "v_path": int(tab.timing.get("v_path", 0)),
                     "instances": (pt.plan.cluster_stats()["instances"] if pt.plan is not None and int(tab.timing.get("v_path", 0)) == 2 else 0)}
            tab.collapse(flank_key, download_perm=False)
            stats["collapse_ms"], stats["groups"] = tab.collapse_ms, tab.n_groups
            g = None
            if export:
                g = tab.export_groups()
                g.is_ref_hap = np.asarray(ds.is_ref, dtype=bool)
            tab.close()
            return g, stats
        finally:
            if pt.plan is None:
                pt._first = ds  # REF-only tile: the prepared set is the only copy
            elif not fused:
                ds.close()      # (a view lives and dies with its plan)

    # ---- the whole region ---------------------------------------------------------------------------
    def run(self, cfd=None, flank_key: Tuple[int, int] = (0, 0), cfd_na_on_ambiguous: bool = True, keep_plans: bool = False):
        """All tiles; returns a MergedGroups (report groups of the whole region, seam groups merged)."""
        acc = None
        stats = []
        base = 0
        labels: List[Optional[RowLabel]] = []
        for t in range(len(self.tiles)):
            pt = self.prepared[t] or self.prepare_tile(t, keep_plan=keep_plans)
            g, st = self.run_tile(t, cfd, flank_key, cfd_na_on_ambiguous)
            stats.append(st)
            part = _groups_of_tile(g, base)
            labels.extend(pt.labels)
            base += pt.n_hap
            seam = self.tiles[t].own_lo
            acc = part if acc is None else _merge_at_seam(acc, part, seam, self.L + 64, self.guidelen, len(self.pam), flank_key)
            if not keep_plans:
                pt.close()
                self.prepared[t] = None
        return MergedGroups(acc, labels, self.guidelen, len(self.pam), self.right, stats)

    def close(self):
        for pt in self.prepared:
            if pt is not None:
                pt.close()
        self.prepared = [None] * len(self.tiles)


# ---------------------------------------------------------------------------------------------------
# report groups across tiles
# ---------------------------------------------------------------------------------------------------
_REPCOLS = ("pos", "strand", "start", "stop", "flags", "cfdon", "gc_num", "gc_den")


def _groups_of_tile(g: GroupTable, hap_base: int) -> Dict[str, np.ndarray]:
    """A tile's groups as plain arrays with haplotype ids moved into the region-wide numbering."""
    members = g.member_hap.astype(np.int64) + hap_base
    first = g.member_hap[g.member_off[:-1]] if g.n_groups else np.zeros(0, np.uint32)
    out = {k: getattr(g, k) for k in _REPCOLS}
    out["win"] = np.ascontiguousarray(g.win.T)
    out["origin"] = g.is_ref_hap[first].astype(np.uint8)
    out["sizes"] = np.diff(g.member_off)
    out["members"] = members
    return out


def _take(part: Dict[str, np.ndarray], idx: np.ndarray) -> Dict[str, np.ndarray]:
    """The groups `idx` (ascending) of a part, CSR members included."""
    off = np.concatenate(([0], np.cumsum(part["sizes"])))
    sizes = part["sizes"][idx]
    if len(idx):
        starts = off[idx]
        rep = np.repeat(np.arange(len(idx)), sizes)
        within = np.arange(int(sizes.sum())) - np.repeat(np.cumsum(sizes) - sizes, sizes)
        members = part["members"][starts[rep] + within]
    else:
        members = np.zeros(0, np.int64)
    out = {k: v[idx] for k, v in part.items() if k not in ("members",)}
    out["members"] = members
    return out


def _concat(parts: Sequence[Dict[str, np.ndarray]]) -> Dict[str, np.ndarray]:
    return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}


def _merge_at_seam(acc, part, seam: int, reach: int, guidelen: int, pamlen: int, flank_key):
    """Groups of the accumulated result and of the next tile that start within `reach` of the seam are merged on the
    full key; everything else passes through (both inputs are ordered by start)."""
    a_near = np.flatnonzero(acc["start"] > seam - reach)
    b_near = np.flatnonzero(part["start"] < seam + reach)
    a_far = np.flatnonzero(acc["start"] <= seam - reach)
    b_far = np.flatnonzero(part["start"] >= seam + reach)
    near = _concat([_take(acc, a_near), _take(part, b_near)])
    if len(near["sizes"]):
        rep = {k: near[k] for k in ("pos", "strand", "start", "stop", "flags", "cfdon", "gc_num", "gc_den", "win", "origin")}
        merged, moff, mem = parallel.merge_groups(dict(rep), near["origin"], near["sizes"], near["members"], guidelen, pamlen, flank_key)
        merged.pop("hap", None)
        merged["sizes"] = np.diff(moff)
        merged["members"] = mem.astype(np.int64)
        near = merged
    return _concat([_take(acc, a_far), near, _take(part, b_far)])


def gather_tile_groups(comm, part: Optional[Dict[str, np.ndarray]], seam_lo: Optional[int], L: int, guidelen: int, pamlen: int,
                       flank_key: Tuple[int, int] = (0, 0), dst: int = 0):
    """The exchange of a REGION-sharded search (one stretch of the interval x all samples per rank, ranks in stretch order): every
    rank hands its collapsed report groups (`_groups_of_tile`: one representative row per group + the members' haplotype ids) to
    `dst`, which folds them together in rank order, merging on the full key around each seam (`seam_lo` = the genomic position
    where this rank's stretch starts, None for the first).  What crosses the links is 75 B per report group + 8 B per member -
    not the 64-byte guide rows.  Returns (parts dict, bytes received) on `dst`, (None, 0) elsewhere."""
    if part is None:
        raise ValueError("every rank contributes a (possibly empty) part")
    keys = sorted(part.keys())
    seams = comm.allgather_i64([-1 if seam_lo is None else int(seam_lo)])[:, 0]
    got = {k: comm.gatherv_bytes(np.ascontiguousarray(part[k]), dst) for k in keys}
    if comm.rank != dst:
        return None, 0
    nbytes = sum(a.nbytes for k in keys for r, a in enumerate(got[k]) if r != dst)
    acc = None
    for r in range(comm.world):
        p = {k: got[k][r] for k in keys}
        acc = p if acc is None else _merge_at_seam(acc, p, int(seams[r]), L + 64, guidelen, pamlen, flank_key)
    return acc, nbytes


class MergedGroups:
    """Report groups of a whole region: representative columns, CSR members over the region-wide haplotype numbering
    and one RowLabel per haplotype row."""

    def __init__(self, parts: Dict[str, np.ndarray], labels, guidelen: int, pamlen: int, right: bool, stats):
        self.cols, self.labels, self.guidelen, self.pamlen, self.right, self.stats = parts, labels, guidelen, pamlen, right, stats
        self.member_off = np.concatenate(([0], np.cumsum(parts["sizes"]))).astype(np.int64)
        self.members = parts["members"]
        self.n_groups = len(parts["sizes"])

    def groups(self):
        """reports.ReportGroups view (the input of reports.report_from_groups)."""
        from .reports import ReportGroups
        c = self.cols
        return ReportGroups(self.guidelen, self.pamlen, self.right, c["pos"], c["strand"], c["start"], c["stop"], c["cfdon"],
                            np.ascontiguousarray(c["win"].T), c["gc_num"], c["gc_den"], self.member_off, self.members)

    def report_input(self):
        """A reports.ReportInput over pseudo-rows (one per member; the representative's fields at a group's first row)."""
        from .hapset import decode_windows
        from .reports import ReportInput
        c = self.cols
        n = len(self.members)
        first = self.member_off[:-1]
        W = self.guidelen + self.pamlen + 20

        def spread(a, fill=0):
            out = np.full(n, fill, dtype=a.dtype)
            out[first] = a
            return out
        wins: List[Optional[str]] = [None] * n
        for i, w in zip(first.tolist(), decode_windows(np.ascontiguousarray(c["win"].T), W)):
            wins[i] = w
        return ReportInput(spread(c["start"]), spread(c["stop"]), spread(c["strand"]), self.members.astype(np.int64), spread(c["pos"]),
                           wins, spread(c["cfdon"], np.nan), np.arange(n, dtype=np.int64), self.member_off, c["gc_num"], c["gc_den"],
                           self.guidelen, self.pamlen, self.right)
