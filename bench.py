#!/usr/bin/env python3
"""bench.py — candidate guides scored per second on the BASELINE.json workloads.

    python bench.py [--gpus N --steps K --warmup W] [--config c3|c1|c4|c5] [--weak]

A "step" is one pass of the device hot path over inputs that are already resident in HBM:

  c3 (default, the configuration the metric is quoted on: 1 Mb region x 2504 phased samples, NGG, 20 nt)
      hawk_search (PAM scan fused with the in-range / REF-identical filters -> redundancy verdict -> coordinates ->
      window gather -> CFDon) over the haplotype planes.
  c1  the same on the 10 kb variant-free region (latency case; one haplotype: ranks are replicas).
  c4  whole chr22-sized contig x 2504 samples, region-tiled: per tile hawk_xplan_run (expansion) -> hawk_search ->
      hawk_table_collapse; resident inputs are the REF planes, the variant table and the carried-variant lists.
  c5  off-target enumeration: TTTV / 23 nt / <= 4 mismatches against a packed synthetic genome (hawk_offtarget_scan).

N > 1: one process per GPU.  Launched by the driver under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in the environment) or, when `--gpus N` is given and WORLD_SIZE is not set, by this script itself (N child
processes started before anything touches the GPU).  The ONE haplotype set is block-partitioned over the ranks
(strong scaling, REF on every rank; `--weak` gives every rank its own 2504-sample panel instead); c5 shards the genome.
There is no data-path collective: barriers and the timing / count reductions run over the package's TCP control plane,
and the single exchange of the job - the RCCL gather of the guide tables over xGMI (hawk_table_gather) - is measured
once after the timed loop and reported as `gather` next to `value`.

One JSON line on stdout (rank 0): the driver's contract fields plus `roofline` (dominant kernel, algorithmic bytes over
its HIP-event duration, against the 8 TB/s HBM peak), `cpu_baseline` (the C oracle on one host thread, bounded sample),
`cpu_baseline_all_cores`, `end_to_end` (in-memory records -> expand -> search+CFDon -> collapse -> D2H of the report
groups, wall clock) and per-kernel times.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "crispr-hawk_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_LANE_OPS = 78.6e12  # 256 CUs x 4 SIMDs x 32 lanes/cycle x 2.4 GHz (MI355X_MICROARCH.md: F32 vector peak 157.3 TF = 2 flop x this)
# Algorithmic bytes of one fused launch: every scanned haplotype position is read once as
# 0.5 B of IUPAC code + 0.125 B of the variant plane (K3's REF-identical filter);
# the emit pass additionally writes each guide row once (74 B: K3 record + packed window + K4 score).
READ_BYTES_PER_POS = 0.625
ROW_BYTES = 74
LIST_BYTES = 4  # one u32 hand-over entry per guide row (count pass writes it, k_emit_list reads it)
REC_BYTES = 32  # one record per carried variant of a chromosome copy (hawk_hx.h HxVar): what the fused step reads instead of planes
PROFILE_TRAFFIC = os.path.join(ROOT, "profiles", "r02_traffic.json")
PROFILE_TRAFFIC_R3 = os.path.join(ROOT, "profiles", "r03_traffic.json")
PROFILE_C5_PMC = os.path.join(ROOT, "profiles", "r04_c5_pmc.json")  # per-wave SQ counters of the off-target kernels at the full C5 size
REFERENCE_TIMING = os.path.join(ROOT, "profiles", "r02_reference_python_timing.json")  # tools/time_reference.py, build container


def log(msg):
    print(f"[bench r{os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def usable_cores() -> int:
    """Host threads this process may really use: the affinity mask, cut by the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


# ---------------------------------------------------------------------------------------------------
# launcher: `--gpus N` without a torch.distributed.run environment
# ---------------------------------------------------------------------------------------------------
def launch(n: int, argv) -> int:
    """Start N rank processes (before this process has touched the GPU) and relay rank 0's stdout."""
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rdzv = tempfile.mkdtemp(prefix="hawk_rdzv_")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HAWK_RDZV_DIR=rdzv)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:  # a failed rank leaves the others waiting at a barrier: stop exactly the processes started here
                    q.terminate()
    try:
        os.rmdir(rdzv)
    except OSError:
        pass
    return rc


class Ranks:
    """rank / world / device of this process + the control-plane communicator (TCP; no torch, no RCCL)."""

    def __init__(self, args):
        from crisprhawk_hip import _lib, parallel
        self.rank, self.world, self.local = parallel.env_rank_world()
        self.backend = os.environ.get("HAWK_BENCH_BACKEND", "rccl")
        if self.backend == "gloo":  # older name of the rehearsal backend
            self.backend = "tcp"
        ndev = _lib.device_count()
        if ndev == 0:
            raise _lib.HawkDeviceError("bench.py needs an MI355X: there is no CPU fallback on the product path")
        # rehearsal on a box with fewer GPUs than ranks (HAWK_BENCH_ONE_GPU=1 or the tcp backend): wrap around
        self.device = self.local % ndev if (os.environ.get("HAWK_BENCH_ONE_GPU") == "1" or self.backend == "tcp") else self.local
        self.ctl = parallel.TcpComm(self.rank, self.world)
        self._lib = _lib
        self.ctx = _lib.context(self.device)

    def barrier(self):
        self._lib.check(self._lib.lib().hawk_sync(self.ctx), "hawk_sync")
        self.ctl.barrier()

    def max_float(self, x: float) -> float:
        return max(self.ctl.allgather_obj(float(x)))

    def sum_ints(self, vec):
        return self.ctl.allgather_i64(vec).sum(axis=0).tolist()


def timed_steps(R: Ranks, step, steps: int, warmup: int):
    """W warm-up steps, then exactly K steps bracketed by device sync + barrier on both sides; the MAX over ranks."""
    for _ in range(warmup):
        step()
    R.barrier()
    t0 = time.perf_counter()
    outs = [step() for _ in range(steps)]
    R.barrier()
    return R.max_float(time.perf_counter() - t0), outs


def pam_need_planes(pam) -> int:
    from crisprhawk_hip.pam import IUPAC_BITS
    need = 0
    for nib in pam.bits_list + [IUPAC_BITS[c] for c in pam.pamrc.upper()]:
        if nib != 15:
            need |= nib
    return need


def search_roofline(pam, positions, rows, count_ms, emit_list_ms, emit_ms, traffic_key=None):
    """The dominant kernel of the fused search (the longer of the count pass and the list-driven emit pass), priced with
    the bytes it has to move."""
    need = pam_need_planes(pam)
    count_bpp = 0.125 * (bin(need).count("1") + 1)
    if emit_list_ms == 0.0:  # HAWK_LIST_EMIT=0: the recompute-everything emit pass
        emit_name, e_ms, emit_bytes = "k_search_emit", emit_ms, READ_BYTES_PER_POS * positions + ROW_BYTES * rows
        count_bytes = count_bpp * positions
    else:
        emit_name, e_ms, emit_bytes = "k_emit_list", emit_list_ms, READ_BYTES_PER_POS * positions + (ROW_BYTES + LIST_BYTES) * rows
        count_bytes = count_bpp * positions + LIST_BYTES * rows
    cands = [(emit_name, e_ms, emit_bytes, READ_BYTES_PER_POS), ("k_search_count", count_ms, count_bytes, count_bpp)]
    cands.sort(key=lambda c: -c[1])
    (dom, dom_ms, dom_bytes, bpp), (oth, oth_ms, oth_bytes, _) = cands
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms else 0.0
    traffic = None
    if traffic_key and os.path.exists(PROFILE_TRAFFIC):
        tk = json.load(open(PROFILE_TRAFFIC)).get(traffic_key, {}).get(dom)
        traffic = tk and tk["fetch_bytes"] + tk["write_bytes"]  # PMC bytes per launch of the committed profile
    step_bytes = READ_BYTES_PER_POS * positions + ROW_BYTES * rows  # every plane bit read once + every row written once
    return {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "launch_ms": dom_ms, "algorithmic_bytes_per_launch": dom_bytes, "read_bytes_per_position": bpp,
            "row_bytes": ROW_BYTES, "list_bytes_per_row": LIST_BYTES,
            "other_kernel": {"kernel": oth, "launch_ms": oth_ms, "algorithmic_bytes_per_launch": oth_bytes,
                             "frac": (oth_bytes / (oth_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if oth_ms else None},
            "step_level": {"algorithmic_bytes": step_bytes, "what": "planes read once (0.625 B/position) + rows written once (74 B)"}}


def plan_carried(ds) -> int:
    """Carried-variant records of the set's expansion plan (one 32-byte record each)."""
    return int(getattr(ds.plan, "n_records", 0))


def vsearch_roofline(rows, records, positions, hits, L, scored, count_ms, emit_ms, traffic_key=None):
    """The per-word search of a plan view (hawk_vsearch.hip; what runs when a plan's cluster dictionary is not usable, or under
    HAWK_VIEW_SEARCH=words).  `achieved` = the bytes a launch has to move through HBM - every carried-variant record in (32 B)
    and, in the emit pass, every row out (74 B) - over its HIP-event duration; `survey_priced` is what SURVEY 8(d)'s per-unit
    figures (K2: 0.75 B per scanned haplotype position ...) would charge for the same work, which the kernels do not move:
    positions are never materialised.  Neither kernel is HBM-bound: per-wave counters
    (profiles/r03_pmc_vsearch_final.txt) show vector-instruction issue and s_waitcnt latency."""
    k2 = 0.75 * positions
    k3_read, k3_write = hits * L / 8.0 + rows * (L + 20) / 2.0, rows * 32.0
    k4 = rows * (2 * ((L + 1) // 2) + 8.0) if scored else 0.0
    emit_alg, count_alg = k2 + k3_read + k3_write + k4, k2 + k3_read
    emit_moved, count_moved = ROW_BYTES * rows + REC_BYTES * records, REC_BYTES * records
    cands = [("k_vsearch<1>", emit_ms, emit_alg, emit_moved), ("k_vsearch<0>", count_ms, count_alg, count_moved)]
    cands.sort(key=lambda c: -c[1])
    (dom, dom_ms, dom_alg, dom_moved), (oth, oth_ms, oth_alg, oth_moved) = cands
    gbps = lambda b, ms: b / (ms * 1e-3) / 1e9 if ms else 0.0
    traffic = None
    if traffic_key and os.path.exists(PROFILE_TRAFFIC_R3):
        tk = json.load(open(PROFILE_TRAFFIC_R3)).get(traffic_key, {}).get(dom)
        traffic = tk and tk["fetch_bytes"] + tk["write_bytes"]
    tot_ms = count_ms + emit_ms
    step_alg = k2 + k3_read + k3_write + k4  # every unit priced once for the step
    return {"bound": "hbm", "kernel": dom, "achieved": gbps(dom_moved, dom_ms), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbps(dom_moved, dom_ms) / HBM_PEAK_GBS, "traffic": traffic, "launch_ms": dom_ms, "algorithmic_bytes_per_launch": dom_moved,
            "pricing": "bytes the launch has to move given the plan representation: 32 B per carried-variant record in"
                       + (" + 74 B per guide row out" if dom.endswith("<1>") else ""),
            "survey_priced": {"bytes": dom_alg, "GBps": gbps(dom_alg, dom_ms),
                              "what": "SURVEY 8(d): K2 0.75 B/position + K3 (L/8 B/hit, (L+20)/2 + 32 B/row) + K4 (2*ceil(L/2) + 8 B/scored row)"},
            "records": records, "row_bytes": ROW_BYTES, "record_bytes": REC_BYTES,
            "other_kernel": {"kernel": oth, "launch_ms": oth_ms, "algorithmic_bytes_per_launch": oth_moved, "frac": gbps(oth_moved, oth_ms) / HBM_PEAK_GBS},
            "step_level": {"algorithmic_bytes": step_alg, "what": "SURVEY 8(d) K2 + K3 + K4, every unit priced once",
                           "moved_bytes": ROW_BYTES * rows + 2 * REC_BYTES * records}}


PACKED_ROW_BYTES = 64     # a packed guide row of the cluster search (include/hawk.h)
CS_INSTANCE_BYTES = 16    # what k_cs_emit_rows reads per cluster instance: row count, first template row, haplotype row, position
PROFILE_TRAFFIC_R4 = os.path.join(ROOT, "profiles", "r04_traffic.json")


def csearch_roofline(rows, ref_rows, positions, hits, L, scored, cl, templates_ms, count_ms, emit_ms, traffic_key=None):
    """The cluster search of a plan view (hawk_csearch.hip): templates (once per distinct cluster) -> counts per instance ->
    k_cs_emit_rows, the dominant kernel, which copies 64-byte template rows into the guide table (packed 64-byte rows, one
    linear write stream).  `achieved` = the bytes the launch has to move through HBM - 64 B per row out, 16 B per cluster
    instance in (count, first template row, haplotype row, position) + one 8-byte offset per 64 instances; the template rows
    (17 MB on C3) come out of L2 / the memory-side cache - over its HIP-event duration.  SURVEY 8(d)'s per-unit figures price
    work this step no longer does position by position; they are given as `survey_priced` only."""
    k2 = 0.75 * positions
    k3_read, k3_write = hits * L / 8.0 + rows * (L + 20) / 2.0, rows * 32.0
    k4 = rows * (2 * ((L + 1) // 2) + 8.0) if scored else 0.0
    step_alg = k2 + k3_read + k3_write + k4
    vrows = rows - ref_rows
    emit_alg = k3_write / max(rows, 1) * vrows + (k4 / max(rows, 1) * vrows) + vrows * (L + 20) / 2.0
    emit_moved = PACKED_ROW_BYTES * vrows + (CS_INSTANCE_BYTES + 8.0 / 64) * cl["instances"]
    gbps = lambda b, ms: b / (ms * 1e-3) / 1e9 if ms else 0.0
    traffic = None
    if traffic_key and os.path.exists(PROFILE_TRAFFIC_R4):
        tk = [v for k, v in json.load(open(PROFILE_TRAFFIC_R4)).get(traffic_key, {}).items() if k.startswith("k_cs_emit_rows")]
        traffic = tk[0]["fetch_bytes"] + tk[0]["write_bytes"] if tk else None
    count_moved = 16.0 * cl["instances"]  # cluster id + run start in, row count + first template row out (the clusters' 32-byte entries come out of L2)
    return {"bound": "hbm", "kernel": "k_cs_emit_rows", "achieved": gbps(emit_moved, emit_ms), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbps(emit_moved, emit_ms) / HBM_PEAK_GBS, "traffic": traffic, "launch_ms": emit_ms,
            "algorithmic_bytes_per_launch": emit_moved,
            "pricing": "bytes the launch has to move: 64 B per guide row written (packed row: include/hawk.h) + 16 B per cluster instance read "
                       "+ 8 B per 64 instances (their offset); SURVEY 8(d)'s per-row figure for a finished row is (L + 20) / 2 + 32 + 2 ceil(L / 2) + 8 "
                       "= %.1f B" % ((L + 20) / 2.0 + 32 + 2 * ((L + 1) // 2) + 8),
            "survey_priced": {"bytes": emit_alg, "GBps": gbps(emit_alg, emit_ms)},
            "row_bytes": PACKED_ROW_BYTES, "instance_bytes": CS_INSTANCE_BYTES, "template_row_bytes": 64,
            "other_kernels": {"k_cs_templates": {"launch_ms": templates_ms, "distinct_clusters": cl["distinct"]},
                              "k_cs_count": {"launch_ms": count_ms, "instances": cl["instances"], "moved_bytes": count_moved,
                                             "frac": gbps(count_moved, count_ms) / HBM_PEAK_GBS}},
            "step_level": {"algorithmic_bytes": step_alg, "what": "SURVEY 8(d) K2 + K3 + K4 over every haplotype position / hit / row, every unit priced once",
                           "moved_bytes": PACKED_ROW_BYTES * vrows + ROW_BYTES * ref_rows + (CS_INSTANCE_BYTES + 16.0) * cl["instances"]}}


def tab_ref_rows(reg, pam, args, mm, pt, device):
    """guide rows of the REF haplotype (row 0 of a plan view: written by the plane kernels, not by k_cs_emit)"""
    from crisprhawk_hip.workload import _ref_only_set
    ref_ds = _ref_only_set(reg.sequence, reg.startp, reg.stopp, len(pam), device)
    t = ref_ds.search(pam.bits, pam.bitsrc, len(pam), args.guidelen, args.right, mm, pt, download=False)
    n = t.n_rows
    t.close()
    ref_ds.close()
    return n


def pam_scan_kernel(ds, pam):
    """The K2 PAM-scan kernel on its own (what `pam_search` runs): north_star's >= 40 % of HBM peak target."""
    import ctypes as C
    from crisprhawk_hip import _lib
    ps_ms, ps_pos = C.c_float(0), C.c_uint64(0)
    _lib.check(_lib.lib().hawk_pam_scan_time(ds._h, C.c_uint64(pam.bits), C.c_uint64(pam.bitsrc), len(pam), 20, C.byref(ps_ms),
                                             C.byref(ps_pos)), "hawk_pam_scan_time")
    # bytes the scan really moves: one bit per position for each plane the PAM names (NGG/CCN: G and C only) plus the
    # forward and reverse hit bits; SURVEY.md §8(d) prices the nibble formulation at 0.75 B/position
    ps_bpp = 0.125 * bin(pam_need_planes(pam)).count("1") + 0.25
    ps_bytes = ps_bpp * ps_pos.value
    ach = ps_bytes / (ps_ms.value * 1e-3) / 1e9
    return {"kernel": "k_scan_raw", "launch_ms": ps_ms.value, "bytes_per_position": ps_bpp, "bytes_per_launch": ps_bytes, "achieved": ach,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "survey_algorithmic_bytes_per_position": 0.75,
            "positions_per_s": ps_pos.value / (ps_ms.value * 1e-3)}


def base_line(args, R: Ranks, value, elapsed, scaling, dtype, config):
    return {"metric": "candidate guides scored/sec", "value": value, "unit": "candidates/s", "n_gpus": R.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": dtype, "data": "synthetic", "config": config}


def region_tiles(args, R, reg, pam, n_tiles):
    """The C3 region cut into `n_tiles` stretches (tiling.TiledRegionSearch: exact seams, flanks, REF partners across seams) -
    the REGION shard of a multi-GPU run: rank r takes stretch r x all samples, so rows, distinct variant clusters, REF work and
    template rows all divide by the number of ranks (a sample block still carries nearly every distinct cluster of the panel)."""
    from crisprhawk_hip.tiling import DenseGenotypes, TiledRegionSearch, VariantPanel
    vs = reg.variants
    panel = VariantPanel(np.array([v.pos for v in vs], dtype=np.int64), [v.ref for v in vs], [v.alt for v in vs],
                         [f"{reg.contig}-{v.pos}-{v.ref}/{v.alt}" for v in vs], np.array([v.af for v in vs], dtype=np.float64),
                         list(reg.samples), DenseGenotypes(reg.gt_matrix if reg.gt_matrix is not None else np.stack([v.gt.reshape(-1) for v in vs])))
    span = reg.bed_stop - reg.bed_start
    trs = TiledRegionSearch(lambda lo, hi: reg.contig_seq[lo - 1:hi], reg.contig, reg.startp, reg.stopp, panel, pam, args.guidelen, args.right,
                            tile_nt=-(-span // n_tiles), device=R.device)
    if len(trs.tiles) != n_tiles:
        raise RuntimeError(f"region of {span} nt cut into {len(trs.tiles)} stretches, {n_tiles} asked for")
    return trs


def time_plan_steps(R, plan, pam, args, mm, pt, steps):
    """(ms per step with the dictionary built inside the step, ms per step with it resident, last table's counts) for one plan"""
    view = plan.view()
    use_dict = plan.cluster_stats()["usable"]

    def step(rebuild):
        if rebuild and use_dict:
            plan.rebuild_dictionary()
        t = view.search(pam.bits, pam.bitsrc, len(pam), args.guidelen, args.right, mm, pt, download=False)
        c = (t.n_candidates, t.n_rows, int(t.timing["v_path"]))
        t.close()
        return c
    out = []
    for rebuild in (True, False):
        step(rebuild)
        R._lib.check(R._lib.lib().hawk_sync(R.ctx), "hawk_sync")
        t0 = time.perf_counter()
        for _ in range(steps):
            c = step(rebuild)
        R._lib.check(R._lib.lib().hawk_sync(R.ctx), "hawk_sync")
        out.append((time.perf_counter() - t0) / steps * 1e3)
    return out[0], out[1], c


def shares_projection(args, R, reg, pam, mm, pt, whole_ms, whole_resident_ms):
    """What ONE rank's share of an N-way split of this workload takes on this one GPU, for both partitions, N = 2 / 4 / 8: the
    step's scaling before any exchange.  NOT a multi-GPU measurement - no N > 1 run has existed so far - but it shows which
    partition can scale: efficiency_before_exchange = whole-panel step / (N x share)."""
    from crisprhawk_hip.parallel import shard_range
    from crisprhawk_hip.workload import expand_on_device
    res = {"what": "one rank's share of an N-way split timed on this one GPU (step = dictionary + search; *_resident: search only); "
                   "efficiency_before_exchange = whole / (N x share); unmeasured on N GPUs",
           "whole_ms": whole_ms, "whole_resident_ms": whole_resident_ms, "region": {}, "samples": {}}
    k = max(5, min(20, args.steps))
    for n in (2, 4, 8):
        trs = region_tiles(args, R, reg, pam, n)
        ptile = trs.prepare_tile(n // 2, keep_plan=True)
        if ptile._first is not None:
            ptile._first.close()
            ptile._first = None
        ms, ms_res, c = time_plan_steps(R, ptile.plan, pam, args, mm, pt, k)
        res["region"][str(n)] = {"ms_per_step": ms, "ms_per_step_dictionary_resident": ms_res, "share_of_whole": ms / whole_ms,
                                 "efficiency_before_exchange": whole_ms / (n * ms), "efficiency_dictionary_resident": whole_resident_ms / (n * ms_res),
                                 "candidates": c[0], "rows": c[1], "v_path": c[2], "haplotype_rows": ptile.n_hap}
        trs.close()
        lo, hi = shard_range(len(reg.samples), n // 2, n)
        ds2, _i, _m, _k = expand_on_device(reg, len(pam), device=R.device, sample_range=(lo, hi), keep_plan=True)
        ms, ms_res, c = time_plan_steps(R, ds2.plan, pam, args, mm, pt, k)
        res["samples"][str(n)] = {"ms_per_step": ms, "ms_per_step_dictionary_resident": ms_res, "share_of_whole": ms / whole_ms,
                                  "efficiency_before_exchange": whole_ms / (n * ms), "efficiency_dictionary_resident": whole_resident_ms / (n * ms_res),
                                  "candidates": c[0], "rows": c[1], "v_path": c[2], "haplotype_rows": ds2.n_hap}
        ds2.plan.close()
        ds2.close()
    return res


# ---------------------------------------------------------------------------------------------------
# c3 / c1: one region, haplotype planes resident
# ---------------------------------------------------------------------------------------------------
def run_region(args, R: Ranks):
    from crisprhawk_hip import synth
    from crisprhawk_hip.pam import PAM
    from crisprhawk_hip.parallel import shard_range
    from crisprhawk_hip.workload import _ref_only_set, expand_on_device

    c1 = args.config == "c1"
    t0 = time.time()
    if c1:
        reg = synth.config_c1()
    else:
        reg = synth.make_region(1003, "chr22", args.region_len + 200_000, 100_000, 100_000 + args.region_len)
        # strong scaling: ONE panel, block-partitioned; --weak: every rank its own panel (round 1's measurement)
        synth.add_phased_variants(reg, 1003_1 + (7919 * R.rank if args.weak else 0), args.sites, args.samples, panel=args.panel)
    pam = PAM(args.pam, args.right, True)
    pam.encode(0)
    n_samples = len(reg.samples)
    # N > 1, one panel: `--shard region` (default) gives rank r the r-th stretch of the interval x all samples, `--shard samples`
    # the r-th block of the samples x the whole interval (north_star's wording; REF then on every rank)
    by_region = R.world > 1 and not (args.weak or c1) and args.shard == "region"
    slo, shi = (0, n_samples) if (args.weak or c1 or by_region) else shard_range(n_samples, R.rank, R.world)
    t1 = time.time()
    trs = ptile = None
    if by_region:
        trs = region_tiles(args, R, reg, pam, R.world)
        ptile = trs.prepare_tile(R.rank, keep_plan=True)
        ds, info, expand_ms, kept = ptile._first, None, 0.0, None
        ds.plan = ptile.plan
    else:
        ds, info, expand_ms, kept = expand_on_device(reg, len(pam), device=R.device, sample_range=(slo, shi), keep_plan=True)
    region_nt = int(ds.hap_len[0])
    log(f"workload: {ds.n_hap} haplotype rows x {region_nt} nt (samples {slo}..{shi} of {n_samples}), {len(reg.variants)} sites; "
        f"synthesised in {t1 - t0:.1f}s, expanded on the device in {time.time() - t1:.1f}s (kernels {expand_ms:.2f} ms, "
        f"{5 * ds.n_hap * ds.stride * 4 / 1e9:.2f} GB of planes)")
    score = (not args.right) and pam.cas_system in (3, 4)  # scoring.py:749-792: CFDon for SpCas9/xCas9 PAMs
    mm, pt = synth.cfd_tables() if score else (None, None)
    plan = getattr(ds, "plan", None)
    # The step SURVEY §8(d) defines: encode + search + reverse_guides + CFDon.  What is resident when the clock starts is the
    # INPUT of encode - REF's planes and every chromosome copy's variant records (the expansion plan) - not the haplotype
    # planes: a view of the plan (hawk_xplan_view) goes from there to the finished guide table without writing a plane.
    # `--planes` times the round-2 step instead (hawk_search over planes already materialised in HBM).
    fused = plan is not None and not args.planes
    target = plan.view() if fused else ds
    # The cluster dictionary of the plan (hawk_csearch.hip) is derived from the resident inputs and does not depend on the PAM - but
    # the product builds ONE per plan and searches it once (pipeline.search_files, TiledRegionSearch.run_tile), so a timed step
    # builds it too: `value` is the rate of dictionary + search; the dictionary-resident rate (several PAMs on one plan) is
    # measured right after and reported as `value_dictionary_resident`.
    with_dict = fused and plan.cluster_stats()["usable"]

    def step(keep=False, rebuild=with_dict):
        if rebuild:
            plan.rebuild_dictionary()
        tab = target.search(pam.bits, pam.bitsrc, len(pam), args.guidelen, args.right, mm, pt, download=False)
        if not keep:
            tab.close()
        return tab

    elapsed, tabs = timed_steps(R, step, args.steps, args.warmup)
    elapsed_resident = None
    if with_dict:
        for t in tabs:
            t.close()
        elapsed_resident, tabs = timed_steps(R, lambda keep=False: step(keep, rebuild=False), args.steps, 1)
    tab = tabs[-1]
    tm = [t.timing for t in tabs]
    avg = lambda k: float(np.mean([t[k] for t in tm]))
    cand, rows, positions = tab.n_candidates, tab.n_rows, tab.timing["scanned_positions"]
    # REF is scanned by every rank: count it once in the whole-job totals
    ref_c = ref_r = ref_p = 0
    if R.world > 1 and not args.weak and not c1 and not by_region:
        ref_ds = _ref_only_set(reg.sequence, reg.startp, reg.stopp, len(pam), R.device)
        rt = ref_ds.search(pam.bits, pam.bitsrc, len(pam), args.guidelen, args.right, mm, pt, download=False)
        ref_c, ref_r, ref_p = rt.n_candidates, rt.n_rows, rt.timing["scanned_positions"]
        rt.close()
        ref_ds.close()
    tot = R.sum_ints([cand, rows, positions, ds.n_hap])
    dup = 0 if by_region else R.world - 1
    cand_all, rows_all, pos_all = tot[0] - dup * ref_c, tot[1] - dup * ref_r, tot[2] - dup * ref_p
    if c1:  # replicas: every rank searched the same haplotype
        cand_all, rows_all, pos_all = cand * R.world, rows * R.world, positions * R.world

    workload = ("C1: 10 kb region, no VCF (BASELINE.json configs[1])" if c1 else
                "C3: 1 Mb region x 2504 phased samples (BASELINE.json configs[2])" if (args.samples, args.region_len) == (2504, 1_000_000)
                else "custom")
    out = None
    if R.rank == 0:
        out = base_line(args, R, cand_all * args.steps / elapsed, elapsed, "weak" if (args.weak or c1) else "strong", "u32",
                        {"workload": workload, "pam": args.pam, "guidelen": args.guidelen, "right": args.right, "region_nt": region_nt,
                         "haplotypes_rank0": ds.n_hap, "haplotype_rows_all_ranks": tot[3] - dup, "samples": n_samples,
                         "partition": "replicas" if c1 else ("own panel per rank" if args.weak else
                                                                 "stretches of the region x all samples (exact seams)" if by_region else
                                                                 "sample blocks of one panel, REF on every rank"),
                         "variant_sites": len(reg.variants), "scored": "CFDon (synthetic tables, seed 2001)" if score else "none",
                         "candidates_per_step": cand_all, "guide_rows_per_step": rows_all, "scanned_positions_per_step": pos_all})
        tkey = "c3" if workload.startswith("C3") and R.world == 1 else None
        out["config"]["step"] = ("encode + search + CFDon from the expansion plan (REF planes + variant records + their cluster dictionary resident; "
                                 "no haplotype plane written)" if fused else "hawk_search over haplotype planes resident in HBM")
        by_cluster = fused and int(tm[-1]["v_path"]) == 2
        if elapsed_resident is not None:
            out["value_dictionary_resident"] = cand_all * args.steps / elapsed_resident
            out["ms_per_step_dictionary_resident"] = elapsed_resident / args.steps * 1e3
            out["config"]["step"] = ("cluster dictionary of the plan + encode + search + CFDon (REF planes + variant records resident; no haplotype plane "
                                     "written); value_dictionary_resident: the same without the dictionary build (a second PAM on the same plan)")
        if by_cluster:
            cl = plan.cluster_stats()
            ref_rows = tab_ref_rows(reg, pam, args, mm, pt, R.device) // (R.world if by_region else 1)  # (a stretch holds its share of REF's rows)
            out["roofline"] = csearch_roofline(rows, ref_rows, positions, tab.n_hits, args.guidelen + len(pam), score, cl, avg("v_templates_ms"),
                                               avg("v_count_ms") - avg("v_templates_ms"), avg("v_emit_rows_ms"), tkey)
            out["cluster_dictionary"] = dict(cl, share=cl["instances"] / max(cl["distinct"], 1),
                                             what="built once per plan (hawk_csearch.hip): the rows' carried variants cut into clusters (alleles "
                                             "within 64 nt), identical clusters of different rows numbered once; `share` = instances per distinct "
                                             "cluster (below 3 the per-word search of hawk_vsearch.hip takes the plan); inside every timed step and every "
                                             "end_to_end pass")
        elif fused:
            out["roofline"] = vsearch_roofline(rows, int(plan_carried(ds)), positions, tab.n_hits, args.guidelen + len(pam), score,
                                               avg("v_count_ms"), avg("v_emit_ms"), tkey)
        else:
            out["roofline"] = search_roofline(pam, positions, rows, avg("count_ms"), avg("emit_list_ms"), avg("emit_ms"), tkey)
        sl = out["roofline"]["step_level"]
        sl["ms"] = avg("total_ms")
        # a view's step is priced by the bytes it moves: the survey's per-position figures describe work the step no longer does
        # (a fraction of peak over them would exceed 1) and are given as an equivalent rate only
        sl_bytes = sl["moved_bytes"] if fused else sl["algorithmic_bytes"]
        sl["frac"] = sl_bytes / (sl["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if sl["ms"] else None
        if fused:
            sl["survey_equivalent_GBps"] = sl["algorithmic_bytes"] / (sl["ms"] * 1e-3) / 1e9 if sl["ms"] else None
        out["kernels_ms"] = {"count": avg("count_ms"), "offsets": avg("offsets_ms"), "emit": avg("emit_ms"), "emit_list": avg("emit_list_ms"),
                             "vsearch_count": avg("v_count_ms"), "vsearch_emit": avg("v_emit_ms"), "view_templates": avg("v_templates_ms"),
                             "view_path": {0: "planes", 1: "per dirty word (hawk_vsearch.hip)", 2: "per distinct cluster (hawk_csearch.hip)"}[int(tm[-1]["v_path"])],
                             "device_total": avg("total_ms")}
        out["pam_scan_kernel"] = pam_scan_kernel(ds, pam)
        out["haplotype_expansion"] = {"kernels_ms": expand_ms, "rows": ds.n_hap,
                                      "where": "device (hawk_xplan_run); not part of the fused step, which writes no planes" if fused
                                      else "device (hawk_xplan_run), outside the timed steps"}
        if fused:  # the materialised alternative, for scale: planes written (hawk_xplan_run) and then searched (round 2's kernels)
            k = max(3, min(10, args.steps))
            R._lib.check(R._lib.lib().hawk_sync(R.ctx), "hawk_sync")
            t_a = time.perf_counter()
            tms = []
            for _ in range(k):
                tb = ds.search(pam.bits, pam.bitsrc, len(pam), args.guidelen, args.right, mm, pt, download=False)
                tms.append(tb.timing)
                tb.close()
            R._lib.check(R._lib.lib().hawk_sync(R.ctx), "hawk_sync")
            dt = (time.perf_counter() - t_a) / k
            out["search_on_resident_planes"] = {
                "ms_per_step": dt * 1e3, "candidates_per_s": cand / dt,
                "kernels_ms": {kk: float(np.mean([t[kk] for t in tms])) for kk in ("count_ms", "offsets_ms", "emit_list_ms", "emit_ms", "total_ms")},
                "plus_expansion_ms": expand_ms, "candidates_per_s_with_expansion": cand / (dt + expand_ms * 1e-3),
                "what": "round 2's step: hawk_search over planes hawk_xplan_run has already written; with the expansion kernels added it is "
                        "the same work as the fused step"}
            if by_cluster:  # what the step costs when a plan's clusters are not shared (dictionary unusable): the per-word search of the view
                os.environ["HAWK_VIEW_SEARCH"] = "words"
                try:
                    R._lib.check(R._lib.lib().hawk_sync(R.ctx), "hawk_sync")
                    step(rebuild=False)
                    t_a = time.perf_counter()
                    for _ in range(k):
                        step(rebuild=False)
                    R._lib.check(R._lib.lib().hawk_sync(R.ctx), "hawk_sync")
                    dt = (time.perf_counter() - t_a) / k
                finally:
                    del os.environ["HAWK_VIEW_SEARCH"]
                out["search_per_dirty_word"] = {"ms_per_step": dt * 1e3, "candidates_per_s": cand / dt,
                                                "what": "the same view searched per dirty word of every row (hawk_vsearch.hip): the path a plan takes when "
                                                        "fewer than 3 instances share a distinct cluster, or a variant chain exceeds 4096 records"}
    for t in tabs:
        t.close()

    # ---- the one exchange of the job (N > 1): guide tables to rank 0 over RCCL ----------------------
    if R.world > 1 and not args.no_gather:
        # RCCL with peers cannot be rehearsed on the one-GPU boxes this was built on: if the exchange hangs rather than
        # fails, the measured line must still come out - a watchdog prints it (gather: timeout) and ends the rank
        import threading

        def _bail():
            if R.rank == 0:
                out["gather"] = {"error": f"no completion within {args.gather_timeout} s (watchdog)"}
                print(json.dumps(out), flush=True)
            os._exit(3)  # the measured line is out, but a hung exchange is a failed run: every rank ends non-zero
        wd = threading.Timer(args.gather_timeout, _bail)
        wd.daemon = True
        wd.start()
        g = gather_region(R, trs, cfd=(mm, pt) if score else None) if by_region else gather_once(R, ds, step, tot[3])
        wd.cancel()
        if R.rank == 0:
            out["gather"] = g
    if R.rank == 0 and R.world == 1:
        if not args.no_end_to_end:
            out["end_to_end"] = end_to_end(args, R, reg, pam, mm, pt, info, kept, c1)
            out["value_end_to_end"] = out["end_to_end"].get("candidates_per_s")  # records + genotypes in host memory -> report groups on the host
            if not c1 and not args.no_files:
                out["end_to_end"]["files_to_tsv"] = files_to_tsv(args, reg, mm, pt, R.device)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(reg, pam, args, mm, pt)
            if not c1 and os.path.exists(REFERENCE_TIMING):
                # NOT measured here: the reference's own Python path, timed in the build container (it cannot travel to the
                # GPU box) on C3 restricted to a few haplotypes - the committed number, quoted for scale
                ext = json.load(open(REFERENCE_TIMING)).get("extrapolation_to_c3", {})
                out["reference_python"] = {"candidates_per_s": ext.get("c3_candidates_per_s"), "c3_seconds_extrapolated": ext.get("c3_seconds_5009_haplotypes"),
                                           "measured": "build container, 8 vCPU, single thread (profiles/r02_reference_python_timing.json); not on this host"}
            if not c1 and not args.no_cpu_all_cores:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args)
    if R.rank == 0 and R.world == 1 and not c1 and not args.no_shares and fused and out.get("ms_per_step_dictionary_resident"):
        out["shares_on_one_gpu"] = shares_projection(args, R, reg, pam, mm, pt, out["ms_per_step"], out["ms_per_step_dictionary_resident"])
    if trs is not None:
        trs.close()
    elif plan is not None:
        plan.close()
    if trs is None:
        ds.close()
    return out


def gather_region(R: Ranks, trs, cfd):
    """The one exchange of a region-sharded job: per rank search -> collapse -> export of its stretch (device), then the
    collapsed report groups - one representative row per group + the members' haplotype rows - travel to rank 0 over RCCL
    (hawk_comm_gatherv) and are merged at the seams (tiling.gather_tile_groups).  Measured once after the timed loop."""
    from crisprhawk_hip import parallel
    from crisprhawk_hip.tiling import _groups_of_tile, gather_tile_groups
    if R.backend != "rccl":
        comm = R.ctl
    else:
        comm = None
    res = {}
    try:
        t0 = time.perf_counter()
        g, st = trs.run_tile(R.rank, cfd)
        n_all = R.ctl.allgather_i64([st["n_hap"]])[:, 0]
        part = _groups_of_tile(g, int(n_all[:R.rank].sum()))
        t1 = time.perf_counter()
        if comm is None:
            comm = parallel.RcclComm(R.ctl, R.device)
        comm.barrier()
        t2 = time.perf_counter()
        acc, nbytes = gather_tile_groups(comm, part, trs.tiles[R.rank].own_lo, trs.L, trs.guidelen, len(trs.pam))
        comm.barrier()
        t3 = time.perf_counter()
        rows = R.sum_ints([st["rows"]])[0]
        if R.rank == 0:
            res["collapsed_region"] = {"search_collapse_export_s": t1 - t0, "exchange_and_seam_merge_s": t3 - t2, "bytes_to_rank0": int(nbytes),
                                       "groups": int(len(acc["sizes"])), "members": int(len(acc["members"])), "guide_rows_behind_them": int(rows),
                                       "backend": R.backend}
        if comm is not R.ctl:
            comm.close()
    except Exception as e:  # the value line must survive a failed exchange
        res["error"] = f"{type(e).__name__}: {e}"
        log(f"gather failed: {res['error']}")
    return res


def gather_once(R: Ranks, ds, step, n_hap_total):
    """Measured once after the timed loop and reported next to `value`, not inside it: the per-step path has no
    data-path collective.  Full table (74 B/row) device to device, then the collapsed form (one
    representative row per report group + 4 B per member)."""
    from crisprhawk_hip import parallel
    if R.backend != "rccl":
        return {"skipped": f"backend {R.backend}: the RCCL exchange needs one GPU per rank"}
    res = {}
    # every rank says over the TCP control plane whether it can enter the exchange at all (librccl loads, its table exists)
    # BEFORE anybody calls into RCCL: one rank failing on its own must not strand the others in a collective
    ok, why = 1, ""
    try:
        tab = step(keep=True)
        R._lib.check(R._lib.lib().hawk_comm_unique_id(np.zeros(128, dtype=np.uint8).ctypes.data_as(__import__("ctypes").c_void_p)), "hawk_comm_unique_id")
    except Exception as e:
        ok, why, tab = 0, f"{type(e).__name__}: {e}", None
    oks = R.ctl.allgather_i64([ok])[:, 0]
    if not oks.all():
        if tab is not None:
            tab.close()
        return {"error": f"rank(s) {np.flatnonzero(oks == 0).tolist()} cannot enter the exchange" + (f" ({why})" if why else "")}
    try:
        comm = parallel.RcclComm(R.ctl, R.device)
        n_all = R.ctl.allgather_i64([ds.n_hap])[:, 0]
        hap_off = int(n_all[:R.rank].sum() - R.rank)  # rows of the earlier ranks, their REF rows not counted
        comm.barrier()
        t0 = time.perf_counter()
        merged, ms = comm.gather_table(tab, hap_off, 0)
        comm.barrier()
        wall = time.perf_counter() - t0
        rows = R.sum_ints([tab.n_rows])[0]
        if R.rank == 0:
            res["full_table"] = {"ms_on_stream": ms, "wall_ms": wall * 1e3, "rows": rows, "bytes_to_rank0": rows * ROW_BYTES,
                                 "GBps": rows * ROW_BYTES / wall / 1e9, "merged_rows": merged.n_rows}
            merged.close()
        # collapsed: per-rank device collapse + group export, then representatives + member lists
        t0 = time.perf_counter()
        tab.collapse(download_perm=False)
        g = tab.export_groups()
        tab.close()
        t1 = time.perf_counter()
        rep = {k: getattr(g, k) for k in ("pos", "strand", "start", "stop", "flags", "cfdon")}
        rep["win"] = np.ascontiguousarray(g.win.T)
        mem = g.member_hap.astype(np.int64)
        mem = np.where(mem == 0, 0, mem + hap_off).astype(np.uint32)
        first = g.member_hap[g.member_off[:-1]] if g.n_groups else np.zeros(0, np.uint32)
        origin = np.asarray(ds.is_ref, dtype=np.uint8)[first]
        parts = {k: comm.gatherv_bytes(v, 0) for k, v in rep.items()}
        o_p, s_p, m_p = comm.gatherv_bytes(origin, 0), comm.gatherv_bytes(np.diff(g.member_off), 0), comm.gatherv_bytes(mem, 0)
        t2 = time.perf_counter()
        if R.rank == 0:
            repc = {k: np.concatenate(v) for k, v in parts.items()}
            mg, moff, members = parallel.merge_groups(repc, np.concatenate(o_p), np.concatenate(s_p), np.concatenate(m_p),
                                                      g.guidelen, g.pamlen)
            t3 = time.perf_counter()
            nbytes = sum(sum(a.nbytes for a in v) for v in parts.values()) + sum(a.nbytes for a in m_p) + sum(a.nbytes for a in s_p)
            res["collapsed"] = {"collapse_export_s": t1 - t0, "exchange_s": t2 - t1, "merge_on_rank0_s": t3 - t2, "bytes_to_rank0": nbytes,
                                "groups": len(moff) - 1, "members": len(members)}
        comm.barrier()
        comm.close()
    except Exception as e:  # the value line must survive a failed exchange
        res["error"] = f"{type(e).__name__}: {e}"
        log(f"gather failed: {res['error']}")
    return res


def end_to_end(args, R: Ranks, reg, pam, mm, pt, info, kept, c1):
    """in-memory records -> device expansion -> search + CFDon -> collapse -> D2H of the report groups,
    wall clock (rank 0, N = 1).  With --report the guide report (f2) is assembled from the groups as well.
    The pipeline runs twice: the first pass pays for whatever HBM the library's allocator does not hold yet (hipMalloc of
    the 2 GB of guide columns alone is ~50 ms) and is reported as `first_run_wall_s`; `wall_s` is the second pass."""
    first = _end_to_end_once(args, R, reg, pam, mm, pt, c1, False)
    res = _end_to_end_once(args, R, reg, pam, mm, pt, c1, True)
    res["first_run_wall_s"] = first["wall_s"]
    res["first_run_stages_s"] = first["stages_s"]
    return res


def _end_to_end_once(args, R: Ranks, reg, pam, mm, pt, c1, extras):
    from crisprhawk_hip.workload import expand_on_device
    t0 = time.perf_counter()
    ds, info2, ems, kept2 = expand_on_device(reg, len(pam), device=R.device, keep_plan=True)
    t1 = time.perf_counter()
    plan2 = getattr(ds, "plan", None)
    target = plan2.view() if (plan2 is not None and not args.planes) else ds
    tab = target.search(pam.bits, pam.bitsrc, len(pam), args.guidelen, args.right, mm, pt, download=False)
    R._lib.check(R._lib.lib().hawk_sync(R.ctx), "hawk_sync")
    t2 = time.perf_counter()
    tab.collapse(download_perm=False)
    g = tab.export_groups()
    t3 = time.perf_counter()
    d2h = sum(getattr(g, k).nbytes for k in ("rep_row", "pos", "strand", "start", "stop", "flags", "cfdon", "win", "member_hap")) + \
        g.member_off.nbytes + g.gc_num.nbytes + g.gc_den.nbytes
    res = {"wall_s": t3 - t0, "candidates_per_s": tab.n_candidates / (t3 - t0),
           "stages_s": {"plan (variant table, genotype matrix up, lists / records / segments / scan bounds on the device, one expansion for "
                        "the content hashes, which rows collapse)": t1 - t0, "search + CFDon (from the plan)": t2 - t1,
                        "collapse + D2H of the report groups": t3 - t2},
           "kernels_ms": {"expand": ems, "search": tab.timing["total_ms"], "collapse": tab.collapse_ms, "export": g.export_ms},
           "rows": tab.n_rows, "groups": g.n_groups, "d2h_bytes": int(d2h),
           "what": "in-memory variant records + genotype matrix -> hawk_gt_lists -> hawk_xplan_create_gt -> hawk_xplan_view + hawk_search -> "
                   "hawk_table_collapse -> hawk_table_collapse_export; FASTA/VCF text ingest (f3) is timed by --vcf"}
    if extras and args.report and not c1:
        from crisprhawk_hip import reports
        from crisprhawk_hip.workload import hap_labels
        t4 = time.perf_counter()
        df = reports.report_from_groups(g, hap_labels(reg.contig, reg.variants, ds, info2, kept2), pam, reg.contig,
                                        f"{reg.contig}:{reg.bed_start}-{reg.bed_stop}", is_ref_hap=np.asarray(ds.is_ref, dtype=bool))
        t5 = time.perf_counter()
        txt = reports.to_tsv(df)
        t6 = time.perf_counter()
        res["report"] = {"assemble_s": t5 - t4, "report_rows": int(len(df)), "tsv_text_s": t6 - t5, "tsv_bytes": len(txt),
                         "what": "reports.report_from_groups (columnar; samples / haplotype ids joined by the library's host helpers) "
                                 "+ reports.to_tsv"}
        res["wall_with_report_s"] = t6 - t0
        del txt
    if extras and args.vcf and not c1:
        res["vcf_ingest"] = time_vcf_ingest(reg, ds, len(pam), R.device)
    tab.close()
    if plan2 is not None:
        plan2.close()
    ds.close()
    return res


def files_to_tsv(args, reg, mm, pt, device):
    """What a user of `crisprhawk search` waits for: FASTA + BED + phased VCF FILES -> the guide report TSV on disk
    (pipeline.search_files: readers -> device genotype parse + plan -> dictionary + search + CFDon -> collapse -> report columns ->
    the library's TSV writer).  The input files are written first, untimed; the pipeline runs twice and the second run is reported
    (`first_run_wall_s`: allocator and page cache cold)."""
    import shutil
    import tempfile
    from crisprhawk_hip import synth
    from crisprhawk_hip.pipeline import search_files
    d = tempfile.mkdtemp(prefix="hawk_files_", dir=os.environ.get("HAWK_SCRATCH", tempfile.gettempdir()))
    try:
        t0 = time.perf_counter()
        fa, bed, vcf = synth.write_region_files(reg, d, "c3")
        setup = time.perf_counter() - t0
        runs = []
        for r in range(2):
            tm = {}
            t0 = time.perf_counter()
            paths = search_files(fa, bed, [vcf], args.pam, args.guidelen, args.right, os.path.join(d, f"out{r}"),
                                 cfd_tables=(mm, pt) if mm is not None else None, device=device, timings=tm)
            runs.append((time.perf_counter() - t0, tm, os.path.getsize(list(paths.values())[0])))
        wall, tm, nbytes = runs[1]
        return {"wall_s": wall, "stages_s": tm, "first_run_wall_s": runs[0][0], "vcf_bytes": os.path.getsize(vcf), "tsv_bytes": nbytes,
                "write_inputs_s_untimed": setup,
                "what": "FASTA + BED + VCF text on disk -> crisprhawk_guides__*.tsv on disk (pipeline.search_files), wall clock of the second run"}
    except Exception as e:  # the value line must survive a full scratch disk
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def time_vcf_ingest(reg, ds_ref, pamlen, device):
    """f3 at workload scale: the VCF text of the region's records (what readers.VCF.fetch_block returns) ->
    hawk_gt_parse -> hawk_gt_lists -> expansion, compared with the planes of the in-memory path."""
    from crisprhawk_hip.readers import VcfBlock
    from crisprhawk_hip.workload import expand_from_vcf
    t0 = time.perf_counter()
    ns = len(reg.samples)
    parts, fixed, line_len = [], [], []
    sep = np.array([ord("|")], np.uint8)
    for v in reg.variants:
        head = f"{reg.contig}\t{v.pos}\t.\t{v.ref}\t{v.alt}\t.\tPASS\tAF={v.af:.6g}\tGT\t".encode()
        g = np.empty((ns, 4), np.uint8)
        g[:, 0] = v.gt[:, 0] + 48; g[:, 1] = sep; g[:, 2] = v.gt[:, 1] + 48; g[:, 3] = 9
        body = g.reshape(-1).tobytes()[:-1] + b"\n"
        parts.append(head); parts.append(body)
        fixed.append(head.decode().rstrip("\t").split("\t"))
        line_len.append((len(head), len(head) + len(body)))
    text = np.frombuffer(b"".join(parts), dtype=np.uint8)
    ll = np.array(line_len, dtype=np.uint64)
    line_off = np.zeros(len(ll) + 1, dtype=np.uint64)
    line_off[1:] = np.cumsum(ll[:, 1])
    gt_off = line_off[:-1] + ll[:, 0]
    blk = VcfBlock(text, line_off, gt_off, fixed)
    t1 = time.perf_counter()
    ds2, info2, ms, kept2, vt = expand_from_vcf(reg.sequence, reg.startp, reg.stopp, blk, reg.samples, pamlen, True, device)
    t2 = time.perf_counter()
    same = bool(ds2.n_hap == ds_ref.n_hap and np.array_equal(ds2.planes(), ds_ref.planes()))
    ds2.close()
    log(f"vcf ingest: {len(text) / 1e6:.0f} MB of records -> {ds2.n_hap} haplotype rows in {t2 - t1:.2f}s "
        f"(parse {ms['parse']:.2f} ms, lists {ms['lists']:.2f} ms, expand {ms['expand']:.2f} ms); planes identical: {same}")
    return {"text_bytes": int(len(text)), "records": len(blk), "samples": ns, "make_text_s": t1 - t0, "ingest_wall_s": t2 - t1,
            "kernels_ms": ms, "parse_GBps": len(text) / (ms["parse"] * 1e-3) / 1e9 if ms["parse"] else None,
            "planes_identical_to_in_memory_path": same}


# ---------------------------------------------------------------------------------------------------
# cpu baselines (the oracle = a C port of the reference's algorithm; bounded samples of the same workload)
# ---------------------------------------------------------------------------------------------------
def _oracle_batch(reg, pamlen, b, per):
    from crisprhawk_hip.workload import build_phased_haplotypes
    from oracle import oracle as ora
    n = max(1, len(reg.samples))
    lo = (b * per) % n
    sub, _ = build_phased_haplotypes(reg, pamlen, sample_slice=slice(lo, lo + per))
    hs = ora.HapSet([bytes(h.seq).decode("ascii") for h in sub], [h.seg.full() for h in sub], [h.is_ref for h in sub], [h.scan for h in sub])
    blob_args = hs.packed()
    hs.packed = lambda blob_args=blob_args: blob_args
    return hs, len(sub)


def cpu_baseline(reg, pam, args, mm, pt):
    """The oracle on the first --cpu-haps haplotypes of the same workload, one host thread: search + reverse_guides +
    CFDon.  The sample is walked in batches of 80 samples (REF + <= 160 haplotypes each, so host memory stays bounded);
    building the batch's strings and marshalling them are not timed (the GPU side starts from resident planes too)."""
    from oracle import oracle as ora
    cand, dt, nh, per, b = 0, 0.0, 0, 80, 0
    reps = 0
    while True:
        hs, n_sub = _oracle_batch(reg, len(pam), b, per)
        t0 = time.perf_counter()
        res = ora.search(hs, pam.pam, args.guidelen, args.right)
        ora.reverse_and_cfdon(res, hs.is_ref, args.guidelen, len(pam), mm, pt, decode=False)
        dt += time.perf_counter() - t0
        cand += res.n_candidates
        nh += n_sub
        b += 1
        reps += 1
        if reg.variants:
            if nh >= max(2, args.cpu_haps) or b * per >= len(reg.samples):
                break
        elif dt > 2.0 or reps >= 2000:  # variant-free region: repeat the one haplotype for a measurable time
            break
    return {"value": cand / dt, "unit": "candidates/s", "cores": 1, "kind": "port",
            "sample": f"{nh} haplotype scans of the same workload in {b} batches of REF + <= {2 * per} haplotypes "
                      f"({cand} candidates, {dt:.1f} s, 1 thread of {usable_cores()} usable)"}


def cpu_baseline_c4(seq, panel, n_block, pam, args, mm, pt):
    """The oracle on a bounded sample of the C4 workload: a 1 Mb window of the contig behind the N block, the panel's own
    variants and genotypes there, REF + the first --cpu-haps chromosome copies; one host thread."""
    from crisprhawk_hip import synth
    lo = n_block + 500_000
    hi = min(lo + 1_000_000, len(seq) - 1_000)
    startp = lo - 100
    sub = bytes(seq[startp - 1:hi + 100]).decode()
    v_lo, v_hi = int(np.searchsorted(panel.pos, startp + 1)), int(np.searchsorted(panel.pos, hi + 100 - 12))
    n_s = min(len(panel.samples), max(1, args.cpu_haps // 2))
    G = panel.genotypes.dense(v_lo, v_hi, 0, 2 * n_s)
    reg = synth.SynthRegion("chr22w", sub, 101, 101 + (hi - lo))
    reg.samples = list(panel.samples[:n_s])
    shift = startp - 1  # window position 1 = contig position startp
    reg.variants = [synth.VariantSite(int(panel.pos[v_lo + k]) - shift, panel.ref[v_lo + k], panel.alt[v_lo + k], float(panel.af[v_lo + k]),
                                      G[k].reshape(n_s, 2).astype(np.uint8)) for k in range(v_hi - v_lo)]
    out = cpu_baseline(reg, pam, args, mm, pt)
    out["sample"] = f"a {hi - lo}-nt window of the contig behind the N block, the panel's calls of its first {n_s} samples: " + out["sample"]
    return out


def _cpu_worker(wid, n_workers, batches, argv, barrier, q):
    """One oracle worker (own process, spawn context - the parent has initialised HIP): prepare its batches untimed,
    meet the others at the barrier, then search + CFDon back to back; report start / end stamps and candidates."""
    sys.argv = ["bench.py"] + argv
    args = build_parser().parse_args(argv)
    from crisprhawk_hip import synth
    from crisprhawk_hip.pam import PAM
    from oracle import oracle as ora
    reg = synth.make_region(1003, "chr22", args.region_len + 200_000, 100_000, 100_000 + args.region_len)
    synth.add_phased_variants(reg, 1003_1, args.sites, args.samples)
    pam = PAM(args.pam, args.right, True)
    pam.encode(0)
    score = (not args.right) and pam.cas_system in (3, 4)
    mm, pt = synth.cfd_tables() if score else (None, None)
    sets = [_oracle_batch(reg, len(pam), wid * batches + k, 40)[0] for k in range(batches)]
    barrier.wait()
    t0 = time.time()
    cand = 0
    for hs in sets:
        res = ora.search(hs, pam.pam, args.guidelen, args.right)
        ora.reverse_and_cfdon(res, hs.is_ref, args.guidelen, len(pam), mm, pt, decode=False)
        cand += res.n_candidates
    q.put((wid, t0, time.time(), cand, sum(len(h.seqs) for h in sets)))


def cpu_baseline_all_cores(args):
    """the same oracle on every usable host core at once (one process per core, each with its own
    bounded sample of the workload); rate = all candidates / (last end - first start)."""
    import multiprocessing as mp
    n = min(usable_cores(), int(os.environ.get("HAWK_CPU_WORKERS", "64")))
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(n), ctx.Queue()
    argv = [a for a in sys.argv[1:]]
    procs = [ctx.Process(target=_cpu_worker, args=(w, n, 8, argv, barrier, q)) for w in range(n)]
    t0 = time.time()
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(n):
            res.append(q.get(timeout=600))
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.terminate()
    start, end = min(r[1] for r in res), max(r[2] for r in res)
    cand, nh = sum(r[3] for r in res), sum(r[4] for r in res)
    return {"value": cand / (end - start), "unit": "candidates/s", "cores": n, "kind": "port",
            "sample": f"{n} processes x 8 batches of REF + <= 80 haplotypes ({nh} haplotype scans, {cand} candidates) in "
                      f"{end - start:.1f} s between the common start and the last finish; set-up {start - t0:.0f} s not timed"}


# ---------------------------------------------------------------------------------------------------
# c4: whole contig, region-tiled
# ---------------------------------------------------------------------------------------------------
def run_c4(args, R: Ranks):
    from crisprhawk_hip import synth
    from crisprhawk_hip.pam import PAM
    from crisprhawk_hip.parallel import shard_range
    from crisprhawk_hip.tiling import TiledRegionSearch

    pam = PAM(args.pam, args.right, True)
    pam.encode(0)
    t0 = time.time()
    n_block = int(args.contig_len * args.n_block_frac)
    seq, panel = synth.contig_panel(1004, "chr22", args.contig_len, n_block, args.samples, sites_per_mb=args.sites / (args.region_len / 1e6))
    slo, shi = (0, args.samples) if args.weak else shard_range(args.samples, R.rank, R.world)
    startp, stopp = 1, args.contig_len  # BED 100 .. contig_len - 100: the whole contig, padding inside it
    trs = TiledRegionSearch(lambda lo, hi: seq[lo - 1:hi], "chr22", startp, stopp, panel, pam, args.guidelen, args.right,
                            tile_nt=args.tile_nt, device=R.device, sample_range=(slo, shi))
    log(f"c4 workload: contig {args.contig_len} nt ({n_block} nt leading N), {len(panel.pos)} sites, samples {slo}..{shi} of {args.samples}, "
        f"{len(trs.tiles)} tiles of {args.tile_nt} nt; synthesised in {time.time() - t0:.1f}s")
    t1 = time.time()
    hbm = 0
    for t in range(len(trs.tiles)):
        pt_ = trs.prepare_tile(t, keep_plan=True)
        if pt_._first is not None and pt_.plan is not None:  # the preparation run's planes are not needed again
            pt_._first.close()
            pt_._first = None
        log(f"  tile {t}: {pt_.n_hap} rows prepared ({time.time() - t1:.0f}s)")
    log(f"prepared {len(trs.tiles)} tiles (plans resident in HBM) in {time.time() - t1:.1f}s")
    score = (not args.right) and pam.cas_system in (3, 4)
    cfd = synth.cfd_tables() if score else None

    def step():
        stats = []
        for t in range(len(trs.tiles)):
            _g, st = trs.run_tile(t, cfd, (0, 0), True, export=False)
            stats.append(st)
        return stats

    elapsed, outs = timed_steps(R, step, args.steps, args.warmup)
    st = outs[-1]
    cand = sum(s["candidates"] for s in st)
    rows = sum(s["rows"] for s in st)
    positions = sum(s["scanned_positions"] for s in st)
    groups = sum(s["groups"] for s in st)
    # REF rows are scanned by every rank: count them once
    ref_c = ref_p = 0
    if R.world > 1 and not args.weak:
        sub = TiledRegionSearch(lambda lo, hi: seq[lo - 1:hi], "chr22", startp, stopp, None, pam, args.guidelen, args.right,
                                tile_nt=args.tile_nt, device=R.device)
        for t in range(len(sub.tiles)):
            _g, s_ = sub.run_tile(t, cfd, (0, 0), True, export=False)
            ref_c += s_["candidates"]; ref_p += s_["scanned_positions"]
        sub.close()
    tot = R.sum_ints([cand, rows, positions, groups])
    dup = R.world - 1
    cand_all, pos_all = tot[0] - dup * ref_c, tot[2] - dup * ref_p
    out = None
    if R.rank == 0:
        search_ms = sum(s["search_ms"] for s in st)
        collapse_ms = sum(s["collapse_ms"] for s in st)
        out = base_line(args, R, cand_all * args.steps / elapsed, elapsed, "weak" if args.weak else "strong", "u32",
                        {"workload": f"C4: whole contig ({args.contig_len} nt, {n_block} nt leading N block) x {args.samples} phased samples, "
                                     f"region-tiled (BASELINE.json configs[3])", "pam": args.pam, "guidelen": args.guidelen,
                         "right": args.right, "tiles": len(trs.tiles), "tile_nt": args.tile_nt, "variant_sites": len(panel.pos),
                         "partition": "sample blocks of one panel, REF on every rank" if not args.weak else "own columns per rank",
                         "step": "per tile: search (+CFDon, NA on N) straight from the tile's expansion plan (hawk_xplan_view: no plane "
                                 "written; once per distinct variant cluster) -> hawk_table_collapse (on the search's template rows); tiles "
                                 "without variants search their REF planes",
                         "candidates_per_step": cand_all, "guide_rows_rank0": rows, "report_groups_rank0": groups,
                         "scanned_positions_per_step": pos_all})
        records = sum(s.get("records", 0) for s in st)
        instances = sum(s.get("instances", 0) for s in st)
        by_cluster = any(s.get("v_path", 0) == 2 for s in st)
        # the cluster path times k_cs_emit_rows alone; the per-word path's emit side is k_vsearch<1>
        emit_ms = sum(s.get("v_emit_rows_ms", 0.0) if s.get("v_path", 0) == 2 else s.get("v_emit_ms", 0.0) for s in st)
        vrows = sum(s["rows"] for s in st if s.get("v_path", 0) == 2)           # rows of the tiles searched per cluster (REF's few included)
        # what the tiles' searches move: per cluster path a packed 64-byte row out + 16 B per instance in (k_cs_emit_rows) and
        # 16 B per instance in the count pass; per dirty word a 74-byte row in columns + the records twice
        if by_cluster:
            emit_bytes = PACKED_ROW_BYTES * vrows + (CS_INSTANCE_BYTES + 8.0 / 64) * instances
            step_bytes = PACKED_ROW_BYTES * vrows + ROW_BYTES * (rows - vrows) + (CS_INSTANCE_BYTES + 16.0) * instances
        else:
            emit_bytes = ROW_BYTES * rows + REC_BYTES * records
            step_bytes = ROW_BYTES * rows + 2 * REC_BYTES * records
        out["roofline"] = {"bound": "hbm", "kernel": ("k_cs_emit_rows" if by_cluster else "k_vsearch<1>") + " (summed over tiles)",
                           "achieved": emit_bytes / (emit_ms * 1e-3) / 1e9 if emit_ms else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": emit_bytes / (emit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if emit_ms else 0.0, "traffic": None, "launch_ms": emit_ms,
                           "algorithmic_bytes_per_launch": emit_bytes, "records": records, "cluster_instances": instances,
                           "pricing": "bytes the launches have to move: " +
                                      ("64 B per guide row written (packed row) + 16 B per cluster instance read + 8 B per 64 instances"
                                       if by_cluster else "74 B per guide row written + 32 B per carried-variant record read"),
                           "step_level": {"algorithmic_bytes": step_bytes, "ms": search_ms,
                                          "frac": step_bytes / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if search_ms else None,
                                          "what": "all search kernels of all tiles (the variant-free tiles' plane searches included): rows written "
                                                  "once + what the passes read per cluster instance / record"},
                           "survey_priced": {"bytes": 0.75 * positions + ROW_BYTES * rows,
                                             "effective_GBps": (0.75 * positions + ROW_BYTES * rows) / (search_ms * 1e-3) / 1e9 if search_ms else None}}
        if R.world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_c4(seq, panel, n_block, pam, args, *(cfd if cfd else (None, None)))
        out["kernels_ms"] = {"search_all_tiles": search_ms, "collapse_all_tiles": collapse_ms,
                             "wall_per_step": elapsed / args.steps * 1e3}
        out["per_tile"] = [{k: s[k] for k in ("tile", "n_hap", "rows", "groups", "search_ms", "collapse_ms")} for s in st]
    trs.close()
    return out


# ---------------------------------------------------------------------------------------------------
# c5: off-target enumeration against a packed genome
# ---------------------------------------------------------------------------------------------------
def run_c5(args, R: Ranks):
    from crisprhawk_hip.genome import GenomeIndex
    from crisprhawk_hip.pam import PAM
    pam_s, guidelen, right = ("TTTV", 23, True) if args.pam == "NGG" and not args.right else (args.pam, args.guidelen, args.right)
    pam = PAM(pam_s, right, True)
    pam.encode(0)
    t0 = time.time()
    rng = np.random.default_rng(1006)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    per = args.genome_nt // args.genome_contigs
    contigs = {}
    for i in range(args.genome_contigs):
        contigs[f"chr{i + 1}"] = acgt[rng.integers(0, 4, size=per, dtype=np.uint8)]
    names = list(contigs)
    guides = []
    grng = np.random.default_rng(1005)
    while len(guides) < args.guides:  # spacers cut out of the genome: every guide has its on-target
        c = contigs[names[int(grng.integers(0, len(names)))]]
        p = int(grng.integers(0, per - 64))
        guides.append(c[p:p + guidelen].tobytes().decode())
    t_syn = time.time() - t0
    t0 = time.time()
    idx = GenomeIndex(contigs, guidelen, len(pam), device=R.device, shard=(R.rank, R.world))
    t_idx = time.time() - t0
    del contigs
    log(f"c5 workload: {args.genome_nt} nt genome in {args.genome_contigs} contigs, rows {idx.row_lo}..{idx.row_hi} of {idx.n_rows_total} on this rank; "
        f"synthesised in {t_syn:.1f}s, packed + one-hot in {t_idx:.1f}s; {len(guides)} guides {pam_s}/{guidelen} mm<={args.mm}")

    def step():
        return idx.scan_arrays(guides, pam, right, args.mm)

    elapsed, outs = timed_steps(R, step, args.steps, args.warmup)
    hits, tm = outs[-1]
    tms = [o[1] for o in outs]
    avg = lambda k: float(np.mean([t[k] for t in tms]))
    tot = R.sum_ints([tm["n_sites"], tm["scanned_positions"], len(hits["guide"])])
    pairs_all = tot[0] * len(guides)
    out = None
    if R.rank == 0:
        out = base_line(args, R, pairs_all * args.steps / elapsed, elapsed, "strong", "u64 (2-bit codes, XOR + popcount)",
                        {"workload": f"C5: off-target enumeration, {pam_s} / {guidelen} nt / right={right} / <= {args.mm} mismatches vs a "
                                     f"{args.genome_nt}-nt packed synthetic genome (BASELINE.json configs[4])",
                         "guides": len(guides), "genome_nt": args.genome_nt, "contigs": args.genome_contigs,
                         "partition": "genome rows (4 Mb pieces) block-partitioned over ranks, guides replicated",
                         "pam_sites_all_ranks": tot[0], "hits_all_ranks": tot[2], "scanned_positions": tot[1]})
        out["metric"] = "PAM-site x guide comparisons/sec (off-target enumeration)"
        out["unit"] = "site-guide pairs/s"
        pairs = tm["n_sites"] * len(guides)
        # What bounds the seeded match is vector-instruction issue (it moves almost nothing through HBM): `achieved` = the
        # wave64 VALU instructions the kernel EXECUTES per second x 64 lanes - counted by rocprofv3 (SQ_INSTS_VALU x SQ_WAVES
        # of the committed profile, profiles/r04_c5_pmc.json, scaled by the pairs of this run) over the launch time measured
        # here - against the chip's integer lane-op peak.  Skipped pairs are not priced: that is the filter's gain and shows
        # in `value`, not in the fraction.
        pmc = json.load(open(PROFILE_C5_PMC)) if os.path.exists(PROFILE_C5_PMC) else None
        roof = {"bound": "valu", "kernel": "k_ot_match_pairs (pair-seeded match, candidates dealt evenly over a wave)", "achieved": None,
                "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-ops/s (executed wave64 VALU instructions x 64)", "frac": None,
                "traffic": None, "launch_ms": avg("match_ms")}
        if pmc and avg("match_ms"):
            k = pmc["k_ot_match_pairs"]
            c = pmc["_config"]
            ref_pairs = None
            insts = k["waves"] * k["per_wave"]["SQ_INSTS_VALU"]
            same = (c["genome_nt"], c["guides"], c["pam"], c["guidelen"], c["mm"]) == (args.genome_nt, args.guides, pam_s, guidelen, args.mm)
            roof["profile"] = {"valu_insts_per_launch": insts, "waves": k["waves"], "valu_insts_per_wave": k["per_wave"]["SQ_INSTS_VALU"],
                               "lds_insts_per_wave": k["per_wave"]["SQ_INSTS_LDS"], "lds_bank_conflict_cycles_per_wave": k["per_wave"]["SQ_LDS_BANK_CONFLICT"],
                               "same_config_as_this_run": same}
            roof["profile"]["wait_fraction_of_wave_life"] = k["per_wave"]["SQ_WAIT_ANY"] / k["per_wave"]["SQ_WAVE_CYCLES"]
            roof["note"] = ("not VALU-bound any more: the pair seeds leave 3.3e8 candidate pairs of 7.3e11, the kernel executes a fifth of round 3's "
                            "instructions and its waves wait on L2 / LDS round trips for most of their life - `frac` is kept as executed VALU over peak "
                            "for continuity")
            if same:
                ach = insts * 64 / (avg("match_ms") * 1e-3)
                roof["achieved"], roof["frac"] = ach / 1e12, ach / VALU_PEAK_LANE_OPS
                roof["lane_ops_per_resolved_pair"] = insts * 64 / pairs
        roof["scan_kernel"] = {"kernel": "k_scan_raw", "launch_ms": avg("scan_ms"),
                               "achieved_GBps": 0.75 * tm["scanned_positions"] / (avg("scan_ms") * 1e-3) / 1e9 if avg("scan_ms") else None,
                               "frac_of_hbm": 0.75 * tm["scanned_positions"] / (avg("scan_ms") * 1e-3) / 1e9 / HBM_PEAK_GBS if avg("scan_ms") else None}
        out["roofline"] = roof
        out["kernels_ms"] = {k: avg(k) for k in ("scan_ms", "sites_ms", "match_ms", "total_ms")}
        out["index_build_s"] = t_idx
    if R.world > 1:
        from crisprhawk_hip import parallel
        t0 = time.perf_counter()
        comm = R.ctl
        if R.backend == "rccl":
            try:
                comm = parallel.RcclComm(R.ctl, R.device)
            except Exception as e:
                log(f"RCCL unavailable for the hit gather ({e}); using the TCP control plane")
        parts = {k: comm.gatherv_bytes(v, 0) for k, v in hits.items()}
        if R.rank == 0:
            out["gather"] = {"hits": int(sum(len(p) for p in parts["guide"])), "wall_ms": (time.perf_counter() - t0) * 1e3,
                             "via": type(comm).__name__}
    if R.rank == 0 and R.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_offtargets(args, pam_s, guidelen, right, guides)
    return out


def cpu_baseline_offtargets(args, pam_s, guidelen, right, guides):
    """The oracle's brute force (every window x every guide, both strands) on a genome sample sized for ~10 s."""
    from oracle import oracle as ora
    rng = np.random.default_rng(1006)
    n = 5_000_000
    g = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].tobytes().decode()
    sub = guides[:2000]
    t0 = time.perf_counter()
    hits = ora.offtargets(g, sub, pam_s, right, args.mm)
    dt = time.perf_counter() - t0
    # PAM density of the sample stands in for "sites": pairs the brute force resolves = PAM-bearing windows x guides
    bits, bitsrc, _, _ = ora.pam_encode(pam_s)
    f, r = ora.scan(ora.encode(g), 0, n - len(pam_s) + 1, bits, bitsrc, len(pam_s))
    pairs = (len(f) + len(r)) * len(sub)
    return {"value": pairs / dt, "unit": "site-guide pairs/s", "cores": 1, "kind": "port",
            "sample": f"oracle brute force on a {n}-nt sample of the genome x {len(sub)} guides ({len(hits)} hits, {dt:.1f} s, 1 thread)"}


# ---------------------------------------------------------------------------------------------------
def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["c1", "c3", "c4", "c5"], default="c3")
    ap.add_argument("--weak", action="store_true", help="every rank its own panel (weak scaling) instead of one block-partitioned panel")
    ap.add_argument("--samples", type=int, default=2504)
    ap.add_argument("--sites", type=int, default=31000, help="variant sites per --region-len bases")
    ap.add_argument("--region-len", type=int, default=1_000_000)
    ap.add_argument("--pam", default="NGG")
    ap.add_argument("--guidelen", type=int, default=20)
    ap.add_argument("--right", action="store_true")
    ap.add_argument("--contig-len", type=int, default=50_818_468, help="c4: contig length (chr22)")
    ap.add_argument("--n-block-frac", type=float, default=10.5 / 50.818468, help="c4: leading N block as a fraction of the contig")
    ap.add_argument("--tile-nt", type=int, default=4_000_000, help="c4: bases a tile owns")
    ap.add_argument("--genome-nt", type=int, default=3_100_000_000, help="c5: genome size")
    ap.add_argument("--genome-contigs", type=int, default=24)
    ap.add_argument("--guides", type=int, default=10_000, help="c5: unique spacers")
    ap.add_argument("--mm", type=int, default=4)
    ap.add_argument("--cpu-haps", type=int, default=1280, help="haplotypes in the cpu_baseline sample (about 12 s of single-thread oracle work on C3)")
    ap.add_argument("--panel", choices=["independent", "linked"], default="independent",
                    help="c3: genotypes drawn independently per site and column (SURVEY 8(d): no LD, worst case for sharing) or as mosaics of "
                         "128 founder haplotypes (linkage blocks of ~100 kb)")
    ap.add_argument("--shard", choices=["region", "samples"], default="region",
                    help="c3, N > 1: what a rank takes of the one panel - a stretch of the region x all samples (default; everything "
                         "divides by N), or a block of the samples x the whole region (REF and nearly every distinct cluster on every rank)")
    ap.add_argument("--no-shares", action="store_true", help="skip shares_on_one_gpu (one rank's share of a 2 / 4 / 8-way split, timed on this GPU)")
    ap.add_argument("--planes", action="store_true", help="c3: time round 2's step (hawk_search over materialised planes) instead of the fused step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-all-cores", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-files", action="store_true", help="end_to_end: skip files_to_tsv (FASTA + BED + VCF files -> report TSV on disk)")
    ap.add_argument("--vcf", action="store_true", help="end_to_end: also time the VCF-text ingest of the workload (f3)")
    ap.add_argument("--report", action="store_true", help="end_to_end: also assemble the guide report (f2) of the whole workload")
    ap.add_argument("--no-gather", action="store_true", help="skip the one-off RCCL gather of the guide tables after the timed loop")
    ap.add_argument("--gather-timeout", type=float, default=180.0, help="seconds the post-loop exchange may take before the line is printed without it")
    return ap


def main():
    args = build_parser().parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # our own launcher: N fresh processes, started before this one has made any GPU call
        sys.exit(launch(args.gpus, sys.argv[1:]))
    defaults = {"c3": (20, 3), "c1": (200, 20), "c4": (2, 1), "c5": (3, 1)}[args.config]
    if args.steps is None:
        args.steps = defaults[0]
    if args.warmup is None:
        args.warmup = defaults[1]
    R = Ranks(args)
    if args.config in ("c3", "c1"):
        out = run_region(args, R)
    elif args.config == "c4":
        out = run_c4(args, R)
    else:
        out = run_c5(args, R)
    R.ctl.barrier()
    R.ctl.close()
    if R.rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
