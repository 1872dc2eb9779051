#!/usr/bin/env python3
"""bench.py — candidate guides scored per second on the BASELINE.json workload.

A "step" is one pass of the device hot path (hawk_search: PAM scan fused with the in-range and
REF-identical filters -> compaction -> coordinates / redundancy removal / window gather ->
CFDon) over one haplotype set that is already resident in HBM as bit-sliced planes.  Default
workload = BASELINE.json configs[2] ("C3": 1 Mb region x 2504 phased samples, NGG, 20 nt), the
configuration the metric is quoted on.

    python bench.py [--gpus N --steps K --warmup W]        # N > 1 via torch.distributed.run

One JSON line on stdout (rank 0).  `roofline` prices the dominant kernel (the longer of k_search_count
and k_emit_list) with its algorithmic bytes against the 8 TB/s HBM peak, using its HIP-event duration
measured on the stream it runs on:
  k_emit_list     0.625 B read per scanned haplotype position (five bit planes) + 74 B written and 4 B
                  of hand-over list read per guide row;
  k_search_count  one bit per position for each plane the PAM names plus the variant plane
                  (NGG/CCN: 0.375 B) + 4 B of hand-over list written per guide row.
`cpu_baseline` times the
C oracle (a port of the reference's algorithm, oracle/hawk_oracle.c) on a bounded sample of the
same workload on the host cores (rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "crispr-hawk_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Algorithmic bytes of one fused launch: every scanned haplotype position is read once as
# 0.5 B of IUPAC code + 0.125 B of the variant plane (K3's REF-identical filter);
# the emit pass additionally writes each guide row once (74 B: K3 record + packed window + K4 score).
READ_BYTES_PER_POS = 0.625
ROW_BYTES = 74
LIST_BYTES = 4  # one u32 hand-over entry per guide row (count pass writes it, k_emit_list reads it)


def log(msg):
    print(f"[bench r{os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def time_vcf_ingest(reg, ds_ref, pamlen, device):
    """f3 at workload scale: the VCF text of the region's records (what readers.VCF.fetch_block returns) ->
    hawk_gt_parse -> hawk_gt_lists -> hawk_hapset_expand, compared with the planes of the in-memory path."""
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
    log(f"vcf ingest: {len(text) / 1e6:.0f} MB of records -> {ds2.n_hap} haplotype rows in {t2 - t1:.2f}s "
        f"(parse {ms['parse']:.2f} ms, lists {ms['lists']:.2f} ms, expand {ms['expand']:.2f} ms); planes identical: {same}")
    return {"text_bytes": int(len(text)), "records": len(blk), "samples": ns, "make_text_s": t1 - t0, "ingest_wall_s": t2 - t1,
            "kernels_ms": ms, "parse_GBps": len(text) / (ms["parse"] * 1e-3) / 1e9 if ms["parse"] else None,
            "planes_identical_to_in_memory_path": same}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=2504)
    ap.add_argument("--sites", type=int, default=31000)
    ap.add_argument("--region-len", type=int, default=1_000_000)
    ap.add_argument("--pam", default="NGG")
    ap.add_argument("--guidelen", type=int, default=20)
    ap.add_argument("--right", action="store_true")
    ap.add_argument("--cpu-haps", type=int, default=1280, help="haplotypes in the cpu_baseline sample (about 12 s of single-thread oracle work on C3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-expand", action="store_true", help="build the haplotype strings on the host and pack them (K1) instead of expanding on the device")
    ap.add_argument("--vcf", action="store_true", help="also time the VCF-text ingest of the workload (f3: device genotype parser + carried lists)")
    ap.add_argument("--report", action="store_true", help="also assemble the guide report (f2) of the whole workload once and time it")
    ap.add_argument("--no-collapse", action="store_true", help="skip the one-off report-row collapse after the timed loop")
    ap.add_argument("--no-gather", action="store_true", help="skip the one-off RCCL gather of the guide tables after the timed loop")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    backend = os.environ.get("HAWK_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N > 1 control flow without RCCL
    cdev = "cuda" if backend == "nccl" else "cpu"           # where the small collective tensors live
    if world > 1:
        import torch
        import torch.distributed as dist  # RCCL ("nccl") process group: barrier, timing reduce, table gather

        if os.environ.get("HAWK_BENCH_ONE_GPU") == "1":  # rehearsal: every rank on GPU 0 of a one-GPU box
            local = 0
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from crisprhawk_hip import _lib, synth
    from crisprhawk_hip.hapset import DeviceHapSet
    from crisprhawk_hip.pam import PAM
    from crisprhawk_hip.workload import build_phased_haplotypes, expand_on_device

    if _lib.device_count() == 0:
        raise _lib.HawkDeviceError("bench.py needs an MI355X: there is no CPU fallback on the product path")

    # ---- workload: haplotypes are independent units -> each rank owns its own 2504 samples
    # (weak scaling: per-GPU work fixed); REF is on every rank (needed by the alt==REF filter)
    t0 = time.time()
    reg = synth.make_region(1003, "chr22", args.region_len + 200_000, 100_000, 100_000 + args.region_len)
    synth.add_phased_variants(reg, 1003_1 + 7919 * rank, args.sites, args.samples)
    pam = PAM(args.pam, args.right, True)
    pam.encode(0)
    expand_ms = None
    if args.host_expand:
        haps, _info = build_phased_haplotypes(reg, len(pam))
        log(f"workload: {len(haps)} haplotypes x {len(haps[0].seq)} nt, {len(reg.variants)} sites, built on the host in {time.time() - t0:.1f}s")
        t0 = time.time()
        ds = DeviceHapSet(haps, device=local)
        region_nt = len(haps[0].seq)
        log(f"resident in HBM ({5 * ds.n_hap * ds.stride * 4 / 1e9:.2f} GB of planes) in {time.time() - t0:.1f}s")
    else:  # the haplotypes are expanded on the device from REF + variant table + genotypes
        t1 = time.time()
        ds, _info, expand_ms, _kept = expand_on_device(reg, len(pam), device=local)
        region_nt = int(ds.hap_len[0])
        log(f"workload: {ds.n_hap} haplotype rows x {region_nt} nt, {len(reg.variants)} sites; synthesised in {t1 - t0:.1f}s, "
            f"expanded on the device in {time.time() - t1:.1f}s (kernels {expand_ms:.2f} ms, {5 * ds.n_hap * ds.stride * 4 / 1e9:.2f} GB of planes)")
    score = (not args.right) and pam.cas_system in (3, 4)  # scoring.py:749-792: CFDon for SpCas9/xCas9 PAMs
    mm, pt = synth.cfd_tables() if score else (None, None)

    counts_all = None
    if dist is not None:
        import torch
        counts_all = torch.zeros(2 * world, dtype=torch.int64, device=cdev)

    def step(keep=False):
        tab = ds.search(pam.bits, pam.bitsrc, len(pam), args.guidelen, args.right, mm, pt, download=False)
        if dist is not None:  # the table directory every rank needs before any exchange: rows + candidates per rank
            mine = torch.tensor([tab.n_rows, tab.n_candidates], dtype=torch.int64, device=cdev)
            dist.all_gather_into_tensor(counts_all, mine)
        if not keep:
            tab.close()
        return tab

    def barrier():
        _lib.check(_lib.lib().hawk_sync(ds._ctx), "hawk_sync")
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t_start = time.perf_counter()
    scan_ms, tot_ms, kern = [], [], {"offsets_ms": [], "emit_ms": [], "emit_list_ms": []}
    tab = None
    for _ in range(args.steps):
        tab = step()
        scan_ms.append(tab.timing["count_ms"])
        tot_ms.append(tab.timing["total_ms"])
        for k in kern:
            kern[k].append(tab.timing[k])
    barrier()
    elapsed = time.perf_counter() - t_start
    cand, rows, positions = tab.n_candidates, tab.n_rows, tab.timing["scanned_positions"]
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([cand, rows, positions], dtype=torch.int64, device=cdev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        cand_all, rows_all, pos_all = (int(x) for x in c.tolist())
    else:
        cand_all, rows_all, pos_all = cand, rows, positions

    # the PAM-scan kernel on its own (K2, what `pam_search` runs): north_star's >= 40 % of HBM peak target
    import ctypes as C
    ps_ms, ps_pos = C.c_float(0), C.c_uint64(0)
    _lib.check(_lib.lib().hawk_pam_scan_time(ds._h, C.c_uint64(pam.bits), C.c_uint64(pam.bitsrc), len(pam), 20, C.byref(ps_ms),
                                             C.byref(ps_pos)), "hawk_pam_scan_time")
    # f2 (outside the timed steps): which rows the report merges, sorted and grouped in HBM
    collapse = None
    if rank == 0 and not args.no_collapse:
        tc = step(keep=True) if dist is None else ds.search(pam.bits, pam.bitsrc, len(pam), args.guidelen, args.right, mm, pt, download=False)
        ng, cms = C.c_uint64(), C.c_float()
        _lib.check(_lib.lib().hawk_table_collapse(tc._t, C.byref(ng), C.byref(cms)), "hawk_table_collapse")
        collapse = {"kernels_ms": cms.value, "rows": tc.n_rows, "groups": ng.value,
                    "what": "k_collapse_keys + rocprim radix sort (key, row) + k_collapse_heads + scan + k_collapse_groups"}
        if args.report and not args.host_expand:
            from crisprhawk_hip import reports as hip_reports
            from crisprhawk_hip.workload import row_labels
            t0 = time.perf_counter()
            tc.collapse()
            t1 = time.perf_counter()
            inp = hip_reports.ReportInput.from_table(tc)
            t2 = time.perf_counter()
            df = hip_reports.report_frame(inp, row_labels(reg, ds, _info, _kept), pam, reg.contig, f"{reg.contig}:{reg.bed_start}-{reg.bed_stop}")
            t3 = time.perf_counter()
            collapse["report"] = {"report_rows": int(len(df)), "collapse_and_download_group_arrays_s": t1 - t0,
                                  "table_download_and_decode_s": t2 - t1, "assemble_s": t3 - t2,
                                  "what": "crisprhawk_hip.reports.report_frame: one pass per report row (variants, AFs, samples, order)"}
            log(f"report: {len(df)} rows from {tc.n_rows} guide rows in {t3 - t0:.1f}s")
        tc.close()
    vcf_ingest = None
    if rank == 0 and args.vcf and not args.host_expand:
        vcf_ingest = time_vcf_ingest(reg, ds, len(pam), local)
    gather = None
    if dist is not None and not args.no_gather and backend == "nccl":
        gather = gather_once(ds, step, dist, rank, world)

    out = None
    if rank == 0:
        scan_avg_ms = float(np.mean(scan_ms))
        emit_avg_ms = float(np.mean(kern["emit_ms"]))
        # dominant kernel = the longer of the count pass and the list-driven emit pass, each priced with the
        # bytes it has to move
        need = 0
        for nib in pam.bits_list + [{"A": 1, "C": 2, "G": 4, "T": 8}.get(c, 0) or synth_iupac(c) for c in pam.pamrc.upper()]:
            if nib != 15:
                need |= nib
        count_bpp = 0.125 * (bin(need).count("1") + 1)
        emit_list_avg_ms = float(np.mean(kern["emit_list_ms"]))
        if emit_list_avg_ms == 0.0:  # HAWK_LIST_EMIT=0: the recompute-everything emit pass
            emit_name, emit_ms_, emit_bytes = "k_search_emit", emit_avg_ms, READ_BYTES_PER_POS * positions + ROW_BYTES * rows
            count_bytes = count_bpp * positions
        else:
            emit_name, emit_ms_, emit_bytes = "k_emit_list", emit_list_avg_ms, READ_BYTES_PER_POS * positions + (ROW_BYTES + LIST_BYTES) * rows
            count_bytes = count_bpp * positions + LIST_BYTES * rows
        if emit_ms_ >= scan_avg_ms:
            dom, dom_ms, algo_bytes, bpp = emit_name, emit_ms_, emit_bytes, READ_BYTES_PER_POS
        else:
            dom, dom_ms, algo_bytes, bpp = "k_search_count", scan_avg_ms, count_bytes, count_bpp
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath) and (args.samples, args.sites, args.region_len, args.pam, args.guidelen) == (2504, 31000, 1_000_000, "NGG", 20):
            tk = json.load(open(tpath))["kernels"].get(dom)
            traffic = tk and tk["fetch_bytes"] + tk["write_bytes"]  # PMC bytes per launch of the committed profile
        out = {
            "metric": "candidate guides scored/sec", "value": cand_all * args.steps / elapsed, "unit": "candidates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "C3: 1 Mb region x 2504 phased samples (BASELINE.json configs[2])" if (args.samples, args.region_len) == (2504, 1_000_000) else "custom",
                       "pam": args.pam, "guidelen": args.guidelen, "right": args.right, "region_nt": region_nt,
                       "haplotypes_per_gpu": ds.n_hap, "samples_per_gpu": args.samples, "variant_sites": len(reg.variants),
                       "scored": "CFDon (synthetic tables, seed 2001)" if score else "none",
                       "candidates_per_step": cand_all, "guide_rows_per_step": rows_all, "scanned_positions_per_step": pos_all},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "launch_ms": dom_ms,
                         "algorithmic_bytes_per_launch": algo_bytes, "read_bytes_per_position": bpp,
                         "row_bytes": ROW_BYTES, "list_bytes_per_row": LIST_BYTES,
                         "other_kernel": {"kernel": "k_search_count" if dom != "k_search_count" else emit_name,
                                          "launch_ms": scan_avg_ms if dom != "k_search_count" else emit_ms_,
                                          "algorithmic_bytes_per_launch": count_bytes if dom != "k_search_count" else emit_bytes,
                                          "frac": (count_bytes / (scan_avg_ms * 1e-3) if dom != "k_search_count" else emit_bytes / (emit_ms_ * 1e-3)) / 1e9 / HBM_PEAK_GBS}},
            "kernels_ms": {"count": scan_avg_ms, **{k[:-3]: float(np.mean(v)) for k, v in kern.items()},
                           "device_total": float(np.mean(tot_ms))},
        }
        # bytes the scan really moves: one bit per position for each plane the PAM names (NGG/CCN: G and C only)
        # plus the forward and reverse hit bits; SURVEY.md §8(d) prices the nibble formulation at 0.75 B/position.
        need = 0
        for nib in pam.bits_list + [{"A": 1, "C": 2, "G": 4, "T": 8}.get(c, 0) or synth_iupac(c) for c in pam.pamrc.upper()]:
            if nib != 15:
                need |= nib
        ps_bpp = 0.125 * bin(need).count("1") + 0.25
        ps_bytes = ps_bpp * ps_pos.value
        out["pam_scan_kernel"] = {"kernel": "k_scan_raw", "launch_ms": ps_ms.value, "bytes_per_position": ps_bpp,
                                  "bytes_per_launch": ps_bytes, "achieved": ps_bytes / (ps_ms.value * 1e-3) / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ps_bytes / (ps_ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "survey_algorithmic_bytes_per_position": 0.75,
                                  "positions_per_s": ps_pos.value / (ps_ms.value * 1e-3)}
        if expand_ms is not None:
            out["haplotype_expansion"] = {"kernels_ms": expand_ms, "rows": ds.n_hap, "where": "device (hawk_hapset_expand), outside the timed steps"}
        if collapse is not None:
            out["collapse"] = collapse
        if vcf_ingest is not None:
            out["vcf_ingest"] = vcf_ingest
        if gather is not None:
            out["gather"] = gather
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(reg, pam, args, mm, pt)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def synth_iupac(c):
    from crisprhawk_hip.pam import IUPAC_BITS
    return IUPAC_BITS[c]


def gather_once(ds, step, dist, rank, world):
    """The single exchange of the job: every rank's guide table to rank 0 over RCCL (xGMI), device to
    device.  Measured once after the timed loop and reported next to `value`, not inside it: the
    per-step path has no data-path collective."""
    import torch
    tab = step(keep=True)
    n = tab.n_rows
    cnt = torch.zeros(world, dtype=torch.int64, device="cuda")
    dist.all_gather_into_tensor(cnt, torch.tensor([n], dtype=torch.int64, device="cuda"))
    nmax = int(cnt.max().item())
    widths = (4, 4, 1, 8, 8, 1, 8, 40)  # hap pos strand start stop flags cfdon win[5]
    send = [torch.zeros(nmax * w, dtype=torch.uint8, device="cuda") for w in widths]
    tab.export_to(*[t.data_ptr() for t in send])
    tab.close()
    recv = [[torch.empty_like(t) for _ in range(world)] for t in send] if rank == 0 else [None] * len(send)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for t, r in zip(send, recv):
        dist.gather(t, r, dst=0)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    total = int(cnt.sum().item()) * sum(widths)
    return {"ms": dt * 1e3, "bytes_to_rank0": total, "GBps": total / dt / 1e9, "rows": int(cnt.sum().item())}


def cpu_baseline(reg, pam, args, mm, pt):
    """The oracle (C port of the reference algorithm) on the first --cpu-haps haplotypes of the same workload, one
    host thread: search + reverse_guides + CFDon.  The sample is walked in batches of 80 samples (REF + <= 160
    haplotypes each, so host memory stays bounded); building the batch's strings and marshalling them are not
    timed (the GPU side starts from resident planes too)."""
    from crisprhawk_hip.workload import build_phased_haplotypes
    from oracle import oracle as ora

    cand, dt, nh, per = 0, 0.0, 0, 80
    n_samples = len(reg.samples)
    b = 0
    while nh < max(2, args.cpu_haps) and b * per < max(1, n_samples):
        sub, _ = build_phased_haplotypes(reg, len(pam), sample_slice=slice(b * per, (b + 1) * per))
        hs = ora.HapSet([bytes(h.seq).decode("ascii") for h in sub], [h.seg.full() for h in sub], [h.is_ref for h in sub],
                        [h.scan for h in sub])
        blob_args = hs.packed()
        hs.packed = lambda blob_args=blob_args: blob_args
        t0 = time.perf_counter()
        res = ora.search(hs, pam.pam, args.guidelen, args.right)
        ora.reverse_and_cfdon(res, hs.is_ref, args.guidelen, len(pam), mm, pt, decode=False)
        dt += time.perf_counter() - t0
        cand += res.n_candidates
        nh += len(sub)
        b += 1
        if not reg.variants:
            break
    return {"value": cand / dt, "unit": "candidates/s", "cores": 1, "kind": "port",
            "sample": f"{nh} haplotype scans of the same workload in {b} batches of REF + <= {2 * per} haplotypes "
                      f"({cand} candidates, {dt:.1f} s, 1 thread of {os.cpu_count()})"}


if __name__ == "__main__":
    main()
