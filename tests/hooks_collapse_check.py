"""Run by tests/test_gpu_parity.py::test_collapse_verify_pass_catches_hash_collisions in a process of its own, with
CRISPRHAWK_HIP_LIB naming libhawk_hip_hooks.so - the library built with -DHAWK_TEST_HOOKS, the only build in which the
collapse hash can be weakened and its verification pass switched off.

The grouping is verified against the full keys by default (k_collapse_verify).  With HAWK_COLLAPSE_WEAK_HASH=1 every row of one
(start, strand) hashes alike - the worst collision there can be: unverified, the groups come out merged (fewer than the
oracle's); verified, the call notices and reruns exactly, and the groups are the oracle's."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "crispr-hawk_amd"), HERE]

import numpy as np  # noqa: E402

from crisprhawk_hip import _lib, synth  # noqa: E402
from oracle import oracle as ora  # noqa: E402
from test_gpu_parity import _check_collapse, device_set  # noqa: E402
from util import oracle_haplotypes  # noqa: E402


def main() -> int:
    assert os.path.basename(_lib.LIB_PATH) == "libhawk_hip_hooks.so", _lib.LIB_PATH
    reg = synth.make_region(7501, "chrC", 40_000, 1_000, 38_000)
    synth.add_phased_variants(reg, 7502, 300, 6, af_min=0.3, af_max=0.8)
    fx = dict(region_seq=reg.sequence, startp=reg.startp, samples=reg.samples,
              variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants])
    haps = oracle_haplotypes(fx)
    scan = [ora.scan_bounds(h["posmap"], reg.startp, reg.stopp, 3) for h in haps]
    hs = ora.HapSet([h["seq"] for h in haps], [h["posmap"] for h in haps], [h["samples"] == ["REF"] for h in haps], scan)
    bits, bitsrc, _, _ = ora.pam_encode("NGG")
    ds = device_set(hs)
    for k in ("HAWK_COLLAPSE_EXACT", "HAWK_COLLAPSE_MODE", "HAWK_COLLAPSE_WEAK_HASH", "HAWK_COLLAPSE_VERIFY"):
        os.environ.pop(k, None)
    good = ds.search(bits, bitsrc, 3, 20, False, collapse=True)
    os.environ["HAWK_COLLAPSE_WEAK_HASH"] = "1"
    os.environ["HAWK_COLLAPSE_VERIFY"] = "0"
    merged = ds.search(bits, bitsrc, 3, 20, False, collapse=True)
    assert merged.n_groups < good.n_groups  # the collisions are real: without the check different rows share a group
    os.environ["HAWK_COLLAPSE_VERIFY"] = "1"
    tab = ds.search(bits, bitsrc, 3, 20, False, collapse=True)
    assert tab.n_groups == good.n_groups
    assert np.array_equal(tab.group_perm, good.group_perm) and np.array_equal(tab.group_off, good.group_off)
    _check_collapse(hs, tab, 20, 3, False)
    print("hooks ok")
    return 0


if __name__ == "__main__":
    sys.exit(main())
