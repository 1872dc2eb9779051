"""The RCCL exchange through the C ABI (hawk_comm_*, hawk_table_gather) on the one GPU a test box has: a world of
one rank exercises the library loading (dlopen of librccl), communicator creation from a unique id, the all-gather of
the table directory, the self-copy branch of the grouped send/recv and the haplotype-offset kernel.  The multi-rank
branches are the same calls with peers; their host logic is covered by tests/test_parallel_gloo.py."""
import numpy as np
import pytest

from crisprhawk_hip import parallel, synth
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.workload import expand_on_device

pytestmark = pytest.mark.gpu


def test_rccl_world_of_one_gathers_a_table_onto_itself(capfd):
    reg = synth.make_region(9101, "chrR", 30_000, 1_000, 28_000)
    synth.add_phased_variants(reg, 9102, 300, 5, af_min=0.2, af_max=0.7)
    pam = PAM("NGG", False, True)
    pam.encode(0)
    mm, pt = synth.cfd_tables()
    ds, _info, _ms, _kept = expand_on_device(reg, 3)
    tab = ds.search(pam.bits, pam.bitsrc, 3, 20, False, mm, pt, download=False)
    comm = parallel.RcclComm(parallel.TcpComm(0, 1))
    assert comm.allgather_i64([7, -3]).tolist() == [[7, -3]]
    parts = comm.gatherv_bytes(np.arange(10, dtype=np.int32).reshape(5, 2), 0)
    assert len(parts) == 1 and parts[0].tolist() == np.arange(10).reshape(5, 2).tolist()
    merged, ms = comm.gather_table(tab, hap_offset=1000, dst=0)
    assert merged.n_rows == tab.n_rows and merged.n_candidates == tab.n_candidates and ms >= 0
    merged.download()
    tab.download()
    want_hap = np.where(tab.hap == 0, 0, tab.hap.astype(np.int64) + 1000).astype(np.uint32)
    assert np.array_equal(merged.hap, want_hap)
    for col in ("pos", "strand", "start", "stop", "flags", "win"):
        assert np.array_equal(getattr(merged, col), getattr(tab, col)), col
    assert np.array_equal(np.isnan(merged.cfdon), np.isnan(tab.cfdon))
    assert np.array_equal(np.nan_to_num(merged.cfdon), np.nan_to_num(tab.cfdon))
    comm.close()
    # RCCL's start-up banner must not reach stdout: bench.py's stdout is one JSON line
    assert "RCCL version" not in capfd.readouterr().out


def test_stale_table_is_refused():
    """A second search on the same set overwrites the first table's columns: the old handle must say so."""
    from crisprhawk_hip import _lib
    reg = synth.make_region(9111, "chrS", 12_000, 1_000, 11_000)
    pam = PAM("NGG", False, True)
    pam.encode(0)
    ds, _i, _m, _k = expand_on_device(reg, 3)
    t1 = ds.search(pam.bits, pam.bitsrc, 3, 20, False, download=False)
    t2 = ds.search(pam.bits, pam.bitsrc, 3, 20, False, download=False)
    with pytest.raises(_lib.HawkStatusError):
        t1.download()
    with pytest.raises(_lib.HawkStatusError):
        t1.collapse()
    t2.collapse()
    t2.download()
    assert t2.n_rows > 1000 and t2.n_groups == t2.n_rows
