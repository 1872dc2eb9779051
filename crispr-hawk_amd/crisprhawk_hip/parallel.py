"""One process per GPU: haplotype sharding and the guide-table exchange.

Haplotypes are independent units for the scan / filter / scoring kernels
(search_guides.py:111-131, 530-547); the only cross-haplotype inputs are the REF haplotype's
windows (redundancy filter, CFDon wild type), so REF is replicated on every rank and the
samples are block-partitioned (`shard_range`).  No collective sits on the search path.  After the
search each rank holds its own guide table in HBM; the single exchange is the gather of those
tables to one rank.

Two communicators implement the same small interface (rank, world, barrier, allgather_i64,
gatherv_bytes, gather_table):

* ``RcclComm``  - the product path: RCCL over xGMI through the C ABI (hawk_comm_* / hawk_table_gather,
  csrc/hawk_comm.hip; grouped ncclSend / ncclRecv straight from the tables' device columns).
* ``TcpComm``   - plain sockets between the ranks of one node: the bootstrap that carries RCCL's
  128-byte unique id and the barrier / timing reductions of bench.py, and the CPU stand-in for the
  exchange in tests (host arrays only).

No torch anywhere: rendezvous is a port file in a directory all local ranks agree on
(MASTER_PORT + the launcher's pid), written by rank 0.
"""
import ctypes as C
import os
import pickle
import socket
import struct
import tempfile
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

COLUMNS = (("hap", np.uint32), ("pos", np.uint32), ("strand", np.uint8), ("start", np.int64), ("stop", np.int64),
           ("flags", np.uint8), ("cfdon", np.float64))


def shard_range(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of units [lo, hi) owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, world, local_rank) as torch.distributed.run / bench.py's own launcher export them."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


# ---------------------------------------------------------------------------------------------
# TCP communicator (control plane + CPU stand-in)
# ---------------------------------------------------------------------------------------------
def _send_msg(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("peer closed the connection")
        got += k
    return bytes(buf)


def _recv_msg(sock: socket.socket) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


def rendezvous_dir() -> str:
    """A directory every rank of this launch (and no other launch) computes identically: keyed by MASTER_PORT and the
    pid of the common parent (the torch.distributed.run agent or bench.py's launcher)."""
    d = os.environ.get("HAWK_RDZV_DIR")
    if d:
        return d
    return os.path.join(tempfile.gettempdir(), f"hawk_rdzv_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")


class TcpComm:
    """Star topology over localhost sockets: rank 0 listens on an ephemeral port and publishes it in a file."""

    def __init__(self, rank: int, world: int, rdzv: Optional[str] = None, timeout: float = 600.0, tag: str = "ctl"):
        self.rank, self.world = rank, world
        self._peers: List[Optional[socket.socket]] = [None] * world
        self._sock: Optional[socket.socket] = None
        if world == 1:
            return
        rdzv = rdzv or rendezvous_dir()
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port_file = os.path.join(rdzv, f"{tag}.port")
        if rank == 0:
            os.makedirs(rdzv, exist_ok=True)
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("0.0.0.0" if addr not in ("127.0.0.1", "localhost") else "127.0.0.1", 0))
            srv.listen(world)
            tmp = port_file + f".tmp{os.getpid()}"
            with open(tmp, "w") as f:
                f.write(str(srv.getsockname()[1]))
            os.replace(tmp, port_file)  # atomic: readers never see a partial file
            srv.settimeout(timeout)
            for _ in range(world - 1):
                conn, _a = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                r = struct.unpack("<I", _recv_exact(conn, 4))[0]
                self._peers[r] = conn
            srv.close()
            try:
                os.remove(port_file)
                os.rmdir(rdzv)
            except OSError:
                pass
        else:
            t0 = time.time()
            while not os.path.exists(port_file):
                if time.time() - t0 > timeout:
                    raise TimeoutError(f"rank {rank}: no rendezvous file {port_file}")
                time.sleep(0.01)
            port = int(open(port_file).read())
            s = socket.create_connection((addr if addr != "0.0.0.0" else "127.0.0.1", port), timeout=timeout)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            s.sendall(struct.pack("<I", rank))
            self._sock = s

    # -- primitives ------------------------------------------------------------------------
    def allgather_obj(self, obj) -> list:
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            objs = [obj] + [None] * (self.world - 1)
            for r in range(1, self.world):
                objs[r] = pickle.loads(_recv_msg(self._peers[r]))
            blob = pickle.dumps(objs)
            for r in range(1, self.world):
                _send_msg(self._peers[r], blob)
            return objs
        _send_msg(self._sock, pickle.dumps(obj))
        return pickle.loads(_recv_msg(self._sock))

    def bcast_obj(self, obj, src: int = 0):
        return self.allgather_obj(obj if self.rank == src else None)[src]

    def barrier(self) -> None:
        self.allgather_obj(None)

    def allgather_i64(self, vec: Sequence[int]) -> np.ndarray:
        return np.asarray(self.allgather_obj([int(v) for v in vec]), dtype=np.int64).reshape(self.world, -1)

    def gatherv_bytes(self, arr: np.ndarray, dst: int = 0) -> Optional[List[np.ndarray]]:
        """Variable-length gather of a contiguous array's rows to `dst` (list of per-rank arrays there, None elsewhere)."""
        a = np.ascontiguousarray(arr)
        if self.world == 1:
            return [a]
        if dst != 0:
            raise ValueError("TcpComm gathers to rank 0 (the star's centre)")
        if self.rank == 0:
            out = [a]
            for r in range(1, self.world):
                shape, dt = pickle.loads(_recv_msg(self._peers[r]))
                raw = _recv_msg(self._peers[r])
                out.append(np.frombuffer(raw, dtype=dt).reshape(shape))
            return out
        _send_msg(self._sock, pickle.dumps((a.shape, a.dtype.str)))
        _send_msg(self._sock, a.tobytes())
        return None

    def close(self) -> None:
        for s in self._peers + [self._sock]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._sock = [None] * self.world, None


# ---------------------------------------------------------------------------------------------
# RCCL communicator (through the C ABI)
# ---------------------------------------------------------------------------------------------
class RcclComm:
    """hawk_comm_* over RCCL.  `ctl` (a TcpComm) carries the unique id from rank 0 and serves the host-side barrier."""

    def __init__(self, ctl: TcpComm, device: Optional[int] = None):
        from . import _lib
        self._lib, self._L = _lib, _lib.lib()
        self.ctl = ctl
        self.rank, self.world = ctl.rank, ctl.world
        self._ctx = _lib.context(device)
        uid = np.zeros(128, dtype=np.uint8)
        if self.rank == 0:
            _lib.check(self._L.hawk_comm_unique_id(uid.ctypes.data_as(C.c_void_p)), "hawk_comm_unique_id")
        uid = np.frombuffer(ctl.bcast_obj(uid.tobytes()), dtype=np.uint8).copy()
        self._c = C.c_void_p()
        _lib.check(self._L.hawk_comm_init(self._ctx, self.world, self.rank, uid.ctypes.data_as(C.c_void_p), C.byref(self._c)),
                   "hawk_comm_init")

    def barrier(self) -> None:
        self._lib.check(self._L.hawk_sync(self._ctx), "hawk_sync")
        self.ctl.barrier()

    def allgather_i64(self, vec: Sequence[int]) -> np.ndarray:
        mine = np.asarray(vec, dtype=np.int64).view(np.uint64)
        out = np.zeros(self.world * len(mine), dtype=np.uint64)
        self._lib.check(self._L.hawk_comm_allgather_u64(self._c, mine.ctypes.data_as(C.c_void_p), len(mine),
                                                        out.ctypes.data_as(C.c_void_p)), "hawk_comm_allgather_u64")
        return out.view(np.int64).reshape(self.world, -1)

    def gatherv_bytes(self, arr: np.ndarray, dst: int = 0) -> Optional[List[np.ndarray]]:
        a = np.ascontiguousarray(arr)
        row = int(np.prod(a.shape[1:], dtype=np.int64)) * a.dtype.itemsize if a.ndim > 1 else a.dtype.itemsize
        counts = self.allgather_i64([len(a)])[:, 0]
        off = np.zeros(self.world + 1, dtype=np.uint64)
        off[1:] = np.cumsum(counts.astype(np.uint64) * np.uint64(row))
        recv = np.empty(int(off[-1]), dtype=np.uint8) if self.rank == dst else np.empty(0, np.uint8)
        self._lib.check(self._L.hawk_comm_gatherv(self._c, a.ctypes.data_as(C.c_void_p), C.c_uint64(a.nbytes), 0,
                                                  recv.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), 0, dst),
                        "hawk_comm_gatherv")
        if self.rank != dst:
            return None
        return [recv[int(off[r]):int(off[r + 1])].view(a.dtype).reshape((-1,) + a.shape[1:]) for r in range(self.world)]

    def gather_table(self, tab, hap_offset: int, dst: int = 0):
        """hawk_table_gather: every rank's device-resident GuideTable to `dst`, device to device.  Returns
        (merged GuideTable or None, milliseconds on the stream)."""
        from .hapset import GuideTable
        merged, ms = C.c_void_p(), C.c_float(0)
        self._lib.check(self._L.hawk_table_gather(self._c, tab._t, int(hap_offset), dst, C.byref(merged), C.byref(ms)),
                        "hawk_table_gather")
        if self.rank != dst:
            return None, ms.value
        return GuideTable(tab._hs, merged, tab.guidelen, tab.pamlen, tab.right, tab._timing_struct), ms.value

    def close(self) -> None:
        if getattr(self, "_c", None):
            self._L.hawk_comm_destroy(self._c)
            self._c = None


def make_comm(backend: str = "rccl", device: Optional[int] = None):
    """Communicator for this process from RANK / WORLD_SIZE: "rccl" (product) or "tcp" (stand-in, no GPU exchange)."""
    rank, world, _ = env_rank_world()
    ctl = TcpComm(rank, world)
    if backend != "rccl":
        return ctl
    return RcclComm(ctl, device)


# ---------------------------------------------------------------------------------------------
# table exchange on host columns (any communicator with gatherv_bytes)
# ---------------------------------------------------------------------------------------------
def table_columns(tab) -> Dict[str, np.ndarray]:
    tab.download()
    cols = {k: np.ascontiguousarray(getattr(tab, k)) for k, _ in COLUMNS}
    cols["win"] = np.ascontiguousarray(tab.win.T)  # [rows, 5]
    return cols


def gather_tables(cols: Dict[str, np.ndarray], hap_offset: int, comm, dst: int = 0) -> Optional[Dict[str, np.ndarray]]:
    """Concatenate every rank's table columns on rank `dst`.  ``hap_offset`` is added to this rank's local haplotype
    indices (REF, index 0 on every rank, stays 0) so the merged table indexes the global haplotype list.  This is the
    host-array form of the exchange (tests, TcpComm); the device form is RcclComm.gather_table."""
    hap = cols["hap"].astype(np.int64)
    send = dict(cols, hap=np.where(hap == 0, 0, hap + hap_offset).astype(np.uint32))  # local haplotype 0 is REF on every rank
    out = {}
    for k, a in send.items():
        parts = comm.gatherv_bytes(a, dst)
        if parts is not None:
            out[k] = np.concatenate(parts, axis=0)
    return out if comm.rank == dst else None


def gather_collapsed(cols: Dict[str, np.ndarray], group_perm: np.ndarray, group_off: np.ndarray, is_ref_row: np.ndarray,
                     hap_offset: int, guidelen: int, pamlen: int, comm, dst: int = 0):
    """The exchange after a per-rank `GuideTable.collapse()` (SURVEY §8 f2 x §8e): instead of every row, each rank
    sends one representative row per group of report-identical rows plus the group's member haplotype ids
    (4 B per row instead of 74 B: C3 moves ~130 MB per rank instead of 2.1 GB).  Rank `dst` merges groups that
    occur on several ranks (the same guide carried by samples of different ranks; REF rows, present on every rank)
    by their full key - start, stop, strand, origin, the five core slices - and returns
    ({column: representative rows}, member_off[n_groups + 1], members[global haplotype ids, ascending per group]);
    other ranks return None.  `cols`: table_columns(tab); `is_ref_row`: whether each row is from the REF haplotype."""
    perm = np.asarray(group_perm, dtype=np.int64)
    off = np.asarray(group_off, dtype=np.int64)
    reps = perm[off[:-1]]
    hap = cols["hap"].astype(np.int64)
    ghap = np.where(hap == 0, 0, hap + hap_offset)  # local haplotype 0 is REF on every rank
    parts = {k: comm.gatherv_bytes(np.ascontiguousarray(cols[k][reps]), dst) for k in cols}
    origin = comm.gatherv_bytes(np.ascontiguousarray(is_ref_row[reps].astype(np.uint8)), dst)
    sizes = comm.gatherv_bytes(np.diff(off), dst)
    members = comm.gatherv_bytes(ghap[perm].astype(np.uint32), dst)
    if comm.rank != dst:
        return None
    rep = {k: np.concatenate(v) for k, v in parts.items()}
    return merge_groups(rep, np.concatenate(origin), np.concatenate(sizes), np.concatenate(members), guidelen, pamlen)


def merge_groups(rep: Dict[str, np.ndarray], origin: np.ndarray, sizes: np.ndarray, members: np.ndarray, guidelen: int, pamlen: int,
                 flank: Tuple[int, int] = (0, 0)):
    """Second-level merge of report groups that arrive from several producers (ranks, or the tiles of a region whose
    seam a group straddles): groups with the same full key - start, stop, strand, origin, compared window slice - become
    one, their member lists are united (ascending, duplicates such as REF = 0 dropped).  `rep["win"]` is [groups, 5]."""
    L = guidelen + pamlen
    up, down = flank
    strand = rep["strand"].astype(np.int64)
    fl = np.where(strand != 0, down, up)
    fr = np.where(strand != 0, up, down)
    width = (L + fl + fr).astype(np.uint64)
    mask = np.where(width >= 64, np.uint64(0xFFFFFFFFFFFFFFFF), (np.uint64(1) << width) - np.uint64(1))
    core = (rep["win"] >> (np.uint64(10) - fl.astype(np.uint64))[:, None]) & mask[:, None]
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
    keep[1:] = (gm[1:] != gm[:-1]) | (mm[1:] != mm[:-1])  # REF (0) arrives once per producer
    gm, mm = gm[keep], mm[keep]
    member_off = np.zeros(n_groups + 1, dtype=np.int64)
    np.add.at(member_off, gm + 1, 1)
    member_off = np.cumsum(member_off)
    out = {k: v[first] for k, v in rep.items()}
    out["hap"] = mm[member_off[:-1]].astype(np.uint32)  # a representative's haplotype: the group's first member
    return out, member_off, mm.astype(np.uint32)
