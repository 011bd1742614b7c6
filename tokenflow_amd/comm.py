"""RCCL exchange steps through the library's own C ABI (`tf_comm_*`, include/tokenflow_hip.h) -- for hosts that do not
use torch.distributed.  `tokenflow_amd.sharded.FrameShard`, the host path this package ships and measures, runs the
same two exchange steps through torch.distributed (backend "nccl" = RCCL); this module is the ctypes binding of the
entry points SURVEY.md section 8b lists for a native host, and what the GPU tests drive them through.

One process per GPU:

    uid = HipComm.unique_id() on rank 0, handed to the other ranks by the host (file, socket, ...)
    comm = HipComm(uid, rank, world)              # binds the current device
    comm.allgather(local, bank)                   # K/V bank of the pivotal pass (tokenflow_utils.py:133-138)
    comm.allgather_rows(local, bank, rows)        # the same for runs of different lengths
    comm.all_to_all_rows(send, recv, send_rows, recv_rows)      # frames <-> heads re-sharding
    comm.sendrecv([piv_last, inv_last, ...], rank + 1, [piv_halo, inv_halo, ...], rank - 1)   # 331-333

All calls are asynchronous on the current torch stream.  The reference is single-process; nothing here replaces a line
of it."""
import ctypes
from typing import Optional, Sequence

import torch

from . import _lib

_DT = {torch.bfloat16: _lib.TF_BF16, torch.float16: _lib.TF_F16, torch.float32: _lib.TF_F32}


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _dev(*ts: torch.Tensor) -> int:
    for t in ts:
        if not t.is_cuda:
            raise _lib.TokenflowHipError("tokenflow_amd.comm: tensors must live on the GPU (there is no CPU path)")
        if not t.is_contiguous():
            raise ValueError("tokenflow_amd.comm: tensors must be contiguous")
    if len({t.dtype for t in ts}) != 1 or ts[0].dtype not in _DT:
        raise TypeError("tokenflow_amd.comm: one of bf16 / f16 / f32 for all tensors of a call")
    return _DT[ts[0].dtype]


_I64P = ctypes.POINTER(ctypes.c_int64)
_VPP = ctypes.POINTER(ctypes.c_void_p)
# the function table of a host-provided transport (tf_comm_hooks, include/tokenflow_hip.h): sizes in BYTES
A2A_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _I64P, _I64P, ctypes.c_int64,
                          ctypes.c_void_p)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _I64P, ctypes.c_int64,
                                ctypes.c_void_p)
SENDRECV_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, _VPP, _I64P, ctypes.c_int, ctypes.c_int, _VPP, _I64P,
                               ctypes.c_int, ctypes.c_int, ctypes.c_void_p)


class _Hooks(ctypes.Structure):
    _fields_ = [("all_to_all_rows", A2A_FN), ("allgather_rows", ALLGATHER_FN), ("sendrecv", SENDRECV_FN),
                ("user", ctypes.c_void_p)]


class HipComm:
    def __init__(self, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != 128:
            raise ValueError("unique_id: 128 bytes from HipComm.unique_id()")
        lib = _lib.load()
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(unique_id, 128)
        _lib.check(lib.tf_comm_init(buf, rank, world, ctypes.byref(h)), "tf_comm_init")
        self._h, self.rank, self.world = h, rank, world

    @classmethod
    def loopback(cls, rank: int, world: int, copies: bool = True, wire=None) -> "HipComm":
        """The wire-less stand-in (tf_comm_init_loopback): every exchange becomes device-to-device copies of the sizes a
        rank of a `world`-GPU run receives, out of this rank's own buffers.  For timing one rank's launch sequence on
        one GPU (tools/rank_step_microbench.py); the received data is meaningless for world > 1.
        `wire = (latency_us, gbps_per_link)`: the wire MODEL (tf_comm_loopback_wire) -- every exchange also holds its
        stream for latency + bytes on its busiest link / bandwidth, so that the schedule's overlap is executed."""
        self = cls.__new__(cls)
        h = ctypes.c_void_p()
        _lib.check(_lib.load().tf_comm_init_loopback(rank, world, ctypes.byref(h)), "tf_comm_init_loopback")
        self._h, self.rank, self.world = h, rank, world
        if not copies:      # the exchanges enqueue nothing at all: the stand-in copies out of the timing too
            _lib.check(_lib.load().tf_comm_loopback_copies(h, 0), "tf_comm_loopback_copies")
        if wire is not None:
            _lib.check(_lib.load().tf_comm_loopback_wire(h, float(wire[0]), float(wire[1])), "tf_comm_loopback_wire")
        return self

    @classmethod
    def from_hooks(cls, rank: int, world: int, all_to_all_rows, allgather_rows, sendrecv) -> "HipComm":
        """A host-provided transport (tf_comm_init_hooks): three Python callables with the signatures of A2A_FN /
        ALLGATHER_FN / SENDRECV_FN (device pointers, sizes in bytes, the stream; return 0).  The multi-process tests
        carry it over gloo so that W ranks can share one GPU."""
        self = cls.__new__(cls)
        self._fns = (A2A_FN(all_to_all_rows), ALLGATHER_FN(allgather_rows), SENDRECV_FN(sendrecv))   # kept alive
        self._hooks = _Hooks(self._fns[0], self._fns[1], self._fns[2], None)
        h = ctypes.c_void_p()
        _lib.check(_lib.load().tf_comm_init_hooks(ctypes.byref(self._hooks), rank, world, ctypes.byref(h)),
                   "tf_comm_init_hooks")
        self._h, self.rank, self.world = h, rank, world
        return self

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().tf_comm_unique_id(buf), "tf_comm_unique_id")
        return buf.raw

    def close(self):
        if self._h is not None:
            h, self._h = self._h, None
            _lib.check(_lib.load().tf_comm_destroy(h), "tf_comm_destroy")

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def allgather(self, local: torch.Tensor, bank: torch.Tensor, stream: Optional[int] = None):
        """bank [world * local.numel()] <- every rank's `local`, in rank order.  stream (all methods): the raw
        hipStream_t to enqueue on; None = the tensors' device's current torch stream."""
        dt = _dev(local, bank)
        if bank.numel() != self.world * local.numel():
            raise ValueError(f"bank has {bank.numel()} elements, expected {self.world} x {local.numel()}")
        _lib.check(_lib.load().tf_allgather_kv(self._h, local.data_ptr(), bank.data_ptr(), local.numel(), dt,
                                               _stream(local) if stream is None else stream), "tf_allgather_kv")
        return bank

    def allgather_rows(self, local: torch.Tensor, bank: torch.Tensor, rows: Sequence[int],
                       stream: Optional[int] = None):
        """bank [sum(rows), ...] <- rank p's `local` [rows[p], ...] for every p, in rank order (runs of different
        lengths: K keyframes over W ranks with K % W != 0)."""
        dt = _dev(local, bank)
        rows = list(rows)
        row = bank[0].numel()
        if len(rows) != self.world or sum(rows) != bank.shape[0] or local.shape[0] != rows[self.rank] or \
                (local.shape[0] and local[0].numel() != row):
            raise ValueError("allgather_rows: row counts do not match the buffers")
        rr = (ctypes.c_int64 * self.world)(*rows)
        _lib.check(_lib.load().tf_allgather_rows(self._h, local.data_ptr(), bank.data_ptr(), rr, row, dt,
                                                 _stream(local) if stream is None else stream), "tf_allgather_rows")
        return bank

    def all_to_all_rows(self, send: torch.Tensor, recv: torch.Tensor, send_rows: Optional[Sequence[int]] = None,
                        recv_rows: Optional[Sequence[int]] = None, stream: Optional[int] = None):
        """send [sum(send_rows), ...] -> rows send_rows[p] to peer p; recv [sum(recv_rows), ...] <- recv_rows[p] rows
        from peer p (None = equal parts), like dist.all_to_all_single over dim 0."""
        dt = _dev(send, recv)
        W = self.world
        send_rows = list(send_rows) if send_rows is not None else [send.shape[0] // W] * W
        recv_rows = list(recv_rows) if recv_rows is not None else [recv.shape[0] // W] * W
        row = send[0].numel() if send.shape[0] else recv[0].numel()
        if len(send_rows) != W or len(recv_rows) != W or sum(send_rows) != send.shape[0] or \
                sum(recv_rows) != recv.shape[0] or (recv.shape[0] and recv[0].numel() != row):
            raise ValueError("all_to_all_rows: row counts do not match the buffers")
        sr, rr = (ctypes.c_int64 * W)(*send_rows), (ctypes.c_int64 * W)(*recv_rows)
        _lib.check(_lib.load().tf_all_to_all_rows(self._h, send.data_ptr(), recv.data_ptr(), sr, rr, row, dt,
                                                  _stream(send) if stream is None else stream), "tf_all_to_all_rows")
        return recv

    def sendrecv(self, send: Sequence[torch.Tensor], send_peer: int, recv: Sequence[torch.Tensor], recv_peer: int,
                 stream: Optional[int] = None):
        """`send` tensors to send_peer while `recv` tensors arrive from recv_peer, one grouped exchange; a peer of -1
        skips that direction."""
        ts = list(send if send_peer >= 0 else []) + list(recv if recv_peer >= 0 else [])
        if not ts:
            return
        dt = _dev(*ts)
        ns, nr = (len(send) if send_peer >= 0 else 0), (len(recv) if recv_peer >= 0 else 0)
        sp = (ctypes.c_void_p * max(ns, 1))(*[t.data_ptr() for t in send[:ns]])
        se = (ctypes.c_int64 * max(ns, 1))(*[t.numel() for t in send[:ns]])
        rp = (ctypes.c_void_p * max(nr, 1))(*[t.data_ptr() for t in recv[:nr]])
        re_ = (ctypes.c_int64 * max(nr, 1))(*[t.numel() for t in recv[:nr]])
        _lib.check(_lib.load().tf_sendrecv_pivot(self._h, sp, se, ns, send_peer, rp, re_, nr, recv_peer, dt,
                                                 _stream(ts[0]) if stream is None else stream), "tf_sendrecv_pivot")


def bootstrap(rank: int, world: int, n: int = 1, group=None, make=None):
    """Create `n` communicators on every rank of a torch.distributed group that is used as the control plane only
    (gloo: unique ids out, one agreement flag back; no tensor of the path touches it).  Every rank returns the same
    kind of result: (list of n communicators, None), or (None, reason) when ANY rank failed to create one -- no rank
    is left holding half a set.  A pre-flight round first agrees that EVERY rank can load RCCL and reach its device
    (the failures that are local to one rank: a missing library, a bad device); only then do the ranks enter
    ncclCommInitRank, which is itself collective -- a rank that dies INSIDE it can still leave the others waiting
    there (RCCL's own timeout applies); what the agreement flags exclude is a rank that never enters it.
    make(unique_id, rank, world) -> communicator; default `HipComm` (RCCL through the C ABI)."""
    import torch.distributed as dist
    make = make or HipComm
    comms, why = [], None
    pre = None
    if make is HipComm:
        try:
            # dlopen of librccl + every symbol, on THIS rank; a loader-only call: ncclGetUniqueId would start a
            # bootstrap root listener (thread + socket) on every rank for nothing
            _lib.check(_lib.load().tf_comm_available(), "tf_comm_available")
            torch.cuda.current_device()
        except Exception as e:  # noqa: BLE001
            pre = str(e)
        ok = torch.tensor([0 if pre else 1])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if not int(ok.item()):
            reasons = [None] * world
            dist.all_gather_object(reasons, pre, group=group)
            return None, next((r for r in reasons if r), "a rank failed its RCCL pre-flight")
    for _ in range(n):
        uid, err = [None], None
        if rank == 0:
            try:
                uid[0] = HipComm.unique_id() if make is HipComm else b"\0" * 128
            except Exception as e:  # noqa: BLE001  (RCCL not loadable ...)
                err = str(e)
        dist.broadcast_object_list(uid, src=0, group=group)
        c = None
        if uid[0] is not None:
            try:
                c = make(uid[0], rank, world)
            except Exception as e:  # noqa: BLE001
                err = str(e)
        flag = torch.tensor([1 if c is not None else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)       # every rank takes the same branch
        if not int(flag.item()):
            reasons = [None] * world
            dist.all_gather_object(reasons, err, group=group)
            why = next((r for r in reasons if r), "a rank failed to create its communicator")
            for x in comms + ([c] if c is not None else []):
                x.close()
            return None, why
        comms.append(c)
    return comms, None
