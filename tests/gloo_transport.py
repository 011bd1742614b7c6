"""A host-provided transport for `tokenflow_amd.comm.HipComm.from_hooks` carried by gloo (TEST INFRASTRUCTURE).

RCCL refuses two ranks on one device, so the multi-process GPU tests that share cuda:0 cannot exchange through it.
The library's exchange entry points accept a function table instead (tf_comm_init_hooks); this one stages every
message through host memory and moves it with torch.distributed's gloo backend.  Each callback synchronises the
device first (the library hands over device pointers whose producers are enqueued, not finished) and returns after
the received bytes are on the device, so whatever the library enqueues next on any stream sees them."""
import ctypes

import torch
import torch.distributed as dist

from tokenflow_amd.comm import HipComm

_hip = None


def _memcpy(dst: int, src: int, nbytes: int, kind: int):
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    if nbytes:
        rc = _hip.hipMemcpy(dst, src, nbytes, kind)
        assert rc == 0, f"hipMemcpy failed: {rc}"


def _to_host(ptr: int, nbytes: int) -> torch.Tensor:
    t = torch.empty(max(nbytes, 1), dtype=torch.uint8)
    _memcpy(t.data_ptr(), ptr, nbytes, 2)            # device -> host
    return t[:nbytes]


def _to_device(ptr: int, t: torch.Tensor):
    _memcpy(ptr, t.data_ptr(), t.numel(), 1)         # host -> device


def gloo_comm(rank: int, world: int) -> HipComm:
    def all_to_all_rows(user, send, recv, send_rows, recv_rows, row_bytes, stream):
        torch.cuda.synchronize()
        sb = [send_rows[p] * row_bytes for p in range(world)]
        rb = [recv_rows[p] * row_bytes for p in range(world)]
        src = _to_host(send, sum(sb))
        dst = torch.empty(max(sum(rb), 1), dtype=torch.uint8)[:sum(rb)]
        dist.all_to_all_single(dst, src, rb, sb)
        _to_device(recv, dst)
        return 0

    def allgather_rows(user, local, bank, rows, row_bytes, stream):
        torch.cuda.synchronize()
        nb = [rows[p] * row_bytes for p in range(world)]
        src = _to_host(local, nb[rank])
        parts = [torch.empty(max(n, 1), dtype=torch.uint8)[:n] for n in nb]
        opsl = []
        for p in range(world):
            if p == rank:
                parts[p].copy_(src)
            else:
                opsl += [dist.P2POp(dist.isend, src, p), dist.P2POp(dist.irecv, parts[p], p)]
        for r in (dist.batch_isend_irecv(opsl) if opsl else []):
            r.wait()
        _to_device(bank, torch.cat(parts))
        return 0

    def sendrecv(user, send, send_bytes, n_send, send_peer, recv, recv_bytes, n_recv, recv_peer, stream):
        torch.cuda.synchronize()
        outs = [_to_host(send[i], send_bytes[i]) for i in range(n_send)] if send_peer >= 0 else []
        ins = [torch.empty(max(recv_bytes[i], 1), dtype=torch.uint8)[:recv_bytes[i]] for i in range(n_recv)] \
            if recv_peer >= 0 else []
        opsl = [dist.P2POp(dist.isend, t, send_peer) for t in outs] + [dist.P2POp(dist.irecv, t, recv_peer) for t in ins]
        for r in (dist.batch_isend_irecv(opsl) if opsl else []):
            r.wait()
        for i, t in enumerate(ins):
            _to_device(recv[i], t)
        return 0

    return HipComm.from_hooks(rank, world, all_to_all_rows, allgather_rows, sendrecv)
