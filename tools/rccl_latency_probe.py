#!/usr/bin/env python
"""Fixed cost of one RCCL exchange on this stack, measured with a communicator of ONE rank (the pool has one GPU):
the library's row all-to-all and all-gather entry points (grouped ncclSend / ncclRecv to self, ncclAllGather) on a
side stream, message sizes of a rank of 8 at the four cfg2 levels.  What it shows: host time per call and GPU time per
call of the collective's own launch + copy machinery with no wire at all -- the floor under every exchange of
tokenflow_amd/sharded.py, to put next to the 2.5..5 us same-size copies the wire-less rank microbenchmark uses."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenflow_amd.comm import HipComm  # noqa: E402


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    comm = HipComm(HipComm.unique_id(), 0, 1)
    side = torch.cuda.Stream()
    for name, S, D in (("level 0", 4096, 320), ("level 1", 1024, 640), ("level 2", 256, 1280), ("level 3", 64, 1280)):
        # a rank of 8 in the heads form: first all-to-all 6 slabs of [1, S, D], second 2 slabs; bank form: 6 slabs gathered
        for what, slabs in (("all-to-all #1", 6), ("all-to-all #2", 2), ("all-gather", 6)):
            send = torch.randn(1, slabs * S * D, device="cuda").bfloat16()
            recv = torch.empty_like(send)
            fn = (lambda: comm.allgather(send, recv, stream=side.cuda_stream)) if what == "all-gather" else \
                 (lambda: comm.all_to_all_rows(send, recv, stream=side.cuda_stream))
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            reps = 50
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            host = (time.perf_counter() - t0) / reps * 1e6
            e1.record(side)
            torch.cuda.synchronize()
            gpu = e0.elapsed_time(e1) / reps * 1e3
            print(f"{name} {what:14s} {send.numel() * 2 / 1e6:6.2f} MB  host {host:6.1f} us/call  GPU {gpu:6.1f} us/call "
                  f"(back to back on one stream)", flush=True)
    comm.close()


if __name__ == "__main__":
    main()
