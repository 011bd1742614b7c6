#!/usr/bin/env python
"""Does torch.baddbmm(..., out=<batch-strided view>) write in place on this stack, and what does it cost against
F.linear + copy_ into the same view?  (Decision data for writing attn1.to_out straight into the halo-extended
attention-output state of a sharded rank, tokenflow_amd/hooks.py.)"""
import torch

dev = torch.device("cuda")
torch.set_grad_enabled(False)
for Kl, S, D in [(1, 4096, 320), (1, 1024, 640), (1, 256, 1280), (1, 64, 1280), (4, 4096, 320)]:
    for dt in (torch.bfloat16, torch.float16):
        x = torch.randn(3 * Kl, S, D, device=dev, dtype=dt)
        lin = torch.nn.Linear(D, D).to(dev).to(dt)
        kfo = torch.zeros(3, Kl + 1, S, D, device=dev, dtype=dt)
        dest = kfo[:, 1:]
        d3 = dest.reshape(3, Kl * S, D) if Kl == 1 else None
        assert d3 is None or d3.data_ptr() == dest.data_ptr()
        wT = lin.weight.t().unsqueeze(0).expand(3, D, D)
        bias = lin.bias.view(1, 1, D).expand(3, Kl * S, D)

        def a():
            y = torch.nn.functional.linear(x, lin.weight, lin.bias)
            dest.copy_(y.view(3, Kl, S, D))

        def b():
            torch.baddbmm(bias, x.view(3, Kl * S, D), wT, out=dest.view(3, Kl * S, D) if Kl == 1 else dest.flatten(1, 2))

        res = {}
        for name, fn in (("linear+copy", a), ("baddbmm out=view", b)):
            try:
                kfo.zero_()
                fn()
                got = kfo[:, 1:].clone()
                halo_untouched = not bool(kfo[:, 0].any())
                for _ in range(5):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[name] = (e0.elapsed_time(e1) / 50 * 1e3, got, halo_untouched)
            except Exception as e:  # noqa: BLE001
                res[name] = (float("nan"), None, str(e)[:100])
        ref = res["linear+copy"][1]
        line = f"Kl={Kl} S={S} D={D} {str(dt)[6:]}: "
        for name, (us, got, ok) in res.items():
            diff = float((got.float() - ref.float()).abs().max()) if got is not None else float("nan")
            line += f"{name} {us:.1f} us (max diff vs linear {diff:.2e}, halo slot untouched {ok})  "
        print(line, flush=True)
