"""Compact pins for golden tensors (test infrastructure): a strided sample of
the flattened tensor plus three full-tensor moments in float64, so fixtures
stay a few hundred KiB while every element still contributes to the pin."""
import torch

STRIDE = 13


def digest(t: torch.Tensor, stride: int = STRIDE) -> dict:
    f = t.detach().to(torch.float32).flatten()
    d = f.double()
    return dict(shape=tuple(t.shape), dtype=str(t.dtype), stride=stride, sample=f[::stride].clone(),
                sum=float(d.sum()), abs_sum=float(d.abs().sum()), sq_sum=float((d * d).sum()))


def check(t: torch.Tensor, dg: dict, atol: float, what: str = "") -> float:
    """Assert `t` matches digest `dg`; returns the max abs deviation on the sample."""
    assert tuple(t.shape) == tuple(dg["shape"]), f"{what}: shape {tuple(t.shape)} != {dg['shape']}"
    f = t.detach().to(torch.float32).flatten().cpu()
    err = float((f[::dg["stride"]] - dg["sample"]).abs().max())
    assert err <= atol, f"{what}: sample max abs err {err} > {atol}"
    n = f.numel()
    d = f.double()
    # moments: tolerance scales with element count (errors may all share a sign)
    assert abs(float(d.sum()) - dg["sum"]) <= atol * n + 1e-9 * abs(dg["sum"]), f"{what}: sum"
    assert abs(float(d.abs().sum()) - dg["abs_sum"]) <= atol * n + 1e-9 * dg["abs_sum"], f"{what}: abs_sum"
    return err
