"""The drop-in seam against the reference's OWN driver code.

tests/golden/driver.pt = `TokenFlow.init_method / denoise_step / batched_denoise_step` of run_tokenflow_pnp.py
(195-239) and run_tokenflow_sdedit.py (154-193), cut out of the reference's syntax tree and executed unchanged over
the VERBATIM reference hooks on CPU fp32 (oracle/make_golden.py::gen_driver).  Here the same driver methods run over
THIS repository's `tokenflow_utils`:

* on the CPU with the oracle-backed ops (`-m "not gpu"`): same hook-call trace, same `pivotal_idx` draws, every block
  output / noise prediction / latent within fp32 re-association distance of the golden;
* the restated driver of tests/driver_seam.py (what the GPU box runs, where /root/reference does not exist) is proven
  equal to the verbatim cut: same trace, same draws, same bits (build container only);
* on the GPU (`-m gpu`) over the HIP kernels.
"""
import warnings

import pytest
import torch

import tokenflow_utils as tfu
from oracle import ref_loader
from oracle.golden_util import check
from tests import driver_seam as ds
from tests.conftest import load_golden
from tests.fake_ops import FakeOps
from tokenflow_amd import hooks

KINDS = ("pnp", "sdedit")


def _methods(kind, log, prefer_verbatim=True):
    """The driver methods over this repository's drop-in module: the verbatim cut where the reference is mounted,
    the restatement elsewhere."""
    ns = ds.traced(tfu, log)
    if prefer_verbatim and ref_loader.available():
        from oracle import driver_cut
        return driver_cut.load_reference_driver(kind, ns), "verbatim"
    return ds.restated_driver(kind, ns), "restated"


def _run(kind, methods, tmp_path, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")       # "CUDA is not available ... Disabling autocast" on the CPU
        return ds.run_driver(kind, methods, str(tmp_path / f"latents_{kind}"), **kw)


def _compare(rec, gold, block_tol, unet_tol, x_tol, what):
    """Structure, draws and call counts exactly; tensors against the golden digests.  `*_tol` are fractions of the
    golden sample's range.  Returns the worst fractions seen."""
    worst = dict(block=0.0, unet=0.0, x=0.0)
    assert len(rec["steps"]) == len(gold["steps"])
    assert abs(rec["weights_checksum"] - gold["weights_checksum"]) <= 1e-6 * gold["weights_checksum"], "RNG drift"
    for s, g in zip(rec["steps"], gold["steps"]):
        assert s["t"] == g["t"] and len(s["calls"]) == len(g["calls"])
        for ci, (c, gc_) in enumerate(zip(s["calls"], g["calls"])):
            assert torch.equal(c["indices"], gc_["indices"]), f"{what} t={s['t']} call {ci}: frame indices differ"
            assert len(c["blocks"]) == len(gc_["blocks"]) == 16
            for bi, (b, gb) in enumerate(zip(c["blocks"], gc_["blocks"])):
                rng = float(gb["sample"].abs().max())
                err = check(b, gb, block_tol * rng, f"{what} t={s['t']} call {ci} block {bi}")
                worst["block"] = max(worst["block"], err / rng)
            rng = float(gc_["unet"]["sample"].abs().max())
            worst["unet"] = max(worst["unet"], check(c["unet"], gc_["unet"], unet_tol * rng,
                                                     f"{what} t={s['t']} call {ci} noise_pred") / rng)
        rng = float(g["x"]["sample"].abs().max())
        worst["x"] = max(worst["x"], check(s["x"], g["x"], x_tol * rng, f"{what} t={s['t']} latents") / rng)
    return worst


@pytest.mark.parametrize("kind", KINDS)
def test_driver_over_dropin_hooks_matches_reference_golden_cpu(kind, tmp_path, monkeypatch):
    """Driver methods (verbatim here) over the drop-in hooks with oracle-backed fp32 ops == verbatim driver over
    verbatim hooks: identical trace and draws, tensors within 2e-5 of range (fp32 re-association only)."""
    monkeypatch.setattr(hooks, "ops", FakeOps(round16=False))
    gold = load_golden("driver.pt")[kind]
    log = []
    methods, which = _methods(kind, log)
    rec = _run(kind, methods, tmp_path, keep_tensors=True)
    assert log == gold["trace"], f"hook-call trace differs from the reference driver's ({which} driver)"
    n_load = sum(1 for e in log if e[0] == "load_source_latents_t")
    C = ds.CFG["F"] // ds.CFG["batch_size"]
    assert n_load == ds.CFG["n_steps"] * (C + 1)                      # run_tokenflow_pnp.py:198 once per UNet call
    worst = _compare(rec, gold, 2e-5, 2e-5, 2e-5, f"{kind}/{which}")
    print(f"driver seam cpu {kind} ({which}): worst fraction of range {worst}")


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not mounted")
@pytest.mark.parametrize("kind", KINDS)
def test_restated_driver_equals_the_verbatim_cut(kind, tmp_path, monkeypatch):
    """The restatement the GPU box runs IS the reference driver as far as the hooks can tell: same calls in the same
    order with the same argument types and values, same `pivotal_idx`, bit-equal tensors."""
    monkeypatch.setattr(hooks, "ops", FakeOps(round16=False))
    log_v, log_r = [], []
    rec_v = _run(kind, _methods(kind, log_v)[0], tmp_path, keep_tensors=True)
    rec_r = _run(kind, _methods(kind, log_r, prefer_verbatim=False)[0], tmp_path, keep_tensors=True)
    assert log_v == log_r
    for sv, sr in zip(rec_v["steps"], rec_r["steps"]):
        assert torch.equal(sv["x"], sr["x"])
        for cv, cr in zip(sv["calls"], sr["calls"]):
            assert torch.equal(cv["indices"], cr["indices"]) and torch.equal(cv["unet"], cr["unet"])
            assert all(torch.equal(a, b) for a, b in zip(cv["blocks"], cr["blocks"]))


def _tensor_compare(rec, ref, tol, what):
    """`rec` against another full-tensor record of the same run shape: worst |difference| / range per kind;
    `tol` = {kind: bound}."""
    worst = dict(block=0.0, unet=0.0, x=0.0)
    for s, g in zip(rec["steps"], ref["steps"]):
        pairs = [("x", s["x"], g["x"])]
        for c, gc_ in zip(s["calls"], g["calls"]):
            assert torch.equal(c["indices"], gc_["indices"])
            pairs += [("unet", c["unet"], gc_["unet"])] + [("block", a, b) for a, b in zip(c["blocks"], gc_["blocks"])]
        for k, a, b in pairs:
            worst[k] = max(worst[k], float((a - b).abs().max() / b.abs().max()))
    assert all(worst[k] <= tol[k] for k in worst), f"{what}: {worst} > {tol}"
    return worst


@pytest.mark.gpu
@pytest.mark.parametrize("model_autocast", [False, True])
@pytest.mark.parametrize("kind", KINDS)
def test_driver_over_hip_hooks_matches_reference_golden(kind, model_autocast, tmp_path, monkeypatch):
    """The driver methods over the HIP hook path on the GPU against the reference-generated golden: the hook-call
    trace and the `pivotal_idx` draws exactly; every transformer block output of every UNet call, every noise
    prediction and the latents after each step as fractions of the golden tensor's range:

    model_autocast=False -- the stand-in UNet's own layers stay fp32 (they run outside the decorator's autocast), so
      the only 16-bit roundings are the kernels' boundary: fp32 q / k / v / pivots rounded to bf16, bf16 attention
      output.  (a) against the pure-fp32 golden: 5e-3 of range -- the bf16 boundary itself: the CPU emulation of the
      contract (`FakeOps(round16=True)`) sits at 2.3e-3 of range per block, 1.3e-3 on the noise prediction (a bf16
      half-ulp is 2e-3 of a value); measured on MI355X: 2.2e-3 / 1.3e-3 / 1.7e-3 (block / noise prediction / latents);
      (b) against the SAME driver run on the CPU over the oracle-backed ops with that rounding contract: both sides
      round at the same points, but the fp32 layers in front of a rounding point differ in their last bits between
      hipBLASLt and the CPU, which flips individual bf16 roundings (one ulp = 4e-3 of that element) -- bound 3e-3 of
      range per block, **1.5e-3 on the noise prediction and the latents** (measured 1.4e-3 / 7.6e-4 / 7.9e-4).
    model_autocast=True -- exactly what the decorator of `batched_denoise_step` asks for on a GPU: fp16 autocast of
      every Linear / conv of the model (the reference's operating mode); the kernels then compute
      in f16.  Bound against the fp32 golden: 6e-3 of range (the fp16 layers of the model over three steps; measured
      3.3e-3 / 2.1e-3 / 1.9e-3).
    north_star's "1e-3" is a per-token bound on the attention kernel against the oracle on identical inputs
    (tests/test_fullsize_gpu.py, bench.py `parity`); a whole UNet pass through a bf16 boundary cannot meet it whatever
    the kernel (see (a): the emulated contract alone is at 1.3e-3 of range on the noise prediction).
    """
    gold = load_golden("driver.pt")[kind]
    log = []
    methods, which = _methods(kind, log)
    rec = _run(kind, methods, tmp_path, device="cuda", model_autocast=model_autocast, keep_tensors=True)
    assert log == gold["trace"], f"hook-call trace differs from the reference driver's ({which} driver)"
    tol = 6e-3 if model_autocast else 5e-3
    worst = _compare(rec, gold, tol, tol, tol, f"{kind}/{which}/autocast={model_autocast}")
    print(f"driver seam gpu {kind} ({which}, model autocast {model_autocast}) vs fp32 golden: worst fraction of "
          f"range {worst}")
    if not model_autocast:
        monkeypatch.setattr(hooks, "ops", FakeOps(round16=True))
        emu = _run(kind, _methods(kind, [])[0], tmp_path, keep_tensors=True)
        w2 = _tensor_compare(rec, emu, dict(block=3e-3, unet=1.5e-3, x=1.5e-3), f"{kind} vs rounding-matched CPU run")
        print(f"driver seam gpu {kind} vs rounding-matched CPU hooks: worst fraction of range {w2}")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_driver_fp32_model_with_f16_boundary_within_1e3_of_range(kind, tmp_path, monkeypatch):
    """The same run with the op boundary rounding fp32 tensors to f16 instead of bf16 (`TOKENFLOW_FP32_AS=f16`: 11
    significand bits, the reference's own GPU dtype): every block output, every noise prediction and the latents of
    all three steps within **1e-3 of range** of the reference-generated fp32 golden."""
    from tokenflow_amd import ops
    monkeypatch.setattr(ops, "FP32_AS", torch.float16)
    gold = load_golden("driver.pt")[kind]
    log = []
    methods, which = _methods(kind, log)
    rec = _run(kind, methods, tmp_path, device="cuda", model_autocast=False, keep_tensors=True)
    assert log == gold["trace"]
    worst = _compare(rec, gold, 1e-3, 1e-3, 1e-3, f"{kind}/{which}/f16 boundary")
    print(f"driver seam gpu {kind} ({which}, fp32 model, f16 op boundary) vs fp32 golden: worst fraction of range {worst}")
