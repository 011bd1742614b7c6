"""Host logic of the drop-in hook layer (tokenflow_utils.py / tokenflow_amd/hooks.py) on CPU:
the HIP ops are replaced by the oracle-backed FakeOps, so what is tested is everything
around the kernels -- installation, per-module state, schedules, keyframe order, dtype
promotion -- against the verbatim reference's golden outputs."""
import pytest
import torch

import tokenflow_utils as tfu
from oracle import golden_cases as gc
from oracle.golden_util import check
from tests import fake_diffusers as fd
from tests.fake_ops import FakeOps
from tokenflow_amd import hooks


@pytest.fixture
def fake_ops(monkeypatch):
    f = FakeOps()
    monkeypatch.setattr(hooks, "ops", f)
    return f


def _pipe():
    cfg = gc.BLOCKS_CFG
    torch.manual_seed(cfg["seed"])
    return fd.FakePipeline(dims=cfg["dims"], heads=cfg["heads"], cross_dim=cfg["cross_dim"]).eval()


def test_api_surface_matches_reference():
    # tokenflow_utils.py:7,13,20,43,49,106,216,296,432 + the run scripts' `from util import ...`
    import inspect
    import util
    sigs = {
        "register_pivotal": ["diffusion_model", "is_pivotal"], "register_batch_idx": ["diffusion_model", "batch_idx"],
        "register_time": ["model", "t"], "load_source_latents_t": ["t", "latents_path"],
        "register_conv_injection": ["model", "injection_schedule"],
        "register_extended_attention_pnp": ["model", "injection_schedule"], "register_extended_attention": ["model"],
        "make_tokenflow_attention_block": ["block_class"], "set_tokenflow": ["model"]}
    for name, params in sigs.items():
        assert list(inspect.signature(getattr(tfu, name)).parameters) == params, name
    for name in ("save_video", "seed_everything", "isinstance_str", "batch_cosine_sim"):
        assert callable(getattr(util, name))


def test_blocks_match_reference_golden(fake_ops, golden_blocks):
    cfg = gc.BLOCKS_CFG
    pipe = _pipe()
    assert gc.checksum(*pipe.parameters()) == golden_blocks["weights_checksum"], "RNG drift"
    tfu.register_extended_attention_pnp(pipe, torch.tensor(cfg["schedule"]))     # tensor schedule, as the driver passes
    tfu.register_conv_injection(pipe, torch.tensor(cfg["conv_schedule"]))
    tfu.set_tokenflow(pipe.unet)
    blocks = [b for _, b in pipe.unet.transformer_blocks_in_order()]
    for t in cfg["timesteps"]:
        tfu.register_time(pipe, t)
        inp = gc.blocks_inputs(t)
        run = golden_blocks["runs"][t]
        fake_ops.calls.clear()
        with torch.no_grad():
            tfu.register_pivotal(pipe, True)
            for i, (blk, x) in enumerate(zip(blocks, inp["pivotal"])):
                check(blk(x, encoder_hidden_states=inp["enc"]), run["pivotal"][i], 3e-5, f"t{t}/pivotal/{i}")
            tfu.register_pivotal(pipe, False)
            for c in range(cfg["n_chunks"]):
                tfu.register_batch_idx(pipe, c)
                for i, (blk, x) in enumerate(zip(blocks, inp["chunks"][c])):
                    check(blk(x, encoder_hidden_states=inp["enc_n"]), run["chunks"][c][i], 3e-5, f"t{t}/chunk{c}/{i}")
                    # the state attribute the reference leaves after a propagation pass (361-363): the selected
                    # keyframe outputs [3, len(batch_idxs), S, D] in the order [i, i-1]
                    check(blk.attn_output, run["chunk_attn_state"][c][i], 3e-5, f"t{t}/chunk{c}/{i}/attn_output")
                    assert blk.attn_output.shape[1] == (1 if c == 0 else 2)
            y = pipe.unet.up_blocks[1].resnets[1](inp["res_x"], inp["res_temb"])
            check(y, run["resnet"], 1e-6, f"t{t}/resnet")
        # injection fires on exactly the 8 decoder blocks and only for scheduled timesteps (206-214)
        n_inj = sum(1 for c in fake_ops.calls if c[0] == "ext_attn" and c[2])
        assert n_inj == (8 if t in cfg["schedule"] else 0)
        assert sum(1 for c in fake_ops.calls if c[0] == "inject_copy_") == (1 if t in cfg["conv_schedule"] else 0)
        # keyframe order [i, i-1] (331-333)
        kfs = [c[2] for c in fake_ops.calls if c[0] == "nn_search"]
        assert kfs[:16] == [(0,)] * 16 and kfs[16:32] == [(1, 0)] * 16 and kfs[32:48] == [(2, 1)] * 16


def test_sdedit_variant_never_injects(fake_ops):
    pipe = _pipe()
    tfu.register_extended_attention(pipe)
    tfu.set_tokenflow(pipe.unet)
    tfu.register_time(pipe, 1000)              # t == 1000 forces injection only in the pnp variant
    tfu.register_pivotal(pipe, True)
    blk = pipe.unet.up_blocks[3].attentions[0].transformer_blocks[0]
    with torch.no_grad():
        blk(torch.randn(6, 16, 80), encoder_hidden_states=torch.randn(6, 7, 32))
    assert [c for c in fake_ops.calls if c[0] == "ext_attn"] == [("ext_attn", (6, 16, 80), False)]


def test_t_1000_forces_injection_everywhere(fake_ops):
    pipe = _pipe()
    tfu.register_extended_attention_pnp(pipe, [])
    tfu.set_tokenflow(pipe.unet)
    tfu.register_time(pipe, 1000)
    tfu.register_pivotal(pipe, True)
    blk = pipe.unet.down_blocks[0].attentions[0].transformer_blocks[0]    # not one of the 8 injected blocks
    with torch.no_grad():
        blk(torch.randn(6, 16, 80), encoder_hidden_states=torch.randn(6, 7, 32))
    assert fake_ops.calls[0] == ("ext_attn", (6, 16, 80), True)


def test_state_attributes_and_class_swap(fake_ops):
    pipe = _pipe()
    tfu.register_extended_attention_pnp(pipe, [5])
    tfu.set_tokenflow(pipe.unet)
    blk = pipe.unet.mid_block.attentions[0].transformer_blocks[0]
    assert type(blk).__name__ == "TokenFlowBlock" and tfu.isinstance_str(blk, "BasicTransformerBlock")
    assert blk.attn1.injection_schedule == [] and "forward" in blk.attn1.__dict__
    assert pipe.unet.up_blocks[2].attentions[1].transformer_blocks[0].attn1.injection_schedule == [5]
    tfu.register_time(pipe, 5)
    tfu.register_pivotal(pipe, True)
    tfu.register_batch_idx(pipe, 3)
    assert blk.pivotal_pass is True and blk.batch_idx == 3 and blk.attn1.t == 5 and blk.attn2.t == 5
    assert pipe.unet.up_blocks[1].resnets[1].t == 5                        # tokenflow_utils.py:21-22
    with torch.no_grad():
        x = torch.randn(6, 8, 320)
        blk(x, encoder_hidden_states=torch.randn(6, 7, 32))
    assert blk.pivot_hidden_states.shape == (3, 2, 8, 320) and blk.kf_attn_output.shape == (6, 8, 320)
    assert blk.attn_output is blk.kf_attn_output


def test_propagation_dtype_promotion(fake_ops):
    """chunk 0 keeps the stream dtype, chunks >= 1 promote to fp32 (tokenflow_utils.py:385-390)."""
    blk = fd.BasicTransformerBlock(80, 2).eval().to(torch.bfloat16)
    holder = torch.nn.Module()
    holder.blk = blk
    tfu.set_tokenflow(holder)
    blk.pivotal_pass = False
    K, n, S, D = 2, 2, 8, 80
    blk.pivot_hidden_states = torch.randn(3, K, S, D)
    blk._tf_pivots = blk.pivot_hidden_states[0].contiguous()
    blk._tf_pivot_inv_norm = 1.0 / blk._tf_pivots.norm(dim=-1)
    blk.kf_attn_output = torch.randn(3 * K, S, D).bfloat16()
    seen = {}
    orig = fake_ops.gather_blend

    def spy(*a):
        seen["dtype"] = a[-1]
        return orig(*a)
    fake_ops.gather_blend = spy
    enc = torch.randn(3 * n, 7, 32).bfloat16()
    with torch.no_grad():
        blk.batch_idx = 0
        blk(torch.randn(3 * n, S, D).bfloat16(), encoder_hidden_states=enc)
        assert seen["dtype"] == torch.bfloat16
        blk.batch_idx = 1
        with pytest.raises(RuntimeError):      # fp32 stream into bf16 weights: same failure mode as the reference
            blk(torch.randn(3 * n, S, D).bfloat16(), encoder_hidden_states=enc)
        assert seen["dtype"] == torch.float32


def test_adazero_block_matches_reference_golden(fake_ops):
    """AdaLayerNormZero blocks: the propagation pass gates the selected keyframe outputs with gate_msa before
    the gather, as the reference does (tokenflow_utils.py:362-366) -- pinned to the verbatim reference."""
    from tests.conftest import load_golden
    g = load_golden("adazero.pt")
    blk = gc.adazero_block()
    assert gc.checksum(*blk.parameters()) == g["weights_checksum"], "RNG drift"
    holder = torch.nn.Module()
    holder.unet = torch.nn.Module()
    holder.unet.blk = blk
    blk.attn1.forward = hooks._make_sa_forward(blk.attn1, pnp=True)
    hooks._set_schedule(blk.attn1, [])
    blk.attn1.t = 7
    tfu.set_tokenflow(holder)
    inp = gc.adazero_inputs()
    with torch.no_grad():
        tfu.register_pivotal(holder, True)
        check(blk(inp["pivotal"], encoder_hidden_states=inp["enc"], timestep=inp["timestep"]), g["pivotal"], 3e-5,
              "adazero/pivotal")
        tfu.register_pivotal(holder, False)
        for c in range(gc.ADAZERO_CFG["K"]):
            tfu.register_batch_idx(holder, c)
            check(blk(inp["chunks"][c], encoder_hidden_states=inp["enc_n"], timestep=inp["timestep"]),
                  g["chunks"][c], 3e-5, f"adazero/chunk{c}")


@pytest.mark.parametrize("first", [0, 1])
def test_multi_chunk_pass_equals_per_chunk_passes(fake_ops, first):
    """Extension: `batch_idx` may be a run of consecutive chunks carried by ONE pass (frames chunk-major inside
    each branch).  The block output must equal the per-chunk passes of the reference API, row for row."""
    cfg = gc.BLOCKS_CFG
    pipe = _pipe()
    tfu.register_extended_attention_pnp(pipe, [])
    tfu.set_tokenflow(pipe.unet)
    tfu.register_time(pipe, 1)
    K, n, S, D = 4, 2, 16, cfg["dims"][0]
    blk = pipe.unet.down_blocks[0].attentions[0].transformer_blocks[0]
    g = torch.Generator().manual_seed(5)
    enc, enc_n = torch.randn(3 * K, 7, 32, generator=g), torch.randn(3 * n, 7, 32, generator=g)
    chunks = [torch.randn(3 * n, S, D, generator=g) for _ in range(K)]
    with torch.no_grad():
        tfu.register_pivotal(pipe, True)
        blk(torch.randn(3 * K, S, D, generator=g), encoder_hidden_states=enc)
        tfu.register_pivotal(pipe, False)
        ref = []
        for c in range(first, K):
            tfu.register_batch_idx(pipe, c)
            ref.append(blk(chunks[c], encoder_hidden_states=enc_n).view(3, n, S, D))
        C = K - first
        x_all = torch.stack([chunks[c].view(3, n, S, D) for c in range(first, K)], dim=1).reshape(3 * C * n, S, D)
        tfu.register_batch_idx(pipe, range(first, K))
        got = blk(x_all, encoder_hidden_states=enc_n.view(3, n, 7, 32).repeat(1, C, 1, 1).reshape(3 * C * n, 7, 32))
    want = torch.stack(ref, dim=1).reshape(3 * C * n, S, D)
    assert got.dtype == want.dtype and torch.allclose(got, want, atol=1e-6, rtol=0)
    with pytest.raises(ValueError):
        tfu.register_batch_idx(pipe, [0, 2])
        blk(x_all, encoder_hidden_states=enc_n)


def test_cfg1_harness_dry_run_on_cpu(monkeypatch):
    """The config-1 end-to-end harness of tests/test_baseline_configs_gpu.py (public installers, the driver's call
    sequence, per-op oracle checks, block outputs against `oracle.block_forward`) run on the CPU with the
    oracle-backed ops standing in for the HIP library: validates the harness itself and the host logic at the
    real SD1.5 widths, one step."""
    from tests import test_baseline_configs_gpu as e2e
    e2e.run_cfg1(FakeOps(round16=True), torch.device("cpu"), monkeypatch, steps=[0], full_steps={0})


def test_fused_qkv_equals_three_projections(fake_ops, monkeypatch):
    """Row f2: q, k, v as column slabs of one GEMM against the cached concatenated weight -- on CPU the fused
    path is disabled (x.is_cuda), so emulate it by calling the helper's math directly: the slabs must equal the
    three Linear outputs bit for bit (same dot products, same fp32 accumulation order per output column)."""
    attn = fd.Attention(80, 2).eval()
    x = torch.randn(6, 16, 80)
    wcat = torch.cat([attn.to_q.weight, attn.to_k.weight, attn.to_v.weight], 0)
    qkv = torch.nn.functional.linear(x, wcat)
    for i, lin in enumerate((attn.to_q, attn.to_k, attn.to_v)):
        assert torch.allclose(qkv[..., 80 * i:80 * (i + 1)], lin(x), atol=1e-6, rtol=0)
    assert hooks._fused_qkv(attn, x) is None            # CPU tensors: the caller issues the three projections


def test_latents_cache_is_bounded(tmp_path):
    for t in (981, 961, 941, 921):
        torch.save(torch.full((2, 4, 8, 8), float(t)), tmp_path / f"noisy_latents_{t}.pt")
    for t in (981, 961, 941, 921, 981):
        assert float(tfu.load_source_latents_t(t, str(tmp_path))[0, 0, 0, 0]) == t
    assert len(hooks._latents_cache) <= hooks._LATENTS_CACHE_ENTRIES


def test_launcher_shadows_same_named_modules_in_the_script_directory(tmp_path):
    """`python script.py` resolves `tokenflow_utils` / `util` to the script's OWN directory (sys.path[0] beats
    PYTHONPATH): inside a reference checkout the drop-in would silently not be used.  The launcher must win."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "tokenflow_utils.py").write_text("def register_pivotal(m, p):\n    raise SystemExit('decoy used')\n")
    (tmp_path / "util.py").write_text("def seed_everything(s):\n    raise SystemExit('decoy used')\n"
                                      "save_video = None\n")
    (tmp_path / "local_helper.py").write_text("VALUE = 41\n")
    (tmp_path / "run_stub.py").write_text(
        "import sys\n"
        "from tokenflow_utils import *\n"
        "from util import save_video, seed_everything\n"
        "import local_helper\n"
        "if __name__ == '__main__':\n"
        "    print('ARGV', sys.argv[1:])\n"
        "    print('MOD', register_pivotal.__module__, seed_everything.__module__, local_helper.VALUE)\n")
    env = dict(os.environ, PYTHONPATH=root, TOKENFLOW_QUIET="1")
    # the documented-but-wrong way: PYTHONPATH alone -> the decoy is imported
    plain = subprocess.run([sys.executable, str(tmp_path / "run_stub.py")], cwd=tmp_path, env=env,
                           capture_output=True, text=True)
    assert "MOD tokenflow_utils util" in plain.stdout
    # the launcher: this repository's modules, the script's other local imports intact, argv passed through
    res = subprocess.run([sys.executable, "-m", "tokenflow_amd.run", str(tmp_path / "run_stub.py"), "--x", "1"],
                         cwd=tmp_path, env=env, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert "ARGV ['--x', '1']" in res.stdout
    assert "MOD tokenflow_amd.hooks util 41" in res.stdout


def test_hook_installation_announces_the_hip_path(capsys, monkeypatch):
    monkeypatch.setattr(hooks, "_announced", False)
    monkeypatch.delenv("TOKENFLOW_QUIET", raising=False)
    tfu.set_tokenflow(_pipe().unet)
    assert "HIP hook path active" in capsys.readouterr().err
    tfu.set_tokenflow(_pipe().unet)
    assert capsys.readouterr().err == ""            # once per process


def test_load_source_latents_cache(tmp_path):
    x = torch.randn(4, 4, 8, 8)
    torch.save(x, tmp_path / "noisy_latents_981.pt")
    a = tfu.load_source_latents_t(981, str(tmp_path))
    b = tfu.load_source_latents_t(981, str(tmp_path))
    assert torch.equal(a, x) and a is b
    with pytest.raises(AssertionError, match="Missing latents"):
        tfu.load_source_latents_t(1, str(tmp_path))


def test_ops_fail_loudly_on_cpu():
    """No CPU fallback: the real ops refuse CPU tensors."""
    from tokenflow_amd import ops
    from tokenflow_amd._lib import TokenflowHipError
    x = torch.zeros(3, 8, 80, dtype=torch.bfloat16)
    with pytest.raises(TokenflowHipError, match="no CPU fallback"):
        ops.ext_attn(x, x, x, 2, 1.0, False)
    with pytest.raises(TokenflowHipError):
        ops.inject_copy_(torch.zeros(3, 16))


def test_library_exports_every_declared_symbol():
    """Every function declared in include/tokenflow_hip.h is exported by the built library."""
    import os
    import re
    from tokenflow_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "tokenflow_hip.h")).read()
    declared = set(re.findall(r"\b(tf_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.tf_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define TF_ABI_VERSION (\d+)", hdr).group(1))
    # ... and NOTHING else with the library's prefix (built with -fvisibility=hidden: internal C++ helpers stay inside)
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    syms = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in syms.splitlines() if " T " in line or " W " in line}
    assert {n for n in exported if "tf_" in n} == declared, {n for n in exported if "tf_" in n} ^ declared


def test_loopback_wire_model_is_a_loopback_only_switch():
    """tf_comm_loopback_wire (ABI 7) configures the wire model of a LOOPBACK communicator (no launch, no GPU needed to set
    it); any other handle is refused with TF_ERR_COMM."""
    import ctypes
    from tokenflow_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.tf_comm_init_loopback(1, 8, ctypes.byref(h)) == 0
    assert lib.tf_comm_loopback_wire(h, 25.0, 50.0) == 0
    assert lib.tf_comm_loopback_wire(h, 0.0, 0.0) == 0            # off again
    assert lib.tf_comm_loopback_copies(h, 0) == 0
    assert lib.tf_comm_destroy(h) == 0
    assert lib.tf_comm_loopback_wire(None, 25.0, 50.0) == _lib.TF_ERR_COMM and lib.tf_last_error()


def test_comm_available_is_a_loader_only_probe():
    """tf_comm_available: 0 where RCCL and every entry point the library binds can be loaded, else TF_ERR_COMM with the
    reason in tf_last_error -- and no thread / socket either way (`bootstrap`'s pre-flight calls it on every rank;
    ncclGetUniqueId, which it replaced there, starts a bootstrap listener per call)."""
    import os
    from tokenflow_amd import _lib
    lib = _lib.load()
    tasks = lambda: len(os.listdir("/proc/self/task"))       # native threads of this process
    before = tasks()
    rc = lib.tf_comm_available()
    assert rc in (0, _lib.TF_ERR_COMM)
    if rc:
        assert lib.tf_last_error()
    assert lib.tf_comm_available() == rc          # resolved once per process: the answer is stable
    assert tasks() == before


@pytest.mark.parametrize("dtype,sfx", [(torch.float32, ""), (torch.float16, "_f16")])
def test_ddim_inversion_matches_reference_golden(tmp_path, monkeypatch, dtype, sfx):
    """Row f4: `tokenflow_amd.inversion.ddim_inversion` / `ddim_sample` (latent update through the oracle-backed
    `ddim_step`) against what the VERBATIM `Preprocess.ddim_inversion` / `ddim_sample` (preprocess.py:198-261,
    executed unchanged by oracle/make_golden.py) wrote and returned: same file names, same contents bit for bit
    (same operation order, same rounding points, no fused multiply-add), same in-place update of the caller's
    tensor -- in fp32 and in the float16 the reference inverts in (preprocess.py:195)."""
    import os
    from tests.conftest import load_golden
    from tokenflow_amd import inversion

    class CpuScalarOps(FakeOps):
        """torch on the CPU casts a 0-dim FIRST operand (`sigma_prev * eps`, `mu * pred_x0`, `sigma * eps`: 0-dim fp32
        tensors) to the tensor dtype before a 16-bit multiply, while a 0-dim divisor (`/ mu_prev`) stays in fp32
        opmath; on a GPU every host scalar stays in fp32 opmath -- the form `tf_ddim_step` and the oracle
        implement.  The golden was written by a CPU run of the reference, so its f16 case is reproduced by the same
        update with the three MULTIPLIED coefficients rounded to f16 first: every rounding point of the tensor
        arithmetic (product, difference, quotient, product, product, sum) is the reference's own."""

        def ddim_step(self, x, eps, mu_a, sigma_a, mu_b, sigma_b, out=None):
            rnd = lambda c: float(torch.tensor(c, dtype=torch.float32).to(x.dtype))
            return super().ddim_step(x, eps, mu_a, rnd(sigma_a), rnd(mu_b), rnd(sigma_b), out=out)

    monkeypatch.setattr(inversion, "ops", CpuScalarOps())
    g = load_golden("inversion.pt")
    model = gc.InversionModel()
    latents, cond = gc.inversion_inputs(dtype)
    assert gc.checksum(latents, cond) == g["input_checksum" + sfx], "RNG drift"
    os.makedirs(tmp_path / "latents")
    work = latents.clone()
    inv = inversion.ddim_inversion(model, cond, work, str(tmp_path), gc.INVERSION_CFG["batch_size"], save_latents=True,
                                   timesteps_to_save=model.scheduler.timesteps[::2])
    assert inv is work and inv.dtype == dtype                            # updated in place, as the reference
    assert sorted(os.listdir(tmp_path / "latents")) == sorted(g["files" + sfx])
    for name, dg in g["files" + sfx].items():
        saved = torch.load(tmp_path / "latents" / name)
        assert str(saved.dtype) == dg["dtype"]
        check(saved, dg, 0.0, name)
        assert tfu.load_source_latents_t(int(name.split("_")[-1][:-3]), str(tmp_path / "latents")).shape == dg["shape"]
    check(inv, g["inverted" + sfx], 0.0, "inverted")
    check(inversion.ddim_sample(model, inv.clone(), cond, gc.INVERSION_CFG["batch_size"]), g["reconstructed" + sfx], 0.0,
          "reconstructed")
    assert inversion.latents_save_path("latents", "2.1", "data/wolf.mp4", 500, 40) == \
        os.path.join("latents", "sd_2.1", "wolf", "steps_500", "nframes_40")       # preprocess.py:305-309


def test_fp32_as_environment_switch_is_validated():
    import os
    """TOKENFLOW_FP32_AS accepts bf16 / f16 and their common spellings; anything else fails at import with a message that
    names the variable (a bare KeyError before round 6)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import torch; from tokenflow_amd import ops; print(ops.FP32_AS)"
    for val, want in (("fp16", "torch.float16"), ("BF16", "torch.bfloat16"), ("half", "torch.float16")):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, TOKENFLOW_FP32_AS=val),
                           capture_output=True, text=True)
        assert r.returncode == 0 and want in r.stdout, (val, r.stdout, r.stderr[-300:])
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, TOKENFLOW_FP32_AS="fp8"),
                       capture_output=True, text=True)
    assert r.returncode != 0 and "TOKENFLOW_FP32_AS" in r.stderr and "ValueError" in r.stderr
