"""The oracle restatement must reproduce what the VERBATIM reference produced
(tests/golden/*.pt, written by oracle/make_golden.py).  CPU only."""
import pytest
import torch

from tests.conftest import load_golden

from oracle import golden_cases as gc
from oracle import tokenflow_oracle as orc
from oracle.golden_util import check
from tests import fake_diffusers as fd


@pytest.mark.parametrize("name", list(gc.ATTN_CASES))
def test_attn_core_matches_reference(name, golden_attn):
    K, S, h, d, sched, t = gc.ATTN_CASES[name]
    q, k, v = gc.attn_inputs(name)
    g = golden_attn[name]
    assert gc.checksum(q, k, v) == g["input_checksum"], "RNG drift: regenerate goldens"
    inject = orc.should_inject(t, sched)
    for fn in (orc.ext_attn_core, orc.ext_attn_core_bmm):
        check(fn(q, k, v, h, d ** -0.5, inject), g["out_pnp"], 2e-6, f"{name}/pnp/{fn.__name__}")
        check(fn(q, k, v, h, d ** -0.5, False), g["out_sdedit"], 2e-6, f"{name}/sdedit/{fn.__name__}")


def test_should_inject_semantics():
    # tokenflow_utils.py:86,124
    assert orc.should_inject(1000, [])
    assert not orc.should_inject(999, [])
    assert not orc.should_inject(1000, None)
    assert orc.should_inject(5, torch.tensor([7, 5]))
    assert not orc.should_inject(6, torch.tensor([7, 5]))
    assert not orc.should_inject(6, torch.tensor([]))


@pytest.mark.parametrize("name", list(gc.PROP_CASES))
def test_propagation_matches_reference(name, golden_prop):
    K, n, S, D, dt = gc.PROP_CASES[name]
    piv, kf_out, hidden = gc.prop_inputs(name)
    g = golden_prop[name]
    assert gc.checksum(piv, kf_out, *hidden) == g["input_checksum"], "RNG drift: regenerate goldens"
    for bi in range(K):
        norm = hidden[bi].float().view(3, n, S, D)
        idx, _ = orc.nn_search(norm[0], piv[0], bi)
        gi = g["chunks"][bi]["idx"]
        assert len(idx) == len(gi)
        for a, b in zip(idx, gi):
            assert torch.equal(a, b.long())           # integer work: bit-exact
        out = orc.gather_blend(kf_out, idx, bi, n, residual=hidden[bi])
        assert str(out.dtype) == g["chunks"][bi]["out_dtype"]
        check(out, g["chunks"][bi]["out"], 0.0, f"{name}/chunk{bi}")   # same op order: bit-exact


def test_blend_weights_closed_form():
    # n=4 -> [0.6225, 0.6792, 0.7311, 0.6971]; independent of the chunk index
    w = orc.blend_weights(4, 1)
    assert torch.allclose(w, torch.tensor([0.6225, 0.6792, 0.7311, 0.6971]), atol=1e-4)
    assert torch.equal(w, orc.blend_weights(4, 7))


def test_blocks_match_reference(golden_blocks):
    cfg = gc.BLOCKS_CFG
    torch.manual_seed(cfg["seed"])
    pipe = fd.FakePipeline(dims=cfg["dims"], heads=cfg["heads"], cross_dim=cfg["cross_dim"]).eval()
    assert gc.checksum(*pipe.parameters()) == golden_blocks["weights_checksum"], "RNG drift"
    blocks = [b for _, b in pipe.unet.transformer_blocks_in_order()]
    injected = set(id(pipe.unet.up_blocks[r].attentions[a].transformer_blocks[0])
                   for r, aa in {1: [1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}.items() for a in aa)
    for t in cfg["timesteps"]:
        inp = gc.blocks_inputs(t)
        run = golden_blocks["runs"][t]
        states = [orc.BlockState() for _ in blocks]
        with torch.no_grad():
            for i, (blk, x) in enumerate(zip(blocks, inp["pivotal"])):
                inj = orc.should_inject(t, cfg["schedule"] if id(blk) in injected else [])
                y = orc.block_forward(blk, states[i], x, pivotal=True, inject=inj,
                                      encoder_hidden_states=inp["enc"])
                check(y, run["pivotal"][i], 3e-5, f"t{t}/pivotal/{i}")
            for c in range(cfg["n_chunks"]):
                for i, (blk, x) in enumerate(zip(blocks, inp["chunks"][c])):
                    y = orc.block_forward(blk, states[i], x, pivotal=False, batch_idx=c,
                                          encoder_hidden_states=inp["enc_n"])
                    check(y, run["chunks"][c][i], 3e-5, f"t{t}/chunk{c}/{i}")
            # patched resnet: plain forward, then the injection copy after conv2 (86-91)
            res = pipe.unet.up_blocks[1].resnets[1]
            x, temb = inp["res_x"], inp["res_temb"]
            h = res.conv1(res.nonlinearity(res.norm1(x)))
            h = h + res.time_emb_proj(res.nonlinearity(temb))[:, :, None, None]
            h = res.conv2(res.dropout(res.nonlinearity(res.norm2(h))))
            if orc.should_inject(t, cfg["conv_schedule"]):
                orc.conv_inject_(h)
            check((x + h) / res.output_scale_factor, run["resnet"], 1e-6, f"t{t}/resnet")


def test_oracle_adazero_block_matches_reference_golden():
    """AdaLayerNormZero block (gate_msa in both passes, tokenflow_utils.py:365-366; scale/shift/gate on the
    feed-forward, 417-424) against the verbatim reference's outputs."""
    g = load_golden("adazero.pt")
    blk = gc.adazero_block()
    assert gc.checksum(*blk.parameters()) == g["weights_checksum"], "RNG drift"
    inp = gc.adazero_inputs()
    assert gc.checksum(inp["pivotal"], *inp["chunks"]) == g["input_checksum"]
    st = orc.BlockState()
    with torch.no_grad():
        y = orc.block_forward(blk, st, inp["pivotal"], pivotal=True, encoder_hidden_states=inp["enc"],
                              timestep=inp["timestep"])
        check(y, g["pivotal"], 3e-5, "adazero/pivotal")
        for c in range(gc.ADAZERO_CFG["K"]):
            y = orc.block_forward(blk, st, inp["chunks"][c], pivotal=False, batch_idx=c,
                                  encoder_hidden_states=inp["enc_n"], timestep=inp["timestep"])
            check(y, g["chunks"][c], 3e-5, f"adazero/chunk{c}")
