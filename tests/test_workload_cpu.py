"""Algorithmic-work accounting used by bench.py."""
from tokenflow_amd import workload


def test_step_totals_match_survey():
    fa, fn, _ = workload.step_work(workload.CONFIGS["cfg2"])
    assert abs(fa / 1e12 - 16.66) < 0.01 and abs(fn / 1e12 - 4.59) < 0.01          # 21.25 TFLOP per step
    fa, fn, _ = workload.step_work(workload.CONFIGS["cfg4"])
    assert abs(fa / 1e12 - 130.23) < 0.01 and abs(fn / 1e12 - 47.13) < 0.01
    fa, fn, _ = workload.step_work(workload.CONFIGS["cfg5"])
    assert abs(fa / 1e12 - 156.18) < 0.01 and abs(fn / 1e12 - 24.01) < 0.01


def test_block_order_and_injected_set():
    # 16 blocks: down 2+2+2, mid 1, up 3+3+3; 8 injected decoder blocks (tokenflow_utils.py:208-214)
    assert len(workload.BLOCKS) == 16
    assert [l for l, _ in workload.BLOCKS] == [0, 0, 1, 1, 2, 2, 3, 2, 2, 2, 1, 1, 1, 0, 0, 0]
    assert sum(inj for _, inj in workload.BLOCKS) == 8
    assert [inj for _, inj in workload.BLOCKS][7:10] == [False, True, True]     # up_blocks[1].attentions[0] is NOT injected


def test_level_geometry():
    c = workload.CONFIGS["cfg2"]
    assert c.K == 8 and c.levels[0] == (4096, 320, 8) and c.levels[3] == (64, 1280, 8)
    assert workload.attn_flops(8, 4096, 320) == 4 * 8 * 4096 * 320 * (4096 + 2 * 8 * 4096)
