"""Tensor parallelism: sharded decoder (2, 4 and 8 gloo ranks) == unsharded model."""
import json
import os
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world,port", [(2, 29541), (4, 29542), (8, 29543)])
def test_tp_decoder_matches_full_model(tmp_path, world, port):
    out = str(tmp_path / "tp.json")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(HERE, "_tp_worker.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=220)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.load(open(out))
    assert len(got["ranks"]) == world
    for rk in got["ranks"]:
        assert rk["err"] < 1e-4, rk
        assert rk["tok"] == got["want_tok"]
        assert rk["gather_err"] < 1e-4, rk
    assert got["i_local"] % 64 == 0          # intermediate shard padded for the MFMA GEMM


def test_shard_helpers_cover_everything_once():
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    from eventgpt_amd.parallel.tp import shard_layer, shard_vocab
    cfg = EventChatConfig(vocab_size=203, hidden_size=64, intermediate_size=172,
                          num_hidden_layers=1, num_attention_heads=4, torch_dtype="float32")
    m = EventChatModel.from_config(cfg, torch.float32, "cpu", seed=1)
    lyr = m.model.layers[0]
    x = torch.randn(3, 64)
    ref_gu = torch.nn.functional.silu(lyr.mlp.gate_proj(x)) * lyr.mlp.up_proj(x)
    ref = lyr.mlp.down_proj(ref_gu)
    acc = torch.zeros_like(ref)
    rows = []
    for r in range(4):
        w_qkv, w_o, w_gu, w_down = shard_layer(lyr, r, 4, 4, 4, 16, 172)
        assert w_qkv.shape == (3 * 16, 64) and w_o.shape == (64, 16)
        i_l = w_gu.shape[0] // 2
        gu = x @ w_gu.t()
        acc += (torch.nn.functional.silu(gu[:, :i_l]) * gu[:, i_l:]) @ w_down.t()
        sh, lo = shard_vocab(m.lm_head.weight.data, r, 4)
        rows.append(sh[: max(0, min(sh.shape[0], 203 - lo))])
    assert torch.allclose(acc, ref, atol=1e-5)
    assert torch.equal(torch.cat(rows), m.lm_head.weight.data)


def test_shard_helpers_at_eight_ranks_with_7b_head_and_mlp_counts():
    """32 heads / 11008 intermediate / 32003 vocabulary over 8 ranks (the batched serving config):
    11008 / 8 = 1376 is no multiple of 64, so every rank's MLP shard is zero-padded to 1408; the padded
    shards still sum to the unsharded layer and the vocabulary rows are covered exactly once."""
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    from eventgpt_amd.parallel.tp import shard_layer, shard_vocab
    cfg = EventChatConfig(vocab_size=32003, hidden_size=256, intermediate_size=11008,
                          num_hidden_layers=1, num_attention_heads=32, torch_dtype="float32")
    m = EventChatModel.from_config(cfg, torch.float32, "cpu", seed=2)
    lyr = m.model.layers[0]
    x = torch.randn(2, 256)
    ref = lyr.mlp.down_proj(torch.nn.functional.silu(lyr.mlp.gate_proj(x)) * lyr.mlp.up_proj(x))
    q_ref = lyr.self_attn.q_proj(x)
    acc = torch.zeros_like(ref)
    rows, q_parts = [], []
    for r in range(8):
        w_qkv, w_o, w_gu, w_down = shard_layer(lyr, r, 8, 32, 32, 8, 11008)
        assert w_qkv.shape == (3 * 4 * 8, 256) and w_o.shape == (256, 4 * 8)
        i_l = w_gu.shape[0] // 2
        assert i_l == 1408 and w_down.shape == (256, 1408)
        gu = x @ w_gu.t()
        acc += (torch.nn.functional.silu(gu[:, :i_l]) * gu[:, i_l:]) @ w_down.t()
        q_parts.append(x @ w_qkv[:32].t())
        sh, lo = shard_vocab(m.lm_head.weight.data, r, 8)
        assert sh.shape[0] == 4001 and lo == r * 4001
        rows.append(sh[: max(0, min(sh.shape[0], 32003 - lo))])
    assert torch.allclose(acc, ref, atol=1e-4)
    assert torch.allclose(torch.cat(q_parts, 1), q_ref, atol=1e-5)
    assert torch.equal(torch.cat(rows), m.lm_head.weight.data)
