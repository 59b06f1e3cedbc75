import json
import os

import pytest
import torch

from eventgpt_amd.align import store
from eventgpt_amd.align.adapters import (LEVEL_NAMES, AttentionAdapter, EAGLEStyleAdapter,
                                         FusedEAGLEAdapter, HiddenStateAdapter,
                                         MultiLayerHiddenAdapter, create_adapter, load_any_adapter)
from eventgpt_amd.align.losses import acceptance_rates, alignment_losses, lora_distill_loss


@pytest.mark.parametrize("level,expected_m", [(1, 2.1), (2, 6.3), (3, 16.8), (4, 100.7), (5, 102.8), (6, 170.0)])
def test_parameter_counts_at_4096(level, expected_m):
    with torch.device("meta"):
        m = create_adapter(level, 4096)
    assert abs(m.get_num_parameters() / 1e6 - expected_m) < 0.25, m.get_num_parameters()


def test_l1_exact_count():
    with torch.device("meta"):
        m = create_adapter(1, 4096)
    assert m.get_num_parameters() == 2 * 4096 * 256 + 2 * 4096 + 1 + 2 * 4096


@pytest.mark.parametrize("level", [1, 2, 3, 4, 5, 6])
def test_forward_loss_and_checkpoint_roundtrip(level, tmp_path):
    torch.manual_seed(level)
    m = create_adapter(level, 64, bottleneck_dim=16, num_heads=4, ffn_dim=32)
    e, v = torch.randn(3, 7, 64), torch.randn(3, 7, 64)
    mask = torch.tensor([[1] * 7, [1] * 5 + [0] * 2, [1] * 2 + [0] * 5], dtype=torch.float32)
    out = m.compute_loss(e, v, mask)
    assert out["total_loss"].requires_grad and torch.isfinite(out["total_loss"])
    assert ("pred_loss" in out) == (level >= 5)
    out["total_loss"].backward()
    p = str(tmp_path / "a.pt")
    m.save_checkpoint(p, epoch=3, metrics={"loss": 1.0})
    back, ck = load_any_adapter(p)
    assert type(back) is type(m) and ck["adapter_type"] == LEVEL_NAMES[level] and ck["epoch"] == 3
    m.eval(); back.eval()
    assert torch.allclose(m.align(e, v, mask), back.align(e, v, mask))


def test_adapters_start_near_identity():
    h = torch.randn(2, 5, 64)
    for level in (1, 2):
        m = create_adapter(level, 64, bottleneck_dim=8).eval()
        assert torch.allclose(m(h), torch.nn.functional.layer_norm(h, (64,)), atol=1e-5)
    m4 = AttentionAdapter(64, 4, 32).eval()
    out = m4(h)
    assert out.shape == h.shape and (out - h).abs().max() < (h.abs().max() * 0.5 + 1)


def test_padding_does_not_leak_into_valid_positions():
    torch.manual_seed(0)
    m = AttentionAdapter(64, 4, 32).eval()
    h = torch.randn(1, 6, 64)
    mask = torch.tensor([[1, 1, 1, 1, 0, 0]])
    a = m(h, mask)
    h2 = h.clone(); h2[:, 4:] = 99.0
    assert torch.allclose(a[:, :4], m(h2, mask)[:, :4], atol=1e-5)
    eag = EAGLEStyleAdapter(64, 4, 32).eval()
    full = eag(h)
    assert torch.allclose(full[:, :3], eag(h[:, :3]), atol=1e-5)   # causal


def test_eagle_rollout_and_fused_requires_target():
    eag = EAGLEStyleAdapter(64, 4, 32).eval()
    assert eag.speculative_decode(torch.randn(2, 1, 64), 5).shape == (2, 5, 64)
    with pytest.raises(ValueError):
        FusedEAGLEAdapter(64, 4, 32).align(torch.randn(1, 2, 64))
    ml = MultiLayerHiddenAdapter(64, 8, num_input_layers=4)
    assert ml(torch.randn(2, 4, 5, 64)).shape == (2, 5, 64)


def test_losses_hand_computed():
    pred = torch.tensor([[[1.0, 0.0], [0.0, 2.0], [5.0, 5.0]]])
    tgt = torch.tensor([[[1.0, 0.0], [0.0, -2.0], [0.0, 0.0]]])
    mask = torch.tensor([[1.0, 1.0, 0.0]])
    out = alignment_losses(pred, tgt, mask)
    assert out["mse_loss"].item() == pytest.approx(16.0 / 4)       # (0+0+0+16)/(2 pos * 2 dims)
    assert out["cos_sim"].item() == pytest.approx(0.0)             # (1 + -1)/2
    assert out["total_loss"].item() == pytest.approx(4.0 + 0.5)
    out5 = alignment_losses(pred, tgt, mask, prediction_weight=0.5)
    assert out5["pred_loss"].item() == pytest.approx((1 + 4) / 2)  # pred[0] vs tgt[1]; pos 2 masked
    assert out5["total_loss"].item() == pytest.approx(0.5 * 4.5 + 0.5 * 2.5)
    acc = acceptance_rates(torch.tensor([0.5, 0.82, 0.91, 0.99]))
    assert acc == {"accept_80": 0.75, "accept_85": 0.5, "accept_90": 0.5, "accept_95": 0.25}
    d = lora_distill_loss(torch.randn(4, 8), torch.randn(4, 8), torch.randn(20, 8))
    assert torch.isfinite(d["total_loss"])


def _fill(dirname, n=23, chunk=10, hd=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    w = store.ChunkedHiddenStateWriter(dirname, chunk_size=chunk, config={"k": 1})
    for i in range(n):
        s = 5 + i % 4
        w.add_sample(torch.randn(s, hd, generator=g), torch.randn(s, hd, generator=g),
                     {"sample_idx": i // 3, "question_idx": i % 3, "seq_len": s, "question": "q",
                      "egpt_text": "a", "vl_text": "b"})
    w.flush()
    return w


def test_chunked_store_layout_resume_and_loaders(tmp_path):
    d = str(tmp_path / "hs")
    w = _fill(d)
    idx = json.load(open(os.path.join(d, "index.json")))
    assert idx["total_samples"] == 23 and idx["chunk_size"] == 10
    assert [c["n_samples"] for c in idx["chunks"]] == [10, 10, 3]
    assert [c["path"] for c in idx["chunks"]] == ["chunk_000000.pt", "chunk_000010.pt", "chunk_000020.pt"]
    ck = torch.load(os.path.join(d, "chunks", "chunk_000010.pt"), weights_only=True)
    assert set(ck) == {"egpt_hidden", "vl_hidden", "seq_lens", "metadata", "chunk_id", "n_samples"}
    assert ck["egpt_hidden"].dtype == torch.float32 and ck["egpt_hidden"].shape[0] == 10
    # resume: a new writer knows what is done and appends after it
    w2 = store.ChunkedHiddenStateWriter(d, chunk_size=10)
    done = w2.get_processed_pairs()
    assert (0, 0) in done and (7, 1) in done and len(done) == 23
    w2.add_sample(torch.randn(4, 16), torch.randn(4, 16), {"sample_idx": 99, "question_idx": 0})
    w2.flush()
    assert json.load(open(os.path.join(d, "index.json")))["total_samples"] == 24
    allv = store.load_chunked_hidden_states(d)
    assert allv["egpt_hidden"].shape == (24, 8, 16) and len(allv["metadata"]) == 24
    # loaders: every sample exactly once per epoch, masks match lengths
    tl = store.ChunkedTrainLoader(d, batch_size=4, seed=1, pin=False)
    seen = 0
    for b in tl:
        assert b["mask"].sum(1).tolist() == b["seq_lens"].tolist()
        seen += b["egpt_hidden"].shape[0]
    assert seen == 24 and len(tl) == 3 + 3 + 1 + 1
    vl = store.ChunkedValLoader(d, batch_size=4, pin=False)
    assert sum(b["egpt_hidden"].shape[0] for b in vl) == 24
    # rank sharding: disjoint, equal batch counts
    r0 = store.ChunkedTrainLoader(d, batch_size=2, seed=1, rank=0, world=2, pin=False)
    r1 = store.ChunkedTrainLoader(d, batch_size=2, seed=1, rank=1, world=2, pin=False)
    b0, b1 = list(r0), list(r1)
    assert len(b0) == len(b1) == len(r0)
    s0 = {float(b["egpt_hidden"][i, 0, 0]) for b in b0 for i in range(2)}
    s1 = {float(b["egpt_hidden"][i, 0, 0]) for b in b1 for i in range(2)}
    assert not (s0 & s1)


def test_trainer_learns_and_resumes(tmp_path):
    from eventgpt_amd.align.trainer import main
    d = str(tmp_path / "train"); v = str(tmp_path / "val")
    hd = 32
    g = torch.Generator().manual_seed(0)
    proj = torch.randn(hd, hd, generator=g) * 0.3 + torch.eye(hd)
    for dirname, n in ((d, 64), (v, 16)):
        w = store.ChunkedHiddenStateWriter(dirname, chunk_size=32)
        for i in range(n):
            e = torch.randn(6, hd, generator=g)
            w.add_sample(e, torch.nn.functional.layer_norm(e @ proj, (hd,)), {"sample_idx": i, "question_idx": 0})
        w.flush()
    out = str(tmp_path / "tasks")
    save = main(["--train_data", d, "--val_data", v, "--output_dir", out, "--adapter_level", "1",
                 "--hidden_dim", str(hd), "--bottleneck_dim", "16", "--num_epochs", "6",
                 "--batch_size", "16", "--learning_rate", "0.01", "--cpu"])
    hist = json.load(open(os.path.join(save, "history.json")))
    assert len(hist["train_loss"]) == 6 and hist["train_loss"][-1] < hist["train_loss"][0]
    assert len(hist["val_accept_90"]) == 6
    for f in ("best_model.pt", "final_model.pt", "config.json", "last_rank0.pt"):
        assert os.path.exists(os.path.join(save, f)), f
    assert "/L1/L1_" in save
    model, _ = load_any_adapter(os.path.join(save, "best_model.pt"))
    assert isinstance(model, HiddenStateAdapter)
    # resume continues the epoch count and the history
    main(["--train_data", d, "--val_data", v, "--adapter_level", "1", "--hidden_dim", str(hd),
          "--bottleneck_dim", "16", "--num_epochs", "8", "--batch_size", "16",
          "--learning_rate", "0.01", "--cpu", "--resume", save])
    hist2 = json.load(open(os.path.join(save, "history.json")))
    assert len(hist2["train_loss"]) == 8 and hist2["train_loss"][:6] == hist["train_loss"]


def test_adapter_self_test_table():
    from eventgpt_amd.align.adapters import self_test
    rows = self_test(hidden_dim=64, batch=2, seq=7, iters=1, out=lambda *_: None)
    assert [r["level"].split("_")[0] for r in rows] == ["L1", "L2", "L3", "L4", "L5", "L5F"]
    assert all(r["parameters"] > 0 and r["forward_ms"] >= 0 for r in rows)
    assert rows[3]["parameters"] > rows[0]["parameters"]          # attention adapter > bottleneck
