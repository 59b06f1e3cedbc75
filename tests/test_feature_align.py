"""Vision-feature alignment strategies on synthetic paired features (CPU)."""
import pytest
import torch

from eventgpt_amd.align import feature_align as FA


def _pairs(n=256, de=24, dt=32, seed=0, text=False):
    g = torch.Generator().manual_seed(seed)
    ev = torch.randn(n, de, generator=g)
    w = torch.randn(de, dt, generator=g) / de ** 0.5
    rgb = torch.tanh(ev @ w) + 0.05 * torch.randn(n, dt, generator=g)
    tx = torch.tanh(ev @ torch.randn(de, 16, generator=g)) if text else None
    return FA.AlignmentDataset(ev, rgb, tx)


def test_base_losses_and_retrieval():
    m = FA.LightweightAlignmentModule(FA.LightweightAlignmentConfig(event_dim=8, target_dim=8, adapter_type="linear"))
    x = torch.randn(16, 8)
    r = m.evaluate_retrieval(x, x)
    assert r == {"R@1": 1.0, "R@5": 1.0, "R@10": 1.0}
    perm = torch.arange(16).roll(1)
    assert m.evaluate_retrieval(x, x[perm])["R@1"] < 0.3
    good, bad = m.compute_contrastive_loss(x, x), m.compute_contrastive_loss(x, x[perm])
    assert float(good) < float(bad)
    # hand check of the symmetric InfoNCE on a 2x2 case
    a = torch.eye(2)
    loss, logits = FA.info_nce(a, a, 2.0)
    assert float(loss) == pytest.approx(float(torch.log(1 + torch.exp(torch.tensor(-2.0)))), rel=1e-5)
    assert m.compute_similarity(x, x).diagonal().allclose(torch.ones(16), atol=1e-5)
    assert torch.allclose(m(x), x * 1.1, atol=1e-6)     # identity adapter + 0.1 residual, scale 1


@pytest.mark.parametrize("strategy,kw", [("lightweight", dict(adapter_type="mlp")),
                                         ("lightweight", dict(adapter_type="adapter", use_whitening=True)),
                                         ("contrastive", dict(projection_dim=16)),
                                         ("contrastive", dict(projection_dim=16, use_queue=True, queue_size=64)),
                                         ("triple_modal", dict(projection_dim=16, text_dim=16, num_attention_heads=4))])
def test_strategies_learn(strategy, kw, tmp_path):
    torch.manual_seed(0)
    ds = _pairs(text=strategy == "triple_modal")
    tr, va = ds.split(0.75, seed=1)
    assert len(tr) == 192 and len(va) == 64
    model = FA.create_alignment(strategy, event_dim=24, target_dim=32, hidden_dim=48, dropout=0.0, **kw)
    trainer = FA.AlignmentTrainer(model, tr, va, batch_size=64, learning_rate=3e-3, warmup_epochs=2)
    before = trainer.validate()
    hist = trainer.train(30, save_path=str(tmp_path / "best.pt"), early_stopping_patience=30)
    after = trainer.validate()
    assert after["loss"] < before["loss"] and hist[-1]["train_loss"] < hist[0]["train_loss"]
    if strategy == "lightweight":
        assert after["R@5"] > max(before["R@5"], 0.3)
    else:       # the contrastive objectives are optimised in the projection space
        model.eval()
        with torch.no_grad():
            args = (va.event_features, va.rgb_features) + ((va.text_features,) if strategy == "triple_modal" else ())
            met = model.compute_loss(*args, return_metrics=True)
        assert float(met["acc_e2t" if strategy == "contrastive" else "acc_event_image"]) > 0.3
    again = FA.create_alignment(strategy, event_dim=24, target_dim=32, hidden_dim=48, dropout=0.0, **kw)
    again.load_checkpoint(str(tmp_path / "best.pt"))
    # best.pt is the best-epoch snapshot of the same architecture (not the final weights)
    assert {k: v.shape for k, v in again.state_dict().items()} == {k: v.shape for k, v in model.state_dict().items()}
    fresh = FA.create_alignment(strategy, event_dim=24, target_dim=32, hidden_dim=48, dropout=0.0, **kw)
    assert any(not torch.equal(again.state_dict()[k], fresh.state_dict()[k]) for k in fresh.state_dict())
    rep = FA.evaluate_alignment_comprehensive(model, va)
    assert rep["quality"] in ("excellent", "medium", "poor") and rep["num_samples"] == 64


def test_triple_modal_zero_shot_and_queue():
    torch.manual_seed(0)
    m = FA.TripleModalAlignmentModule(FA.TripleModalAlignmentConfig(event_dim=8, target_dim=8, text_dim=8,
                                                                    hidden_dim=16, projection_dim=8,
                                                                    num_attention_heads=2, dropout=0.0))
    enc = m.encode_all_modalities(torch.randn(4, 8), torch.randn(4, 8), torch.randn(4, 8))
    assert set(enc) == {"event", "image", "text", "event_attended", "event_text_attended"}
    assert torch.allclose(enc["event_attended"].norm(dim=-1), torch.ones(4), atol=1e-5)
    assert m.zero_shot_classify(torch.randn(5, 8), torch.randn(3, 8)).shape == (5,)
    c = FA.ContrastiveAlignmentModule(FA.ContrastiveAlignmentConfig(event_dim=8, target_dim=8, hidden_dim=16,
                                                                    projection_dim=4, use_queue=True, queue_size=10))
    c.train()
    for _ in range(3):
        c.compute_loss(torch.randn(4, 8), torch.randn(4, 8))
    assert int(c.queue_ptr) == 2 and int(c.queue_filled) == 10        # 12 keys wrapped around a queue of 10


def test_whitening():
    torch.manual_seed(0)
    w = FA.WhiteningLayer(6, momentum=1.0)
    x = torch.randn(4000, 6) @ torch.randn(6, 6) + 3.0
    w.train()
    y = w(x)
    cov = (y - y.mean(0)).t() @ (y - y.mean(0)) / y.shape[0]
    assert torch.allclose(cov, torch.eye(6), atol=2e-2) and y.mean(0).abs().max() < 1e-3
    w.eval()
    y2 = w(x)
    assert torch.allclose(y, y2, atol=1e-3) and w._cached is not None


def test_reconstruction_bridge():
    torch.manual_seed(0)
    net = FA.SimpleE2VIDNet(5, base_channels=4)
    vox = torch.randn(2, 5, 32, 48)
    img, h = net(vox)
    assert img.shape == (2, 1, 32, 48) and h.shape == (2, 16, 8, 12) and 0 <= float(img.min()) <= float(img.max()) <= 1
    img2, _ = net(vox, h)
    assert not torch.allclose(img, img2)                       # recurrent state matters
    # GRU gate algebra against the textbook form
    cell = FA.ConvGRUCell(3, 3, 1)
    x, hid = torch.randn(1, 3, 2, 2), torch.randn(1, 3, 2, 2)
    r, u = torch.sigmoid(cell.gates(torch.cat([x, hid], 1))).chunk(2, 1)
    cand = torch.tanh(cell.candidate(torch.cat([x, r * hid], 1)))
    assert torch.allclose(cell(x, hid), (1 - u) * hid + u * cand, atol=1e-6)

    class Enc(torch.nn.Module):
        def forward(self, px):
            return px.mean((2, 3)).repeat(1, 4)                 # [B, 12]
    cfg = FA.ReconstructionBridgingConfig(event_dim=12, target_dim=12, reconstruction_size=(16, 16), num_temporal_bins=5)
    bridge = FA.ReconstructionBridgingModule(cfg, e2vid_model=FA.E2VIDWrapper(model=net), clip_encoder=Enc())
    f = bridge(vox)
    assert f.shape == (2, 12) and bridge.e2vid.hidden_state is not None
    grey = torch.full((1, 1, 4, 4), 0.5)
    rgb = bridge.preprocess_image(grey)
    assert torch.allclose(rgb[0, :, 0, 0], (0.5 - torch.tensor(FA.CLIP_MEAN)) / torch.tensor(FA.CLIP_STD), atol=1e-6)
    st = bridge.get_latency_stats()
    assert st["avg_reconstruction_ms"] > 0 and st["total_avg_ms"] >= st["avg_encoding_ms"]
    bridge.reset_e2vid_state(); bridge.reset_latency_stats()
    assert bridge.e2vid.hidden_state is None and bridge.get_latency_stats()["total_avg_ms"] == 0.0
    assert not any(p.requires_grad for p in bridge.e2vid.parameters())
    # voxel grids built by this package feed the net directly
    from eventgpt_amd.data.events import synthetic_events, voxel_grid
    v = torch.from_numpy(voxel_grid(synthetic_events(5000, 64, 48, seed=1), bins=5)).float()[None]
    assert net(v)[0].shape == (1, 1, 48, 64)
    losses = FA.finetune_e2vid_on_events(FA.SimpleE2VIDNet(5, 4), torch.randn(4, 5, 16, 16), torch.rand(4, 1, 16, 16), epochs=3, lr=1e-2)
    assert losses[-1] < losses[0]


def test_utilities(tmp_path):
    ds = _pairs(32)
    ds.save(str(tmp_path / "f"))
    back = FA.AlignmentDataset.from_directory(str(tmp_path / "f"))
    assert torch.equal(back.rgb_features, ds.rgb_features) and set(back[0]) == {"event_features", "rgb_features"}
    ident = torch.nn.Identity()
    loader = [{"event_features": torch.randn(8, 16), "rgb_features": torch.randn(8, 16)}]
    assert FA.quick_feasibility_test(ident, ident, loader)["needs_alignment"] == "strong"
    same = torch.randn(8, 16)
    assert FA.quick_feasibility_test(ident, ident, [{"event_features": same, "rgb_features": same}])[
        "needs_alignment"] == "light"
    assert FA.diagnose_alignment_quality(same, same).overall_quality == "excellent"
    assert FA.diagnose_alignment_quality(same, -same).overall_quality == "poor"
    pick = FA.auto_select_alignment_strategy
    assert pick(True, 20000, False, target_performance="best").strategy.startswith("Triple")
    assert pick(True, 6000, True).strategy.startswith("Contrastive")
    assert pick(False, 0, True).strategy == "Reconstruction Bridging"
    assert pick(True, 2000, False).strategy.endswith("Fine-tune") and pick(False, 10, False).confidence == "low"
    cmp = FA.compare_alignment_strategies(*_pairs(128).split(0.75), num_epochs=3, event_dim=24, target_dim=32,
                                          hidden_dim=32)
    assert set(cmp) == {"lightweight", "contrastive"} and cmp["lightweight"]["parameters"] > 0
