"""Token-level alignment modules, dataset IO, trainer, SD simulation, tokenizer tools (CPU)."""
import json
import math
import os

import pytest
import torch
from torch import nn

from eventgpt_amd.align import token_align as T
from eventgpt_amd.align.tokenizers import TokenizerAligner, check_tokenizers
from eventgpt_amd.data.tokenizer import ByteTokenizer

V, H = 50, 32


def _cfg(cls, **kw):
    base = dict(draft_vocab_size=V, target_vocab_size=V, draft_hidden_size=H, target_hidden_size=H, hidden_dim=32,
                dropout=0.0)
    base.update(kw)
    return cls(**base)


def _shift_dataset(n=64, s=12, seed=0):
    """target[t+1] = (draft[t] + 1) mod V: learnable from the previous draft token."""
    g = torch.Generator().manual_seed(seed)
    d = torch.randint(1, V, (n, s), generator=g)
    t = torch.ones(n, s, dtype=torch.long)
    t[:, 1:] = (d[:, :-1] % (V - 1)) + 1
    return T.TokenAlignmentDataset(d, t, draft_hidden=torch.randn(n, s, H, generator=g),
                                   target_hidden=torch.randn(n, s, H, generator=g))


def test_encoder_matches_torch_transformer():
    """Same parameters, same outputs as nn.TransformerEncoder (post-norm, GELU, causal)."""
    torch.manual_seed(0)
    ref_layer = nn.TransformerEncoderLayer(32, 4, 64, dropout=0.0, activation="gelu", batch_first=True)
    ref = nn.TransformerEncoder(ref_layer, num_layers=2).eval()
    mine = T.CausalEncoder(32, 4, 64, 2, dropout=0.0).eval()
    missing, unexpected = mine.load_state_dict(ref.state_dict(), strict=True)
    x = torch.randn(3, 7, 32)
    mask = torch.triu(torch.ones(7, 7, dtype=torch.bool), 1)
    with torch.no_grad():
        a = ref(x, mask=mask)
        b = mine(x)
    assert torch.allclose(a, b, atol=1e-5)
    pad = torch.zeros(3, 7, dtype=torch.bool)
    pad[1, 5:] = True
    with torch.no_grad():
        a = ref(x, mask=mask, src_key_padding_mask=pad)
        b = mine(x, pad)
    assert torch.allclose(a[~pad], b[~pad], atol=1e-5)


def test_token_adapter_trains_and_roundtrips(tmp_path):
    torch.manual_seed(0)
    ds, val = _shift_dataset(96, seed=1), _shift_dataset(32, seed=2)
    cfg = T.TokenAdapterConfig(draft_vocab_size=V, target_vocab_size=V, embed_dim=32, num_layers=1, num_heads=8,
                               ffn_dim=64, max_context=16, dropout=0.0, use_token_bias=True, use_auxiliary_loss=True)
    base = T.compute_baseline_acceptance(val)["acceptance_rate"]
    model, stats = T.train_token_adapter(ds, val, cfg, num_epochs=30, batch_size=32, learning_rate=1e-2,
                                         save_path=str(tmp_path / "best.pt"), early_stopping=30)
    m = T.compute_acceptance_metrics(model, val, batch_size=16)
    assert m["acceptance_rate"] > 0.9 > base + 0.5
    assert m["top5_rate"] >= m["acceptance_rate"] and len(stats["val_losses"]) >= 1
    again = T.TokenAdapter.load(str(tmp_path / "best.pt"))
    x = val.draft_tokens[:4]
    assert again.config.use_token_bias and again(x).shape == (4, 12, V)
    # a reference-style checkpoint (state dict only) loads through shape inference
    torch.save({"model_state_dict": model.state_dict(), "epoch": 3, "val_acc": 0.5}, tmp_path / "ref.pt")
    ref = T.TokenAdapter.load(str(tmp_path / "ref.pt"))
    assert torch.allclose(ref(x), model.eval()(x), atol=1e-6)
    sd = T.TokenAdapterForSD(model.eval())
    out = sd(val.draft_tokens[0])
    assert out[0] == val.draft_tokens[0, 0] and (out[1:] == val.target_tokens[0, 1:]).float().mean() > 0.8
    table = torch.arange(V).roll(1)
    assert torch.equal(T.TokenAdapterForSD(None, table)(torch.tensor([1, 2])), torch.tensor([0, 1]))


def test_dataset_io(tmp_path):
    ds = _shift_dataset(5)
    ds.save(str(tmp_path / "d"))
    back = T.TokenAlignmentDataset.from_directory(str(tmp_path / "d"))
    assert torch.equal(back.draft_tokens, ds.draft_tokens) and torch.equal(back.target_hidden, ds.target_hidden)
    assert back[2]["sample_id"] == "sample_2" and "draft_hidden" in back[2]
    json.dump([{"id": "a", "egpt_token_ids": [5, 6, 7], "llava_token_ids": [5, 9]},
               {"id": "b", "egpt_token_ids": [], "llava_token_ids": [1]}], open(tmp_path / "a.json", "w"))
    a = T.TokenAlignmentDataset.from_json(str(tmp_path / "a.json"), max_seq_len=4)
    assert a.draft_tokens.tolist() == [[5, 6, 7, 0]] and a.target_tokens.tolist() == [[5, 9, 0, 0]]
    json.dump({"results": [{"sample_idx": 3, "egpt_tokens": [1, 2], "vl_tokens": [1, 2]}]},
              open(tmp_path / "b.json", "w"))
    b = T.TokenAlignmentDataset.from_json(str(tmp_path / "b.json"), max_seq_len=3)
    assert b.sample_ids == ["3"] and len(b) == 1


def test_eagle_fusion_and_simulation():
    torch.manual_seed(0)
    ds = _shift_dataset(64, seed=3)
    # make the target a function of the drafter's hidden state: token = argmax of first V dims
    hid = torch.randn(64, 12, H)
    tgt = torch.ones(64, 12, dtype=torch.long)
    tgt[:, 1:] = hid[:, :-1, : V - 1].argmax(-1) + 1
    ds = T.TokenAlignmentDataset(ds.draft_tokens, tgt, draft_hidden=hid, target_hidden=torch.randn(64, 12, H))
    model = T.EAGLEFusionModule(_cfg(T.EAGLEFusionConfig, num_fusion_layers=2, num_attention_heads=4,
                                     num_speculation_heads=2))
    out = model(hid[:2], input_tokens=tgt[:2])
    assert out["logits"].shape == (2, 12, V) and out["features"].shape == (2, 12, H)
    assert out["speculation_logits"].shape == (2, 1, 12, V)
    full = model.compute_loss(hid[:4], tgt[:4], torch.randn(4, 12, H), torch.randn(4, 12, V), torch.ones(4, 12))
    assert all(float(full[k].detach()) > 0 for k in ("kl_loss", "feature_loss", "spec_loss"))
    tr = T.AlignmentTrainer(model, ds, ds, batch_size=32, learning_rate=5e-3)
    first = tr.validate()["loss"]
    stats = tr.train(25, early_stopping_patience=25)
    assert stats["val_losses"][-1] < first * 0.8
    sim = T.simulate_speculative_decoding(model, hid, tgt, gamma=3)
    assert 0.0 <= sim["acceptance_rate"] <= 1.0 and sim["simulated_speedup"] >= 1.0
    assert sim["total_speculated"] > 0 and sim["baseline_steps"] == 12
    sweeps = T.evaluate_at_different_gammas(model, hid, tgt, gammas=(1, 5))
    assert set(sweeps) == {"gamma_1", "gamma_5"}
    toks, probs = model.generate_speculation(hid[:3, :1], tgt[:3, 0], num_tokens=4, temperature=0.0)
    assert toks.shape == (3, 4) and probs.shape == (3, 4) and bool((probs > 0).all())
    assert T.compute_theoretical_speedup(1.0, 5) == 6 and T.compute_theoretical_speedup(0.0, 5) == 1.0
    assert T.compute_theoretical_speedup(0.5, 1) == pytest.approx(1.5)


def test_simulation_walk_hand_case():
    class Oracle(nn.Module):            # predicts target token p+1 correctly except at listed positions
        def __init__(self, wrong):
            super().__init__()
            self.wrong = wrong

        def forward(self, h, input_tokens=None):
            nxt = h[..., 0].long()
            logits = torch.nn.functional.one_hot(nxt, V).float()
            return {"logits": logits}
    tgt = torch.arange(1, 11)[None]                       # 10 tokens, no padding
    hid = torch.zeros(1, 10, 1)
    hid[0, :-1, 0] = tgt[0, 1:].float()                   # perfect predictions
    hid[0, 2, 0] = 0                                      # draft made at position 2 is wrong
    sim = T.simulate_speculative_decoding(Oracle(None), hid, tgt, gamma=4)
    # step1: pos0 window 4 -> accepts p0,p1 then miss (2) -> pos=3; step2: pos3 window 4 -> 4 accepted -> pos=8;
    # step3: pos8 window 1 -> 1 accepted -> pos=10
    assert sim["total_accepted"] == 7 and sim["total_speculated"] == 9 and sim["speculative_steps"] == 3
    assert sim["simulated_speedup"] == pytest.approx(10 / 3)


def test_projection_logit_distillation_modules():
    torch.manual_seed(0)
    ds = _shift_dataset(32)
    for ptype in ("linear", "mlp", "bottleneck"):
        m = T.TokenProjectionModule(_cfg(T.TokenProjectionConfig, projection_type=ptype, bottleneck_dim=8))
        out = m.compute_loss(ds.draft_hidden, ds.target_hidden, ds.draft_tokens, ds.target_tokens,
                             torch.ones(32, 12))
        assert out["total_loss"].requires_grad and -1 <= float(out["cos_sim"]) <= 1
    # identity-initialised vocabulary map: a token most likely maps to itself
    assert torch.equal(m.project_tokens(torch.tensor([[3, 7]])).argmax(-1), torch.tensor([[3, 7]]))
    one_hot = torch.nn.functional.one_hot(torch.tensor([[3, 7]]), V).float()
    assert torch.allclose(m(draft_one_hot=one_hot), m.token_logits(torch.tensor([[3, 7]])))
    la = T.LogitAlignmentModule(_cfg(T.LogitAlignmentConfig, top_k=10, use_residual=True))
    dl, tl = torch.randn(4, 6, 10), torch.randn(4, 6, 10)
    out = la.compute_loss(dl, tl, torch.ones(4, 6))
    assert la(dl).shape == (4, 6, 10) and float(out["kl_loss"]) >= 0
    tds = T.TokenAlignmentDataset(ds.draft_tokens[:, :6], ds.target_tokens[:, :6], draft_logits=torch.randn(32, 6, 10),
                                  target_logits=torch.randn(32, 6, 10))
    assert len(T.AlignmentTrainer(la, tds, tds, 16, 1e-3).train(2)["val_losses"]) == 2
    sd = T.SequenceDistillationModule(_cfg(T.SequenceDistillationConfig, num_transformer_layers=1,
                                           num_attention_heads=4))
    out = sd.compute_loss(ds.draft_hidden, ds.target_tokens, torch.ones(32, 12))
    assert sd(ds.draft_hidden).shape == (32, 12, V) and 0 <= float(out["accuracy"]) <= 1
    base = T.BaseTokenAlignmentModule
    assert base.compute_acceptance_rate(torch.tensor([[1, 2, 0]]), torch.tensor([[1, 3, 0]])) == (0.5, 1, 2)
    assert float(base.compute_kl_loss(dl, dl)) == pytest.approx(0.0, abs=1e-6)


def test_tokenizer_alignment_tools():
    a, b = ByteTokenizer(300), ByteTokenizer(400)
    a.add_tokens(["<ev_start>", "zz_only_in_draft"])
    b.add_tokens(["<video>", "<ev_start>", "xx_zz_only_in_draft_yy"])
    al = TokenizerAligner(a, b)
    s = al.summary()
    assert s["shared"] == 259 + 1 and s["draft_only"] == 1 and s["target_only"] == 2
    tmap = al.create_translation_map()
    assert tmap[a.added["<ev_start>"]] == b.added["<ev_start>"]
    assert tmap[a.added["zz_only_in_draft"]] == b.added["xx_zz_only_in_draft_yy"]       # substring fallback
    assert al.n_direct == 260 and al.n_substring == 1
    table = al.translation_tensor(tmap)
    assert int(table[a.added["<ev_start>"]]) == b.added["<ev_start>"] and int(table[70]) == 70
    ids = a.encode("hi<ev_start>", add_special_tokens=False)
    tr, unmapped = al.translate_tokens(ids, tmap)
    assert tr == b.encode("hi<ev_start>", add_special_tokens=False) and not unmapped
    rate, acc, n = al.compute_acceptance_rate_with_alignment(ids, tr, tmap)
    assert (rate, acc, n) == (1.0, 3, 3)
    rep = check_tokenizers(a, b, probes=("plain text",))
    assert rep["special_match"] and rep["probes"][0]["match_rate"] == 1.0 and rep["compatible"]


def test_token_alignment_driver(tmp_path):
    """extract (benchmark json / hidden store) -> train -> evaluate for each strategy."""
    import json as _json

    from eventgpt_amd.align import run_token_alignment as R
    from eventgpt_amd.align.store import ChunkedHiddenStateWriter
    torch.manual_seed(0)
    V, H, N, S = 24, 16, 40, 10
    # (a) token pairs from a benchmark-style json: target = (draft + 1) mod V, learnable map
    rows = []
    for i in range(N):
        d = torch.randint(1, V, (S,)).tolist()
        rows.append({"id": i, "egpt_token_ids": d, "llava_token_ids": [(x % (V - 1)) + 1 for x in d]})
    bj = tmp_path / "bench.json"
    bj.write_text(_json.dumps(rows))
    cdir = str(tmp_path / "cached")
    ds = R.main(["--mode", "extract", "--benchmark_json", str(bj), "--cached_dir", cdir, "--max_seq_len", "12"])
    assert len(ds) == N and ds.draft_tokens.shape == (N, 12)
    common = ["--cached_dir", cdir, "--device", "cpu", "--hidden_dim", "32", "--num_layers", "1",
              "--num_attention_heads", "2", "--batch_size", "8", "--learning_rate", "5e-3"]
    out = R.main(["--mode", "train", "--strategy", "token_adapter", "--num_epochs", "30", *common])
    assert os.path.exists(out["checkpoint"])
    res = R.main(["--mode", "evaluate", "--strategy", "token_adapter", *common,
                  "--output_path", str(tmp_path / "eval.json")])
    assert res["aligned"]["acceptance_rate"] > res["baseline"]["acceptance_rate"]
    assert _json.load(open(tmp_path / "eval.json"))["strategy"] == "token_adapter"

    # (b) hidden store -> projected tokens + hidden rows (+ logits) -> hidden-based strategies
    w = ChunkedHiddenStateWriter(str(tmp_path / "store"), chunk_size=16)
    W = torch.randn(V, H)
    for i in range(N):
        e = torch.randn(S, H)
        w.add_sample(e, e + 0.1 * torch.randn(S, H), {"sample_idx": i, "question_idx": 0})
    w.flush()
    torch.save({"lm_head_weight": W}, tmp_path / "lm.pt")
    hdir = str(tmp_path / "cached_h")
    ds = R.main(["--mode", "extract", "--hidden_store", str(tmp_path / "store"), "--lm_head", str(tmp_path / "lm.pt"),
                 "--cached_dir", hdir, "--max_seq_len", "10", "--with_logits"])
    assert ds.draft_hidden.shape == (N, S, H) and ds.target_logits.shape == (N, S, V)
    common = ["--cached_dir", hdir, "--device", "cpu", "--hidden_dim", "32", "--num_layers", "1",
              "--num_attention_heads", "2", "--batch_size", "8", "--learning_rate", "2e-3", "--num_epochs", "3"]
    for strat in ("eagle_fusion", "token_projection", "sequence_distillation", "logit_alignment"):
        out = R.main(["--mode", "train", "--strategy", strat, *common])
        assert os.path.exists(out["checkpoint"]), strat
        assert all(math.isfinite(x) for x in out["stats"]["train_losses"]), strat
        res = R.main(["--mode", "evaluate", "--strategy", strat, *common])
        assert "aligned" in res and "baseline" in res, strat
    assert "gamma_5" in res["gamma_sweep"] if strat == "eagle_fusion" else True


def test_benchmark_result_analysis_and_sampling_sweep(tmp_path):
    from eventgpt_amd.align import tokenizers as TK
    rows = [{"id": "a", "egpt_token_ids": [1, 2, 3, 4, 5], "llava_token_ids": [1, 2, 9, 4]},      # diverges at 2
            {"id": "b", "egpt_token_ids": [7, 8], "llava_token_ids": [7, 8, 9]},                   # never within overlap
            {"id": "c", "egpt_token_ids": [], "llava_token_ids": [1]}]                              # skipped
    p = tmp_path / "bench.json"
    p.write_text(json.dumps(rows))
    rep = TK.analyze_benchmark_results(str(p), positions=(1, 3))
    assert rep["n_samples"] == 2 and rep["total_draft_tokens"] == 7 and rep["total_accepted"] == 5
    assert rep["alpha"] == pytest.approx(5 / 7)
    assert rep["match_at_position"][1] == 1.0 and rep["match_at_position"][3] == 0.0
    assert rep["first_divergence_hist"] == {2: 2}          # sample b "diverges" at the end of the overlap (2)
    # translating the draft ids first (9 <- 3) repairs sample a
    rep2 = TK.analyze_benchmark_results(rows, tmap={1: 1, 2: 2, 3: 9, 4: 4, 5: 5, 7: 7, 8: 8})
    assert rep2["total_accepted"] == 6 and rep2["samples"][0]["first_divergence"] == 4
    assert TK.main(["--analyze", str(p)])["n_samples"] == 2

    # sweep: the "models" agree on a prefix that shrinks with temperature
    def pair(i, temp, top_p):
        keep = max(1, int(6 * (1 - temp)))
        d = list(range(10, 16))
        return d, d[:keep] + [99] * (6 - keep)
    sw = TK.matched_sampling_sweep(pair, settings=((0.0, 1.0), (0.5, 1.0), (1.0, 0.9)), n_samples=3)
    assert [round(r["alpha"], 3) for r in sw] == [1.0, 0.5, round(1 / 6, 3)]
    assert sw[0]["mean_first_divergence"] == 6 and sw[2]["top_p"] == 0.9
