"""LoRA modules (math, PEFT-layout IO, merge) and the L6 trainer on a tiny model (CPU)."""
import json
import os

import pytest
import torch

from eventgpt_amd.align import lora as L
from eventgpt_amd.models import EventChatConfig, EventChatModel


def _model(seed=0):
    return EventChatModel.from_config(EventChatConfig.tiny(), torch.float32, "cpu", seed=seed)


def test_lora_math_io_and_merge(tmp_path):
    torch.manual_seed(0)
    m = _model()
    emb = torch.randn(1, 9, m.config.hidden_size)
    ref = m.model(inputs_embeds=emb).last_hidden_state
    wrapped = L.apply_lora(m, r=4, alpha=8, dropout=0.0)
    assert len(wrapped) == 4 * m.config.num_hidden_layers and all(w.split(".")[-1] in L.DEFAULT_TARGETS for w in wrapped)
    c = L.count_parameters(m)
    assert c["trainable"] == sum(p.numel() for p in L.lora_parameters(m)) and c["trainable"] < 0.2 * c["total"]
    # B = 0: wrapped model equals the base model
    assert torch.allclose(m.model(inputs_embeds=emb).last_hidden_state, ref, atol=1e-6)
    for p in L.lora_parameters(m):
        torch.nn.init.normal_(p, std=0.05)
    mod = m.model.layers[0].self_attn.q_proj
    x = torch.randn(3, m.config.hidden_size)
    want = x @ mod.base.weight.t() + (x @ mod.lora_A.weight.t() @ mod.lora_B.weight.t()) * 2.0
    assert torch.allclose(mod(x), want, atol=1e-5)
    res = torch.randn(3, mod.out_features)
    assert torch.allclose(mod(x, residual=res), want + res, atol=1e-5)
    out = m.model(inputs_embeds=emb).last_hidden_state
    assert not torch.allclose(out, ref, atol=1e-4)
    out.sum().backward()
    assert all(p.grad is not None for p in L.lora_parameters(m)) and m.model.layers[0].mlp.gate_proj.weight.grad is None
    L.save_lora(m, str(tmp_path / "ad"), "EventGPT-7b")
    cfg = json.load(open(tmp_path / "ad" / "adapter_config.json"))
    assert cfg["peft_type"] == "LORA" and cfg["r"] == 4 and cfg["target_modules"] == list(L.DEFAULT_TARGETS)
    from safetensors.torch import load_file
    keys = load_file(str(tmp_path / "ad" / "adapter_model.safetensors")).keys()
    assert "base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight" in keys
    m2 = _model()
    assert len(L.load_lora(m2, str(tmp_path / "ad"))) == 2 * len(wrapped)
    assert torch.allclose(m2.model(inputs_embeds=emb).last_hidden_state, out.detach(), atol=1e-6)
    assert L.merge_lora(m2) == len(wrapped) and not L.lora_modules(m2)
    assert torch.allclose(m2.model(inputs_embeds=emb).last_hidden_state, out.detach(), atol=1e-5)
    assert "lora" not in " ".join(m2.state_dict().keys())


def test_l6_trainer_end_to_end(tmp_path):
    from eventgpt_amd.align import train_lora as TL
    from eventgpt_amd.pipeline import dataset as D
    from eventgpt_amd.pipeline import extract_hidden_states as X
    data = D.make_synthetic_dataset(str(tmp_path / "ds"), n_samples=3, n_events=3000)
    X.main(["--dataset_dir", data, "--output_dir", str(tmp_path / "hs"), "--random_init", "tiny", "--device", "cpu",
            "--max_new_tokens", "8", "--max_questions", "2", "--chunk_size", "4", "--event_image_key",
            "event_image_1f"])
    store = str(tmp_path / "hs" / "chunked_train_1s_bf16")
    torch.save({"lm_head_weight": torch.randn(320, 64)}, tmp_path / "lm.pt")
    res = TL.main(["--existing_chunks", store, "--dataset_dir", data, "--val_chunks", store, "--random_init", "tiny",
                   "--cpu", "--num_epochs", "4", "--gradient_accumulation", "2", "--learning_rate", "5e-3",
                   "--lora_rank", "4", "--lora_alpha", "8", "--lora_dropout", "0.0", "--lm_head_path",
                   str(tmp_path / "lm.pt"), "--max_questions", "2", "--output_dir", str(tmp_path / "L6")])
    h = res["history"]
    assert len(h) == 4 and h[0]["train_samples"] == 6 and h[0]["val_samples"] == 6
    assert h[-1]["train_total_loss"] < h[0]["train_total_loss"] and h[-1]["val_cos_sim"] > h[0]["val_cos_sim"] - 1e-3
    assert "train_token_match" in h[0] and "train_ce_loss" in h[0]
    for f in ("best_model/adapter_model.safetensors", "final_model/adapter_config.json", "history.json", "config.json"):
        assert os.path.exists(tmp_path / "L6" / f), f
    cfg = json.load(open(tmp_path / "L6" / "config.json"))
    assert cfg["lora"]["r"] == 4 and cfg["parameters"]["trainable"] > 0


def test_l6_micro_batches_equal_single_sample_passes(tmp_path):
    """``--micro_batch 2`` (two samples of an accumulation window as one right-padded pass) trains to the
    same losses as one sample per pass: same windows, same optimizer steps, fp32 on the CPU."""
    from eventgpt_amd.align import train_lora as TL
    from eventgpt_amd.pipeline import dataset as D
    from eventgpt_amd.pipeline import extract_hidden_states as X
    data = D.make_synthetic_dataset(str(tmp_path / "ds"), n_samples=3, n_events=3000)
    X.main(["--dataset_dir", data, "--output_dir", str(tmp_path / "hs"), "--random_init", "tiny", "--device", "cpu",
            "--max_new_tokens", "9", "--max_questions", "2", "--chunk_size", "4", "--event_image_key",
            "event_image_1f"])
    store = str(tmp_path / "hs" / "chunked_train_1s_bf16")
    torch.save({"lm_head_weight": torch.randn(320, 64)}, tmp_path / "lm.pt")
    hist = []
    for mb in (1, 2, 3):
        torch.manual_seed(3)
        res = TL.main(["--existing_chunks", store, "--dataset_dir", data, "--random_init", "tiny", "--cpu",
                       "--num_epochs", "2", "--gradient_accumulation", "4", "--micro_batch", str(mb),
                       "--learning_rate", "5e-3", "--lora_rank", "4", "--lora_alpha", "8", "--lora_dropout", "0.0",
                       "--lm_head_path", str(tmp_path / "lm.pt"), "--max_questions", "2",
                       "--output_dir", str(tmp_path / f"L6_{mb}")])
        hist.append(res["history"])
    for other in hist[1:]:
        for a, b in zip(hist[0], other):
            assert a["train_samples"] == b["train_samples"] == 6
            for k in ("train_total_loss", "train_mse_loss", "train_cos_sim", "train_ce_loss"):
                assert abs(a[k] - b[k]) <= 2e-4 * max(1.0, abs(a[k])), (k, a[k], b[k])


def test_l6_sample_falls_back_to_recorded_token_ids():
    """An answer whose text decoded to "" (ids outside the tokenizer's round-trip range)
    is trained on from the drafter's recorded ids instead of being dropped."""
    from eventgpt_amd.align import train_lora as TL

    class Tok:
        def encode(self, text, add_special_tokens=False):
            return [ord(c) for c in text]

    class DS(list):
        pass

    t = TL.LoRATrainer.__new__(TL.LoRATrainer)
    t.tok, t.questions = Tok(), ["q"]
    t.prepare_input = lambda ds, sample, q: {"input_ids": torch.zeros(1, 3, dtype=torch.long)}
    ds = DS([{}])
    meta = {"sample_idx": 0, "question_idx": 0, "egpt_text": "", "egpt_tokens": list(range(1000, 1012))}
    got = t._prepare_sample(ds, meta, 8)
    assert got["gen_token_ids"].tolist() == list(range(1000, 1008))
    assert t._prepare_sample(ds, dict(meta, egpt_tokens=[]), 8) is None
    assert t._prepare_sample(ds, dict(meta, egpt_text="hello world"), 8)["gen_token_ids"].tolist() == \
        [ord(c) for c in "hello wo"]


@pytest.mark.timeout(240)
def test_l6_trainer_two_ranks_gloo(tmp_path):
    """Two data-parallel ranks: samples are split, gradients averaged once per optimizer
    step, both ranks end with identical adapters and rank 0 writes them."""
    import subprocess
    import sys
    from eventgpt_amd.pipeline import dataset as D
    from eventgpt_amd.pipeline import extract_hidden_states as X
    data = D.make_synthetic_dataset(str(tmp_path / "ds"), n_samples=4, n_events=3000)
    X.main(["--dataset_dir", data, "--output_dir", str(tmp_path / "hs"), "--random_init", "tiny", "--device", "cpu",
            "--max_new_tokens", "8", "--max_questions", "1", "--chunk_size", "4", "--event_image_key",
            "event_image_1f"])
    store = str(tmp_path / "hs" / "chunked_train_1s_bf16")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", "-m", "eventgpt_amd.align.train_lora", "--existing_chunks", store,
           "--dataset_dir", data, "--random_init", "tiny", "--cpu", "--num_epochs", "2", "--gradient_accumulation",
           "2", "--lora_rank", "4", "--lora_alpha", "8", "--max_questions", "1", "--output_dir", str(tmp_path / "L6")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=220,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-3000:]
    h = json.load(open(tmp_path / "L6" / "history.json"))
    assert len(h) == 2 and h[0]["train_samples"] == 4          # 2 per rank, summed
    assert json.load(open(tmp_path / "L6" / "config.json"))["world_size"] == 2
    assert os.path.exists(tmp_path / "L6" / "final_model" / "adapter_model.safetensors")
