"""Instruction fine-tuning loop on a tiny CPU model: label masking, the three tuning modes,
output formats, resume, and 2 data-parallel ranks."""
import json
import os
import subprocess
import sys

import pytest
import torch

from eventgpt_amd.cli import train_sft as T
from eventgpt_amd.constants import EVENT_TOKEN_INDEX, IGNORE_INDEX
from eventgpt_amd.data.tokenizer import ByteTokenizer
from eventgpt_amd.pipeline.dataset import make_synthetic_dataset


@pytest.fixture(scope="module")
def dataset_dir(tmp_path_factory):
    return make_synthetic_dataset(str(tmp_path_factory.mktemp("sft") / "ds"), n_samples=6, n_events=3000,
                                  width=64, height=48)


def test_build_example_masks_prompt_and_event():
    tok = ByteTokenizer()
    ids, labels = T.build_example(tok, "What moves?", "A car.")
    assert ids.shape == labels.shape and (ids == EVENT_TOKEN_INDEX).sum() == 1
    sup = labels != IGNORE_INDEX
    n_sup = int(sup.sum())
    # the supervised tail is exactly " A car." + EOS, everything before it is ignored
    assert torch.equal(sup.nonzero().flatten(), torch.arange(ids.shape[0] - n_sup, ids.shape[0]))
    assert ids[-1] == tok.eos_token_id and labels[-1] == tok.eos_token_id
    assert tok.decode(ids[sup][:-1].tolist()).strip() == "A car."
    assert labels[ids == EVENT_TOKEN_INDEX].item() == IGNORE_INDEX


@pytest.mark.parametrize("tune", ["projector", "lora", "full"])
def test_modes_train_and_save(dataset_dir, tmp_path, tune):
    out = str(tmp_path / tune)
    args = ["--random_init", "tiny", "--dataset_dir", dataset_dir, "--output_dir", out, "--tune", tune, "--cpu",
            "--num_epochs", "3", "--learning_rate", "5e-3", "--num_frames", "2", "--gradient_accumulation", "2",
            "--lora_rank", "4"]
    torch.manual_seed(0)
    res = T.main(args)
    hist = res["history"]
    assert len(hist) == 3 and hist[-1]["train_loss"] < hist[0]["train_loss"], hist
    assert res["steps"] == 9                                    # 6 samples / accumulation 2, 3 epochs
    sd = torch.load(res["paths"]["projector"], weights_only=True)
    assert any(k.startswith("model.visual_projector.0.") for k in sd)
    assert any(k.startswith("model.feature_adaptor.") for k in sd)
    meta = json.load(open(os.path.join(out, "history.json")))
    assert meta["tune"] == tune and meta["trainable_parameters"] > 0
    if tune == "lora":
        assert os.path.exists(os.path.join(res["paths"]["lora"], "adapter_config.json"))
    if tune == "full":
        assert os.path.exists(os.path.join(res["paths"]["model"], "config.json"))
    # a later stage can start from the projector file
    T.main(["--random_init", "tiny", "--dataset_dir", dataset_dir, "--output_dir", str(tmp_path / "s2"), "--cpu",
            "--pretrain_mm_mlp_adapter", res["paths"]["projector"], "--num_epochs", "1", "--max_samples", "2",
            "--num_frames", "2"])


def test_projector_mode_leaves_decoder_untouched_and_resumes(dataset_dir, tmp_path):
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    from eventgpt_amd.parallel.dist import DistInfo
    from eventgpt_amd.pipeline.dataset import InstructionDataset
    model = EventChatModel.from_config(EventChatConfig.tiny(), torch.float32, "cpu", seed=0)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    tr = T.SFTTrainer(model, ByteTokenizer(), InstructionDataset(dataset_dir), DistInfo(), "projector", 2,
                      learning_rate=5e-3, output_dir=str(tmp_path / "a"))
    tr.train(num_epochs=2)
    changed = {n for n, p in model.named_parameters() if not torch.equal(p, before[n])}
    assert changed and all(("visual_projector" in n) or ("feature_adaptor" in n) for n in changed), changed
    # resume: 2 epochs, then 1 more == state of epoch 2 carried over (step counter continues)
    model2 = EventChatModel.from_config(EventChatConfig.tiny(), torch.float32, "cpu", seed=0)
    tr2 = T.SFTTrainer(model2, ByteTokenizer(), InstructionDataset(dataset_dir), DistInfo(), "projector", 2,
                       learning_rate=5e-3, output_dir=str(tmp_path / "a"))
    r = tr2.train(num_epochs=3, resume=True)
    assert [h["epoch"] for h in r["history"]] == [1, 2, 3] and r["steps"] == 18


@pytest.mark.timeout(300)
def test_two_ranks(dataset_dir, tmp_path):
    out = str(tmp_path / "dp")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29553", "-m", "eventgpt_amd.cli.train_sft", "--random_init", "tiny",
           "--dataset_dir", dataset_dir, "--output_dir", out, "--cpu", "--num_epochs", "2", "--num_frames", "2",
           "--learning_rate", "5e-3"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-2000:]
    meta = json.load(open(os.path.join(out, "history.json")))
    assert meta["world_size"] == 2 and len(meta["history"]) == 2
    assert meta["history"][-1]["steps"] == 6                     # 3 samples per rank per epoch
    assert os.path.exists(os.path.join(out, "mm_projector.pt"))


def test_micro_batches_take_the_same_steps_as_single_samples(tmp_path):
    """Samples of an accumulation window run as ONE right-padded forward / backward (no attention mask:
    causal attention never looks at the padding behind a sequence) and the mean of the per-sample losses:
    parameters after training equal those of the one-sample-at-a-time loop."""
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    from eventgpt_amd.parallel.dist import DistInfo
    from eventgpt_amd.pipeline.dataset import InstructionDataset
    # answers of different lengths, so the windows really are padded
    ds_dir = make_synthetic_dataset(str(tmp_path / "ds"), n_samples=8, n_events=3000, width=64, height=48)
    meta = json.load(open(os.path.join(ds_dir, "EventGPT_Instruction_Subset.json")))
    for i, smp in enumerate(meta):
        smp["conversations"][1]["value"] = "A synthetic scene" + " with cars" * (i % 4) + "."
    json.dump(meta, open(os.path.join(ds_dir, "EventGPT_Instruction_Subset.json"), "w"))
    got = {}
    for mb in (1, 4, 3):
        model = EventChatModel.from_config(EventChatConfig.tiny(), torch.float32, "cpu", seed=0)
        tr = T.SFTTrainer(model, ByteTokenizer(), InstructionDataset(ds_dir), DistInfo(), "full", 2,
                          learning_rate=2e-3, gradient_accumulation=4, micro_batch=mb,
                          output_dir=str(tmp_path / f"mb{mb}"))
        res = tr.train(num_epochs=2, save_every_epoch=False)
        assert res["steps"] == 4
        got[mb] = (torch.cat([p.detach().reshape(-1) for p in model.parameters()]),
                   [h["train_loss"] for h in res["history"]])
    for mb in (4, 3):
        assert torch.allclose(got[mb][0], got[1][0], atol=2e-5), (mb, (got[mb][0] - got[1][0]).abs().max())
        assert all(abs(a - b) < 1e-4 for a, b in zip(got[mb][1], got[1][1])), (got[mb][1], got[1][1])
