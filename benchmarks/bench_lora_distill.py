"""One step of the L6 LoRA distillation trainer at the 7B geometry (random weights, a synthetic
1-frame sample, a teacher-forced answer of 50 tokens against random target hidden states): the
trainer's own forward / loss / backward / optimizer code, timed per sample.
``python benchmarks/bench_lora_distill.py [n_steps] [on|off] [micro_batch]`` (second argument: block
recomputation; third: samples per forward / backward, with one optimizer step per batch)."""
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.align.train_lora import LoRATrainer  # noqa: E402
from eventgpt_amd.constants import EVENT_TOKEN_INDEX  # noqa: E402
from eventgpt_amd.data import prepare_event_prompt, tokenizer_event_token  # noqa: E402
from eventgpt_amd.data.tokenizer import ByteTokenizer  # noqa: E402
from eventgpt_amd.models import EventChatConfig, EventChatModel  # noqa: E402
from eventgpt_amd.pipeline.dataset import make_synthetic_dataset  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ckpt = sys.argv[2] if len(sys.argv) > 2 else "off"
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda")
cfg = EventChatConfig.eventgpt_7b()
model = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=0)
tok = ByteTokenizer(cfg.vocab_size)
ds_dir = make_synthetic_dataset(tempfile.mkdtemp(prefix="evgpt_l6b_"), n_samples=2, n_events=20000, width=640, height=480)
query = "What are the key elements in this scene?"
tr = LoRATrainer(model, tok, train_chunks="", dataset_dir=ds_dir, questions=[query], gradient_accumulation=mb, micro_batch=mb,
                 output_dir=tempfile.mkdtemp(prefix="evgpt_l6b_out_"), gradient_checkpointing=ckpt)
g = torch.Generator().manual_seed(0)
img = (np.random.default_rng(0).random((480, 640, 3)) > 0.9).astype(np.uint8) * 255
px = model.get_visual_tower().event_processor(img, return_tensors="pt")["pixel_values"][0]
ids = tokenizer_event_token(prepare_event_prompt(query, "eventgpt_v1"), tok, EVENT_TOKEN_INDEX, return_tensors="pt")[None]
# answers of 40..50 tokens: the batch is ragged, as real ones are
lens = [50 - (3 * j) % 11 for j in range(mb)]
inps = [{"pixel_values": px, "input_ids": ids, "event_image_size": [480, 640],
         "gen_token_ids": torch.randint(3, 250, (n,), generator=g)} for n in lens]
targets = [torch.randn(n, cfg.hidden_size, generator=g).to(dev) for n in lens]
inp, target = inps[0], targets[0]
model.train()
acc = {"forward": 0.0, "backward": 0.0, "optimizer": 0.0}


def tick():
    torch.cuda.synchronize()
    return time.perf_counter()


tr.opt.zero_grad()
for i in range(steps + 3):
    t0 = tick()
    if mb > 1:
        preds = tr.teacher_forced_forward_batch(inps)
        total = sum(tr._loss(p, t)["total_loss"] for p, t in zip(preds, targets)) / mb
        pred = preds[0]
    else:
        pred = tr.teacher_forced_forward(inp)
        total = tr._loss(pred, target)["total_loss"]
    t1 = tick()
    total.backward()
    t2 = tick()
    tr.opt.step(max_grad_norm=1.0)
    tr.opt.zero_grad()
    t3 = tick()
    if i >= 3:
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2)):
            acc[k] += v
n_tok = int(pred.shape[0])
print("L6 distillation step, recomputation %s, micro_batch %d:" % (ckpt, mb),
      {k: round(v / steps * 1e3, 2) for k, v in acc.items()},
      "ms per step; %.1f ms per sample; peak memory %.1f GiB" % (sum(acc.values()) / steps * 1e3 / mb,
                                                                 torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)
