"""Where one instruction-tuning step goes (7B geometry, random weights, synthetic data):
``python benchmarks/bench_sft_step.py [projector|lora|full] [n_steps] [micro_batch]`` prints the mean
host-preparation, forward, backward and optimizer times, and how long the host needed to issue
the forward and the backward (close to the phase's own time = the GPU waits for the host).  With
``micro_batch`` > 1 a step is one right-padded forward / backward over that many samples and one
optimizer step (an accumulation window of that size); the per-sample time is printed as well."""
import sys
import tempfile
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.cli.train_sft import SFTTrainer  # noqa: E402
from eventgpt_amd.data.tokenizer import ByteTokenizer  # noqa: E402
from eventgpt_amd.models import EventChatConfig, EventChatModel  # noqa: E402
from eventgpt_amd.parallel.dist import init_distributed  # noqa: E402
from eventgpt_amd.pipeline.dataset import InstructionDataset, make_synthetic_dataset  # noqa: E402

tune = sys.argv[1] if len(sys.argv) > 1 else "projector"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
info = init_distributed(use_gpu=True)
cfg = EventChatConfig.eventgpt_7b()
model = EventChatModel.from_config(cfg, torch.bfloat16, info.device, seed=0)
tok = ByteTokenizer(cfg.vocab_size)
ds = InstructionDataset(make_synthetic_dataset(tempfile.mkdtemp(prefix="evgpt_sftb_"), n_samples=(steps + 3) * mb,
                                               n_events=60000, width=640, height=480))
tr = SFTTrainer(model, tok, ds, info, tune=tune, output_dir=tempfile.mkdtemp(prefix="evgpt_sftb_out_"),
                gradient_accumulation=mb, micro_batch=mb)
model.train()
acc = {"prepare": 0.0, "forward": 0.0, "backward": 0.0, "optimizer": 0.0}
issue = {"forward": 0.0, "backward": 0.0}


def tick():
    torch.cuda.synchronize()
    return time.perf_counter()


for i in range(steps + 3):
    t0 = tick()
    preps = [tr.prepare_sample(ds[i * mb + j]) for j in range(mb)]
    prep = preps[0]
    t1 = tick()
    loss = tr.batch_loss(preps)
    e1 = time.perf_counter()
    t2 = tick()
    loss.backward()
    e2 = time.perf_counter()
    t3 = tick()
    tr.opt.step(max_grad_norm=1.0)
    tr.opt.zero_grad()
    t4 = tick()
    if i >= 3:
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            acc[k] += v
        issue["forward"] += e1 - t1
        issue["backward"] += e2 - t2
tokens = int(prep["ids"].shape[0]) + 5 + 576
print(tune, {k: round(v / steps * 1e3, 2) for k, v in acc.items()}, "ms per step; ~%d tokens per sample" % tokens,
      "; host issue", {k: round(v / steps * 1e3, 2) for k, v in issue.items()}, flush=True)
if mb > 1:
    gpu = (acc["forward"] + acc["backward"] + acc["optimizer"]) / steps * 1e3
    print(f"{tune} micro_batch {mb}: {gpu / mb:.2f} ms per sample (forward + backward + optimizer step / {mb}); "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
