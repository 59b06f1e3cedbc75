"""Continuous-batching throughput on one GPU: R requests (636-token multimodal prompts,
T new tokens each) through a B-slot engine.
``python benchmarks/bench_serving.py [slots] [requests] [new_tokens] [max_admit_per_step]``"""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.models import EventChatConfig, EventChatModel  # noqa: E402
from eventgpt_amd.parallel.tp import TensorParallelEngine  # noqa: E402
from eventgpt_amd.runtime.batcher import ContinuousBatcher  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
R = int(sys.argv[2]) if len(sys.argv) > 2 else 96
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ADMIT = int(sys.argv[4]) if len(sys.argv) > 4 else 8      # most prompts admitted (prefilled as one pack) per step
dev = torch.device("cuda")
cfg = EventChatConfig.eventgpt_7b(vocab_size=32003)
model = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=0)
eng = TensorParallelEngine(model, None, max_len=1024, max_batch=B, max_new_tokens=T + 8)
cb = ContinuousBatcher(eng, eos_token_id=-1, poll_every=8, max_admit_per_step=ADMIT)
g = torch.Generator(device=dev).manual_seed(0)
prompts = [(torch.randn(636, cfg.hidden_size, device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(8)]
# warm-up: captures the decode graph and the prefill kernels
cb.submit(prompts[0], 8)
cb.run_until_idle()
cb.finished.clear()
cb.steps = 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(R):
    cb.submit(prompts[i % len(prompts)], T)
cb.run_until_idle()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = cb.stats()
print(json.dumps({"slots": B, "max_admit_per_step": ADMIT, "requests": R, "new_tokens": T, "wall_s": round(dt, 3),
                  "tokens_per_s": round(R * T / dt, 1), "requests_per_s": round(R / dt, 2),
                  "mean_latency_s": round(st["mean_latency_s"], 3), "mean_ttft_s": round(st["mean_ttft_s"], 3),
                  "decode_steps": st["decode_steps"]}), flush=True)
