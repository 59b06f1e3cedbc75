"""Per-step device time of the first decode steps after a synchronisation (7B geometry, random weights):
an event after every graph replay, differences between consecutive events.
``python benchmarks/bench_decode_steps.py [steps] [warmup]``"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.models import EventChatConfig, EventChatModel  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda")
cfg = EventChatConfig.eventgpt_7b(vocab_size=32003)
model = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=1234)
eng = model.engine(max_len=2048, max_new_tokens=steps + warm + 8)
eng.eos = -1
emb = (torch.randn(636, cfg.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
for rep in range(3):
    eng.reset()
    eng.prefill(emb)
    for _ in range(warm):
        eng.decode_step(True)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    host = []
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        eng.decode_step(True)
        ev[i + 1].record()
        host.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) * 1e3
    d = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    print(f"rep {rep}: total {total:.2f} ms = {total / steps:.4f} per step; device per step:",
          " ".join(f"{x:.2f}" for x in d), flush=True)
    print("   host enqueue times (ms since start):", " ".join(f"{x:.1f}" for x in host[:24]), flush=True)
