"""A/B of the decode step with and without weight prefetch into the die-level cache on a second stream
(7B geometry, random weights, 636-token prompt): the configurations alternate on one engine, several
rounds, so clock state and weight placement are shared.
``python benchmarks/bench_decode_prefetch.py [steps] [rounds]``"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventgpt_amd.models import EventChatConfig, EventChatModel  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda")
cfg = EventChatConfig.eventgpt_7b(vocab_size=32003)
model = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=1234)
eng = model.engine(max_len=2048, max_new_tokens=steps + 128)
eng.eos = -1
emb = (torch.randn(636, cfg.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
MB = 1 << 20
configs = [("off", None)]
for join in ("o_proj", "step"):
    for gu in (0, 24 * MB, 64 * MB):
        for blocks in (256, 1024):
            configs.append((f"join={join} gate_up={gu // MB}MB blocks={blocks}",
                            {"blocks": blocks, "gate_up_bytes": gu, "join": join}))
want = None
results = {name: [] for name, _ in configs}
for rnd in range(rounds):
    for name, pf in configs:
        eng.l3_prefetch = pf
        eng._graph_greedy.clear()
        eng.reset()
        eng.prefill(emb)
        for _ in range(100):
            eng.decode_step(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.decode_step(True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        toks = eng.out_tokens[0, :100 + steps].clone()
        if want is None:
            want = toks
        same = bool(torch.equal(want, toks))
        results[name].append(ms)
        print(f"round {rnd} {name}: {ms:.4f} ms/step tokens_equal={same} graph={eng.graph_captured}", flush=True)
print("--- medians")
for name, _ in configs:
    v = sorted(results[name])
    print(f"{name}: {v[len(v) // 2]:.4f} ms/step (min {v[0]:.4f})")
