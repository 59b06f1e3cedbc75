"""LLM prefill latency of the 7B decoder at the prompt lengths the reference quotes: 636 tokens
(EventGPT, 1 event frame), 2115 (Video-LLaVA verifier, 8 frames) and 4643 (EventGPT, 8 event frames).
``python benchmarks/bench_prefill.py [len ...]``"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.models import EventChatConfig, EventChatModel  # noqa: E402

import os  # noqa: E402

from eventgpt_amd.ops import hip  # noqa: E402

lens = [int(a) for a in sys.argv[1:]] or [636, 2115, 4643]
if os.environ.get("EVGPT_GROUP_M"):        # row tiles per group of the GEMM tile walk (0: whole columns)
    hip.raw.gemm_group_m(int(os.environ["EVGPT_GROUP_M"]))
if os.environ.get("EVGPT_GEMM_SPLITK"):    # 1: never split K in GEMMs of more than 32 rows
    hip.raw.gemm_force_splitk(int(os.environ["EVGPT_GEMM_SPLITK"]))
if os.environ.get("EVGPT_ATTN_WIDE"):      # 0: 64-row attention blocks only, 2: 128-row blocks for every shape
    hip.raw.attn_enable_wide(int(os.environ["EVGPT_ATTN_WIDE"]))
dev = torch.device("cuda")
cfg = EventChatConfig.eventgpt_7b()
cfg.max_position_embeddings = 8192
model = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=0)
eng = model.engine(max_len=max(lens) + 64)
eng.eos = -1
for L in lens:
    emb = (torch.randn(L, cfg.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
    for _ in range(3):
        eng.reset()
        eng.prefill(emb, select="greedy")
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        eng.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.prefill(emb, select="greedy")
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"prefill {L} tokens: median {ts[len(ts) // 2]:.2f} ms, min {ts[0]:.2f} ms", flush=True)
