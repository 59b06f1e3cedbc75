"""Pieces of one decoder layer's prefill at a given row count, timed alone (7B shapes): the row-parallel
projections with their residual + following RMSNorm as (a) GEMM with residual epilogue + norm kernel and
(b) GEMM to fp32 K-split slabs + one consumer that sums, adds the residual and normalises; and the whole
prefill eagerly launched against one captured graph.  ``python benchmarks/bench_prefill_parts.py [rows]``"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.models import EventChatConfig, EventChatModel  # noqa: E402
from eventgpt_amd.ops import hip  # noqa: E402

raw = hip.raw
M = int(sys.argv[1]) if len(sys.argv) > 1 else 636
dev = torch.device("cuda")
H, I = 4096, 11008
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05).to(torch.bfloat16)
x, o, h = rnd(M, H), rnd(M, H), rnd(M, I)
w_o, w_down, nw = rnd(H, H), rnd(H, I), rnd(H).abs() + 0.5


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, a, w in (("o_proj", o, w_o), ("down", h, w_down)):
    def plain():
        y = raw.linear(a, w, None, x, 0, False)
        return y, raw.rmsnorm(y, nw, 1e-6)
    t_plain = timeit(plain)
    row = [f"{name} {M}x{w.shape[0]}x{w.shape[1]}: GEMM(+residual) + rmsnorm {t_plain:.1f} us"]
    ref_y, ref_n = plain()
    for sk in (0, 1, 2, 3, 4):
        raw.gemm_force_splitk(sk)
        def fused():
            return raw.reduce_residual_norm(raw.linear_partial(a, w), x, nw, 1e-6, None)
        t = timeit(fused)
        y, yn = fused()
        err = (y.float() - ref_y.float()).abs().max().item()
        row.append(f"slabs(split {'auto' if sk == 0 else sk}, {raw.linear_partial(a, w).shape[0]} slabs) + consumer {t:.1f} us (max diff {err:.3f})")
    raw.gemm_force_splitk(0)
    print("; ".join(row), flush=True)

# ---- whole prefill: eager launches vs one captured graph of the same launches
cfg = EventChatConfig.eventgpt_7b()
model = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=0)
eng = model.engine(max_len=2048)
eng.eos = -1
emb = (torch.randn(M, H, device=dev) * 0.02).to(torch.bfloat16)


def eager():
    eng.reset()
    eng.prefill(emb, select="greedy")


t_eager = timeit(eager, 20)
static = emb.clone()
pos0 = torch.zeros(1, dtype=torch.int32, device=dev)
box = {}


def body():
    box["x"] = eng._forward_rows(static, 0, pos0)


gph = eng._capture(body)


def graphed():
    eng.reset()
    static.copy_(emb)
    gph.replay()
    eng.pos[0] = M - 1
    eng.x[0].copy_(box["x"][0, -1])
    eng._logits_and_commit(greedy=True)


t_graph = timeit(graphed, 20)
print(f"prefill {M} tokens: eager {t_eager / 1e3:.3f} ms, captured graph {t_graph / 1e3:.3f} ms", flush=True)
