"""The 64-query-row attention block with one group of four waves against two groups that share the key
tiles (`attn_force_groups`): per-call time on the LLM prefill, fine-tuning and ViT shapes."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.ops import hip
dev = torch.device("cuda")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (D, H, S, causal) in ((128, 32, 636, True), (64, 16, 577, False), (128, 32, 836, True), (64, 16, 2885, False)):
    q, k, v = (torch.randn(1, S, H, D, device=dev).to(torch.bfloat16) for _ in range(3))
    out = []
    for g in (1, 2):
        hip.raw.attn_force_groups(g)
        out.append(t(lambda: hip.attention(q, k, v, causal, None, 0)))
    hip.raw.attn_force_groups(0)
    print(f"D={D} H={H} S={S} causal={causal}: one group {out[0]:.1f} us, two groups {out[1]:.1f} us", flush=True)
