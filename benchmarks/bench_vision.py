"""Vision path only (CLIP ViT-L/14-336 + projector + adaptor + pooling), T frames:
``python benchmarks/bench_vision.py [T] [iters]`` prints ms per call and per frame."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.models import EventChatConfig, EventChatModel  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda")
if __import__("os").environ.get("EVGPT_ATTN_WIDE"):      # 0: 64-row attention blocks only (A/B runs)
    from eventgpt_amd.ops import hip
    hip.raw.attn_enable_wide(int(__import__("os").environ["EVGPT_ATTN_WIDE"]))
cfg = EventChatConfig.eventgpt_7b(vocab_size=32003)
cfg.num_hidden_layers = 1                      # the decoder is not used here
model = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=0)
frames = [torch.randn(3, 336, 336, device=dev).to(torch.bfloat16) for _ in range(T)]
with torch.no_grad():
    for _ in range(3):
        model.encode_event_frames(frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = model.encode_event_frames(frames)
    torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / iters
print(f"T={T}: {ms:.3f} ms per call, {ms / T:.3f} ms per frame, out {tuple(out.shape)}", flush=True)
