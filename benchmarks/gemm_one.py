"""One GEMM shape, one implementation, a few launches: the target of a counter pass.
``python benchmarks/gemm_one.py M N K intree|lib [iters]`` (bf16, x [M, K] @ w [N, K]^T)."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.ops import hip  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
which = sys.argv[4] if len(sys.argv) > 4 else "intree"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda")
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
if __import__("os").environ.get("EVGPT_FORCE_CFG"):      # tile configuration of the in-tree kernel (tuning runs)
    hip.raw.gemm_force_config(int(__import__("os").environ["EVGPT_FORCE_CFG"]))
fn = (lambda: hip.raw.linear(x, w, None, None, 0, False)) if which == "intree" else (lambda: torch.matmul(x, w.t()))
for _ in range(3):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    fn()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"{which} {M}x{N}x{K}: {dt * 1e6:.1f} us, {2.0 * M * N * K / dt / 1e12:.0f} TFLOP/s", flush=True)
