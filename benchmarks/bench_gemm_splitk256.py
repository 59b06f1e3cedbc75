"""Narrow outputs over long contractions (a few hundred rows into 4096 columns): which tile and how
many K splits.  Prints TFLOP/s for the 8-wave 256x256 kernel, the 128x128 and the 64x128 tile at
several split counts, next to the default choice and the vendor library."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.ops import hip  # noqa: E402
from benchmarks.bench_gemm import timeit  # noqa: E402

raw = hip.raw
SHAPES = [(636, 4096, 11008), (836, 4096, 11008), (836, 4096, 12288), (836, 4096, 22016), (636, 4096, 4096),
          (836, 4096, 4096), (577, 4096, 4096), (577, 1024, 4096)]
dev = torch.device("cuda")
combos = [(5, s) for s in (1, 2, 3, 4, 6, 8)] + [(0, s) for s in (1, 2, 3, 4)] + [(1, s) for s in (1, 2, 3)]
names = {5: "w8", 0: "128", 1: "64x128"}
print(f"{'shape':22s} {'auto':>6s} " + " ".join(f"{names[c] + '/' + str(s):>8s}" for c, s in combos) + f" {'lib':>6s}")
for M, N, K in SHAPES:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    ref = x.float() @ w.float().t()
    fl = 2.0 * M * N * K
    tf = lambda ms: fl / ms / 1e9
    raw.gemm_force_config(-1); raw.gemm_force_splitk(0)
    auto = timeit(lambda: raw.linear(x, w, None, None, 0, False))
    out = []
    for c, s in combos:
        raw.gemm_force_config(c); raw.gemm_force_splitk(s)
        y = raw.linear(x, w, None, None, 0, False)
        err = (y.float() - ref).abs().max().item()
        assert err < 0.05 * ref.abs().max().item() + 0.05, (M, N, K, c, s, err)
        out.append(timeit(lambda: raw.linear(x, w, None, None, 0, False)))
    raw.gemm_force_config(-1); raw.gemm_force_splitk(0)
    lib = timeit(lambda: torch.matmul(x, w.t()))
    print(f"{f'{M}x{N}x{K}':22s} {tf(auto):6.0f} " + " ".join(f"{tf(t):8.0f}" for t in out) + f" {tf(lib):6.0f}", flush=True)
