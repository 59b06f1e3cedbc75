"""Tile config x K-split sweep of the in-tree GEMM on narrow outputs over long contractions (a few
hundred rows into 4096 columns over 11008; the ViT's 4096 -> 1024 projection; dgrad of the wide
layers), the shapes the split-K rules of ``csrc/kernels/gemm.hip`` were set from.
``python benchmarks/bench_gemm_splitk.py [rows ...]`` (default 636 836)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.ops import differentiable as diff  # noqa: E402
from eventgpt_amd.ops import hip  # noqa: E402

raw = hip.raw


def timeit(fn, iters=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda")
    rows = [int(a) for a in sys.argv[1:]] or [636, 836]
    fmt = lambda t, fl: f"{t * 1e3:6.1f}us/{fl / t / 1e9:4.0f}TF"   # noqa: E731
    print("forward  y = x W^T: automatic choice, then tile config (0 = 128x128, 1 = 64x128) x K splits")
    for M in rows:
        for N, K in [(4096, 4096), (4096, 11008), (12288, 4096), (22016, 4096), (1024, 4096), (4096, 1024)]:
            x = torch.randn(M, K, device=dev).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
            fl = 2.0 * M * N * K
            raw.gemm_force_config(-1); raw.gemm_force_splitk(0)
            cells = ["auto " + fmt(timeit(lambda: raw.linear(x, w, None, None, 0, False)), fl)]
            for cfg in (0, 1):
                for sk in (1, 2, 3, 4):
                    raw.gemm_force_config(cfg); raw.gemm_force_splitk(sk)
                    cells.append(f"c{cfg}sk{sk} " + fmt(timeit(lambda: raw.linear(x, w, None, None, 0, False)), fl))
            raw.gemm_force_config(-1); raw.gemm_force_splitk(0)
            print(f"{M}x{N}x{K}: " + "  ".join(cells), flush=True)
    print("backward: automatic choice (sk0), then forced K splits on the 128x128 tile")
    for M in rows:
        for N, K, name in [(4096, 4096, "o_proj"), (4096, 11008, "down"), (12288, 4096, "qkv"), (22016, 4096, "gate_up")]:
            gy = torch.randn(M, N, device=dev).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
            xx = torch.randn(M, K, device=dev).to(torch.bfloat16)
            fl = 2.0 * M * N * K
            for kind, fn in (("dgrad", lambda: diff.linear_dgrad(gy, w)), ("wgrad", lambda: diff.linear_wgrad(gy, xx))):
                cells = []
                for sk in (0, 1, 2, 3, 4):
                    raw.gemm_force_splitk(sk)
                    cells.append(f"sk{sk} " + fmt(timeit(fn), fl))
                raw.gemm_force_splitk(0)
                print(f"{kind} {name} M={M}: " + "  ".join(cells), flush=True)


if __name__ == "__main__":
    main()
