"""Backward GEMMs of a linear layer at fine-tuning shapes (one ~836-token sample; 8 samples): dgrad
dX = dY W (weight K-strided) and wgrad dW = dY^T X (both operands K-strided), every tile config of the
in-tree kernel vs torch.matmul (hipBLASLt).  Median of interleaved rounds, TFLOP/s."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.ops import differentiable as diff  # noqa: E402
from eventgpt_amd.ops import hip  # noqa: E402

raw = hip.raw
CFGS = {-1: "auto", 0: "128x128", 1: "64x128", 2: "64x64", 5: "256x256w8"}
LAYERS = [(12288, 4096, "qkv"), (4096, 4096, "o_proj"), (22016, 4096, "gate_up"), (4096, 11008, "down")]


def timeit(fn, iters=15):
    for _ in range(3):
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
    rows = [int(a) for a in sys.argv[1:]] or [836, 6688]
    print(f"{'product':34s} " + " ".join(f"{v:>10s}" for v in CFGS.values()) + f" {'hipBLASLt':>10s}")
    for M in rows:
        for N, K, name in LAYERS:
            gy = torch.randn(M, N, device=dev).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
            x = torch.randn(M, K, device=dev).to(torch.bfloat16)
            fl = 2.0 * M * N * K
            for kind, fn, lib in (("dgrad", lambda: diff.linear_dgrad(gy, w), lambda: gy @ w),
                                  ("wgrad", lambda: diff.linear_wgrad(gy, x), lambda: gy.t() @ x)):
                res = []
                for c in CFGS:
                    raw.gemm_force_config(c)
                    res.append(fl / timeit(fn) / 1e9)
                raw.gemm_force_config(-1)
                res.append(fl / timeit(lib) / 1e9)
                print(f"{kind + ' ' + name + f' M={M} {N}x{K}':34s} " + " ".join(f"{r:10.0f}" for r in res), flush=True)


if __name__ == "__main__":
    main()
