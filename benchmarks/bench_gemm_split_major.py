"""Split-K slab GEMMs of a few hundred rows: workgroups ordered tile by tile (the splits of a tile are neighbours)
against split by split (an XCD stays inside one K range).  ``python benchmarks/bench_gemm_split_major.py [rows]``"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.ops import hip  # noqa: E402

raw = hip.raw
M = int(sys.argv[1]) if len(sys.argv) > 1 else 636
dev = torch.device("cuda")


def timeit(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print(f"{'shape / tile':34s} " + " ".join(f"{'sk%d %s' % (sk, 'tile' if not sm else 'split'):>10s}" for sk in (1, 2, 3, 4) for sm in (0, 1) if sk > 1 or not sm))
for N, K, label in [(4096, 4096, "o_proj"), (4096, 11008, "down"), (12288, 4096, "qkv"), (22016, 4096, "gate_up (no glu)")]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    ref = x.float() @ w.float().t()
    for cfg, cname in ((0, "128x128"), (1, "64x128")):
        raw.gemm_force_config(cfg)
        row = []
        for sk in (1, 2, 3, 4):
            for sm in (0, 1):
                if sk == 1 and sm:
                    continue
                raw.gemm_force_splitk(sk)
                raw.gemm_split_major(sm)
                got = raw.linear_partial(x, w).sum(0)
                assert (got - ref).abs().max().item() < 0.05 * ref.abs().max().item() + 0.05
                row.append(timeit(lambda: raw.linear_partial(x, w)))
        print(f"{label + ' %dx%dx%d ' % (M, N, K) + cname:34s} " + " ".join(f"{t:10.1f}" for t in row), flush=True)
raw.gemm_force_config(-1)
raw.gemm_force_splitk(0)
raw.gemm_split_major(0)
