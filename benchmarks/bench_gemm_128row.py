"""The 128 x 256 tile on the 8-phase schedule (config 6) against the automatic choice and hipBLASLt on the
636-row shapes of the headline prefill.  ``python benchmarks/bench_gemm_128row.py [rows]``"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.ops import hip  # noqa: E402

raw = hip.raw
M = int(sys.argv[1]) if len(sys.argv) > 1 else 636
dev = torch.device("cuda")


def timeit(fn, iters=30):
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


print(f"{'shape':26s} {'auto':>8s} {'128x256':>8s} {'+sk2':>8s} {'+sk3':>8s} {'+sk4':>8s} {'hipBLASLt':>10s}   (TFLOP/s)")
for N, K, label in [(12288, 4096, "qkv"), (22016, 4096, "gate_up"), (4096, 4096, "o_proj"), (4096, 11008, "down")]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    fl = 2.0 * M * N * K
    f = lambda: raw.linear(x, w, None, None, 0, False)  # noqa: E731
    row = [timeit(f)]
    raw.gemm_force_config(6)
    row.append(timeit(f))
    for sk in (2, 3, 4):
        raw.gemm_force_splitk(sk)
        row.append(timeit(f))
    raw.gemm_force_splitk(0)
    raw.gemm_force_config(-1)
    row.append(timeit(lambda: torch.matmul(x, w.t())))
    print(f"{label + ' %dx%dx%d' % (M, N, K):26s} " + " ".join(f"{fl / t / 1e9:8.0f}" for t in row[:-1]) + f" {fl / row[-1] / 1e9:10.0f}", flush=True)
