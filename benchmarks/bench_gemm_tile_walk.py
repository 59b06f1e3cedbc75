"""Tile walk of the 8-wave GEMM: row tiles per group (``gemm_group_m``; 0 = whole columns) and persistent
workgroups on / off, on tall outputs (batched ViT, long prompts, training batches).  TFLOP/s, 8-wave kernel forced.
``python benchmarks/bench_gemm_tile_walk.py``"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.ops import hip  # noqa: E402

raw = hip.raw
dev = torch.device("cuda")


def timeit(fn, iters=20):
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


walks = [(0, True), (0, False), (4, True), (8, True), (16, True), (8, False)]
print(f"{'shape':24s} " + " ".join(f"{'g%d%s' % (g, '' if pz else '/1wg'):>8s}" for g, pz in walks) + f" {'lib':>8s}")
for M, N, K in [(23080, 3072, 1024), (23080, 1024, 1024), (23080, 4096, 1024), (23080, 1024, 4096), (4643, 12288, 4096),
                (4643, 4096, 11008), (6688, 12288, 4096), (6688, 4096, 11008), (20352, 4096, 4096), (2115, 12288, 4096),
                (8192, 8192, 8192)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    fl = 2.0 * M * N * K
    raw.gemm_force_config(5)
    row = []
    for g, pz in walks:
        raw.gemm_group_m(g)
        raw.gemm_persist_256(pz)
        row.append(timeit(lambda: raw.linear(x, w, None, None, 0, False)))
    raw.gemm_group_m(8)
    raw.gemm_persist_256(True)
    raw.gemm_force_config(-1)
    lib = timeit(lambda: torch.matmul(x, w.t()))
    print(f"{'%dx%dx%d' % (M, N, K):24s} " + " ".join(f"{fl / t / 1e9:8.0f}" for t in row) + f" {fl / lib / 1e9:8.0f}", flush=True)


print()
print("automatic dispatch and the 4-wave tiles, whole columns (g0) against groups of 8 row tiles (g8):")
print(f"{'shape':24s} {'auto g0':>8s} {'auto g8':>8s} {'128 g0':>8s} {'128 g8':>8s} {'64x128 g0':>10s} {'64x128 g8':>10s}")
for M, N, K in [(636, 12288, 4096), (636, 22016, 4096), (836, 12288, 4096), (2885, 3072, 1024), (2885, 4096, 1024),
                (4616, 4096, 1024), (4616, 1024, 4096), (2115, 12288, 4096), (2115, 4096, 11008), (4643, 12288, 4096),
                (4643, 22016, 4096), (6688, 12288, 4096), (6688, 22016, 4096), (23080, 4096, 1024)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    fl = 2.0 * M * N * K
    row = []
    for cfg in (-1, 0, 1):
        raw.gemm_force_config(cfg)
        for g in (0, 8):
            raw.gemm_group_m(g)
            row.append(timeit(lambda: raw.linear(x, w, None, None, 0, False)))
    raw.gemm_force_config(-1)
    raw.gemm_group_m(8)
    print(f"{'%dx%dx%d' % (M, N, K):24s} " + " ".join(f"{fl / t / 1e9:8.0f}" for t in row[:4])
          + " " + " ".join(f"{fl / t / 1e9:10.0f}" for t in row[4:]), flush=True)
