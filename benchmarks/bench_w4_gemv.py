#!/usr/bin/env python
"""The four one-row projections of a 7B decoder layer on 4-bit weights, one by one.

    python benchmarks/bench_w4_gemv.py [copies] [iters]

Every projection cycles through `copies` distinct weight matrices (default 12: 300-600 MB, more than the
256 MB memory-side cache holds) so that each launch streams its codes from HBM as a decode step does.
Prints microseconds per launch and the bytes per second of codes + scales.  EVGPT_W4_ROW1=0 routes the
calls to the generic multi-row kernel for comparison; EVGPT_W4_ROW1_BLOCKS sets the grid of the one-row one.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventgpt_amd import ops  # noqa: E402
from eventgpt_amd.ops import hip  # noqa: E402
from eventgpt_amd.quant import w4 as Q  # noqa: E402


def timed(fn, n, iters):
    for i in range(2 * n):
        fn(i % n)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % n)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 240
    dev = torch.device("cuda")
    raw = hip.raw
    H, I, Hq, D, L = 4096, 11008, 32, 128, 1024
    g = torch.Generator().manual_seed(0)

    def copies(N, K):
        base = Q.quantize_weight(torch.randn(N, K, generator=g) * 0.02).to(dev)
        out = []
        for i in range(n):
            out.append((base.packed.roll(i * 3, 0).contiguous(), base.absmax.roll(i * 3, 0).contiguous(), base.code))
        return out

    x = (torch.randn(1, H, generator=g) * 0.5).to(dev, torch.bfloat16)
    xi = (torch.randn(1, I, generator=g) * 0.5).to(dev, torch.bfloat16)
    nw = torch.ones(H, dtype=torch.bfloat16, device=dev)
    cos, sin = ops.rope_tables(L, D, 10000.0, dev)
    pos = torch.tensor([77], dtype=torch.int32, device=dev)
    qo = torch.zeros(1, Hq * D, dtype=torch.bfloat16, device=dev)
    kc = torch.zeros(1, Hq, L, D, dtype=torch.bfloat16, device=dev)
    vc = torch.zeros_like(kc)
    h = torch.empty(1, I, dtype=torch.bfloat16, device=dev)
    y = torch.zeros(1, H, dtype=torch.bfloat16, device=dev)

    rows = []
    w = copies(3 * H, H)
    rows.append(("qkv + rope   12288 x 4096", 3 * H * H, timed(
        lambda i: raw.decode_qkv_w4(x, nw, w[i][0], w[i][1], w[i][2], qo, kc, vc, cos, sin, pos, 1e-6, Hq, Hq, D, False),
        n, iters)))
    del w
    w = copies(H, H)
    rows.append(("o_proj + res  4096 x 4096", H * H, timed(
        lambda i: raw.decode_gemv_residual_w4(w[i][0], w[i][1], w[i][2], x, y), n, iters)))
    del w
    w = copies(2 * I, H)
    rows.append(("gate/up      22016 x 4096", 2 * I * H, timed(
        lambda i: raw.decode_mlp_up_w4(w[i][0], w[i][1], w[i][2], x, nw, 1e-6, h), n, iters)))
    del w
    w = copies(H, I)
    rows.append(("down + res   4096 x 11008", H * I, timed(
        lambda i: raw.decode_gemv_residual_w4(w[i][0], w[i][1], w[i][2], xi, y), n, iters)))
    total = 0.0
    for name, weights, us in rows:
        nbytes = weights // 2 + weights // 64 * 4
        total += us
        print(f"{name}: {us:6.2f} us  {nbytes / us * 1e-6:5.2f} TB/s")
    print(f"sum {total:.2f} us per layer, {total * 32e-3:.3f} ms per 32 layers")


if __name__ == "__main__":
    main()
