"""Few-row projection micro-benchmark on the 7B decoder shapes: the split-K GEMM alone,
the GEMM + fused slab consumer, and the single-row GEMV, as GB/s of weight bytes.
``python benchmarks/bench_skinny.py [M]``"""
import sys
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.ops import hip  # noqa: E402

raw = hip.raw
M = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda")
H, I = 4096, 11008


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)


# several weight copies so that successive calls do not hit in the 256 MB memory-side cache
NW = 6
shapes = {"o_proj": (H, H, False), "qkv": (3 * H, H, True), "gate_up": (2 * I, H, True), "down": (H, I, False)}
print(f"M={M}")
for name, (N, K, norm) in shapes.items():
    ws = [rnd(N, K, scale=0.02) for _ in range(NW)]
    x, x1 = rnd(M, K, scale=0.5), rnd(1, K, scale=0.5)
    nw = rnd(K) if norm else None
    res, res1 = rnd(M, N), rnd(1, N)
    i = [0]

    def nxt():
        i[0] = (i[0] + 1) % NW
        return ws[i[0]]
    gb = N * K * 2 / 1e3   # bytes -> GB/s when divided by us
    if name == "gate_up":
        t_sk = timeit(lambda: raw.linear_partial(x, nxt()))
        t_gm = timeit(lambda: raw.reduce_swiglu(raw.linear_partial(raw.rmsnorm(x, nw, 1e-5), nxt())))
        h1 = torch.empty(1, I, device=dev, dtype=torch.bfloat16)
        t_gv = timeit(lambda: raw.decode_mlp_up(nxt(), x1, nw, 1e-5, h1))
    elif norm:
        t_sk = timeit(lambda: raw.linear_partial(x, nxt()))
        t_gm = timeit(lambda: raw.linear(raw.rmsnorm(x, nw, 1e-5), nxt(), None, None, 0, False))
        t_gv = float("nan")
    else:
        t_sk = timeit(lambda: raw.linear_partial(x, nxt()))
        t_gm = timeit(lambda: raw.reduce_residual_norm(raw.linear_partial(x, nxt()), res, None, 0.0, res))
        t_gv = timeit(lambda: raw.decode_gemv_residual(nxt(), x1, res1))
    print(f"{name:8s} N={N:6d} K={K:6d}  gemm only {t_sk:7.1f} us ({gb / t_sk:5.0f} GB/s)   gemm+reduce {t_gm:7.1f} us "
          f"({gb / t_gm:5.0f} GB/s)   gemv(M=1) {t_gv:7.1f} us ({gb / t_gv:5.0f} GB/s)", flush=True)
