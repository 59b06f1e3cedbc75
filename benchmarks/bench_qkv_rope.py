import sys, time, torch
sys.path.insert(0, '.')
from eventgpt_amd import ops
from eventgpt_amd.ops import hip  # noqa: F401  (loads the extension)
raw = torch.ops.evgpt
dev = torch.device("cuda")
S, Hq, Hkv, K, L = 636, 32, 32, 4096, 2048
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(S, K, device=dev, generator=g).to(torch.bfloat16)
w = (torch.randn(3 * 4096, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
cos, sin = ops.rope_tables(L, 128, 10000.0, dev); cos, sin = cos.contiguous(), sin.contiguous()
pos0 = torch.zeros(1, dtype=torch.int32, device=dev)
q = torch.empty(1, S, Hq, 128, dtype=torch.bfloat16, device=dev)
kc = torch.zeros(1, Hkv, L, 128, dtype=torch.bfloat16, device=dev); vc = torch.zeros_like(kc)
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
qkv = raw.linear(x, w, None, None, 0, False).view(1, S, -1)
print("gemm %.1f us; rope kernel %.1f us; gemm + rope kernel %.1f us; gemm with rope epilogue %.1f us" % (
    timeit(lambda: raw.linear(x, w, None, None, 0, False)),
    timeit(lambda: raw.prefill_rope_kv(qkv, q, kc, vc, cos, sin, pos0, Hq, Hkv, 128)),
    timeit(lambda: raw.prefill_rope_kv(raw.linear(x, w, None, None, 0, False).view(1, S, -1), q, kc, vc, cos, sin, pos0, Hq, Hkv, 128)),
    timeit(lambda: raw.linear_qkv_rope(x, w, q, kc, vc, cos, sin, pos0, Hq, Hkv))), flush=True)
