"""GEMM micro-benchmark: every tile config of the in-tree MFMA GEMM vs the
vendor library (torch.matmul -> hipBLASLt) on the shapes the models use.
Interleaved rounds in one process; random bf16 operands; reports median TF/s."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.ops import hip  # noqa: E402

raw = hip.raw
ONLY = sys.argv[1] if len(sys.argv) > 1 else None   # substring filter on the label
SHAPES = [  # (M, N, K, label)
    (636, 12288, 4096, "llm qkv"), (636, 22016, 4096, "llm gate_up"), (636, 4096, 4096, "llm o_proj"),
    (636, 4096, 11008, "llm down"), (577, 3072, 1024, "vit qkv"), (577, 1024, 1024, "vit out"),
    (577, 4096, 1024, "vit fc1"), (577, 1024, 4096, "vit fc2"), (577, 4096, 4096, "projector"),
    (2885, 3072, 1024, "vit qkv T=5"), (2885, 1024, 4096, "vit fc2 T=5"), (22, 32064, 4096, "verify lm_head"),
    (2115, 12288, 4096, "verifier qkv"), (2115, 22016, 4096, "verifier gate_up"), (2115, 4096, 11008, "verifier down"),
    (4643, 12288, 4096, "8f qkv"), (4643, 22016, 4096, "8f gate_up"), (4643, 4096, 4096, "8f o_proj"),
    (4643, 4096, 11008, "8f down"), (20352, 4096, 4096, "bs32 prefill o_proj"),
    (32, 4096, 4096, "bs32 o_proj"), (3200, 4096, 256, "adapter up"), (3200, 256, 4096, "adapter down"),
]
CFGS = {0: "128x128s2", 1: "64x128s3", 2: "64x64s4", 3: "32x64s4", 4: "32x128s4", 5: "256x256w8"}


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


def main():
    dev = torch.device("cuda")
    print(f"{'shape':28s} {'auto':>9s} " + " ".join(f"{v:>10s}" for v in CFGS.values())
          + " " + " ".join(f"{'sk' + str(k):>6s}" for k in (2, 4, 8)) + f" {'hipBLASLt':>10s}")
    for M, N, K, label in SHAPES:
        if ONLY and ONLY not in label:
            continue
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        fl = 2.0 * M * N * K
        res = {}
        ref = (x.float() @ w.float().t())
        for k in (2, 4, 8):
            raw.gemm_force_splitk(k)
            y = raw.linear(x, w, None, None, 0, False)
            err = (y.float() - ref).abs().max().item()
            assert err < 0.05 * ref.abs().max().item() + 0.05, (label, "sk", k, err)
            res["sk%d" % k] = timeit(lambda: raw.linear(x, w, None, None, 0, False))
        raw.gemm_force_splitk(1)
        for c in CFGS:
            if c in (3, 4) and M > 1024 and N * K > 8e6:
                res[c] = float("nan"); continue
            raw.gemm_force_config(c)
            y = raw.linear(x, w, None, None, 0, False)
            err = (y.float() - ref).abs().max().item()
            assert err < 0.05 * ref.abs().max().item() + 0.05, (label, c, err)
            res[c] = timeit(lambda: raw.linear(x, w, None, None, 0, False))
        raw.gemm_force_config(-1)
        raw.gemm_force_splitk(0)
        # the default choice is timed after the forced ones: the first timing on freshly allocated
        # operands runs a few per cent low (clocks and caches still settling), which used to show as
        # "auto" trailing the very configuration it selects
        res["auto"] = timeit(lambda: raw.linear(x, w, None, None, 0, False))
        res["lib"] = timeit(lambda: torch.matmul(x, w.t()))
        tf = lambda ms: fl / ms / 1e9
        print(f"{label + f' {M}x{N}x{K}':28s} {tf(res['auto']):9.0f} " +
              " ".join(f"{tf(res[c]):10.0f}" for c in CFGS) + " "
              + " ".join(f"{tf(res['sk%d' % k]):6.0f}" for k in (2, 4, 8)) + f" {tf(res['lib']):10.0f}", flush=True)


if __name__ == "__main__":
    main()
