import sys, torch
sys.path.insert(0, "/root/repo")
from eventgpt_amd.ops import hip
raw = hip.raw
def timeit(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts=[]
    for _ in range(iters):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
dev=torch.device("cuda")
for M in (768, 836, 1000):
  for N,K in [(12288,4096),(22016,4096),(4096,4096),(4096,11008),(3072,1024),(4096,1024)]:
    x=torch.randn(M,K,device=dev).to(torch.bfloat16); w=(torch.randn(N,K,device=dev)*0.02).to(torch.bfloat16)
    out=[]
    for cfg in (-1,0,1,5):
        raw.gemm_force_config(cfg)
        t=timeit(lambda: raw.linear(x,w,None,None,0,False))
        out.append(f"c{cfg}:{t*1e3:6.1f}us/{2*M*N*K/t/1e9:4.0f}TF")
    raw.gemm_force_config(-1)
    t=timeit(lambda: x@w.t())
    print(f"{M}x{N}x{K} "+" ".join(out)+f" lib:{t*1e3:6.1f}us/{2*M*N*K/t/1e9:4.0f}TF", flush=True)
