import sys, torch
sys.path.insert(0, '.')
from eventgpt_amd.ops import hip
raw = hip.raw
dev = torch.device("cuda")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts=[]
    for _ in range(iters):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
print(f"{'shape':24s} {'8wave':>7s} {'128x128':>8s} {'64x128':>7s} {'lib':>7s}")
for M in (23080, 11540, 6688, 4616, 2885, 1154):
    for N,K in ((3072,1024),(1024,1024),(4096,1024),(1024,4096),(4096,2048),(4096,256)):
        x=torch.randn(M,K,device=dev).to(torch.bfloat16); w=(torch.randn(N,K,device=dev)*0.02).to(torch.bfloat16)
        fl=2.0*M*N*K
        f=lambda: raw.linear(x,w,None,None,0,False)
        row=[]
        for c in (5,0,1):
            raw.gemm_force_config(c); row.append(timeit(f))
        raw.gemm_force_config(-1)
        row.append(timeit(lambda: torch.matmul(x,w.t())))
        print(f"{'%dx%dx%d'%(M,N,K):24s} "+" ".join(f"{fl/t/1e9:7.0f}" for t in row), flush=True)
