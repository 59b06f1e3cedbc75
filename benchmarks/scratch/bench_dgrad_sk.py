import sys, torch
sys.path.insert(0, "/root/repo")
from eventgpt_amd.ops import hip
from eventgpt_amd.ops import differentiable as diff
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
M=836
for N,K,name in [(4096,4096,"o_proj"),(4096,11008,"down"),(12288,4096,"qkv"),(22016,4096,"gate_up")]:
    gy=torch.randn(M,N,device=dev).to(torch.bfloat16); w=(torch.randn(N,K,device=dev)*0.02).to(torch.bfloat16); x=torch.randn(M,K,device=dev).to(torch.bfloat16)
    for kind,fn in (("dgrad",lambda: diff.linear_dgrad(gy,w)),("wgrad",lambda: diff.linear_wgrad(gy,x))):
        out=[]
        for sk in (0,1,2,3,4):
            raw.gemm_force_splitk(sk)
            t=timeit(fn)
            out.append(f"sk{sk}:{t*1e3:6.1f}us/{2*M*N*K/t/1e9:4.0f}TF")
        raw.gemm_force_splitk(0)
        print(kind,name," ".join(out),flush=True)
