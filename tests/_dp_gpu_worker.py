"""Worker for the GPU data-parallel test: two ranks share GPU 0 over gloo (host-copied collectives).
A bf16 model of framework ``Linear`` layers trains three steps with the sharded AdamW, once with the
wgrad GEMM writing straight into the flat gradient buffer and once through zero-and-accumulate."""
import json
import os
import sys

os.environ["LOCAL_RANK"] = "0"          # every rank of this test drives the one GPU of the box
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventgpt_amd.models.llama import Linear, RMSNorm  # noqa: E402
from eventgpt_amd.parallel.dist import init_distributed, shutdown  # noqa: E402
from eventgpt_amd.parallel.dp import ShardedAdamW  # noqa: E402


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = Linear(256, 512, bias=False)
        self.b = Linear(512, 264, bias=False)
        self.c = Linear(264, 8, bias=False)         # a small tail: buckets of uneven sizes
        self.norm = RMSNorm(256)

    def forward(self, x):
        return self.c(self.b(self.a(self.norm(x))))


def run(direct: bool, info, xs):
    torch.manual_seed(0)
    net = Net().to(info.device, torch.bfloat16)
    opt = ShardedAdamW(net.parameters(), lr=1e-2, weight_decay=0.01, bucket_bytes=1 << 18, direct_grads=direct)
    norms = []
    for x in xs:
        opt.zero_grad()
        net(x[info.rank::info.world]).float().pow(2).mean().backward()
        if direct:
            assert net.a.weight.grad.data_ptr() == net.a.weight._grad_sink.data_ptr()
        norms.append(float(opt.step(max_grad_norm=1.0)))
    flat = torch.cat([p.detach().float().reshape(-1) for p in net.parameters()]).cpu()
    return flat, norms


def main(out_path):
    info = init_distributed(backend="gloo", use_gpu=True)
    g = torch.Generator(device=info.device).manual_seed(1)
    xs = [torch.randn(64, 256, device=info.device, generator=g).to(torch.bfloat16) for _ in range(3)]
    pd, nd = run(True, info, xs)
    pz, nz = run(False, info, xs)
    gathered = [None] * info.world
    dist.all_gather_object(gathered, pd.tolist())
    if info.rank == 0:
        json.dump({"equal": bool(torch.equal(pd, pz)), "norms_direct": nd, "norms_zero": nz,
                   "ranks_agree": all(r == gathered[0] for r in gathered), "world": info.world}, open(out_path, "w"))
    shutdown()


if __name__ == "__main__":
    main(sys.argv[1])
