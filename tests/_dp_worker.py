"""Worker for the multi-process DP tests (launched by torch.distributed.run)."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventgpt_amd.parallel.dist import init_distributed, shutdown  # noqa: E402
from eventgpt_amd.parallel.dp import ShardedAdamW, broadcast_parameters  # noqa: E402


def make_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(12, 33), torch.nn.GELU(), torch.nn.Linear(33, 7))


def _train(model, opt, info, x, y, steps, norms):
    for step in steps:
        xb = x[step, info.rank::info.world]
        yb = y[step, info.rank::info.world]
        opt.zero_grad()
        loss = (model(xb) - yb).pow(2).mean()
        loss.backward()
        if step % 2:                    # reading the norm first must not reduce the gradients a second time
            opt.global_grad_norm()
        norms.append(float(opt.step(max_grad_norm=0.5)))


def main(out_path, resume=False):
    info = init_distributed(use_gpu=False)
    model = make_model()
    if info.rank != 0:
        for p in model.parameters():
            p.data.add_(1.0)            # ranks start different; broadcast must fix it
    broadcast_parameters(model)
    opt = ShardedAdamW(model.parameters(), lr=1e-2, weight_decay=0.01, bucket_bytes=1 << 10)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(6, 8, 12, generator=g)
    y = torch.randn(6, 8, 7, generator=g)
    norms = []
    if not resume:
        _train(model, opt, info, x, y, range(6), norms)
    else:
        # stop after 3 steps, rebuild everything from the saved state (each rank its own shard), continue
        _train(model, opt, info, x, y, range(3), norms)
        weights = {k: v.clone() for k, v in model.state_dict().items()}
        shard = opt.state_dict()
        model = make_model()
        for p in model.parameters():
            p.data.zero_()
        model.load_state_dict(weights)
        opt = ShardedAdamW(model.parameters(), lr=1e-2, weight_decay=0.01, bucket_bytes=1 << 10)
        opt.load_state_dict(shard)
        _train(model, opt, info, x, y, range(3, 6), norms)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(info.world)]
    dist.all_gather(gathered, flat)
    if info.rank == 0:
        same = all(torch.equal(gathered[0], t) for t in gathered)
        json.dump({"params": flat.tolist(), "norms": norms, "ranks_agree": same,
                   "n_buckets": len(opt.buckets), "world": info.world, "reduces_launched": opt.reduces_launched,
                   "padding": [b.padded - b.numel for b in opt.buckets]}, open(out_path, "w"))
    shutdown()


if __name__ == "__main__":
    main(sys.argv[1], resume=len(sys.argv) > 2 and sys.argv[2] == "resume")
