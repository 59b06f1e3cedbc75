"""Fused AdamW shard kernel against torch.optim.AdamW (fp32 reference of the same update)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(w0, grads, lr, wd, clip_to):
    w = torch.nn.Parameter(w0.float().clone())
    opt = torch.optim.AdamW([w], lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    for g in grads:
        w.grad = g.float().clone()
        if clip_to is not None:
            torch.nn.utils.clip_grad_norm_([w], clip_to)
        opt.step()
    return w.detach()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_adamw_kernel_matches_torch(dtype):
    from eventgpt_amd.ops import hip
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    n = 100003                                  # not a multiple of the 4-element inner loop
    w0 = (torch.randn(n, generator=g) * 0.05).to(dtype)
    grads = [(torch.randn(n, generator=g) * 0.01).to(dtype) for _ in range(4)]
    master = w0.float().to(dev)
    m, v = torch.zeros_like(master), torch.zeros_like(master)
    out = torch.empty(n, dtype=dtype, device=dev)
    for t, gr in enumerate(grads, 1):
        hip.raw.fused_adamw(gr.to(dev), master, m, v, out, 1e-2, 0.9, 0.999, 1e-8, 0.1, t, 1.0)
    want = _reference(w0, grads, 1e-2, 0.1, None).to(dev)
    assert torch.allclose(master, want, rtol=2e-5, atol=1e-6), (master - want).abs().max().item()
    assert torch.equal(out, master.to(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sharded_adamw_on_gpu_matches_torch(dtype):
    """The optimizer object end to end on one GPU: flat buckets, clipping, fused update."""
    from eventgpt_amd.parallel.dp import ShardedAdamW
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    shapes = [(64, 48), (48,), (130, 7)]
    ws = [(torch.randn(*s, generator=g) * 0.1).to(dtype) for s in shapes]
    params = [torch.nn.Parameter(w.clone().to(dev)) for w in ws]
    opt = ShardedAdamW(params, lr=5e-3, weight_decay=0.05, bucket_bytes=4 << 10)   # several buckets
    assert len(opt.buckets) >= 2
    ref = [torch.nn.Parameter(w.float().clone()) for w in ws]
    ropt = torch.optim.AdamW(ref, lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    for step in range(3):
        opt.zero_grad()
        grads = [(torch.randn(*s, generator=g) * 2.0).to(dtype) for s in shapes]
        for p, r, gr in zip(params, ref, grads):
            if p.grad is None:          # bf16 matrices: the slot is handed out again with the next gradient
                p.grad = gr.to(dev)
            else:
                p.grad.copy_(gr.to(dev))
            r.grad = gr.float().clone()
        want_norm = float(torch.nn.utils.clip_grad_norm_(ref, 1.0))
        norm = float(opt.step(max_grad_norm=1.0))
        ropt.step()
        assert abs(norm - want_norm) < 1e-3 * want_norm
    for p, r in zip(params, ref):
        tol = 1e-5 if dtype == torch.float32 else 1e-2
        assert torch.allclose(p.detach().float().cpu(), r.detach(), rtol=tol, atol=tol)
    # the fp32 master shards follow the reference closely in both storage types
    flat_ref = torch.cat([r.detach().reshape(-1) for r in reversed(ref)])
    flat_master = torch.cat([b.master[:b.numel] for b in opt.buckets]).cpu()
    assert torch.allclose(flat_master, flat_ref, rtol=1e-4, atol=1e-6)


def test_device_resident_loaders_yield_the_host_loaders_batches(tmp_path):
    """Batches cut on the GPU (whole set resident, or chunk by chunk) equal the host loaders'."""
    from eventgpt_amd.align import store
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    w = store.ChunkedHiddenStateWriter(str(tmp_path / "s"), chunk_size=5)
    for i in range(13):
        n = 6 + i % 4
        e = torch.randn(n, 32, generator=g)
        w.add_sample(e, e * 2, {"sample_idx": i, "question_idx": 0})
    w.flush()
    d = str(tmp_path / "s")
    for cls, kw in ((store.ChunkedTrainLoader, dict(seed=3)), (store.ChunkedValLoader, {})):
        host = list(cls(d, batch_size=4, pin=False, **kw))
        gpu = list(cls(d, batch_size=4, pin=True, device=dev, **kw))
        assert len(host) == len(gpu) > 0
        for a, b in zip(host, gpu):
            assert b["egpt_hidden"].is_cuda and b["mask"].is_cuda
            for k in ("egpt_hidden", "vl_hidden", "mask"):
                assert torch.equal(a[k], b[k].cpu())
            assert torch.equal(a["seq_lens"], b["seq_lens"])
    data = store.load_hidden_states(d)
    assert store.DeviceBatchLoader.fits(data, dev) and not store.DeviceBatchLoader.fits(data, "cpu")
    ld = store.DeviceBatchLoader(data, [0, 2, 5, 7, 11], 2, dev, shuffle=False)
    got = list(ld)
    assert len(ld) == len(got) == 3 and got[-1]["egpt_hidden"].shape[0] == 1
    want = store.make_batch(data["egpt_hidden"][[0, 2]], data["vl_hidden"][[0, 2]], data["seq_lens"][[0, 2]])
    assert torch.equal(got[0]["egpt_hidden"].cpu(), want["egpt_hidden"]) and torch.equal(got[0]["mask"].cpu(), want["mask"])
    # two epochs of a shuffled loader visit every sample once each, in different orders
    sh = store.DeviceBatchLoader(data, None, 4, dev, shuffle=True, seed=1)
    e1 = torch.cat([b["seq_lens"] for b in sh]); e2 = torch.cat([b["seq_lens"] for b in sh])
    assert sorted(e1.tolist()) == sorted(e2.tolist()) == sorted(data["seq_lens"].tolist())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n", [1, 7, 8, 4099, 1 << 20, (1 << 22) + 5])
def test_sumsq_accumulate(dtype, n):
    from eventgpt_amd.ops import hip
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(n)
    x = (torch.randn(n, device=dev, generator=g) * 3).to(dtype)
    acc = torch.full((1,), 2.5, device=dev)
    hip.raw.sumsq_accumulate(x, acc)
    hip.raw.sumsq_accumulate(x, acc)                       # accumulates; same bits both times
    want = 2.5 + 2 * float(x.double().pow(2).sum())
    assert abs(float(acc) - want) < 1e-4 * want


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("masked", [False, True])
def test_fused_alignment_loss_matches_tensor_expressions(dtype, masked):
    """Masked MSE + cosine alignment loss: the two-kernel Function vs autograd through the tensor
    expressions (values of every reported loss, and the gradient of the total)."""
    from eventgpt_amd.align import losses as L
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(4)
    B, S, H = 6, 11, 512
    pred = torch.randn(B, S, H, device=dev, generator=g).to(dtype).requires_grad_(True)
    target = (torch.randn(B, S, H, device=dev, generator=g) * 1.5).to(dtype)
    mask = (torch.rand(B, S, device=dev, generator=g) > 0.3).float() if masked else None
    for w in (0.0, 0.3):
        out = L.alignment_losses(pred, target, mask, prediction_weight=w)
        assert type(out["mse_loss"].grad_fn).__name__ == "_RowPairLossBackward"
        out["total_loss"].backward()
        got_grad, pred.grad = pred.grad.clone(), None
        p32 = pred.detach().float().requires_grad_(True)
        t32 = target.float()
        mse = L.masked_mse(p32, t32, mask)
        cos = L.masked_mean(L.cosine_per_position(p32, t32), mask)
        align = mse + 0.5 * (1 - cos)
        if w > 0:
            pl = L.masked_mse(p32[:, :-1], t32[:, 1:], None if mask is None else mask[:, 1:])
            total = (1 - w) * align + w * pl
        else:
            total = align
        total.backward()
        tol = 1e-5 if dtype == torch.float32 else 2e-2
        assert abs(float(out["mse_loss"]) - float(mse)) < tol * float(mse)
        assert abs(float(out["cos_sim"]) - float(cos)) < max(tol, 1e-5)
        assert abs(float(out["total_loss"]) - float(total)) < tol * float(total)
        scale = float(p32.grad.abs().max())
        assert torch.allclose(got_grad.float(), p32.grad, atol=(1e-6 if dtype == torch.float32 else 1e-2) * scale + 1e-9,
                              rtol=1e-4 if dtype == torch.float32 else 2e-2)


def test_wgrad_written_straight_into_the_flat_gradient_buffer():
    """bf16 matrices get the first gradient of a step from the wgrad GEMM directly in their slot of the
    optimizer's flat buffer (no zero fill, no accumulation pass): same parameters, bit for bit, as the
    zero-and-accumulate path, also with a layer applied twice and a parameter that gets no gradient."""
    from eventgpt_amd.models.llama import Linear, RMSNorm
    from eventgpt_amd.parallel.dp import ShardedAdamW
    dev = torch.device("cuda:0")

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = Linear(256, 512, bias=False)
            self.b = Linear(512, 256, bias=False)
            self.c = Linear(512, 256, bias=False)
            self.unused = Linear(256, 64, bias=False)
            self.norm = RMSNorm(256)

        def forward(self, x):
            h = self.b(self.a(self.norm(x)))
            return self.c(self.a(h))            # `a` is used twice: autograd sums its two gradients

    def build(direct):
        torch.manual_seed(0)
        net = Net().to(dev, torch.bfloat16)
        opt = ShardedAdamW(net.parameters(), lr=1e-2, weight_decay=0.01, bucket_bytes=1 << 18, direct_grads=direct)
        return net, opt

    g = torch.Generator(device=dev).manual_seed(1)
    xs = [torch.randn(96, 256, device=dev, generator=g).to(torch.bfloat16) for _ in range(3)]
    results = []
    for direct in (True, False):
        net, opt = build(direct)
        for x in xs:
            opt.zero_grad()
            if direct:
                assert net.a.weight.grad is None and net.norm.weight.grad is not None
            net(x).float().pow(2).mean().backward()
            if direct:
                # autograd adopted the slot itself as .grad (nothing was copied)
                assert net.b.weight.grad.data_ptr() == net.b.weight._grad_sink.data_ptr()
                assert net.c.weight.grad.data_ptr() == net.c.weight._grad_sink.data_ptr()
                assert net.unused.weight.grad is None
            opt.step(max_grad_norm=1.0)
            if direct:
                assert float(net.unused.weight.grad.abs().max()) == 0.0      # settled as a zero gradient
        results.append([p.detach().clone() for p in net.parameters()])
    for pd, pz in zip(*results):
        assert torch.equal(pd, pz)


@pytest.mark.timeout(240)
def test_direct_gradients_under_data_parallel_two_ranks(tmp_path):
    """Two ranks on GPU 0 (gloo, host-copied reduce-scatter / all-gather): the sharded AdamW with the
    wgrad GEMM writing into its flat buffer ends at the same parameters, bit for bit, as the
    zero-and-accumulate path, and the ranks agree."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = str(tmp_path / "dp_gpu.json")
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29567", os.path.join(here, "_dp_gpu_worker.py"), out]
    print("launching 2-rank DP worker", flush=True)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=220)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    got = json.load(open(out))
    assert got["world"] == 2 and got["ranks_agree"] and got["equal"], got
    assert got["norms_direct"] == got["norms_zero"]


_RCCL_DP_ONE_RANK = r"""
import os, sys, json
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1])
sys.path.insert(0, sys.argv[2])
import torch, torch.distributed as dist
from eventgpt_amd.models.llama import Linear, RMSNorm
from eventgpt_amd.parallel.dp import ShardedAdamW
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = Linear(256, 512, bias=False)
        self.b = Linear(512, 256, bias=False)
        self.norm = RMSNorm(256)

    def forward(self, x):
        return self.b(self.a(self.norm(x)))


def run(direct):
    torch.manual_seed(0)
    net = Net().to(dev, torch.bfloat16)
    opt = ShardedAdamW(net.parameters(), lr=1e-2, weight_decay=0.01, bucket_bytes=1 << 18, direct_grads=direct)
    g = torch.Generator(device=dev).manual_seed(1)
    hooked = bool(opt.overlap)
    for _ in range(3):
        x = torch.randn(64, 256, device=dev, generator=g).to(torch.bfloat16)
        opt.zero_grad()
        net(x).float().pow(2).mean().backward()
        pending = [b.pending for b in opt.buckets]      # the post-accumulate hooks counted every gradient in
        opt.step(max_grad_norm=1.0)
    return torch.cat([p.detach().float().reshape(-1) for p in net.parameters()]).cpu(), hooked, pending


plain, _, _ = run(False)                                   # no process group yet: no hooks
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
direct, hooked, pending = run(True)                        # RCCL group of one: hooks + direct gradients
print(json.dumps({"equal": bool(torch.equal(plain, direct)), "hooked": hooked, "pending": pending}))
dist.destroy_process_group()
"""


@pytest.mark.timeout(240)
def test_direct_gradients_with_the_rccl_overlap_hooks_single_rank():
    """On an RCCL group the sharded AdamW counts gradients in from post-accumulate hooks (to start a
    bucket's reduce-scatter during backward).  With direct gradient slots the hook also settles the
    slot; a group of one rank (all a one-GPU box can form) runs exactly that code: same parameters as
    the hook-free zero-and-accumulate path, every bucket counted down to zero."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    print("launching single-rank RCCL DP worker", flush=True)
    r = subprocess.run([sys.executable, "-c", _RCCL_DP_ONE_RANK, "29569", root], env=env, capture_output=True,
                       text=True, timeout=220)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2500:])
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["hooked"] and got["equal"] and all(p == 0 for p in got["pending"]), got
