"""One-shot all-reduce (csrc/kernels/comm.hip) on ONE GPU: W virtual ranks driven by a single
launch (every block co-resident), checked against a plain fp32 sum; graph replays; the
bounded wait's give-up path forced from the host; the vocab-parallel argmax pair kernels."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _oracle(partials, res, norm_w, eps):
    total = torch.zeros_like(partials[0][0])
    for p in partials:                      # rank order, slabs first: the kernel's order
        total = total + p.sum(0)
    x = total.to(torch.bfloat16).float() + (res.float() if res is not None else 0.0)
    x = x.to(torch.bfloat16)
    xn = None
    if norm_w is not None:
        xf = x.float()
        xn = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(torch.bfloat16).float() * norm_w.float()
        xn = xn.to(torch.bfloat16)
    return total, x, xn


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("shape", [(1, 4096, 1), (3, 4096, 2), (6, 4096, 4), (16, 4096, 3), (1, 11008, 1),
                                   (5, 1024, 8), (2, 512, 1)])
def test_oneshot_allreduce_matches_sum(dev, world, shape):
    from eventgpt_amd.parallel.comm import VirtualGroup, plan
    M, N, ns = shape
    blocks, chunk, bpr = plan(M, N)
    vg = VirtualGroup(world, dev)
    g = torch.Generator(device=dev).manual_seed(7 * world + M)
    norm_w = (torch.rand(N, device=dev, generator=g) + 0.5).to(torch.bfloat16) if bpr == 1 else None
    for it in range(5):                     # epochs advance, the buffer parity flips
        parts = [torch.randn(ns, M, N, device=dev, generator=g) for _ in range(world)]
        res = (torch.randn(M, N, device=dev, generator=g)).to(torch.bfloat16)
        xs = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(world)]
        xns = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(world)] if bpr == 1 else None
        sums = [torch.empty(M, N, device=dev) for _ in range(world)]
        assert vg.reduce(parts, res=[res] * world, x_out=xs, norm_w=norm_w, xn_out=xns, sum_out=sums, eps=1e-6)
        torch.cuda.synchronize()
        vg.check()
        total, x, xn = _oracle(parts, res, norm_w, 1e-6)
        for r in range(world):
            assert torch.equal(sums[r], sums[0]), (it, r)            # bit-identical on every rank
            assert torch.allclose(sums[r], total, rtol=1e-5, atol=1e-4), (it, r)
            # bf16 outputs: the oracle rounds the same fp32 sum, so at most one ulp apart
            assert torch.allclose(xs[r].float(), x.float(), rtol=1e-2, atol=1e-2), (it, r)
            assert torch.equal(xs[r], xs[0])
            if xn is not None:
                assert torch.allclose(xns[r].float(), xn.float(), rtol=2e-2, atol=2e-2), (it, r)


def test_oneshot_allreduce_32_rows_four_ranks(dev):
    """The batched-decode message (32 x 4096 fp32 = 512 KB per rank), in place on the residual."""
    from eventgpt_amd.parallel.comm import VirtualGroup
    vg = VirtualGroup(4, dev)
    g = torch.Generator(device=dev).manual_seed(5)
    parts = [torch.randn(2, 32, 4096, device=dev, generator=g) for _ in range(4)]
    x0 = torch.randn(32, 4096, device=dev, generator=g).to(torch.bfloat16)
    xs = [x0.clone() for _ in range(4)]
    assert vg.reduce(parts, res=xs, x_out=xs)            # x += sum, in place
    torch.cuda.synchronize()
    vg.check()
    _, want, _ = _oracle(parts, x0, None, 1e-6)
    for r in range(4):
        assert torch.allclose(xs[r].float(), want.float(), rtol=1e-2, atol=1e-2)


def test_oneshot_allreduce_graph_replays(dev):
    """Three collectives of different sizes captured in one graph; every replay (with the
    program's own eager work and synchronisation in between) matches the eager result."""
    from eventgpt_amd.parallel.comm import VirtualGroup
    W = 4
    vg = VirtualGroup(W, dev)
    g = torch.Generator(device=dev).manual_seed(11)
    shapes = [(1, 4096, 1), (6, 4096, 2), (2, 4096, 1)]
    parts = [[torch.randn(ns, M, N, device=dev, generator=g) for _ in range(W)] for (M, N, ns) in shapes]
    outs = [[torch.zeros(M, N, device=dev) for _ in range(W)] for (M, N, ns) in shapes]

    def body():
        for p, o in zip(parts, outs):
            assert vg.reduce(p, sum_out=o)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        body()
    for rep in range(4):
        for ps in parts:                 # new data between replays (eager work)
            for p in ps:
                p.normal_(generator=g)
        for os_ in outs:
            for o in os_:
                o.fill_(-1.0)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        vg.check()
        for ps, os_ in zip(parts, outs):
            want = sum(p.sum(0) for p in ps)
            for o in os_:
                assert torch.allclose(o, want, rtol=1e-5, atol=1e-4), rep


def test_oneshot_allreduce_wait_gives_up(dev):
    """Only rank 0 of a 2-rank group is driven, so its peer's flag never arrives: with a tiny
    spin limit the wait gives up, the kernel exits normally and the host sees the error."""
    from eventgpt_amd.parallel.comm import VirtualGroup
    vg = VirtualGroup(2, dev, spin_limit=64)
    parts = [torch.ones(1, 1, 4096, device=dev) for _ in range(2)]
    out = [torch.zeros(1, 4096, device=dev) for _ in range(2)]
    assert vg.reduce(parts, sum_out=out, ranks=[0])
    torch.cuda.synchronize()                  # the launch has drained
    with pytest.raises(RuntimeError, match="timed out"):
        vg.check()
    vg.clear_errors()
    vg.check()


def test_message_that_does_not_fit_is_refused(dev):
    from eventgpt_amd.parallel.comm import VirtualGroup, plan
    assert plan(65, 4096) is None and plan(1, 4096 * 9 * 8) is None
    vg = VirtualGroup(2, dev)
    parts = [torch.ones(1, 65, 4096, device=dev) for _ in range(2)]
    assert vg.reduce(parts, sum_out=[torch.zeros(65, 4096, device=dev) for _ in range(2)]) is False


def test_sum_slabs_and_gemv_partial(dev):
    from eventgpt_amd.ops import hip
    g = torch.Generator(device=dev).manual_seed(3)
    p = torch.randn(5, 7, 4096, device=dev, generator=g)
    assert torch.allclose(hip.raw.sum_slabs(p)[0], p.sum(0), rtol=1e-6, atol=1e-5)
    w = (torch.randn(4096, 1408, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    x = torch.randn(3, 1408, device=dev, generator=g).to(torch.bfloat16)
    out = torch.zeros(1, 3, 4096, device=dev)
    hip.raw.decode_gemv_partial(w, x, out)
    want = x.float() @ w.float().t()
    assert torch.allclose(out[0], want, rtol=2e-3, atol=2e-3)


def test_argmax_pair_kernels(dev):
    """Per-rank (max, index) pairs and their selection == argmax over the concatenated shards,
    lowest index on ties."""
    from eventgpt_amd.ops import hip
    W, B, P, per = 4, 6, 37, 1000
    g = torch.Generator(device=dev).manual_seed(9)
    pairs = torch.zeros(W, B, 2, dtype=torch.int32, device=dev)
    vals, idxs = [], []
    for r in range(W):
        pv = torch.randn(B, P, device=dev, generator=g)
        pi = torch.randint(0, per, (B, P), device=dev, generator=g, dtype=torch.int32)
        if r == 1:
            pv[0, 5] = 50.0; pi[0, 5] = 9
        if r == 2:
            pv[0, 3] = 50.0; pi[0, 3] = 4       # tie with rank 1: the lower global index wins
        hip.raw.argmax_pair_local(pv.contiguous(), pi.contiguous(), B, P, r * per, pairs[r])
        vals.append(pv); idxs.append(pi.long() + r * per)
    tok = torch.zeros(B, dtype=torch.int32, device=dev)
    hip.raw.argmax_pair_select(pairs, W, B, tok)
    v = torch.cat(vals, 1); i = torch.cat(idxs, 1)
    best = v.max(1, keepdim=True).values
    want = torch.where(v == best, i, torch.full_like(i, 1 << 40)).min(1).values
    assert tok.long().tolist() == want.tolist()
    assert int(tok[0]) == 1 * per + 9


_RCCL_CAPTURE = r"""
import os, sys, json
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1])
import torch, torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(4, 4096, device=dev)
pair = torch.arange(8, device=dev, dtype=torch.int32)
pairs = torch.zeros(8, device=dev, dtype=torch.int32)
dist.all_reduce(t)                      # communicator comes up before any capture
dist.all_gather_into_tensor(pairs, pair)
torch.cuda.synchronize()
src = torch.zeros(4, 4096, device=dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    t.copy_(src)
    t.mul_(2.0)
    dist.all_reduce(t)
    dist.all_gather_into_tensor(pairs, pair)
    out = t + 1.0
ok = []
for i in range(4):
    src.fill_(float(i))
    pair.fill_(i)
    g.replay()
    torch.cuda.synchronize()
    ok.append(bool((out == 2.0 * i + 1.0).all()) and bool((pairs == i).all()))
    _ = torch.randn(64, device=dev).sum().item()       # eager work between the replays
print(json.dumps({"ok": ok, "backend": dist.get_backend()}))
dist.destroy_process_group()
"""


@pytest.mark.timeout(240)
def test_rccl_collectives_record_into_a_graph_single_rank():
    """The library collectives of the tensor-parallel decode step (fp32 all-reduce, int32
    all-gather) are recorded into a HIP graph and replayed, on the RCCL backend itself; one rank is
    all a one-GPU box can form, so this pins the capture path, not the fabric."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    print("launching single-rank RCCL capture worker", flush=True)
    r = subprocess.run([sys.executable, "-c", _RCCL_CAPTURE, "29561"], env=env, capture_output=True, text=True,
                       timeout=220)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2500:])
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["backend"] == "nccl" and got["ok"] == [True] * 4, got


def _bf16_sum_oracle(xs):
    total = torch.zeros_like(xs[0], dtype=torch.float32)
    for x in xs:                            # rank order, fp32: the kernel's order
        total = total + x.float()
    return total.to(torch.bfloat16)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("elems", [8, 4096, 636 * 64, 100_000, 3 * 2048 * 8])
def test_twoshot_allreduce_matches_sum(dev, world, elems):
    """Reduce-scatter + all-gather by direct pushes: every rank ends with the bf16 rounding of the fp32
    sum in rank order, bit for bit, for sizes that do and do not fill the last piece; five calls in a
    row flip the buffer parity; out of place and in place."""
    from eventgpt_amd.parallel.comm import VirtualTwoShot
    vg = VirtualTwoShot(world, dev, max_elems=120_000)
    g = torch.Generator(device=dev).manual_seed(11 * world + elems)
    for it in range(5):
        xs = [torch.randn(elems, device=dev, generator=g).to(torch.bfloat16) for _ in range(world)]
        want = _bf16_sum_oracle(xs)
        if it % 2 == 0:
            outs = [torch.empty_like(x) for x in xs]
            assert vg.reduce(xs, outs)
        else:
            outs = [x.clone() for x in xs]
            assert vg.reduce(outs)          # in place
        torch.cuda.synchronize()
        vg.check()
        for r in range(world):
            assert torch.equal(outs[r], want), (it, r, (outs[r].float() - want.float()).abs().max().item())


def test_twoshot_allreduce_graph_replays_and_refusal(dev):
    """A captured two-shot call replayed with new inputs (eager work and synchronisation in between)
    matches the oracle every time; a message beyond the buffers is refused without a launch."""
    from eventgpt_amd.parallel.comm import VirtualTwoShot
    world, elems = 4, 50_000
    vg = VirtualTwoShot(world, dev, max_elems=elems)
    g = torch.Generator(device=dev).manual_seed(3)
    xs = [torch.zeros(elems, dtype=torch.bfloat16, device=dev) for _ in range(world)]
    outs = [torch.empty_like(x) for x in xs]
    assert vg.reduce(xs, outs)              # warm-up outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        assert vg.reduce(xs, outs)
    for rep in range(4):
        fresh = [torch.randn(elems, device=dev, generator=g).to(torch.bfloat16) for _ in range(world)]
        for x, f in zip(xs, fresh):
            x.copy_(f)
        graph.replay()
        torch.cuda.synchronize()
        vg.check()
        want = _bf16_sum_oracle(fresh)
        for r in range(world):
            assert torch.equal(outs[r], want), (rep, r)
        _ = torch.randn(256, device=dev).sum().item()
    too_big = [torch.zeros(elems * 3, dtype=torch.bfloat16, device=dev) for _ in range(world)]
    assert vg.reduce(too_big) is False


def test_twoshot_allreduce_wait_gives_up(dev):
    """Only rank 0 of two is driven: its waits for the peer's scatter and gather flags give up after a
    tiny spin limit, the kernel exits normally and the host sees the error."""
    from eventgpt_amd.parallel.comm import VirtualTwoShot
    vg = VirtualTwoShot(2, dev, max_elems=8192, spin_limit=64)
    xs = [torch.ones(4096, dtype=torch.bfloat16, device=dev) for _ in range(2)]
    assert vg.reduce(xs, ranks=[0])
    torch.cuda.synchronize()                  # the launch has drained
    with pytest.raises(RuntimeError, match="timed out"):
        vg.check()
    vg.clear_errors()
    vg.check()
