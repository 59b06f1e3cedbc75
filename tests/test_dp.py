"""Data-parallel sharded optimizer: 2 gloo ranks == single-process AdamW on the full batch."""
import json
import os
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _reference(world=2):
    sys.path.insert(0, HERE)
    from _dp_worker import make_model
    model = make_model()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.01)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(6, 8, 12, generator=g)
    y = torch.randn(6, 8, 7, generator=g)
    norms = []
    for step in range(6):
        opt.zero_grad()
        # mean of the per-rank batch means == what averaging rank gradients computes
        loss = sum((model(x[step, r::world]) - y[step, r::world]).pow(2).mean() for r in range(world)) / world
        loss.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)))
        opt.step()
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]), norms


@pytest.mark.timeout(180)
def test_sharded_adamw_two_ranks_matches_adamw(tmp_path):
    out = str(tmp_path / "dp.json")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(HERE, "_dp_worker.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.load(open(out))
    want, norms = _reference()
    assert got["ranks_agree"] and got["n_buckets"] >= 2
    # one gradient collective per bucket and step, also on the steps that read the norm before stepping
    assert got["reduces_launched"] == 6 * got["n_buckets"], got["reduces_launched"]
    assert torch.allclose(torch.tensor(got["params"]), want, atol=2e-5), \
        (torch.tensor(got["params"]) - want).abs().max()
    assert torch.allclose(torch.tensor(got["norms"]), torch.tensor(norms), rtol=1e-4)


@pytest.mark.timeout(240)
@pytest.mark.parametrize("mode", ["straight", "resume"])
def test_sharded_adamw_four_ranks_uneven_buckets_and_resume(tmp_path, mode):
    """Four gloo ranks, buckets whose length is no multiple of 4 (padded shards), with and without a
    stop / rebuild-from-shards / continue in the middle: both equal single-process AdamW."""
    out = str(tmp_path / "dp4.json")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4",
           "--master-addr", "127.0.0.1", "--master-port", "29537" if mode == "straight" else "29539",
           os.path.join(HERE, "_dp_worker.py"), out] + (["resume"] if mode == "resume" else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=230)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.load(open(out))
    want, norms = _reference(4)
    assert got["ranks_agree"] and got["world"] == 4 and max(got["padding"]) > 0
    assert torch.allclose(torch.tensor(got["params"]), want, atol=2e-5), \
        (torch.tensor(got["params"]) - want).abs().max()
    assert torch.allclose(torch.tensor(got["norms"]), torch.tensor(norms), rtol=1e-4)


def test_sharded_adamw_single_process_equals_adamw():
    from eventgpt_amd.parallel.dp import ShardedAdamW
    torch.manual_seed(0)
    a = torch.nn.Linear(5, 4)
    b = torch.nn.Linear(5, 4)
    b.load_state_dict(a.state_dict())
    oa = torch.optim.AdamW(a.parameters(), lr=1e-2, weight_decay=0.01)
    ob = ShardedAdamW(b.parameters(), lr=1e-2, weight_decay=0.01)
    x = torch.randn(3, 5)
    for _ in range(4):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad()
            m(x).pow(2).sum().backward()
            o.step()
    assert torch.allclose(a.weight, b.weight, atol=1e-6) and torch.allclose(a.bias, b.bias, atol=1e-6)
    sd = ob.state_dict()
    ob.load_state_dict(sd)
    assert sd["step"] == 4
