"""Training kernels: the MFMA GEMM with K-strided operands (dgrad / wgrad of a linear layer) and
the flash-attention backward, each against a plain fp32 PyTorch computation of the same thing."""
import math

import pytest
import torch

from eventgpt_amd.ops import reference as ref

pytestmark = pytest.mark.gpu


def _rand(*shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed + sum(shape))
    return (torch.randn(*shape, device=dev, generator=g) * scale).to(torch.bfloat16)


def _close(a, b, atol, rtol):
    a, b = a.float(), b.float()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"max err {err.max().item():.4g}, worst excess {(err - tol).max().item():.4g}"


@pytest.mark.parametrize("M,N,K", [(840, 4096, 4096), (636, 11008, 4096), (64, 4096, 1024), (200, 16, 4096),
                                   (130, 4096, 16), (577, 1000, 1024), (2115, 4096, 11008),
                                   (836, 12288, 4096), (636, 22016, 4096)])   # dgrad splits its long contraction
def test_dgrad_and_wgrad_match_fp32(dev, M, N, K):
    """dX = dY W and dW = dY^T X through the K-strided operand paths (any M, padded contraction)."""
    from eventgpt_amd.ops import differentiable as diff
    gy, w, x = _rand(M, N, dev=dev, scale=0.5), _rand(N, K, dev=dev, seed=1, scale=0.05), _rand(M, K, dev=dev, seed=2)
    dx = diff.linear_dgrad(gy, w)
    assert dx.shape == (M, K)
    _close(dx, gy.float() @ w.float(), 2e-2 * math.sqrt(N / 64), 2e-2)
    dw = diff.linear_wgrad(gy, x)
    assert dw.shape == (N, K)
    _close(dw, gy.float().t() @ x.float(), 2e-2 * math.sqrt(M / 16), 2e-2)


@pytest.mark.parametrize("M,N,K", [(811, 4096, 16), (6, 4096, 16), (33, 64, 24), (130, 512, 72), (640, 1024, 200),
                                   (811, 16, 4104), (6, 4096, 4104), (1030, 4096, 136)])
def test_contraction_that_is_not_a_multiple_of_64(dev, M, N, K):
    """K-contiguous operands whose rows end inside the last K-tile (low-rank products: K = 16): the
    kernel stages zeros past the rows' end, so the result equals the zero-padded product bit for bit
    and no operand is copied.  Every element is non-zero, so a chunk taken from the next row shows."""
    from eventgpt_amd.ops import hip
    x = _rand(M, K, dev=dev) + 0.25
    w = _rand(N, K, dev=dev, seed=1, scale=0.1) + 0.125
    res = _rand(M, N, dev=dev, seed=2)
    y = hip.raw.linear(x, w, None, res, 0, False)
    _close(y, x.float() @ w.float().t() + res.float(), 3e-2 * math.sqrt(K / 64 + 1), 2e-2)
    if K < 2048:      # same tile configuration and split either way
        pad = (-K) % 64
        yp = hip.raw.linear(torch.nn.functional.pad(x, (0, pad)), torch.nn.functional.pad(w, (0, pad)), None, res, 0, False)
        assert torch.equal(y, yp)
    # dX = dY W with dY's columns ending inside the K-tile (contraction over N here)
    if N % 8 == 0 and N % 64:
        from eventgpt_amd.ops import differentiable as diff
        gy = _rand(M, N, dev=dev, seed=3) + 0.25
        _close(diff.linear_dgrad(gy, w), gy.float() @ w.float(), 3e-2 * math.sqrt(N / 64 + 1), 2e-2)


def test_kstrided_operand_asymmetric_identity(dev):
    """A = I with an asymmetric other operand: a swapped row/column map or a wrong k order inside
    the transposed fragments shows up as exact mismatches."""
    from eventgpt_amd.ops import hip
    n = 256
    eye = torch.eye(n, device=dev, dtype=torch.bfloat16)
    w = (torch.arange(n * 128, device=dev).reshape(n, 128) % 251).to(torch.bfloat16)      # [K=n rows][N=128]
    y = hip.raw.linear_t(eye, w, False, True, None, None, 0, False, n)                     # I @ w
    assert torch.equal(y, w)
    a = (torch.arange(n * 192, device=dev).reshape(n, 192) % 241).to(torch.bfloat16)      # [K=n][M=192]
    y2 = hip.raw.linear_t(a, w, True, True, None, None, 0, True, n)                        # a^T @ w, fp32 out
    assert torch.equal(y2, a.float().t() @ w.float())
    y3 = hip.raw.linear_t(a, eye, True, False, None, None, 0, False, n)                    # a^T @ I^T
    assert torch.equal(y3, a.t())


@pytest.mark.parametrize("shape", [(3, 280, 1024, 4096), (1, 840, 4096, 512), (2, 50, 512, 24)])
def test_linear_function_gradients(dev, shape):
    """autograd through ops.linear (in-tree forward, dgrad, wgrad) vs autograd through fp32 matmul."""
    from eventgpt_amd import ops
    B, S, K, N = shape
    x = _rand(B, S, K, dev=dev).requires_grad_(True)
    w = _rand(N, K, dev=dev, seed=1, scale=0.05).requires_grad_(True)
    bias = _rand(N, dev=dev, seed=2).requires_grad_(True)
    gy = _rand(B, S, N, dev=dev, seed=3)
    y = ops.linear(x, w, bias)
    assert type(y.grad_fn).__name__ == "_LinearBackward", type(y.grad_fn).__name__
    y.backward(gy)
    xf, wf, bf = (t.detach().float().requires_grad_(True) for t in (x, w, bias))
    yf = xf @ wf.t() + bf
    yf.backward(gy.float())
    _close(y, yf, 3e-2, 2e-2)
    _close(x.grad, xf.grad, 2e-2 * math.sqrt(N / 64), 2e-2)
    _close(w.grad, wf.grad, 2e-2 * math.sqrt(B * S / 16), 2e-2)
    _close(bias.grad, bf.grad, 2e-2 * math.sqrt(B * S / 16), 2e-2)


def test_frozen_weight_skips_wgrad(dev):
    from eventgpt_amd import ops
    x = _rand(70, 512, dev=dev).requires_grad_(True)
    w = _rand(256, 512, dev=dev, seed=1, scale=0.05)
    ops.linear(x, w).sum().backward()
    assert w.grad is None and x.grad is not None


def test_frozen_weight_dgrad_through_transposed_copy(dev):
    """Frozen matrices may carry a row-major copy of their transpose; the input gradient then runs the
    plain GEMM on it.  Same numbers as the fp32 product, also after the weight is written in place."""
    from eventgpt_amd.models.llama import Linear
    from eventgpt_amd.ops.differentiable import enable_transposed_dgrad
    lin = Linear(512, 1088, bias=False).to(dev, torch.bfloat16)
    train = Linear(512, 64, bias=False).to(dev, torch.bfloat16)
    lin.weight.requires_grad_(False)
    mod = torch.nn.ModuleList([lin, train])
    spent = enable_transposed_dgrad(mod, min_free_bytes=0)
    assert spent == lin.weight.numel() * 2 and not hasattr(train.weight, "_dgrad_t")
    g = _rand(836, 1088, dev=dev, seed=3)
    for rnd in range(2):
        x = _rand(836, 512, dev=dev, seed=2).requires_grad_(True)
        lin(x).backward(g)
        want = g.float() @ lin.weight.float()
        _close(x.grad, want, 2e-2 * math.sqrt(1088 / 64), 2e-2)
        assert torch.equal(lin.weight._dgrad_t[0], lin.weight.t())
        with torch.no_grad():
            lin.weight.mul_(0.5)        # the copy is stale now; the next backward refreshes it
    from eventgpt_amd.ops.differentiable import release_transposed_dgrad
    assert release_transposed_dgrad(mod) == spent and not hasattr(lin.weight, "_dgrad_t")


@pytest.mark.parametrize("M,R,K", [(836, 16, 4096), (5, 16, 4096), (16, 8, 4096), (17, 16, 1376), (1030, 16, 11008),
                                   (33, 3, 72), (836, 16, 8)])
def test_rank_r_projection_matches_fp32(dev, M, R, K):
    """alpha * X W^T for R <= 16 output columns: rows past M, W rows past R and K-steps past K are
    staged as zeros (every element is non-zero, so a chunk taken from a neighbour shows)."""
    from eventgpt_amd.ops import hip
    x = _rand(M, K, dev=dev) + 0.25
    w = _rand(R, K, dev=dev, seed=1, scale=0.1) + 0.125
    y = hip.raw.lowrank_out(x, w, 2.0)
    assert y.shape == (M, R) and y.dtype == torch.bfloat16
    _close(y, 2.0 * (x.float() @ w.float().t()), 3e-2 * math.sqrt(K / 64 + 1), 2e-2)
    y3 = hip.raw.lowrank_out(x.view(1, M, K), w, 2.0)
    assert y3.shape == (1, M, R) and torch.equal(y3.view(M, R), y)


@pytest.mark.parametrize("M", [836, 6])
def test_low_rank_update_gradients(dev, M):
    """(s x A^T) B^T as one autograd node: output and the three gradients against an fp32 computation
    with the same two bf16 intermediates (t = s x A^T forward, dt = s dY B backward)."""
    from eventgpt_amd import ops
    K, N, r, s = 4096, 4096, 16, 2.0
    x = _rand(1, M, K, dev=dev).requires_grad_(True)
    a = _rand(r, K, dev=dev, seed=1, scale=0.02).requires_grad_(True)
    b = _rand(N, r, dev=dev, seed=2, scale=0.05).requires_grad_(True)
    go = _rand(1, M, N, dev=dev, seed=3)
    y = ops.lowrank_delta(x, a, b, s)
    assert type(y.grad_fn).__name__ == "_LowRankBackward"
    y.backward(go)
    xf, af, bf, gf = (t.detach().float().view(-1, t.shape[-1]) for t in (x, a, b, go))
    t = (s * (xf @ af.t())).bfloat16().float()
    dt = (s * (gf @ bf)).bfloat16().float()
    yf = t @ bf.t()
    _close(y.view(M, N), yf, 2e-2, 2e-2)
    _close(x.grad.view(M, K), dt @ af, 2e-2, 2e-2)
    _close(a.grad, dt.t() @ xf, 2e-2 * math.sqrt(M / 16), 2e-2)
    _close(b.grad, gf.t() @ t, 2e-2 * math.sqrt(M / 16), 2e-2)
    with torch.no_grad():      # inference takes the same two kernels
        _close(ops.lowrank_delta(x.detach(), a.detach(), b.detach(), s).view(M, N), yf, 2e-2, 2e-2)


def _attn_ref_grads(q, k, v, go, causal, q_offset):
    qf, kf, vf = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    rep = q.shape[2] // k.shape[2]          # grouped KV heads: the oracle takes them expanded
    ke, ve = (kf.repeat_interleave(rep, 2), vf.repeat_interleave(rep, 2)) if rep > 1 else (kf, vf)
    o = ref.attention(qf, ke, ve, causal, None, q_offset)
    o.backward(go.float())
    return o, qf.grad, kf.grad, vf.grad


def test_frozen_gate_up_grouped_in_one_matrix(dev):
    """Frozen gate / up projections re-homed as row-slices of one matrix: same values and gradients as
    the two separate products, one forward GEMM and one input-gradient GEMM, one transposed copy for the
    group; a member that trains again sends the module back to the separate path."""
    from types import SimpleNamespace
    from eventgpt_amd.models.llama import LlamaMLP, group_intact
    from eventgpt_amd.ops.differentiable import enable_transposed_dgrad, release_transposed_dgrad
    torch.manual_seed(0)
    mlp = LlamaMLP(SimpleNamespace(hidden_size=512, intermediate_size=1408)).to(dev, torch.bfloat16)
    for p in mlp.parameters():
        p.requires_grad_(False)
    before = {k: v.clone() for k, v in mlp.state_dict().items()}
    x = _rand(2, 150, 512, dev=dev).requires_grad_(True)
    gy = _rand(2, 150, 512, dev=dev, seed=5)
    y0 = mlp(x)
    y0.backward(gy)
    dx0, x.grad = x.grad.clone(), None

    spent = enable_transposed_dgrad(mlp, min_free_bytes=0)
    w = mlp.__dict__["_w_gate_up"]
    assert w.shape == (2 * 1408, 512) and group_intact(w, (mlp.gate_proj, mlp.up_proj), x)
    assert mlp.gate_proj.weight.data_ptr() == w.data_ptr() and mlp.up_proj.weight.data_ptr() == w[1408:].data_ptr()
    assert all(torch.equal(v, before[k]) for k, v in mlp.state_dict().items())
    assert spent == 2 * (w.numel() + mlp.down_proj.weight.numel())        # one copy for the group, one for down
    assert getattr(mlp.gate_proj.weight, "_dgrad_t", None) is None and w._dgrad_t[0].shape == (512, 2 * 1408)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        y1 = mlp(x)
        y1.backward(gy)
    names = [e.name for e in prof.events()]
    assert names.count("evgpt::linear") == 4 and "aten::cat" not in names and "evgpt::linear_t" not in names, \
        sorted(set(n for n in names if "evgpt" in n or "cat" in n))
    _close(y1, y0, 2e-2, 2e-2)
    _close(x.grad, dx0, 2e-2, 2e-2)
    # an in-place write through a member refreshes the group's transposed copy
    with torch.no_grad():
        mlp.up_proj.weight.mul_(0.5)
    x.grad = None
    mlp(x).backward(gy)
    xf = x.detach().float().requires_grad_(True)
    g, u = xf @ mlp.gate_proj.weight.float().t(), xf @ mlp.up_proj.weight.float().t()
    ((torch.nn.functional.silu(g) * u) @ mlp.down_proj.weight.float().t()).backward(gy.float())
    _close(x.grad, xf.grad, 3e-2, 3e-2)
    assert release_transposed_dgrad(mlp, ungroup=False) == spent and getattr(w, "_dgrad_t", None) is None
    mlp.up_proj.weight.requires_grad_(True)             # trains again: separate products
    assert not group_intact(w, (mlp.gate_proj, mlp.up_proj), x)
    x.grad = None
    mlp(x).backward(gy)
    assert mlp.up_proj.weight.grad is not None and mlp.up_proj.weight.grad.shape == (1408, 512)
    release_transposed_dgrad(mlp)                       # dissolves the group: own storage again, same values
    assert "_w_gate_up" not in mlp.__dict__ and mlp.up_proj.weight.data_ptr() != w[1408:].data_ptr()
    assert torch.equal(mlp.gate_proj.weight, before["gate_proj.weight"]) and torch.equal(mlp.up_proj.weight, w[1408:])


@pytest.mark.parametrize("hq,hkv,d", [(8, 8, 128), (8, 2, 64)])
def test_rope_qkv_split_and_its_backward(dev, hq, hkv, d):
    """One launch splits a packed q|k|v projection and rotates q and k: bit-equal to the separate RoPE
    kernel on the slices; the backward (one launch) equals rotating each gradient back and concatenating."""
    from eventgpt_amd import ops
    from eventgpt_amd.ops import hip
    b, s = 2, 75
    qkv = _rand(b, s, (hq + 2 * hkv) * d, dev=dev)
    pos = (torch.arange(s, device=dev)[None] + torch.tensor([[3], [11]], device=dev)).contiguous()
    inv = 1.0 / (10000 ** (torch.arange(0, d, 2, device=dev).float() / d))
    ang = torch.arange(256, device=dev).float()[:, None] * inv[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    q, k, v = ops.rope_qkv_split(qkv, pos, cos, sin, hq, hkv, d)
    qs, ks, vs = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    assert torch.equal(q, hip.raw.rope(qs.reshape(b, s, hq, d).contiguous(), pos, cos, sin))
    assert torch.equal(k, hip.raw.rope(ks.reshape(b, s, hkv, d).contiguous(), pos, cos, sin))
    assert torch.equal(v, vs.reshape(b, s, hkv, d))
    x = qkv.clone().requires_grad_(True)
    q2, k2, v2 = ops.rope_qkv_split(x, pos, cos, sin, hq, hkv, d)
    assert type(q2.grad_fn).__name__ == "_RopeQKVBackward" and torch.equal(q2, q)
    gq, gk, gv = _rand(*q.shape, dev=dev, seed=1), _rand(*k.shape, dev=dev, seed=2), _rand(*v.shape, dev=dev, seed=3)
    torch.autograd.backward([q2, k2, v2], [gq, gk, gv])
    want = torch.cat([hip.raw.rope(gq, pos, cos, -sin).reshape(b, s, -1), hip.raw.rope(gk, pos, cos, -sin).reshape(b, s, -1),
                      gv.reshape(b, s, -1)], dim=-1)
    assert torch.equal(x.grad, want)
    # and against plain fp32 arithmetic
    xf = qkv.float().requires_grad_(True)
    qf = xf[..., :hq * d].reshape(b, s, hq, d)
    c, sn = cos[pos].unsqueeze(2), sin[pos].unsqueeze(2)
    q1, q2h = qf[..., :d // 2], qf[..., d // 2:]
    qr = torch.cat([q1 * c - q2h * sn, q2h * c + q1 * sn], dim=-1)
    _close(q, qr, 2e-2, 1e-2)
    qr.backward(gq.float())
    _close(x.grad[..., :hq * d], xf.grad[..., :hq * d], 2e-2, 1e-2)


def test_frozen_qkv_grouped_in_one_matrix(dev):
    """The attention block over grouped frozen q / k / v weights: same output and input gradient as the
    three separate projections, with one projection GEMM and one split+RoPE launch each way."""
    from types import SimpleNamespace
    from eventgpt_amd.models.llama import LlamaAttention
    from eventgpt_amd.ops.differentiable import enable_transposed_dgrad, release_transposed_dgrad
    torch.manual_seed(0)
    cfg = SimpleNamespace(hidden_size=512, num_attention_heads=4, num_key_value_heads=4)
    att = LlamaAttention(cfg).to(dev, torch.bfloat16)
    for p in att.parameters():
        p.requires_grad_(False)
    b, s, d = 2, 90, 128
    x = (_rand(b, s, 512, dev=dev) * 0.5).requires_grad_(True)
    res = _rand(b, s, 512, dev=dev, seed=7)
    gy = _rand(b, s, 512, dev=dev, seed=5)
    pos = torch.arange(s, device=dev)[None].expand(b, s).contiguous()
    inv = 1.0 / (10000 ** (torch.arange(0, d, 2, device=dev).float() / d))
    ang = torch.arange(128, device=dev).float()[:, None] * inv[None]
    rope = (ang.cos().contiguous(), ang.sin().contiguous())
    y0, _ = att(x, pos, rope, None, None, res)
    y0.backward(gy)
    dx0, x.grad = x.grad.clone(), None
    enable_transposed_dgrad(att, min_free_bytes=0)
    w = att.__dict__["_w_qkv"]
    assert w.shape == (3 * 512, 512) and att.v_proj.weight.data_ptr() == w[1024:].data_ptr()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        y1, kv = att(x, pos, rope, None, None, res)
        y1.backward(gy)
    names = [e.name for e in prof.events()]
    assert names.count("evgpt::rope_qkv_split") == 1 and names.count("evgpt::rope_qkv_merge") == 1
    assert names.count("evgpt::linear") == 4 and "evgpt::rope" not in names, sorted(set(n for n in names if "evgpt" in n))
    assert kv[0].shape == (b, 4, s, d)
    _close(y1, y0, 2e-2, 2e-2)
    _close(x.grad, dx0, 3e-2, 2e-2)
    with torch.no_grad():                       # the inference path (no autograd) takes the group as well
        y2, _ = att(x.detach(), pos, rope, None, None, res)
    _close(y2, y0, 2e-2, 2e-2)
    release_transposed_dgrad(att)


def test_low_rank_wrappers_over_grouped_qkv(dev):
    """LoRA-wrapped q / k / v whose frozen bases are grouped: the base projection is one GEMM, the
    low-rank updates are added onto their column ranges; output, input gradient and the adapters'
    gradients equal the wrapped layers run one by one."""
    from types import SimpleNamespace
    from eventgpt_amd.align.lora import apply_lora, lora_parameters
    from eventgpt_amd.models.llama import LlamaAttention
    torch.manual_seed(0)
    att = LlamaAttention(SimpleNamespace(hidden_size=512, num_attention_heads=4, num_key_value_heads=4)).to(dev, torch.bfloat16)
    apply_lora(att, r=16, alpha=32, dropout=0.0)
    with torch.no_grad():
        for m in (att.q_proj, att.k_proj, att.v_proj, att.o_proj):
            m.lora_B.weight.copy_(_rand(*m.lora_B.weight.shape, dev=dev, seed=9, scale=0.05))
    b, s, d = 2, 70, 128
    x = (_rand(b, s, 512, dev=dev) * 0.5).requires_grad_(True)
    res, gy = _rand(b, s, 512, dev=dev, seed=7), _rand(b, s, 512, dev=dev, seed=5)
    pos = torch.arange(s, device=dev)[None].expand(b, s).contiguous()
    inv = 1.0 / (10000 ** (torch.arange(0, d, 2, device=dev).float() / d))
    ang = torch.arange(128, device=dev).float()[:, None] * inv[None]
    rope = (ang.cos().contiguous(), ang.sin().contiguous())
    y0, _ = att(x, pos, rope, None, None, res)
    y0.backward(gy)
    want = [x.grad.clone()] + [p.grad.clone() for p in lora_parameters(att)]
    x.grad = None
    for p in lora_parameters(att):
        p.grad = None
    assert att.group_frozen() and att.__dict__["_w_qkv"].shape == (1536, 512)
    assert att.k_proj.base.weight.data_ptr() == att.__dict__["_w_qkv"][512:].data_ptr()
    y1, _ = att(x, pos, rope, None, None, res)
    y1.backward(gy)
    got = [x.grad] + [p.grad for p in lora_parameters(att)]
    _close(y1, y0, 2e-2, 2e-2)
    for g, w in zip(got, want):
        assert g is not None and g.shape == w.shape
        _close(g, w, 3e-2 * max(1.0, float(w.float().abs().max())), 3e-2)
    att.q_proj.merge()                              # merged: the base weight (a row-slice of the group) holds the update
    with torch.no_grad():
        y2, _ = att(x.detach(), pos, rope, None, None, res)
    _close(y2, y0, 3e-2, 3e-2)


@pytest.mark.parametrize("D,H,Hkv,Sq,Sk,causal", [
    (128, 4, 4, 200, 200, True), (128, 32, 32, 840, 840, True), (64, 16, 16, 577, 577, False),
    (128, 8, 2, 130, 130, True), (64, 4, 4, 64, 64, True), (128, 4, 4, 70, 300, True), (64, 2, 1, 33, 90, False)])
def test_attention_backward(dev, D, H, Hkv, Sq, Sk, causal):
    from eventgpt_amd import ops
    B = 2 if Sq < 500 else 1
    q = _rand(B, Sq, H, D, dev=dev).requires_grad_(True)
    k = _rand(B, Sk, Hkv, D, dev=dev, seed=1).requires_grad_(True)
    v = _rand(B, Sk, Hkv, D, dev=dev, seed=2).requires_grad_(True)
    go = _rand(B, Sq, H, D, dev=dev, seed=3)
    q_offset = Sk - Sq if causal else 0
    o = ops.attention(q, k, v, causal, None, q_offset)
    assert type(o.grad_fn).__name__ == "_AttentionBackward", type(o.grad_fn).__name__
    o.backward(go)
    o_ref, dq, dk, dv = _attn_ref_grads(q, k, v, go, causal, q_offset)
    _close(o, o_ref, 2e-2, 2e-2)
    _close(q.grad, dq, 3e-2, 3e-2)
    _close(k.grad, dk, 3e-2, 3e-2)
    _close(v.grad, dv, 3e-2, 3e-2)


def test_attention_lse(dev):
    from eventgpt_amd.ops import hip
    q, k, v = _rand(1, 100, 2, 128, dev=dev), _rand(1, 100, 2, 128, dev=dev, seed=1), _rand(1, 100, 2, 128, dev=dev, seed=2)
    scale = 1 / math.sqrt(128)
    _, lse = hip.raw.attention_fwd_lse(q, k, v, True, scale, 0)
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * scale
    s = s.masked_fill(~torch.ones(100, 100, dtype=torch.bool, device=dev).tril(), float("-inf"))
    want = torch.logsumexp(s, -1) * math.log2(math.e)          # the kernel keeps it in log2 units
    assert torch.allclose(lse, want, rtol=1e-3, atol=1e-2)


def test_decoder_layer_training_step_uses_only_intree_kernels(dev):
    """Forward + backward of a small LLaMA block under the profiler: every GEMM and attention
    kernel that runs is an evgpt:: kernel (no Cijk_* Tensile GEMM, no AOTriton attention)."""
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    cfg = EventChatConfig(vocab_size=512, hidden_size=512, intermediate_size=1408, num_hidden_layers=2,
                          num_attention_heads=4, max_position_embeddings=512)
    model = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=0)
    for p in model.parameters():
        p.requires_grad_(True)
    emb = _rand(1, 130, 512, dev=dev).requires_grad_(True)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        out = model.model(inputs_embeds=emb, use_cache=False)
        h = out.last_hidden_state if hasattr(out, "last_hidden_state") else out[0]
        h.float().pow(2).mean().backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert any("gemm_bf16_kernel" in n for n in names), names
    assert any("attn_bwd_dkv_kernel" in n for n in names), names
    bad = [n for n in names if n.startswith("Cijk_") or "attn_fwd" in n or "bwd_kernel_dk_dv" in n]
    assert not bad, bad
    assert emb.grad is not None and torch.isfinite(emb.grad.float()).all()


def test_row_strided_operands_of_the_lora_products(dev):
    """The products that read or write a column block of a wider matrix where it is: the rank-r kernel and
    the K-strided GEMM on a row-strided gradient block, and the GEMM that accumulates onto a block in place."""
    from eventgpt_amd.ops import hip
    raw = hip.raw
    M, N, r = 300, 1024, 16
    wide = _rand(M, N, dev=dev)
    blk = wide[:, 256:768]                                     # rows strided by 1024
    b_t = _rand(r, 512, dev=dev, seed=1, scale=0.1)
    assert torch.equal(raw.lowrank_out(blk, b_t, 1.5), raw.lowrank_out(blk.contiguous(), b_t, 1.5))
    t = _rand(M, r, dev=dev, seed=2)
    got = raw.linear_t(blk, t, True, True, None, None, 0, False, 320)
    want = raw.linear_t(blk.contiguous(), t, True, True, None, None, 0, False, 320)
    assert torch.equal(got, want)
    _close(got, blk.float().t() @ t.float(), 0.3, 2e-2)
    bw = _rand(512, r, dev=dev, seed=3, scale=0.1)
    y = wide.clone()
    raw.linear_rows_(t, bw, y[:, 256:768], True)
    ref_y = wide.float().clone()
    ref_y[:, 256:768] += (t.float() @ bw.float().t()).bfloat16().float()
    _close(y, ref_y, 2e-2, 1e-2)
    assert torch.equal(y[:, :256], wide[:, :256]) and torch.equal(y[:, 768:], wide[:, 768:])
    z = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    raw.linear_rows_(t, bw, z[:, 512:], False)
    _close(z[:, 512:], t.float() @ bw.float().t(), 2e-2, 1e-2)
    assert float(z[:, :512].abs().max()) == 0.0


def test_lora_input_mask_statistics_and_fused_load(dev):
    """The keep mask of (seed, element index): the right fraction, different per seed, and the rank-r kernel
    that applies it while loading equals the kernel run on the masked copy."""
    from eventgpt_amd.ops import hip
    raw = hip.raw
    M, K, p = 333, 4096, 0.1
    ones = torch.ones(M, K, dtype=torch.bfloat16, device=dev)
    m1, m2 = raw.lowrank_mask(ones, p, 11), raw.lowrank_mask(ones, p, 12)
    assert set(m1.unique().tolist()) == {0.0, 1.0}
    assert abs(float(m1.float().mean()) - (1 - p)) < 0.005 and not torch.equal(m1, m2)
    assert abs(float(m1.float().mean(0).min()) - (1 - p)) < 0.12      # no dead columns or rows
    x = _rand(M, K, dev=dev) + 0.25
    a = _rand(16, K, dev=dev, seed=1, scale=0.05)
    fused = raw.lowrank_out(x, a, 2.0, p, 11)
    plain = raw.lowrank_out(raw.lowrank_mask(x, p, 11), a, 2.0 / (1 - p))
    assert torch.equal(fused, plain)
    acc = _rand(M, K, dev=dev, seed=4)
    want = (acc.float() + (x.float() * m1.float())).bfloat16()
    raw.lowrank_mask_add_(acc, x, p, 11)
    assert torch.equal(acc, want)


@pytest.mark.parametrize("p,single", [(0.0, False), (0.1, False), (0.1, True), (0.0, True)])
def test_lora_projection_node_against_fp32(dev, p, single):
    """Frozen projection + low-rank adapters on column ranges as one autograd node: output, input gradient,
    residual gradient and every adapter gradient against fp32 arithmetic with the same masks and the same
    bf16 rank-r intermediates."""
    from eventgpt_amd import ops
    from eventgpt_amd.ops import differentiable as diff, hip
    raw = hip.raw
    K, r, s = 512, 16, 2.0
    widths = [512] if single else [512, 256, 256]
    N, M = sum(widths), 140
    x = (_rand(2, 70, K, dev=dev) * 0.5).requires_grad_(True)
    w = _rand(N, K, dev=dev, seed=1, scale=K ** -0.5)
    res = _rand(2, 70, N, dev=dev, seed=8).requires_grad_(True) if single else None
    ads, off = [], 0
    for i, n in enumerate(widths):
        a = _rand(r, K, dev=dev, seed=10 + i, scale=0.05).requires_grad_(True)
        b = _rand(n, r, dev=dev, seed=20 + i, scale=0.05).requires_grad_(True)
        ads.append((off, a, b))
        off += n
    go = _rand(2, 70, N, dev=dev, seed=3)
    torch.manual_seed(77)
    y = ops.lora_projection(x, w, ads, s, p, res)
    assert y is not None and type(y.grad_fn).__name__ == "_LoRAProjectionBackward"
    y.backward(go)
    torch.manual_seed(77)
    seeds = [diff.next_seed() if p > 0 else 0 for _ in ads]
    xf, gf = x.detach().float().view(M, K), go.float().view(M, N)
    ones = torch.ones(M, K, dtype=torch.bfloat16, device=dev)
    se = s / (1 - p)
    yf = (xf @ w.float().t()).bfloat16().float()
    if res is not None:
        yf = yf + res.detach().float().view(M, N)
    dx = gf @ w.float()
    for (o, a, b), seed in zip(ads, seeds):
        n = b.shape[0]
        mask = raw.lowrank_mask(ones, p, seed).float() if p > 0 else ones.float()
        xm = xf * mask
        t = (se * (xm @ a.detach().float().t())).bfloat16().float()
        yf[:, o:o + n] = (yf[:, o:o + n].bfloat16().float() + (t @ b.detach().float().t()).bfloat16().float())
        gi = gf[:, o:o + n]
        dt = (se * (gi @ b.detach().float())).bfloat16().float()
        _close(a.grad, dt.t() @ xm, 2e-2 * math.sqrt(M / 16), 2e-2)
        _close(b.grad, gi.t() @ t, 2e-2 * math.sqrt(M / 16), 2e-2)
        dx = dx + mask * (dt @ a.detach().float())
    _close(y.view(M, N), yf, 3e-2, 2e-2)
    _close(x.grad.view(M, K), dx, 4e-2, 2e-2)
    if res is not None:
        assert torch.equal(res.grad, go)


def test_linear_with_residual_is_one_node(dev):
    """``ops.linear(..., residual=r)`` under autograd adds the residual in the GEMM's epilogue; the residual's
    gradient is the incoming gradient."""
    from eventgpt_amd import ops
    x = _rand(3, 50, 512, dev=dev).requires_grad_(True)
    w = _rand(1024, 512, dev=dev, seed=1, scale=0.05).requires_grad_(True)
    r = _rand(3, 50, 1024, dev=dev, seed=2).requires_grad_(True)
    go = _rand(3, 50, 1024, dev=dev, seed=3)
    y = ops.linear(x, w, None, None, r)
    assert type(y.grad_fn).__name__ == "_LinearBackward"
    y.backward(go)
    _close(y, (x.detach().float() @ w.detach().float().t()).bfloat16().float() + r.detach().float(), 2e-2, 1e-2)
    assert torch.equal(r.grad, go)
    _close(x.grad, go.float() @ w.detach().float(), 3e-2, 2e-2)
    _close(w.grad, go.float().view(-1, 1024).t() @ x.detach().float().view(-1, 512), 0.1, 2e-2)


def test_lora_projection_same_masks_with_and_without_autograd(dev):
    """The unrecorded pass (evaluation, or the first pass of a checkpointed block) runs the same kernels and
    draws the same seeds as the recorded one: with the CPU generator restored the outputs are identical."""
    from eventgpt_amd import ops
    K, N, r = 512, 768, 16
    x = _rand(3, 40, K, dev=dev)
    w = _rand(N, K, dev=dev, seed=1, scale=K ** -0.5)
    ads = [(0, _rand(r, K, dev=dev, seed=2, scale=0.05).requires_grad_(True), _rand(512, r, dev=dev, seed=3, scale=0.05).requires_grad_(True)),
           (512, _rand(r, K, dev=dev, seed=4, scale=0.05).requires_grad_(True), _rand(256, r, dev=dev, seed=5, scale=0.05).requires_grad_(True))]
    torch.manual_seed(11)
    with torch.no_grad():
        y0 = ops.lora_projection(x, w, ads, 2.0, 0.2)
    torch.manual_seed(11)
    y1 = ops.lora_projection(x, w, ads, 2.0, 0.2)
    assert y0 is not None and not y0.requires_grad and y1.requires_grad
    assert torch.equal(y0, y1.detach())
    torch.manual_seed(12)
    assert not torch.equal(ops.lora_projection(x, w, ads, 2.0, 0.2).detach(), y0)     # another seed, another mask
