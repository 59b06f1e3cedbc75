"""Training kernels of the hidden-state adapters (``csrc/kernels/adapter_train.hip``) against plain
fp32 PyTorch references: head_dim-512 attention forward + backward (masks, dropout), LayerNorm
backward, GELU forward / backward, counter-based dropout; and whole L4 / L5F training steps under
the profiler (no library GEMM, attention, normalisation, activation or dropout kernel)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(*shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed + sum(shape))
    return (torch.randn(*shape, device=dev, generator=g) * scale).to(torch.bfloat16)


def _attn_ref(qkv, heads, causal, kpm, keep, drop_p):
    """fp32 attention over a packed projection; ``keep`` [B,H,64,64] is the kernel's own dropout mask."""
    b, s, _ = qkv.shape
    q, k, v = qkv.float().view(b, s, 3, heads, 512).permute(2, 0, 3, 1, 4)      # [B,H,S,512]
    sc = q @ k.transpose(-1, -2) / math.sqrt(512)
    dead = torch.zeros(b, 1, s, s, dtype=torch.bool, device=qkv.device)
    if causal:
        dead = dead | ~torch.ones(s, s, dtype=torch.bool, device=qkv.device).tril()
    if kpm is not None:
        dead = dead | kpm[:, None, None, :]
    p = torch.softmax(sc.masked_fill(dead, float("-inf")), -1)
    if drop_p > 0:
        p = p * keep[:, :, :s, :s].float() / (1 - drop_p)
    return (p @ v).transpose(1, 2).reshape(b, s, heads * 512)


@pytest.mark.parametrize("B,S,H,causal,masked,drop_p", [
    (2, 1, 2, False, False, 0.0), (3, 17, 8, True, False, 0.0), (2, 50, 8, False, True, 0.0),
    (2, 64, 4, True, True, 0.0), (4, 33, 8, True, True, 0.1), (2, 50, 8, False, False, 0.25)])
def test_attention_d512_forward_backward(dev, B, S, H, causal, masked, drop_p):
    from eventgpt_amd.ops import hip
    from eventgpt_amd.ops import differentiable as diff
    qkv = _rand(B, S, 3 * H * 512, dev=dev, scale=0.7, seed=S).requires_grad_(True)
    kpm = None
    if masked:
        lens = torch.tensor([max(1, S - 3 * i) for i in range(B)], device=dev)
        kpm = torch.arange(S, device=dev)[None, :] >= lens[:, None]
    seed = 12345
    out, probs = hip.raw.attn512_fwd(qkv.detach(), H, causal, kpm, 1 / math.sqrt(512), drop_p, seed, True)
    keep = hip.raw.attn512_mask(B, H, drop_p, seed, qkv.detach()).bool()
    if drop_p > 0:
        frac = keep.float().mean().item()
        assert abs(frac - (1 - drop_p)) < 0.01, frac                 # the hash is uniform
    ref_in = qkv.detach().clone().requires_grad_(True)
    want = _attn_ref(ref_in, H, causal, kpm, keep, drop_p)
    assert torch.allclose(out.float(), want, rtol=2e-2, atol=2e-2), (out.float() - want).abs().max()
    # probabilities: rows sum to one over the live keys, zero elsewhere
    ps = probs.float()[:, :, :S, :S].sum(-1)
    assert torch.allclose(ps, torch.ones_like(ps), atol=2e-2)
    assert float(probs.float()[:, :, S:, :].abs().max()) == 0.0 if S < 64 else True
    # backward against autograd through the fp32 reference, same upstream gradient
    g = _rand(B, S, H * 512, dev=dev, scale=0.5, seed=7)
    want.backward(g.float())
    dqkv = hip.raw.attn512_bwd(qkv.detach(), g, probs, H, 1 / math.sqrt(512), drop_p, seed)
    err = (dqkv.float() - ref_in.grad).abs().max().item()
    assert err < 3e-2 * ref_in.grad.abs().max().item() + 2e-3, err
    # the autograd Function returns the same (eval-style call: no dropout, so seeds do not matter)
    if drop_p == 0:
        o2 = diff.attention_d512(qkv, H, causal, kpm, 0.0)
        o2.backward(g)
        assert torch.equal(o2, out) and torch.equal(qkv.grad, dqkv)


def test_attention_d512_is_deterministic_and_seeded(dev):
    from eventgpt_amd.ops import hip
    qkv = _rand(2, 40, 3 * 8 * 512, dev=dev, scale=0.5)
    a, _ = hip.raw.attn512_fwd(qkv, 8, True, None, 1 / math.sqrt(512), 0.2, 99, False)
    b, _ = hip.raw.attn512_fwd(qkv, 8, True, None, 1 / math.sqrt(512), 0.2, 99, False)
    c, _ = hip.raw.attn512_fwd(qkv, 8, True, None, 1 / math.sqrt(512), 0.2, 100, False)
    assert torch.equal(a, b) and not torch.equal(a, c)


@pytest.mark.parametrize("rows,H,bias", [(3200, 4096, True), (37, 1024, True), (130, 512, False), (5, 8192, True)])
def test_layernorm_forward_backward(dev, rows, H, bias):
    from eventgpt_amd.ops import differentiable as diff
    x = _rand(rows, H, dev=dev, seed=1).requires_grad_(True)
    w = (1 + 0.1 * torch.randn(H, device=dev)).requires_grad_(True)          # fp32 parameters, as under autocast
    b = (0.1 * torch.randn(H, device=dev)).requires_grad_(True) if bias else None
    g = _rand(rows, H, dev=dev, seed=2)
    y = diff.layernorm(x, w, b, 1e-5)
    assert y.dtype == torch.bfloat16
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = None if b is None else b.detach().to(torch.bfloat16).float().requires_grad_(True)
    yr = F.layer_norm(xr, (H,), wr, br, 1e-5)
    yr.backward(g.float())
    assert torch.allclose(y.float(), yr, rtol=2e-2, atol=2e-2)
    assert (x.grad.float() - xr.grad).abs().max() < 2e-2 * xr.grad.abs().max() + 1e-3
    assert w.grad.dtype == torch.float32
    assert (w.grad - wr.grad).abs().max() < 1e-2 * wr.grad.abs().max() + 1e-3
    if bias:
        assert (b.grad - br.grad).abs().max() < 1e-2 * br.grad.abs().max() + 1e-3


def test_gelu_and_dropout(dev):
    from eventgpt_amd.ops import differentiable as diff
    x = _rand(640, 2048, dev=dev, scale=2.0).requires_grad_(True)
    g = _rand(640, 2048, dev=dev, seed=3)
    y = diff.gelu(x)
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    yr = F.gelu(xr)
    yr.backward(g.float())
    assert torch.allclose(y.float(), yr, rtol=1e-2, atol=1e-2)
    assert torch.allclose(x.grad.float(), xr.grad, rtol=2e-2, atol=1e-2)
    # dropout: inverted scaling, the expected keep rate, and a backward that uses the forward's mask
    torch.manual_seed(5)
    z = torch.ones(512, 4096, device=dev, dtype=torch.bfloat16, requires_grad=True)
    d = diff.dropout(z, 0.1, True)
    kept = d != 0
    assert abs(kept.float().mean().item() - 0.9) < 0.005
    assert torch.allclose(d[kept].float(), torch.full_like(d[kept].float(), 1 / 0.9), rtol=1e-2)
    d.backward(torch.ones_like(d))
    assert torch.equal(z.grad != 0, kept)
    assert diff.dropout(z, 0.1, False) is z and diff.dropout(z, 0.0, True) is z
    torch.manual_seed(5)
    d2 = diff.dropout(z, 0.1, True)
    assert torch.equal(d2, d)                   # seeds come from torch's generator: repeatable
    assert not torch.equal(diff.dropout(z, 0.1, True), d)


@pytest.mark.parametrize("level", [4, 6])
def test_attention_adapter_training_step_matches_library(dev, level):
    """Loss and parameter gradients of an L4 / L5F step (eval mode: no dropout, so the functions are
    equal) on the in-tree kernels against the library path (fp32 LayerNorm / GELU, library attention)."""
    from eventgpt_amd.align.adapters import create_adapter
    from eventgpt_amd.ops.intree_mode import InTreeLinear
    torch.manual_seed(0)
    a = create_adapter(level, 1024, num_heads=2, ffn_dim=256).to(dev)
    b = create_adapter(level, 1024, num_heads=2, ffn_dim=256).to(dev)
    b.load_state_dict(a.state_dict())
    a.eval(), b.eval()
    h = torch.randn(6, 40, 1024, device=dev)
    t = h + 0.1 * torch.randn_like(h)
    mask = torch.ones(6, 40, device=dev)
    mask[1, 30:] = 0
    mask[4, 12:] = 0
    la = a.float().compute_loss(h, t, mask)["total_loss"]          # fp32, library kernels throughout
    with torch.autocast("cuda", dtype=torch.bfloat16), InTreeLinear() as mode:
        lb = b.compute_loss(h, t, mask)["total_loss"]
    assert mode.routed >= 8, (mode.routed, mode.passed)
    la.backward()
    lb.backward()
    assert abs(float(la) - float(lb)) < 3e-2 * abs(float(la)) + 1e-3, (float(la), float(lb))
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is None:
            assert pb.grad is None, n
            continue
        assert pb.grad is not None and pb.grad.dtype == torch.float32, n
        err = (pa.grad - pb.grad).abs().max().item()
        assert err < 6e-2 * pa.grad.abs().max().item() + 1e-5, (n, err, pa.grad.abs().max().item())


@pytest.mark.parametrize("level", [4, 6])
def test_attention_adapter_training_step_uses_only_intree_kernels(dev, level):
    """An L4 / L5F training step (train mode: dropout everywhere, 4096 wide, 8 heads of 512) under the
    profiler: no Tensile GEMM, no library attention, no ATen LayerNorm / GELU / dropout / softmax kernel."""
    from eventgpt_amd.align.adapters import create_adapter
    from eventgpt_amd.ops.intree_mode import InTreeLinear
    torch.manual_seed(1)
    m = create_adapter(level, 4096).to(dev).train()
    h = torch.randn(16, 48, 4096, device=dev)
    t = h + 0.1 * torch.randn_like(h)
    mask = torch.ones(16, 48, device=dev)
    mask[3, 20:] = 0
    for _ in range(2):                      # first pass: module loads, outside the profile
        m.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16), InTreeLinear():
            m.compute_loss(h, t, mask)["total_loss"].backward()
    m.zero_grad()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        with torch.autocast("cuda", dtype=torch.bfloat16), InTreeLinear():
            loss = m.compute_loss(h, t, mask)["total_loss"]
        loss.backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    for need in ("attn512_fwd_kernel", "attn512_bwd_kernel", "layernorm_bwd_kernel", "adapter_elementwise_kernel",
                 "gemm"):
        assert any(need in n for n in names), (need, names)
    bad = [n for n in names if n.startswith("Cijk_") or "attn_fwd" in n or "bwd_kernel_dk_dv" in n
           or "layer_norm" in n.lower() or "GammaBeta" in n or "gelu" in n.lower().replace("adapter_elementwise", "")
           or "fused_dropout" in n or "softmax" in n.lower()]
    assert not bad, bad
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


@pytest.mark.parametrize("M,V,K,n_ignored", [(48, 32003, 4096, 0), (840, 1003, 512, 800), (130, 2048, 512, 64),
                                             (17, 32003, 4096, 5)])
def test_linear_cross_entropy_matches_fp32(dev, M, V, K, n_ignored):
    """Loss and both gradients of the fused LM-head cross-entropy against F.cross_entropy over fp32
    logits of the same bf16 operands; unsupervised rows take no part (and get zero gradient)."""
    from eventgpt_amd.ops import differentiable as diff
    x = _rand(M, K, dev=dev, scale=1.0, seed=1).requires_grad_(True)
    w = _rand(V, K, dev=dev, scale=0.05, seed=2).requires_grad_(True)
    g = torch.Generator(device=dev).manual_seed(3)
    labels = torch.randint(0, V, (M,), device=dev, generator=g)
    if n_ignored:
        labels[torch.randperm(M, device=dev, generator=g)[:n_ignored]] = -100
    loss = diff.linear_cross_entropy(x, w, labels)
    (loss * 0.5).backward()                       # an upstream factor, as gradient accumulation applies
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().float().requires_grad_(True)
    want = F.cross_entropy(xr @ wr.t(), labels, ignore_index=-100)
    (want * 0.5).backward()
    assert abs(float(loss) - float(want)) < 2e-2 * abs(float(want)) + 1e-3, (float(loss), float(want))
    assert x.grad.shape == x.shape and w.grad.shape == w.shape
    assert (x.grad.float() - xr.grad).abs().max() < 3e-2 * xr.grad.abs().max() + 1e-5
    assert (w.grad.float() - wr.grad).abs().max() < 3e-2 * wr.grad.abs().max() + 1e-5
    if n_ignored:
        assert float(x.grad[labels == -100].abs().max()) == 0.0


def test_model_training_loss_uses_the_fused_head(dev):
    """EventChatModel.forward with labels in a training step: same loss and gradients as the full-logits
    path, no fp32 softmax kernel, and ``logits`` is not materialised."""
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    cfg = EventChatConfig(vocab_size=1003, hidden_size=512, intermediate_size=1408, num_hidden_layers=2,
                          num_attention_heads=4, max_position_embeddings=512)
    a = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=0)
    b = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=0)
    for m in (a, b):
        for p in m.parameters():
            p.requires_grad_(True)
    ids = torch.randint(3, 1000, (1, 90), device=dev)
    labels = ids.clone()
    labels[:, :70] = -100
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        out = a(input_ids=ids, labels=labels)
        out.loss.backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert out.logits is None and any("ce_fwd_kernel" in n for n in names), names
    assert not [n for n in names if "softmax" in n.lower() or "nll_loss" in n.lower()], names
    # reference: all logits in fp32
    h = b.model(input_ids=ids, use_cache=False).last_hidden_state
    logits = b.lm_head(h).float()
    want = F.cross_entropy(logits[:, :-1].reshape(-1, 1003), labels[:, 1:].reshape(-1), ignore_index=-100)
    want.backward()
    assert abs(float(out.loss) - float(want)) < 2e-2 * abs(float(want)) + 1e-3
    ga, gb = a.lm_head.weight.grad.float(), b.lm_head.weight.grad.float()
    assert (ga - gb).abs().max() < 5e-2 * gb.abs().max() + 1e-4
    ga, gb = a.model.layers[0].mlp.down_proj.weight.grad.float(), b.model.layers[0].mlp.down_proj.weight.grad.float()
    assert (ga - gb).abs().max() < 8e-2 * gb.abs().max() + 1e-4


def test_padded_batch_step_equals_single_sample_steps(dev):
    """Three samples of different lengths in ONE right-padded forward / backward (fused attention, no mask,
    ``loss_reduction="sequence_mean"``) against the three single-sample passes averaged: same loss, same
    gradients, and the attention launches are the in-tree kernels."""
    from eventgpt_amd.constants import EVENT_TOKEN_INDEX
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    cfg = EventChatConfig(vocab_size=1003, hidden_size=512, intermediate_size=1408, num_hidden_layers=2,
                          num_attention_heads=4, max_position_embeddings=1024)
    a = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=0)
    b = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=0)
    for m in (a, b):
        m.train()
        for p in m.parameters():
            p.requires_grad_(True)
    g = torch.Generator().manual_seed(0)
    lens, n_ev = [70, 96, 83], 40
    ids, labels, feats = [], [], []
    for n in lens:
        t = torch.randint(3, 1000, (n,), generator=g)
        t[5] = EVENT_TOKEN_INDEX
        lab = t.clone()
        lab[: n - 12] = -100
        ids.append(t), labels.append(lab)
        feats.append((torch.randn(n_ev, 512, generator=g) * 0.1).to(torch.bfloat16).to(dev))
    width = max(lens)
    pid = torch.zeros(3, width, dtype=torch.long)
    plab = torch.full((3, width), -100, dtype=torch.long)
    pmask = torch.zeros(3, width, dtype=torch.long)
    for i, n in enumerate(lens):
        pid[i, :n], plab[i, :n], pmask[i, :n] = ids[i], labels[i], 1
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        out = a(input_ids=pid.to(dev), attention_mask=pmask.to(dev), labels=plab.to(dev),
                event_features=torch.stack(feats), already_adapted=True, loss_reduction="sequence_mean")
        out.loss.backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert any("attn_bwd_dkv_kernel" in n for n in names) and not [n for n in names if "attn_fwd" in n], names
    total = 0.0
    for i in range(3):
        o = b(input_ids=ids[i][None].to(dev), labels=labels[i][None].to(dev), event_features=feats[i][None],
              already_adapted=True)
        (o.loss / 3).backward()
        total += float(o.loss) / 3
    assert abs(float(out.loss) - total) < 1e-2 * abs(total) + 1e-3, (float(out.loss), total)
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pb.grad is None:
            continue
        ga, gb = pa.grad.float(), pb.grad.float()
        assert (ga - gb).abs().max() < 6e-2 * gb.abs().max() + 1e-4, (n, float((ga - gb).abs().max()), float(gb.abs().max()))
