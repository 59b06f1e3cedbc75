"""The 64-query-row attention block as two groups of four waves that share the key tiles (even / odd)
and merge at the end, forced onto every shape class the one-group kernel serves: causal with and without
an offset into a longer key range, non-causal, grouped KV heads, ragged tails, fewer tiles than groups,
a device-side key count, strided inputs, a late spike that forces the rescale in one group only, and the
log-sum-exp that feeds the backward.  Against the fp32 reference and against the one-group kernel."""
import pytest
import torch

from eventgpt_amd.ops import reference as ref

pytestmark = pytest.mark.gpu


def _rand(*shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed + sum(shape))
    return (torch.randn(*shape, device=dev, generator=g) * scale).to(torch.bfloat16)


@pytest.fixture()
def two():
    from eventgpt_amd.ops import hip
    hip.raw.attn_enable_wide(0)
    hip.raw.attn_force_groups(2)
    yield hip
    hip.raw.attn_force_groups(0)
    hip.raw.attn_enable_wide(1)


@pytest.mark.parametrize("D,H,Hkv,Sq,Sk,causal", [
    (128, 32, 32, 636, 636, True), (128, 8, 2, 300, 300, True), (128, 2, 2, 129, 700, True), (128, 2, 2, 64, 64, True),
    (64, 16, 16, 577, 577, False), (64, 4, 4, 250, 250, False), (128, 2, 2, 1, 130, True), (64, 2, 2, 130, 259, True),
    (128, 2, 1, 257, 1000, False), (128, 2, 2, 5, 40, True), (64, 2, 2, 70, 65, False), (128, 4, 4, 836, 836, True)])
def test_two_group_attention_matches_reference_and_one_group(dev, two, D, H, Hkv, Sq, Sk, causal):
    B = 2
    q, k, v = _rand(B, Sq, H, D, dev=dev), _rand(B, Sk, Hkv, D, dev=dev, seed=1), _rand(B, Sk, Hkv, D, dev=dev, seed=2)
    rep = H // Hkv
    want = ref.attention(q, k.repeat_interleave(rep, 2), v.repeat_interleave(rep, 2), causal, q_offset=Sk - Sq)
    first = None
    for i in range(3):
        o = two.attention(q, k, v, causal, None, Sk - Sq)
        assert torch.allclose(o.float(), want.float(), atol=2e-2, rtol=2e-2), (i, float((o.float() - want.float()).abs().max()))
        first = o if first is None else first
        assert torch.equal(o, first)
    two.raw.attn_force_groups(1)
    one = two.attention(q, k, v, causal, None, Sk - Sq)
    two.raw.attn_force_groups(2)
    assert torch.allclose(first.float(), one.float(), atol=1e-2, rtol=1e-2)


def test_two_group_attention_strided_inputs_and_spike(dev, two):
    B, S, H, D = 1, 400, 4, 128
    qkv = _rand(B, S, 3, H, D, dev=dev)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    k[0, 333] = q[0, 399] * 4.0           # key tile 5: group 1 only
    o = two.attention(q, k, v, True, None, 0)
    want = ref.attention(q.contiguous(), k.contiguous(), v.contiguous(), True)
    assert torch.allclose(o.float(), want.float(), atol=2e-2, rtol=2e-2)


def test_two_group_attention_against_cache_with_device_length(dev, two):
    from eventgpt_amd.ops import hip, reference
    D, Hq, Hkv, Sq, kv, Lmax = 128, 4, 2, 300, 900, 1024
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn(1, Sq, Hq, D, device=dev, generator=g).to(torch.bfloat16)
    kc = torch.randn(1, Hkv, Lmax, D, device=dev, generator=g).to(torch.bfloat16)
    vc = torch.randn(1, Hkv, Lmax, D, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty_like(q)
    hip.raw.attention_cache(q, kc, vc, out, torch.tensor([kv], dtype=torch.int32, device=dev), D ** -0.5)
    k = kc[:, :, :kv].transpose(1, 2).float().repeat_interleave(2, dim=2)
    v = vc[:, :, :kv].transpose(1, 2).float().repeat_interleave(2, dim=2)
    want = reference.attention(q.float(), k, v, causal=True, q_offset=kv - Sq)
    assert torch.allclose(out.float(), want, atol=2e-2, rtol=2e-2)


def test_two_group_forward_feeds_backward(dev, two):
    from eventgpt_amd import ops
    B, S, H, D = 1, 384, 4, 128
    q, k, v = (_rand(B, S, H, D, dev=dev, seed=i, scale=0.5).requires_grad_(True) for i in range(3))
    go = _rand(B, S, H, D, dev=dev, seed=7)
    o = ops.attention(q, k, v, True)
    assert type(o.grad_fn).__name__ == "_AttentionBackward"
    o.backward(go)
    qf, kf, vf = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    ref.attention(qf, kf, vf, True).backward(go.float())
    for got, want in ((q.grad, qf.grad), (k.grad, kf.grad), (v.grad, vf.grad)):
        assert torch.allclose(got.float(), want, atol=3e-2, rtol=3e-2), float((got.float() - want).abs().max())
