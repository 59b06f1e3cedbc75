"""Draft-head kernels: adapter forward on the in-tree kernels vs the nn.Module, head_dim-512
attention, device-index gather, argmax from partials, device-resident SD rounds."""
import math

import pytest
import torch

from eventgpt_amd.ops import reference as ref

pytestmark = pytest.mark.gpu


def _rand(*shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed + sum(shape))
    return (torch.randn(*shape, device=dev, generator=g) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("S,H,causal", [(1, 8, True), (7, 8, True), (22, 8, False), (50, 2, True)])
def test_attn_d512(dev, S, H, causal):
    from eventgpt_amd.ops import hip
    qkv = _rand(2, S, 3, H, 512, dev=dev, scale=0.5)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    o = hip.raw.attn_d512(q, k, v, causal, 1 / math.sqrt(512))
    want = ref.attention(q.contiguous(), k.contiguous(), v.contiguous(), causal, None, 0)
    assert torch.allclose(o.float(), want.float(), rtol=2e-2, atol=2e-2)


def test_gather_rows_dev_and_argmax_from_partials(dev):
    from eventgpt_amd.ops import hip
    src = _rand(10, 512, dev=dev)
    idx = torch.tensor([4], dtype=torch.int32, device=dev)
    nv = torch.tensor([7], dtype=torch.int32, device=dev)
    out = hip.raw.gather_rows_dev(src, idx, -1, nv, 5)          # rows 3,4,5,6,6 (clamped to n_valid - 1)
    assert torch.equal(out, src[[3, 4, 5, 6, 6]])
    idx.fill_(0)
    assert torch.equal(hip.raw.gather_rows_dev(src, idx, -1, None, 2), src[[0, 0]])   # clamped at 0
    pv = torch.randn(6, 40, device=dev)
    pi = torch.randint(0, 900, (6, 40), device=dev, dtype=torch.int32)
    pv[2, 7] = 9.0; pv[2, 30] = 9.0; pi[2, 7] = 500; pi[2, 30] = 17      # tie -> lower index
    tok = torch.zeros(6, dtype=torch.int32, device=dev)
    hip.raw.argmax_from_partials(pv, pi, tok, 1000)
    want = pi.gather(1, pv.argmax(1, keepdim=True))[:, 0]
    want[2] = 17
    assert tok.tolist() == want.tolist()


@pytest.mark.parametrize("level,hidden,heads", [(1, 512, 4), (2, 512, 4), (3, 512, 4), (4, 512, 4), (5, 512, 4),
                                                (6, 512, 4), (4, 4096, 8), (6, 4096, 8), (1, 4096, 8)])
@pytest.mark.parametrize("S", [1, 6])
def test_adapter_kernels_match_module(dev, level, hidden, heads, S):
    from eventgpt_amd.align.adapter_engine import AdapterKernels
    from eventgpt_amd.align.adapters import create_adapter
    torch.manual_seed(level)
    kw = dict(bottleneck_dim=64) if level <= 3 else dict(num_heads=heads, ffn_dim=256)
    adapter = create_adapter(level, hidden, **kw).to(dev).eval()
    with torch.no_grad():
        for n, p in adapter.named_parameters():       # zero-initialised projections would hide errors
            if p.dim() == 2 and float(p.abs().sum()) == 0:
                p.normal_(0, 0.02)
    h = _rand(S, hidden, dev=dev)
    vl = _rand(S, hidden, dev=dev, seed=9)
    kern = AdapterKernels(adapter)
    got = kern(h, vl if level == 6 else None)
    with torch.no_grad():
        want = adapter(h.float()[None], vl.float()[None])[0] if level == 6 else adapter(h.float()[None])[0]
    cos = torch.nn.functional.cosine_similarity(got.float(), want, dim=-1)
    assert float(cos.min()) > 0.998, float(cos.min())
    assert torch.allclose(got.float(), want, rtol=5e-2, atol=5e-2 * float(want.abs().max()))


@pytest.mark.parametrize("mode,level", [("adapter", 1), ("adapter", 4), ("vlm_only", 5), ("fused", 6)])
def test_device_rounds_equal_host_rounds_and_greedy(dev, mode, level):
    """Captured draft/verify rounds replayed without host reads emit exactly the verifier's greedy
    tokens, like the host-driven loop, and count their rounds on the device."""
    from eventgpt_amd import ops
    from eventgpt_amd.align.adapters import create_adapter
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    from eventgpt_amd.runtime.spec_decode import AdapterDraftHead, SpecDecodeEngine, collect_decode_hidden
    cfg = EventChatConfig(vocab_size=1024, hidden_size=512, intermediate_size=1408, num_hidden_layers=2,
                          num_attention_heads=4, max_position_embeddings=512)
    target = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=1)
    drafter = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=2)
    emb = _rand(28, 512, dev=dev, scale=0.5, seed=5)
    v = target.engine(max_len=512, max_new_tokens=128)
    greedy = v.generate(emb, 24, eos_token_id=-1)
    dh = collect_decode_hidden(drafter.engine(max_len=512, max_new_tokens=128), emb, 24)
    torch.manual_seed(0)
    adapter = create_adapter(level, 512, bottleneck_dim=32, num_heads=4, ffn_dim=64).to(dev).to(torch.bfloat16)
    head = AdapterDraftHead(adapter, target.lm_head.weight.data)
    assert head.kernels is not None
    sd = SpecDecodeEngine(v, head, gamma=4)
    assert sd.device_loop_available()
    from _lossless import assert_lossless
    outs = {}
    for dl in (False, True, True):                 # the third run replays the graph captured by the second
        v.reset(); v.eos = -1
        h = v.prefill(emb, select="greedy", return_hidden=True)
        vh0 = ops.rmsnorm(h[-1:], v.final_norm, v.eps)
        toks, stats = sd.run(dh["hidden"], 24, -1, mode, vh0, device_loop=dl, poll_every=3)
        assert_lossless(v, emb, toks, greedy[:24])
        assert stats.total_drafted > 0 and stats.iterations > 0 and stats.tokens == 24
        outs[dl] = stats
    assert outs[True].iterations > 0 and outs[True].total_drafted >= outs[True].total_accepted


def test_verify_past_the_cache_is_refused(dev):
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    cfg = EventChatConfig(vocab_size=512, hidden_size=512, intermediate_size=1408, num_hidden_layers=1,
                          num_attention_heads=4, max_position_embeddings=512)
    model = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=1)
    v = model.engine(max_len=64, max_new_tokens=64)
    v.eos = -1
    v.reset()
    v.prefill(_rand(58, 512, dev=dev, scale=0.5))
    draft = torch.zeros(5, dtype=torch.int32, device=dev)
    v.verify(draft, 5)                              # 58 + 6 = 64 rows: fits exactly
    with pytest.raises(ValueError, match="does not fit the cache"):
        v.verify(draft, 5)
    assert int(v.pos[0]) <= 64


@pytest.mark.parametrize("level", [1, 3])
def test_adapter_training_step_on_in_tree_linear(level):
    """Inside ``InTreeLinear`` the nn.Linear layers of an adapter (fp32 parameters, bf16 autocast) run
    the in-tree GEMM forward and backward; loss and parameter gradients agree with the library path."""
    from eventgpt_amd.align.adapters import create_adapter
    from eventgpt_amd.ops.intree_mode import InTreeLinear
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    a = create_adapter(level, 512, bottleneck_dim=64).to(dev)
    b = create_adapter(level, 512, bottleneck_dim=64).to(dev)
    b.load_state_dict(a.state_dict())
    a.eval(), b.eval()                       # no dropout: the two paths see the same function
    h = torch.randn(8, 40, 512, device=dev)
    t = h + 0.1 * torch.randn_like(h)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        la = a.compute_loss(h, t)["total_loss"]
    with torch.autocast("cuda", dtype=torch.bfloat16), InTreeLinear() as mode:
        lb = b.compute_loss(h, t)["total_loss"]
    assert mode.routed >= 2, (mode.routed, mode.passed)
    la.backward()
    lb.backward()
    assert abs(float(la) - float(lb)) < 2e-2 * abs(float(la)) + 1e-4
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is None:
            assert pb.grad is None, n
            continue
        assert pb.grad is not None and pb.grad.dtype == torch.float32, n
        err = (pa.grad - pb.grad).abs().max().item()
        # (LayerNorm inside the mode reads its input rounded to bf16: its parameters' gradients carry that rounding)
        tol = 8e-2 if "norm" in n else 3e-2
        assert err < tol * pa.grad.abs().max().item() + 1e-6, (n, err)
