"""Two to four x rows through the decode projections (the rows kernel) against the same projection applied
row by row with the single-row kernels."""
import pytest
import torch

pytestmark = pytest.mark.gpu

GEOMS = [(512, 1408, 8, 64),          # small: rows end inside a step
         (4096, 11008, 32, 128),      # the 7B layer
         (1152, 3200, 6, 96)]         # K = 1152: a row ends in the second step; 3200 / 128 = 25 loads


def _dev():
    return torch.device("cuda")


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(_dev(), torch.bfloat16)


def _close(a, b, what):
    a, b = a.float(), b.float()
    typical = b.abs()[b != 0].mean()
    err = (a - b).abs()
    assert (err <= 2e-2 * b.abs() + 2e-2 * typical).all(), (what, err.max().item(), typical.item())
    assert (a[b == 0] == 0).all(), what


def _rows(raw, K):
    return [B for B in (2, 3, 4) if B <= int(raw.gemv_max_batch()) and B * K * 2 <= 96 * 1024]


@pytest.mark.parametrize("geom", GEOMS)
def test_residual_rows_match_single_rows(geom):
    from eventgpt_amd.ops import hip
    raw = hip.raw
    H, I, Hq, D = geom
    for (N, K) in ((H, Hq * D), (H, I)):
        w = _rand((N, K), N + K, 0.03)
        for B in _rows(raw, K):
            x = _rand((B, K), 3 + B, 0.5)
            y0 = _rand((B, N), 5 + B)
            y = y0.clone()
            raw.decode_gemv_residual(w, x, y)
            want = y0.clone()
            for b in range(B):
                raw.decode_gemv_residual(w, x[b:b + 1], want[b:b + 1])
            _close(y, want, ("residual", N, K, B))
            out = torch.zeros(1, B, N, dtype=torch.float32, device=_dev())
            raw.decode_gemv_partial(w, x, out)
            ref = x.float() @ w.float().t()
            assert (out[0] - ref).abs().max() <= 2e-3 * ref.abs().mean() + 1e-3 * ref.abs().max()


@pytest.mark.parametrize("geom", GEOMS)
def test_mlp_up_rows_match_single_rows(geom):
    from eventgpt_amd.ops import hip
    raw = hip.raw
    H, I, Hq, D = geom
    w = _rand((2 * I, H), 7, 0.03)
    nw = (1 + 0.1 * _rand((H,), 8).float()).to(torch.bfloat16)
    for B in _rows(raw, H):
        x = _rand((B, H), 9 + B)
        h = torch.empty(B, I, dtype=torch.bfloat16, device=_dev())
        raw.decode_mlp_up(w, x, nw, 1e-6, h)
        want = torch.empty_like(h)
        for b in range(B):
            raw.decode_mlp_up(w, x[b:b + 1], nw, 1e-6, want[b:b + 1])
        _close(h, want, ("mlp_up", H, I, B))


@pytest.mark.parametrize("one_seq", [False, True])
@pytest.mark.parametrize("geom", GEOMS)
def test_qkv_rope_rows_match_single_rows(geom, one_seq):
    from eventgpt_amd import ops
    from eventgpt_amd.ops import hip
    raw = hip.raw
    H, I, Hq, D = geom
    Hkv, L = Hq, 96
    w = _rand(((Hq + 2 * Hkv) * D, H), 11, 0.03)
    nw = (1 + 0.1 * _rand((H,), 12).float()).to(torch.bfloat16)
    cos, sin = ops.rope_tables(L, D, 10000.0, _dev())
    for B in _rows(raw, H):
        x = _rand((B, H), 13 + B)
        S = 1 if one_seq else B
        pos = torch.tensor([17] if one_seq else [5 + 7 * b for b in range(B)], dtype=torch.int32, device=_dev())
        q = torch.zeros(B, Hq * D, dtype=torch.bfloat16, device=_dev())
        kc = torch.zeros(S, Hkv, L, D, dtype=torch.bfloat16, device=_dev())
        vc = torch.zeros_like(kc)
        raw.decode_qkv(x, nw, w, q, kc, vc, cos, sin, pos, 1e-6, Hq, Hkv, D, one_seq)
        q1, k1, v1 = torch.zeros_like(q), torch.zeros_like(kc), torch.zeros_like(vc)
        for b in range(B):
            pb = (pos + b) if one_seq else pos[b:b + 1]
            sl = slice(0, 1) if one_seq else slice(b, b + 1)
            raw.decode_qkv(x[b:b + 1], nw, w, q1[b:b + 1], k1[sl], v1[sl], cos, sin, pb.contiguous(), 1e-6,
                           Hq, Hkv, D, False)
        _close(q, q1, ("q", geom, B, one_seq))
        _close(kc, k1, ("k", geom, B, one_seq))
        _close(vc, v1, ("v", geom, B, one_seq))
        assert k1.abs().sum() > 0


def test_verify_pass_on_the_rows_kernel_is_the_few_row_path():
    """A verify pass of three drafts (four rows) on a 7B-geometry engine of two layers: the weight-streaming
    path and the few-row GEMM path leave the same hidden state up to rounding."""
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    from eventgpt_amd.runtime.engine import GenerationEngine
    cfg = EventChatConfig.eventgpt_7b()
    cfg.num_hidden_layers = 2
    m = EventChatModel.from_config(cfg, torch.bfloat16, _dev(), seed=2)
    fast = GenerationEngine(m, max_len=512, max_new_tokens=16)
    slow = GenerationEngine(m, max_len=512, max_new_tokens=16)
    slow.gemv_rows = 0
    assert fast.gemv_rows >= 4
    emb = _rand((40, cfg.hidden_size), 21, 0.02)
    draft = torch.randint(0, 32000, (3,), device=_dev(), dtype=torch.int32)
    outs = []
    for e in (fast, slow):
        e.eos = -1
        e.reset()
        e.prefill(emb, select="greedy")
        outs.append(e.verify(draft, 3, return_hidden=True).float())
    cosim = torch.nn.functional.cosine_similarity(outs[0].flatten(), outs[1].flatten(), dim=0).item()
    assert cosim > 0.9995, cosim
