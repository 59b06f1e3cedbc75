"""The decode projections on narrow matrices -- short contractions (512, 1408: a weight row ends inside the
first step of the rows kernel), few output rows (1536, 2816) and an odd row count for the LM head (4001) --
against fp32 products.  A tensor-parallel rank holds matrices of these shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu

H, D = 4096, 128
SHARDS = [(4, 1408), (8, 2752)]        # (heads, MLP columns) of the matrices under test


def _dev():
    return torch.device("cuda")


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(_dev(), torch.bfloat16)


def _close(a, b, what, tol=2e-2):
    a, b = a.float(), b.float()
    err = (a - b).abs()
    assert (err <= tol * b.abs() + tol * b.abs().mean()).all(), (what, err.max().item(), b.abs().mean().item())


def _norm(x, w, eps=1e-6):
    xf = x.float()
    return (xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + eps)).to(torch.bfloat16).float() * w.float()


@pytest.mark.parametrize("heads,inter", SHARDS)
@pytest.mark.parametrize("B", [1, 3])
def test_residual_projections_over_short_contractions(heads, inter, B):
    from eventgpt_amd.ops import hip
    raw = hip.raw
    for K in (heads * D, inter):
        w = _rand((H, K), K, 0.03)
        x = _rand((B, K), 3 + B, 0.5)
        ref = x.float() @ w.float().t()
        out = torch.zeros(1, B, H, dtype=torch.float32, device=_dev())
        raw.decode_gemv_partial(w, x, out)
        assert (out[0] - ref).abs().max() <= 2e-3 * ref.abs().mean() + 1e-3 * ref.abs().max(), (K, B)
        y0 = _rand((B, H), 5 + B)
        y = y0.clone()
        raw.decode_gemv_residual(w, x, y)
        _close(y, y0.float() + ref, ("residual", K, B))


@pytest.mark.parametrize("heads,inter", SHARDS)
@pytest.mark.parametrize("B", [1, 3])
def test_gate_up_with_few_columns(heads, inter, B):
    from eventgpt_amd.ops import hip
    raw = hip.raw
    w = _rand((2 * inter, H), 7, 0.03)
    nw = (1 + 0.1 * _rand((H,), 8).float()).to(torch.bfloat16)
    x = _rand((B, H), 9 + B)
    h = torch.empty(B, inter, dtype=torch.bfloat16, device=_dev())
    raw.decode_mlp_up(w, x, nw, 1e-6, h)
    gu = _norm(x, nw).to(torch.bfloat16).float() @ w.float().t()
    _close(h, torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:], ("mlp_up", inter, B), tol=3e-2)


@pytest.mark.parametrize("heads,inter", SHARDS)
@pytest.mark.parametrize("B", [1, 3])
def test_qkv_of_few_heads(heads, inter, B):
    from eventgpt_amd import ops
    from eventgpt_amd.ops import hip
    raw = hip.raw
    L, hh = 96, D // 2
    w = _rand((3 * heads * D, H), 11, 0.03)
    nw = (1 + 0.1 * _rand((H,), 12).float()).to(torch.bfloat16)
    cos, sin = ops.rope_tables(L, D, 10000.0, _dev())
    x = _rand((B, H), 13 + B)
    pos = torch.tensor([5 + 7 * b for b in range(B)], dtype=torch.int32, device=_dev())
    q = torch.zeros(B, heads * D, dtype=torch.bfloat16, device=_dev())
    kc = torch.zeros(B, heads, L, D, dtype=torch.bfloat16, device=_dev())
    vc = torch.zeros_like(kc)
    raw.decode_qkv(x, nw, w, q, kc, vc, cos, sin, pos, 1e-6, heads, heads, D, False)
    qkv = (_norm(x, nw).to(torch.bfloat16).float() @ w.float().t()).view(B, 3, heads, D)

    def rope(t, b):
        c, s = cos[int(pos[b])].float(), sin[int(pos[b])].float()
        lo, hi = t[..., :hh], t[..., hh:]
        return torch.cat([lo * c - hi * s, hi * c + lo * s], -1)

    for b in range(B):
        p = int(pos[b])
        _close(q[b].view(heads, D), rope(qkv[b, 0], b), ("q", heads, b), tol=3e-2)
        _close(kc[b, :, p], rope(qkv[b, 1], b), ("k", heads, b), tol=3e-2)
        _close(vc[b, :, p], qkv[b, 2], ("v", heads, b), tol=3e-2)
        assert kc[b].abs().sum(-1).count_nonzero() == heads        # one cache row per head was written


@pytest.mark.parametrize("rows", [4001, 8001])
def test_lm_head_with_an_odd_row_count(rows):
    """Greedy selection from (max, index) partials of a matrix whose row count is odd."""
    from eventgpt_amd.ops import hip
    raw = hip.raw
    w = _rand((rows, H), 17, 0.03)
    nw = (1 + 0.1 * _rand((H,), 18).float()).to(torch.bfloat16)
    x = _rand((1, H), 19)
    n_part = int(raw.logits_partials(rows))
    pv = torch.zeros(n_part, dtype=torch.float32, device=_dev())
    pi = torch.zeros(n_part, dtype=torch.int32, device=_dev())
    raw.decode_logits(w, x, nw, 1e-6, None, pv, pi)
    logits = _norm(x, nw).to(torch.bfloat16).float() @ w.float().t()
    best = int(pi[int(pv.argmax())])
    assert 0 <= int(pi.min()) and int(pi.max()) < rows
    top = logits[0].max()
    assert logits[0, best] >= top - 2e-2 * logits.abs().mean() - 1e-2 * top.abs(), (best, int(logits.argmax()))
    # six rows: the GEMM path's partials
    x6 = _rand((6, H), 20)
    hn = _norm(x6, nw).to(torch.bfloat16)
    pv6, pi6 = raw.linear_argmax_partials(hn, w, None)
    l6 = hn.float() @ w.float().t()
    for r in range(6):
        b = int(pi6[r, int(pv6[r].argmax())])
        assert 0 <= b < rows and l6[r, b] >= l6[r].max() - 2e-2 * l6.abs().mean() - 1e-2 * l6[r].max().abs(), r


@pytest.mark.parametrize("heads,inter", SHARDS)
@pytest.mark.parametrize("M", [32, 636])
def test_prompt_and_batch_gemms_on_narrow_matrices(heads, inter, M):
    """The products of a 636-row prompt and of a 32-stream step: row-parallel projections over the short
    contractions (plain, with residual, as one fp32 slab and as split-K slabs), the gated MLP input
    projection with few columns, and q|k|v of few heads with rotation and cache append in the epilogue
    against GEMM + rotation kernel."""
    from eventgpt_amd import ops
    from eventgpt_amd.ops import hip
    raw = hip.raw
    for K in (heads * D, inter):
        w = _rand((H, K), K + 1, 0.03)
        x = _rand((M, K), 23, 0.5)
        res = _rand((M, H), 24)
        ref = x.float() @ w.float().t()
        _close(raw.linear(x, w, None, None, 0, False), ref, ("linear", M, K))
        _close(raw.linear(x, w, None, res, 0, False), ref + res.float(), ("linear+res", M, K))
        for nslab in (1, None):
            part = raw.linear_partial(x, w, 1) if nslab else raw.linear_partial(x, w)
            assert part.dtype == torch.float32 and part.shape[1:] == (M, H) and (nslab is None or part.shape[0] == 1)
            got = part.sum(0)
            assert (got - ref).abs().max() <= 2e-3 * ref.abs().mean() + 2e-3 * ref.abs().max(), (M, K, nslab)
    w_gu = _rand((2 * inter, H), 25, 0.03)
    xn = _rand((M, H), 26, 0.5)
    gu = xn.float() @ w_gu.float().t()
    want = torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]
    if M > 32:
        _close(raw.linear_swiglu(xn, w_gu), want, ("linear_swiglu", M, inter), tol=3e-2)
    _close(raw.reduce_swiglu(raw.linear_partial(xn, w_gu)), want, ("reduce_swiglu", M, inter), tol=3e-2)
    if M > 32:
        L = 768
        w_qkv = _rand((3 * heads * D, H), 27, 0.03)
        cos, sin = ops.rope_tables(L, D, 10000.0, _dev())
        pos0 = torch.tensor([9], dtype=torch.int32, device=_dev())
        q = torch.zeros(1, M, heads, D, dtype=torch.bfloat16, device=_dev())
        kc = torch.zeros(1, heads, L, D, dtype=torch.bfloat16, device=_dev())
        vc = torch.zeros_like(kc)
        assert raw.linear_qkv_rope(xn, w_qkv, q, kc, vc, cos, sin, pos0, heads, heads)
        qkv = raw.linear(xn, w_qkv, None, None, 0, False).view(1, M, -1)
        q1, k1, v1 = torch.zeros_like(q), torch.zeros_like(kc), torch.zeros_like(vc)
        raw.prefill_rope_kv(qkv, q1, k1, v1, cos, sin, pos0, heads, heads, D)
        for a, b, what in ((q, q1, "q"), (kc, k1, "k"), (vc, v1, "v")):     # (the two GEMMs may tile differently)
            assert (a.float() - b.float()).abs().max() <= 0.02 * b.float().abs().max() + 1e-3, what
            assert torch.equal(a == 0, b == 0), what
        assert k1[0, :, 9:9 + M].abs().sum() > 0 and k1[0, :, 9 + M:].abs().sum() == 0
