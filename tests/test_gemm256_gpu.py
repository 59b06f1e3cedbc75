"""The 256 x 256 8-wave GEMM (phase-staggered wave rows) forced onto shapes that exercise its edges:
rows / columns that do not fill a tile, one K-tile, odd and even K-tile counts, every epilogue, the
K-strided operand layouts of the backward products; each against an fp32 matmul.  Every case runs
several times: its synchronisation is counted waits and barriers, so a race would show as a
difference between repeats."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(*shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed + sum(shape))
    return (torch.randn(*shape, device=dev, generator=g) * scale).to(torch.bfloat16)


@pytest.fixture()
def force256():
    from eventgpt_amd.ops import hip
    hip.raw.gemm_force_config(5)
    yield hip.raw
    hip.raw.gemm_force_config(-1)


def _check(y, want, K):
    err = (y.float() - want).abs()
    tol = 2e-2 * math.sqrt(K / 64) * float(want.abs().max()) / 8 + 2e-2 * want.abs()
    assert bool((err <= tol).all()), (float(err.max()), float(want.abs().max()))


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (256, 256, 192), (512, 768, 4096), (1000, 520, 1024),
                                   (37, 4096, 4096), (2115, 4096, 11008), (4643, 1000, 1024), (300, 264, 320)])
def test_forced_256_matches_fp32(dev, force256, M, N, K):
    x, w = _rand(M, K, dev=dev), _rand(N, K, dev=dev, seed=1, scale=0.05)
    want = x.float() @ w.float().t()
    first = None
    for rep in range(4):
        y = force256.linear(x, w, None, None, 0, False)
        _check(y, want, K)
        first = y if first is None else first
        assert torch.equal(y, first), rep            # deterministic: same schedule, same bits


def test_forced_256_epilogues(dev, force256):
    M, N, K = 700, 1032, 512
    x, w = _rand(M, K, dev=dev), _rand(N, K, dev=dev, seed=1, scale=0.05)
    bias, res = _rand(N, dev=dev, seed=2), _rand(M, N, dev=dev, seed=3)
    lin = x.float() @ w.float().t() + bias.float()
    y = force256.linear(x, w, bias, res, 1, False)                      # bias + GELU + residual
    want = torch.nn.functional.gelu(lin).to(torch.bfloat16).float() + res.float()
    _check(y, want, K)
    yf = force256.linear(x, w, bias, None, 0, True)                      # fp32 output
    assert yf.dtype == torch.float32
    _check(yf, lin, K)
    pv, pi = force256.linear_argmax_partials(x, w, bias)
    got = pi.gather(1, pv.argmax(1, keepdim=True))[:, 0].long()
    assert torch.equal(lin.gather(1, got[:, None]).to(torch.bfloat16), lin.max(1, keepdim=True).values.to(torch.bfloat16))


@pytest.mark.parametrize("M,N,K", [(1024, 512, 4096), (836, 4096, 1000), (2115, 1024, 264), (264, 11008, 2120)])
def test_forced_256_kstrided_operands(dev, force256, M, N, K):
    """dX = dY W (weight K-strided) and dW = dY^T X (both K-strided) on the 8-wave kernel."""
    from eventgpt_amd.ops import differentiable as diff
    gy, w, x = _rand(M, N, dev=dev, scale=0.5), _rand(N, K, dev=dev, seed=1, scale=0.05), _rand(M, K, dev=dev, seed=2)
    for rep in range(3):
        dx = diff.linear_dgrad(gy, w)
        _check(dx, gy.float() @ w.float(), N)
        dw = diff.linear_wgrad(gy, x)
        _check(dw, gy.float().t() @ x.float(), M)


def test_forced_256_identity_exact(dev, force256):
    """A = I against an asymmetric operand: a wrong row / column / k map gives exact mismatches."""
    n = 512
    eye = torch.eye(n, device=dev, dtype=torch.bfloat16)
    w = (torch.arange(n * 320, device=dev).reshape(320, n) % 251).to(torch.bfloat16)     # [N=320][K=n]
    assert torch.equal(force256.linear(eye, w, None, None, 0, False), w.t())
    wt = (torch.arange(n * 264, device=dev).reshape(n, 264) % 241).to(torch.bfloat16)    # [K=n][N=264]
    assert torch.equal(force256.linear_t(eye, wt, False, True, None, None, 0, False, n), wt)
    a = (torch.arange(n * 328, device=dev).reshape(n, 328) % 239).to(torch.bfloat16)     # [K=n][M=328]
    assert torch.equal(force256.linear_t(a, wt, True, True, None, None, 0, True, n), a.float().t() @ wt.float())
    assert torch.equal(force256.linear_t(a, eye, True, False, None, None, 0, False, n), a.t())


def test_auto_dispatch_uses_256_for_large_shapes(dev):
    from eventgpt_amd.ops import hip
    x, w = _rand(4643, 4096, dev=dev), _rand(12288, 4096, dev=dev, seed=1, scale=0.05)
    y_auto = hip.raw.linear(x, w, None, None, 0, False)
    hip.raw.gemm_enable_256(False)
    try:
        y_old = hip.raw.linear(x, w, None, None, 0, False)
    finally:
        hip.raw.gemm_enable_256(True)
    want = x.float() @ w.float().t()
    _check(y_auto, want, 4096)
    _check(y_old, want, 4096)


@pytest.mark.parametrize("M,N,K", [(4643, 4608, 4096), (4643, 4096 + 264, 4096), (4643, 4096, 512), (2600, 6400, 128)])
def test_auto_dispatch_column_split_for_partial_rounds(dev, M, N, K):
    """Tile counts just over whole rounds of CUs (342 = 256 + 86, 323 = 256 + 67) run as two launches over
    disjoint column ranges -- the tail on the 4-wave kernel or K-split on the 8-wave one; every epilogue operand is
    offset with them.  (Contractions under 2048 take the 4-wave tiles throughout.)"""
    from eventgpt_amd.ops import hip
    x, w = _rand(M, K, dev=dev), _rand(N, K, dev=dev, seed=1, scale=0.05)
    bias, res = _rand(N, dev=dev, seed=2), _rand(M, N, dev=dev, seed=3)
    lin = x.float() @ w.float().t() + bias.float()
    y = hip.raw.linear(x, w, bias, res, 1, False)
    want = torch.nn.functional.gelu(lin).to(torch.bfloat16).float() + res.float()
    _check(y, want, K)
    yf = hip.raw.linear(x, w, bias, None, 0, True)
    _check(yf, lin, K)
    r2 = res.clone()
    y2 = hip.raw.linear(x, w, None, r2, 0, False)          # plain residual
    _check(y2, (x.float() @ w.float().t()).to(torch.bfloat16).float() + res.float(), K)


@pytest.mark.parametrize("M,I,K,cfg", [(636, 11008, 4096, -1), (4643, 1408, 512, -1), (100, 96, 64, -1), (300, 2816, 1024, 0),
                                       (300, 2816, 1024, 1), (300, 2816, 1024, 2), (1030, 704, 256, 5), (257, 32, 128, 5)])
def test_linear_swiglu_epilogue(dev, M, I, K, cfg):
    """Gate/up projection with the activation in the GEMM epilogue == GEMM, then the activation kernel
    (bitwise: both round the two factors to bf16 before multiplying), on every tile config."""
    from eventgpt_amd.ops import hip
    x, w = _rand(M, K, dev=dev), _rand(2 * I, K, dev=dev, seed=1, scale=0.05)
    hip.raw.gemm_force_config(cfg)
    try:
        fused = hip.raw.linear_swiglu(x, w)
        again = hip.raw.linear_swiglu(x, w)
    finally:
        hip.raw.gemm_force_config(-1)
    plain = hip.raw.swiglu(hip.raw.linear(x, w, None, None, 0, False))
    assert fused.shape == (M, I) and torch.equal(fused, again)
    gu = x.float() @ w.float().t()
    want = torch.nn.functional.silu(gu[:, :I]) * gu[:, I:]
    _check(fused, want, K)
    assert torch.allclose(fused.float(), plain.float(), atol=2e-2 * float(want.abs().max()), rtol=2e-2)


@pytest.mark.parametrize("M,N,K", [(4643, 4096, 4096), (4643, 4096, 11008), (4000, 4608, 4160),
                                   (2115, 4096, 11008), (1500, 4096, 8192)])   # the last two: one partial round, all tiles split
def test_last_round_tiles_with_k_split(dev, M, N, K):
    """Default dispatch of a shape whose 256 x 256 tiles are one full round plus a few: the few run the
    8-wave kernel with K split over the idle CUs (fp32 slabs + reduce, bias / activation / residual applied
    by the reduce).  Same result as with the split switched off (those columns on the 4-wave kernel) up to
    the summation order, repeatable, and right against fp32."""
    from eventgpt_amd.ops import hip
    raw = hip.raw
    x, w = _rand(M, K, dev=dev), _rand(N, K, dev=dev, seed=1, scale=0.05)
    bias, res = _rand(N, dev=dev, seed=2), _rand(M, N, dev=dev, seed=3)
    want = torch.nn.functional.gelu(x.float() @ w.float().t() + bias.float()).to(torch.bfloat16).float() + res.float()
    try:
        raw.gemm_enable_tail_splitk(True)
        first = None
        for _ in range(3):
            y = raw.linear(x, w, bias, res, 1, False)
            first = y if first is None else first
            assert torch.equal(y, first)
        raw.gemm_enable_tail_splitk(False)
        off = raw.linear(x, w, bias, res, 1, False)
    finally:
        raw.gemm_enable_tail_splitk(True)
    _check(first, want, K)
    assert torch.allclose(first.float(), off.float(), atol=3e-2, rtol=2e-2)
    # fp32 output path as well
    yf = raw.linear(x, w, None, None, 0, True)
    _check(yf, x.float() @ w.float().t(), K)


@pytest.mark.parametrize("M,I,K", [(2115, 4096, 1024), (1500, 11008, 512), (2115, 11008, 4096)])
def test_swiglu_gemm_last_tiles_on_small_tiles(dev, M, I, K):
    """The gate/up GEMM with the SwiGLU epilogue when its 256 x 128 output tiles are whole rounds plus a
    handful: those columns run 64 x 64 tiles (a second launch over a column range of the packed
    [gate | up] matrix).  Same result as the single launch, right against fp32."""
    from eventgpt_amd.ops import hip
    raw = hip.raw
    x, w = _rand(M, K, dev=dev), _rand(2 * I, K, dev=dev, seed=1, scale=0.05)
    gu = x.float() @ w.float().t()
    want = torch.nn.functional.silu(gu[:, :I].to(torch.bfloat16).float()) * gu[:, I:].to(torch.bfloat16).float()
    try:
        raw.gemm_enable_tail_splitk(True)
        y = raw.linear_swiglu(x, w)
        assert torch.equal(y, raw.linear_swiglu(x, w))
        raw.gemm_enable_tail_splitk(False)
        one = raw.linear_swiglu(x, w)
    finally:
        raw.gemm_enable_tail_splitk(True)
    _check(y, want, K)
    assert torch.allclose(y.float(), one.float(), atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("M,N,K", [(128, 256, 64), (128, 256, 192), (636, 4096, 4096), (636, 12288, 4096), (300, 264, 320),
                                   (1000, 520, 1024), (37, 4096, 512)])
def test_forced_128_row_variant_matches_fp32(dev, M, N, K):
    """The 8-phase schedule on a 128 x 256 tile (config 6): values, determinism, bias / activation / residual
    epilogue, an identity operand against an asymmetric one, and K split over its slabs."""
    from eventgpt_amd.ops import hip
    raw = hip.raw
    x, w = _rand(M, K, dev=dev), _rand(N, K, dev=dev, seed=1, scale=0.05)
    want = x.float() @ w.float().t()
    raw.gemm_force_config(6)
    try:
        first = None
        for rep in range(3):
            y = raw.linear(x, w, None, None, 0, False)
            _check(y, want, K)
            first = y if first is None else first
            assert torch.equal(y, first), rep
        bias, res = _rand(N, dev=dev, seed=2), _rand(M, N, dev=dev, seed=3)
        y = raw.linear(x, w, bias, res, 1, False)
        _check(y, torch.nn.functional.gelu(want + bias.float()).to(torch.bfloat16).float() + res.float(), K)
        if K >= 1024:
            raw.gemm_force_splitk(2)
            _check(raw.linear(x, w, None, None, 0, False), want, K)
            raw.gemm_force_splitk(0)
        n = 256
        eye = torch.eye(n, device=dev, dtype=torch.bfloat16)
        wa = (torch.arange(n * 520, device=dev).reshape(520, n) % 251).to(torch.bfloat16)
        assert torch.equal(raw.linear(eye, wa, None, None, 0, False), wa.t())
    finally:
        raw.gemm_force_splitk(0)
        raw.gemm_force_config(-1)


@pytest.mark.parametrize("M,N,K", [(4643, 4608, 192), (4643, 4608, 64), (5000, 5000, 1024), (4643, 12288, 512),
                                   (23080, 1024, 320)])
def test_forced_256_walks_several_tiles_per_workgroup(dev, force256, M, N, K):
    """More tiles than CUs: a workgroup runs its tiles one after the other and stages the next tile's first
    K-tile under the epilogue of the current one.  Values, determinism, bias + residual, and equality with the
    one-workgroup-per-tile launch."""
    x, w = _rand(M, K, dev=dev), _rand(N, K, dev=dev, seed=1, scale=0.05)
    bias, res = _rand(N, dev=dev, seed=2), _rand(M, N, dev=dev, seed=3)
    want = x.float() @ w.float().t()
    y = force256.linear(x, w, None, None, 0, False)
    _check(y, want, K)
    for rep in range(3):
        assert torch.equal(force256.linear(x, w, None, None, 0, False), y), rep
    yb = force256.linear(x, w, bias, res, 0, False)
    _check(yb, (want + bias.float()).to(torch.bfloat16).float() + res.float(), K)
    force256.gemm_persist_256(False)
    try:
        assert torch.equal(force256.linear(x, w, None, None, 0, False), y)
        assert torch.equal(force256.linear(x, w, bias, res, 0, False), yb)
    finally:
        force256.gemm_persist_256(True)


def test_persistent_256_with_k_split_glu_and_kstrided_operands(dev, force256):
    from eventgpt_amd.ops import differentiable as diff
    # K split: tile x split pairs beyond the CU count
    x, w = _rand(2600, 2048, dev=dev), _rand(2560, 2048, dev=dev, seed=1, scale=0.05)
    force256.gemm_force_splitk(4)
    try:
        _check(force256.linear(x, w, None, None, 0, False), x.float() @ w.float().t(), 2048)
    finally:
        force256.gemm_force_splitk(0)
    # SwiGLU epilogue, 19 x 22 = 418 tiles
    x, wg = _rand(4643, 512, dev=dev), _rand(2 * 2816, 512, dev=dev, seed=2, scale=0.05)
    h = force256.linear_swiglu(x, wg)
    g, u = (x.float() @ wg.float().t()).to(torch.bfloat16).float().chunk(2, -1)
    _check(h, torch.nn.functional.silu(g) * u, 512)
    # K-strided operands: dgrad and wgrad of a 5000-row batch
    gy, w2, x2 = _rand(5000, 4608, dev=dev, scale=0.5), _rand(4608, 4352, dev=dev, seed=1, scale=0.05), _rand(5000, 4352, dev=dev, seed=2)
    _check(diff.linear_dgrad(gy, w2), gy.float() @ w2.float(), 4608)
    _check(diff.linear_wgrad(gy, x2), gy.float().t() @ x2.float(), 5000)
