"""4-bit (NF4 / FP4) weight format: quantiser on the CPU, HIP kernels on the GPU."""
import pytest
import torch

from eventgpt_amd import quant as Q


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def test_pack_roundtrip_and_word_layout():
    idx = torch.randint(0, 16, (5, 128), dtype=torch.uint8, generator=torch.Generator().manual_seed(0))
    packed = Q.pack_codes(idx)
    assert packed.shape == (5, 64) and packed.dtype == torch.uint8
    assert torch.equal(Q.unpack_codes(packed), idx)
    # byte j of a word: weight j in the low nibble, weight 4+j in the high nibble
    one = torch.arange(8, dtype=torch.uint8)[None]
    assert Q.pack_codes(one)[0].tolist() == [0x40, 0x51, 0x62, 0x73]


@pytest.mark.parametrize("kind", ["nf4", "fp4"])
def test_quantize_error_bound_and_exact_levels(kind):
    w = _rand((16, 256), 1, 0.02)
    q = Q.quantize_weight(w, kind)
    assert q.shape == (16, 256) and q.absmax.shape == (16, 4) and q.nbytes == 16 * 128 + 16 * 4 * 4
    d = q.dequantize().float()
    levels = Q.codebook(kind).float().sort().values
    half_gap = (levels[1:] - levels[:-1]).max() / 2
    am = q.absmax.repeat_interleave(64, dim=1)
    # nearest level: the error is at most half the widest gap (plus bf16 rounding of the product)
    assert ((d - w).abs() <= half_gap * am * 1.01 + am * 2 ** -8).all()
    # weights that sit on levels come back exactly (up to the bf16 rounding of level * scale)
    on = (Q.codebook(kind).float()[torch.randint(0, 16, (4, 128), generator=torch.Generator().manual_seed(2))]) * 0.5
    on[:, 0] = 0.5; on[:, 64] = -0.5     # pin every block's absmax
    assert torch.allclose(Q.quantize_weight(on, kind).dequantize().float(), on, rtol=2 ** -8, atol=0)


def test_nf4_beats_fp4_on_gaussian_weights():
    w = _rand((64, 1024), 3)
    err = {k: (Q.quantize_weight(w, k).dequantize().float() - w).pow(2).mean().sqrt().item() for k in ("nf4", "fp4")}
    assert err["nf4"] < err["fp4"] and err["nf4"] / w.std().item() < 0.11


def test_double_quant_state_dict_roundtrip(tmp_path):
    q = Q.quantize_weight(_rand((32, 512), 4, 0.05))
    for dq in (True, False):
        torch.save(q.state_dict(double_quant=dq), tmp_path / "w.pt")
        q2 = Q.W4Weight.from_state_dict(torch.load(tmp_path / "w.pt", weights_only=True))
        assert torch.equal(q2.packed, q.packed) and torch.equal(q2.code, q.code) and q2.kind == "nf4"
        rel = ((q2.absmax - q.absmax).abs() / q.absmax).max().item()
        assert rel == 0 if not dq else rel < 0.02
    st = Q.quantize_absmax(q.absmax)
    assert st["q"].dtype == torch.uint8 and st["q"].shape == q.absmax.shape and st["scale"].numel() == 1


def test_fake_quantize_model_matches_matmul_reference():
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    cfg = EventChatConfig(vocab_size=128, hidden_size=64, intermediate_size=192, num_hidden_layers=2,
                          num_attention_heads=4, max_position_embeddings=64, torch_dtype="float32")
    m = EventChatModel.from_config(cfg, torch.float32, "cpu", seed=0)
    w0 = m.model.layers[0].mlp.down_proj.weight.data.clone()
    n = Q.fake_quantize_model_(m)
    assert n == 7 * m.config.num_hidden_layers
    q = Q.quantize_weight(w0)
    x = _rand((3, w0.shape[1]), 5)
    got = torch.nn.functional.linear(x, m.model.layers[0].mlp.down_proj.weight.data)
    assert torch.allclose(got, q.matmul_reference(x), rtol=2e-2, atol=2e-3)


# ------------------------------------------------------------------- GPU
def _dev():
    return torch.device("cuda", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["nf4", "fp4"])
def test_dequant_kernel_equals_oracle(kind):
    w = _rand((300, 704), 6, 0.03)
    q = Q.quantize_weight(w, kind)
    want = q.dequantize()
    got = q.to(_dev()).dequantize()
    assert got.dtype == torch.bfloat16 and torch.equal(got.cpu(), want)
    out = torch.zeros(300, 704, dtype=torch.bfloat16, device=_dev())
    assert q.to(_dev()).dequantize(out).data_ptr() == out.data_ptr() and torch.equal(out.cpu(), want)
    # quantising on the GPU gives the same codes
    qg = Q.quantize_weight(w.to(_dev()), kind)
    assert torch.equal(qg.packed.cpu(), q.packed) and torch.equal(qg.absmax.cpu(), q.absmax)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 2, 3])
@pytest.mark.parametrize("N,K", [(4096, 4096), (512, 11008), (12288, 256), (24576, 64), (260, 192), (4096, 11008),
                                 (8, 8192), (20004, 4160)])
def test_gemv_residual_w4(B, N, K):
    from eventgpt_amd.ops import hip
    q = Q.quantize_weight(_rand((N, K), N + K, 0.02)).to(_dev())
    x = _rand((B, K), 7).to(_dev(), torch.bfloat16)
    y0 = _rand((B, N), 8).to(_dev(), torch.bfloat16)
    y = y0.clone()
    hip.raw.decode_gemv_residual_w4(q.packed, q.absmax, q.code, x, y)
    proj = q.matmul_reference(x)
    want = proj.to(torch.bfloat16).float() + y0.float()
    tol = 2e-2 * (proj.abs() + y0.float().abs()) + 1e-3
    assert ((y.float() - want).abs() <= tol).all(), ((y.float() - want).abs() / tol).max().item()


def _close_to_bf16_kernel(a, b):
    """The 4-bit kernel scales fp32 sums of exact code values; the bf16 kernel multiplies weights that were
    rounded to bf16 after scaling.  Each projection so differs by up to ~1 % of the typical output (measured
    against an fp32 product of the codes: 0.8-0.9 % for either), and a gated or rotated output combines two."""
    a, b = a.float(), b.float()
    typical = b.abs()[b != 0].mean()   # a cache is zero except at the written position
    err = (a - b).abs()
    assert (err <= 3e-2 * b.abs() + 3e-3 + 5e-2 * typical).all(), (err.max().item(), typical.item())
    assert err[b != 0].mean() <= 1e-2 * typical + 1e-4, (err.mean().item(), typical.item())
    assert (a[b == 0] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("B,geom", [(1, (512, 1408, 8, 64)), (3, (512, 1408, 8, 64)),
                                    (1, (4096, 11008, 32, 128)),   # the 7B layer: several steps per wave
                                    (1, (2112, 2880, 6, 96))])     # rows end inside a 64-lane iteration
def test_mlp_up_and_qkv_w4_match_bf16_kernels_on_dequantised_weights(B, geom):
    """The fused-epilogue variants against the bf16 kernels fed the dequantised matrix (one row: the
    dedicated one-row kernel; more rows: the generic one)."""
    from eventgpt_amd import ops
    from eventgpt_amd.ops import hip
    raw = hip.raw
    dev = _dev()
    H, I, Hq, D = geom
    L = 64
    x = _rand((B, H), 9).to(dev, torch.bfloat16)
    nw = (1 + 0.1 * _rand((H,), 10)).to(dev, torch.bfloat16)
    qgu = Q.quantize_weight(_rand((2 * I, H), 11, 0.03)).to(dev)
    h4 = torch.empty(B, I, dtype=torch.bfloat16, device=dev)
    hb = torch.empty_like(h4)
    raw.decode_mlp_up_w4(qgu.packed, qgu.absmax, qgu.code, x, nw, 1e-6, h4)
    raw.decode_mlp_up(qgu.dequantize(), x, nw, 1e-6, hb)
    _close_to_bf16_kernel(h4, hb)

    qqkv = Q.quantize_weight(_rand((3 * Hq * D, H), 12, 0.03)).to(dev)
    cos, sin = ops.rope_tables(L, D, 10000.0, dev)
    pos = torch.tensor([5, 9, 0][:B], dtype=torch.int32, device=dev)
    outs = []
    for fmt in ("w4", "bf16"):
        qo = torch.zeros(B, Hq * D, dtype=torch.bfloat16, device=dev)
        kc = torch.zeros(B, Hq, L, D, dtype=torch.bfloat16, device=dev)
        vc = torch.zeros_like(kc)
        if fmt == "w4":
            raw.decode_qkv_w4(x, nw, qqkv.packed, qqkv.absmax, qqkv.code, qo, kc, vc, cos, sin, pos, 1e-6,
                              Hq, Hq, D, False)
        else:
            raw.decode_qkv(x, nw, qqkv.dequantize(), qo, kc, vc, cos, sin, pos, 1e-6, Hq, Hq, D, False)
        outs.append((qo, kc, vc))
    for a, b in zip(*outs):
        _close_to_bf16_kernel(a, b)
    assert outs[0][1].abs().sum() > 0


def _tiny_pair(fmt):
    """(4-bit engine on a model, bf16 engine on the fake-quantised copy of that model)."""
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    from eventgpt_amd.runtime.engine import GenerationEngine
    cfg = EventChatConfig(vocab_size=512, hidden_size=256, intermediate_size=704, num_hidden_layers=3,
                          num_attention_heads=4, max_position_embeddings=256)
    m4 = EventChatModel.from_config(cfg, torch.bfloat16, _dev(), seed=3)
    mb = EventChatModel.from_config(cfg, torch.bfloat16, _dev(), seed=3)
    Q.fake_quantize_model_(mb, fmt)
    e4 = GenerationEngine(m4, max_len=128, max_new_tokens=40, weight_format=fmt)
    eb = GenerationEngine(mb, max_len=128, max_new_tokens=40)
    return e4, eb


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["nf4", "fp4"])
def test_engine_w4_follows_bf16_engine_on_dequantised_weights(fmt):
    e4, eb = _tiny_pair(fmt)
    assert e4.w4 and e4.weight_bytes < 0.3 * eb.weight_bytes
    emb = _rand((21, 256), 13, 0.5).to(_dev(), torch.bfloat16)
    # prefill expands each matrix to exactly the bf16 engine's weights: same hidden states
    h4 = e4.prefill(emb, select="greedy", return_hidden=True).clone()
    hb = eb.prefill(emb, select="greedy", return_hidden=True).clone()
    assert torch.equal(h4, hb) and int(e4.cur_token[0]) == int(eb.cur_token[0])
    # decode: same arithmetic up to rounding order; the residual streams stay together
    for step in range(12):
        e4.decode_step(True); eb.decode_step(True)
        eb.cur_token.copy_(e4.cur_token)          # keep both on the 4-bit engine's token path
        eb.x.copy_(eb.embed[e4.cur_token.long()])
    torch.cuda.synchronize()
    k4, kb = e4.layers[-1].k_cache[0, :, :33].float(), eb.layers[-1].k_cache[0, :, :33].float()
    cos = torch.nn.functional.cosine_similarity(k4.flatten(), kb.flatten(), dim=0).item()
    assert cos > 0.999, cos


@pytest.mark.gpu
def test_engine_w4_generate_graph_equals_eager_and_verify_paths():
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    from eventgpt_amd.runtime.engine import GenerationEngine
    cfg = EventChatConfig(vocab_size=512, hidden_size=256, intermediate_size=704, num_hidden_layers=2,
                          num_attention_heads=4, max_position_embeddings=256)
    m = EventChatModel.from_config(cfg, torch.bfloat16, _dev(), seed=4)
    emb = _rand((17, 256), 14, 0.5).to(_dev(), torch.bfloat16)
    eg = GenerationEngine(m, max_len=128, max_new_tokens=40, weight_format="nf4", use_graph=True)
    ee = GenerationEngine(m, max_len=128, max_new_tokens=40, weight_format="nf4", use_graph=False)
    tg = eg.generate(emb, 24, eos_token_id=-1)
    te = ee.generate(emb, 24, eos_token_id=-1)
    assert torch.equal(tg, te) and tg.numel() == 24
    # replays after eager work in between give the same tokens again
    _ = ee.generate(emb, 5, eos_token_id=-1)
    assert torch.equal(eg.generate(emb, 24, eos_token_id=-1), te)
    # the memory-lean mode (bf16 weights dropped, matrices expanded on the fly) gives the same tokens
    m2 = EventChatModel.from_config(cfg, torch.bfloat16, _dev(), seed=4)
    er = GenerationEngine(m2, max_len=128, max_new_tokens=40, weight_format="nf4", release_bf16=True)
    assert er.layers[0].w_qkv is None and torch.equal(er.generate(emb, 24, eos_token_id=-1), te)
    # by default the module path computes the same quantised model (its weights were rewritten in place)
    w_mod = m.model.layers[0].self_attn.q_proj.weight.data
    assert torch.equal(w_mod, eg.layers[0].q4["w_qkv"].dequantize()[: w_mod.shape[0]])
    # speculative verify: drafting the engine's own continuation is fully accepted, both on the
    # GEMV path (2 drafts -> 3 rows) and on the dequantise + GEMM path (6 drafts -> 7 rows)
    for G in (2, 6):
        eg.reset(); eg.prefill(emb, select="greedy")
        draft = te[1:1 + G].to(torch.int32)
        eg.verify(draft)
        res = eg.sd_result.tolist()
        assert res[0] == G and res[1] == int(te[1 + G]), (G, res, te[:G + 2].tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("release", [False, True])
def test_batched_engine_w4_more_streams_than_the_4bit_gemv_takes(release):
    """Six streams on a 4-bit batched engine (one rank): the step multiplies by the dequantised matrices with
    the bf16 kernels, so it is the bf16 engine on the fake-quantised model, token for token -- also when the
    bf16 copies were dropped and every matrix is expanded into scratch where it is used."""
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    from eventgpt_amd.parallel.tp import TensorParallelEngine
    from eventgpt_amd.ops import hip
    cfg = EventChatConfig(vocab_size=512, hidden_size=256, intermediate_size=704, num_hidden_layers=2,
                          num_attention_heads=4, max_position_embeddings=256)
    B = int(hip.raw.gemv_w4_max_batch()) + 3
    m4 = EventChatModel.from_config(cfg, torch.bfloat16, _dev(), seed=5)
    mb = EventChatModel.from_config(cfg, torch.bfloat16, _dev(), seed=5)
    Q.fake_quantize_model_(mb, "nf4")
    e4 = TensorParallelEngine(m4, None, max_len=128, max_batch=B, max_new_tokens=24, weight_format="nf4",
                              release_bf16=release)
    eb = TensorParallelEngine(mb, None, max_len=128, max_batch=B, max_new_tokens=24)
    embs = [_rand((9 + 2 * b, 256), 20 + b, 0.5).to(_dev(), torch.bfloat16) for b in range(B)]
    e4.prefill_batch(embs)
    eb.prefill_batch(embs)
    for _ in range(10):
        e4.decode_step(True); eb.decode_step(True)
    torch.cuda.synchronize()
    assert torch.equal(e4.out_tokens[:, :11], eb.out_tokens[:, :11])
    assert torch.equal(e4.pos, eb.pos)
    with pytest.raises(RuntimeError, match="TensorParallelEngine"):
        from eventgpt_amd.runtime.engine import GenerationEngine
        GenerationEngine(mb, max_len=128, max_batch=B, max_new_tokens=8, use_graph=False).decode_step(True)
