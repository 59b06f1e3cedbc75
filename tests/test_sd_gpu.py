"""Speculative decoding on the GPU: losslessness, fused verify kernel, scheduler."""
import pytest
import torch

from eventgpt_amd.models import EventChatConfig, EventChatModel

pytestmark = pytest.mark.gpu


def _cfg(layers=2):
    return EventChatConfig(vocab_size=128, hidden_size=512, intermediate_size=1408,
                           num_hidden_layers=layers, num_attention_heads=4,
                           max_position_embeddings=1024)


@pytest.fixture(scope="module")
def target(dev):
    return EventChatModel.from_config(_cfg(3), torch.bfloat16, dev, seed=11)


@pytest.fixture(scope="module")
def drafter(dev):
    return EventChatModel.from_config(_cfg(1), torch.bfloat16, dev, seed=12)


from _lossless import assert_lossless as _assert_lossless  # noqa: E402


def _emb(dev, n=40, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(n, 512, device=dev, generator=g) * 0.5).to(torch.bfloat16)


def test_verify_accept_kernel_cases(target, dev):
    from eventgpt_amd.ops import hip
    eng = target.engine(max_len=512, max_new_tokens=128)
    emb = _emb(dev)
    greedy = eng.generate(emb, 12, eos_token_id=-1)
    # all-correct draft: accepted gamma, bonus token follows
    eng.reset(); eng.eos = -1
    eng.prefill(emb, select="greedy")
    pos0 = int(eng.pos[0])
    draft = greedy[1:6].to(torch.int32)
    eng.verify(draft, 5)
    n_acc, nxt, emitted = eng.sd_result[:3].tolist()
    assert (n_acc, emitted) == (5, 6) and nxt == int(greedy[6])
    assert int(eng.pos[0]) == pos0 + 6 and int(eng.step[0]) == 7
    assert torch.equal(eng.out_tokens[0, :7].long(), greedy[:7])
    # continue autoregressively from the rolled state: still the greedy sequence
    rest = eng.generate_continue(12, -1)
    assert torch.equal(rest, greedy)
    # wrong at index 2: two accepted, correction emitted, cache rolled back
    eng.reset(); eng.prefill(emb, select="greedy")
    bad = greedy[1:6].to(torch.int32).clone()
    bad[2] = (bad[2] + 1) % 128
    eng.verify(bad, 5)
    n_acc, nxt, emitted = eng.sd_result[:3].tolist()
    assert (n_acc, emitted) == (2, 3) and nxt == int(greedy[3])
    assert int(eng.pos[0]) == pos0 + 3
    assert torch.equal(eng.generate_continue(12, -1), greedy)
    # EOS inside the accepted run stops emission there
    eng.reset(); eng.prefill(emb, select="greedy")
    k = next(i for i in range(1, 6) if int(greedy[i]) not in greedy[1:i].tolist())
    eng.eos = int(greedy[k])
    eng.verify(greedy[1:6].to(torch.int32), 5)
    assert int(eng.done[0]) == 1 and int(eng.step[0]) == k + 1


def test_verify_graph_replays_match_eager(target, dev):
    """The captured verify is replayed with other eager work and syncs in between;
    every replay must equal the eager (uncaptured) verify from the same state."""
    from eventgpt_amd.runtime.engine import GenerationEngine
    emb = _emb(dev, 37, seed=9)
    eager = GenerationEngine(target, max_len=512, max_new_tokens=128, use_graph=False)
    graph = GenerationEngine(target, max_len=512, max_new_tokens=128, use_graph=True)
    greedy = graph.generate(emb, 40, eos_token_id=-1)
    for e in (eager, graph):
        e.eos = -1
        e.reset()
        e.prefill(emb, select="greedy")
    g = torch.Generator(device=dev).manual_seed(1)
    for it in range(7):
        n = int(graph.step[0].item())
        draft = greedy[n:n + 4].to(torch.int32).clone()
        if it % 2:                                  # corrupt one position on odd rounds
            k = int(torch.randint(0, 4, (1,), device=dev, generator=g))
            draft[k] = (draft[k] + 1) % 128
        he = eager.verify(draft, 4, return_hidden=True)
        hg = graph.verify(draft, 4, return_hidden=True)
        _ = (torch.randn(256, 256, device=dev) @ torch.randn(256, 256, device=dev)).sum().item()
        assert torch.equal(eager.sd_result[:3], graph.sd_result[:3]), it
        assert torch.equal(eager.pos, graph.pos) and torch.equal(eager.step, graph.step)
        assert torch.allclose(he.float(), hg.float(), atol=2e-2, rtol=2e-2)
    n = int(graph.step[0].item())
    assert torch.equal(graph.out_tokens[0, :n].long(), greedy[:n])
    assert torch.equal(eager.out_tokens[0, :n].long(), greedy[:n])


@pytest.mark.parametrize("device_loop", [True, False])
def test_naive_sd_is_lossless(target, drafter, dev, device_loop):
    from eventgpt_amd.runtime.spec_decode import NaiveSpeculativeDecoder
    emb = _emb(dev, 33, seed=3)
    t_eng = target.engine(max_len=512, max_new_tokens=128)
    greedy = t_eng.generate(emb, 30, eos_token_id=-1)
    d_eng = drafter.engine(max_len=512, max_new_tokens=128)
    sd = NaiveSpeculativeDecoder(d_eng, t_eng)
    toks, hist = sd.generate(emb, None, 30, 4, -1, device_loop=device_loop)
    _assert_lossless(t_eng, emb, toks, greedy[:len(toks)])
    assert len(toks) >= 30 and sd.stats.iterations > 0 and sd.stats.total_drafted == 4 * sd.stats.iterations
    # the target drafting for itself accepts everything
    twin = EventChatModel.from_config(_cfg(3), torch.bfloat16, dev, seed=11)
    sd2 = NaiveSpeculativeDecoder(twin.engine(max_len=512, max_new_tokens=128), t_eng)
    toks2, hist2 = sd2.generate(emb, None, 30, 4, -1, device_loop=device_loop)
    _assert_lossless(t_eng, emb, toks2[:30], greedy[:30])
    assert sd2.stats.accept_rate > 0.9               # a drafter with the target's weights is accepted (ties aside)
    assert sd2.stats.iterations <= 8                 # 30 tokens in rounds of up to 5


def test_naive_sd_device_rounds_replay_and_prompt_offset(target, drafter, dev):
    """The captured round is replayed across calls with other work in between, with drafter and
    target prompts of different lengths (the drafter's positions are the target's plus an offset),
    and with an EOS id that ends the sequence inside a round."""
    from eventgpt_amd.runtime.spec_decode import NaiveSpeculativeDecoder
    t_emb, d_emb = _emb(dev, 57, seed=21), _emb(dev, 23, seed=22)
    t_eng = target.engine(max_len=512, max_new_tokens=128)
    d_eng = drafter.engine(max_len=512, max_new_tokens=128)
    sd = NaiveSpeculativeDecoder(d_eng, t_eng)
    greedy = t_eng.generate(t_emb, 40, eos_token_id=-1)
    for rep in range(3):
        toks, _ = sd.generate(d_emb, t_emb, 40, 3, -1, poll_every=2 + rep)
        _ = (torch.randn(256, 256, device=dev) @ torch.randn(256, 256, device=dev)).sum().item()
        _assert_lossless(t_eng, t_emb, toks, greedy[:len(toks)])
        assert len(toks) == 40
    assert len(sd._graphs) == 1
    # the host-driven loop gives the same tokens from the same prompts
    toks_h, _ = sd.generate(d_emb, t_emb, 40, 3, -1, device_loop=False)
    _assert_lossless(t_eng, t_emb, toks_h, greedy[:40])
    # EOS: stop at the first occurrence of a token of the greedy answer
    k = next(i for i in range(3, 30) if int(greedy[i]) not in greedy[:i].tolist())
    toks_e, _ = sd.generate(d_emb, t_emb, 40, 3, int(greedy[k]))
    assert int(t_eng.done[0]) == 1 and int(toks_e[-1]) == int(greedy[k])
    _assert_lossless(t_eng, t_emb, toks_e, greedy[:len(toks_e)])


@pytest.mark.timeout(600)
def test_naive_sd_lossless_at_7b_layer_geometry(dev):
    """Two layers of the 7B geometry (4096 wide, 32 heads of 128, 11008 intermediate, 32003 tokens):
    a drafter with the target's own weights is accepted throughout, an unrelated one almost never,
    and both leave the target's greedy answer."""
    from eventgpt_amd.runtime.spec_decode import NaiveSpeculativeDecoder
    cfg = EventChatConfig(vocab_size=32003, hidden_size=4096, intermediate_size=11008, num_hidden_layers=2,
                          num_attention_heads=32, max_position_embeddings=4096, torch_dtype="bfloat16")
    tgt = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=5)
    twin = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=5)
    other = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=6)
    g = torch.Generator(device=dev).manual_seed(1)
    emb = (torch.randn(300, 4096, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    d_emb = emb[:120].clone()
    t_eng = tgt.engine(max_len=1024, max_new_tokens=128)
    greedy = t_eng.generate(emb, 48, eos_token_id=-1)
    sd = NaiveSpeculativeDecoder(twin.engine(max_len=1024, max_new_tokens=128), t_eng)
    toks, _ = sd.generate(emb, None, 48, 5, -1)
    _assert_lossless(t_eng, emb, toks, greedy[:48])
    assert sd.stats.accept_rate > 0.85 and sd.stats.iterations <= 12
    sd0 = NaiveSpeculativeDecoder(other.engine(max_len=1024, max_new_tokens=128), t_eng)
    toks0, _ = sd0.generate(d_emb, emb, 48, 5, -1)
    _assert_lossless(t_eng, emb, toks0, greedy[:48])
    assert sd0.stats.iterations >= 20


@pytest.mark.parametrize("mode,level", [("adapter", 1), ("adapter", 4), ("vlm_only", 5), ("fused", 6)])
def test_adapter_sd_modes_are_lossless(target, drafter, dev, mode, level):
    from eventgpt_amd.align.adapters import create_adapter
    from eventgpt_amd.runtime.spec_decode import (AdapterDraftHead, SpecDecodeEngine,
                                                  collect_decode_hidden)
    emb = _emb(dev, 28, seed=5)
    v = target.engine(max_len=512, max_new_tokens=128)
    greedy = v.generate(emb, 24, eos_token_id=-1)
    d = drafter.engine(max_len=512, max_new_tokens=128)
    dh = collect_decode_hidden(d, emb, 24)
    assert dh["hidden"].shape == (24, 512)
    torch.manual_seed(0)
    adapter = create_adapter(level, 512, bottleneck_dim=32, num_heads=4, ffn_dim=64).to(dev).to(torch.bfloat16)
    head = AdapterDraftHead(adapter, target.lm_head.weight.data)
    v.reset(); v.eos = -1
    h = v.prefill(emb, select="greedy", return_hidden=True)
    from eventgpt_amd import ops
    vh0 = ops.rmsnorm(h[-1:], v.final_norm, v.eps)
    toks, stats = SpecDecodeEngine(v, head, gamma=4).run(dh["hidden"], 24, -1, mode, vh0)
    _assert_lossless(v, emb, toks, greedy[:24])
    assert stats.total_drafted > 0 and stats.iterations > 0


def test_prefill_hiding_and_parallel_prefill(target, drafter, dev):
    from eventgpt_amd.runtime.scheduler import DualModelScheduler
    d = drafter.engine(max_len=512, max_new_tokens=128)
    v = target.engine(max_len=512, max_new_tokens=128)
    sch = DualModelScheduler(d, v)
    d_emb, v_emb = _emb(dev, 20, seed=7), _emb(dev, 200, seed=8)
    base = sch.verifier_baseline(v_emb, 20, -1)
    res = sch.prefill_hiding(d_emb, v_emb, max_new_tokens=20, max_free=8, eos_token_id=-1)
    assert torch.equal(res["tokens"], base["tokens"])
    assert 1 <= res["n_free"] <= 8
    pp = sch.parallel_prefill(d_emb, v_emb)
    assert pp["wall_ms"] > 0 and 0.0 <= pp["overlap_ratio"] <= 1.0


def test_device_sampler_distribution(dev):
    from eventgpt_amd.ops import hip
    V = 1003
    logits = torch.full((1, V), -20.0, device=dev)
    logits[0, [5, 77, 900]] = torch.tensor([2.0, 1.0, 0.0], device=dev)
    out = torch.zeros(1, dtype=torch.int32, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    counts = {}
    for i in range(600):
        step.fill_(i)
        hip.raw.sample_top_p(logits, out, 1.0, 1.0, 1234, step)
        counts[int(out[0])] = counts.get(int(out[0]), 0) + 1
    p = torch.softmax(torch.tensor([2.0, 1.0, 0.0]), 0)
    assert set(counts) <= {5, 77, 900}
    assert abs(counts.get(5, 0) / 600 - p[0]) < 0.07 and abs(counts.get(77, 0) / 600 - p[1]) < 0.07
    # tight nucleus keeps only the mode; low temperature too
    for i in range(50):
        step.fill_(i)
        hip.raw.sample_top_p(logits, out, 1.0, 0.5, 99, step)
        assert int(out[0]) == 5
    big = torch.randn(2, 32003, device=dev)
    out2 = torch.zeros(2, dtype=torch.int32, device=dev)
    hip.raw.sample_top_p(big, out2, 0.05, 1.0, 7, torch.zeros(2, dtype=torch.int32, device=dev))
    assert torch.equal(out2.long(), big.argmax(-1))


def test_device_sampler_top_k(dev):
    """top-k keeps exactly the k most likely tokens (all of a tie at the k-th value), then
    top-p acts on what is left -- the order of temperature / top-k / top-p warpers."""
    from eventgpt_amd.ops import hip
    g = torch.Generator(device=dev).manual_seed(3)
    V = 32003
    logits = torch.randn(1, V, device=dev, generator=g)
    out = torch.zeros(1, dtype=torch.int32, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    keep = set(logits[0].topk(7).indices.tolist())
    seen = set()
    for i in range(300):
        step.fill_(i)
        hip.raw.sample_top_p(logits, out, 3.0, 1.0, 5, step, 7)
        seen.add(int(out[0]))
    assert seen <= keep and len(seen) >= 5
    # a tie at the k-th value keeps both tied tokens
    tie = torch.full((1, 1003), -9.0, device=dev)
    tie[0, [3, 10, 20, 30]] = torch.tensor([1.0, 0.5, 0.5, 0.0], device=dev)
    seen = set()
    for i in range(200):
        step.fill_(i)
        hip.raw.sample_top_p(tie, out, 1.0, 1.0, 11, step, 2)
        seen.add(int(out[0]))
    assert seen == {3, 10, 20}
    # top-k 3 then top-p 0.5 of the renormalised three: only the mode survives
    for i in range(50):
        step.fill_(i)
        hip.raw.sample_top_p(tie, out, 1.0, 0.4, 11, step, 3)
        assert int(out[0]) == 3
    # k = 1 is greedy
    hip.raw.sample_top_p(logits, out, 1.0, 1.0, 1, step, 1)
    assert int(out[0]) == int(logits.argmax())


def test_linear_argmax_partials(dev):
    from eventgpt_amd.ops import hip
    x = (torch.randn(22, 512, device=dev)).to(torch.bfloat16)
    w = (torch.randn(1003, 512, device=dev) * 0.05).to(torch.bfloat16)
    pv, pi = hip.raw.linear_argmax_partials(x, w, None)
    got = pi.gather(1, pv.argmax(1, keepdim=True))[:, 0].long()
    ref = hip.linear(x, w).float()
    assert torch.equal(ref.gather(1, got[:, None]), ref.max(1, keepdim=True).values)


@pytest.mark.parametrize("G", [1, 2])
def test_short_verify_on_decode_kernels(target, dev, G):
    """Verifies of up to 3 rows run on the decode step's fused weight-streaming kernels
    (one-sequence QKV epilogue); they must leave the same cache rows, hidden states and
    accept decisions as the generic row path."""
    from eventgpt_amd.runtime.engine import GenerationEngine
    emb = _emb(dev, 29, seed=4)
    fast = GenerationEngine(target, max_len=256, max_new_tokens=64, use_graph=False)
    slow = GenerationEngine(target, max_len=256, max_new_tokens=64, use_graph=False)
    slow.gemv_rows = 0                                   # force the generic path
    assert fast._gemv_rows_ok(G + 1) and not slow._gemv_rows_ok(G + 1)
    greedy = fast.generate(emb, 12, eos_token_id=-1)
    for e in (fast, slow):
        e.eos = -1
        e.reset()
        e.prefill(emb, select="greedy")
    for it in range(3):
        n = int(fast.step[0].item())
        draft = greedy[n:n + G].to(torch.int32).clone()
        if it == 1:
            draft[-1] = (draft[-1] + 1) % 128            # reject the last draft token
        hf = fast.verify(draft, G, return_hidden=True)
        hs = slow.verify(draft, G, return_hidden=True)
        assert torch.equal(fast.sd_result[:3], slow.sd_result[:3]), (it, fast.sd_result, slow.sd_result)
        assert torch.equal(fast.pos, slow.pos)
        assert torch.allclose(hf.float(), hs.float(), atol=3e-2, rtol=3e-2)
        p = int(fast.pos[0].item())
        for lf, ls in zip(fast.layers, slow.layers):
            assert torch.allclose(lf.k_cache[0, :, :p].float(), ls.k_cache[0, :, :p].float(), atol=3e-2, rtol=3e-2)
            assert torch.allclose(lf.v_cache[0, :, :p].float(), ls.v_cache[0, :, :p].float(), atol=3e-2, rtol=3e-2)
    n = int(fast.step[0].item())
    assert torch.equal(fast.out_tokens[0, :n].long(), greedy[:n])
