"""Decode attention over the static cache for every KV split count the engines choose (1, 2, 4, 8, 16),
several streams of different lengths, grouped KV heads, both head sizes, against fp32 softmax attention."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("B,Hq,Hkv", [(1, 8, 8), (4, 8, 2), (5, 4, 4)])
def test_decode_attention_for_every_split_count(B, Hq, Hkv, D):
    from eventgpt_amd.ops import hip
    raw = hip.raw
    dev = torch.device("cuda")
    L = 700
    g = torch.Generator().manual_seed(B * 100 + Hq + D)
    q = torch.randn(B, Hq * D, generator=g).to(dev, torch.bfloat16)
    kc = torch.randn(B, Hkv, L, D, generator=g).to(dev, torch.bfloat16)
    vc = torch.randn(B, Hkv, L, D, generator=g).to(dev, torch.bfloat16)
    pos = torch.tensor([5, 130, 636, 17, 0][:B], dtype=torch.int32, device=dev)     # the newest key's index
    scale = D ** -0.5
    # the oracle runs on the host in double precision: no library kernel, no shared device state
    want = torch.empty(B, Hq, D, dtype=torch.float64)
    qh, kh, vh = q.cpu().double(), kc.cpu().double(), vc.cpu().double()
    for b in range(B):
        n = int(pos[b]) + 1
        k = kh[b, :, :n].repeat_interleave(Hq // Hkv, 0)
        v = vh[b, :, :n].repeat_interleave(Hq // Hkv, 0)
        w = torch.softmax((qh[b].view(Hq, 1, D) * k).sum(-1) * scale, -1)
        want[b] = (w.unsqueeze(-1) * v).sum(1)
    want = want.to(dev, torch.float32)
    for nsplit in (1, 2, 4, 8, 16):
        out = torch.zeros(B, Hq * D, dtype=torch.bfloat16, device=dev)
        part_o = torch.zeros(B * Hq * nsplit * D, dtype=torch.float32, device=dev)
        part_ml = torch.zeros(B * Hq * nsplit * 2, dtype=torch.float32, device=dev)
        raw.decode_attn(q, kc, vc, out, part_o, part_ml, pos, nsplit, scale)
        err = (out.view(B, Hq, D).float() - want).abs().amax(dim=(1, 2)).tolist()
        assert max(err) < 2e-2, (nsplit, err)


def test_engines_choose_splits_by_block_count():
    from eventgpt_amd.runtime.engine import GenerationEngine
    ks = GenerationEngine.kv_splits
    assert [ks(b, 32) for b in (1, 2, 3, 4, 8, 16, 32)] == [16, 16, 4, 4, 1, 1, 1]
    assert [ks(32, h) for h in (4, 8, 16)] == [4, 1, 1]          # tensor-parallel ranks hold fewer heads
    assert ks(4, 4) == 16
