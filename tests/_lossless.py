"""Shared check of speculative-decoding tests: output == the verifier's greedy output up to bf16 near-ties."""
import torch


def assert_lossless(v_eng, emb, got, want, tol=0.08):
    """Speculative output == the verifier's own greedy output.  The verify pass scores several rows
    with other kernels than the one-row decode step, so two candidates whose logits are closer
    than bf16 rounding can swap; the first difference (if any) must be such a near-tie."""
    n = min(len(got), len(want))
    got, want = got[:n], want[:n]
    if torch.equal(got, want):
        return
    i = int((got != want).nonzero()[0])
    toks = want[:i].long()
    ctx = torch.cat([emb, v_eng.embed[toks]]) if i else emb
    v_eng.reset()
    v_eng.prefill(ctx, select="logits")
    row = v_eng.logits[0].float()
    gap = abs(float(row[int(want[i])]) - float(row[int(got[i])]))
    top2 = row.topk(2).indices.tolist()
    assert gap < tol and int(want[i]) in top2 and int(got[i]) in top2, (i, gap, got.tolist(), want.tolist())
