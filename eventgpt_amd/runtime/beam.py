"""Beam search over the module path (``--num_beams > 1`` in the inference CLI;
the reference forwards the flag to HF ``generate`` — inference.py:75-88).

Beams live in the batch dimension of one dynamic KV cache; after every step the
cache rows are re-gathered by the surviving beams' parent indices. Scores are
sums of fp32 log-probabilities with HF's length normalisation
(``score / len ** length_penalty``); a sample finishes when ``num_beams``
hypotheses ended and no live beam can still beat the worst of them.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def _reorder(past, idx):
    return tuple((k.index_select(0, idx), v.index_select(0, idx)) for k, v in past)


@torch.no_grad()
def beam_search(model, inputs_embeds: torch.Tensor, attention_mask: Optional[torch.Tensor],
                num_beams: int, max_new_tokens: int, eos_token_id: int,
                length_penalty: float = 1.0) -> torch.Tensor:
    """Returns ``[B, n_new]`` (right-padded with the pad id) best hypotheses."""
    pad = model.config.pad_token_id or 0
    outs: List[torch.Tensor] = []
    for b in range(inputs_embeds.shape[0]):
        n = int(attention_mask[b].sum()) if attention_mask is not None else inputs_embeds.shape[1]
        outs.append(_beam_one(model, inputs_embeds[b, :n], num_beams, max_new_tokens, eos_token_id,
                              length_penalty))
    longest = max(o.shape[0] for o in outs)
    res = torch.full((len(outs), longest), pad, dtype=torch.long, device=inputs_embeds.device)
    for i, o in enumerate(outs):
        res[i, :o.shape[0]] = o
    return res


def _beam_one(model, embeds: torch.Tensor, K: int, max_new: int, eos: int, lp: float) -> torch.Tensor:
    dev = embeds.device
    out = model.model(inputs_embeds=embeds[None], use_cache=True)
    logp = torch.log_softmax(model.lm_head(out.last_hidden_state[:, -1]).float(), -1)[0]
    V = logp.shape[0]
    K0 = min(K, V)
    scores, tok = logp.topk(K0)
    past = _reorder(out.past_key_values, torch.zeros(K0, dtype=torch.long, device=dev))
    seqs = tok[:, None]
    pos = embeds.shape[0]
    finished: List[Tuple[float, torch.Tensor]] = []

    def finish(score: float, seq: torch.Tensor) -> None:
        finished.append((score / (seq.shape[0] ** lp), seq))
        finished.sort(key=lambda t: -t[0])
        del finished[K:]

    for step in range(1, max_new + 1):
        # retire beams that just emitted EOS
        ended = seqs[:, -1] == eos
        for i in torch.nonzero(ended).flatten().tolist():
            finish(float(scores[i]), seqs[i])
        live = torch.nonzero(~ended).flatten()
        if live.numel() == 0 or step == max_new:
            for i in live.tolist():
                finish(float(scores[i]), seqs[i])
            break
        if len(finished) >= K:
            # upper bound for a live beam: its (negative) score cannot rise, and the
            # largest divisor it can still get is the length cap
            best_live = float(scores[live].max()) / (max_new ** lp if lp > 0 else 1.0)
            if best_live <= finished[-1][0]:
                break
        seqs, scores, past = seqs[live], scores[live], _reorder(past, live)
        n = seqs.shape[0]
        mask = torch.ones(n, pos + 1, dtype=torch.bool, device=dev)
        out = model.model(inputs_embeds=model.model.embed_tokens(seqs[:, -1])[:, None], attention_mask=mask,
                          position_ids=torch.full((n, 1), pos, device=dev, dtype=torch.long),
                          past_key_values=past, use_cache=True)
        pos += 1
        logp = torch.log_softmax(model.lm_head(out.last_hidden_state[:, -1]).float(), -1)
        total = (scores[:, None] + logp).reshape(-1)
        scores, flat = total.topk(min(K, total.numel()))
        parent, tok = flat // V, flat % V
        seqs = torch.cat([seqs[parent], tok[:, None]], dim=1)
        past = _reorder(out.past_key_values, parent)
    return finished[0][1]
