"""Speculative decoding runtime: a fast drafter proposes, a slow verifier checks.

Pieces (reference counterparts in brackets):

* ``NaiveSpeculativeDecoder`` — token-level greedy SD between two models that
  share a vocabulary, with KV caches on both sides
  [feasible/naive_speculative_decoding.py:7-60, which re-runs the whole sequence
  every step].
* ``AdapterDraftHead`` — drafter hidden states -> hidden adapter -> verifier
  ``lm_head`` -> argmax, with the GEMM's epilogue producing the argmax so logits
  are never materialised [benchmark_e2e_wallclock.py:996-1001].
* ``SpecDecodeEngine`` — the cross-modal loop: adapter-drafted SD with a fixed
  gamma, the gamma=1 feedback modes (B1 / L5F consume verifier hidden states),
  and prefill hiding (verify every token the drafter produced while the
  verifier was still prefilling, in one batch) [run_sd_decode :860-1032,
  vl_verify_batch :569-637, sequential_egpt_vl_prefill :722-853].

Every verify step is ``GenerationEngine.verify``: one forward over gamma+1 rows
and a fused accept kernel; acceptance, the bonus token, EOS and KV rollback
happen on the device.  Output is lossless: the emitted sequence equals the
verifier's own greedy decode.

Device-resident rounds (``SpecDecodeEngine.run(..., device_loop=True)``, the default on
a GPU when the adapter has a kernel forward): one round -- gather the drafter rows that
belong to the current position (the position is a device counter), adapter forward on
the in-tree kernels, lm_head GEMM with argmax epilogue, verify forward, accept kernel --
is ONE captured hipGraph.  The host replays it ``poll_every`` times back to back and
then reads six integers; rounds that land after the sequence finished change nothing
(the accept kernel freezes once ``done`` is set or the token budget is out).  The
reference's loop synchronises several times per round (``.tolist()`` of the argmax,
Python prefix match, KV slicing: benchmark_e2e_wallclock.py:569-637, 860-1032).
"""
from __future__ import annotations

import dataclasses
import time
from typing import Callable, Dict, List, Optional, Sequence

import torch

from .. import ops
from .engine import persistent_state


@dataclasses.dataclass
class SDStats:
    total_drafted: int = 0
    total_accepted: int = 0
    iterations: int = 0
    fallback_steps: int = 0
    tokens: int = 0
    decode_ms: float = 0.0

    @property
    def accept_rate(self) -> float:
        return self.total_accepted / self.total_drafted if self.total_drafted else 0.0

    @property
    def tokens_per_iter(self) -> float:
        return self.tokens / self.iterations if self.iterations else 0.0

    def as_dict(self) -> Dict[str, float]:
        return {"total_drafted": self.total_drafted, "total_accepted": self.total_accepted,
                "iterations": self.iterations, "fallback_steps": self.fallback_steps,
                "accept_rate": self.accept_rate, "tokens_per_iter": self.tokens_per_iter,
                "tokens": self.tokens, "decode_ms": self.decode_ms}


NEAR_TIE_LOGIT_GAP = 0.08   # below this, bf16 kernels of different shapes may order two tokens either way


@torch.no_grad()
def matches_greedy(verifier_engine, prompt_embeds: torch.Tensor, got: torch.Tensor, ref: torch.Tensor,
                   tol: float = NEAR_TIE_LOGIT_GAP) -> Dict[str, object]:
    """Is ``got`` the verifier's greedy answer ``ref``?  Exactly -- or up to a near-tie: a verify pass
    scores several rows with other kernels than the one-row decode step, so where the verifier's own
    teacher-forced logits hold two candidates closer than bf16 rounding the choice may fall either way
    (and everything after it legitimately differs).  Only the first difference is examined.
    Scoring resets ``verifier_engine``."""
    got, ref = torch.as_tensor(got), torch.as_tensor(ref)
    n = min(got.shape[0], ref.shape[0])
    got, ref = got[:n].to(ref.device), ref[:n]
    if torch.equal(got, ref):
        return {"lossless": True, "exact": True, "first_mismatch": None, "mismatch_logit_gap": None}
    i = int((got != ref).nonzero()[0])
    row = verifier_engine.score(prompt_embeds, ref[:i].to(verifier_engine.device))[-1]
    a, b = int(ref[i]), int(got[i])
    gap = abs(float(row[a]) - float(row[b]))
    top2 = row.topk(2).indices.tolist()
    return {"lossless": bool(gap < tol and a in top2 and b in top2), "exact": False, "first_mismatch": i,
            "mismatch_logit_gap": gap}


def _emitted(engine) -> torch.Tensor:
    n = int(engine.step[0].item())
    return engine.out_tokens[0, :n].to(torch.long)


class AdapterDraftHead:
    """``adapter(hidden) @ lm_head^T -> argmax`` as draft tokens (device int32)."""

    def __init__(self, adapter: torch.nn.Module, lm_head_weight: torch.Tensor):
        self.adapter = adapter.eval()
        self.dtype = next(adapter.parameters()).dtype
        self.w = lm_head_weight
        self._w_bf16 = None
        self.kernels = None
        if lm_head_weight.is_cuda and lm_head_weight.shape[1] % 64 == 0:
            from ..align.adapter_engine import AdapterKernels
            if AdapterKernels.supports(adapter):
                self.kernels = AdapterKernels(adapter)

    @torch.no_grad()
    def draft_device(self, hidden: torch.Tensor, verifier_hidden: Optional[torch.Tensor],
                     out: torch.Tensor) -> None:
        """Draft tokens of ``hidden [G,H]`` into ``out`` (device int32 ``[G]``) on in-tree kernels
        only: no allocation through a library workspace, no host sync -- safe to record into
        the graph of a draft/verify round."""
        from ..ops import hip
        if self._w_bf16 is None:
            self._w_bf16 = self.w.to(torch.bfloat16).contiguous()
        aligned = self.kernels(hidden, verifier_hidden)
        pv, pi = hip.raw.linear_argmax_partials(aligned, self._w_bf16, None)
        hip.raw.argmax_from_partials(pv, pi, out, int(self._w_bf16.shape[0]))

    @torch.no_grad()
    def __call__(self, hidden: torch.Tensor, verifier_hidden: Optional[torch.Tensor] = None) -> torch.Tensor:
        h = hidden.to(self.dtype)
        if h.ndim == 2:
            h = h[None]
        if getattr(self.adapter, "takes_target_input", False):
            aligned = self.adapter(h, verifier_hidden.to(self.dtype).view(1, -1, h.shape[-1]))
        else:
            aligned = self.adapter(h)
        aligned = aligned[0]
        if aligned.is_cuda and aligned.shape[-1] % 64 == 0:
            from ..ops import hip
            if self._w_bf16 is None:
                self._w_bf16 = self.w.to(torch.bfloat16).contiguous()
            pv, pi = hip.raw.linear_argmax_partials(aligned.to(torch.bfloat16).contiguous(),
                                                    self._w_bf16, None)
            best = pv.argmax(dim=1, keepdim=True)
            return pi.gather(1, best)[:, 0].to(torch.int32)
        return (aligned.float() @ self.w.float().t()).argmax(-1).to(torch.int32)


class NaiveSpeculativeDecoder:
    """Greedy token-level SD.  ``draft`` and ``target`` are ``GenerationEngine``s over
    the same vocabulary; both keep their KV caches and roll back by position
    [feasible/naive_speculative_decoding.py:7-60 re-runs both models over the whole
    sequence for every draft token and every verification].

    Device-resident rounds (the default): one round -- point the drafter at the target's
    position and stage its two input rows (``naive_sd_sync``), a two-row drafter pass that
    yields the first draft, gamma - 1 drafter decode steps, the target's verify pass over
    gamma + 1 rows and the fused accept kernel -- is ONE captured hipGraph.  The two-row
    start replaces the conditional catch-up pass of a host-driven loop (after a round whose
    drafts were all accepted the drafter has not ingested the last of them): re-writing the
    slot before the current token costs the second row of a weight-streaming pass, and
    needs no decision the host would have to read a counter for.  The host replays the
    graph ``poll_every`` times and then reads six integers.
    """

    def __init__(self, draft_engine, target_engine, tokenizer=None):
        self.draft, self.target, self.tokenizer = draft_engine, target_engine, tokenizer
        self._graphs = {}
        self._rows = None

    # ------------------------------------------------------ device-resident rounds
    @persistent_state
    def _state(self) -> None:
        d, t = self.draft, self.target
        if self._rows is None:
            dev = d.device
            self._rows = torch.zeros(2, d.H, dtype=torch.bfloat16, device=dev)
            self._prompt_last = torch.zeros(d.H, dtype=torch.bfloat16, device=dev)
            self._pos0 = torch.zeros(1, dtype=torch.int32, device=dev)
        if t.sd_limit is None:
            t.sd_limit = torch.zeros(1, dtype=torch.int32, device=t.device)
            t.sd_stats = torch.zeros(4, dtype=torch.int32, device=t.device)
            t._verify_graphs.clear()         # earlier verify graphs were captured without them

    def _round_body(self, gamma: int, offset: int) -> None:
        d, t = self.draft, self.target
        raw = t.raw
        raw.naive_sd_sync(t.pos, t.step, t.cur_token, t.out_tokens, offset, d.embed, self._prompt_last,
                          self._rows, self._pos0, d.pos, d.step, d.done, d.l_max)
        x = d._forward_rows(self._rows, 0, self._pos0)       # two rows: positions pos - 1, pos
        d.x[0].copy_(x[0, -1])
        d._logits_and_commit(greedy=True)                    # draft 1; the commit stages its embedding
        for _ in range(gamma - 1):
            d._layers_step()
            d._logits_and_commit(greedy=True)
        t._verify_body(d.out_tokens[0, :gamma], gamma)

    def _round_graph(self, gamma: int, offset: int):
        d, t = self.draft, self.target
        key = (gamma, offset, t.eos)
        g = self._graphs.get(key)
        if g is None:
            keep_t = (t.out_tokens, t.sd_result, t.sd_stats, d.step, d.pos, d.done, d.cur_token, d.x, d.out_tokens)
            keep = [x.clone() for x in keep_t]
            try:
                g = t._capture(lambda: self._round_body(gamma, offset))
            finally:
                for x, k in zip(keep_t, keep):
                    x.copy_(k)
            self._graphs[key] = g
        return g

    @torch.no_grad()
    def generate(self, draft_embeds: torch.Tensor, target_embeds: Optional[torch.Tensor] = None,
                 max_new_tokens: int = 50, num_draft_tokens: int = 5, eos_token_id: int = -1,
                 device_loop: bool = True, poll_every: int = 4):
        """Returns ``(tokens, acceptance_history)``; ``tokens`` equals the target's own
        greedy continuation of its prompt.  The history has one entry per round when the
        host drives the rounds and one per ``poll_every`` rounds (their mean) otherwise;
        ``self.stats`` holds the totals of the last call."""
        d, t = self.draft, self.target
        target_embeds = draft_embeds if target_embeds is None else target_embeds
        t.eos = eos_token_id
        d.eos = -1                       # the drafter proposes; only the target ends a sequence
        d.reset(); t.reset()
        gamma = num_draft_tokens
        if not 1 <= gamma <= 63:
            raise ValueError(f"1..63 draft tokens per round, got {gamma}")
        if d.embed.shape[0] != t.embed.shape[0]:
            raise ValueError("token-level speculative decoding needs one vocabulary on both sides")
        # every verify appends up to gamma+1 cache rows on both sides: keep the budget inside
        room = min(t.l_max - target_embeds.shape[0], d.l_max - draft_embeds.shape[0]) - (gamma + 1)
        if room < 1:
            raise ValueError(f"prompt of {target_embeds.shape[0]} / {draft_embeds.shape[0]} tokens leaves no room "
                             f"for speculative rounds of {gamma} in caches of {t.l_max} / {d.l_max}")
        max_new_tokens = min(max_new_tokens, room, t.max_out, d.max_out)
        t.prefill(target_embeds, select="greedy")          # first token comes from the target
        d.prefill(draft_embeds, select=None)
        offset = draft_embeds.shape[0] - target_embeds.shape[0]
        self.stats = SDStats()
        if device_loop:
            return self._generate_device(draft_embeds, max_new_tokens, gamma, offset, poll_every)
        history: List[float] = []
        # one read per round: [n_accepted, next, emitted, step, done, pos] of the last verify
        info = [0, 0, 0, 1, int(t.done[0].item()), target_embeds.shape[0]]
        while info[3] < max_new_tokens and not info[4]:
            # drafter continues from the target's current token at its own position
            base = info[5] + offset
            d.pos.fill_(base)
            d.step.zero_()
            d.x[0].copy_(d.embed[t.cur_token[0].long()])
            for _ in range(gamma):
                d.decode_step(True)
            draft = d.out_tokens[0, :gamma].clone()
            t.verify(draft, gamma)
            info = t.sd_result[:6].tolist()
            n_acc = info[0]
            history.append(n_acc / gamma)
            self.stats.total_drafted += gamma
            self.stats.total_accepted += n_acc
            self.stats.iterations += 1
            if n_acc == gamma and not info[4]:
                # every draft token was accepted: the drafter has not yet ingested the
                # last one (its cache holds pos < base+gamma)
                d.pos.fill_(base + gamma)
                d.x[0].copy_(d.embed[draft[gamma - 1].long()])
                d._layers_step()
        toks = _emitted(t)[:max_new_tokens]
        self.stats.tokens = int(toks.shape[0])
        return toks, history

    def _generate_device(self, draft_embeds, max_new_tokens: int, gamma: int, offset: int, poll_every: int):
        d, t = self.draft, self.target
        self._state()
        self._prompt_last.copy_(draft_embeds[-1].to(torch.bfloat16))
        t.sd_result.zero_()
        t.sd_limit.fill_(max_new_tokens)
        t.sd_stats.zero_()
        graph = self._round_graph(gamma, offset) if t.use_graph else None
        history: List[float] = []
        step, done = int(t.step[0].item()), int(t.done[0].item())
        seen = [0, 0]
        while step < max_new_tokens and not done:
            for _ in range(min(poll_every, max_new_tokens - step)):    # a round emits at least one token
                if graph is not None:
                    graph.replay()
                else:
                    self._round_body(gamma, offset)
            info = torch.cat([t.sd_result[:6], t.sd_stats[:3]]).tolist()   # the only host read of these rounds
            step, done = info[3], info[4]
            drafted, accepted = info[6] - seen[0], info[7] - seen[1]
            seen = [info[6], info[7]]
            if drafted:
                history.append(accepted / drafted)
        st = t.sd_stats.tolist()
        self.stats.total_drafted, self.stats.total_accepted, self.stats.iterations = st[0], st[1], st[2]
        t.sd_limit.fill_(1 << 30)
        t._pos_hi = int(t.pos[0].item())
        d._pos_hi = int(d.pos[0].item())
        toks = _emitted(t)[:max_new_tokens]
        self.stats.tokens = int(toks.shape[0])
        return toks, history


class SpecDecodeEngine:
    """Adapter-drafted speculative decoding against a ``GenerationEngine`` verifier."""

    def __init__(self, verifier_engine, draft_head: Optional[AdapterDraftHead] = None,
                 gamma: int = 5):
        self.v = verifier_engine
        self.head = draft_head
        self.gamma = gamma
        self._round_graphs = {}        # (mode, G, eos) -> captured draft/verify round
        self._dh = None                # static copy of the drafter states the rounds gather from
        self._nh = None

    # ------------------------------------------------------ device-resident rounds
    def device_loop_available(self) -> bool:
        v = self.v
        return bool(self.head is not None and self.head.kernels is not None and v.device.type == "cuda"
                    and v.B == 1 and v.H % 64 == 0)

    def _round_state(self, G: int, n_rows: int):
        v, dev = self.v, self.v.device
        if self._dh is None or self._dh.shape[0] < max(n_rows, 1):
            self._dh = torch.zeros(max(n_rows, v.max_out, 64), v.H, dtype=torch.bfloat16, device=dev)
            self._nh = torch.ones(1, dtype=torch.int32, device=dev)
            self._round_graphs.clear()       # graphs bake the buffer addresses
        if getattr(self, "_vh", None) is None:
            self._vh = torch.zeros(64, v.H, dtype=torch.bfloat16, device=dev)
            self._draft = torch.zeros(63, dtype=torch.int32, device=dev)
        if v.sd_limit is None:
            v.sd_limit = torch.zeros(1, dtype=torch.int32, device=dev)
            v.sd_stats = torch.zeros(4, dtype=torch.int32, device=dev)
            v._verify_graphs.clear()         # earlier verify graphs were captured without them

    def _round_body(self, mode: str, G: int) -> None:
        from ..ops import hip
        raw, v = hip.raw, self.v
        draft = self._draft[:G]
        if mode == "adapter":
            rows = raw.gather_rows_dev(self._dh, v.step, -1, self._nh, G)
            self.head.draft_device(rows, None, draft)
        else:
            vh = raw.gather_rows_dev(self._vh, v.sd_result, 0, None, 1)   # the row that predicts the next token
            if mode == "vlm_only":
                self.head.draft_device(vh, None, draft)
            else:
                rows = raw.gather_rows_dev(self._dh, v.step, -1, self._nh, 1)
                self.head.draft_device(rows, vh, draft)
        hn = v._verify_body(draft, G)
        if mode != "adapter":
            self._vh[:G + 1].copy_(hn)

    def _round_graph(self, mode: str, G: int):
        v = self.v
        key = (mode, G, v.eos)
        g = self._round_graphs.get(key)
        if g is None:
            keep_t = (v.out_tokens, v.sd_result, v.sd_stats, self._vh)
            keep = [t.clone() for t in keep_t]
            try:
                g = v._capture(lambda: self._round_body(mode, G))
            finally:
                for t, k in zip(keep_t, keep):
                    t.copy_(k)
            self._round_graphs[key] = g
        return g

    @torch.no_grad()
    def _run_device(self, drafter_hidden, max_new_tokens, mode, verifier_prefill_hidden, poll_every, stats):
        v = self.v
        need_vh = mode in ("vlm_only", "fused")
        G = 1 if need_vh else self.gamma
        n_h = 0 if drafter_hidden is None else int(drafter_hidden.shape[0])
        self._round_state(G, n_h)
        if n_h:
            self._dh[:n_h].copy_(drafter_hidden.to(torch.bfloat16))
        self._nh.fill_(max(n_h, 1))
        start = v.sd_result.new_tensor([0, 0, 0, 0, 0, 0, 0, 0])
        v.sd_result.copy_(start)
        if need_vh:
            self._vh[0].copy_(verifier_prefill_hidden.reshape(-1).to(torch.bfloat16))
        step0, pos0 = int(v.step[0].item()), int(v.pos[0].item())
        # every round (also one that lands after the end) writes G+1 cache rows at the position
        budget = min(max_new_tokens, step0 + (v.l_max - pos0) - (G + 1))
        if budget <= step0:
            raise ValueError(f"no room for speculative rounds of {G} at position {pos0} in a cache of {v.l_max}")
        v.sd_limit.fill_(budget)
        v.sd_stats.zero_()
        graph = self._round_graph(mode, G) if v.use_graph else None
        step, done = step0, int(v.done[0].item())
        while step < budget and not done:
            if mode == "adapter" and step - 1 >= n_h:       # drafter states used up: plain decoding
                for _ in range(min(poll_every, budget - step)):
                    v.decode_step(True)
                    stats.fallback_steps += 1
                    stats.iterations += 1
                step, done = int(v.step[0].item()), int(v.done[0].item())
                continue
            for _ in range(min(poll_every, budget - step)):  # a round emits at least one token
                if graph is not None:
                    graph.replay()
                else:
                    self._round_body(mode, G)
            info = v.sd_result[:6].tolist()                  # the only host read of these rounds
            step, done = info[3], info[4]
        st = v.sd_stats.tolist()
        stats.total_drafted += st[0]
        stats.total_accepted += st[1]
        stats.iterations += st[2]
        v.sd_limit.fill_(1 << 30)
        v._pos_hi = int(v.pos[0].item())

    # ---------------------------------------------------------------- drafting
    def _verify(self, draft: torch.Tensor, stats: SDStats, want_hidden: bool):
        g = int(draft.shape[0])
        hidden = self.v.verify(draft, g, return_hidden=want_hidden)
        n_acc = int(self.v.sd_result[0].item())
        stats.total_drafted += g
        stats.total_accepted += n_acc
        stats.iterations += 1
        return n_acc, hidden

    def _ar_step(self, stats: SDStats) -> None:
        self.v.decode_step(True)
        stats.fallback_steps += 1
        stats.iterations += 1

    @torch.no_grad()
    def run(self, drafter_hidden: Optional[torch.Tensor], max_new_tokens: int = 50,
            eos_token_id: int = 2, mode: str = "adapter",
            verifier_prefill_hidden: Optional[torch.Tensor] = None,
            device_loop: Optional[bool] = None, poll_every: int = 4) -> tuple:
        """SD decode after the verifier has prefilled and committed its first token.

        ``drafter_hidden [N, H]``: the drafter's per-step last-layer hidden states
        for its own greedy answer (position i drafts verifier token i+1).
        ``mode``: ``"adapter"`` (L1-L5, one-shot gamma drafts), ``"vlm_only"`` (B1,
        gamma=1 from the verifier's own last hidden state) or ``"fused"`` (L5F,
        gamma=1 from drafter + verifier hidden).  Returns ``(tokens, SDStats)``.

        ``device_loop`` (default: on when the adapter has a kernel forward): rounds are
        captured graphs replayed ``poll_every`` at a time with one host read per batch of
        rounds; off, every round is driven and read back by the host.
        """
        v = self.v
        v.eos = eos_token_id
        stats = SDStats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        need_vh = mode in ("vlm_only", "fused")
        if device_loop is None:
            device_loop = self.device_loop_available() and (not need_vh or verifier_prefill_hidden is not None) \
                and (mode != "fused" or drafter_hidden is not None)
        if device_loop:
            self._run_device(drafter_hidden, max_new_tokens, mode, verifier_prefill_hidden, poll_every, stats)
            torch.cuda.synchronize()
            stats.decode_ms = (time.perf_counter() - t0) * 1e3
            toks = _emitted(v)[:max_new_tokens]
            stats.tokens = int(toks.shape[0])
            return toks, stats
        # cache room: a verify appends up to gamma+1 rows
        room = v.l_max - int(v.pos[0].item()) - ((1 if need_vh else self.gamma) + 1)
        max_new_tokens = min(max_new_tokens, int(v.step[0].item()) + max(room, 0))
        last_vh = verifier_prefill_hidden  # [1, H] verifier hidden that predicts the next token
        n_h = 0 if drafter_hidden is None else drafter_hidden.shape[0]
        while True:
            n_out = int(v.step[0].item())
            if n_out >= max_new_tokens or int(v.done[0].item()):
                break
            start = n_out - 1  # index of the drafter state aligned with the current token
            if need_vh:
                if last_vh is None or (mode == "fused" and start >= n_h):
                    self._ar_step(stats)
                    last_vh = None
                    continue
                if mode == "vlm_only":
                    draft = self.head(last_vh.view(1, -1))
                else:
                    draft = self.head(drafter_hidden[start:start + 1], last_vh.view(1, -1))
                n_acc, vh = self._verify(draft[:1], stats, True)
                last_vh = vh[n_acc:n_acc + 1]
            else:
                end = min(start + self.gamma, n_h)
                if end - start <= 0:
                    self._ar_step(stats)
                    continue
                draft = self.head(drafter_hidden[start:end])
                self._verify(draft, stats, False)
        torch.cuda.synchronize()
        stats.decode_ms = (time.perf_counter() - t0) * 1e3
        toks = _emitted(v)[:max_new_tokens]
        stats.tokens = int(toks.shape[0])
        return toks, stats

    @torch.no_grad()
    def verify_free_tokens(self, draft_tokens: torch.Tensor, max_new_tokens: int = 50,
                           eos_token_id: int = 2, drop_first: bool = True,
                           continue_ar: bool = True, want_hidden: bool = False) -> tuple:
        """Prefill hiding: the drafter produced ``draft_tokens`` while the verifier was
        prefilling.  They are checked in ONE batched verify, then the verifier
        continues autoregressively.  ``drop_first``: the first draft token
        duplicates the verifier's own first token and is skipped
        (benchmark_e2e_wallclock.py:1766-1773).  With ``continue_ar=False`` only the
        batched verify runs (a decode-phase drafter takes over afterwards);
        ``want_hidden`` keeps the verifier hidden row that predicts the next token
        in ``self.next_hidden`` for the gamma=1 feedback modes."""
        v = self.v
        self.next_hidden = None
        v.eos = eos_token_id
        stats = SDStats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d = draft_tokens[1:] if drop_first else draft_tokens
        d = d[: min(d.shape[0], 63, max_new_tokens - 1)]
        if d.shape[0] > 0:
            n_acc, vh = self._verify(d.to(torch.int32), stats, want_hidden)
            if want_hidden:
                self.next_hidden = vh[n_acc:n_acc + 1].clone()
        while continue_ar and int(v.step[0].item()) < max_new_tokens and not int(v.done[0].item()):
            remaining = max_new_tokens - int(v.step[0].item())
            for _ in range(min(v.poll_every, remaining)):
                v.decode_step(True)
                stats.fallback_steps += 1
        torch.cuda.synchronize()
        stats.decode_ms = (time.perf_counter() - t0) * 1e3
        toks = _emitted(v)[:max_new_tokens]
        if eos_token_id >= 0:
            hit = (toks == eos_token_id).nonzero()
            if hit.numel():
                toks = toks[: int(hit[0]) + 1]
        stats.tokens = int(toks.shape[0])
        return toks, stats


def collect_decode_hidden(engine, embeds: torch.Tensor, max_new_tokens: int,
                          eos_token_id: int = -1) -> Dict[str, torch.Tensor]:
    """Greedy decode on ``engine`` keeping, per generated token, the final-norm hidden
    state that produced it — the drafter-side input of the hidden adapters
    [egpt_decode_collect_hidden, benchmark_e2e_wallclock.py:417-472].  Runs eagerly
    (one small copy per step between graph replays)."""
    engine.eos = eos_token_id
    engine.reset()
    h = engine.prefill(embeds, select="greedy", return_hidden=True)
    hidden = [ops.rmsnorm(h[-1:], engine.final_norm, engine.eps)]
    n = 1
    while n < max_new_tokens and not int(engine.done[0].item()):
        engine._layers_step()
        hidden.append(ops.rmsnorm(engine.x[:1].clone(), engine.final_norm, engine.eps))
        engine._logits_and_commit(greedy=True)
        n += 1
    return {"tokens": _emitted(engine)[:n], "hidden": torch.cat(hidden)[:n]}
