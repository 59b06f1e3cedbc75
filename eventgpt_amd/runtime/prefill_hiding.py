"""User-facing prefill-hiding runtimes built on ``DualModelScheduler``.

* ``PrefillOnlyInference`` — drafter and verifier each prefill and decode their own
  answer concurrently; the verifier's text is the result and the drafter's text is a
  *preview* that is ready much earlier
  (reference: feasible/egpt_prefill_only/prefill_only.py:72-502).
* ``PrefillThenVerifyInference`` — the drafter prefills and keeps drafting until the
  verifier's prefill completes (or ``max_draft_tokens`` / EOS), the drafts are mapped
  into the verifier's vocabulary (``TokenAdapterForSD``), verified in ONE batched
  forward and the verifier continues autoregressively
  (reference: prefill_then_verify.py:50-103, 147-724).
* ``AlignedPrefillInference`` — the same with drafts produced by a hidden-state
  adapter + the verifier's LM head instead of the drafter's own tokens
  (reference: prefill_with_alignment.py).
* analytic helpers ``compute_prefill_benefit`` / ``compute_theoretical_speedup``
  (prefill_only.py:72-97, compute_speedup.py:183-228) and ``evaluate_sd_outputs``
  (evaluate_sd_outputs.py).

Both models run on their serving engines; concurrency is two host threads feeding two
default-priority HIP streams (one GPU) or two devices.
"""
from __future__ import annotations

import dataclasses
import time
from typing import Callable, Dict, List, Optional, Sequence

import torch

from .scheduler import DualModelScheduler, _run_threads
from .spec_decode import AdapterDraftHead, SpecDecodeEngine


def compute_prefill_benefit(egpt_prefill_ms: float, vl_prefill_ms: float, egpt_decode_rate_ms: float = 18.0) -> Dict:
    window = max(0.0, vl_prefill_ms - egpt_prefill_ms)
    return {"overlap_window_ms": window, "free_tokens": int(window / egpt_decode_rate_ms) if egpt_decode_rate_ms > 0 else 0,
            "prefill_speedup": vl_prefill_ms / max(egpt_prefill_ms, 1e-9), "egpt_prefill_ms": egpt_prefill_ms,
            "vl_prefill_ms": vl_prefill_ms}


def compute_theoretical_speedup(egpt_prefill_ms: float, vl_prefill_ms: float, vl_token_time_ms: float = 14.5,
                                max_new_tokens: int = 50, acceptance_rate: float = 0.279,
                                egpt_token_time_ms: float = 10.0, verify_time_ms: float = 50.0) -> Dict:
    overlap = vl_prefill_ms - egpt_prefill_ms
    free = max(int(overlap / egpt_token_time_ms), 0) if egpt_token_time_ms > 0 else 0
    accepted = int(free * acceptance_rate)
    base = vl_prefill_ms + max_new_tokens * vl_token_time_ms
    sd = vl_prefill_ms + verify_time_ms + (max_new_tokens - accepted) * vl_token_time_ms
    return {"egpt_prefill_ms": egpt_prefill_ms, "vl_prefill_ms": vl_prefill_ms, "overlap_ms": overlap,
            "free_drafts": free, "accepted": accepted, "time_saved_ms": accepted * vl_token_time_ms,
            "vl_baseline_ms": base, "sd_time_ms": sd, "prefill_speedup": vl_prefill_ms / max(egpt_prefill_ms, 1e-9),
            "e2e_speedup": base / sd if sd > 0 else 1.0}


@dataclasses.dataclass
class PrefillOnlyResult:
    output_text: str
    output_tokens: List[int]
    preview_text: Optional[str]
    preview_tokens: Optional[List[int]]
    egpt_prefill_time: float
    egpt_decode_time: float
    vl_prefill_time: float
    vl_decode_time: float
    wall_clock_time: float
    free_tokens_count: int
    overlap_window: float
    prefill_speedup: float
    total_speedup_vs_sequential: float

    def as_dict(self) -> Dict:
        return dataclasses.asdict(self)


@dataclasses.dataclass
class VerifyResult:
    output_text: str
    output_tokens: List[int]
    draft_tokens: List[int]
    n_drafted: int
    n_accepted: int
    accept_rate: float
    egpt_prefill_time: float
    vl_prefill_time: float
    prefill_wall_time: float
    decode_time: float
    wall_clock_time: float

    def as_dict(self) -> Dict:
        return dataclasses.asdict(self)


def _decode_text(tokenizer, toks) -> str:
    if tokenizer is None:
        return ""
    return tokenizer.decode(list(toks), skip_special_tokens=True).strip()


class PrefillOnlyInference:
    def __init__(self, drafter_engine, verifier_engine, drafter_tokenizer=None, verifier_tokenizer=None,
                 max_new_tokens: int = 50, return_preview: bool = True):
        self.sched = DualModelScheduler(drafter_engine, verifier_engine)
        self.d_tok, self.v_tok = drafter_tokenizer, verifier_tokenizer
        self.max_new, self.return_preview = max_new_tokens, return_preview

    def _generate(self, engine, stream, lock, embeds, eos) -> Dict:
        with lock, torch.cuda.device(engine.device), torch.cuda.stream(stream):
            stream.synchronize()
            t0 = time.perf_counter()
            engine.eos = eos
            engine.reset()
            engine.prefill(embeds, select="greedy")
            stream.synchronize()
            t1 = time.perf_counter()
            toks = engine.generate_continue(self.max_new, eos)
            stream.synchronize()
            t2 = time.perf_counter()
        return {"tokens": toks.tolist(), "prefill": t1 - t0, "decode": t2 - t1, "done_at": t2}

    @torch.no_grad()
    def generate(self, d_embeds, v_embeds, eos_token_id: int = 2) -> PrefillOnlyResult:
        s = self.sched
        for eng in (s.d, s.v):
            with torch.cuda.device(eng.device):
                eng.ensure_graphs(eos_token_id)
        s._sync_all()
        t0 = time.perf_counter()
        d, v = _run_threads([lambda: self._generate(s.d, s.d_stream, s.d_lock, d_embeds, eos_token_id),
                             lambda: self._generate(s.v, s.v_stream, s.v_lock, v_embeds, eos_token_id)])
        s._sync_all()
        wall = time.perf_counter() - t0
        ben = compute_prefill_benefit(d["prefill"] * 1e3, v["prefill"] * 1e3,
                                      1e3 * d["decode"] / max(len(d["tokens"]), 1))
        seq = d["prefill"] + d["decode"] + v["prefill"] + v["decode"]
        return PrefillOnlyResult(
            _decode_text(self.v_tok, v["tokens"]), v["tokens"],
            _decode_text(self.d_tok, d["tokens"]) if self.return_preview else None,
            d["tokens"] if self.return_preview else None, d["prefill"], d["decode"], v["prefill"], v["decode"],
            wall, ben["free_tokens"], ben["overlap_window_ms"] / 1e3, ben["prefill_speedup"],
            seq / wall if wall > 0 else 1.0)


class PrefillThenVerifyInference:
    def __init__(self, drafter_engine, verifier_engine, token_map: Optional[Callable] = None,
                 verifier_tokenizer=None, max_new_tokens: int = 50, max_draft_tokens: int = 30,
                 draft_head: Optional[AdapterDraftHead] = None):
        self.sched = DualModelScheduler(drafter_engine, verifier_engine, draft_head)
        self.token_map, self.v_tok = token_map, verifier_tokenizer
        self.max_new, self.max_draft = max_new_tokens, max_draft_tokens

    @torch.no_grad()
    def generate(self, d_embeds, v_embeds, eos_token_id: int = 2) -> VerifyResult:
        s = self.sched
        t0 = time.perf_counter()
        pre = s.hidden_prefill(d_embeds, v_embeds, self.max_draft, eos_token_id, self.token_map)
        sd = SpecDecodeEngine(s.v, s.head)
        toks, st = sd.verify_free_tokens(pre["draft"], self.max_new, eos_token_id, drop_first=True)
        wall = time.perf_counter() - t0
        toks = toks.tolist()
        return VerifyResult(_decode_text(self.v_tok, toks), toks, pre["draft"].tolist(), st.total_drafted,
                            st.total_accepted, st.accept_rate, pre["egpt_prefill_ms"] / 1e3,
                            pre["vl_prefill_ms"] / 1e3, pre["prefill_wall_ms"] / 1e3, st.decode_ms / 1e3, wall)


class AlignedPrefillInference(PrefillThenVerifyInference):
    """Drafts come from ``adapter(drafter hidden) @ verifier lm_head``."""

    def __init__(self, drafter_engine, verifier_engine, adapter: torch.nn.Module, verifier_lm_head: torch.Tensor,
                 **kw):
        head = AdapterDraftHead(adapter.to(verifier_engine.device).to(torch.bfloat16), verifier_lm_head)
        super().__init__(drafter_engine, verifier_engine, None, draft_head=head, **kw)


def evaluate_sd_outputs(results: Sequence[Dict], baseline: Sequence[Dict]) -> Dict:
    """Aggregate a run of ``VerifyResult``/baseline dict pairs: losslessness, acceptance,
    mean times and speedup."""
    n = min(len(results), len(baseline))
    if n == 0:
        return {"n": 0}
    # a runner that compared logits at the first difference says so in `identical_to_baseline`
    same = sum(1 for r, b in zip(results, baseline)
               if r.get("identical_to_baseline", r["output_tokens"] == b["tokens"][:len(r["output_tokens"])]))
    sd_ms = sum(r["wall_clock_time"] for r in results[:n]) / n * 1e3
    bl_ms = sum(b["total_ms"] for b in baseline[:n]) / n
    drafted = sum(r["n_drafted"] for r in results[:n])
    return {"n": n, "identical_outputs": same, "lossless": same == n,
            "accept_rate": sum(r["n_accepted"] for r in results[:n]) / drafted if drafted else 0.0,
            "mean_accepted": sum(r["n_accepted"] for r in results[:n]) / n, "sd_ms": sd_ms, "baseline_ms": bl_ms,
            "speedup": bl_ms / sd_ms if sd_ms > 0 else 1.0}
