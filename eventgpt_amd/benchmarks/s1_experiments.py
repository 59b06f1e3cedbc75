"""Stage-1 ("S1") speculative-decoding experiments between the drafter and the verifier.

* ``SharedDecoderPipeline`` — align the drafter's *visual* features to the verifier's
  projected visual space with a small residual MLP, inject them at ``<video>`` and let
  the verifier's decoder generate; compares original / raw-injected / adapted-injected
  outputs (reference: feasible/benchmark_inference/shared_decoder_speculative_S1.py:70-533).
* ``extract_token_pairs`` — greedy token ids (and optional logits/hidden) of both
  models per (scene, question), in the JSON shape the token-alignment trainers read
  (reference: feasible/token_alignment/extract_tokens_train.py, extract_features.py:52-519).
* ``benchmark_alpha`` — acceptance-rate sweeps: raw token match, match after a
  ``TokenAdapter``, consecutive-prefix statistics per gamma, theoretical speedups
  (reference: feasible/benchmark_inference/benchmark_alpha.py, benchmark_alignment_S1.py,
  benchmark_with_adapter_S1.py, speculative_decoding_S1.py).

Both models are this package's; all loops work on CPU with tiny configs and on the GPU
with the 7B geometries.
"""
from __future__ import annotations

import dataclasses
import json
import os
import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..data.events import load_image
from ..models.video_llava import VIDEO_PROMPT, load_video_frames
from ..pipeline.dataset import InstructionDataset
from .stages import calculate_acceptance_rate, calculate_speculative_speedup, compute_aligned_acceptance_rate


@dataclasses.dataclass
class AdapterConfig:
    input_dim: int = 4096
    output_dim: int = 4096
    hidden_dim: int = 2048
    num_layers: int = 2
    dropout: float = 0.1


class FeatureAlignmentAdapter(nn.Module):
    """MLP (+ identity residual when the widths match) between projected visual spaces."""

    def __init__(self, config: AdapterConfig):
        super().__init__()
        self.config = config
        mods, d = [], config.input_dim
        for i in range(config.num_layers):
            last = i == config.num_layers - 1
            o = config.output_dim if last else config.hidden_dim
            mods.append(nn.Linear(d, o))
            if not last:
                mods += [nn.LayerNorm(o), nn.GELU(), nn.Dropout(config.dropout)]
            d = o
        self.adapter = nn.Sequential(*mods)
        self.use_residual = config.input_dim == config.output_dim

    def forward(self, x):
        y = self.adapter(x)
        return y + x if self.use_residual else y


def _resample_tokens(f: torch.Tensor, target_len: int) -> torch.Tensor:
    """Linear interpolation along the token axis: ``[S, C] -> [target_len, C]``."""
    if f.shape[0] == target_len:
        return f
    return F.interpolate(f.t()[None].float(), size=target_len, mode="linear", align_corners=False)[0].t().to(f.dtype)


class SharedDecoderPipeline:
    def __init__(self, egpt_model, vl_model, vl_processor, target_len: Optional[int] = None):
        self.egpt, self.vl, self.proc = egpt_model, vl_model, vl_processor
        self.target_len = target_len or vl_model.config.tokens_per_frame
        self.adapter: Optional[FeatureAlignmentAdapter] = None

    @torch.no_grad()
    def extract_eventgpt_features(self, event_image_paths: Sequence[str]) -> torch.Tensor:
        """Projected (+adapted) drafter features, mean over frames, resampled to ``target_len`` rows."""
        m = self.egpt
        proc = m.get_visual_tower().event_processor
        px = torch.stack([proc(np.asarray(load_image(p)), return_tensors="pt")["pixel_values"][0]
                          for p in event_image_paths]).to(m.device, m.dtype)
        f = m.visval_encode(px)
        if hasattr(m.get_model(), "feature_adaptor"):
            f = m.get_model().feature_adaptor(f)
        return _resample_tokens(f.mean(0), self.target_len).float()

    @torch.no_grad()
    def extract_videollava_features(self, frames: Sequence[np.ndarray]) -> torch.Tensor:
        m = self.vl
        px = self.proc.frames(frames).to(m.device, m.dtype)
        f = m.get_video_features(px)[0].view(px.shape[1], -1, m.config.text_config.hidden_size)
        return _resample_tokens(f.mean(0), self.target_len).float()

    def train_adapter(self, eventgpt_features: torch.Tensor, videollava_features: torch.Tensor, num_epochs: int = 10,
                      lr: float = 1e-3, batch_size: int = 16, hidden_dim: int = 512, seed: int = 0) -> List[float]:
        """``[N, S, C]`` pairs, trained row-wise with MSE + 0.5 (1 - cos)."""
        dev = self.vl.device
        torch.manual_seed(seed)
        self.adapter = FeatureAlignmentAdapter(AdapterConfig(eventgpt_features.shape[-1], videollava_features.shape[-1],
                                                             hidden_dim, 2)).to(dev)
        opt = torch.optim.AdamW(self.adapter.parameters(), lr=lr, weight_decay=1e-4)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, num_epochs)
        src = eventgpt_features.reshape(-1, eventgpt_features.shape[-1]).to(dev).float()
        tgt = videollava_features.reshape(-1, videollava_features.shape[-1]).to(dev).float()
        losses = []
        for _ in range(num_epochs):
            self.adapter.train()
            perm = torch.randperm(src.shape[0], device=dev)
            tot, n = 0.0, 0
            for i in range(0, src.shape[0], batch_size):
                idx = perm[i:i + batch_size]
                out = self.adapter(src[idx])
                loss = F.mse_loss(out, tgt[idx]) + 0.5 * (1 - F.cosine_similarity(out, tgt[idx], dim=-1).mean())
                opt.zero_grad(set_to_none=True)
                loss.backward()
                torch.nn.utils.clip_grad_norm_(self.adapter.parameters(), 1.0)
                opt.step()
                tot, n = tot + float(loss.detach()), n + 1
            sched.step()
            losses.append(tot / max(n, 1))
        self.adapter.eval()
        return losses

    @torch.no_grad()
    def generate_with_shared_decoder(self, visual_features: torch.Tensor, query: str, max_new_tokens: int = 50,
                                     use_adapter: bool = False):
        """``(text, token ids, seconds)`` from the verifier's decoder with ``visual_features
        [S, C]`` spliced at the ``<video>`` placeholder."""
        m = self.vl
        f = visual_features.to(m.device)
        if use_adapter and self.adapter is not None:
            f = self.adapter(f.float())
        ids = self.proc.prompt_ids(query).to(m.device)
        emb = m.merge_video_features(ids, f.to(m.dtype)[None], None)
        sync = torch.cuda.synchronize if emb.is_cuda else (lambda: None)
        sync()
        t0 = time.perf_counter()
        eos = m.config.text_config.eos_token_id
        if emb.is_cuda and emb.dtype == torch.bfloat16:
            toks = m.engine().generate(emb[0], max_new_tokens, eos_token_id=eos).tolist()
        else:
            lm = m.language_model
            out = lm.model(inputs_embeds=emb, use_cache=True)
            past, last, pos, toks = out.past_key_values, out.last_hidden_state[:, -1], emb.shape[1], []
            for _ in range(max_new_tokens):
                nxt = lm.lm_head(last).float().argmax(-1)
                toks.append(int(nxt))
                if toks[-1] == eos:
                    break
                out = lm.model(inputs_embeds=lm.model.embed_tokens(nxt)[:, None],
                               position_ids=torch.full((1, 1), pos, device=emb.device), past_key_values=past,
                               use_cache=True)
                past, last, pos = out.past_key_values, out.last_hidden_state[:, -1], pos + 1
        sync()
        return self.proc.tokenizer.decode(toks, skip_special_tokens=True).strip(), toks, time.perf_counter() - t0

    def run(self, ds: InstructionDataset, query: str, max_samples: int = 10, max_new_tokens: int = 20,
            num_epochs: int = 10, event_image_key: str = "event_image") -> Dict:
        """Extract paired features, train the adapter on them, then compare the verifier's
        outputs on its own features vs raw vs adapted drafter features."""
        ef, vf, keep = [], [], []
        for s in ds.samples[:max_samples]:
            paths = ds.event_image_paths(s, event_image_key)
            src = ds.video_source(s)
            frames = load_video_frames(src, self.vl.config.num_frames) if src else None
            if not paths or not frames:
                continue
            ef.append(self.extract_eventgpt_features(paths).cpu())
            vf.append(self.extract_videollava_features(frames).cpu())
            keep.append(s)
        if not ef:
            return {"n": 0}
        E, V = torch.stack(ef), torch.stack(vf)
        cos_before = float(F.cosine_similarity(E, V, dim=-1).mean()) if E.shape == V.shape else float("nan")
        losses = self.train_adapter(E, V, num_epochs=num_epochs)
        with torch.no_grad():
            cos_after = float(F.cosine_similarity(self.adapter(E.to(self.vl.device)).cpu(), V, dim=-1).mean())
        rows = []
        for e, v in zip(E, V):
            _, ref, t_ref = self.generate_with_shared_decoder(v, query, max_new_tokens)
            _, raw, _ = self.generate_with_shared_decoder(e, query, max_new_tokens)
            _, ada, t_ada = self.generate_with_shared_decoder(e, query, max_new_tokens, use_adapter=True)
            rows.append({"accept_raw": calculate_acceptance_rate(raw, ref)[0],
                         "accept_adapted": calculate_acceptance_rate(ada, ref)[0], "t_ref": t_ref, "t_adapted": t_ada})
        return {"n": len(rows), "cos_before": cos_before, "cos_after": cos_after, "adapter_losses": losses,
                "accept_raw_mean": float(np.mean([r["accept_raw"] for r in rows])),
                "accept_adapted_mean": float(np.mean([r["accept_adapted"] for r in rows])), "samples": rows}


# ------------------------------------------------------------ token extraction
@torch.no_grad()
def extract_token_pairs(extractor, ds: InstructionDataset, questions: Sequence[str], out_json: Optional[str] = None,
                        max_samples: int = -1, with_hidden: bool = False, resume: bool = True) -> Dict:
    """Greedy answers of both models for every pair; ``extractor`` is a
    ``pipeline.extract_hidden_states.HiddenStateExtractor``.  Output:
    ``{"results": [{sample_idx, question_idx, egpt_tokens, vl_tokens, egpt_text, vl_text}]}``."""
    results: List[Dict] = []
    if resume and out_json and os.path.exists(out_json):
        with open(out_json) as f:
            results = json.load(f).get("results", [])
    done = {(r["sample_idx"], r["question_idx"]) for r in results}
    hidden = {"egpt": [], "vl": []}
    n = len(ds) if max_samples <= 0 else min(max_samples, len(ds))
    for si in range(n):
        for qi, q in enumerate(questions):
            if (si, qi) in done:
                continue
            e = extractor.extract_egpt_hidden(ds, ds[si], q)
            v = extractor.extract_vl_hidden(ds, ds[si], q)
            if e is None or v is None:
                continue
            results.append({"sample_idx": si, "question_idx": qi, "question": q, "egpt_tokens": e[1],
                            "vl_tokens": v[1], "egpt_text": e[2], "vl_text": v[2]})
            if with_hidden:
                hidden["egpt"].append(e[0])
                hidden["vl"].append(v[0])
    out = {"results": results, "num_pairs": len(results)}
    if out_json:
        os.makedirs(os.path.dirname(os.path.abspath(out_json)), exist_ok=True)
        with open(out_json + ".tmp", "w") as f:
            json.dump(out, f)
        os.replace(out_json + ".tmp", out_json)
    if with_hidden:
        out["hidden"] = hidden
    return out


# ----------------------------------------------------------------- alpha sweeps
def consecutive_prefix(draft: Sequence[int], target: Sequence[int], start: int, gamma: int) -> int:
    k = 0
    while k < gamma and start + k < min(len(draft), len(target)) and draft[start + k] == target[start + k]:
        k += 1
    return k


def benchmark_alpha(pairs: Sequence[Dict], gammas: Sequence[int] = (1, 2, 3, 4, 5, 7, 10), c_ratio: float = 1.0,
                    token_adapter=None, device="cpu") -> Dict:
    """Acceptance statistics over extracted token pairs: position-wise match rate
    (alpha), match after the token adapter, and per-gamma mean accepted prefix when SD
    windows are walked along the target sequence, with the theoretical speedups."""
    rates, aligned, lens = [], [], []
    for r in pairs:
        d, t = r["egpt_tokens"], r["vl_tokens"]
        rates.append(calculate_acceptance_rate(d, t)[0])
        lens.append(min(len(d), len(t)))
        if token_adapter is not None:
            aligned.append(compute_aligned_acceptance_rate(d, t, token_adapter, device)["aligned_acceptance_rate"])
    alpha = float(np.mean(rates)) if rates else 0.0
    out: Dict = {"n_pairs": len(pairs), "alpha": alpha, "alpha_std": float(np.std(rates)) if rates else 0.0,
                 "mean_len": float(np.mean(lens)) if lens else 0.0, "c_ratio": c_ratio, "per_gamma": {}}
    if aligned:
        out["alpha_aligned"] = float(np.mean(aligned))
    for g in gammas:
        acc, steps = [], []
        for r in pairs:
            d, t = r["egpt_tokens"], r["vl_tokens"]
            n, pos, a, s = min(len(d), len(t)), 0, 0, 0
            while pos < n:
                k = consecutive_prefix(d, t, pos, g)
                a, s, pos = a + k, s + 1, pos + k + 1
            if s:
                acc.append(a / s)
                steps.append(n / s)
        out["per_gamma"][str(g)] = {
            "mean_accepted_per_step": float(np.mean(acc)) if acc else 0.0,
            "tokens_per_target_call": float(np.mean(steps)) if steps else 1.0,
            "theoretical_speedup": calculate_speculative_speedup(c_ratio, out.get("alpha_aligned", alpha), g)}
    return out
