"""Tokenizer alignment and checking between a drafter and a verifier.

``TokenizerAligner`` diffs two vocabularies, builds a draft-id -> target-id translation
map (shared strings map directly; a draft-only string maps to the longest target string
that contains it or is contained in it) and scores acceptance after translation
(reference: feasible/tokenizer_alignment/align_tokenizers.py:18-169).
``check_tokenizers`` reports special tokens, vocabulary sizes and round-trip
behaviour on probe sentences (reference: feasible/tokenizer_check/tokenizer_check.py:35-175).
Tokenizers only need ``get_vocab()``, ``encode`` and ``decode``; the translation map
is also exported as a dense id tensor for use on the device.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch


def _vocab(tok) -> Dict[str, int]:
    if hasattr(tok, "get_vocab"):
        return dict(tok.get_vocab())
    raise TypeError(f"{type(tok).__name__} has no get_vocab()")


def _encode(tok, text: str) -> List[int]:
    try:
        return list(tok.encode(text, add_special_tokens=False))
    except TypeError:
        return list(tok.encode(text))


class TokenizerAligner:
    def __init__(self, draft_tokenizer, target_tokenizer, verbose: bool = False):
        self.draft, self.target = draft_tokenizer, target_tokenizer
        self.draft_str2id, self.target_str2id = _vocab(draft_tokenizer), _vocab(target_tokenizer)
        self.draft_id2str = {i: s for s, i in self.draft_str2id.items()}
        self.target_id2str = {i: s for s, i in self.target_str2id.items()}
        d, t = set(self.draft_str2id), set(self.target_str2id)
        self.shared_tokens, self.draft_only_tokens, self.target_only_tokens = d & t, d - t, t - d
        if verbose:
            print(self.summary())

    def summary(self) -> Dict[str, float]:
        big = max(len(self.draft_str2id), len(self.target_str2id), 1)
        return {"draft_vocab": len(self.draft_str2id), "target_vocab": len(self.target_str2id),
                "shared": len(self.shared_tokens), "draft_only": len(self.draft_only_tokens),
                "target_only": len(self.target_only_tokens), "overlap": len(self.shared_tokens) / big,
                "same_ids": sum(1 for s in self.shared_tokens if self.draft_str2id[s] == self.target_str2id[s])}

    def compare_token_sequences(self, text: str, max_length: int = 100) -> Dict:
        a, b = _encode(self.draft, text)[:max_length], _encode(self.target, text)[:max_length]
        n = max(min(len(a), len(b)), 1)
        return {"draft_tokens": a, "target_tokens": b, "draft_decoded": [self.draft.decode([t]) for t in a],
                "target_decoded": [self.target.decode([t]) for t in b],
                "token_match_rate": sum(1 for x, y in zip(a, b) if x == y) / n}

    def create_translation_map(self) -> Dict[int, int]:
        tmap = {self.draft_str2id[s]: self.target_str2id[s] for s in self.shared_tokens}
        self.n_direct = len(tmap)
        targets = sorted(self.target_str2id.items(), key=lambda kv: -len(kv[0]))
        for s in self.draft_only_tokens:
            if not s:
                continue
            best, best_len = None, 0
            for ts, tid in targets:
                span = len(ts) if s in ts else (len(s) if ts and ts in s else 0)
                if span > best_len:
                    best, best_len = tid, span
            if best is not None:
                tmap[self.draft_str2id[s]] = best
        self.n_substring = len(tmap) - self.n_direct
        return tmap

    def translation_tensor(self, tmap: Optional[Dict[int, int]] = None, device="cpu") -> torch.Tensor:
        """``ids_target = table[ids_draft]``; unmapped ids map to themselves."""
        tmap = self.create_translation_map() if tmap is None else tmap
        size = max(self.draft_id2str) + 1 if self.draft_id2str else 0
        table = torch.arange(size, dtype=torch.long)
        if tmap:
            k = torch.tensor(list(tmap.keys()), dtype=torch.long)
            table[k] = torch.tensor(list(tmap.values()), dtype=torch.long)
        return table.to(device)

    @staticmethod
    def translate_tokens(token_ids: Sequence[int], tmap: Dict[int, int]) -> Tuple[List[int], List[int]]:
        out, unmapped = [], []
        for t in token_ids:
            if t in tmap:
                out.append(tmap[t])
            else:
                out.append(t)
                unmapped.append(t)
        return out, unmapped

    def compute_acceptance_rate_with_alignment(self, draft_tokens: Sequence[int], target_tokens: Sequence[int],
                                               tmap: Dict[int, int]) -> Tuple[float, int, int]:
        tr, _ = self.translate_tokens(draft_tokens, tmap)
        acc = sum(1 for a, b in zip(tr, target_tokens) if a == b)
        return (acc / len(draft_tokens) if draft_tokens else 0.0), acc, len(draft_tokens)


PROBE_SENTENCES = ("What are the key elements in this scene?", "A car is driving down a wet road at night.",
                   "USER: <video>\nDescribe the motion. ASSISTANT:")


def check_tokenizers(draft_tokenizer, target_tokenizer, probes: Sequence[str] = PROBE_SENTENCES) -> Dict:
    al = TokenizerAligner(draft_tokenizer, target_tokenizer)
    rep: Dict = {"vocab": al.summary(), "special": {}, "probes": []}
    for name in ("bos_token_id", "eos_token_id", "pad_token_id", "unk_token_id"):
        rep["special"][name] = {"draft": getattr(draft_tokenizer, name, None),
                                "target": getattr(target_tokenizer, name, None)}
    rep["special_match"] = all(v["draft"] == v["target"] for v in rep["special"].values())
    for text in probes:
        cmp = al.compare_token_sequences(text)
        rt_d = draft_tokenizer.decode(cmp["draft_tokens"])
        rt_t = target_tokenizer.decode(cmp["target_tokens"])
        rep["probes"].append({"text": text, "n_draft": len(cmp["draft_tokens"]), "n_target": len(cmp["target_tokens"]),
                              "match_rate": cmp["token_match_rate"], "roundtrip_draft": rt_d.strip() == text.strip(),
                              "roundtrip_target": rt_t.strip() == text.strip()})
    rep["compatible"] = rep["special_match"] and all(p["match_rate"] == 1.0 for p in rep["probes"])
    return rep


def analyze_benchmark_results(rows_or_path, tmap: Optional[Dict[int, int]] = None,
                              positions: Sequence[int] = (1, 5, 10, 20)) -> Dict:
    """Re-analyse a benchmark result file (list of ``{egpt_token_ids, llava_token_ids}``):
    overall and per-sample acceptance, match rate at given positions, first-divergence
    histogram — the analysis behind "the tokenizers agree, the *models* diverge"
    (feasible/tokenizer_alignment/test_benchmark_alignment.py:12-148).  With ``tmap`` the
    draft ids are translated first."""
    import json
    rows = rows_or_path
    if isinstance(rows_or_path, str):
        with open(rows_or_path) as f:
            rows = json.load(f)
        if isinstance(rows, dict):
            rows = rows.get("results", rows.get("samples", []))
    per, tot_d, tot_a = [], 0, 0
    at_pos = {p: [0, 0] for p in positions}
    first_div: Dict[int, int] = {}
    for i, r in enumerate(rows):
        d = list(r.get("egpt_token_ids") or r.get("egpt_tokens") or [])
        t = list(r.get("llava_token_ids") or r.get("vl_tokens") or [])
        if not d or not t:
            continue
        if tmap is not None:
            d, _ = TokenizerAligner.translate_tokens(d, tmap)
        n = min(len(d), len(t))
        hits = [d[j] == t[j] for j in range(n)]
        acc = sum(hits)
        div = next((j for j, h in enumerate(hits) if not h), n)
        first_div[div] = first_div.get(div, 0) + 1
        for p_ in positions:
            if p_ - 1 < n:
                at_pos[p_][1] += 1
                at_pos[p_][0] += int(hits[p_ - 1])
        per.append({"sample_id": r.get("id", i), "draft_len": len(d), "target_len": len(t), "accepted": acc,
                    "alpha": acc / len(d), "first_divergence": div})
        tot_d, tot_a = tot_d + len(d), tot_a + acc
    return {"n_samples": len(per), "total_draft_tokens": tot_d, "total_accepted": tot_a,
            "alpha": tot_a / tot_d if tot_d else 0.0,
            "match_at_position": {p_: (h / n if n else 0.0) for p_, (h, n) in at_pos.items()},
            "first_divergence_hist": dict(sorted(first_div.items())),
            "mean_first_divergence": (sum(k * v for k, v in first_div.items()) / max(sum(first_div.values()), 1)),
            "samples": per}


def matched_sampling_sweep(generate_pair, settings: Sequence[Tuple[float, float]] = ((0.0, 1.0), (0.3, 1.0), (0.6, 1.0),
                                                                                   (1.0, 0.9)),
                           n_samples: int = 5) -> List[Dict]:
    """Does matching the sampling parameters of drafter and verifier raise acceptance?
    (test_matched_sampling.py:13-140).  ``generate_pair(i, temperature, top_p)`` returns
    ``(draft_ids, target_ids)`` for sample ``i``; every setting is applied to BOTH models."""
    out = []
    for temp, top_p in settings:
        rows = []
        for i in range(n_samples):
            d, t = generate_pair(i, temp, top_p)
            rows.append({"egpt_token_ids": list(d), "llava_token_ids": list(t)})
        rep = analyze_benchmark_results(rows)
        out.append({"temperature": temp, "top_p": top_p, "alpha": rep["alpha"],
                    "mean_first_divergence": rep["mean_first_divergence"], "n_samples": rep["n_samples"]})
    return out


def main(argv=None) -> Dict:
    import argparse
    import json

    from ..data.tokenizer import load_tokenizer
    p = argparse.ArgumentParser(description="Compare / align drafter and verifier tokenizers")
    p.add_argument("--eventgpt_model_path", default=None)
    p.add_argument("--videollava_model_path", default=None)
    p.add_argument("--save_map", default=None, help="write the id translation table (.pt)")
    p.add_argument("--analyze", default=None, help="benchmark result JSON to re-analyse")
    args = p.parse_args(argv)
    if args.analyze:
        tmap = None
        if args.eventgpt_model_path and args.videollava_model_path:
            tmap = TokenizerAligner(load_tokenizer(args.eventgpt_model_path),
                                    load_tokenizer(args.videollava_model_path)).create_translation_map()
        rep = analyze_benchmark_results(args.analyze, tmap)
        print(json.dumps({k: v for k, v in rep.items() if k != "samples"}, indent=2))
        return rep
    if not (args.eventgpt_model_path and args.videollava_model_path):
        p.error("--eventgpt_model_path and --videollava_model_path are required")
    d, t = load_tokenizer(args.eventgpt_model_path), load_tokenizer(args.videollava_model_path)
    rep = check_tokenizers(d, t)
    print(json.dumps(rep, indent=2, default=str))
    if args.save_map:
        torch.save({"table": TokenizerAligner(d, t).translation_tensor()}, args.save_map)
    return rep


if __name__ == "__main__":
    main()
