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


def main(argv=None) -> Dict:
    import argparse
    import json

    from ..data.tokenizer import load_tokenizer
    p = argparse.ArgumentParser(description="Compare / align drafter and verifier tokenizers")
    p.add_argument("--eventgpt_model_path", required=True)
    p.add_argument("--videollava_model_path", required=True)
    p.add_argument("--save_map", default=None, help="write the id translation table (.pt)")
    args = p.parse_args(argv)
    d, t = load_tokenizer(args.eventgpt_model_path), load_tokenizer(args.videollava_model_path)
    rep = check_tokenizers(d, t)
    print(json.dumps(rep, indent=2, default=str))
    if args.save_map:
        torch.save({"table": TokenizerAligner(d, t).translation_tensor()}, args.save_map)
    return rep


if __name__ == "__main__":
    main()
