"""Token-alignment driver: extract -> train -> evaluate.

One CLI for what the reference spreads over feasible/token_alignment/
run_token_alignment.py:402-468 (modes extract/train/evaluate, strategies),
extract_tokens_parallel.py:243-252 and extract_features.py (running both models
for token ids, hidden rows and logits), train_eagle_fusion.py:121-143 and
evaluate_speculative.py:282-288 (gamma sweep of the simulated SD walk).

    extract   --benchmark_json F            token pairs from a benchmark result file
              --dataset_dir D [--random_init ...]   run drafter and verifier (engines)
              --hidden_store S --lm_head P  project a chunked hidden-state store
    train     --strategy {token_adapter,eagle_fusion,token_projection,
                          sequence_distillation,logit_alignment}
    evaluate  --checkpoint C                baseline vs aligned acceptance; EAGLE: gamma sweep

The cached directory holds ``draft_tokens.pt``/``target_tokens.pt`` (+ optional
``draft_hidden.pt``/``target_hidden.pt``/``*_logits.pt``) and ``sample_ids.json``.
Token ids are stored shifted by +1 where they come from hidden-state projection
(0 is the pad id); ids from token files are stored as they are.
"""
from __future__ import annotations

import argparse
import json
import os
from typing import Dict, Optional

import torch

from . import token_align as TA

STRATEGIES = {
    "token_adapter": (TA.TokenAdapter, TA.TokenAdapterConfig),
    "eagle_fusion": (TA.EAGLEFusionModule, TA.EAGLEFusionConfig),
    "token_projection": (TA.TokenProjectionModule, TA.TokenProjectionConfig),
    "sequence_distillation": (TA.SequenceDistillationModule, TA.SequenceDistillationConfig),
    "logit_alignment": (TA.LogitAlignmentModule, TA.LogitAlignmentConfig),
}
NEEDS = {"token_adapter": (), "eagle_fusion": ("draft_hidden",), "token_projection": ("draft_hidden", "target_hidden"),
         "sequence_distillation": ("draft_hidden",), "logit_alignment": ("draft_logits", "target_logits")}


# ---------------------------------------------------------------------- extract
def extract_from_benchmark_results(benchmark_json: str, cached_dir: str, max_seq_len: int = 128) -> TA.TokenAlignmentDataset:
    ds = TA.TokenAlignmentDataset.from_json(benchmark_json, max_seq_len)
    ds.save(cached_dir)
    return ds


def extract_from_models(args, cached_dir: str) -> TA.TokenAlignmentDataset:
    """Run drafter and verifier over an instruction dataset; keep token ids and the
    per-step last-layer hidden rows (the inputs of every strategy but logit alignment)."""
    from ..benchmarks.s1_experiments import extract_token_pairs
    from ..pipeline.dataset import DEFAULT_QUESTION, InstructionDataset, load_questions
    from ..pipeline.extract_hidden_states import HiddenStateExtractor, load_models
    egpt, tok, vl, vproc = load_models(args)
    ext = HiddenStateExtractor(egpt, tok, vl, vproc, max_new_tokens=args.max_new_tokens,
                               double_adaptor=not args.single_adaptor)
    ds = InstructionDataset(args.dataset_dir, args.max_samples)
    qs = load_questions(args.questions_file, args.max_questions if args.max_questions > 0 else 10) \
        if args.questions_file else [DEFAULT_QUESTION]
    out = extract_token_pairs(ext, ds, qs, os.path.join(cached_dir, "token_pairs.json"), args.max_samples,
                              with_hidden=True, resume=False)
    L = args.max_seq_len
    rows = out["results"]
    if not rows:
        raise RuntimeError("no pairs extracted")

    def pad_h(hs):
        H = hs[0].shape[-1]
        t = torch.zeros(len(hs), L, H)
        for i, h in enumerate(hs):
            n = min(L, h.shape[0])
            t[i, :n] = h[:n].float().cpu()
        return t
    # shift by one so that id 0 stays the pad id
    d = torch.tensor([TA._pad_ids([x + 1 for x in r["egpt_tokens"]], L) for r in rows])
    t = torch.tensor([TA._pad_ids([x + 1 for x in r["vl_tokens"]], L) for r in rows])
    tds = TA.TokenAlignmentDataset(d, t, draft_hidden=pad_h(out["hidden"]["egpt"]),
                                   target_hidden=pad_h(out["hidden"]["vl"]),
                                   sample_ids=[f"{r['sample_idx']}_{r['question_idx']}" for r in rows])
    tds.save(cached_dir)
    return tds


def extract_from_store(store_dir: str, lm_head_path: str, cached_dir: str, max_seq_len: int = 64,
                       with_logits_topk: int = 0) -> TA.TokenAlignmentDataset:
    from ..eval.acceptance import load_lm_head
    w = load_lm_head(lm_head_path)
    ds = TA.TokenAlignmentDataset.from_hidden_store(store_dir, w, None, max_seq_len)
    if with_logits_topk:
        # logits for the logit-alignment strategy: full rows are V floats per position,
        # so only materialise them when asked to
        wt = w.float().t()
        ds.draft_logits = ds.draft_hidden.float() @ wt
        ds.target_logits = ds.target_hidden.float() @ wt
    ds.save(cached_dir)
    return ds


# ------------------------------------------------------------------------ train
def _split(ds: TA.TokenAlignmentDataset, val_fraction: float, seed: int = 0):
    n = len(ds)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(seed))
    n_val = max(1, int(n * val_fraction)) if n > 1 else 0
    def take(idx):
        kw = {f: (getattr(ds, f)[idx] if getattr(ds, f) is not None else None) for f in ds.FIELDS}
        return TA.TokenAlignmentDataset(ds.draft_tokens[idx], ds.target_tokens[idx],
                                        sample_ids=[ds.sample_ids[i] for i in idx.tolist()], **kw)
    return take(perm[n_val:]), (take(perm[:n_val]) if n_val else None)


def _infer_config(strategy: str, ds: TA.TokenAlignmentDataset, args) -> object:
    _, cfg_cls = STRATEGIES[strategy]
    vocab = int(max(ds.draft_tokens.max(), ds.target_tokens.max())) + 1
    kw: Dict = {}
    fields = getattr(cfg_cls, "__dataclass_fields__", {})
    def put(name, val):
        if name in fields and val is not None:
            kw[name] = val
    put("draft_vocab_size", max(vocab, args.vocab_size or 0))
    put("target_vocab_size", max(vocab, args.vocab_size or 0))
    if strategy == "logit_alignment":       # the logit rows define the vocabularies
        kw["draft_vocab_size"] = ds.draft_logits.shape[-1]
        kw["target_vocab_size"] = ds.target_logits.shape[-1]
        kw["use_top_k_logits"] = False      # the cache holds full-width rows
    if ds.draft_hidden is not None:
        put("draft_hidden_size", ds.draft_hidden.shape[-1])
    if ds.target_hidden is not None:
        put("target_hidden_size", ds.target_hidden.shape[-1])
    put("hidden_dim", args.hidden_dim)
    put("embed_dim", args.hidden_dim)
    if args.hidden_dim:
        put("ffn_dim", 4 * args.hidden_dim)
        put("bottleneck_dim", max(args.hidden_dim // 4, 8))
    put("num_layers", args.num_layers)
    put("num_fusion_layers", args.num_layers)
    put("num_transformer_layers", args.num_layers)
    put("num_heads", args.num_attention_heads)
    put("num_attention_heads", args.num_attention_heads)
    put("max_context", max(ds.draft_tokens.shape[1], 8))
    if "use_kl_loss" in fields:
        kw["use_kl_loss"] = bool(args.use_kl_loss)
        kw["kl_weight"] = args.kl_weight
    return cfg_cls(**kw)


def train_alignment(cached_dir: str, strategy: str, output_dir: Optional[str], args) -> Dict:
    ds = TA.TokenAlignmentDataset.from_directory(cached_dir)
    for need in NEEDS[strategy]:
        if getattr(ds, need) is None:
            raise ValueError(f"strategy {strategy} needs {need}.pt in {cached_dir}")
    train, val = _split(ds, args.val_fraction)
    model_cls, _ = STRATEGIES[strategy]
    model = model_cls(_infer_config(strategy, ds, args))
    out_dir = output_dir or os.path.join(cached_dir, f"checkpoints_{strategy}")
    os.makedirs(out_dir, exist_ok=True)
    ckpt = os.path.join(out_dir, f"{strategy}_alignment.pt")
    tr = TA.AlignmentTrainer(model, train, val, args.batch_size, args.learning_rate, args.device)
    stats = tr.train(args.num_epochs, ckpt, args.early_stopping)
    if not os.path.exists(ckpt):
        tr.save_checkpoint(ckpt)
    with open(os.path.join(out_dir, "training_stats.json"), "w") as f:
        json.dump({"strategy": strategy, "n_train": len(train), "n_val": len(val) if val else 0,
                   "parameters": sum(p.numel() for p in model.parameters()), **stats}, f, indent=2)
    return {"checkpoint": ckpt, "stats": stats}


# --------------------------------------------------------------------- evaluate
def evaluate_alignment(checkpoint: str, cached_dir: str, strategy: str, device: str = "cpu",
                       output_path: Optional[str] = None) -> Dict:
    ds = TA.TokenAlignmentDataset.from_directory(cached_dir)
    model_cls, _ = STRATEGIES[strategy]
    model = model_cls.load(checkpoint, device)
    res: Dict = {"strategy": strategy, "baseline": TA.compute_baseline_acceptance(ds)}
    if strategy == "token_adapter":
        m = TA.compute_acceptance_metrics(model, ds, device)
        res["aligned"] = {k: m[k] for k in ("acceptance_rate", "top5_rate", "total_correct", "total_tokens")}
    elif strategy == "eagle_fusion":
        res["gamma_sweep"] = TA.evaluate_at_different_gammas(model, ds.draft_hidden, ds.target_tokens, device)
        best = max((v for v in res["gamma_sweep"].values() if "error" not in v),
                   key=lambda v: v["simulated_speedup"], default=None)
        res["aligned"] = {"acceptance_rate": best["acceptance_rate"] if best else 0.0,
                          "best_simulated_speedup": best["simulated_speedup"] if best else 1.0}
    else:
        tr = TA.AlignmentTrainer(model, ds, ds, 32, 1e-4, device)
        res["aligned"] = tr.validate()
    base, al = res["baseline"]["acceptance_rate"], res["aligned"]["acceptance_rate"]
    print(f"[token_alignment] {strategy}: acceptance {100 * base:.2f} % -> {100 * al:.2f} %")
    if output_path:
        with open(output_path, "w") as f:
            json.dump(res, f, indent=2)
    return res


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="token alignment: extract / train / evaluate")
    p.add_argument("--mode", default="train", choices=["extract", "train", "evaluate"])
    p.add_argument("--benchmark_json", default=None)
    p.add_argument("--dataset_dir", default=None)
    p.add_argument("--hidden_store", default=None)
    p.add_argument("--lm_head", default=None)
    p.add_argument("--with_logits", action="store_true")
    p.add_argument("--questions_file", default=None)
    p.add_argument("--max_samples", type=int, default=-1)
    p.add_argument("--max_questions", type=int, default=-1)
    p.add_argument("--max_new_tokens", type=int, default=50)
    p.add_argument("--random_init", default=None, choices=["7b", "tiny"])
    p.add_argument("--egpt_model_path", default=None)
    p.add_argument("--vl_model_path", default=None)
    p.add_argument("--single_adaptor", action="store_true")
    p.add_argument("--cached_dir", default="./cached_outputs")
    p.add_argument("--output_dir", default=None)
    p.add_argument("--strategy", default="token_projection", choices=sorted(STRATEGIES))
    p.add_argument("--num_epochs", type=int, default=20)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    p.add_argument("--max_seq_len", type=int, default=128)
    p.add_argument("--hidden_dim", type=int, default=None)
    p.add_argument("--num_layers", "--num_fusion_layers", dest="num_layers", type=int, default=None)
    p.add_argument("--num_attention_heads", type=int, default=None)
    p.add_argument("--vocab_size", type=int, default=None)
    p.add_argument("--use_kl_loss", action="store_true")
    p.add_argument("--kl_weight", type=float, default=0.5)
    p.add_argument("--early_stopping", type=int, default=10)
    p.add_argument("--val_fraction", type=float, default=0.1)
    p.add_argument("--checkpoint", default=None)
    p.add_argument("--output_path", default=None)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.mode == "extract":
        if args.benchmark_json:
            ds = extract_from_benchmark_results(args.benchmark_json, args.cached_dir, args.max_seq_len)
        elif args.hidden_store:
            if not args.lm_head:
                raise SystemExit("--hidden_store needs --lm_head")
            ds = extract_from_store(args.hidden_store, args.lm_head, args.cached_dir, args.max_seq_len,
                                    with_logits_topk=int(args.with_logits))
        elif args.dataset_dir:
            ds = extract_from_models(args, args.cached_dir)
        else:
            raise SystemExit("extract needs --benchmark_json, --hidden_store or --dataset_dir")
        print(f"[token_alignment] cached {len(ds)} pairs in {args.cached_dir}")
        return ds
    if args.mode == "train":
        return train_alignment(args.cached_dir, args.strategy, args.output_dir, args)
    ckpt = args.checkpoint or os.path.join(args.cached_dir, f"checkpoints_{args.strategy}",
                                           f"{args.strategy}_alignment.pt")
    return evaluate_alignment(ckpt, args.cached_dir, args.strategy, args.device, args.output_path)


if __name__ == "__main__":
    main()
