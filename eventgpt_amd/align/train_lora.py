"""L6 trainer: LoRA on the drafter's decoder so that its decode hidden states land in
the verifier's space.

Per sample (a stored verifier hidden sequence + the drafter's own answer text):
teacher-forced single forward over ``[prompt ‖ answer tokens]``, the last ``gen_len``
final-norm hidden rows are the prediction, loss
``MSE + 0.5 CosineEmbedding + w_tok CE(vl_lm_head(pred), argmax vl_lm_head(target))``;
micro-batch 1 with gradient accumulation, AdamW 1e-4, cosine schedule over epochs,
activation checkpointing, validation on a held-out store, best/final adapters saved
in PEFT layout, ``history.json`` every epoch
(reference: pipeline/adapter_train/train_lora_adapter.py:253-997).

MI355X-first differences: base weights are bf16 (no NF4); the frozen vision tower
runs on the graph-captured engine; samples are sharded over data-parallel ranks
(one process per GPU) and the 19 M LoRA gradients are reduce-scattered once per
optimizer step into a sharded AdamW (``parallel.dp.ShardedAdamW``); host-side sample
preparation (PNG decode, CLIP preprocessing, tokenisation) is prefetched on a thread.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from datetime import datetime
from typing import Dict, Iterator, List, Optional

import numpy as np
import torch

from ..constants import EVENT_TOKEN_INDEX
from ..data.conversation import prepare_event_prompt
from ..data.events import load_image
from ..data.tokenizer import tokenizer_event_token
from ..models.event_chat import get_spatio_temporal_features
from ..parallel.dist import DistInfo, all_reduce_scalar, init_distributed
from ..parallel.dp import ShardedAdamW, broadcast_parameters
from ..pipeline.dataset import InstructionDataset, load_questions
from .lora import apply_lora, count_parameters, lora_parameters, save_lora
from .losses import lora_distill_loss
from .store import _load_pt, read_index

MIN_LEN = 5


class LoRATrainer:
    def __init__(self, model, tokenizer, train_chunks: str, dataset_dir: str, questions: List[str],
                 vl_lm_head: Optional[torch.Tensor] = None, val_chunks: Optional[str] = None,
                 val_dataset_dir: Optional[str] = None, event_image_key: str = "event_image_1f",
                 lora_rank: int = 16, lora_alpha: int = 32, lora_dropout: float = 0.05,
                 token_loss_weight: float = 0.1, learning_rate: float = 1e-4, gradient_accumulation: int = 16,
                 num_epochs: int = 20, early_stopping: int = 5, output_dir: str = "tasks/L6",
                 info: Optional[DistInfo] = None, double_adaptor: bool = True, max_samples: int = -1,
                 gradient_checkpointing: str = "auto", micro_batch: int = 1):
        self.model, self.tok = model, tokenizer
        # samples of one accumulation window that share a forward / backward (right-padded batch)
        self.micro_batch = max(1, min(int(micro_batch), int(gradient_accumulation)))
        self.train_chunks, self.val_chunks = train_chunks, val_chunks
        self.ds = InstructionDataset(dataset_dir)
        self.val_ds = InstructionDataset(val_dataset_dir) if val_dataset_dir else self.ds
        self.questions, self.key = questions, event_image_key
        self.token_loss_weight = token_loss_weight if vl_lm_head is not None else 0.0
        self.lr, self.accum, self.num_epochs, self.early_stopping = learning_rate, gradient_accumulation, \
            num_epochs, early_stopping
        self.output_dir, self.info = output_dir, info or DistInfo()
        self.double_adaptor, self.max_samples = double_adaptor, max_samples
        self.device = model.device
        self.wrapped = apply_lora(model, lora_rank, lora_alpha, lora_dropout)
        # The reference recomputes every block in backward to fit a 24 GB card
        # (train_lora_adapter.py:335-402); the activations of one ~700-token sample are a few
        # GB, so with tens of GB free the recomputation (one extra forward per step) is skipped.
        if gradient_checkpointing == "auto":
            roomy = self.device.type == "cuda" and torch.cuda.mem_get_info(self.device)[0] > (48 << 30)
            gradient_checkpointing = "off" if roomy else "on"
        model.model.gradient_checkpointing = gradient_checkpointing == "on"
        self.vl_lm_head = vl_lm_head.to(self.device, torch.float32) if vl_lm_head is not None else None
        if self.info.world > 1:
            # LoRA A matrices are randomly initialised: every rank starts from rank 0's
            for p in lora_parameters(model):
                if self.info.backend == "gloo" and p.is_cuda:
                    host = p.data.cpu()
                    torch.distributed.broadcast(host, 0)
                    p.data.copy_(host)
                else:
                    torch.distributed.broadcast(p.data, 0)
        self.opt = ShardedAdamW(lora_parameters(model), lr=learning_rate, weight_decay=0.01, overlap=False)
        # the frozen base matrices keep K-contiguous copies for the input-gradient GEMMs while the
        # device has room for them (ops/differentiable.py)
        if self.device.type == "cuda":
            from ..ops.differentiable import enable_transposed_dgrad
            enable_transposed_dgrad(model.model.layers)
        self.history: List[Dict] = []
        self.params = count_parameters(model)

    # ------------------------------------------------------------ sample prep
    @torch.no_grad()
    def prepare_input(self, ds: InstructionDataset, sample: Dict, query: str) -> Optional[Dict]:
        paths = ds.event_image_paths(sample, self.key)
        if not paths or not os.path.exists(paths[0]):
            return None
        img = np.asarray(load_image(paths[0]))
        proc = self.model.get_visual_tower().event_processor
        px = proc(img, return_tensors="pt")["pixel_values"][0]
        ids = tokenizer_event_token(prepare_event_prompt(query, "eventgpt_v1"), self.tok, EVENT_TOKEN_INDEX,
                                    return_tensors="pt")[None]
        return {"pixel_values": px, "input_ids": ids, "event_image_size": list(img.shape[:2])}

    def _prepare_sample(self, ds: InstructionDataset, meta: Dict, target_len: int) -> Optional[Dict]:
        """Host-only work (runs on the prefetch thread)."""
        si, qi = meta["sample_idx"], meta["question_idx"]
        text = meta.get("egpt_text", "")
        if si >= len(ds) or qi >= len(self.questions):
            return None
        # The answer is re-tokenised from its text; a store whose tokenizer cannot
        # round-trip the ids (random weights over a byte vocabulary decode to "")
        # falls back to the drafter's recorded token ids.
        gen = self.tok.encode(text, add_special_tokens=False) if text else list(meta.get("egpt_tokens", []))
        gen = gen[:target_len]
        if len(gen) < MIN_LEN:
            return None
        inp = self.prepare_input(ds, ds[si], self.questions[qi])
        if inp is None:
            return None
        inp["gen_token_ids"] = torch.tensor(gen, dtype=torch.long)
        return inp

    # --------------------------------------------------------------- forward
    def teacher_forced_forward(self, inp: Dict) -> torch.Tensor:
        """Final-norm hidden states at the answer positions ``[gen_len, H]``."""
        m = self.model
        feats = self._event_features(inp)[None]
        ids = torch.cat([inp["input_ids"], inp["gen_token_ids"][None]], 1).to(self.device)
        embeds = m.prepare_inputs_labels_for_multimodal(
            ids, None, None, None, None, event_features=feats, event_image_sizes=[inp["event_image_size"]],
            already_adapted=not self.double_adaptor)[4]
        embeds = embeds.detach().requires_grad_(True)       # lets checkpointed blocks propagate gradients
        out = m.model(inputs_embeds=embeds, use_cache=False)
        n = inp["gen_token_ids"].shape[0]
        return out.last_hidden_state[0, -n:]

    def _event_features(self, inp: Dict) -> torch.Tensor:
        """Pooled event tokens ``[n, H]`` of one sample; tower, projector, adaptor and pooling are frozen."""
        m = self.model
        with torch.no_grad():
            px = inp["pixel_values"].to(self.device, m.dtype)
            if px.is_cuda and m.dtype == torch.bfloat16:
                # one replay of the serving engine's captured launch sequence instead of ~300 eager launches
                return m.encode_event_frames([px]).detach()
            feats = m.get_model().feature_adaptor(m.visval_encode(px[None]))[0]
            return get_spatio_temporal_features([feats]).detach()

    def teacher_forced_forward_batch(self, inps: List[Dict]) -> List[torch.Tensor]:
        """``teacher_forced_forward`` of several samples in ONE pass over the decoder: the sequences are
        right-padded to the longest (causal attention: a sequence's rows never see its own padding, so no
        mask is needed) and each sample's answer rows are sliced out of the batch's hidden states.  The
        projections then run over all samples' rows together -- the regime the GEMMs are fast in -- and
        the values equal the one-by-one passes up to the kernels' tile-dependent rounding."""
        m = self.model
        feats = [self._event_features(inp) for inp in inps]
        seqs = [torch.cat([inp["input_ids"][0], inp["gen_token_ids"]]) for inp in inps]
        width = max(int(q.shape[0]) for q in seqs)
        ids = torch.zeros(len(seqs), width, dtype=torch.long)
        mask = torch.zeros(len(seqs), width, dtype=torch.long)
        for i, q in enumerate(seqs):
            ids[i, :q.shape[0]] = q
            mask[i, :q.shape[0]] = 1
        out = m.prepare_inputs_labels_for_multimodal(
            ids.to(self.device), None, mask.to(self.device), None, None, event_features=feats,
            event_image_sizes=[inp["event_image_size"] for inp in inps], already_adapted=not self.double_adaptor)
        lengths = out[2].sum(1).tolist()
        embeds = out[4].detach().requires_grad_(True)
        hidden = m.model(inputs_embeds=embeds, use_cache=False).last_hidden_state
        preds = []
        for i, inp in enumerate(inps):
            n, end = int(inp["gen_token_ids"].shape[0]), int(lengths[i])
            preds.append(hidden[i, end - n:end])
        return preds

    def _loss(self, pred: torch.Tensor, target: torch.Tensor) -> Dict[str, torch.Tensor]:
        if self.vl_lm_head is not None:
            out = lora_distill_loss(pred, target, self.vl_lm_head, 0.5, self.token_loss_weight)
            with torch.no_grad():
                w = self.vl_lm_head
                out["token_match"] = ((pred.float() @ w.t()).argmax(-1) == (target.float() @ w.t()).argmax(-1)) \
                    .float().mean()
            return out
        p, t = pred.float(), target.float()
        mse = torch.nn.functional.mse_loss(p, t)
        cos = torch.nn.functional.cosine_embedding_loss(p, t, torch.ones(p.shape[0], device=p.device))
        return {"total_loss": mse + 0.5 * cos, "mse_loss": mse, "cos_loss": cos,
                "cos_sim": torch.nn.functional.cosine_similarity(p, t, dim=-1).mean()}

    # ---------------------------------------------------------------- epochs
    def _iter_samples(self, chunks_dir: str, ds: InstructionDataset) -> Iterator:
        """(prepared input, target hidden) pairs of this rank, prepared one ahead."""
        pool = ThreadPoolExecutor(max_workers=1)
        seen = 0
        try:
            for info in read_index(chunks_dir)["chunks"]:
                chunk = _load_pt(os.path.join(chunks_dir, "chunks", info["path"]))
                mine = [i for i in range(len(chunk["metadata"])) if (seen + i) % self.info.world == self.info.rank]
                seen += len(chunk["metadata"])
                if self.max_samples > 0:
                    mine = mine[: self.max_samples]
                fut = None
                for k, i in enumerate(mine):
                    if fut is None:
                        fut = pool.submit(self._prepare_sample, ds, chunk["metadata"][i], int(chunk["seq_lens"][i]))
                    prepared = fut.result()
                    fut = pool.submit(self._prepare_sample, ds, chunk["metadata"][mine[k + 1]],
                                      int(chunk["seq_lens"][mine[k + 1]])) if k + 1 < len(mine) else None
                    if prepared is not None:
                        n = min(prepared["gen_token_ids"].shape[0], int(chunk["seq_lens"][i]))
                        yield prepared, chunk["vl_hidden"][i, :n]
        finally:
            pool.shutdown(wait=False)

    def _run_epoch(self, chunks_dir: str, ds: InstructionDataset, train: bool) -> Dict[str, float]:
        self.model.train(train)
        sums: Dict[str, float] = {}
        n = micro = 0
        if train:
            self.opt.zero_grad()
        buf: List = []

        def flush() -> int:
            """One forward / backward over the buffered samples; returns how many of them counted."""
            nonlocal n
            counted, total = 0, None
            with torch.set_grad_enabled(train):
                if len(buf) > 1:
                    preds = self.teacher_forced_forward_batch([b[0] for b in buf])
                else:
                    preds = [self.teacher_forced_forward(buf[0][0])]
                for pred, (_, target) in zip(preds, buf):
                    k = min(pred.shape[0], target.shape[0])
                    if k < MIN_LEN:
                        continue
                    out = self._loss(pred[:k], target[:k].to(self.device))
                    total = out["total_loss"] if total is None else total + out["total_loss"]
                    for key, v in out.items():
                        sums[key] = sums.get(key, 0.0) + float(v.detach())
                    n += 1
                    counted += 1
                if train and total is not None:
                    (total / self.accum).backward()
            buf.clear()
            return counted

        def after(counted: int) -> None:
            nonlocal micro
            if train and counted:
                micro += counted
                if micro % self.accum == 0:
                    self.opt.step(max_grad_norm=1.0)
                    self.opt.zero_grad()

        for inp, target in self._iter_samples(chunks_dir, ds):
            buf.append((inp, target))
            # a batch never straddles an optimizer step: at most the samples left in the window
            room = self.accum - micro % self.accum if train else self.micro_batch
            if len(buf) >= min(self.micro_batch, room):
                after(flush())
        if buf:
            after(flush())
        if train and micro % self.accum != 0:
            self.opt.step(max_grad_norm=1.0)
            self.opt.zero_grad()
        # average over every rank's samples
        tot_n = all_reduce_scalar(float(n), device=self.device)
        res = {k: all_reduce_scalar(v, device=self.device) / max(tot_n, 1.0) for k, v in sorted(sums.items())}
        res["samples"] = tot_n
        return res

    def train(self) -> Dict:
        os.makedirs(self.output_dir, exist_ok=True)
        best, patience, t0 = float("inf"), 0, time.time()
        for epoch in range(self.num_epochs):
            lr = 1e-6 + 0.5 * (self.lr - 1e-6) * (1 + math.cos(math.pi * epoch / max(self.num_epochs, 1)))
            self.opt.param_groups[0]["lr"] = lr
            tr = self._run_epoch(self.train_chunks, self.ds, True)
            rec = {"epoch": epoch + 1, "lr": lr, **{"train_" + k: v for k, v in tr.items()}}
            metric = tr.get("total_loss", float("inf"))
            if self.val_chunks:
                with torch.no_grad():
                    va = self._run_epoch(self.val_chunks, self.val_ds, False)
                rec.update({"val_" + k: v for k, v in va.items()})
                metric = va.get("total_loss", metric)
            rec["elapsed_s"] = time.time() - t0
            self.history.append(rec)
            improved = metric < best
            if improved:
                best, patience = metric, 0
            else:
                patience += 1
            if self.info.is_main:
                if improved:
                    save_lora(self.model, os.path.join(self.output_dir, "best_model"))
                with open(os.path.join(self.output_dir, "history.json"), "w") as f:
                    json.dump(self.history, f, indent=2)
                print(f"[L6] epoch {epoch + 1}: " + ", ".join(f"{k}={v:.4f}" for k, v in rec.items()
                                                                 if isinstance(v, float) and k != "elapsed_s"),
                      flush=True)
            if patience >= self.early_stopping:
                break
        if self.info.is_main:
            save_lora(self.model, os.path.join(self.output_dir, "final_model"))
            with open(os.path.join(self.output_dir, "config.json"), "w") as f:
                json.dump({"lora": getattr(self.model, "_lora_config", {}), "parameters": self.params,
                           "learning_rate": self.lr, "gradient_accumulation": self.accum,
                           "micro_batch": self.micro_batch,
                           "token_loss_weight": self.token_loss_weight, "event_image_key": self.key,
                           "world_size": self.info.world, "best_metric": best,
                           "created": datetime.now().isoformat()}, f, indent=2)
        if self.device.type == "cuda":
            from ..ops.differentiable import release_transposed_dgrad
            release_transposed_dgrad(self.model)
        return {"history": self.history, "best": best}


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="L6: LoRA fine-tuning of the drafter towards verifier hidden states")
    p.add_argument("--existing_chunks", required=True, help="chunked hidden-state store (training)")
    p.add_argument("--dataset_dir", required=True)
    p.add_argument("--val_chunks", default=None)
    p.add_argument("--val_dataset_dir", default=None)
    p.add_argument("--questions_file", default=None)
    p.add_argument("--max_questions", type=int, default=10)
    p.add_argument("--event_image_key", default="event_image_1f")
    p.add_argument("--lora_rank", type=int, default=16)
    p.add_argument("--lora_alpha", type=int, default=32)
    p.add_argument("--lora_dropout", type=float, default=0.05)
    p.add_argument("--token_loss_weight", type=float, default=0.1)
    p.add_argument("--lm_head_path", default=None)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--batch_size", type=int, default=1, help="micro-batch is one sample; use --gradient_accumulation")
    p.add_argument("--gradient_accumulation", type=int, default=16)
    p.add_argument("--micro_batch", type=int, default=1,
                   help="samples of an accumulation window run as one right-padded forward / backward")
    p.add_argument("--num_epochs", type=int, default=20)
    p.add_argument("--early_stopping", type=int, default=5)
    p.add_argument("--output_dir", default=None)
    p.add_argument("--egpt_model_path", default=None)
    p.add_argument("--random_init", default=None, choices=["7b", "tiny"])
    p.add_argument("--single_adaptor", action="store_true")
    p.add_argument("--max_samples", type=int, default=-1, help="cap on samples per chunk and rank (smoke runs)")
    p.add_argument("--gradient_checkpointing", default="auto", choices=["auto", "on", "off"],
                   help="auto: recompute blocks in backward only when the device has < 48 GB free")
    p.add_argument("--cpu", action="store_true")
    return p


def main(argv=None) -> Dict:
    from ..data.tokenizer import ByteTokenizer, load_tokenizer
    from ..eval.acceptance import load_lm_head
    from ..models import EventChatConfig, EventChatModel
    args = build_parser().parse_args(argv)
    info = init_distributed(use_gpu=False if args.cpu else None)
    dtype = torch.bfloat16 if info.device.type == "cuda" else torch.float32
    if args.random_init:
        cfg = EventChatConfig.eventgpt_7b() if args.random_init == "7b" else EventChatConfig.tiny()
        model, tok = EventChatModel.from_config(cfg, dtype, info.device, seed=0), ByteTokenizer()
    else:
        model = EventChatModel.from_pretrained(args.egpt_model_path, dtype, device=info.device)
        tok = load_tokenizer(args.egpt_model_path)
    questions = load_questions(args.questions_file or os.path.join(args.dataset_dir, "top50_questions.json"),
                               args.max_questions)
    head = load_lm_head(args.lm_head_path) if args.lm_head_path and args.token_loss_weight > 0 else None
    out = args.output_dir or os.path.join("tasks", "L6", "L6_" + datetime.now().strftime("%Y%m%d_%H%M%S"))
    trainer = LoRATrainer(model, tok, args.existing_chunks, args.dataset_dir, questions, head, args.val_chunks,
                          args.val_dataset_dir, args.event_image_key, args.lora_rank, args.lora_alpha,
                          args.lora_dropout, args.token_loss_weight, args.learning_rate,
                          args.gradient_accumulation, args.num_epochs, args.early_stopping, out, info,
                          not args.single_adaptor, args.max_samples, args.gradient_checkpointing, args.micro_batch)
    if info.is_main:
        print(f"[L6] {trainer.params['trainable']:,} trainable of {trainer.params['total']:,} parameters "
              f"({trainer.params['percent']:.3f}%), {len(trainer.wrapped)} wrapped layers, world {info.world}",
              flush=True)
    res = trainer.train()
    res["output_dir"] = out
    return res


if __name__ == "__main__":
    main()
