"""Instruction fine-tuning of the event-stream MLLM: projector / adaptor alignment, LoRA,
or the full decoder, data-parallel over RCCL.

The reference ships the *model side* of training — ``initialize_vision_modules`` /
``initialize_vision_tokenizer`` with ``tune_mm_mlp_adapter`` and the ``pretrain_*`` partial
checkpoints (model/EventChatModel.py:110-172, 202-244) — but no training loop
(script/train.py is an unrelated GPT-2 template).  This is that loop, LLaVA-recipe shaped:

  * samples come from the instruction JSON (``<event>\\n question`` / answer pairs with
    ``event_image`` rasters or raw ``event_npy`` streams);
  * the prompt is the ``eventgpt_v1`` conversation, labels are ``IGNORE_INDEX`` up to and
    including ``ASSISTANT:`` and on every event position, the answer + ``</s>`` is supervised;
  * the CLIP tower is frozen; ``--tune projector`` trains ``visual_projector`` +
    ``feature_adaptor`` (stage 1), ``--tune lora`` adds LoRA on the decoder's q/k/v/o,
    ``--tune full`` trains the whole decoder in bf16 with fp32 master shards;
  * one process per GPU: rank-sharded samples, gradient accumulation, reduce-scatter +
    sharded AdamW + all-gather (``parallel.dp.ShardedAdamW``), warm-up/cosine schedule,
    global-norm clipping, resumable state;
  * ``--micro_batch N``: N samples of an accumulation window share ONE forward / backward -- right-padded
    to the longest, the projections over all their rows (8 samples are ~6.7 K rows: the GEMMs run at
    1.0-1.1 PFLOP/s instead of the 0.5-0.75 of one 840-row sample), causal attention per sequence with
    no mask (padding sits at the end), and the loss the mean of the per-sample means, so the step is
    the one that N separate passes accumulate (the reference runs batch 1 x 16 accumulation steps,
    pipeline/adapter_train/train_lora_adapter.py:713-734);
  * outputs: ``mm_projector.pt`` with the ``model.visual_projector.*`` / ``model.feature_adaptor.*``
    keys that ``initialize_vision_modules(pretrain_mm_mlp_adapter=…, pretrain_feature_adaptor=…)``
    loads, a PEFT-layout LoRA directory, or a full HF-layout checkpoint.

    torchrun --nproc-per-node 8 -m eventgpt_amd.cli.train_sft --model_path ckpt --dataset_dir D --tune projector
"""
from __future__ import annotations

import argparse
import json
import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from ..constants import EVENT_TOKEN_INDEX, IGNORE_INDEX
from ..data.conversation import prepare_event_prompt
from ..data.events import event_frames, load_event_npy, load_image
from ..data.tokenizer import tokenizer_event_token
from ..ops.event_ops import preprocess_frames
from ..parallel.dist import all_reduce_scalar, init_distributed, shutdown
from ..parallel.dp import ShardedAdamW, broadcast_parameters
from ..pipeline.dataset import InstructionDataset
from ..utils.schedules import LinearWarmupCosineLR


# ------------------------------------------------------------------------ data
def answer_of(sample: Dict) -> Optional[str]:
    for turn in sample.get("conversations", []):
        if turn.get("from") == "gpt":
            return turn.get("value", "")
    return None


def build_example(tokenizer, question: str, answer: str, conv_mode: str = "eventgpt_v1",
                  max_len: int = 2048) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(input_ids, labels)``: the conversation with one ``EVENT_TOKEN_INDEX`` placeholder;
    only the answer and the closing EOS are supervised."""
    prompt = prepare_event_prompt(question, conv_mode)
    p_ids = tokenizer_event_token(prompt, tokenizer, EVENT_TOKEN_INDEX, return_tensors="pt")
    full = tokenizer_event_token(prompt + " " + answer, tokenizer, EVENT_TOKEN_INDEX, return_tensors="pt")
    eos = getattr(tokenizer, "eos_token_id", None)
    if eos is not None:
        full = torch.cat([full, torch.tensor([eos], dtype=full.dtype)])
    full = full[:max_len]
    labels = full.clone()
    labels[: min(p_ids.shape[0], labels.shape[0])] = IGNORE_INDEX
    labels[full == EVENT_TOKEN_INDEX] = IGNORE_INDEX
    return full, labels


def load_event_rasters(ds: InstructionDataset, sample: Dict, num_frames: int, key: str) -> Optional[List[np.ndarray]]:
    paths = [p for p in ds.event_image_paths(sample, key)[:num_frames] if os.path.exists(p)]
    if paths:
        return [np.asarray(load_image(p)) for p in paths]
    npy = ds.event_npy_path(sample)
    if npy and os.path.exists(npy):
        return event_frames(load_event_npy(npy, trusted=True), num_frames)   # this package's own datasets
    return None


# ----------------------------------------------------------------------- model
def select_trainable(model, tune: str, lora_rank: int = 16, lora_alpha: int = 32, lora_dropout: float = 0.05):
    """Freeze everything, then open what ``tune`` names.  Returns the trainable parameters."""
    model.requires_grad_(False)
    inner = model.get_model()
    opened = [inner.visual_projector]
    if getattr(inner, "feature_adaptor", None) is not None:
        opened.append(inner.feature_adaptor)
    for m in opened:
        m.requires_grad_(True)
    if tune == "lora":
        from ..align.lora import apply_lora, lora_parameters
        apply_lora(model, r=lora_rank, alpha=lora_alpha, dropout=lora_dropout)
        for p in lora_parameters(model):
            p.requires_grad_(True)
        for m in opened:                       # wrapping the decoder must not re-freeze the bridge
            m.requires_grad_(True)
    elif tune == "full":
        inner.layers.requires_grad_(True)
        inner.norm.requires_grad_(True)
        inner.embed_tokens.requires_grad_(True)
        model.lm_head.requires_grad_(True)
    elif tune != "projector":
        raise ValueError(f"--tune {tune}")
    return [p for p in model.parameters() if p.requires_grad]


def projector_state(model) -> Dict[str, torch.Tensor]:
    """The partial checkpoint of the multimodal bridge, under the key prefixes the
    training-time loader strips (EventChatModel.py:132-172)."""
    inner = model.get_model()
    sd = {f"model.visual_projector.{k}": v.detach().cpu() for k, v in inner.visual_projector.state_dict().items()}
    if getattr(inner, "feature_adaptor", None) is not None:
        sd.update({f"model.feature_adaptor.{k}": v.detach().cpu()
                   for k, v in inner.feature_adaptor.state_dict().items()})
    return sd


class SFTTrainer:
    def __init__(self, model, tokenizer, dataset: InstructionDataset, info, tune: str = "projector",
                 num_frames: int = 5, event_image_key: str = "event_image", learning_rate: float = 1e-3,
                 weight_decay: float = 0.0, warmup_ratio: float = 0.03, gradient_accumulation: int = 1,
                 max_len: int = 2048, output_dir: str = "tasks/sft", lora: Optional[Dict] = None,
                 dgrad_copies: bool = True, broadcast: bool = True, micro_batch: int = 1):
        self.model, self.tok, self.ds, self.info = model, tokenizer, dataset, info
        self.tune, self.num_frames, self.key = tune, num_frames, event_image_key
        self.accum, self.max_len, self.out = max(1, gradient_accumulation), max_len, output_dir
        self.micro_batch = max(1, min(int(micro_batch), self.accum))
        self.lr, self.warmup_ratio = learning_rate, warmup_ratio
        self.params = select_trainable(model, tune, **(lora or {}))
        if info.world > 1 and broadcast:       # (off when every rank built the model from one seed)
            broadcast_parameters(model)
        self.opt = ShardedAdamW(self.params, lr=learning_rate, weight_decay=weight_decay, overlap=False)
        # frozen decoder weights (projector / LoRA tuning): keep K-contiguous copies for the
        # input-gradient GEMMs when the device has the room (one more copy of the frozen matrices)
        self.dgrad_copy_bytes = 0
        if dgrad_copies and tune != "full" and next(model.parameters()).is_cuda:
            from ..ops.differentiable import enable_transposed_dgrad
            self.dgrad_copy_bytes = enable_transposed_dgrad(model.get_model().layers)
        self.history: List[Dict] = []
        self.step = 0
        os.makedirs(output_dir, exist_ok=True)

    def prepare_sample(self, sample: Dict) -> Optional[Dict]:
        """Host-only half of a sample (image decode / rasterisation, tokenisation); the
        training loop runs it one sample ahead on a worker thread."""
        ans = answer_of(sample)
        imgs = load_event_rasters(self.ds, sample, self.num_frames, self.key)
        if not ans or not imgs:
            return None
        ids, labels = build_example(self.tok, self.ds.question(sample), ans, max_len=self.max_len)
        return {"imgs": imgs, "ids": ids, "labels": labels}

    # one (sample) -> loss; None if the sample has no usable event data / answer
    def sample_loss(self, sample: Dict, prepared: Optional[Dict] = None) -> Optional[torch.Tensor]:
        m = self.model
        prepared = prepared if prepared is not None else self.prepare_sample(sample)
        if prepared is None:
            return None
        imgs, ids, labels = prepared["imgs"], prepared["ids"], prepared["labels"]
        size = m.get_visual_tower().event_processor.size
        frames = list(preprocess_frames(imgs, m.device, size, m.dtype))
        out = m(input_ids=ids[None].to(m.device), labels=labels[None].to(m.device), event_tensors=[frames],
                event_image_sizes=[list(imgs[0].shape[:2])])
        return out.loss

    def batch_loss(self, prepared: List[Dict]) -> Optional[torch.Tensor]:
        """Mean of the per-sample losses of ``prepared`` (the host halves of several samples) from ONE
        forward: ids and labels right-padded to the longest sample, every sample's frames through the
        vision path together.  Samples whose frame counts differ from the first one's are run on their own."""
        prepared = [p for p in prepared if p is not None]
        if not prepared:
            return None
        if len(prepared) == 1:
            return self.sample_loss({}, prepared[0])
        m = self.model
        t = len(prepared[0]["imgs"])
        same = [p for p in prepared if len(p["imgs"]) == t]
        rest = [p for p in prepared if len(p["imgs"]) != t]
        size = m.get_visual_tower().event_processor.size
        n, width = len(same), max(int(p["ids"].shape[0]) for p in same)
        ids = torch.zeros(n, width, dtype=torch.long)
        labels = torch.full((n, width), IGNORE_INDEX, dtype=torch.long)
        mask = torch.zeros(n, width, dtype=torch.long)
        for i, p in enumerate(same):
            k = int(p["ids"].shape[0])
            ids[i, :k], labels[i, :k], mask[i, :k] = p["ids"], p["labels"], 1
        frames = [list(preprocess_frames(p["imgs"], m.device, size, m.dtype)) for p in same]
        out = m(input_ids=ids.to(m.device), attention_mask=mask.to(m.device), labels=labels.to(m.device),
                event_tensors=frames, event_image_sizes=[list(p["imgs"][0].shape[:2]) for p in same],
                loss_reduction="sequence_mean")
        total = out.loss * len(same)
        for p in rest:
            total = total + self.sample_loss({}, p)
        return total / len(prepared)

    def _state_path(self) -> str:
        return os.path.join(self.out, f"trainer_state_rank{self.info.rank}.pt")

    def save_state(self, epoch: int) -> None:
        trainable = {n: p.detach().cpu() for n, p in self.model.named_parameters() if p.requires_grad}
        tmp = self._state_path() + ".tmp"
        torch.save({"trainable": trainable, "opt": self.opt.state_dict(), "step": self.step, "epoch": epoch,
                    "history": self.history}, tmp)
        os.replace(tmp, self._state_path())

    def load_state(self) -> int:
        st = torch.load(self._state_path(), map_location="cpu", weights_only=False)   # written by save_state
        named = dict(self.model.named_parameters())
        with torch.no_grad():
            for n, v in st["trainable"].items():
                named[n].copy_(v.to(named[n].device, named[n].dtype))
        self.opt.load_state_dict(st["opt"])
        self.step, self.history = st["step"], st["history"]
        return st["epoch"]

    def save_outputs(self, tag: str = "") -> Dict[str, str]:
        paths = {}
        if not self.info.is_main:
            return paths
        paths["projector"] = os.path.join(self.out, f"mm_projector{tag}.pt")
        torch.save(projector_state(self.model), paths["projector"])
        if self.tune == "lora":
            from ..align.lora import save_lora
            paths["lora"] = os.path.join(self.out, f"lora{tag}")
            save_lora(self.model, paths["lora"])
        elif self.tune == "full":
            paths["model"] = os.path.join(self.out, f"checkpoint{tag}")
            self.model.save_pretrained(paths["model"], self.tok)
        return paths

    def train(self, num_epochs: int = 1, max_samples: int = -1, resume: bool = False, logging_steps: int = 10,
              save_every_epoch: bool = True, seed: int = 0) -> Dict:
        info = self.info
        idx_all = list(range(len(self.ds)))
        if max_samples > 0:
            idx_all = idx_all[:max_samples]
        mine = idx_all[info.rank::info.world]
        n_local = len(idx_all) // info.world if info.world > 1 else len(idx_all)   # equal step counts on every rank
        mine = mine[:max(n_local, 1)]
        steps_per_epoch = max(1, math.ceil(len(mine) / self.accum))
        total = steps_per_epoch * num_epochs
        warm = max(1, int(total * self.warmup_ratio))
        sched = LinearWarmupCosineLR(self.opt, total, min_lr=0.0, init_lr=self.lr, warmup_steps=warm,
                                     warmup_start_lr=0.0)
        start_epoch = 0
        if resume and os.path.exists(self._state_path()):
            start_epoch = self.load_state() + 1
        import time
        gen = torch.Generator().manual_seed(seed)
        for _ in range(start_epoch):                     # keep the shuffles of a resumed run aligned
            torch.randperm(len(mine), generator=gen)
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1)
        for epoch in range(start_epoch, num_epochs):
            self.model.train()
            order = torch.randperm(len(mine), generator=gen).tolist()
            run, n_run, micro, skipped = 0.0, 0, 0, 0
            t_epoch, steps0 = time.perf_counter(), self.step
            self.opt.zero_grad()
            # windows of ``micro_batch`` samples that never straddle an optimizer step; the host half of
            # the next window is prepared on a worker thread while this one runs on the GPU
            windows, j = [], 0
            while j < len(order):
                room = self.accum - (j % self.accum)
                k = min(self.micro_batch, room, len(order) - j)
                windows.append(order[j:j + k])
                j += k
            prep_window = lambda w: [self.prepare_sample(self.ds[mine[oi]]) for oi in w]
            fut = pool.submit(prep_window, windows[0]) if windows else None
            for wi, win in enumerate(windows):
                prepared = fut.result()
                fut = pool.submit(prep_window, windows[wi + 1]) if wi + 1 < len(windows) else None
                usable = [p for p in prepared if p is not None]
                # every rank must take the same number of optimizer steps: a rank that
                # cannot use a sample contributes a zero gradient for it
                skipped += len(prepared) - len(usable)
                loss = self.batch_loss(usable) if usable else None
                if loss is not None:
                    (loss * (len(usable) / self.accum)).backward()
                    run, n_run = run + float(loss.detach()) * len(usable), n_run + len(usable)
                micro += len(win)
                if micro % self.accum and wi + 1 < len(windows):
                    continue
                sched.step(0 if self.step < warm else self.step, self.step)
                self.opt.step(max_grad_norm=1.0)
                self.opt.zero_grad()
                self.step += 1
                if info.is_main and self.step % logging_steps == 0:
                    print(f"[train_sft] epoch {epoch + 1} step {self.step}/{total} loss {run / max(n_run, 1):.4f} "
                          f"lr {self.opt.param_groups[0]['lr']:.2e}", flush=True)
            mean = all_reduce_scalar(run / max(n_run, 1), device=info.device) / info.world
            if info.device.type == "cuda":
                torch.cuda.synchronize(info.device)
            dt = time.perf_counter() - t_epoch
            self.history.append({"epoch": epoch + 1, "train_loss": mean, "steps": self.step, "skipped": skipped,
                                 "seconds": dt, "samples_per_s": info.world * len(order) / dt if dt > 0 else 0.0,
                                 "s_per_step": dt / max(self.step - steps0, 1)})
            if save_every_epoch:
                self.save_state(epoch)
        pool.shutdown(wait=True)
        if self.dgrad_copy_bytes:               # training is over: hand the frozen matrices' copies back
            from ..ops.differentiable import release_transposed_dgrad
            release_transposed_dgrad(self.model)
            self.dgrad_copy_bytes = 0
        paths = self.save_outputs()
        if info.is_main:
            with open(os.path.join(self.out, "history.json"), "w") as f:
                json.dump({"history": self.history, "tune": self.tune, "world_size": info.world,
                           "trainable_parameters": sum(p.numel() for p in self.params)}, f, indent=2)
        return {"history": self.history, "paths": paths, "steps": self.step}


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="instruction fine-tuning (projector / LoRA / full), data parallel")
    p.add_argument("--model_path", default=None)
    p.add_argument("--random_init", default=None, choices=["7b", "tiny"])
    p.add_argument("--dataset_dir", required=True)
    p.add_argument("--output_dir", default="tasks/sft")
    p.add_argument("--tune", default="projector", choices=["projector", "lora", "full"])
    p.add_argument("--pretrain_mm_mlp_adapter", default=None, help="mm_projector.pt of an earlier stage")
    p.add_argument("--num_frames", type=int, default=5)
    p.add_argument("--event_image_key", default="event_image")
    p.add_argument("--num_epochs", type=int, default=1)
    p.add_argument("--learning_rate", type=float, default=1e-3)
    p.add_argument("--weight_decay", type=float, default=0.0)
    p.add_argument("--warmup_ratio", type=float, default=0.03)
    p.add_argument("--gradient_accumulation", type=int, default=1)
    p.add_argument("--micro_batch", type=int, default=1,
                   help="samples of an accumulation window that share one forward / backward (<= gradient_accumulation)")
    p.add_argument("--model_max_length", type=int, default=2048)
    p.add_argument("--lora_rank", type=int, default=16)
    p.add_argument("--lora_alpha", type=int, default=32)
    p.add_argument("--lora_dropout", type=float, default=0.05)
    p.add_argument("--max_samples", type=int, default=-1)
    p.add_argument("--resume", action="store_true")
    p.add_argument("--no_save_state", action="store_true",
                   help="skip the per-epoch resumable state (full tuning of a 7B writes ~95 GB per rank)")
    p.add_argument("--cpu", action="store_true")
    return p


def load_model(args, info):
    from ..data.tokenizer import ByteTokenizer, load_tokenizer
    from ..models import EventChatConfig, EventChatModel
    dtype = torch.float32 if args.cpu else torch.bfloat16
    if args.random_init:
        tok = ByteTokenizer()
        if args.random_init == "7b":
            cfg = EventChatConfig.eventgpt_7b()
        else:
            cfg = EventChatConfig.tiny()
        model = EventChatModel.from_config(cfg, dtype, info.device, seed=0)
    else:
        model = EventChatModel.from_pretrained(args.model_path, dtype, device=info.device)
        tok = load_tokenizer(args.model_path)
    if args.pretrain_mm_mlp_adapter:
        sd = torch.load(args.pretrain_mm_mlp_adapter, map_location="cpu", weights_only=True)
        inner = model.get_model()
        inner.visual_projector.load_state_dict({k[len("model.visual_projector."):]: v for k, v in sd.items()
                                                if k.startswith("model.visual_projector.")})
        fa = {k[len("model.feature_adaptor."):]: v for k, v in sd.items() if k.startswith("model.feature_adaptor.")}
        if fa and getattr(inner, "feature_adaptor", None) is not None:
            inner.feature_adaptor.load_state_dict(fa)
    return model, tok


def main(argv=None) -> Dict:
    args = build_parser().parse_args(argv)
    info = init_distributed(use_gpu=not args.cpu)
    model, tok = load_model(args, info)
    ds = InstructionDataset(args.dataset_dir)
    tr = SFTTrainer(model, tok, ds, info, args.tune, args.num_frames, args.event_image_key, args.learning_rate,
                    args.weight_decay, args.warmup_ratio, args.gradient_accumulation, args.model_max_length,
                    args.output_dir, dict(lora_rank=args.lora_rank, lora_alpha=args.lora_alpha,
                                          lora_dropout=args.lora_dropout), micro_batch=args.micro_batch)
    res = tr.train(args.num_epochs, args.max_samples, args.resume, save_every_epoch=not args.no_save_state)
    if info.is_main:
        last = res["history"][-1] if res["history"] else {}
        print(f"[train_sft] done: {res['steps']} steps, final loss {last.get('train_loss', float('nan')):.4f}")
    if info.world > 1:
        shutdown()
    return res


if __name__ == "__main__":
    main()
