"""Adapter trainer (L1-L5F, B1) — single GPU or data-parallel over RCCL.

Training recipe of the reference (pipeline/adapter_train/train_hidden_adapter.py:
270-548): AdamW(lr 1e-3, wd 0.01), cosine annealing over the epoch count down to
1e-6, gradient clipping at 1.0, per-epoch validation (loss, cosine, accept@
{.80,.85,.90,.95}), best/final checkpoints, early stopping, ``history.json``,
``training_curves.png``, ``config.json`` in ``tasks/<L?>/<L?>_<timestamp>/``.

What is different here:
  * the optimizer is ``parallel.dp.ShardedAdamW``: gradients are reduce-scattered
    in flat buckets while backward is still running, AdamW state is sharded,
    parameters are all-gathered; with one process it degenerates to plain AdamW;
  * every rank reads a disjoint slice of each chunk; validation metrics and the
    early-stopping decision are reduced so all ranks agree;
  * training is resumable: ``last.pt`` holds model, optimizer shard, scheduler,
    history and patience, and ``--resume`` continues from it (the reference saved
    the optimizer state but could not reload it);
  * bf16 autocast on GPU (the data is stored fp32).
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import time
from datetime import datetime
from typing import Dict, Optional

import torch
import torch.distributed as dist

from ..parallel.dist import DistInfo, all_reduce_scalar, init_distributed, shutdown
from ..parallel.dp import ShardedAdamW, broadcast_parameters
from ..utils.schedules import CosineAnnealing
from . import store
from .adapters import LEVEL_NAMES, create_adapter
from .losses import acceptance_rates, cosine_per_position

HISTORY_KEYS = ("train_loss", "train_cos_sim", "val_loss", "val_cos_sim", "val_accept_80",
                "val_accept_85", "val_accept_90", "val_accept_95")


class HiddenAdapterTrainer:
    def __init__(self, model, train_loader, val_loader=None, learning_rate: float = 1e-3,
                 weight_decay: float = 0.01, device="cuda", vlm_only: bool = False,
                 fused: bool = False, num_epochs: int = 100, info: Optional[DistInfo] = None,
                 autocast: Optional[bool] = None, bucket_bytes: int = 64 << 20, intree_gemm: bool = True):
        self.intree_gemm, self.linear_mode = intree_gemm, None
        self.info = info or DistInfo(device=torch.device(device))
        self.device = torch.device(device)
        self.model = model.to(self.device)
        broadcast_parameters(self.model)
        self.train_loader, self.val_loader = train_loader, val_loader
        self.vlm_only, self.fused = vlm_only, fused or getattr(model, "takes_target_input", False)
        self.optimizer = ShardedAdamW(self.model.parameters(), lr=learning_rate,
                                      weight_decay=weight_decay, bucket_bytes=bucket_bytes)
        self.scheduler = CosineAnnealing(self.optimizer, t_max=num_epochs, eta_min=1e-6)
        self.autocast = (self.device.type == "cuda") if autocast is None else autocast
        self.history: Dict[str, list] = {k: [] for k in HISTORY_KEYS}
        self.best_val_loss = float("inf")
        self.patience = 0
        self.start_epoch = 1
        self.max_grad_norm = 1.0

    # ---------------------------------------------------------------- helpers
    def _batch(self, batch):
        nb = self.device.type == "cuda"
        tgt = batch["vl_hidden"].to(self.device, non_blocking=nb)
        src = tgt if self.vlm_only else batch["egpt_hidden"].to(self.device, non_blocking=nb)
        return src, tgt, batch["mask"].to(self.device, non_blocking=nb)

    def _ctx(self):
        """bf16 autocast on a GPU, with the adapter's ``nn.Linear`` layers on the in-tree GEMM
        (``ops/intree_mode.py``) unless ``intree_gemm`` is off."""
        if not self.autocast:
            return torch.autocast("cpu", enabled=False)
        stack = contextlib.ExitStack()
        stack.enter_context(torch.autocast("cuda", dtype=torch.bfloat16))
        if self.intree_gemm:
            from ..ops.intree_mode import InTreeLinear
            self.linear_mode = stack.enter_context(InTreeLinear())
        return stack

    def _mean_over_ranks(self, total: float, count: float) -> float:
        t = all_reduce_scalar(total, device=self.device)
        c = all_reduce_scalar(count, device=self.device)
        return t / max(c, 1.0)

    # ------------------------------------------------------------------ epochs
    def train_epoch(self, epoch: int) -> Dict[str, float]:
        self.model.train()
        if hasattr(self.train_loader, "set_epoch"):
            self.train_loader.set_epoch(epoch)
        tot_loss = tot_cos = 0.0
        n = 0
        for batch in self.train_loader:
            src, tgt, mask = self._batch(batch)
            self.optimizer.zero_grad()
            with self._ctx():
                losses = self.model.compute_loss(src, tgt, mask)
            losses["total_loss"].backward()
            self.optimizer.step(max_grad_norm=self.max_grad_norm)
            tot_loss += float(losses["total_loss"].detach())
            tot_cos += float(losses["cos_sim"].detach())
            n += 1
        self.scheduler.step()
        return {"loss": self._mean_over_ranks(tot_loss, n), "cos_sim": self._mean_over_ranks(tot_cos, n)}

    @torch.no_grad()
    def validate(self) -> Dict[str, float]:
        if self.val_loader is None:
            return {}
        self.model.eval()
        tot_loss = tot_cos = 0.0
        n = 0
        counts = torch.zeros(5, dtype=torch.float64)  # [valid, >.80, >.85, >.90, >.95]
        for batch in self.val_loader:
            src, tgt, mask = self._batch(batch)
            with self._ctx():
                pred = self.model.align(src, tgt, mask)
                from .losses import alignment_losses
                losses = alignment_losses(pred.float(), tgt, mask, self.model.prediction_weight)
            tot_loss += float(losses["total_loss"])
            tot_cos += float(losses["cos_sim"])
            n += 1
            cos = cosine_per_position(pred.float(), tgt)[mask > 0]
            counts += torch.tensor([cos.numel()] + [float((cos > t).sum()) for t in
                                                    (0.80, 0.85, 0.90, 0.95)], dtype=torch.float64)
        if dist.is_available() and dist.is_initialized():
            c = counts.to(self.device) if dist.get_backend() == "nccl" else counts
            dist.all_reduce(c)
            counts = c.cpu()
        valid = max(float(counts[0]), 1.0)
        out = {"loss": self._mean_over_ranks(tot_loss, n), "cos_sim": self._mean_over_ranks(tot_cos, n)}
        for i, k in enumerate(("accept_80", "accept_85", "accept_90", "accept_95")):
            out[k] = float(counts[i + 1]) / valid
        return out

    # ------------------------------------------------------------------- train
    def train(self, num_epochs: int, save_dir: str, early_stopping: int = 10,
              verbose: bool = True) -> Dict[str, list]:
        main = self.info.is_main
        if main:
            os.makedirs(save_dir, exist_ok=True)
        epoch = self.start_epoch - 1
        for epoch in range(self.start_epoch, num_epochs + 1):
            t0 = time.time()
            tm = self.train_epoch(epoch)
            self.history["train_loss"].append(tm["loss"])
            self.history["train_cos_sim"].append(tm["cos_sim"])
            msg = f"epoch {epoch}/{num_epochs} train loss {tm['loss']:.4f} cos {tm['cos_sim']:.4f}"
            stop = False
            if self.val_loader is not None:
                vm = self.validate()
                self.history["val_loss"].append(vm["loss"])
                self.history["val_cos_sim"].append(vm["cos_sim"])
                for t in (80, 85, 90, 95):
                    self.history[f"val_accept_{t}"].append(vm.get(f"accept_{t}", 0.0))
                msg += (f" | val loss {vm['loss']:.4f} cos {vm['cos_sim']:.4f} "
                        f"acc@.90 {vm['accept_90']:.2%}")
                if vm["loss"] < self.best_val_loss:   # same reduced value on every rank
                    self.best_val_loss, self.patience = vm["loss"], 0
                    if main:
                        self.model.save_checkpoint(os.path.join(save_dir, "best_model.pt"),
                                                   None, epoch, vm)
                else:
                    self.patience += 1
                stop = self.patience >= early_stopping
            if verbose and main:
                print(msg + f" ({time.time() - t0:.1f}s)", flush=True)
            self._save_resume(save_dir, epoch)
            if stop:
                if verbose and main:
                    print(f"early stopping at epoch {epoch}")
                break
        if main:
            self.model.save_checkpoint(os.path.join(save_dir, "final_model.pt"), None, epoch)
            self.save_history(save_dir)
            self.plot_curves(save_dir)
        return self.history

    # --------------------------------------------------------- resume / outputs
    def _save_resume(self, save_dir: str, epoch: int) -> None:
        state = {"optimizer": self.optimizer.state_dict(), "scheduler": self.scheduler.state_dict()}
        if self.info.is_main:
            state.update(model_state_dict=self.model.state_dict(), epoch=epoch,
                         history=self.history, best_val_loss=self.best_val_loss,
                         patience=self.patience)
        torch.save(state, os.path.join(save_dir, f"last_rank{self.info.rank}.pt"))

    def resume(self, save_dir: str) -> bool:
        path = os.path.join(save_dir, f"last_rank{self.info.rank}.pt")
        main_path = os.path.join(save_dir, "last_rank0.pt")
        if not (os.path.exists(path) and os.path.exists(main_path)):
            return False
        mine = torch.load(path, map_location="cpu", weights_only=True)
        main = mine if self.info.rank == 0 else torch.load(main_path, map_location="cpu",
                                                           weights_only=True)
        self.model.load_state_dict(main["model_state_dict"])
        # parameters are views of the optimizer's flat buffers, so loading in place
        # keeps them bound; the fp32 master shard comes from the optimizer state
        self.optimizer.load_state_dict(mine["optimizer"])
        self.scheduler.load_state_dict(mine["scheduler"])
        self.history = main["history"]
        self.best_val_loss, self.patience = main["best_val_loss"], main["patience"]
        self.start_epoch = main["epoch"] + 1
        return True

    def save_history(self, save_dir: str) -> None:
        with open(os.path.join(save_dir, "history.json"), "w") as f:
            json.dump(self.history, f, indent=2)

    def plot_curves(self, save_dir: str) -> None:
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:
            return
        fig, ax = plt.subplots(1, 3, figsize=(15, 4))
        ep = range(1, len(self.history["train_loss"]) + 1)
        ax[0].plot(ep, self.history["train_loss"], label="train")
        if self.history["val_loss"]:
            ax[0].plot(ep, self.history["val_loss"], label="val")
        ax[0].set_title("loss"); ax[0].legend()
        ax[1].plot(ep, self.history["train_cos_sim"], label="train")
        if self.history["val_cos_sim"]:
            ax[1].plot(ep, self.history["val_cos_sim"], label="val")
        ax[1].set_title("cosine similarity"); ax[1].legend()
        for t in (80, 85, 90, 95):
            if self.history[f"val_accept_{t}"]:
                ax[2].plot(ep, self.history[f"val_accept_{t}"], label=f"@0.{t}")
        ax[2].set_title("acceptance rate"); ax[2].legend()
        fig.tight_layout()
        fig.savefig(os.path.join(save_dir, "training_curves.png"), dpi=120)
        plt.close(fig)


# ------------------------------------------------------------------------- CLI
def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Train Hidden State Adapter")
    p.add_argument("--train_data", type=str, required=True,
                   help="hidden_states_train.pt or a chunked directory")
    p.add_argument("--val_data", type=str, default=None)
    p.add_argument("--val_split", type=float, default=0.1)
    p.add_argument("--output_dir", type=str, default="tasks")
    p.add_argument("--adapter_level", type=int, default=1, choices=[1, 2, 3, 4, 5, 6])
    p.add_argument("--hidden_dim", type=int, default=4096)
    p.add_argument("--bottleneck_dim", type=int, default=None)
    p.add_argument("--num_blocks", type=int, default=3)
    p.add_argument("--num_heads", type=int, default=8)
    p.add_argument("--num_layers", type=int, default=1)
    p.add_argument("--ffn_dim", type=int, default=2048)
    p.add_argument("--num_epochs", type=int, default=50)
    p.add_argument("--batch_size", type=int, default=64, help="per-rank batch size")
    p.add_argument("--learning_rate", type=float, default=1e-3)
    p.add_argument("--early_stopping", type=int, default=10)
    p.add_argument("--vlm_only", action="store_true",
                   help="B1 baseline: verifier states as both input and target")
    p.add_argument("--resume", type=str, default=None, help="task directory to continue")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--cpu", action="store_true")
    p.add_argument("--library_gemm", action="store_true",
                   help="leave the adapter's nn.Linear layers on the vendor GEMM library (default: in-tree GEMM)")
    return p


def _loaders(args, info: DistInfo):
    if os.path.isdir(args.train_data) and (args.val_data is None or os.path.isdir(args.val_data)) \
            and args.val_data is not None:
        dev = info.device if info.device.type == "cuda" else None
        return (store.ChunkedTrainLoader(args.train_data, args.batch_size, seed=args.seed,
                                         rank=info.rank, world=info.world, device=dev),
                store.ChunkedValLoader(args.val_data, args.batch_size, rank=info.rank,
                                       world=info.world, device=dev))
    data = store.load_hidden_states(args.train_data)
    n = data["egpt_hidden"].shape[0]
    g = torch.Generator().manual_seed(args.seed)
    perm = torch.randperm(n, generator=g).tolist()
    if args.val_data:
        train_idx, val = perm, store.HiddenStateDataset(store.load_hidden_states(args.val_data))
    else:
        k = max(int(n * args.val_split), 1)
        train_idx, val = perm[k:], store.HiddenStateDataset(data, perm[:k])
    train_idx = train_idx[info.rank::info.world]
    usable = (len(train_idx) // args.batch_size) * args.batch_size if info.world > 1 else len(train_idx)
    if store.DeviceBatchLoader.fits(data, info.device):
        # the whole set lives on the accelerator; a step uploads only its sample indices
        if args.val_data:
            vdata, vidx = {"egpt_hidden": val.e, "vl_hidden": val.v, "seq_lens": val.lens}, None
        else:
            vdata, vidx = data, val.indices
        return (store.DeviceBatchLoader(data, train_idx[:max(usable, 1)], args.batch_size, info.device,
                                        shuffle=True, seed=args.seed, drop_last=info.world > 1),
                store.DeviceBatchLoader(vdata, vidx, args.batch_size, info.device, shuffle=False))
    train = store.HiddenStateDataset(data, train_idx[:max(usable, 1)])
    dl = torch.utils.data.DataLoader
    return (dl(train, batch_size=args.batch_size, shuffle=True, collate_fn=store.collate_fn,
               drop_last=info.world > 1),
            dl(val, batch_size=args.batch_size, shuffle=False, collate_fn=store.collate_fn))


def main(argv=None) -> str:
    args = build_parser().parse_args(argv)
    info = init_distributed(use_gpu=False if args.cpu else None)
    torch.manual_seed(args.seed)
    train_loader, val_loader = _loaders(args, info)
    kw = dict(num_blocks=args.num_blocks, num_heads=args.num_heads, num_layers=args.num_layers,
              ffn_dim=args.ffn_dim)
    if args.bottleneck_dim is not None:
        kw["bottleneck_dim"] = args.bottleneck_dim
    if args.adapter_level == 3 and "num_blocks" in kw and args.num_blocks == 3:
        kw["num_blocks"] = 2
    model = create_adapter(args.adapter_level, args.hidden_dim, verbose=info.is_main, **kw)
    tag = "B1" if args.vlm_only else ("L5F" if args.adapter_level == 6 else f"L{args.adapter_level}")
    if args.resume:
        save_dir = args.resume
    else:
        stamp = datetime.now().strftime("%Y%m%d_%H%M%S")
        from ..parallel.dist import broadcast_object
        save_dir = broadcast_object(os.path.join(args.output_dir, tag, f"{tag}_{stamp}"))
    os.makedirs(save_dir, exist_ok=True)
    trainer = HiddenAdapterTrainer(model, train_loader, val_loader, args.learning_rate,
                                   device=info.device, vlm_only=args.vlm_only,
                                   num_epochs=args.num_epochs, info=info, intree_gemm=not args.library_gemm)
    if args.resume and trainer.resume(save_dir) and info.is_main:
        print(f"resumed from {save_dir} at epoch {trainer.start_epoch}")
    if info.is_main:
        cfg = dict(vars(args), adapter_type=LEVEL_NAMES[args.adapter_level],
                   num_parameters=model.get_num_parameters(), world_size=info.world,
                   global_batch=args.batch_size * info.world, tag=tag)
        with open(os.path.join(save_dir, "config.json"), "w") as f:
            json.dump(cfg, f, indent=2)
    trainer.train(args.num_epochs, save_dir, args.early_stopping)
    shutdown()
    return save_dir


if __name__ == "__main__":
    main()
