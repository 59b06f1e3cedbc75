"""Additional measurements of a multi-GPU ``bench.py`` run (never its headline value).

* ``dp_train`` -- one instruction-tuning step of the 7B model per rank, gradients reduce-scattered
  into the sharded AdamW (``parallel/dp.py``) over the run's own process group: the projector stage
  and the LoRA stage of ``cli/train_sft.py``, each with the collective launched from the gradient
  hooks (on the communication stream, under the rest of backward) and again after backward, so the
  difference is the communication time the overlap hides.  The loops it parallelises in the
  reference are single-process: pipeline/adapter_train/train_lora_adapter.py:594-764,
  train_hidden_adapter.py:320-359.
* ``sd_two_gpu`` -- drafter and verifier on two devices (rank 0 drives its own GPU and rank 1's; the
  other ranks wait): prefill hiding end to end -- drafter prefill + free-running decode while the
  verifier prefills, a peer copy of <= 64 token ids, one batched verify, autoregressive rest --
  next to the same run with both models on one GPU and two streams.  The reference draws GPU0 /
  GPU1 (pipeline/benchmark_e2e/README.md:25-42) and runs everything on one device.

Every function returns a plain dict for the JSON line and raises nothing the caller has to know
about: ``bench.py`` wraps each in try / except and reports the error under the extra's key.
"""
from __future__ import annotations

import tempfile
import time
from typing import Dict

import torch


def _sync_ranks(dist, dev) -> None:
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)


def dp_train_extra(model_kind: str, dist, dev, world: int, rank: int, steps: int = 8, warmup: int = 3) -> Dict:
    """ms per step and samples / s / GPU of the projector and LoRA tuning steps, data parallel."""
    from ..cli.train_sft import SFTTrainer
    from ..data.tokenizer import ByteTokenizer
    from ..models import ClipVisionConfig, EventChatConfig, EventChatModel
    from ..parallel.dist import DistInfo
    from ..parallel.dp import ShardedAdamW
    from ..pipeline.dataset import InstructionDataset, make_synthetic_dataset

    if model_kind == "7b":
        cfg = EventChatConfig.eventgpt_7b()
        ev = dict(n_events=60000, width=640, height=480)
    else:
        cfg = EventChatConfig(vocab_size=32003, hidden_size=1024, intermediate_size=2816, num_hidden_layers=4,
                              num_attention_heads=8, mm_visual_tower="small", event_feature_adaptor=True)
        cfg.vision_config = ClipVisionConfig(hidden_size=256, intermediate_size=1024, num_hidden_layers=2,
                                             num_attention_heads=4)
        ev = dict(n_events=20000, width=160, height=120)
    backend = "none" if dist is None else dist.get_backend()
    info = DistInfo(rank, world, dev.index or 0, backend, dev)
    model = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=0)
    tok = ByteTokenizer(cfg.vocab_size)
    n = steps + warmup
    ds = InstructionDataset(make_synthetic_dataset(tempfile.mkdtemp(prefix=f"evgpt_dpx{rank}_"), n_samples=n,
                                                   seed=100 * rank, **ev))
    out: Dict = {"global_batch": world, "steps": steps, "backend": backend}
    for tune in ("projector", "lora"):
        tr = SFTTrainer(model, tok, ds, info, tune=tune, output_dir=tempfile.mkdtemp(prefix="evgpt_dpx_out_"),
                        broadcast=False)        # every rank built the model from the same seed
        model.train()
        row = {"trainable_parameters": sum(p.numel() for p in tr.params)}
        for label, overlap in (("overlap", True), ("no_overlap", False)):
            if overlap and not (dist is not None and backend == "nccl"):
                continue        # hooks + communication stream exist on an RCCL group only
            tr.opt.detach()
            tr.opt = ShardedAdamW(tr.params, lr=1e-4, weight_decay=0.0, overlap=overlap)
            tr.opt.zero_grad()
            fb = st = 0.0
            for i in range(n):
                prep = tr.prepare_sample(ds[i])
                if i == warmup:
                    _sync_ranks(dist, dev)
                    t_all = time.perf_counter()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                loss = tr.sample_loss(ds[i], prep)
                loss.backward()
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                tr.opt.step(max_grad_norm=1.0)
                tr.opt.zero_grad()
                torch.cuda.synchronize(dev)
                t2 = time.perf_counter()
                if i >= warmup:
                    fb, st = fb + (t1 - t0), st + (t2 - t1)
            _sync_ranks(dist, dev)
            dt = time.perf_counter() - t_all
            row[label] = {"ms_per_step": round(dt / steps * 1e3, 2),
                          "forward_backward_ms": round(fb / steps * 1e3, 2),
                          "reduce_scatter_adamw_all_gather_ms": round(st / steps * 1e3, 2),
                          "samples_per_s_per_gpu": round(steps / dt, 2), "loss": round(float(loss.detach()), 4)}
        if "overlap" in row:
            row["comm_hidden_ms"] = round(row["no_overlap"]["ms_per_step"] - row["overlap"]["ms_per_step"], 2)
        out[tune] = row
        tr.opt.detach()
        if tr.dgrad_copy_bytes:
            from ..ops.differentiable import release_transposed_dgrad
            release_transposed_dgrad(model)
    return out


def sd_two_gpu_extra(model_kind: str, dev0: torch.device, dev1: torch.device, new_tokens: int = 50,
                     drafter_prompt: int = 636, verifier_prompt: int = 2115, max_free: int = 30) -> Dict:
    """Prefill hiding end to end with the verifier on ``dev1`` and the drafter on ``dev0``, and with
    both on ``dev0``.  ``dev1 == dev0`` (a one-GPU rehearsal) runs the one-device placement only."""
    from ..models import EventChatConfig, EventChatModel
    from ..runtime.scheduler import DualModelScheduler

    if model_kind == "7b":
        cfg = EventChatConfig(vocab_size=32003, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                              num_attention_heads=32, max_position_embeddings=4096, torch_dtype="bfloat16")
    else:
        cfg = EventChatConfig(vocab_size=32003, hidden_size=1024, intermediate_size=2816, num_hidden_layers=4,
                              num_attention_heads=8, max_position_embeddings=4096, torch_dtype="bfloat16")
    H = cfg.hidden_size
    g = torch.Generator(device="cpu").manual_seed(3)
    d_emb = (torch.randn(drafter_prompt, H, generator=g) * 0.02).to(torch.bfloat16)
    v_emb = (torch.randn(verifier_prompt, H, generator=g) * 0.02).to(torch.bfloat16)
    out: Dict = {"new_tokens": new_tokens, "drafter_prompt": drafter_prompt, "verifier_prompt": verifier_prompt}
    with torch.cuda.device(dev0):
        drafter = EventChatModel.from_config(cfg, torch.bfloat16, dev0, seed=21)
        d_eng = drafter.engine(max_len=4096, max_new_tokens=new_tokens + 80)
    placements = [("one_gpu_two_streams", dev0)] + ([("two_gpus", dev1)] if dev1 != dev0 else [])
    for label, vdev in placements:
        with torch.cuda.device(vdev):
            verifier = EventChatModel.from_config(cfg, torch.bfloat16, vdev, seed=22)
            v_eng = verifier.engine(max_len=4096, max_new_tokens=new_tokens + 80)
        sch = DualModelScheduler(d_eng, v_eng)
        de, ve = d_emb.to(dev0), v_emb.to(vdev)
        best = None
        for rep in range(3):                      # the first repetition captures the graphs
            base = sch.verifier_baseline(ve, new_tokens, -1)
            res = sch.prefill_hiding(de, ve, max_new_tokens=new_tokens, max_free=max_free, eos_token_id=-1)
            if rep and (best is None or res["total_ms"] < best[1]["total_ms"]):
                best = (base, res)
        base, res = best
        n = min(len(res["tokens"]), len(base["tokens"]))
        out[label] = {"verifier_only_ms": round(base["total_ms"], 2), "prefill_hiding_ms": round(res["total_ms"], 2),
                      "speedup": round(base["total_ms"] / res["total_ms"], 3), "free_tokens": int(res["n_free"]),
                      "accepted": int(res["stats"]["total_accepted"]),
                      "prefill_wall_ms": round(res["prefill_wall_ms"], 2),
                      "drafter_prefill_ms": round(res["egpt_prefill_ms"], 2),
                      "verifier_prefill_ms": round(res["vl_prefill_ms"], 2),
                      "same_tokens_as_verifier": bool(torch.equal(res["tokens"][:n].cpu(), base["tokens"][:n].cpu())),
                      "devices": [str(dev0), str(vdev)]}
        del sch, v_eng, verifier
        torch.cuda.empty_cache()
    return out
