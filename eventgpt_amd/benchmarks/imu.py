"""5-stage timing of IMU-conditioned chat models on synthetic inertial data.

The reference times two third-party IMU MLLMs — LLaSA-7B (LIMU-BERT encoder, 72-wide,
sensor tokens spliced at ``<image>``) and OneLLM-7B (conv tokenizer + universal
encoder + 30 learned query tokens) — with the same load / preprocess / encode /
prefill / decode split as the event benchmark, falling back to synthetic timings when
the checkpoints are missing (feasible_imu/benchmark_inference_5stages.py:100-457,
benchmark_onellm_5stages.py:61-345).  Here both encoder styles are small modules of
this package in front of the framework's LLaMA decoder and serving engine, so the
benchmark always measures real work; weights are random unless a checkpoint directory
is given, and the data is synthetic of the reference's shapes (120x6 at 60 Hz;
2000x6 at 400 Hz).
"""
from __future__ import annotations

import argparse
import os
from datetime import datetime
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import nn

from ..align.token_align import CausalEncoder
from ..eval.reports import write_json
from ..eval.timers import StageTimer
from ..models.config import EventChatConfig
from ..models.video_llava import LlamaForCausalLM
from .stages import _finish, llm_prefill_decode, summarize_5stage

IMU_QUERY = "What activity is the person doing based on this motion sensor data?"


def generate_synthetic_imu_data(num_samples: int, seq_len: int = 120, feature_num: int = 6, seed: int = 0):
    """Walking-like accelerometer + gyroscope traces ``[seq_len, 6]`` float32 (2 s at 60 Hz)."""
    rng = np.random.default_rng(seed)
    t = np.linspace(0, 2, seq_len)
    out = []
    for _ in range(num_samples):
        n = lambda s: rng.standard_normal(seq_len) * s      # noqa: E731
        cols = [0.1 * np.sin(4 * np.pi * t) + n(0.05), 0.05 * np.sin(8 * np.pi * t + np.pi / 4) + n(0.05),
                9.8 + 0.3 * np.sin(4 * np.pi * t + np.pi / 2) + n(0.1), 0.1 * np.sin(2 * np.pi * t) + n(0.02),
                0.05 * np.cos(2 * np.pi * t) + n(0.02), 0.02 * np.sin(np.pi * t) + n(0.01)]
        out.append(np.stack(cols[:feature_num], 1).astype(np.float32))
    return out


def generate_synthetic_imu_ego4d(num_samples: int, duration_sec: float = 5.0, rate_hz: int = 400, seed: int = 0):
    """Ego4D-style clips: ``{"signal": [6, T] float32, "duration": s}`` at 400 Hz."""
    rng = np.random.default_rng(seed)
    n = int(duration_sec * rate_hz)
    t = np.arange(n) / rate_hz
    out = []
    for _ in range(num_samples):
        f = rng.uniform(0.5, 3.0, 6)[:, None]
        sig = np.sin(2 * np.pi * f * t[None]) * rng.uniform(0.05, 0.5, 6)[:, None] + rng.standard_normal((6, n)) * 0.05
        sig[2] += 9.8
        out.append({"signal": sig.astype(np.float32), "duration": duration_sec})
    return out


def normalize_imu_data(imu: np.ndarray) -> np.ndarray:
    """Per-feature z-score over time, ``[T, F]``."""
    return (imu - imu.mean(0, keepdims=True)) / (imu.std(0, keepdims=True) + 1e-8)


def pad_imu_signal(signal: torch.Tensor, length: int = 2000) -> torch.Tensor:
    """``[6, T]`` -> ``[6, length]`` (zero-pad / truncate)."""
    if signal.shape[1] >= length:
        return signal[:, :length]
    return torch.nn.functional.pad(signal, (0, length - signal.shape[1]))


class LimuBertEncoder(nn.Module):
    """Per-timestep embedding (6 -> 72) + position + norm, 4 shared-width attention blocks."""

    def __init__(self, feature_num: int = 6, hidden: int = 72, layers: int = 4, heads: int = 4, seq_len: int = 120):
        super().__init__()
        self.embed, self.pos = nn.Linear(feature_num, hidden), nn.Embedding(seq_len, hidden)
        self.norm = nn.LayerNorm(hidden)
        self.blocks = CausalEncoder(hidden, heads, hidden * 2, layers, dropout=0.0)

    def forward(self, x):                               # [B, T, F] -> [B, T, hidden]
        h = self.norm(self.embed(x) + self.pos(torch.arange(x.shape[1], device=x.device))[None])
        return self.blocks(h, causal=False)


class UniversalImuEncoder(nn.Module):
    """Conv1d tokenizer (kernel 10, stride 10) -> transformer -> ``n_query`` learned queries
    attend over the tokens (one cross-attention resampler)."""

    def __init__(self, hidden: int = 1024, layers: int = 4, heads: int = 16, n_query: int = 30, in_ch: int = 6):
        super().__init__()
        self.tokenizer = nn.Conv1d(in_ch, hidden, kernel_size=10, stride=10)
        self.blocks = CausalEncoder(hidden, heads, hidden * 4, layers, dropout=0.0, pre_norm=True)
        self.query = nn.Parameter(torch.randn(n_query, hidden) * 0.02)
        self.attn = nn.MultiheadAttention(hidden, heads, batch_first=True)
        self.norm = nn.LayerNorm(hidden)

    def forward(self, x):                               # [B, 6, T] -> [B, n_query, hidden]
        tok = self.blocks(self.tokenizer(x).transpose(1, 2), causal=False)
        q = self.query[None].expand(x.shape[0], -1, -1)
        return self.norm(self.attn(q, tok, tok, need_weights=False)[0])


class ImuChatModel(nn.Module):
    """IMU encoder -> 2-layer projector -> LLaMA; sensor tokens replace the ``<image>`` placeholder."""

    def __init__(self, text_config: EventChatConfig, style: str = "llasa", encoder_hidden: Optional[int] = None,
                 encoder_layers: int = 4):
        super().__init__()
        self.style = style
        if style == "llasa":
            eh = encoder_hidden or 72
            self.encoder = LimuBertEncoder(hidden=eh, layers=encoder_layers, heads=4)
        elif style == "onellm":
            eh = encoder_hidden or 1024
            self.encoder = UniversalImuEncoder(hidden=eh, layers=encoder_layers, heads=max(eh // 64, 1))
        else:
            raise ValueError(f"unknown IMU model style {style}")
        h = text_config.hidden_size
        self.projector = nn.Sequential(nn.Linear(eh, h), nn.GELU(), nn.Linear(h, h))
        self.language_model = LlamaForCausalLM(text_config)
        self.config = text_config

    @property
    def device(self):
        return self.language_model.device

    @property
    def dtype(self):
        return self.language_model.dtype

    def engine(self, **kw):
        return self.language_model.engine(**kw)

    @torch.no_grad()
    def encode_imu(self, x: torch.Tensor) -> torch.Tensor:
        return self.projector(self.encoder(x.to(self.device, self.dtype)))

    @torch.no_grad()
    def build_embeds(self, ids_before: torch.Tensor, feats: torch.Tensor, ids_after: torch.Tensor) -> torch.Tensor:
        emb = self.language_model.model.embed_tokens
        return torch.cat([emb(ids_before.to(self.device)), feats[0].to(emb.weight.dtype),
                          emb(ids_after.to(self.device))])

    @classmethod
    def random_init(cls, style: str, size: str = "7b", device="cpu", dtype=None, seed: int = 0) -> "ImuChatModel":
        if size == "7b":
            text = EventChatConfig(vocab_size=32000)
            kw = {}
        else:
            text = EventChatConfig(vocab_size=320, hidden_size=64, intermediate_size=160, num_hidden_layers=2,
                                   num_attention_heads=4, max_position_embeddings=512, torch_dtype="float32")
            kw = dict(encoder_hidden=24 if style == "llasa" else 64, encoder_layers=1)
        dtype = dtype or (torch.bfloat16 if str(device).startswith("cuda") else torch.float32)
        torch.manual_seed(seed)
        with torch.device(device):
            m = cls(text, style, **kw)
        for p in m.parameters():
            if p.ndim >= 2:
                nn.init.normal_(p, std=0.02)
        return m.to(dtype).eval()


def run_imu_5stage_benchmark(model: ImuChatModel, tokenizer, samples: List, max_new_tokens: int = 64,
                             warmup: int = 1) -> List[Dict]:
    results = []
    head, tail = f"USER: ", f"\n{IMU_QUERY}\nASSISTANT:"
    for i, raw in enumerate(samples):
        timer = StageTimer(model.device)
        with timer.stage("stage1_time", host=True):
            data = raw["signal"].copy() if isinstance(raw, dict) else raw.copy()
        with timer.stage("stage2_time", host=True):
            if model.style == "llasa":
                x = torch.from_numpy(normalize_imu_data(data).astype(np.float32))[None]
            else:
                x = pad_imu_signal(torch.from_numpy(data))[None]
            before = torch.tensor(tokenizer(head).input_ids, dtype=torch.long)
            after = torch.tensor(tokenizer(tail, add_special_tokens=False).input_ids, dtype=torch.long)
        with timer.stage("stage3_time"):
            feats = model.encode_imu(x)
            emb = model.build_embeds(before, feats, after)
        toks, kv_mb = llm_prefill_decode(model, emb, max_new_tokens, model.config.eos_token_id, timer)
        rec = _finish(timer, toks, emb.shape[0], kv_mb, tokenizer, {"imu_input": list(x.shape),
                                                                   "imu_tokens": list(feats.shape)})
        rec["sample"] = i
        if i >= warmup:
            results.append(rec)
    return results


def main(argv=None) -> Dict:
    from ..data.tokenizer import ByteTokenizer
    p = argparse.ArgumentParser(description="5-stage benchmark of IMU chat models on synthetic data")
    p.add_argument("--style", default="both", choices=["llasa", "onellm", "both"])
    p.add_argument("--size", default="7b", choices=["7b", "tiny"])
    p.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    p.add_argument("--max_samples", type=int, default=20)
    p.add_argument("--max_new_tokens", type=int, default=64)
    p.add_argument("--imu_seq_len", type=int, default=120)
    p.add_argument("--duration_sec", type=float, default=5.0)
    p.add_argument("--output_dir", default=None)
    p.add_argument("--output_json", default=None)
    args = p.parse_args(argv)
    tok = ByteTokenizer(32000 if args.size == "7b" else 320)
    out: Dict = {"timestamp": datetime.now().strftime("%Y%m%d_%H%M%S"), "data": "synthetic", "weights": "random",
                 "size": args.size, "summaries": {}, "samples": {}}
    for style in (("llasa", "onellm") if args.style == "both" else (args.style,)):
        model = ImuChatModel.random_init(style, args.size, args.device)
        data = generate_synthetic_imu_data(args.max_samples + 1, args.imu_seq_len) if style == "llasa" else \
            generate_synthetic_imu_ego4d(args.max_samples + 1, args.duration_sec)
        res = run_imu_5stage_benchmark(model, tok, data, args.max_new_tokens)
        name = "LLaSA-shaped" if style == "llasa" else "OneLLM-shaped"
        out["summaries"][name] = summarize_5stage(res, name)
        out["samples"][name] = res
        del model
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    path = args.output_json or os.path.join(args.output_dir or os.path.join(os.getcwd(), "tasks"),
                                            f"imu_5stages_{out['timestamp']}.json")
    write_json(out, path)
    print(f"[imu] wrote {path}")
    return out


if __name__ == "__main__":
    main()
