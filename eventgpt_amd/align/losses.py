"""Losses for hidden-state alignment.

One implementation shared by every adapter level (the reference repeats it per
class, e.g. pipeline/adapter_train/hidden_adapter.py:164-215, 831-898):

    align  = masked MSE(pred, target) + 0.5 * (1 - masked mean cosine)
    pred   = masked MSE(pred[:, :-1], target[:, 1:])            (next-position)
    total  = (1 - w) * align + w * pred                          (w = 0 for L1-L4)

Masks are ``[B, S]`` with 1 for valid positions.  The MSE is averaged over valid
*elements* (positions x hidden), the cosine over valid positions.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F


def masked_mse(pred: torch.Tensor, target: torch.Tensor,
               mask: Optional[torch.Tensor]) -> torch.Tensor:
    se = (pred - target).pow(2)
    if mask is None:
        return se.mean()
    m = mask.to(se.dtype).unsqueeze(-1)
    return (se * m).sum() / (m.sum() * se.shape[-1]).clamp_min(1.0)


def cosine_per_position(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """``[B, S]`` cosine similarity between matching rows."""
    return (F.normalize(pred, dim=-1) * F.normalize(target, dim=-1)).sum(dim=-1)


def masked_mean(values: torch.Tensor, mask: Optional[torch.Tensor]) -> torch.Tensor:
    if mask is None:
        return values.mean()
    m = mask.to(values.dtype)
    return (values * m).sum() / m.sum().clamp_min(1.0)


class _RowPairLoss(torch.autograd.Function):
    """``(masked MSE, masked mean cosine)`` of ``pred`` vs ``target`` on the GPU in one pass each way.

    Both are functions of four sums per row -- ``p.t, |p|^2, |t|^2, |p-t|^2`` -- which one kernel
    produces (``row_stats``); the gradient is ``A_r (p - t) + B_r t + C_r p`` with per-row
    coefficients, which one kernel applies (``row_affine_grad``).  Autograd through the tensor
    expressions above writes and re-reads about a dozen full-size temporaries, a third of an
    L1 adapter's training step.  ``target`` gets no gradient (it is data)."""

    EPS = 1e-12      # F.normalize's clamp on the norms

    @staticmethod
    def forward(ctx, pred, target, mask):
        from ..ops import hip
        h = pred.shape[-1]
        st = hip.raw.row_stats(pred, target)                       # [rows, 4]
        m = torch.ones(st.shape[0], device=st.device) if mask is None else mask.reshape(-1).to(torch.float32)
        n_valid = m.sum().clamp_min(1.0)
        dot, na, nb, se = st.unbind(1)
        pn, tn = na.sqrt().clamp_min(_RowPairLoss.EPS), nb.sqrt().clamp_min(_RowPairLoss.EPS)
        cos = dot / (pn * tn)
        ctx.save_for_backward(pred, target, m, n_valid, pn, tn, cos)
        ctx.h = h
        return (se * m).sum() / (n_valid * h), (cos * m).sum() / n_valid

    @staticmethod
    def backward(ctx, g_mse, g_cos):
        from ..ops import hip
        pred, target, m, n_valid, pn, tn, cos = ctx.saved_tensors
        w = m / n_valid
        a = g_mse * 2.0 * w / ctx.h                                # d mse / d p = 2 (p - t) m / (n H)
        b = g_cos * w / (pn * tn)                                  # d cos / d p = t / (|p||t|) - cos p / |p|^2
        c = -g_cos * w * cos / (pn * pn)
        coef = torch.stack([a, b, c], dim=1).contiguous().float()
        return hip.raw.row_affine_grad(pred, target, coef), None, None


def _fused_pair_ok(pred: torch.Tensor, target: torch.Tensor) -> bool:
    return (pred.is_cuda and target.is_cuda and pred.dtype == target.dtype and pred.shape == target.shape
            and pred.dtype in (torch.float32, torch.bfloat16) and pred.shape[-1] % 8 == 0
            and pred.is_contiguous() and target.is_contiguous() and not target.requires_grad)


def alignment_losses(pred: torch.Tensor, target: torch.Tensor,
                     mask: Optional[torch.Tensor] = None,
                     prediction_weight: float = 0.0) -> Dict[str, torch.Tensor]:
    if _fused_pair_ok(pred, target):
        mse, cos_sim = _RowPairLoss.apply(pred, target, mask)
    else:
        mse = masked_mse(pred, target, mask)
        cos_sim = masked_mean(cosine_per_position(pred, target), mask)
    cos_loss = 1.0 - cos_sim
    align = mse + 0.5 * cos_loss
    out = {"mse_loss": mse, "cos_loss": cos_loss, "cos_sim": cos_sim}
    if prediction_weight > 0.0:
        if pred.shape[1] > 1:
            pred_loss = masked_mse(pred[:, :-1], target[:, 1:],
                                   None if mask is None else mask[:, 1:])
        else:
            pred_loss = pred.new_zeros(())
        out["pred_loss"] = pred_loss
        out["align_loss"] = align
        out["total_loss"] = (1.0 - prediction_weight) * align + prediction_weight * pred_loss
    else:
        out["total_loss"] = align
    return out


def acceptance_rates(cos: torch.Tensor, thresholds=(0.80, 0.85, 0.90, 0.95)) -> Dict[str, float]:
    """Fraction of positions whose cosine exceeds each threshold (keys ``accept_80`` ...)."""
    cos = cos.float()
    if cos.numel() == 0:
        return {f"accept_{int(round(t * 100))}": 0.0 for t in thresholds}
    return {f"accept_{int(round(t * 100))}": float((cos > t).float().mean()) for t in thresholds}


_BF16_COPIES = {}


def _bf16_copy(w: torch.Tensor) -> torch.Tensor:
    """A frozen matrix in bf16, converted once (keyed by storage and version)."""
    if w.dtype == torch.bfloat16 and w.is_contiguous():
        return w
    key = (w.data_ptr(), w._version, tuple(w.shape))
    hit = _BF16_COPIES.get("w")
    if hit is None or hit[0] != key:
        _BF16_COPIES["w"] = (key, w.detach().to(torch.bfloat16).contiguous())
    return _BF16_COPIES["w"][1]


def lora_distill_loss(pred_hidden: torch.Tensor, target_hidden: torch.Tensor,
                      lm_head_weight: torch.Tensor, cos_weight: float = 0.5,
                      ce_weight: float = 0.1) -> Dict[str, torch.Tensor]:
    """``MSE + 0.5*CosineEmbedding + 0.1*CE(lm_head(pred), argmax lm_head(target))``
    (reference: pipeline/adapter_train/train_lora_adapter.py:692-707)."""
    p = pred_hidden.float().reshape(-1, pred_hidden.shape[-1])
    t = target_hidden.float().reshape(-1, target_hidden.shape[-1])
    mse = F.mse_loss(p, t)
    cos = F.cosine_embedding_loss(p, t, torch.ones(p.shape[0], device=p.device))
    from ..ops import differentiable as diff
    if pred_hidden.is_cuda and pred_hidden.dtype == torch.bfloat16 and lm_head_weight.shape[1] % 64 == 0:
        # token term on the in-tree kernels: target tokens from the GEMM's argmax epilogue, the loss from
        # bf16 logits without an fp32 [rows, vocab] tensor (``ops.differentiable.linear_cross_entropy``)
        wb = _bf16_copy(lm_head_weight)
        ph = pred_hidden.reshape(-1, pred_hidden.shape[-1])
        if diff.linear_cross_entropy_ok(ph, wb):
            from ..ops import hip
            with torch.no_grad():
                pv, pi = hip.raw.linear_argmax_partials(target_hidden.reshape(-1, ph.shape[-1]).to(torch.bfloat16)
                                                        .contiguous(), wb, None)
                tgt_tok = pi.gather(1, pv.argmax(1, keepdim=True))[:, 0]
            ce = diff.linear_cross_entropy(ph, wb, tgt_tok.long(), -100, None)
            return {"total_loss": mse + cos_weight * cos + ce_weight * ce, "mse_loss": mse,
                    "cos_loss": cos, "ce_loss": ce, "cos_sim": cosine_per_position(p, t).mean()}
    w = lm_head_weight.float()
    with torch.no_grad():
        tgt_tok = (t @ w.t()).argmax(-1)
    ce = F.cross_entropy(p @ w.t(), tgt_tok)
    return {"total_loss": mse + cos_weight * cos + ce_weight * ce, "mse_loss": mse,
            "cos_loss": cos, "ce_loss": ce,
            "cos_sim": cosine_per_position(p, t).mean()}
