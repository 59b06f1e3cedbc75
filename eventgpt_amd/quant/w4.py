"""Block-wise 4-bit weight quantisation with a 16-entry codebook.

The reference loads both 7B models through bitsandbytes ``load_in_4bit`` with
``bnb_4bit_quant_type="nf4"`` and double quantisation
(pipeline/feature_extraction/extract_hidden_states.py:117-122,
pipeline/benchmark_e2e/benchmark_e2e_wallclock.py:292-329): every linear weight is
cut into blocks of 64 values, each block is divided by its absolute maximum and
every value is replaced by the index of the nearest of 16 fixed levels; the fp32
block maxima are themselves quantised to 8 bits in groups of 256.

Here the same scheme is a storage format of the serving engine:

* ``codes``: the 4-bit indices, two per byte, laid out for the HIP kernels (see
  :func:`pack_codes`);
* ``absmax``: one fp32 scale per 64 weights, ``[N, K/64]``;
* ``code``: the 16 levels, rounded to bf16 (the kernels look them up as bf16 and
  accumulate in fp32, and the oracle below uses the same rounded table).

Double quantisation is kept as an on-disk option (:func:`quantize_absmax`): the
engine expands the 8-bit maxima to fp32 once at load, because 0.4 GB of scales is
nothing next to 288 GB of HBM while a second table lookup per block is not free.
The 8-bit code for the maxima is a plain linear one (offset + scale per group of
256); bitsandbytes uses its dynamic 8-bit map there, so files are not
interchangeable with it — checkpoints are quantised at load from bf16/fp16 weights
in both stacks.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, Optional, Tuple

import torch

BLOCK = 64

# The 16 NF4 levels: quantiles of a standard normal scaled to [-1, 1], with an exact
# zero (Dettmers et al., QLoRA, 2023, appendix E).
NF4_CODE = (-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
            -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
            0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
            0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0)
# Sign/2-bit exponent/1-bit mantissa 4-bit floats, normalised to [-1, 1] (the other
# `bnb_4bit_quant_type`).
FP4_CODE = (0.0, 0.0052083333, 0.6666666667, 1.0, 0.3333333333, 0.5, 0.1666666667, 0.25,
            -0.0, -0.0052083333, -0.6666666667, -1.0, -0.3333333333, -0.5, -0.1666666667, -0.25)


def codebook(kind: str = "nf4") -> torch.Tensor:
    """The 16 levels as a host bf16 tensor (what the kernels and the oracle look up)."""
    table = {"nf4": NF4_CODE, "fp4": FP4_CODE}.get(kind)
    if table is None:
        raise ValueError(f"unknown 4-bit codebook {kind!r} (nf4 or fp4)")
    return torch.tensor(table, dtype=torch.float32).to(torch.bfloat16)


def pack_codes(idx: torch.Tensor) -> torch.Tensor:
    """``[N, K]`` uint8 indices (0..15) -> ``[N, K/2]`` bytes in kernel order.

    Every 8 consecutive weights share one little-endian 32-bit word: byte ``j`` of the
    word holds weight ``j`` in its low nibble and weight ``4+j`` in its high nibble, so
    one mask and one shift split a word into two registers of four byte-sized indices.
    """
    n, k = idx.shape
    if k % 8:
        raise ValueError("K must be a multiple of 8")
    g = idx.reshape(n, k // 8, 2, 4).to(torch.uint8)
    return (g[:, :, 0] | (g[:, :, 1] << 4)).reshape(n, k // 2).contiguous()


def unpack_codes(packed: torch.Tensor) -> torch.Tensor:
    n, kb = packed.shape
    b = packed.reshape(n, kb // 4, 4)
    return torch.stack([b & 0xF, b >> 4], dim=2).reshape(n, kb * 2)


def quantize_absmax(absmax: torch.Tensor, group: int = 256) -> Dict[str, torch.Tensor]:
    """Second-level (double) quantisation of the block maxima to 8 bits per block."""
    flat = absmax.float().reshape(-1)
    pad = (-flat.numel()) % group
    v = torch.cat([flat, flat.new_zeros(pad)]).reshape(-1, group)
    lo = v.min(dim=1, keepdim=True).values
    scale = ((v.max(dim=1, keepdim=True).values - lo) / 255.0).clamp_min(1e-12)
    q = torch.round((v - lo) / scale).clamp_(0, 255).to(torch.uint8)
    return {"q": q.reshape(-1)[: flat.numel()].reshape(absmax.shape), "offset": lo.reshape(-1),
            "scale": scale.reshape(-1)}


def dequantize_absmax(state: Dict[str, torch.Tensor], group: int = 256) -> torch.Tensor:
    q = state["q"]
    flat = q.reshape(-1).float()
    gi = torch.arange(flat.numel(), device=q.device) // group
    return (flat * state["scale"].to(q.device)[gi] + state["offset"].to(q.device)[gi]).reshape(q.shape)


@dataclasses.dataclass
class W4Weight:
    """One ``[N, K]`` linear weight in the 4-bit format."""
    packed: torch.Tensor            # uint8 [N, K/2]
    absmax: torch.Tensor            # fp32  [N, K/64]
    code: torch.Tensor              # host bf16 [16]
    kind: str = "nf4"

    @property
    def shape(self) -> Tuple[int, int]:
        return self.packed.shape[0], self.packed.shape[1] * 2

    @property
    def nbytes(self) -> int:
        return self.packed.numel() + 4 * self.absmax.numel()

    def dequantize(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """bf16 ``[N, K]``: ``round_bf16(code[idx] * absmax)``.  On a GPU this is the HIP
        kernel; on the CPU the plain-torch oracle of the same arithmetic."""
        if self.packed.is_cuda:
            from ..ops import hip
            return hip.raw.w4_dequant(self.packed, self.absmax, self.code, out)
        n, k = self.shape
        vals = self.code.float()[unpack_codes(self.packed).long()]
        w = (vals.reshape(n, k // BLOCK, BLOCK) * self.absmax.reshape(n, k // BLOCK, 1)).reshape(n, k)
        w = w.to(torch.bfloat16)
        if out is not None:
            out.copy_(w)
            return out
        return w

    def matmul_reference(self, x: torch.Tensor) -> torch.Tensor:
        """fp32 ``x @ Wᵀ`` with the decode kernels' arithmetic (levels × x summed per
        32-weight half block, then one multiply by the block scale)."""
        n, k = self.shape
        vals = self.code.float().to(x.device)[unpack_codes(self.packed).long()]
        w = (vals.reshape(n, k // BLOCK, BLOCK) * self.absmax.reshape(n, k // BLOCK, 1)).reshape(n, k)
        return x.float() @ w.t()

    def to(self, device) -> "W4Weight":
        return W4Weight(self.packed.to(device), self.absmax.to(device), self.code, self.kind)

    def state_dict(self, double_quant: bool = True) -> Dict[str, torch.Tensor]:
        sd = {"packed": self.packed.cpu(), "code": self.code.float(), "kind": self.kind}
        if double_quant:
            sd.update({f"absmax_{k}": v.cpu() for k, v in quantize_absmax(self.absmax).items()})
        else:
            sd["absmax"] = self.absmax.cpu()
        return sd

    @classmethod
    def from_state_dict(cls, sd: Dict[str, torch.Tensor], device="cpu") -> "W4Weight":
        if "absmax" in sd:
            absmax = sd["absmax"].float()
        else:
            absmax = dequantize_absmax({k[len("absmax_"):]: v for k, v in sd.items() if k.startswith("absmax_")})
        return cls(sd["packed"].to(device), absmax.to(device).contiguous(),
                   sd["code"].to(torch.bfloat16).cpu(), str(sd.get("kind", "nf4")))


def quantize_weight(w: torch.Tensor, kind: str = "nf4", rows_per_pass: int = 4096) -> W4Weight:
    """Quantise ``w [N, K]`` (``K % 64 == 0``) on its own device, a slab of rows at a time
    (the nearest-level search materialises 16 distances per weight)."""
    n, k = w.shape
    if k % BLOCK:
        raise ValueError(f"K={k} is not a multiple of the {BLOCK}-weight block")
    code = codebook(kind)
    levels = code.float().to(w.device)
    packed = torch.empty(n, k // 2, dtype=torch.uint8, device=w.device)
    absmax = torch.empty(n, k // BLOCK, dtype=torch.float32, device=w.device)
    for r0 in range(0, n, rows_per_pass):
        blk = w[r0:r0 + rows_per_pass].float().reshape(-1, k // BLOCK, BLOCK)
        am = blk.abs().amax(dim=-1, keepdim=True).clamp_min(1e-12)
        idx = ((blk / am).unsqueeze(-1) - levels).abs().argmin(dim=-1).to(torch.uint8)
        packed[r0:r0 + rows_per_pass] = pack_codes(idx.reshape(-1, k))
        absmax[r0:r0 + rows_per_pass] = am.squeeze(-1)
    return W4Weight(packed, absmax, code, kind)


DECODER_LINEARS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


@torch.no_grad()
def fake_quantize_model_(model, kind: str = "nf4") -> int:
    """Replace every decoder-layer projection of ``model`` by its quantise→dequantise image
    (in place, any device).  The module path then computes what the 4-bit engine computes up
    to rounding order, which makes it the accuracy oracle for the engine and a way to study
    4-bit quality on the CPU.  Returns the number of matrices rewritten."""
    n = 0
    for lyr in model.model.layers:
        for holder in (lyr.self_attn, lyr.mlp):
            for name in DECODER_LINEARS:
                lin = getattr(holder, name, None)
                if lin is None:
                    continue
                w = lin.weight.data
                lin.weight.data.copy_(quantize_weight(w, kind).dequantize().to(w.dtype))
                n += 1
    return n
