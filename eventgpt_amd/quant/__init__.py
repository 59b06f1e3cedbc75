"""Weight-only 4-bit quantisation (NF4 / FP4 codebooks) for the serving engine."""
from .w4 import (FP4_CODE, NF4_CODE, W4Weight, codebook, dequantize_absmax, fake_quantize_model_, pack_codes, quantize_absmax,
                 quantize_weight, unpack_codes)

__all__ = ["add_quant_args", "apply_quant_args", "set_weight_format", "FP4_CODE", "NF4_CODE", "W4Weight", "codebook", "dequantize_absmax", "fake_quantize_model_", "pack_codes",
           "quantize_absmax", "quantize_weight", "unpack_codes"]


def add_quant_args(parser) -> None:
    """``--use_4bit`` / ``--quant_type``: run the decoders' projections from 4-bit weights, the
    configuration of the reference's benchmark scripts (``BitsAndBytesConfig(load_in_4bit=True,
    bnb_4bit_quant_type="nf4")``, pipeline/benchmark_e2e/benchmark_e2e_wallclock.py:292-329).
    Off by default here: bf16 weights fit the GPU many times over."""
    parser.add_argument("--use_4bit", action="store_true",
                        help="serve the decoder projections from 4-bit weights (GPU engine only)")
    parser.add_argument("--quant_type", choices=["nf4", "fp4"], default="nf4")


def set_weight_format(model, fmt: str) -> None:
    """Make the serving engine that ``model`` builds on first use hold ``fmt`` weights."""
    target = getattr(model, "language_model", model)      # Video-LLaVA wraps its decoder
    if getattr(target, "_engine", None) is not None:
        raise RuntimeError("the model's engine is already built; choose the weight format first")
    target.engine_defaults = {**getattr(target, "engine_defaults", {}), "weight_format": fmt}


def apply_quant_args(args, *models) -> None:
    if getattr(args, "use_4bit", False):
        for m in models:
            if m is not None:
                set_weight_format(m, args.quant_type)
