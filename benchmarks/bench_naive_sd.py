"""Token-level draft + verify speculative decoding at 7B (``runtime/spec_decode.py: NaiveSpeculativeDecoder``):
an EventGPT-7B-shaped drafter against a 7B verifier whose prompt is Video-LLaVA-sized (2 115 tokens),
device-resident rounds.  Reference loop: feasible/naive_speculative_decoding.py:7-60 (no KV cache, both
models re-run over the whole sequence per draft token).

Two regimes bracket what any trained drafter can do with random-initialised weights:

* ``same``   drafter == verifier weights and prompt: every draft is accepted (near-ties aside) -- the upper
             bound (gamma + 1 tokens per round) and an end-to-end losslessness check;
* ``indep``  an unrelated 7B drafter on its own 636-token prompt: almost nothing is accepted -- the overhead
             bound (one token per round).

Gaussian random weights give nearly flat logits: the best two of 32 003 candidates are closer than the bf16
kernels' rounding a quarter of the time, so a one-row decode step and a multi-row verify pass of the SAME
model often pick different tokens, which no trained model does.  ``--peaked`` (the default) therefore ties
the LM head to a permutation of the (scaled-up) token embeddings: the next token is a fixed function of the
current one with a wide margin, like a confident trained model; everything else stays random, and the time
of a kernel does not depend on the values it multiplies.  ``--no-peaked`` keeps plain Gaussian weights.

    python benchmarks/bench_naive_sd.py [--gammas 3,5,8] [--tokens 96] [--regime both] [--layers 32]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventgpt_amd.models import EventChatConfig, EventChatModel  # noqa: E402
from eventgpt_amd.runtime.spec_decode import NaiveSpeculativeDecoder, matches_greedy  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gammas", default="3,5,8")
ap.add_argument("--tokens", type=int, default=96)
ap.add_argument("--regime", choices=["both", "same", "indep"], default="both")
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--verifier-prompt", type=int, default=2115)
ap.add_argument("--drafter-prompt", type=int, default=636)
ap.add_argument("--poll-every", type=int, default=4)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--host-loop", action="store_true", help="also time the host-driven loop")
ap.add_argument("--json", default=None, help="write the rows to this file")
ap.add_argument("--no-peaked", action="store_true", help="plain Gaussian weights (near-tied logits)")
args = ap.parse_args()

dev = torch.device("cuda")
cfg = EventChatConfig(vocab_size=32003, hidden_size=4096, intermediate_size=11008, num_hidden_layers=args.layers,
                      num_attention_heads=32, max_position_embeddings=4096, torch_dtype="bfloat16")


def build(seed):
    m = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=seed)
    if not args.no_peaked:
        with torch.no_grad():
            gen = torch.Generator(device=dev).manual_seed(1000 + seed)
            perm = torch.randperm(cfg.vocab_size, device=dev, generator=gen)
            m.model.embed_tokens.weight.mul_(150.0)           # the token's own row dominates the residual stream
            m.lm_head.weight.copy_(m.model.embed_tokens.weight[perm] / 150.0)
    return m, m.engine(max_len=4096, max_new_tokens=args.tokens + 80)


g = torch.Generator(device=dev).manual_seed(1)
v_emb = (torch.randn(args.verifier_prompt, cfg.hidden_size, device=dev, generator=g) * 0.02).to(torch.bfloat16)
d_emb = (torch.randn(args.drafter_prompt, cfg.hidden_size, device=dev, generator=g) * 0.02).to(torch.bfloat16)
print(f"[naive-sd] building verifier ({args.layers} layers)", flush=True)
_, v_eng = build(5)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3


# ---- verifier-only greedy decode (the baseline every speedup is against): prefill, then N-1 steps
greedy = None
base_ms = []
for _ in range(args.reps):
    v_eng.eos = -1
    v_eng.reset()
    v_eng.prefill(v_emb, select="greedy")
    _, ms = timed(lambda: v_eng.generate_continue(args.tokens, -1))
    base_ms.append(ms)
greedy = v_eng.out_tokens[0, :args.tokens].long().clone()
base = min(base_ms)
print(f"[naive-sd] verifier-only decode after a {args.verifier_prompt}-token prompt: {base:.1f} ms for "
      f"{args.tokens - 1} steps = {base / (args.tokens - 1):.3f} ms/token", flush=True)

rows = []
regimes = ("same", "indep") if args.regime == "both" else (args.regime,)
for regime in regimes:
    print(f"[naive-sd] building drafter ({regime})", flush=True)
    _, d_eng = build(5 if regime == "same" else 6)
    sd = NaiveSpeculativeDecoder(d_eng, v_eng)
    demb = v_emb if regime == "same" else d_emb
    # the two prefills a call to generate() starts with (subtracted from the host-driven loop's wall time)
    def prefills():
        v_eng.reset(); d_eng.reset()
        v_eng.prefill(v_emb, select="greedy"); d_eng.prefill(demb, select=None)
    prefills()
    prefill_ms = min(timed(prefills)[1] for _ in range(3))
    for gamma in [int(x) for x in args.gammas.split(",")]:
        for device_loop in ([True, False] if args.host_loop else [True]):
            best = None
            for rep in range(args.reps + 1):          # the first repetition captures the round's graph
                (toks, hist), gen_ms = timed(lambda: sd.generate(demb, v_emb, args.tokens, gamma, -1,
                                                                  device_loop=device_loop, poll_every=args.poll_every))
                st = sd.stats
                # the rounds alone: replay them once more from the same start for a clean clock
                if device_loop:
                    graph = sd._round_graph(gamma, demb.shape[0] - v_emb.shape[0])
                    v_eng.reset(); d_eng.reset()
                    v_eng.prefill(v_emb, select="greedy"); d_eng.prefill(demb, select=None)
                    v_eng.sd_limit.fill_(args.tokens); v_eng.sd_stats.zero_(); v_eng.sd_result.zero_()
                    sd._prompt_last.copy_(demb[-1])
                    n_rounds = st.iterations

                    def rounds():
                        done = 0
                        reads = 0
                        while done < n_rounds:
                            for _ in range(min(args.poll_every, n_rounds - done)):
                                graph.replay()
                            done += args.poll_every
                            v_eng.sd_result[:6].tolist()
                            reads += 1
                        return reads
                    reads, ms = timed(rounds)
                    v_eng.sd_limit.fill_(1 << 30)
                else:
                    reads, ms = st.iterations, gen_ms - prefill_ms
                if rep and (best is None or ms < best[0]):
                    best = (ms, reads)
            chk = matches_greedy(v_eng, v_emb, toks, greedy[:len(toks)])
            n_tok = int(toks.shape[0]) - 1             # tokens produced by the rounds (the first comes from the prefill)
            row = {"regime": regime, "gamma": gamma, "loop": "device" if device_loop else "host",
                   "rounds": st.iterations, "accept_rate": round(st.accept_rate, 4), "tokens": n_tok,
                   "lossless": chk["lossless"], "exact": chk["exact"], "first_mismatch": chk["first_mismatch"],
                   "mismatch_logit_gap": chk["mismatch_logit_gap"]}
            if best is not None:
                ms, reads = best
                row.update({"ms": round(ms, 2), "ms_per_round": round(ms / st.iterations, 3),
                            "tokens_per_s": round(n_tok / ms * 1e3, 1), "host_reads": reads,
                            "speedup_vs_verifier_only": round((base / (args.tokens - 1)) / (ms / n_tok), 3)})
            rows.append(row)
            print("[naive-sd] " + json.dumps(row), flush=True)
    del sd, d_eng
    torch.cuda.empty_cache()

out = {"verifier_only_ms_per_token": round(base / (args.tokens - 1), 4), "verifier_prompt": args.verifier_prompt,
       "drafter_prompt": args.drafter_prompt, "layers": args.layers, "tokens": args.tokens,
       "peaked_logits": not args.no_peaked, "rows": rows}
if args.json:
    os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
    with open(args.json, "w") as f:
        json.dump(out, f, indent=1)
print(json.dumps(out), flush=True)
