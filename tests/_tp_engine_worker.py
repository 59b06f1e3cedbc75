"""Worker for the GPU tensor-parallel engine tests: the ranks (two or eight) share GPU 0, the process
group is gloo (host tensors only), graphs are off (a host copy cannot be captured).

Two ranks run the sharded engine in four configurations -- 2 and 6 streams (the weight-streaming
GEMV path and the split-K GEMM + fused-consumer path), each with the library all-reduce
(gloo through host copies) and with the push all-reduces over IPC-mapped buffers (one-shot for the
decode steps, two-shot for the prefill's [rows, hidden] products; two processes, one GPU: real
``hipIpcMemHandle`` exchange, every device wait bounded) --
and compare each against the unsharded engine on the same prompts; eight ranks (one attention
head, 192 padded MLP columns and 65 vocabulary rows each) run the library all-reduce only."""
import json
import os
import sys

os.environ["LOCAL_RANK"] = "0"          # every rank of this test drives the one GPU of the box
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventgpt_amd.models import EventChatConfig, EventChatModel  # noqa: E402
from eventgpt_amd.parallel.dist import init_distributed, shutdown  # noqa: E402
from eventgpt_amd.parallel.tp import TensorParallelEngine  # noqa: E402

cos = torch.nn.functional.cosine_similarity


def reference(model, prompts):
    single = model.engine(max_len=256, max_new_tokens=32, use_graph=False)
    want, want_x = [], []
    for p in prompts:
        single.eos = -1
        single.reset()
        single.prefill(p)
        xs = [single.x[0].float().clone()]          # next-step input embedding after the prefill commit
        single._layers_step()
        xs.append(single.x[0].float().clone())      # residual stream after the first decode step's layers
        single._logits_and_commit(greedy=True)
        want_x.append(xs)
        want.append(single.generate(p, 8, eos_token_id=-1).tolist())
    return want, want_x


def run_config(model_fn, info, prompts, want_x, allreduce):
    B = len(prompts)
    eng = TensorParallelEngine(model_fn(), dist.group.WORLD, max_len=256, max_batch=B, max_new_tokens=32,
                               use_graph=False, allreduce=allreduce)
    eng.eos = -1
    packed = []
    inner = eng._forward_rows_packed
    eng._forward_rows_packed = lambda e, s: (packed.append(len(s)), inner(e, s))[1]
    big = []
    if eng.twoshot is not None:
        inner2 = eng.twoshot.reduce
        eng.twoshot.reduce = lambda x, out=None: (lambda ok: (big.append(bool(ok)), ok)[1])(inner2(x, out))
    eng.prefill_batch(prompts)
    first_tok = [int(eng.out_tokens[b, 0]) for b in range(B)]
    c0 = [float(cos(eng.x[b].float(), want_x[b][0], dim=0)) for b in range(B)]
    eng._layers_step()
    c1 = [float(cos(eng.x[b].float(), want_x[b][1], dim=0)) for b in range(B)]
    eng._logits_and_commit(greedy=True)
    for _ in range(6):
        eng.decode_step(True)
    torch.cuda.synchronize()
    got = [eng.tokens(b).tolist() for b in range(B)]
    # sampling path: every rank must hold the same full-vocabulary logits row
    eng._logits_and_commit(greedy=False)
    row = eng.logits[0].cpu()
    timeouts = 0
    if eng.oneshot is not None:
        timeouts = int(eng.oneshot.buf.err[0].item())
    if eng.twoshot is not None:
        timeouts += int(eng.twoshot.buf.err[0].item())
    rows = [None] * info.world
    dist.all_gather_object(rows, row.tolist())
    gathered = [None] * info.world
    dist.all_gather_object(gathered, (got, timeouts))
    res = {"ranks": [g[0] for g in gathered], "timeouts": [g[1] for g in gathered], "first_tok": first_tok,
           "cos_prefill": c0, "cos_step": c1, "heads_local": eng.Hq, "i_local": eng.I,
           "logits_width": int(eng.logits.shape[1]), "logits_equal": all(r == rows[0] for r in rows),
           "allreduce_kind": eng.allreduce_kind, "packed_prompts": sum(packed)}
    res["twoshot_calls"] = sum(big)
    if eng.oneshot is not None:
        eng.oneshot.close()
    if eng.twoshot is not None:
        eng.twoshot.close()
    return res


def main(out_path):
    info = init_distributed(backend="gloo", use_gpu=True)
    dev = info.device
    wide = os.environ.get("EVGPT_TP_WORKER_GEOMETRY") == "7b"
    if wide:
        # two layers of the 7B decoder: a rank of eight multiplies by 1536 x 4096 (q|k|v), 4096 x 512 (o_proj),
        # 2816 x 4096 (gate|up, 1376 columns padded to 1408), 4096 x 1408 (down) and 4001 vocabulary rows
        cfg = EventChatConfig(vocab_size=32003, hidden_size=4096, intermediate_size=11008, num_hidden_layers=2,
                              num_attention_heads=32, max_position_embeddings=512)
    else:
        # head_dim stays 128; with more than four ranks the model gets one head per rank
        heads = max(4, info.world)
        cfg = EventChatConfig(vocab_size=515, hidden_size=128 * heads, intermediate_size=1408, num_hidden_layers=2,
                              num_attention_heads=heads, max_position_embeddings=512)
    H = cfg.hidden_size
    modes = ("rccl", "oneshot") if info.world == 2 else ("rccl",)   # the IPC all-reduce is exercised by two processes
    model_fn = lambda: EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=3)   # noqa: E731
    out = {}
    for B in ((1, 3, 6) if wide else (2, 6)):
        # one stream (7B shards): one prompt of 150 rows through the prompt GEMMs, then the one-row step;
        # two / three streams: short prompts, prefilled one by one; six streams: every prompt longer than 32
        # tokens, so they are prefilled as ONE pack (shared projections, per-slot attention)
        base = 150 if B == 1 else 20 if B <= 3 else 40
        prompts = [(torch.randn(base + 3 * b, H, device=dev, generator=torch.Generator(device=dev).manual_seed(b))
                    * 0.5).to(torch.bfloat16) for b in range(B)]
        want, want_x = reference(model_fn(), prompts)
        print(f"rank {info.rank}: single-engine reference for {B} streams done", flush=True)
        for ar in modes:
            print(f"rank {info.rank}: {B} streams, allreduce={ar}", flush=True)
            res = run_config(model_fn, info, prompts, want_x, ar)
            res["want"] = want
            out[f"b{B}_{ar}"] = res
    if info.rank == 0:
        json.dump(out, open(out_path, "w"))
    shutdown()


if __name__ == "__main__":
    main(sys.argv[1])
