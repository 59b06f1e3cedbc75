"""Worker for the tensor-parallel test: sharded decoder on gloo vs the full model."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventgpt_amd.models import EventChatConfig, EventChatModel  # noqa: E402
from eventgpt_amd.parallel.dist import init_distributed, shutdown  # noqa: E402
from eventgpt_amd.parallel.tp import TensorParallelDecoder, gather_vocab_shards, shard_vocab  # noqa: E402


def main(out_path):
    info = init_distributed(use_gpu=False)
    heads = max(4, info.world)            # every rank owns at least one head
    cfg = EventChatConfig(vocab_size=203, hidden_size=64, intermediate_size=172,
                          num_hidden_layers=2, num_attention_heads=heads, max_position_embeddings=128,
                          torch_dtype="float32")
    model = EventChatModel.from_config(cfg, torch.float32, "cpu", seed=5)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 9, 64, generator=g)
    full = model.model(inputs_embeds=x).last_hidden_state
    full_tok = model.lm_head(full[:, -1]).argmax(-1)
    tp = TensorParallelDecoder(model, dist.group.WORLD)
    h = tp(x)
    tok = tp.greedy_next(h[:, -1])
    err = (h - full).abs().max().item()
    # vocabulary shards (203 rows do not divide evenly) gathered back to full logits
    w_local, _ = shard_vocab(model.lm_head.weight.data, info.rank, info.world)
    logits = gather_vocab_shards(full[:, -1] @ w_local.T, 203, dist.group.WORLD)
    gerr = (logits - model.lm_head(full[:, -1])).abs().max().item() if logits.shape == (2, 203) else 1e9
    gathered = [None] * info.world
    dist.all_gather_object(gathered, {"err": err, "tok": tok.tolist(), "gather_err": gerr})
    if info.rank == 0:
        json.dump({"ranks": gathered, "want_tok": full_tok.tolist(),
                   "i_local": tp.shards[0][2].shape[0] // 2}, open(out_path, "w"))
    shutdown()


if __name__ == "__main__":
    main(sys.argv[1])
