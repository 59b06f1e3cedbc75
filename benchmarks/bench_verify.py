"""Time the speculative-decoding verify pass (G draft rows in one forward) against a
plain decode step on the 7B decoder.  ``python benchmarks/bench_verify.py [G] [prompt_len] [layers] [verify_only]``"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.models import EventChatConfig, EventChatModel  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 5
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2115
dev = torch.device("cuda")
cfg = EventChatConfig.eventgpt_7b()
cfg.num_hidden_layers = int(sys.argv[3]) if len(sys.argv) > 3 else 32
model = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=0)
eng = model.engine(max_len=4096)
eng.eos = -1
emb = (torch.randn(L, cfg.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
eng.reset()
eng.prefill(emb, select="greedy")
draft = torch.randint(0, 32000, (G,), device=dev, dtype=torch.int32)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t_dec = timeit(lambda: eng.decode_step(True))
eng.pos.fill_(L); eng.step.fill_(1)
def ver():
    eng.pos.fill_(L); eng.step.fill_(1)
    eng.verify(draft, G)
t_ver = timeit(ver)
if len(sys.argv) > 4 and sys.argv[4] == "verify_only":      # for a kernel trace of the verify pass alone
    timeit(ver, 200)
    print(f"layers={cfg.num_hidden_layers} kv_len={L} decode_step={t_dec:.3f} ms  verify(G={G})={t_ver:.3f} ms", flush=True)
    sys.exit(0)
# same pass with the few-row fusion switched off (separate reduce / norm / rope / swiglu launches)
eng.fused_rows = 0
eng._verify_graphs.clear()
t_unf = timeit(ver)
print(f"verify(G={G}) unfused={t_unf:.3f} ms", flush=True)
print(f"layers={cfg.num_hidden_layers} kv_len={L} decode_step={t_dec:.3f} ms  verify(G={G})={t_ver:.3f} ms  "
      f"ratio={t_ver / t_dec:.2f}", flush=True)

# ---- speculative-decoding rounds end to end: draft head (B1 adapter from the verifier's own
# hidden state, gamma = 1) + verify + accept, host-driven vs device-resident rounds
from eventgpt_amd import ops  # noqa: E402
from eventgpt_amd.align.adapters import create_adapter  # noqa: E402
from eventgpt_amd.runtime.spec_decode import AdapterDraftHead, SpecDecodeEngine  # noqa: E402

eng.fused_rows = 32
eng._verify_graphs.clear()
torch.manual_seed(0)
adapter = create_adapter(5, cfg.hidden_size).to(dev).to(torch.bfloat16)
head = AdapterDraftHead(adapter, model.lm_head.weight.data)
sd = SpecDecodeEngine(eng, head, gamma=5)
n_tok = 48
for label, dl in (("host loop", False), ("device loop", True)):
    for rep in range(2):        # first repetition captures the graphs
        eng.reset(); eng.eos = -1
        h = eng.prefill(emb, select="greedy", return_hidden=True)
        vh0 = ops.rmsnorm(h[-1:], eng.final_norm, eng.eps)
        toks, st = sd.run(None, n_tok, -1, "vlm_only", vh0, device_loop=dl)
    print(f"SD rounds ({label}, B1 gamma=1): {st.decode_ms:.1f} ms for {st.tokens} tokens, {st.iterations} rounds: "
          f"{st.decode_ms / max(st.iterations, 1):.3f} ms/round, {st.decode_ms / max(st.tokens, 1):.3f} ms/token "
          f"(decode step {t_dec:.3f} ms), accept {st.accept_rate:.2f}", flush=True)
