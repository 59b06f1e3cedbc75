"""Where the forward of one instruction-tuning step goes (7B geometry, projector tuning): wall time with
a device synchronisation around the event encoder, the decoder layers, the LM head and the loss, and
the decoder's wall time against the device time of its kernels (HIP events) -- a gap between the two is
the host's launch rate.  ``python benchmarks/bench_sft_forward_parts.py [n_steps]``"""
import sys
import tempfile
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.cli.train_sft import SFTTrainer  # noqa: E402
from eventgpt_amd.data.tokenizer import ByteTokenizer  # noqa: E402
from eventgpt_amd.models import EventChatConfig, EventChatModel  # noqa: E402
from eventgpt_amd.parallel.dist import init_distributed  # noqa: E402
from eventgpt_amd.pipeline.dataset import InstructionDataset, make_synthetic_dataset  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
info = init_distributed(use_gpu=True)
cfg = EventChatConfig.eventgpt_7b()
model = EventChatModel.from_config(cfg, torch.bfloat16, info.device, seed=0)
tok = ByteTokenizer(cfg.vocab_size)
ds = InstructionDataset(make_synthetic_dataset(tempfile.mkdtemp(prefix="evgpt_sftp_"), n_samples=steps + 3,
                                               n_events=60000, width=640, height=480))
tr = SFTTrainer(model, tok, ds, info, tune="projector", output_dir=tempfile.mkdtemp(prefix="evgpt_sftp_out_"))
model.train()
acc = {}


def timed(name, fn):
    def wrapped(*a, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*a, **kw)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return out
    return wrapped


import eventgpt_amd.cli.train_sft as _ts  # noqa: E402
_pf = _ts.preprocess_frames
_ts.preprocess_frames = timed("preprocess", lambda *a, **kw: list(_pf(*a, **kw)))
model.prepare_inputs_labels_for_multimodal = timed("splice (incl. encoder)", model.prepare_inputs_labels_for_multimodal)
model.encode_event_frames = timed("event encoder", model.encode_event_frames)
model.model.forward = timed("decoder layers", model.model.forward)
model.lm_head.forward = timed("lm_head", model.lm_head.forward)
_ce = torch.nn.functional.cross_entropy
torch.nn.functional.cross_entropy = timed("cross entropy", _ce)
total = 0.0
for i in range(steps + 3):
    if i == 3:
        acc.clear()
        total = 0.0
    prep = tr.prepare_sample(ds[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = tr.sample_loss(ds[i], prep)
    torch.cuda.synchronize()
    total += time.perf_counter() - t0
    loss.backward()
    tr.opt.step(max_grad_norm=1.0)
    tr.opt.zero_grad()
parts = {k: round(v / steps * 1e3, 2) for k, v in acc.items()}
outer = sum(v for k, v in acc.items() if k != "event encoder")
print("forward %.2f ms:" % (total / steps * 1e3), parts, "rest %.2f" % ((total - outer) / steps * 1e3), flush=True)
