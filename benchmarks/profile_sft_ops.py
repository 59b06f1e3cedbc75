"""Which framework ops (not kernels) a fine-tuning step is made of: one step of
``benchmarks/bench_sft_step.py``'s loop under ``torch.profiler``, grouped by op and input shapes,
sorted by device time.  ``python benchmarks/profile_sft_ops.py [projector|lora|full] [micro_batch]``"""
import sys
import tempfile

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.cli.train_sft import SFTTrainer  # noqa: E402
from eventgpt_amd.data.tokenizer import ByteTokenizer  # noqa: E402
from eventgpt_amd.models import EventChatConfig, EventChatModel  # noqa: E402
from eventgpt_amd.parallel.dist import init_distributed  # noqa: E402
from eventgpt_amd.pipeline.dataset import InstructionDataset, make_synthetic_dataset  # noqa: E402

tune = sys.argv[1] if len(sys.argv) > 1 else "lora"
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
info = init_distributed(use_gpu=True)
cfg = EventChatConfig.eventgpt_7b()
model = EventChatModel.from_config(cfg, torch.bfloat16, info.device, seed=0)
tok = ByteTokenizer(cfg.vocab_size)
ds = InstructionDataset(make_synthetic_dataset(tempfile.mkdtemp(prefix="evgpt_sftp_"), n_samples=max(6, 4 * mb),
                                               n_events=60000, width=640, height=480))
tr = SFTTrainer(model, tok, ds, info, tune=tune, output_dir=tempfile.mkdtemp(prefix="evgpt_sftp_out_"),
                gradient_accumulation=mb, micro_batch=mb)
model.train()


def step(i):
    if mb > 1:
        loss = tr.batch_loss([tr.prepare_sample(ds[(i * mb + j) % len(ds)]) for j in range(mb)])
    else:
        prep = tr.prepare_sample(ds[i])
        loss = tr.sample_loss(ds[i], prep)
    loss.backward()
    tr.opt.step(max_grad_norm=1.0)
    tr.opt.zero_grad()


for i in range(3):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(3)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=60,
                                                         max_name_column_width=48, max_shapes_column_width=70))
