"""Adapter-training throughput on synthetic paired hidden states:
``python benchmarks/bench_adapter_train.py [level] [n_samples] [batch] [library]`` prints samples/s
of one epoch of ``align.trainer`` (the reference reports L4: 300 epochs of ~52 K samples
in ~10 h on an RTX 4090, i.e. ~430 samples/s)."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from eventgpt_amd.align.store import ChunkedHiddenStateWriter  # noqa: E402
from eventgpt_amd.align import trainer as TR  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
library = len(sys.argv) > 4 and sys.argv[4] == "library"     # adapter GEMMs on the vendor library instead of in-tree
root = tempfile.mkdtemp(prefix="evgpt_adtrain_")
w = ChunkedHiddenStateWriter(os.path.join(root, "store"), chunk_size=1024)
g = torch.Generator().manual_seed(0)
for i in range(n):
    s = 30 + i % 21
    e = torch.randn(s, 4096, generator=g)
    w.add_sample(e, e + 0.1 * torch.randn(s, 4096, generator=g), {"sample_idx": i, "question_idx": 0})
w.flush()
t0 = time.perf_counter()
TR.main(["--train_data", os.path.join(root, "store"), "--adapter_level", str(level), "--num_epochs", "2",
         "--batch_size", str(batch), "--output_dir", os.path.join(root, "tasks"), "--val_split", "0.05"]
        + (["--library_gemm"] if library else []))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"level {level} ({'library' if library else 'in-tree'} GEMM): {2 * n / dt:.0f} samples/s over 2 epochs incl. validation and start-up ({dt:.1f} s)", flush=True)
