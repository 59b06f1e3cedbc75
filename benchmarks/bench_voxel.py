"""Voxel / polarity-histogram kernels on a DSEC-size canvas (480x640, 5 bins): one global atomic
per event vs per-band counting in LDS.  ``python benchmarks/bench_voxel.py [n_events]``."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from eventgpt_amd.ops import event_ops as eo  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    ev = {"x": torch.randint(0, 640, (n,), device=dev, generator=g, dtype=torch.int32),
          "y": torch.randint(0, 480, (n,), device=dev, generator=g, dtype=torch.int32),
          "p": torch.randint(0, 2, (n,), device=dev, generator=g, dtype=torch.int32),
          "t": torch.sort(torch.randint(0, 50_000, (n,), device=dev, generator=g, dtype=torch.int64)).values}
    for by in ("count", "time"):
        res = {}
        for priv in (False, True):
            for _ in range(2):
                out = eo.voxel_grid_gpu(ev, 5, (480, 640), by=by, privatised=priv)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                out = eo.voxel_grid_gpu(ev, 5, (480, 640), by=by, privatised=priv)
            torch.cuda.synchronize()
            res[priv] = ((time.perf_counter() - t0) / 10 * 1e3, out)
        assert torch.equal(res[False][1][0], res[True][1][0])
        print(f"{n:.0e} events, by {by}: atomic-per-event {res[False][0]:.3f} ms, LDS bands {res[True][0]:.3f} ms "
              f"({n / res[True][0] / 1e6:.1f} G events/s), whole call incl. bounds + output zeroing", flush=True)


if __name__ == "__main__":
    main()
