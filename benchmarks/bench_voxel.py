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
    t = torch.sort(torch.randint(0, 50_000, (n,), device=dev, generator=g, dtype=torch.int64)).values
    p = torch.randint(0, 2, (n,), device=dev, generator=g, dtype=torch.int32)
    uniform = {"x": torch.randint(0, 640, (n,), device=dev, generator=g, dtype=torch.int32),
               "y": torch.randint(0, 480, (n,), device=dev, generator=g, dtype=torch.int32), "p": p, "t": t}
    # a scene: events cluster on a few moving edges (8 blobs, sigma 12 px), as sensor data does
    c = torch.rand(8, 2, device=dev, generator=g) * torch.tensor([640.0, 480.0], device=dev)
    which = torch.randint(0, 8, (n,), device=dev, generator=g)
    pos = c[which] + torch.randn(n, 2, device=dev, generator=g) * 12.0
    scene = {"x": pos[:, 0].clamp(0, 639).to(torch.int32), "y": pos[:, 1].clamp(0, 479).to(torch.int32), "p": p, "t": t}
    for name, ev, by in (("uniform", uniform, "count"), ("uniform", uniform, "time"), ("clustered", scene, "time")):
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
        print(f"{n:.0e} {name} events, by {by}: atomic-per-event {res[False][0]:.3f} ms, LDS bands {res[True][0]:.3f} ms "
              f"({n / res[True][0] / 1e6:.1f} G events/s), whole call incl. bounds + output zeroing", flush=True)


def sweep():
    """LDS-band kernel only: event chunks per (band, bin)."""
    n = 10_000_000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    ev = {"x": torch.randint(0, 640, (n,), device=dev, generator=g, dtype=torch.int32),
          "y": torch.randint(0, 480, (n,), device=dev, generator=g, dtype=torch.int32),
          "p": torch.randint(0, 2, (n,), device=dev, generator=g, dtype=torch.int32),
          "t": torch.arange(n, device=dev, dtype=torch.int64)}
    for chunks in (1, 2, 3, 4, 6, 10, 16):
        for _ in range(2):
            eo.voxel_grid_gpu(ev, 5, (480, 640), by="count", privatised=True, chunks=chunks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eo.voxel_grid_gpu(ev, 5, (480, 640), by="count", privatised=True, chunks=chunks)
        torch.cuda.synchronize()
        print(f"chunks {chunks}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        sweep()
        sys.exit(0)
    main()
