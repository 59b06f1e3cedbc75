"""In-tree build of the HIP extension ``eventgpt_amd/_C.so`` for gfx950.

Drives hipcc directly (no hipify, no JIT cache): every ``csrc/kernels/*.hip``
is compiled to an object for ``--offload-arch=gfx950`` only, ``csrc/*.cpp`` are
compiled as host code against the PyTorch headers, and everything is linked
into one shared library that registers its ops with ``TORCH_LIBRARY`` and is
loaded with ``torch.ops.load_library``.  Objects are cached under ``build/`` by
content hash, so rebuilding after a one-file edit takes seconds.

    python -m eventgpt_amd.build          # build if stale
    python -m eventgpt_amd.build --force  # rebuild everything
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import glob
import hashlib
import os
import shutil
import subprocess
import sys
from typing import List

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
BUILD_DIR = os.path.join(os.path.dirname(PKG_DIR), "build", "evgpt_C")
SO_PATH = os.path.join(PKG_DIR, "_C.so")
ARCH = "gfx950"

# -packed-fp32-ops is disabled on purpose: some packed-fp32 forms have produced
# wrong results on this hardware pool, so the compiler must not emit them.
# No -ffast-math: the kernels use +/-inf as softmax / argmax sentinels, and
# finite-math-only lets the compiler fold `x == -INFINITY` away (NaNs follow).
HIP_FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-math-errno",
             "-ffp-contract=fast",
             "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
             "-Wno-unused-result", "-Wno-unknown-warning-option", "-Wno-unused-command-line-argument",
             "-DNDEBUG"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _torch_paths():
    import torch
    from torch.utils import cpp_extension as ce
    inc = ce.include_paths("cuda")
    lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    return inc, lib


def _digest(path: str, extra: str) -> str:
    h = hashlib.sha256()
    h.update(extra.encode())
    with open(path, "rb") as f:
        h.update(f.read())
    for hdr in sorted(glob.glob(os.path.join(CSRC, "**", "*.h"), recursive=True)):
        with open(hdr, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _compile(src: str, flags: List[str], verbose: bool) -> str:
    os.makedirs(BUILD_DIR, exist_ok=True)
    tag = _digest(src, " ".join(flags))
    obj = os.path.join(BUILD_DIR, f"{os.path.basename(src)}.{tag}.o")
    if os.path.exists(obj):
        return obj
    cmd = [_hipcc(), *flags, "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f"hipcc failed on {src}")
    return obj


def sources():
    hip = sorted(glob.glob(os.path.join(CSRC, "kernels", "*.hip")))
    cpp = sorted(glob.glob(os.path.join(CSRC, "*.cpp")))
    return hip, cpp


def build(force: bool = False, verbose: bool = False, jobs: int = 8, extra: str = "",
          out: str = SO_PATH) -> str:
    """``extra``: more hipcc flags for the kernels (``-DEVGPT_...=n`` tuning variants);
    ``out``: where to link (a variant goes next to ``_C.so`` and is picked with ``EVGPT_SO``)."""
    hip, cpp = sources()
    inc, torch_lib = _torch_paths()
    inc_flags = [f"-I{p}" for p in inc] + [f"-I{CSRC}"]
    host_flags = ["-O2", "-std=c++17", "-fPIC",
                  "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_API_INCLUDE_EXTENSION_H",
                  "-D_GLIBCXX_USE_CXX11_ABI=1", "-Wno-unused-result", "-Wno-deprecated-declarations",
                  "-Wno-unknown-warning-option", "-Wno-unused-command-line-argument"] + inc_flags
    if force and os.path.isdir(BUILD_DIR):
        shutil.rmtree(BUILD_DIR)
    tasks = [(s, HIP_FLAGS + extra.split() + [f"-I{CSRC}"]) for s in hip] + [(s, host_flags) for s in cpp]
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda t: _compile(t[0], t[1], verbose), tasks))
    stamp = hashlib.sha256(" ".join(sorted(objs)).encode()).hexdigest()[:16]
    stamp_file = os.path.join(BUILD_DIR, os.path.basename(out) + ".link.stamp")
    if (not force and os.path.exists(out) and os.path.exists(stamp_file)
            and open(stamp_file).read() == stamp):
        return out
    cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", out,
           f"-L{torch_lib}", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip", "-lc10_hip",
           f"-Wl,-rpath,{torch_lib}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("link failed")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return out


def build_tools(verbose: bool = False) -> list:
    """Stand-alone HIP programs of ``benchmarks/hip`` (timing probes with their own ``main``): one
    hipcc call each, next to their source, rebuilt when the source or the shared header is newer."""
    root = os.path.dirname(PKG_DIR)
    outs = []
    for src in sorted(glob.glob(os.path.join(root, "benchmarks", "hip", "*.hip"))):
        out = src[:-4]
        deps = [src, os.path.join(CSRC, "kernels", "common.h")]
        if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
            outs.append(out)
            continue
        flags = [f for f in HIP_FLAGS if f != "-fPIC"]
        cmd = [_hipcc(), *flags, src, "-o", out]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError(f"hipcc failed on {src}")
        outs.append(out)
    return outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("-j", "--jobs", type=int, default=8)
    ap.add_argument("--extra", default="", help="extra hipcc flags for the kernels (tuning variants)")
    ap.add_argument("--out", default=SO_PATH, help="output library (default eventgpt_amd/_C.so)")
    a = ap.parse_args()
    print(build(a.force, a.verbose, a.jobs, a.extra, a.out))


if __name__ == "__main__":
    main()
