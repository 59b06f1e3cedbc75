"""bench.py as the driver invokes it: ``--gpus N`` without a torchrun environment must start N
ranks (not silently run one), and a torchrun environment must be honoured as is.  The probe
mode exercises the real launch path -- subprocess -> torch.distributed.run -> N ranks ->
rendezvous -> an all-reduce -- on CPU (gloo), without touching a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


@pytest.mark.timeout(180)
@pytest.mark.parametrize("n", [2, 4])
def test_gpus_flag_launches_n_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--model", "small", "--probe"],
                       env=_clean_env(), capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # exactly one JSON line, from rank 0
    got = json.loads(lines[0])
    assert got["n_gpus"] == n and got["rccl_world_size"] == n and got["requested"] == n
    assert got["scaling"] == "strong"           # tensor-parallel is the multi-GPU default


@pytest.mark.timeout(120)
def test_torchrun_environment_is_honoured():
    """The driver's own launch shape: torchrun starts the ranks, bench.py must not nest another job."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29571", BENCH, "--gpus", "2", "--probe"]
    r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_single_gpu_default_does_not_launch():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args([])
    assert a.gpus == 1 and a.parallelism == "tp" and a.batch == 1 and a.settle_steps == 96
    child = bench.oneshot_child_command(bench.parse_args(["--gpus", "4", "--steps", "20", "--settle-steps", "50"]), 4)
    assert child[child.index("--allreduce") + 1] == "auto" and child[child.index("--settle-steps") + 1] == "50"
    assert "--no-oneshot-extra" in child and "--no-dp-extra" in child      # the child job does not nest another
    assert "--no-dp-train-extra" in child and "--no-sd-extra" in child
    # the child job exists at the full width of a node too (two processes per GPU for its duration)
    child8 = bench.oneshot_child_command(bench.parse_args(["--gpus", "8"]), 8)
    assert child8[child8.index("--gpus") + 1] == "8" and child8[child8.index("--allreduce") + 1] == "auto"
    assert float(child8[child8.index("--extras-timeout") + 1]) >= 60
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "5"], 29999)
    assert "--nproc-per-node=8" in cmd and cmd[-4:] == ["--gpus", "8", "--steps", "5"]
    assert "127.0.0.1" in cmd


@pytest.mark.gpu
@pytest.mark.timeout(420)
@pytest.mark.parametrize("n", [2, 4])
def test_multi_rank_flow_rehearsal_on_one_gpu(n):
    """The whole multi-rank bench flow -- launcher, tensor-parallel engine, the 32-stream run and the
    independent replicas, the MAX over ranks, one JSON line -- with the ranks sharing GPU 0 over gloo.
    The 8-rank case belongs to the scaling runs on the 8-GPU node and is not started from here."""
    env = _clean_env()
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    print(f"launching the {n}-rank rehearsal", flush=True)
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--model", "small", "--rehearsal", "--steps", "4",
                        "--warmup", "2", "--max-len", "1024"], env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    got = json.loads(lines[0])
    assert got["n_gpus"] == n and got["rccl_world_size"] == n and got["scaling"] == "strong"
    assert got["value"] > 0 and got["config"]["parallelism"] == f"tp{n}" and got["tp_allreduce"] == "gloo"
    assert "error" not in got["tp_batch32"] and got["tp_batch32"]["tokens_per_s"] > 0, got["tp_batch32"]
    assert "error" not in got["dp_replicas"] and got["dp_replicas"]["global_batch"] == n, got["dp_replicas"]
    # the child job on the in-tree all-reduce (IPC-mapped buffers between the ranks' processes)
    one = got["tp_oneshot"]
    assert "error" not in one and one["tp_allreduce"] == "oneshot" and one["tokens_per_s"] > 0, one
    assert "error" not in one["tp_batch32"], one
    # a fine-tuning step per rank through the sharded AdamW over the run's group, both stages
    dpt = got["dp_train"]
    assert "error" not in dpt and dpt["global_batch"] == n, dpt
    for stage in ("projector", "lora"):
        assert dpt[stage]["no_overlap"]["ms_per_step"] > 0 and dpt[stage]["trainable_parameters"] > 0, dpt
    # prefill hiding through the two-model scheduler (one device here: the rehearsal has no second GPU)
    sd = got["sd_two_gpu"]
    assert "error" not in sd and sd["one_gpu_two_streams"]["prefill_hiding_ms"] > 0, sd
    assert "same_tokens_as_verifier" in sd["one_gpu_two_streams"]      # (may differ at bf16 near-ties)
    assert got["ms_per_step_cold"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_extras_time_limit_keeps_the_headline_line():
    """Additional measurements that outlast their limit (here: ranks idling on the host) do not cost the
    headline line: rank 0 prints it with the extras marked unfinished and every rank ends with status 0."""
    env = _clean_env()
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    print("launching the 2-rank rehearsal with a 2 s limit on the extras", flush=True)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--model", "small", "--rehearsal", "--steps", "4",
                        "--warmup", "2", "--max-len", "1024", "--extras-timeout", "2", "--extras-stall", "30"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    got = json.loads(lines[0])
    assert got["n_gpus"] == 2 and got["value"] > 0
    assert "not finished" in got["tp_batch32"]["error"] and "not finished" in got["dp_replicas"]["error"]
    assert "not finished" in got["tp_oneshot"]["error"]
    assert "not finished" in got["dp_train"]["error"] and "not finished" in got["sd_two_gpu"]["error"]
