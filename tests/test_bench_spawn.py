"""bench.py as the driver starts it: `python bench.py --gpus N` WITHOUT torch.distributed.run must start the
N ranks itself, and the N-rank job must push its metric records through generation.gather_records.  Run here
on CPU with --dry (gloo, no sampler): rendezvous, self-spawn, barrier / max-over-ranks timing and the all-gather
(with a short last shard) are the real code paths of the GPU run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=timeout)


def test_self_spawn_two_ranks_gather_records():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--dry", "--batch", "8"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert out["records_gathered"] == 8 + 5                             # the last shard is short (8 - 3)
    assert out["record_rank_column"] == [0.0, 1.0]                      # rank order preserved


def test_single_rank_dry_and_world_size_mismatch_message():
    r = _run(["--gpus", "1", "--steps", "1", "--dry"])
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["ranks_seen"] == 1
    bad = _run(["--gpus", "4", "--dry"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert bad.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in (bad.stderr + bad.stdout)


def test_eight_rank_dry_run_gathers_in_rank_order():
    """The 8-GPU job of BASELINE configs[3] / configs[4] without hardware: bench.py starts eight
    gloo ranks itself, every rank contributes its shard -- the last one short --, and the gathered records arrive
    concatenated in rank order (generate_samples_distributed.py:84-95), counted by the collective itself."""
    r = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--dry", "--batch", "6"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8 and out["world_size_after_gather"] == 8
    assert out["records_per_rank"] == [6] * 7 + [3] and out["records_gathered"] == 45
    assert out["rank_column_sorted"] and out["record_rank_runs"] == [[float(i), 6 if i < 7 else 3] for i in range(8)]
    # every rank's own time beside the MAX the value is computed from
    pr = out["per_rank"]
    assert len(pr["ms_per_step"]) == 8 and pr["max"] == max(pr["ms_per_step"]) and pr["min"] == min(pr["ms_per_step"])
    assert pr["ms_per_step"][pr["rank_of_max"]] == pr["max"] and abs(out["ms_per_step"] - pr["max"]) < 1e-3


_RCCL_ONE_RANK = r"""
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("PDR_TEST_PORT", "29731"), RANK="0",
                  WORLD_SIZE="1", LOCAL_RANK="0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)          # bench.py main(), world > 1
assert dist.get_backend() == "nccl"
dist.barrier()
t = torch.tensor([3.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                          # max-over-ranks timing
rec = torch.arange(35, device=dev, dtype=torch.float32).view(7, 5)
n = torch.tensor([7], dtype=torch.int64, device=dev)
counts = [torch.zeros_like(n)]
dist.all_gather(counts, n)                                        # generation.gather_records: lengths, then payload
out = [torch.empty_like(rec)]
dist.all_gather(out, rec)
torch.cuda.synchronize()
assert float(t) == 3.5 and int(counts[0]) == 7 and torch.equal(out[0], rec)
dist.destroy_process_group()
print("rccl-one-rank-ok")
"""


import pytest  # noqa: E402


@pytest.mark.gpu
def test_rccl_backend_runs_the_jobs_collectives_on_this_box():
    """What a 1-GPU box can say about the N-GPU job: the process group bench.py creates for world > 1 (backend
    "nccl" = RCCL, device-bound) comes up on this box's GPU, and the collectives of the path -- barrier, the MAX
    all-reduce of the timing, the two all-gathers of generation.gather_records -- run on device tensors.  (One rank:
    gather_records itself returns early at world size 1, so the calls are made directly.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "rccl-one-rank-ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.gpu
def test_driver_launch_line_with_one_rank_on_the_gpu():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 ... bench.py --gpus 1` (the driver's line) on the
    real device: environment parsing, the timed region and the JSON contract of the line."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29733", os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-roofline",
                        "--no-extras"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["scaling"] == "weak" and out["value"] > 0
    assert out["ranks_seen"] == 1 and out["world_size_after_gather"] == 1
