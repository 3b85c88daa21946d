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
