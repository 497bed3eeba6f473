"""CPU: `python bench.py --gpus N` from a plain shell (no WORLD_SIZE) starts its own N ranks through
torch.distributed.run and relays rank 0's JSON line; the default N > 1 workload is the C5 shard per rank.
Rehearsed with --dry-run (touches no GPU) on the gloo backend."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["VFSX_BENCH_BACKEND"] = "gloo"
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    return r


def test_self_launch_two_ranks_dry_run_names_c5():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # only rank 0 prints
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["ranks_joined"] == 2 and out["n_gpus"] == 2
    cfg = out["config"]
    assert cfg["workload"].startswith("C5 shard 12.5Mx768 f32 Euclidean")
    assert cfg["rows_per_gpu"] == 12_500_000 and cfg["rows_total"] == 25_000_000 and cfg["metric_fn"] == "euclidean"
    assert out["steps"] == 3 and out["warmup"] == 1


def test_single_gpu_default_is_c2_and_world_mismatch_is_an_error():
    r = _run(["--dry-run"])
    assert r.returncode == 0, r.stderr[-1500:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["ranks_joined"] == 1 and out["config"]["workload"].startswith("C2 1Mx768 f32 Cosine")
    assert out["config"]["queries_per_step"] >= 128  # the driver's 20 steps cover >= 2 s of device time
    r = _run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2
