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


def test_dry_run_runs_the_communicator_agreement_without_entering_the_collective_init():
    """N > 1: before vfsx_comm_init every rank reports its pre-flight over the process group and all of them raise
    the same error when any failed.  Without GPUs every rank reports 'device not visible'; nobody hangs."""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    pf = out["comm_preflight"]
    assert pf == "ok" or ("no rank enters ncclCommInitRank" in pf and "rank 0" in pf and "rank 1" in pf), pf


def test_a_rank_that_dies_before_the_agreement_fails_the_job_within_the_timeout():
    """First contact with N GPUs: one rank is gone before the communicator is agreed on.  The job must end with a
    non-zero status inside the stated bound (the launcher tears the survivors down; the process group's own
    timeout, VFSX_BENCH_PG_TIMEOUT_S, bounds a launcher that does not)."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
             {"VFSX_BENCH_FAULT": "exit_before_agree:1", "VFSX_BENCH_PG_TIMEOUT_S": "60"})
    assert r.returncode != 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert time.time() - t0 < 150
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{") and "dry_run" in ln]  # no line was printed
