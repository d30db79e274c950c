"""`python bench.py --gpus N` must run by itself (the driver's command shape): the parent starts N fresh rank
processes before anything touches a GPU, relays rank 0's single JSON line and fails loudly otherwise.  Rehearsed here
on CPU with --stub (gloo backend, a fake step): launcher, rendezvous on 127.0.0.1, barrier + max-over-ranks timing and
the aggregation of the result line are the real code path; only the step is a stand-in."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _run(args, env=None, timeout=180):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e, cwd=REPO)


def test_self_launch_two_ranks_stub():
    r = _run(["--gpus", "2", "--stub", "--steps", "6", "--warmup", "2", "--batch", "128"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2
    assert out["config"]["parallelism"] == "dp2" and "allreduce" in out["config"]
    assert out["config"]["ranks_seen"] == 2      # counted through the process group, not read from the environment
    assert out["stub_grad_mean"] == 1.5  # mean of rank-valued "gradients" 1 and 2: the collective ran
    # rank 1 sleeps 2 ms per step, rank 0 1 ms: the reported time is the MAX over ranks
    assert out["ms_per_step"] >= 2.0
    assert abs(out["value"] - 128 * 2 / (out["ms_per_step"] / 1e3)) <= 0.01 * out["value"]  # whole-job rows / time


def test_rank_set_by_torchrun_style_env_is_respected():
    """with WORLD_SIZE already set (torch.distributed.run) the script is a rank, not a launcher"""
    r = _run(["--gpus", "1", "--stub", "--steps", "3", "--warmup", "1"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip())["n_gpus"] == 1


def test_failing_rank_fails_the_job_without_a_result_line():
    r = _run(["--gpus", "2", "--stub", "--steps", "3", "--warmup", "1"], env={"PSVAE_BENCH_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert '"metric"' not in r.stdout
    assert "rank 1 exited" in r.stderr


def test_more_gpus_than_the_box_has_is_refused_cleanly():
    """no GPU in the build container (and one on a gpurun box): --gpus 64 must say so and exit non-zero at once"""
    r = _run(["--gpus", "64", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 2
    assert "needs 64 GPUs" in r.stderr and r.stdout.strip() == ""
