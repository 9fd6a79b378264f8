"""CPU: `python bench.py --gpus N` started WITHOUT a launcher spawns its N ranks itself (torch.distributed.run, one fresh
process per rank), never initialises HIP in the parent, forwards the ranks' output and propagates their status.  The
dry-launch knob stops every rank right after it has joined the process group (gloo here, no GPU needed)."""
import json
import os
import subprocess
import sys

from tests.helpers import REPO


def _launch(extra_env, gpus=2):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"GPMPC_BENCH_DRY_LAUNCH": "1", "OMP_NUM_THREADS": "1"})
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1"],
                          capture_output=True, text=True, env=env, timeout=600)


def _launcher_record(stderr):
    recs = [json.loads(ln) for ln in stderr.splitlines() if ln.startswith('{"launcher"')]
    assert len(recs) == 1, stderr[-2000:]
    return recs[0]["launcher"]


def test_bench_self_launches_its_ranks():
    out = _launch({})
    assert out.returncode == 0, out.stderr[-2000:]
    marks = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith('{"dry_launch"')]
    assert sorted(m["rank"] for m in marks) == [0, 1] and all(m["world"] == 2 for m in marks)
    rec = _launcher_record(out.stderr)
    assert rec["ranks"] == 2 and rec["rc"] == 0
    assert rec["hip_initialised_in_parent"] is False            # the parent only started processes:
    assert rec["torch_imported_in_parent"] is False and rec["hip_mapped_in_parent_at_spawn"] is False   # no torch, no HIP / HSA runtime mapped
    pids = {m["pid"] for m in marks}
    assert len(pids) == 2 and rec["pid"] not in pids             # two fresh processes, neither of them the parent
    assert all(m["parent"] == str(rec["pid"]) for m in marks)


def test_bench_self_launch_propagates_a_failing_rank():
    out = _launch({"GPMPC_BENCH_DRY_FAIL_RANK": "1"})
    assert out.returncode != 0
    assert _launcher_record(out.stderr)["rc"] != 0


def test_bench_with_a_launcher_around_it_does_not_spawn_again():
    """WORLD_SIZE set (torch.distributed.run started us): no self-launch; a mismatching --gpus is refused."""
    env = dict(os.environ)
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         env=env, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr and "launcher" not in out.stderr


import pytest


@pytest.mark.gpu
def test_bench_self_launch_path_on_a_gpu_box():
    """The N > 1 launch path proven with the one GPU there is (GPMPC_BENCH_SELF_LAUNCH=1 at --gpus 1): the parent never maps
    the HIP / HSA runtime (checked in /proc/self/maps at spawn time) nor initialises torch.cuda, counts the GPUs from the
    kfd topology, the child - started by torch.distributed.run - initialises RCCL (the N > 1 code path of the bench with a
    world of one rank), prints the line, and its exit status propagates."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"GPMPC_BENCH_SELF_LAUNCH": "1", "GPMPC_BENCH_FORCE_DIST": "1"})
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--no-extra", "--cpu-sample", "0",
                          "--prewarm", "20", "--reach-ns", "512"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = _launcher_record(out.stderr)
    assert rec["ranks"] == 1 and rec["rc"] == 0 and rec["pid"] != os.getpid()
    assert rec["hip_mapped_in_parent_at_spawn"] is False and rec["hip_initialised_in_parent"] is False
    assert rec["kfd_gpu_nodes"] is None or rec["kfd_gpu_nodes"] >= 1
    lines = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["value"] > 1e6
    assert lines[0]["gather"] is not None and lines[0]["gather"]["standalone_ms"] > 0      # the RCCL path ran in the child


@pytest.mark.gpu
def test_bench_self_launch_propagates_a_failing_rank_on_a_gpu_box():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"GPMPC_BENCH_SELF_LAUNCH": "1"})
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--ns", "-5"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode != 0 and _launcher_record(out.stderr)["rc"] != 0
