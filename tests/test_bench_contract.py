"""GPU: bench.py prints ONE short JSON line on stdout (the driver's contract keys, the roofline and cpu_baseline objects, the
compact `legs` table; < 4096 characters, the last stdout line) and the full objects of every leg as one JSON line on stderr; with
GPMPC_BENCH_FORCE_DIST=1 the N > 1 code path (process group, per-shard base samples, pipelined all-gather) runs with a
world of one rank."""
import json
import os
import subprocess
import sys

import pytest

from tests.helpers import REPO

pytestmark = pytest.mark.gpu


def _run(extra_env=None, args=()):
    env = dict(os.environ)
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "5", "--warmup", "2", "--prewarm", "20",
                          "--cpu-sample", "16", "--reach-ns", "512", *args], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    all_lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    lines = [ln for ln in all_lines if ln.startswith("{")]
    assert len(lines) == 1 and all_lines[-1] == lines[0], out.stdout[-2000:]       # the contract line is the LAST stdout line
    assert len(lines[0]) < 4096, len(lines[0])                                       # ... and short (round 5's 20 KB line did not parse)
    d = json.loads(lines[0])
    ext = [json.loads(ln) for ln in out.stderr.splitlines() if ln.startswith('{"bench_extra"')]
    assert len(ext) == 1                                                             # the extras line parses too
    legend = ext[0]["legend"]

    def expand(o):
        if isinstance(o, dict):
            return {k: expand(v) for k, v in o.items()}
        if isinstance(o, list):
            return [expand(v) for v in o]
        return legend.get(o, o) if isinstance(o, str) else o

    full = expand(ext[0]["bench_extra"])
    assert not (set(full) & {"metric", "value"})                                     # nothing a parser could mistake for the headline
    d["_full"] = full
    return d


def test_bench_line_contract():
    d = _run(args=("--no-extra",))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "cold"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["kernel_path"] == 4 and "rollout_one_kernel" in r["kernel"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["kernel_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert d["value"] > 1e6 and d["cold"]["ms_per_step"] > 0
    assert set(d["legs"]) == {"carI_512x40"} and d["legs"]["carI_512x40"][0] > 0
    rs = d["_full"]["reachable_set"]
    assert rs["finite"] and rs["roofline"]["frac"] <= 1.0 and 0.0 < rs["roofline"]["hbm_frac"] <= 1.0
    e = d["_full"]["end_to_end_ms"]
    assert e["ms"] >= d["ms_per_step"] and e["h2d_bytes"] == 30 * 2 * 1024 * 3 * 8 and e["d2h_bytes"] == 1024 * 2 * 31 * 8


def test_bench_multi_gpu_path_with_one_rank():
    d = _run({"GPMPC_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29547"}, args=("--no-extra", "--cpu-sample", "0"))
    g = d["gather"]
    assert g is not None and g["every"] == 4 and g["standalone_ms"] > 0 and g["bytes_per_rank"] == 1024 * 2 * 31 * 8
    assert d["_full"]["gather"]["overlapped_with_next_rollout"] and d["_full"]["gather"]["bytes_per_collective_per_rank"] == 4 * 1024 * 2 * 31 * 8
    assert d["value"] > 1e6


def test_bench_sharded_closed_loop_leg_with_one_rank():
    """The N > 1 leg of configs[4]: per-rank Agent over its shard, per SQP iteration draw + device gather of the packed
    Jacobians + one D2H copy on rank 0 (here a world of one rank, 32 samples)."""
    d = _run({"GPMPC_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29549"}, args=("--cpu-sample", "0", "--cl-ns", "32"))
    legs = d["_full"]["extra"]
    assert [k.split(".")[0] for k in d["legs"]][:8] == ["carJ_sharded"] * 8
    assert len(legs) == 1 and "error" not in legs[0], legs
    its = legs[0]["iterations"]
    assert [(i["mpc_step"], i["k"]) for i in its] == [(s, k) for s in range(2) for k in range(4)]
    assert all(i["finite"] and i["draw_ms"] > 0 and i["gather_ms"] > 0 and i["wall_ms_per_iteration"] >= i["draw_ms"] for i in its)
    assert its[0]["n_o"] == 45 and its[3]["n_o"] == 45 + 3 * 120 and its[4]["n_o"] == 45 + 4 * 120    # the reset-after-build quirk
    assert its[0]["gathered_bytes"] == 32 * 4 * 40 * 7 * 8
