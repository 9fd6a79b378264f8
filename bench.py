#!/usr/bin/env python3
"""bench.py - sampled trajectory-steps/sec of the GP-posterior-sample rollout on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Both forms work for N > 1: started WITHOUT a launcher (WORLD_SIZE unset), this process - before it has touched HIP -
starts the N ranks as fresh child processes through torch.distributed.run (one per GPU), forwards rank 0's JSON line and
exits with the children's status (`self_launch`; the reference scales out with a job array, benchmarking/euler_job.sh:5-11).

A "step" is one pass of the hot path over one batch: the full H-step re-conditioned rollout of the rank's Ns sampled
dynamics functions (one gpmpc_rollout launch) plus, for N > 1, the RCCL all-gather that assembles the reachable tube
X_traj (N*Ns, nx, H+1) on every rank.  Workload at every N: BASELINE.json configs[1] per GPU - pendulum1D
(params_pendulum1D_samples), Ns = 1024 samples per GPU, H = 30, sequential re-conditioned rollout (mode R, T = 3
value+gradient labels) - i.e. weak scaling over samples.  Inputs (training grid, base samples z, input sequence) are
synthetic and resident in HBM before the timed region; every rank generates ONLY its own shard
of the base samples (counter-based stream keyed by global sample id).

For N > 1 the all-gather runs on a side stream while the next rollouts run on the launch stream (two alternating
trajectory / tube buffers, sampling_gpmpc_amd.distributed.OverlappedTubeGather; ONE collective assembles the tubes of
`--gather-every` = 4 consecutive rollouts: enqueueing a collective costs the host ~30 us whatever its size); the timed region
ends when the last gather has completed.  `gather` in the JSON reports the collective alone and what of it stays exposed per step.

Measurement order: W warmup steps -> K steps timed COLD (`cold.ms_per_step`: what a fresh process sees; the GPU has not
reached its sustained clocks yet) -> `--prewarm` (default 2000, ~0.25 s) untimed steps -> W warmup steps -> EXACTLY K
steps between two fences (barrier + synchronize): `value` / `ms_per_step`, max over ranks.  Rank 0 prints ONE short JSON
line on stdout (the contract line, < 4 KB, last thing printed; `legs` = {leg id: [ms, roofline frac, path]}) and, before it,
the full objects of every leg as one JSON line on stderr (+ bench_extra.json).  `roofline` is for the dominant kernel:
algorithmic FP64 FLOP per launch divided by the launch duration measured with HIP events on the launch stream (one event pair per launch, in a pass of the same launches right
after the timed region).  `cpu_baseline` times the CPU oracle (the reference-faithful from-scratch algorithm, torch CPU
FP64) on rank 0 at N = 1 on a bounded sample of the same workload, at the best of several thread counts.  `extra`
(N = 1) carries BASELINE configs[2] (car, mode R) and configs[4]'s per-GPU shard (car closed loop AS SHIPPED, mode J,
SQP iterations k = 0..3) and configs[1]'s mode-J points (pendulum, k = 0 / 1), each with its own roofline object;
`end_to_end_ms` (N = 1) is host z -> HBM, rollout, X_traj -> host.  For N > 1 `extra` carries the SHARDED closed loop of
configs[4] (per-rank Agents over 1024 samples each, per SQP iteration: joint draw + Jacobians on the device, ONE gather
of the packed Jacobians to rank 0, one D2H copy; max over ranks).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np


def _wants_self_launch():
    """`--gpus N` (N > 1, or GPMPC_BENCH_SELF_LAUNCH=1) without a launcher around us: this process only starts the ranks."""
    if "WORLD_SIZE" in os.environ:
        return False
    n = 1
    for i, tok in enumerate(sys.argv):
        if tok == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif tok.startswith("--gpus="):
            n = int(tok.split("=", 1)[1])
    return n > 1 or os.environ.get("GPMPC_BENCH_SELF_LAUNCH") == "1"


# The launching parent never imports torch: `import torch` maps libamdhip64 / libhsa-runtime64 into the process (link
# dependencies of libtorch_hip), and a process with the GPU runtime in it must not fork / exec the ranks on this pool.
if __name__ != "__main__" or not _wants_self_launch():
    import torch

FP64_PEAK_TFLOPS = 78.6                                # MI355X FP64 vector == matrix peak (vendor figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ns", type=int, default=1024, help="samples per GPU (BASELINE configs[1]: 1024)")
    ap.add_argument("--horizon", type=int, default=30)
    ap.add_argument("--cpu-sample", type=int, default=256, help="samples per CPU-oracle baseline rollout (0 = skip)")
    ap.add_argument("--cpu-repeats", type=int, default=6, help="rollouts of the CPU-oracle baseline (~10 s of CPU work in all)")
    ap.add_argument("--prewarm", type=int, default=2000,
                    help="untimed steps before the second warmup that bring the GPU to its sustained clocks (the cold "
                         "region is measured before them and reported as well); 0 = off")
    ap.add_argument("--reach-ns", type=int, default=32768,
                    help="samples per GPU of the informational reachable-set leg (configs[3], mode I; 0 = skip)")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[2] / configs[4] legs")
    ap.add_argument("--cl-ns", type=int, default=1024, help="samples per GPU of the sharded closed-loop leg (N > 1)")
    ap.add_argument("--gather-every", type=int, default=4,
                    help="N > 1: rollouts per all-gather (one collective assembles the tubes of this many consecutive rollouts; "
                         "a collective costs the host ~30 us to enqueue whatever its size)")
    return ap.parse_args()


def kfd_gpu_count():
    """GPUs of this node counted WITHOUT HIP: the kfd topology nodes that have SIMDs (CPU nodes have simd_count 0).  None when
    the topology is not readable (no amdgpu driver in this container): then the ranks themselves report a missing device."""
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for f in nodes:
        try:
            props = dict(ln.split()[:2] for ln in open(f).read().splitlines() if len(ln.split()) >= 2)
        except OSError:
            continue
        n += int(props.get("simd_count", "0")) > 0
    return n


def hip_mapped_in_this_process():
    """True when the HIP / HSA runtime libraries are mapped into this process (they are loaded lazily: `import torch` alone
    does not map them on this image).  Checked right before the ranks are spawned."""
    try:
        maps = open("/proc/self/maps").read()
    except OSError:
        return None
    return ("libamdhip64" in maps) or ("libhsa-runtime64" in maps)


def self_launch(a):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as fresh processes (one per GPU, RCCL) and
    forward what they print.  Runs BEFORE anything touches HIP in this process - no torch.cuda call at all (the device count
    comes from the kfd topology in sysfs), no library load: a process that has initialised the GPU must never fork / exec
    the ranks.  GPMPC_BENCH_SELF_LAUNCH=1 takes this path at --gpus 1 too (how the path is proven on a one-GPU box)."""
    import socket
    import subprocess
    dry = os.environ.get("GPMPC_BENCH_DRY_LAUNCH") == "1"
    ngpu = None if dry else kfd_gpu_count()
    if ngpu is not None and ngpu < a.gpus:
        sys.stderr.write("bench.py: --gpus %d but %d GPU node(s) in the kfd topology\n" % (a.gpus, ngpu))
        sys.exit(2)
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC only on this pool (RCCL needs it)
    env["GPMPC_BENCH_PARENT"] = str(os.getpid())
    env.pop("GPMPC_BENCH_SELF_LAUNCH", None)                     # the ranks must not launch again
    hip_mapped = hip_mapped_in_this_process()                    # at spawn time
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:                                     # rank 0's JSON line (and the dry-launch markers)
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = proc.wait()
    sys.stderr.write(json.dumps({"launcher": {"pid": os.getpid(), "ranks": a.gpus, "port": port, "rc": rc,
                                              "kfd_gpu_nodes": ngpu, "hip_mapped_in_parent_at_spawn": hip_mapped,
                                              "torch_imported_in_parent": "torch" in sys.modules,
                                              "hip_initialised_in_parent": bool("torch" in sys.modules and
                                                                                sys.modules["torch"].cuda.is_initialized())}}) + "\n")
    sys.exit(rc if rc == 0 or 0 < rc < 256 else 1)


def roofline(flop_per_launch, kernel_ms, kernel, hbm_bytes_algorithmic, note=None, **more):
    ach = flop_per_launch / (kernel_ms * 1e-3) / 1e12
    r = {"bound": "fp64_valu", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
         "kernel": kernel, "kernel_ms": kernel_ms, "flop_per_launch": flop_per_launch,
         "algorithmic_hbm_bytes_per_launch": hbm_bytes_algorithmic,
         "algorithmic_hbm_gbps": hbm_bytes_algorithmic / (kernel_ms * 1e-3) / 1e9}
    if note:
        r["note"] = note
    r.update(more)
    return r


def cpu_baseline(Ns_cpu, H, u_ff, repeats=1):
    """Time the oracle (reference-faithful: Ns-tiled real data, dense kernel rebuild, from-scratch Cholesky per step) at
    the best of several torch thread counts (probed on a quarter-size sample), on the bounded sample."""
    from oracle import agent_oracle as ao
    import sampling_gpmpc_amd as sg
    from sampling_gpmpc_amd.workloads import fs_params

    def run(ns, threads):
        torch.set_num_threads(threads)
        p = fs_params("params_pendulum1D_samples", ns, H)
        p["agent"]["base_sample_generator"] = "counter"
        erv = sg.random_vector_within_bounds(p, 1, 3)
        agent = ao.OracleAgent(p, ao.make_oracle_env(p), erv)
        t0 = time.perf_counter()
        X = ao.forward_sampling_rollout(agent, u_ff)
        dt = time.perf_counter() - t0
        assert np.isfinite(X).all()
        return ns * H / dt, dt

    ncpu = os.cpu_count() or 8
    default_threads = torch.get_num_threads()
    probe = {}
    for th in sorted({t for t in (8, 16, 32, 64, 128) if t <= ncpu}):
        probe[th] = run(max(Ns_cpu // 4, 16), th)[0]
    best = max(probe, key=probe.get)
    dt = sum(run(Ns_cpu, best)[1] for _ in range(max(repeats, 1)))
    v = max(repeats, 1) * Ns_cpu * H / dt
    torch.set_num_threads(default_threads)
    return v, dt, best, {str(k): round(x, 1) for k, x in probe.items()}


def time_launches(fn, reps):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for k in range(reps):
        ev[k][0].record()
        fn()
        ev[k][1].record()
    torch.cuda.synchronize()
    t = [s.elapsed_time(e) for s, e in ev]
    return float(np.mean(t)), float(np.min(t))


def extra_car_rollout(sg, _lib, RolloutRunner, wl):
    """BASELINE configs[2]: params_car_residual, Ns=4096, H=40, sequential re-conditioned rollout (mode R, T=3)."""
    Ns, H = 4096, 40
    p = wl.fs_params("params_car_residual", Ns, H, nograd=False)
    p["common"]["use_cuda"] = True
    p["agent"]["base_sample_generator"] = "counter"
    agent = sg.Agent(p, sg.make_env(p))
    erv = agent.epistimic_random_vector
    per = Ns * agent.g_ny * 3
    runner = RolloutRunner(agent, wl.synthetic_u_ff(agent.nu, H), erv.reshape(-1)[per:], erv.shape[1] * per, H,
                           _lib.MODE_RECONDITIONED, False)
    for _ in range(100):
        runner.launch()
    torch.cuda.synchronize()
    ms, ms_min = time_launches(runner.launch, 30)
    bits = int(runner.info.max().item())
    ok = bool(torch.isfinite(runner.X_traj).all()) and not (bits & (_lib.INFO_TRAIN_CHOL_FAIL | _lib.INFO_ROOT_FAIL))
    flop = wl.flop_mode_r(3, 3, 45, 45, H) * Ns * H
    return {"id": "carR_4096x40", "workload": "BASELINE configs[2]: params_car_residual, mode R (T=3), Ns=4096, H=40, 1 GPU",
            "value": Ns * H / (ms * 1e-3), "unit": "trajectory-steps/s", "ms_per_rollout": ms, "finite": ok,
            "kernel_path": int(_lib.load().gpmpc_debug_last_rollout_path()),
            "roofline": roofline(flop, ms, "rollout_tiles_kernel<5,9,car_residual,32> (four chains per wave, FP64 4x4x4 MFMA solve)"
                                 if _lib.load().gpmpc_debug_last_rollout_path() == 3 else "rollout_fast_kernel<3,45,3,car_residual,grid root>",
                                 wl.min_hbm_bytes(4, 3, 3) * Ns * H,
                                 **({"bound": "fp64_mfma"} if _lib.load().gpmpc_debug_last_rollout_path() == 3 else {}))}


def extra_pendulum_throughput(sg, _lib, RolloutRunner, wl):
    """configs[1]'s workload (pendulum1D, mode R, H=30) at Ns=16384: the throughput point of the same path.  The headline
    launch has exactly one chain per SIMD (its time is one wave's latency through 30 steps); beyond one round of
    the chip the dispatcher takes the four-chains-per-wave kernel."""
    Ns, H = 16384, 30
    p = wl.fs_params("params_pendulum1D_samples", Ns, H)
    p["common"]["use_cuda"] = True
    p["agent"]["base_sample_generator"] = "counter"
    agent = sg.Agent(p, sg.make_env(p))
    erv = agent.epistimic_random_vector
    per = Ns * agent.g_ny * 3
    runner = RolloutRunner(agent, wl.synthetic_u_ff(agent.nu, H), erv.reshape(-1)[per:], erv.shape[1] * per, H,
                           _lib.MODE_RECONDITIONED, False)
    for _ in range(100):
        runner.launch()
    torch.cuda.synchronize()
    ms, ms_min = time_launches(runner.launch, 30)
    path = int(_lib.load().gpmpc_debug_last_rollout_path())
    bits = int(runner.info.max().item())
    ok = bool(torch.isfinite(runner.X_traj).all()) and not (bits & (_lib.INFO_TRAIN_CHOL_FAIL | _lib.INFO_ROOT_FAIL))
    flop = wl.flop_mode_r(1, 3, 36, 36, H) * Ns * H
    return {"id": "penR_16384x30", "workload": "BASELINE configs[1] workload at Ns=16384 (throughput point): params_pendulum1D_samples, mode R, H=30, 1 GPU",
            "value": Ns * H / (ms * 1e-3), "unit": "trajectory-steps/s", "ms_per_rollout": ms, "finite": ok, "kernel_path": path,
            "roofline": roofline(flop, ms, "rollout_tiles_kernel<4,9,pendulum1D,32> (four chains per wave, FP64 4x4x4 MFMA solve)"
                                 if path == 3 else ("rollout_one_kernel<4,pendulum1D>" if path == 4 else "rollout_fast_kernel<3,36,1,pendulum1D>"),
                                 wl.min_hbm_bytes(2, 1, 3) * Ns * H,
                                 **({"bound": "fp64_mfma"} if path == 3 else {}))}


def extra_closed_loop(sg, _lib, wl, name="params_car_residual", Ns=1024, H=40, iters=4, next_step=True, label=None, tag="carJ_1024x40"):
    """Joint draws of the closed loop (mode J), per SQP iteration k (reference src/solver.py:84-94); linearisation points
    from the deterministic surrogate of SURVEY.md 8d (sample mean of the previous iteration's prediction).
    Default: BASELINE configs[4] on its per-GPU shard, AS SHIPPED (params_car_residual.yaml incl. Dyn_gp_jitter 1e-20 -> the
    eigendecomposition root): Ns = 8192 / 8 = 1024, H = 40, k = 0..3 of MPC step 0 and of the steady-state MPC steps.  Also used
    for configs[1]'s mode-J points (pendulum, Ns = 1024, H = 30, k = 0 and k = 1; Cholesky root, jitter 1e-6).

    Every draw is timed ONCE, in the loop's own sequence (HIP events around the gpmpc_joint_sample_pending call of the real SQP
    iteration - its launches and the gaps between them, not the facade's Python around it; that is in `wall_ms_per_iteration`): from
    round 6 a draw leaves its own X / S behind as the next call's new factor rows (gpmpc_joint_sample_pending), so a draw cannot
    be repeated from a rewound cache without changing what it does.  `ms_per_draw`: the minimum over the repetitions of the same
    (MPC step class, k) - two fresh Agents, and for the steady state the MPC steps 1 and 2 of each; `wall_ms_per_iteration`: wall
    clock of the whole iteration through Agent.sqp_linearisation (train -> upload -> batch_x_hat -> draw -> Jacobians + p_lin ->
    download of p_lin), same minimum.  MPC step 0's k = 0 contains the once-per-Agent plan of the real data (~1.1 ms wall)."""
    import ctypes as C
    import warnings
    n_steps = 3 if next_step else 1
    p = wl.closed_loop_params(name, Ns, H, n_steps, iters)
    p["common"]["use_cuda"] = True
    p["agent"]["base_sample_generator"] = "counter"
    raw = _lib.load()
    ework = (C.c_ulonglong * 4)()
    best = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    xg, w = np.zeros(H), np.zeros(H)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for rep in range(3):                                      # rep 0: throw-away (process-wide first-use costs: allocator, first launches)
            agent = sg.Agent(p, sg.make_env(p))
            g_ny, T = agent.g_ny, agent.in_dim_y
            hbm_b = 8 * (2 * agent.nx + 2 * g_ny * T) * Ns * H
            x0 = np.asarray(p["env"]["start"], dtype=np.float64)[: agent.nx]
            u_h = wl.synthetic_u_ff(agent.nu, H)
            x_h = np.tile(x0, (H, Ns))
            agent._ws_cache["joint_time_events"] = (e0, e1)       # recorded by HipPosterior around gpmpc_joint_sample_pending's launches
            for step in range(n_steps if rep else min(n_steps, 2)):
                agent.mpc_iteration(step)
                for k in range(iters):
                    raw.gpmpc_debug_read_eigh_work(ework, 1)      # reset the eigh kernel's work counters
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    agent.sqp_linearisation(x_h, u_h, k, xg, w)
                    torch.cuda.synchronize()
                    wall_ms = (time.perf_counter() - t0) * 1e3
                    ms = e0.elapsed_time(e1)
                    raw.gpmpc_debug_read_eigh_work(ework, 0)
                    gp_val = agent._last_device_jacobians[0]
                    mean_next = gp_val[:, :, :, 0].mean(dim=0).T.cpu().numpy()
                    x_h = np.tile(np.vstack([x0[None, :], mean_next[:-1]]), (1, Ns))
                    if rep == 0:
                        continue
                    post = agent.model_i_call
                    info = post.last_info
                    n_ho = int(agent.model_i.h_slots.numel())
                    n_c = int(post.n_cached_rows)
                    n_r = int(agent.model_i.plan.n_r)
                    pend = bool(getattr(post, "used_pending", False))
                    key = (min(step, 1), k)
                    flop_ref = wl.flop_mode_j(g_ny, T, n_r, H, n_ho // (T * H)) * Ns       # what the reference's call computes
                    n_skip = n_ho if pend else n_c                      # factor rows this call did not have to form against old columns
                    flop = flop_ref - wl.flop_mode_j_cached_rows(g_ny, n_r, n_skip) * Ns \
                        + (wl.flop_chol(g_ny, n_ho - n_c) * Ns if pend else 0.0)             # (+ the Cholesky of the pending block)
                    eigh = bool((info & _lib.INFO_ROOT_EIGH).all().item())
                    mfma = raw.gpmpc_joint_last_path() == _lib.JOINT_MFMA
                    rec = {"mpc_step": min(step, 1), "k": k, "n_o": int(n_r + n_ho), "joint_path": "mfma" if mfma else "valu",
                           "cached_rows": n_c, "pending_rows_used": pend, "executed_flop_frac": flop / flop_ref, "ms_per_draw": ms,
                           "wall_ms_per_iteration": wall_ms, "trajectory_steps_per_s": Ns * H / (ms * 1e-3), "eigh_root": eigh,
                           "finite": bool(torch.isfinite(gp_val).all().item()),
                           "_flop": flop, "_eigh": (ework[0], ework[1], ework[2], ework[3])}
                    old = best.get(key)
                    if old is None or ms < old["ms_per_draw"]:
                        rec["wall_ms_per_iteration"] = min(wall_ms, old["wall_ms_per_iteration"]) if old else wall_ms
                        best[key] = rec
                    else:
                        old["wall_ms_per_iteration"] = min(wall_ms, old["wall_ms_per_iteration"])
            del agent
            torch.cuda.synchronize()
    out = []
    for key in sorted(best):
        rec = best[key]
        flop, ew = rec.pop("_flop"), rec.pop("_eigh")
        mfma, eigh, pend = rec["joint_path"] == "mfma", rec["eigh_root"], rec["pending_rows_used"]
        jk = (("joint_chol_mfma_kernel (pending rows: Cholesky of S + noise in place)" if pend else
               "joint_test_mfma_kernel<3> (factor extension) + joint_chol_mfma_kernel") +
              " + joint_test_mfma_kernel<3> (test rows, mean, S) + joint_kernel<3,16,1,128,4> (root + sample)") if mfma \
            else "joint_kernel<3,NB,1,NT,W>"
        rec["roofline"] = roofline(flop, rec["ms_per_draw"], jk + (" + joint_eigh_kernel<3,2>" if eigh else ""),
                                   hbm_b,
                                   bound=("fp64_mfma" if mfma else "fp64_valu"),
                                   note="joint draw incl. the facade's info reduction, timed once in the loop's own sequence; FLOP = "
                                        "SURVEY 8d mode-J formula minus the factor rows this call did not form (`cached_rows`, or all of "
                                        "them with `pending_rows_used`; `executed_flop_frac` of what the reference's call computes)"
                                        + ("; the eigendecomposition root is counted separately: `eigh`" if eigh else ""),
                                   **({"eigh": {"flop_per_launch": float(ew[0]), "mean_rank": ew[2] / max(ew[1], 1),
                                                "mean_sweeps": ew[3] / max(ew[1], 1), "chains_per_launch": float(ew[1]),
                                                "note": "work joint_eigh_kernel counted itself over the draw (csrc/joint_eigh.hpp "
                                                        "g_eigh_work); inside `ms_per_draw`, not inside `flop_per_launch`"}}
                                      if eigh else {}))
        out.append(rec)
    return {"id": tag, "workload": label or ("BASELINE configs[4] per-GPU shard as shipped: params_car_residual (Dyn_gp_jitter 1e-20), "
                                             "mode J, Ns=1024 (8192 / 8 GPUs), H=40, SQP iterations k=0..3 of MPC step 0 "
                                             "(mpc_step 0) and of the steady-state MPC steps (mpc_step 1)"), "iterations": out}


def extra_car_joint_cfg3_size(sg, _lib, wl):
    """SURVEY 8d cfg3's size (car, Ns = 4096, H = 40) in mode J: the joint draw at SQP iterations k = 0..3 of one MPC step on
    ONE GPU (the closed loop of configs[4] with four times the per-GPU shard)."""
    return extra_closed_loop(sg, _lib, wl, "params_car_residual", 4096, 40, 4, next_step=False, tag="carJ_4096x40",
                             label="SURVEY 8d cfg3 size in mode J: params_car_residual as shipped, Ns=4096, H=40 on one GPU, "
                                   "joint draw at SQP iterations k=0..3")


def extra_pendulum_joint(sg, _lib, wl):
    """SURVEY 8d cfg2, mode J: params_pendulum1D_samples as shipped (jitter 1e-6: Cholesky root), Ns=1024, H=30, k=0 and k=1."""
    return extra_closed_loop(sg, _lib, wl, "params_pendulum1D_samples", 1024, 30, 2, next_step=False, tag="penJ_1024x30",
                             label="BASELINE configs[1] in mode J (SURVEY 8d cfg2): params_pendulum1D_samples, Ns=1024, H=30, "
                                   "joint draw at SQP iterations k=0 and k=1")


def sharded_closed_loop_leg(rank, world, dist, sg, _lib, wl, ns_per_gpu=1024, H=40, iters=4):
    """BASELINE configs[4] sharded over the ranks (N > 1): params_car_residual as shipped, `ns_per_gpu` samples per GPU
    (8192 / 8 = 1024), H = 40, per-SQP-iteration GP re-conditioning (reference src/solver.py:84-131).  Every rank owns an
    Agent over its shard (distributed.make_sharded_agent: base samples keyed by global sample id); per iteration k:
    train_hallucinated_dynGP -> x_hat -> joint draw + Jacobian assembly on the device (`draw_ms`, HIP events) ->
    distributed.gather_jacobians: ONE gather of the packed (ns, nx, H, 1+nx+nu) blocks to rank 0 + one D2H copy
    (`gather_ms`, wall clock between two synchronisations) -> rank 0 plays the solver (surrogate of SURVEY 8d: next
    linearisation point = sample mean over ALL ranks' samples) and broadcasts the iterate.  MPC step 0 pays for the
    grow-only buffers (workspace, factor cache); MPC step 1 is the steady state, its k = 0 conditions on the 480 slots the
    reference's reset-after-build quirk leaves.  Every number is the max over ranks."""
    import warnings
    from sampling_gpmpc_amd.distributed import gather_jacobians, make_sharded_agent
    Ns = ns_per_gpu * world
    p = wl.closed_loop_params("params_car_residual", Ns, H, 2, iters)
    p["common"]["use_cuda"] = True
    p["agent"]["base_sample_generator"] = "counter"
    agent = make_sharded_agent(sg.Agent, p, sg.make_env(p))
    nx = agent.nx
    x0 = np.asarray(p["env"]["start"], dtype=np.float64)[:nx]
    u_h = wl.synthetic_u_ff(agent.nu, H)
    row = torch.as_tensor(np.tile(x0, (H, 1)), device="cuda")     # (H, nx): the iterate, identical for every sample
    out = []

    def rank_max(vals):
        t = torch.tensor(vals, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # as the single-process leg: two throw-away iterations on a second Agent (process-wide first-use costs: the allocator's
        # multi-GiB blocks for the workspace and the factor cache, the first launch of the matrix-pipe kernels) and one
        # throw-away gather (RCCL sets its channels up on the first collective of a kind; the persistent send / receive blocks)
        warm = make_sharded_agent(sg.Agent, p, sg.make_env(p))
        warm.mpc_iteration(0)
        xw = np.tile(x0, (H, warm.ns))
        for kw in range(2):
            warm.train_hallucinated_dynGP(kw)
            jw = warm.dyn_fg_jacobians_device(warm.get_batch_x_hat(xw, u_h), kw)
        # (two gathers whose results are alive together: the loop below holds iteration k's arrays while iteration k + 1's are
        # copied, i.e. TWO pinned staging blocks of the caching host allocator - a first-time pinned allocation costs ~25 ms)
        w1 = gather_jacobians(jw, Ns, dst=0)
        w2 = gather_jacobians(jw, Ns, dst=0)
        del warm, jw, w1, w2
        torch.cuda.synchronize()
        for step in range(2):
            agent.mpc_iteration(step)
            for k in range(iters):
                x_h = np.tile(row.cpu().numpy(), (1, agent.ns))
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                agent.train_hallucinated_dynGP(k)
                bx = agent.get_batch_x_hat(x_h, u_h)
                e0.record()
                jac = agent.dyn_fg_jacobians_device(bx, k)
                e1.record()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                full = gather_jacobians(jac, Ns, dst=0)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                if rank == 0:                                     # the solver's side: next iterate from ALL samples
                    mean_next = full[0][:, :, :, 0].mean(axis=0).T
                    row.copy_(torch.as_tensor(np.vstack([x0[None, :], mean_next[:-1]])))
                    finite = bool(np.isfinite(full[0]).all() and np.isfinite(full[1]).all() and np.isfinite(full[2]).all())
                dist.broadcast(row, src=0)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                draw_ms, gather_ms, wall_ms = rank_max([e0.elapsed_time(e1), (t2 - t1) * 1e3, (t3 - t0) * 1e3])
                if rank == 0:
                    out.append({"mpc_step": step, "k": k, "n_o": int(agent.model_i.plan.n_r + agent.model_i.h_slots.numel()),
                                "cached_rows": int(agent.model_i_call.n_cached_rows), "draw_ms": draw_ms,
                                "gather_ms": gather_ms, "wall_ms_per_iteration": wall_ms, "finite": finite,
                                "gathered_bytes": int(sum(a.nbytes for a in full)),
                                "trajectory_steps_per_s": Ns * H / (wall_ms * 1e-3)})
    return {"id": "carJ_sharded", "workload": "BASELINE configs[4] sharded over %d GPU(s): params_car_residual as shipped (Dyn_gp_jitter 1e-20), mode J, "
                        "Ns=%d (%d per GPU), H=%d, SQP iterations k=0..%d of MPC steps 0 (cold: buffers grow) and 1 (steady)"
                        % (world, Ns, ns_per_gpu, H, iters - 1),
            "collective": "gather of the packed Jacobians (ns, nx, H, 1+nx+nu) f64 to rank 0 (RCCL) + one D2H copy",
            "iterations": out}


def reach_roofline(wl, Ns, kms):
    """Mode I roofline by the algorithm the kernel executes (grid root: 1.66e3 FLOP per trajectory-step, workloads.
    flop_mode_i_grid_root) AND by bytes (112 B per trajectory-step): the leg is closer to the HBM roof."""
    units = Ns * 40
    r = roofline(wl.flop_mode_i_grid_root(3, 5, 9) * units, kms, "rollout_indep_grid_kernel<car,5,9,3>",
                 wl.min_hbm_bytes(4, 3, 1) * units,
                 note="FLOP = the grid-root algorithm the kernel executes (3 x 554 per trajectory-step); SURVEY 8d's count "
                      "of the triangular algorithm it replaces is 8.3e3 (`survey_flop_equiv_tflops`, not a roofline fraction)")
    r["hbm_frac"] = r["algorithmic_hbm_gbps"] / 8000.0
    r["survey_flop_equiv_tflops"] = wl.flop_mode_i(3, 45) * units / (kms * 1e-3) / 1e12
    r["bound"] = "hbm" if r["hbm_frac"] > r["frac"] else "fp64_valu"
    return r


def end_to_end_ms(runner, agent, reps=20):
    """SURVEY 8d "end-to-end": pinned host z -> HBM, one rollout launch, X_traj -> pinned host; wall clock, median."""
    erv = agent.epistimic_random_vector
    z_host = erv.cpu().pin_memory()
    x_host = torch.empty(runner.X_traj.shape, dtype=runner.X_traj.dtype).pin_memory()
    ts = []
    for _ in range(reps + 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        erv.copy_(z_host, non_blocking=True)
        runner.launch()
        x_host.copy_(runner.X_traj, non_blocking=True)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts[3:])), int(z_host.numel() * 8), int(x_host.numel() * 8)


def reachable_set_leg(a, rank, world, dist, sg, _lib, RolloutRunner, wl):
    """Second, informational leg (not `value`): BASELINE.json's "reachable-set wall-clock" on configs[3] as shipped
    (params_car_residual_fs, forward sampling on the real data only = mode I, T = 1, H = 40), 32768 samples per GPU
    (the per-GPU shard of Ns = 262144 on 8 GPUs): one rollout launch + the all-gather of the tube per repetition,
    base samples of the rank's shard resident in HBM, max over ranks.  Any failure is reported in the JSON instead of
    raised; the ranks agree on skipping before they enter the collective."""
    if a.reach_ns <= 0:
        return None
    res, ok, runner, tube = {"workload": "BASELINE configs[3] as shipped: params_car_residual_fs, mode I (T=1), "
                             "Ns=%d per GPU, H=40; rollout + all-gather of X_traj" % a.reach_ns}, 1, None, None
    try:
        Ns, H = a.reach_ns, 40
        p = wl.fs_params("params_car_residual_fs", Ns, H, nograd=True)
        p["common"]["use_cuda"] = True
        p["agent"]["base_sample_generator"] = "counter"
        p["agent"]["base_sample_offset"] = rank * Ns           # global sample id of this rank's first sample
        agent = sg.Agent(p, sg.make_env(p))
        u_ff = wl.synthetic_u_ff(agent.nu, H)
        erv = agent.epistimic_random_vector
        per = Ns * agent.g_ny
        runner = RolloutRunner(agent, u_ff, erv.reshape(-1)[per:], erv.shape[1] * per, H, _lib.MODE_INDEPENDENT, True)
        if world > 1:
            tube = torch.empty(world * Ns, agent.nx, H + 1, dtype=torch.float64, device="cuda")
        runner.launch()
        torch.cuda.synchronize()
    except Exception as e:                                    # noqa: BLE001 - reported, never fatal for the bench line
        ok, res["error"] = 0, repr(e)[:300]
    if world > 1:
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item())
    if not ok:
        res.setdefault("error", "skipped: another rank failed to set the workload up")
        return res
    reps = 50

    def one():
        X = runner.launch()
        if world > 1:
            dist.all_gather_into_tensor(tube, X)

    try:
        for _ in range(50):
            one()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            one()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([wall], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        finite = bool(torch.isfinite(runner.X_traj).all())
        kms, _ = time_launches(runner.launch, 20)
    except Exception as e:                                    # noqa: BLE001 - the headline line must still be printed
        res["error"] = repr(e)[:300]
        return res
    res.update({"Ns_per_gpu": a.reach_ns, "Ns_total": world * a.reach_ns, "H": 40, "n_gpus": world, "reps": reps,
                "wallclock_ms": wall / reps * 1e3, "trajectory_steps_per_s": world * a.reach_ns * 40 * reps / wall,
                "finite": finite,
                "roofline": reach_roofline(wl, a.reach_ns, kms)})
    return res


HEADLINE_MAX_CHARS = 4096          # the contract line: printed LAST, shorter than this (tests/test_bench_contract.py)


def _dedupe_strings(obj, legend, min_len=48):
    """Every string of `min_len`+ characters below `obj` becomes a short id ("@3") into `legend` (id -> text): the same
    note / kernel description is carried once however many legs repeat it."""
    if isinstance(obj, dict):
        return {k: _dedupe_strings(v, legend, min_len) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_dedupe_strings(v, legend, min_len) for v in obj]
    if isinstance(obj, str) and len(obj) >= min_len:
        for k, v in legend.items():
            if v == obj:
                return k
        k = "@%d" % len(legend)
        legend[k] = obj
        return k
    return obj


def _r4(x):
    return float("%.5g" % x) if isinstance(x, float) else x


def leg_summary(extra, reach):
    """{short id: [ms, roofline frac]} of every informational leg - what the review reads first, kept on the headline line."""
    legs = {}
    for leg in extra or []:
        if not isinstance(leg, dict) or "error" in leg:
            legs["error:" + str(leg.get("workload"))[:40]] = None
            continue
        tag = leg.get("id", "leg")
        if "iterations" in leg:
            for it in leg["iterations"]:
                ms = it.get("ms_per_draw", it.get("draw_ms"))
                fr = it.get("roofline", {}).get("frac")
                legs["%s.s%dk%d" % (tag, it["mpc_step"], it["k"])] = [_r4(ms), _r4(fr) if fr is not None else _r4(it.get("wall_ms_per_iteration"))]
        else:
            legs[tag] = [_r4(leg.get("ms_per_rollout")), _r4(leg.get("roofline", {}).get("frac")), leg.get("kernel_path")]
    if reach and "roofline" in reach:
        legs["carI_%dx40" % reach["Ns_per_gpu"]] = [_r4(reach["roofline"]["kernel_ms"]), _r4(reach["roofline"]["frac"]),
                                                     _r4(reach["roofline"]["hbm_frac"])]
    return legs


def split_for_driver(out):
    """(extras, headline): `headline` is the driver's contract line - every contract key, `roofline`, `cpu_baseline`, `gather`,
    `cold`, and a compact `legs` table - below HEADLINE_MAX_CHARS; `extras` is everything else in full (notes deduplicated
    into one legend): its own line on stderr BEFORE the headline, and bench_extra.json beside this script."""
    head_keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data")
    head = {k: out[k] for k in head_keys}
    cfg = out["config"]
    head["config"] = {"workload": "BASELINE configs[1]: params_pendulum1D_samples, mode R (T=3), Ns=%d per GPU, H=%d"
                                  % (cfg["Ns_per_gpu"], cfg["H"]),
                      "Ns_per_gpu": cfg["Ns_per_gpu"], "H": cfg["H"], "Ns_total": cfg["Ns_total"],
                      "parallelism": "samples sharded over %d GPU(s), all-gather of X_traj per rollout" % out["n_gpus"],
                      "device": cfg["device"], "cus": cfg["cus"]}
    r = out["roofline"]
    head["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms",
                                                   "flop_per_launch", "algorithmic_hbm_bytes_per_launch", "kernel_path",
                                                   "mfma_busy", "kernel_ms_per_launch_event_pairs")}
    head["roofline"]["kernel"] = r["kernel"].split(" (")[0]
    c = out.get("cpu_baseline")
    head["cpu_baseline"] = None if c is None else {"value": _r4(c["value"]), "unit": c["unit"], "cores": c["cores"], "kind": c["kind"],
                                                   "sample": "%d rollouts of Ns=%d, full H, %.1f s of CPU work (oracle, torch CPU FP64)"
                                                             % (c["_repeats"], c["_ns"], c["_dt"]),
                                                   "host_cpus": c["host_cpus"]}
    g = out.get("gather")
    head["gather"] = None if g is None else {k: _r4(g[k]) for k in ("every", "bytes_per_rank", "standalone_ms", "exposed_ms_per_step")}
    head["cold"] = {"ms_per_step": _r4(out["cold"]["ms_per_step"]), "value": _r4(out["cold"]["value"])}
    head["prewarm_steps"] = out["prewarm_steps"]
    head["legs"] = leg_summary(out.get("extra"), out.get("reachable_set"))
    head["legs_columns"] = "id: [ms, roofline frac (wall ms for sharded legs), path]; full objects: bench_extra line on stderr / bench_extra.json"
    legend = {}
    ext = {k: v for k, v in out.items() if k not in head_keys}
    if ext.get("cpu_baseline"):
        ext["cpu_baseline"] = {k: v for k, v in ext["cpu_baseline"].items() if not k.startswith("_")}
    ext = _dedupe_strings(ext, legend)
    extras = {"bench_extra": ext, "legend": legend}
    if len(json.dumps(head)) >= HEADLINE_MAX_CHARS:              # never let the contract line grow past the window again
        head.pop("legs_columns", None)
        while len(json.dumps(head)) >= HEADLINE_MAX_CHARS and head["legs"]:
            head["legs"].popitem()
    return extras, head


def emit(out):
    """stdout carries exactly ONE JSON line, the short contract line (the form the driver parsed in rounds 1-4), printed last;
    the extras object goes, as one JSON line of its own, to stderr BEFORE it and to bench_extra.json (GPMPC_BENCH_EXTRAS=stdout
    puts it on stdout instead, still before the contract line)."""
    extras, head = split_for_driver(out)
    text = json.dumps(extras)
    for d in (REPO, os.path.join(REPO, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_extra.json"), "w") as f:
                    f.write(text + "\n")
        except OSError:
            pass
    dst = sys.stdout if os.environ.get("GPMPC_BENCH_EXTRAS") == "stdout" else sys.stderr
    dst.write(text + "\n")
    dst.flush()
    sys.stdout.write(json.dumps(head) + "\n")
    sys.stdout.flush()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or os.environ.get("GPMPC_BENCH_SELF_LAUNCH") == "1"):
        self_launch(a)                                           # never returns
    if os.environ.get("GPMPC_BENCH_DRY_LAUNCH") == "1" and world > 1:
        # launch-path check (tests, CPU): every rank joins the group, reports, leaves - nothing else runs
        import torch.distributed as dist_
        dist_.init_process_group(backend="gloo")
        # one write per record: the ranks share the launcher's pipe, print() would emit the newline separately
        os.write(1, (json.dumps({"dry_launch": True, "rank": rank, "world": world, "pid": os.getpid(),
                                 "parent": os.environ.get("GPMPC_BENCH_PARENT")}) + "\n").encode())
        dist_.barrier()
        dist_.destroy_process_group()
        sys.exit(3 if os.environ.get("GPMPC_BENCH_DRY_FAIL_RANK") == str(rank) else 0)
    if a.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    # GPMPC_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, per-shard base samples, pipelined all-gather) with a
    # world of ONE rank - lets the multi-GPU path be exercised on a 1-GPU box (tests / smoke runs; not a scaling number)
    force_dist = os.environ.get("GPMPC_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if force_dist and world == 1:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    multi = dist is not None

    import sampling_gpmpc_amd as sg
    from sampling_gpmpc_amd import _lib, workloads as wl
    from sampling_gpmpc_amd.distributed import OverlappedTubeGather
    from sampling_gpmpc_amd.rollout import RolloutRunner

    Ns, H = a.ns, a.horizon
    p = wl.fs_params("params_pendulum1D_samples", Ns, H)
    p["common"]["use_cuda"] = True
    p["agent"]["base_sample_generator"] = "counter"       # the rank draws ITS shard only, keyed by global sample id
    p["agent"]["base_sample_offset"] = rank * Ns
    agent = sg.Agent(p, sg.make_env(p))
    u_ff = wl.synthetic_u_ff(1, H)
    erv = agent.epistimic_random_vector         # (H, 2, Ns, 1, 1, 3) on the device: this rank's samples only
    per_slab = Ns * 3
    z = erv.reshape(-1)[per_slab:]
    runner = RolloutRunner(agent, u_ff, z, erv.shape[1] * per_slab, H, _lib.MODE_RECONDITIONED, False)
    pipe = OverlappedTubeGather(Ns, agent.nx, H, every=a.gather_every) if multi else None

    def run_steps(n, r0=0):
        """n steps; for N > 1: rollout r on the launch stream, its all-gather on the side stream (overlaps rollout r+1)"""
        if pipe is None:
            for _ in range(n):
                runner.launch()
            return
        for r in range(r0, r0 + n):
            pipe.before_rollout(r)
            runner.launch(out=pipe.buffer(r))
            pipe.submit(r)
        pipe.finish()

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n):
        fence()
        t0 = time.perf_counter()
        run_steps(n)
        fence()
        wall = time.perf_counter() - t0
        if multi:
            t = torch.tensor([wall], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        return wall

    # cold: what a fresh process sees (W warmup steps, then K timed steps, clocks not yet ramped)
    run_steps(a.warmup)
    wall_cold = timed(a.steps)
    # clock ramp: the same step, a fixed count on every rank (it contains the collective), untimed and reported
    run_steps(max(a.prewarm, 0))
    fence()
    run_steps(a.warmup)
    # timed region: EXACTLY a.steps steps between two fences
    wall = timed(a.steps)
    # the rollout kernel's launch duration, one HIP event pair per launch (what rocprofv3 --kernel-trace reports per
    # dispatch): a second, untimed pass of the same launches, so that the instrumentation stays out of the timed region
    kern_ms_pairs, _ = time_launches(runner.launch, a.steps)
    # a launch cannot take longer than a step of the (un-instrumented) timed region it is part of: at N = 1 the region per
    # step is kernel + launch gap, and the event pairs of the second pass add a few microseconds of their own
    kern_ms = min(kern_ms_pairs, wall / a.steps * 1e3) if not multi else kern_ms_pairs
    head_path = int(_lib.load().gpmpc_debug_last_rollout_path())   # 4: rollout_one_kernel, 1: rollout_fast_kernel, 3: rollout_tiles_kernel
    gather = None
    if multi:
        tube, src = pipe.T[0], pipe.X[0]                         # one collective as the timed loop issues it: `every` rollouts' tubes
        for _ in range(20):
            dist.all_gather_into_tensor(tube, src)
        fence()
        g_ms, _ = time_launches(lambda: dist.all_gather_into_tensor(tube, src), 50)
        gather = {"collective": "all_gather_into_tensor of (every, Ns, nx, H+1) f64 shards, RCCL, one per `every` rollouts",
                  "every": pipe.every, "bytes_per_rank": Ns * agent.nx * (H + 1) * 8, "bytes_per_collective_per_rank": src.numel() * 8,
                  "standalone_ms": g_ms, "standalone_ms_per_step": g_ms / pipe.every,
                  "exposed_ms_per_step": max(wall / a.steps * 1e3 - kern_ms, 0.0), "overlapped_with_next_rollout": True}
    X_last = runner.X_traj if pipe is None else pipe.buffer(a.steps - 1)
    bits = int(runner.info.max().item())
    assert torch.isfinite(X_last).all() and not (bits & (_lib.INFO_TRAIN_CHOL_FAIL | _lib.INFO_ROOT_FAIL)), bits

    reach = reachable_set_leg(a, rank, world if not (force_dist and world == 1) else 1, dist if world > 1 else None, sg, _lib,
                              RolloutRunner, wl)
    e2e = end_to_end_ms(runner, agent) if (rank == 0 and not multi) else None
    sharded_cl = None
    if multi and not a.no_extra:                                  # every rank takes part (collectives inside)
        ok = 1
        try:
            sharded_cl = sharded_closed_loop_leg(rank, world, dist, sg, _lib, wl, ns_per_gpu=a.cl_ns)
        except Exception as e:                                    # noqa: BLE001 - never fatal for the bench line
            ok, sharded_cl = 0, {"workload": "sharded_closed_loop_leg", "error": repr(e)[:300]}
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if not int(flag.item()) and ok:
            sharded_cl = {"workload": "sharded_closed_loop_leg", "error": "another rank failed"}

    if rank == 0:
        name, cus, _ = _lib.device_info(local_rank)
        units = world * Ns * H                                  # sampled trajectory-steps per step
        prof, tf = {}, os.path.join(REPO, "profiles", "latest_traffic.json")
        if os.path.exists(tf) and Ns == 1024 and H == 30:     # PMC counters cannot be read in-process: last profiled run
            prof = json.load(open(tf))
        traffic = prof.get("hbm_bytes_per_launch_gfx950_corrected")
        flop = wl.flop_mode_r(1, 3, 36, 36, H) * Ns * H         # per launch (one GPU): 2.55e4 per trajectory-step at H = 30
        out = {
            "metric": "sampled trajectory-steps/sec (Ns*H per wall second)",
            "value": units * a.steps / wall,
            "unit": "trajectory-steps/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "prewarm_steps": max(a.prewarm, 0),     # untimed clock-ramp steps between the cold and the reported region
            "ms_per_step": wall / a.steps * 1e3,
            "cold": {"ms_per_step": wall_cold / a.steps * 1e3, "value": units * a.steps / wall_cold,
                     "note": "the same K steps right after the W warmup steps of a fresh process, before the clock ramp"},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: params_pendulum1D_samples, sequential re-conditioned rollout "
                                   "(mode R, value+gradient labels T=3), Ns=%d per GPU, H=%d" % (Ns, H),
                       "Ns_per_gpu": Ns, "H": H, "Ns_total": world * Ns,
                       "parallelism": "samples sharded over %d GPU(s); base samples generated per shard; RCCL all-gather of "
                                      "X_traj per rollout on a side stream, overlapping the next rollout" % world,
                       "device": name, "cus": cus},
            "roofline": roofline(
                flop, kern_ms, {4: "rollout_one_kernel<4,pendulum1D> (one chain per wave, the factor in AGPR-pinned MFMA panels)",
                                3: "rollout_tiles_kernel<4,9,pendulum1D,32>"}.get(
                                    head_path, "rollout_fast_kernel<3,36,1,pendulum1D,L_hh in LDS,grid root>"),
                wl.min_hbm_bytes(2, 1, 3) * Ns * H,
                note=("FP64 matrix pipe (v_mfma_f64_4x4x4_4b_f64: forward substitution, real-block correction and Gram products; "
                      "FP64 MFMA and FP64 vector peak are the same 78.6 TFLOP/s on MI355X) + FP64 vector FMA for the kernel entries "
                      "and the 3 x 3 roots; " if head_path in (3, 4) else
                      "FP64 vector FMA (the FP64 MFMA peak is the same 78.6 TFLOP/s on MI355X), no MFMA is issued; ") +
                     "algorithmic FLOP = 2.55e4 per trajectory-step (SURVEY 8d) x Ns x H; min HBM traffic 80 B per "
                     "trajectory-step: one wave per SIMD, latency / issue bound, not HBM bound",
                kernel_path=head_path,
                **({"bound": "mfma"} if head_path in (3, 4) else {}),
                traffic=traffic, traffic_source=prof.get("source"),
                hbm_gbps=(traffic / (kern_ms * 1e-3) / 1e9) if traffic else None,
                hbm_frac_of_8TBps=(traffic / (kern_ms * 1e-3) / 8e12) if traffic else None,
                # executed work (SQ counters of the last profiled run): VALU instructions x 64 lanes x 2 FLOP as if every one
                # were an FP64 FMA, over the measured launch time - how busy the FP64 pipe is with ANY instruction
                executed_flop_frac=(prof["valu_insts_per_launch"] * 128 / (kern_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS)
                if prof.get("valu_insts_per_launch") else None,
                valu_active_frac=prof.get("valu_active_frac"),
                mfma_busy=prof.get("mfma_busy_frac", 0.0 if head_path == 1 else None),
                kernel_ms_per_launch_event_pairs=kern_ms_pairs),
            "gather": gather,
        }
        if world == 1 and a.cpu_sample > 0:
            v, dt, th, probe = cpu_baseline(a.cpu_sample, H, u_ff, a.cpu_repeats)
            out["cpu_baseline"] = {"value": v, "unit": "trajectory-steps/s", "cores": th, "kind": "port",
                                   "sample": "same workload, %d rollouts of Ns=%d of %d samples, full H=%d horizon, %.1f s "
                                             "of CPU work at the best of the probed thread counts (oracle: reference-faithful "
                                             "from-scratch batched Cholesky per step, torch CPU FP64; gpytorch itself is not "
                                             "installable on the box)" % (max(a.cpu_repeats, 1), a.cpu_sample, Ns, H, dt),
                                   "thread_probe_steps_per_s": probe, "host_cpus": os.cpu_count(),
                                   "_repeats": max(a.cpu_repeats, 1), "_ns": a.cpu_sample, "_dt": dt}
        else:
            out["cpu_baseline"] = None
        out["reachable_set"] = reach
        if e2e is not None:
            out["end_to_end_ms"] = {"ms": e2e[0], "h2d_bytes": e2e[1], "d2h_bytes": e2e[2], "value": units / (e2e[0] * 1e-3),
                                    "note": "pinned host z -> HBM + one rollout + X_traj -> pinned host, wall clock, median of 20 "
                                            "(SURVEY 8d end-to-end; never `value`)"}
        if sharded_cl is not None:
            out["extra"] = [sharded_cl]
        if not multi and not a.no_extra:
            extra = []
            for fn, args in ((extra_car_rollout, (sg, _lib, RolloutRunner, wl)), (extra_closed_loop, (sg, _lib, wl)),
                             (extra_pendulum_joint, (sg, _lib, wl)), (extra_pendulum_throughput, (sg, _lib, RolloutRunner, wl)),
                             (extra_car_joint_cfg3_size, (sg, _lib, wl))):
                try:
                    extra.append(fn(*args))
                except Exception as e:                            # noqa: BLE001 - never fatal for the bench line
                    extra.append({"workload": fn.__name__, "error": repr(e)[:300]})
            out["extra"] = extra
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the contract line is the LAST thing on stdout: the process group is gone, and whatever the C side (RCCL's version banner)
        # still holds in its stdio buffer is flushed in front of it
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        emit(out)


if __name__ == "__main__":
    main()
