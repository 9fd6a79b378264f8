#!/usr/bin/env python3
"""bench.py - sampled trajectory-steps/sec of the GP-posterior-sample rollout on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: the full H-step re-conditioned rollout of the rank's Ns sampled
dynamics functions (one gpmpc_rollout launch) plus, for N > 1, the RCCL all-gather that assembles the reachable tube
X_traj (N*Ns, nx, H+1) on every rank.  Workload at every N: BASELINE.json configs[1] per GPU - pendulum1D
(params_pendulum1D_samples), Ns = 1024 samples per GPU, H = 30, sequential re-conditioned rollout (mode R, T = 3
value+gradient labels) - i.e. weak scaling over samples.  Inputs (training grid, base samples z, input sequence)
are synthetic and resident in HBM before the timed region.

Before the W warmup steps the same step runs `--prewarm` (default 2000, ~0.25 s) more untimed times: a cold GPU needs
that long to reach its sustained clocks, and a 50-step region measured right after 5 warmup steps reads ~5 % slower than
the steady state every longer run sees (tools/clock_check.py: 0.1166 vs 0.1087 ms per step); the count is reported as
`prewarm_steps`.  Prints ONE JSON line on rank 0 (see the keys below).  `roofline` is for the dominant kernel (rollout_kernel):
algorithmic FP64 FLOP per launch (2.55e4 FLOP per trajectory-step for this config) divided by
the launch duration measured with HIP events on the launch stream (one event pair per launch, in a pass of the same
launches right after the timed region; the timed region itself carries one event at either end).  `cpu_baseline` times the CPU oracle (the
reference-faithful from-scratch algorithm, torch CPU FP64) on rank 0 at N = 1 on a bounded sample of the same
workload.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np
import torch

FLOP_PER_TRAJ_STEP = {"pendulum1D_R_H30": 2.55e4}     # SURVEY.md section 8(d), append-row algorithm
MIN_HBM_BYTES_PER_TRAJ_STEP = {"pendulum1D_R_H30": 80}
FP64_PEAK_TFLOPS = 78.6                                # MI355X FP64 vector == matrix peak (vendor figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ns", type=int, default=1024, help="samples per GPU (BASELINE configs[1]: 1024)")
    ap.add_argument("--horizon", type=int, default=30)
    ap.add_argument("--cpu-sample", type=int, default=256, help="samples of the CPU-oracle baseline (0 = skip)")
    ap.add_argument("--prewarm", type=int, default=2000,
                    help="untimed steps BEFORE the W warmup steps that bring the GPU to its sustained clocks (a cold "
                         "50-step region measures ~5 %% slower than steady state: tools/clock_check.py); 0 = off")
    ap.add_argument("--reach-ns", type=int, default=32768,
                    help="samples per GPU of the informational reachable-set leg (configs[3], mode I; 0 = skip)")
    return ap.parse_args()


def cpu_baseline(Ns_cpu, H, u_ff):
    """Time the oracle (reference-faithful: Ns-tiled real data, dense kernel rebuild, from-scratch Cholesky per step)."""
    from oracle import agent_oracle as ao
    from tests.helpers import fs_params
    import sampling_gpmpc_amd as sg
    p = fs_params("params_pendulum1D_samples", Ns_cpu, H)
    p["agent"]["base_sample_generator"] = "vectorized"
    torch.manual_seed(99)
    erv = sg.random_vector_within_bounds(p, 1, 3)
    agent = ao.OracleAgent(p, ao.make_oracle_env(p), erv)
    t0 = time.perf_counter()
    X = ao.forward_sampling_rollout(agent, u_ff)
    dt = time.perf_counter() - t0
    assert np.isfinite(X).all()
    return Ns_cpu * H / dt, dt


def reachable_set_leg(a, rank, world, dist, sg, _lib, RolloutRunner, fs_params, synthetic_u_ff):
    """Second, informational leg (not `value`): BASELINE.json's "reachable-set wall-clock" on configs[3] as shipped
    (params_car_residual_fs, forward sampling on the real data only = mode I, T = 1, H = 40), 32768 samples per GPU
    (the per-GPU shard of Ns = 262144 on 8 GPUs): one rollout launch + the all-gather of the tube per repetition,
    base samples resident in HBM, max over ranks.  Any failure is reported in the JSON instead of raised; the ranks
    agree on skipping before they enter the collective."""
    if a.reach_ns <= 0:
        return None
    res, ok, runner, tube = {"workload": "BASELINE configs[3] as shipped: params_car_residual_fs, mode I (T=1), "
                             "Ns=%d per GPU, H=40; rollout + all-gather of X_traj" % a.reach_ns}, 1, None, None
    try:
        Ns, H = a.reach_ns, 40
        p = fs_params("params_car_residual_fs", Ns, H, nograd=True)
        p["common"]["use_cuda"] = True
        p["agent"]["base_sample_generator"] = "vectorized"
        torch.manual_seed(777 + rank)
        agent = sg.Agent(p, sg.make_env(p))
        u_ff = synthetic_u_ff(agent.nu, H)
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
    except Exception as e:                                    # noqa: BLE001 - the headline line must still be printed
        res["error"] = repr(e)[:300]
        return res
    res.update({"Ns_per_gpu": a.reach_ns, "Ns_total": world * a.reach_ns, "H": 40, "n_gpus": world, "reps": reps,
                "wallclock_ms": wall / reps * 1e3, "trajectory_steps_per_s": world * a.reach_ns * 40 * reps / wall,
                "finite": finite, "kernel": "rollout_indep_grid_kernel<car,5,9,3>"})
    return res


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N > 1 with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import sampling_gpmpc_amd as sg
    from sampling_gpmpc_amd import _lib
    from sampling_gpmpc_amd.rollout import RolloutRunner
    from tests.helpers import fs_params, synthetic_u_ff

    Ns, H = a.ns, a.horizon
    p = fs_params("params_pendulum1D_samples", Ns, H)
    p["common"]["use_cuda"] = True
    p["agent"]["base_sample_generator"] = "vectorized"
    torch.manual_seed(123456 + rank)            # every rank owns different samples (global sample id = rank*Ns + s)
    agent = sg.Agent(p, sg.make_env(p))
    u_ff = synthetic_u_ff(1, H)
    erv = agent.epistimic_random_vector         # (H, 2, Ns, 1, 1, 3) on the device
    per_slab = Ns * 3
    z = erv.reshape(-1)[per_slab:]
    runner = RolloutRunner(agent, u_ff, z, erv.shape[1] * per_slab, H, _lib.MODE_RECONDITIONED, False)
    tube = torch.empty(world * Ns, agent.nx, H + 1, dtype=torch.float64, device="cuda") if world > 1 else None

    def step():
        X = runner.launch()
        if world > 1:
            dist.all_gather_into_tensor(tube, X)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # clock ramp: the same step, a fixed count on every rank (it contains the collective), untimed and reported
    for _ in range(max(a.prewarm, 0)):
        step()
    fence()
    for _ in range(a.warmup):
        step()
    fence()
    # timed region: EXACTLY a.steps steps between two fences; one HIP event at either end on the launch stream (per-launch
    # events inside the loop cost ~6 us per step: every record is a queue packet the next dispatch waits for)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for k in range(a.steps):
        X = runner.launch()
        if world > 1:
            dist.all_gather_into_tensor(tube, X)
    ev1.record()
    fence()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    region_ms = ev0.elapsed_time(ev1) / a.steps             # per step over the timed region (includes the collective)
    # the rollout kernel's launch duration, one HIP event pair per launch (what rocprofv3 --kernel-trace reports per
    # dispatch): a second, untimed pass of the same launches, so that the instrumentation stays out of the timed region
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    for k in range(a.steps):
        ev[k][0].record()
        runner.launch()
        ev[k][1].record()
    torch.cuda.synchronize()
    kern_ms_pairs = float(np.mean([s.elapsed_time(e) for s, e in ev]))
    # a launch cannot take longer than a step of the (un-instrumented) timed region it is part of: at N = 1 the region per
    # step is kernel + launch gap, and the event pairs of the second pass add a few microseconds of their own
    kern_ms = min(kern_ms_pairs, region_ms) if world == 1 else kern_ms_pairs
    bits = int(runner.info.max().item())
    assert torch.isfinite(runner.X_traj).all() and not (bits & (_lib.INFO_TRAIN_CHOL_FAIL | _lib.INFO_ROOT_FAIL)), bits

    reach = reachable_set_leg(a, rank, world, dist, sg, _lib, RolloutRunner, fs_params, synthetic_u_ff)

    if rank == 0:
        name, cus, _ = _lib.device_info(local_rank)
        units = world * Ns * H                                  # sampled trajectory-steps per step
        traffic, traffic_src = None, None
        tf = os.path.join(REPO, "profiles", "latest_traffic.json")
        if os.path.exists(tf) and Ns == 1024 and H == 30:     # PMC counters cannot be read in-process: last profiled run
            tj = json.load(open(tf))
            traffic, traffic_src = tj["hbm_bytes_per_launch_gfx950_corrected"], tj["source"]
        flop = FLOP_PER_TRAJ_STEP["pendulum1D_R_H30"] * Ns * H  # per launch (one GPU)
        achieved = flop / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": "sampled trajectory-steps/sec (Ns*H per wall second)",
            "value": units * a.steps / wall,
            "unit": "trajectory-steps/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "prewarm_steps": max(a.prewarm, 0),     # untimed clock-ramp steps before the warmup (see --prewarm)
            "ms_per_step": wall / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: params_pendulum1D_samples, sequential re-conditioned rollout "
                                   "(mode R, value+gradient labels T=3), Ns=%d per GPU, H=%d" % (Ns, H),
                       "Ns_per_gpu": Ns, "H": H, "Ns_total": world * Ns,
                       "parallelism": "samples sharded over %d GPU(s), RCCL all-gather of X_traj per rollout" % world,
                       "device": name, "cus": cus},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_hbm_bytes_per_launch": MIN_HBM_BYTES_PER_TRAJ_STEP["pendulum1D_R_H30"] * Ns * H,
                         "hbm_gbps": (traffic / (kern_ms * 1e-3) / 1e9) if traffic else None,
                         "hbm_frac_of_8TBps": (traffic / (kern_ms * 1e-3) / 8e12) if traffic else None,
                         "mfma_busy": 0.0,      # SQ_VALU_MFMA_BUSY_CYCLES = 0 (profiles/): three right-hand sides cannot fill an FP64 MFMA tile
                         "kernel": "rollout_fast_kernel<3,36,1,pendulum1D,L_hh in LDS,grid root>", "kernel_ms": kern_ms,
                         "timed_region_ms_per_step_hip_events": region_ms,
                         "kernel_ms_per_launch_event_pairs": kern_ms_pairs,
                         "flop_per_launch": flop,
                         "note": "FP64 (vector FMA; FP64 MFMA peak is the same 78.6 TFLOP/s on MI355X); algorithmic "
                                 "FLOP = 2.55e4 per trajectory-step (SURVEY 8d) x Ns x H; min HBM traffic 80 B per "
                                 "trajectory-step, i.e. the kernel is latency/FLOP bound, not HBM bound"},
        }
        if world == 1 and a.cpu_sample > 0:
            v, dt = cpu_baseline(a.cpu_sample, H, u_ff)
            out["cpu_baseline"] = {"value": v, "unit": "trajectory-steps/s", "cores": torch.get_num_threads(),
                                   "kind": "port",
                                   "sample": "same workload, Ns=%d of %d samples, full H=%d horizon, %.1f s of CPU work "
                                             "(oracle: reference-faithful from-scratch batched Cholesky per step, torch "
                                             "CPU FP64; gpytorch itself is not installable on the box)" % (a.cpu_sample, Ns, H, dt),
                                   "host_cpus": os.cpu_count()}
        else:
            out["cpu_baseline"] = None
        out["reachable_set"] = reach
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
