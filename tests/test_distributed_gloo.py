"""CPU, world_size 2 (gloo): the sample-sharding + all-gather logic of sampling_gpmpc_amd.distributed.  The local
rollout is played by the oracle here (tests may use it); on the GPU box the same code path runs gpmpc_rollout + RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import fs_params, synthetic_u_ff


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, Ns, H, erv, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import agent_oracle as ao
    from sampling_gpmpc_amd.distributed import shard_range, sharded_rollout
    u_ff = synthetic_u_ff(1, H)

    def local(lo, hi):
        p = fs_params("params_pendulum1D_samples", hi - lo, H)
        agent = ao.OracleAgent(p, ao.make_oracle_env(p), erv[:, :, lo:hi])      # base samples by GLOBAL sample id
        return torch.from_numpy(ao.forward_sampling_rollout(agent, u_ff))

    tube = sharded_rollout(local, Ns)
    lo, hi = shard_range(Ns, rank, world)
    out_q.put((rank, lo, hi, tube.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("Ns", [8, 7])       # even and ragged shards
def test_sharded_rollout_equals_single_process(Ns):
    H, world = 5, 2
    import sampling_gpmpc_amd as sg
    from oracle import agent_oracle as ao
    p = fs_params("params_pendulum1D_samples", Ns, H)
    torch.manual_seed(3)
    erv = sg.random_vector_within_bounds(p, 1, 3, mode="reference")
    agent = ao.OracleAgent(p, ao.make_oracle_env(p), erv)
    X_ref = ao.forward_sampling_rollout(agent, synthetic_u_ff(1, H))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, Ns, H, erv, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    ranges = sorted((lo, hi) for _, lo, hi, _ in got)
    assert ranges[0][0] == 0 and ranges[-1][1] == Ns and ranges[0][1] == ranges[1][0]
    for _, _, _, tube in got:                       # every rank holds the FULL tube, identical to the 1-process run
        assert tube.shape == X_ref.shape
        np.testing.assert_allclose(tube, X_ref, rtol=1e-12, atol=1e-13)


def test_shard_range_partitions():
    from sampling_gpmpc_amd.distributed import shard_range
    for Ns in [1, 7, 8, 1024, 262144]:
        for world in [1, 2, 3, 8]:
            r = [shard_range(Ns, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == Ns
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


# ---------------------------------------------------------------------------------------------------------------------
# closed loop: the reference's cross-sample couplings under sample sharding
# ---------------------------------------------------------------------------------------------------------------------
def _coupling_worker(rank, world, port, Ns, payload, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy
    import sampling_gpmpc_amd as sg
    from sampling_gpmpc_amd.distributed import (shard_range, filtered_in_all_samples, replace_rejected_samples,
                                                gather_jacobians, make_sharded_agent)
    lo, hi = shard_range(Ns, rank, world)
    filt, X, Y, left, jac, newX, newY, params, seed = payload
    res = {"rank": rank}
    res["all_s"] = filtered_in_all_samples(filt[lo:hi]).numpy()
    Xr, Yr = replace_rejected_samples(X[lo:hi].clone(), Y[lo:hi].clone(), left[lo:hi], Ns, np.random.RandomState(5))
    res["X"], res["Y"] = Xr.numpy(), Yr.numpy()
    g = gather_jacobians([a[lo:hi] for a in jac], Ns)
    res["jac"] = g
    # the Agent wiring: min-distance filter of update_hallucinated_Dyn_dataset with samples sharded over ranks
    torch.manual_seed(seed)
    agent = make_sharded_agent(sg.Agent, copy.deepcopy(params), sg.make_env(params))
    assert agent.ns == hi - lo and agent.ns_global == Ns and agent.shard == (lo, hi)
    agent.update_hallucinated_Dyn_dataset(newX[lo:hi], newY[lo:hi])
    agent.update_hallucinated_Dyn_dataset(newX[lo:hi] + 0.01, newY[lo:hi])          # close to the first batch: filtered
    res["hX"], res["hY"] = agent.Hallcinated_X_train.numpy(), agent.Hallcinated_Y_train.numpy()
    res["erv"] = agent.epistimic_random_vector.numpy()
    out_q.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("Ns", [6, 7])
def test_cross_sample_couplings_sharded_equal_single_process(Ns):
    import copy
    import sampling_gpmpc_amd as sg
    from tests.helpers import load_params
    world, g_ny, m, n, D, T = 2, 1, 5, 4, 2, 3
    gen = torch.Generator().manual_seed(100 + Ns)
    filt = torch.rand(Ns, g_ny, m, generator=gen) < 0.6
    filt[:, :, 1] = True                                          # one point filtered in every sample
    X = torch.randn(Ns, g_ny, n, D, generator=gen, dtype=torch.float64)
    Y = torch.randn(Ns, g_ny, n, T, generator=gen, dtype=torch.float64)
    left = (torch.rand(Ns, generator=gen) < 0.5).to(torch.int64)
    left[0], left[-1] = 1, 0                                      # at least one survivor, one rejected sample
    # the three arrays of dyn_fg_jacobians (gp_val, y_grad, u_grad): packed, ONE gather collective, split on rank 0
    jac = [np.arange(Ns * 2 * 3, dtype=np.float64).reshape(Ns, 2, 3, 1), np.random.RandomState(1).randn(Ns, 2, 3, 2),
           np.random.RandomState(2).randn(Ns, 2, 3, 1)]
    params = copy.deepcopy(load_params("params_pendulum1D_samples"))
    params["common"]["use_cuda"] = False
    params["agent"]["num_dyn_samples"] = Ns
    params["optimizer"]["H"] = m
    params["common"]["num_MPC_itrs"], params["optimizer"]["SEMPC"]["max_sqp_iter"] = 2, 2
    params["agent"]["Dyn_gp_min_data_dist"] = 0.05
    newX = torch.randn(Ns, g_ny, m, D, generator=gen, dtype=torch.float64) + torch.tensor([2.5, 0.0])
    newX[:, :, 2, :] = torch.tensor([2.1, -5.0], dtype=torch.float64)      # ON a real grid point: filtered in every sample
    newY = torch.randn(Ns, g_ny, m, T, generator=gen, dtype=torch.float64)
    seed = 4242
    # single-process references
    all_ref = torch.all(filt, dim=0).numpy()
    rng = np.random.RandomState(5)
    Xs, Ys = X.clone(), Y.clone()
    dead = left == 0
    remaining = torch.arange(Ns)[left > 0].numpy()
    n_rep = int(dead.sum())
    Xs[dead] = Xs[rng.choice(remaining, n_rep).tolist()]
    Ys[dead] = Ys[rng.choice(remaining, n_rep).tolist()]
    torch.manual_seed(seed)
    ref_agent = sg.Agent(copy.deepcopy(params), sg.make_env(params))
    ref_agent.update_hallucinated_Dyn_dataset(newX, newY)
    ref_agent.update_hallucinated_Dyn_dataset(newX + 0.01, newY)
    assert ref_agent.Hallcinated_X_train.shape[2] < 2 * m, "the all-samples drop did not trigger in the reference run"

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    payload = (filt, X, Y, left, jac, newX, newY, params, seed)
    procs = [ctx.Process(target=_coupling_worker, args=(r, world, port, Ns, payload, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = sorted((q.get(timeout=180) for _ in range(world)), key=lambda r: r["rank"])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for r in got:
        np.testing.assert_array_equal(r["all_s"], all_ref)
    np.testing.assert_array_equal(np.concatenate([r["X"] for r in got]), Xs.numpy())
    np.testing.assert_array_equal(np.concatenate([r["Y"] for r in got]), Ys.numpy())
    assert got[1]["jac"] is None
    for a, b in zip(got[0]["jac"], jac):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(np.concatenate([r["hX"] for r in got]), ref_agent.Hallcinated_X_train.numpy())
    np.testing.assert_array_equal(np.concatenate([r["hY"] for r in got]), ref_agent.Hallcinated_Y_train.numpy())
    np.testing.assert_array_equal(np.concatenate([r["erv"] for r in got], axis=2), ref_agent.epistimic_random_vector.numpy())


# ---------------------------------------------------------------------------------------------------------------------
# per-shard base samples (counter-based stream keyed by GLOBAL sample id) and the rank-spanning NaN mask
# ---------------------------------------------------------------------------------------------------------------------
def _counter_worker(rank, world, port, Ns, H, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sampling_gpmpc_amd as sg
    from oracle import agent_oracle as ao
    from sampling_gpmpc_amd.distributed import all_gather_tube, make_sharded_agent, shard_range
    p = fs_params("params_pendulum1D_samples", Ns, H)
    p["agent"]["base_sample_generator"] = "counter"
    p["agent"]["base_sample_seed"] = 77
    agent = make_sharded_agent(sg.Agent, p, sg.make_env(p))
    lo, hi = shard_range(Ns, rank, world)
    assert agent.shard == (lo, hi) and agent.epistimic_random_vector.shape[2] == hi - lo   # HBM for z: 1/G per rank
    # the rank's rollout on ITS shard (the oracle plays the kernel on CPU), then the collective
    pl = fs_params("params_pendulum1D_samples", hi - lo, H)
    oagent = ao.OracleAgent(pl, ao.make_oracle_env(pl), agent.epistimic_random_vector)
    X_local = torch.from_numpy(ao.forward_sampling_rollout(oagent, synthetic_u_ff(1, H)))
    tube = all_gather_tube(X_local, Ns)
    # gpytorch's NaN mask spans the WHOLE batch: a slot that is NaN only in a sample of rank 0 is dropped on every rank
    from sampling_gpmpc_amd.gp_model import _observed_slots
    hy = torch.zeros(hi - lo, 1, 4, 3, dtype=torch.float64)
    if lo == 0:
        hy[0, 0, 2, 1] = float("nan")
    slots = _observed_slots(hy, agent.dist_group)
    out_q.put((rank, tube.numpy(), agent.epistimic_random_vector.numpy(), slots.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_counter_base_samples_make_the_run_independent_of_the_world_size(world):
    """Every rank draws only its own shard of the base samples, yet the assembled tube is bit-identical for 1, 2 and 3
    ranks (even and ragged shards) - and identical to the single-process run on the same stream."""
    Ns, H = 7, 4
    import sampling_gpmpc_amd as sg
    from oracle import agent_oracle as ao
    p = fs_params("params_pendulum1D_samples", Ns, H)
    p["agent"]["base_sample_generator"] = "counter"
    p["agent"]["base_sample_seed"] = 77
    erv = sg.random_vector_within_bounds(p, 1, 3)
    assert float(erv.abs().max()) <= p["agent"]["Dyn_gp_beta"]
    X_ref = ao.forward_sampling_rollout(ao.OracleAgent(p, ao.make_oracle_env(p), erv), synthetic_u_ff(1, H))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_counter_worker, args=(r, world, port, Ns, H, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = sorted((q.get(timeout=180) for _ in range(world)), key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    np.testing.assert_array_equal(np.concatenate([g[2] for g in got], axis=2), erv.numpy())
    for _, tube, _, slots in got:
        np.testing.assert_array_equal(tube, X_ref)
        assert slots.tolist() == [s for s in range(12) if s != 2 * 3 + 1]
