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
