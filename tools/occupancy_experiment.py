"""Experiment: does a second wave per SIMD come for free for the tuned rollout kernel?  (short horizon so LDS allows it)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import _lib
from sampling_gpmpc_amd.rollout import RolloutRunner
from tests.helpers import fs_params, synthetic_u_ff
def run(Ns, H, spw):
    os.environ["GPMPC_MAX_SPW"] = str(spw)
    p = fs_params("params_pendulum1D_samples", Ns, H); p["common"]["use_cuda"] = True; p["agent"]["base_sample_generator"] = "vectorized"
    torch.manual_seed(1)
    agent = sg.Agent(p, sg.make_env(p)); u_ff = synthetic_u_ff(1, H); erv = agent.epistimic_random_vector; per = Ns * 3
    r = RolloutRunner(agent, u_ff, erv.reshape(-1)[per:], erv.shape[1] * per, H, _lib.MODE_RECONDITIONED, False)
    for _ in range(3): r.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): r.launch()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"Ns={Ns} H={H} samples/workgroup={spw}: {ms*1e3:8.1f} us  -> {Ns*H/ms*1e3/1e6:7.1f} M traj-steps/s", flush=True)
for H in (12, 15):
    run(1024, H, 4); run(2048, H, 8); run(2048, H, 4)
