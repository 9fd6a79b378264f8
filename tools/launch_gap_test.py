import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import _lib
from sampling_gpmpc_amd.rollout import RolloutRunner
from tests.helpers import fs_params, synthetic_u_ff
Ns, H = 1024, 30
p = fs_params("params_pendulum1D_samples", Ns, H); p["common"]["use_cuda"] = True; p["agent"]["base_sample_generator"] = "vectorized"
torch.manual_seed(1)
agent = sg.Agent(p, sg.make_env(p)); u_ff = synthetic_u_ff(1, H); erv = agent.epistimic_random_vector; per = Ns * 3
r = RolloutRunner(agent, u_ff, erv.reshape(-1)[per:], erv.shape[1] * per, H, _lib.MODE_RECONDITIONED, False)
for _ in range(5): r.launch()
torch.cuda.synchronize()
K = 200
for mode in ("noevents", "events", "noevents", "events"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "events":
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        t0 = time.perf_counter()
        for k in range(K):
            ev[k][0].record(); r.launch(); ev[k][1].record()
    else:
        for k in range(K): r.launch()
    torch.cuda.synchronize(); w = time.perf_counter() - t0
    print(mode, "wall per step us", w / K * 1e6)
# graph
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    r.launch(); torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g, stream=s):
            for k in range(50): r.launch()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): g.replay()
        torch.cuda.synchronize(); w = time.perf_counter() - t0
        print("graph of 50: wall per step us", w / 200 * 1e6)
    except Exception as e:
        print("graph capture failed:", repr(e)[:200])
