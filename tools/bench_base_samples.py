"""Agent construction at the true-reachable-set size (BASELINE configs[3], Ns = 262144): the counter base samples by
gpmpc_base_samples against the torch-op form of the same stream.   python tools/bench_base_samples.py [Ns]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sampling_gpmpc_amd.agent import counter_base_samples
Ns = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
for g_ny, H, T, beta, tag in ((3, 40, 1, 30.0, "car_residual_fs (mode I, T=1)"), (3, 40, 3, 3.0, "car_residual (T=3, beta 3)")):
    for force in (True, False):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        z = counter_base_samples(1, 1, Ns, g_ny, H, T, beta, seed=1, device="cuda", _force_torch=force)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if not force:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            z = counter_base_samples(1, 1, Ns, g_ny, H, T, beta, seed=1, device="cuda")
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{tag:32s} Ns={Ns}: {'torch ops' if force else 'gpmpc_base_samples'} {1e3 * dt:9.2f} ms  ({z.numel() * 8 / 1e6:.0f} MB)", flush=True)
