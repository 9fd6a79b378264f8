import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import numpy as np, torch, torch.distributed as dist
import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import workloads as wl
from sampling_gpmpc_amd.distributed import make_sharded_agent
dist.init_process_group("nccl", rank=0, world_size=1)
Ns, H = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 40
p = wl.closed_loop_params("params_car_residual", Ns, H, 2, 4)
p["common"]["use_cuda"] = True
p["agent"]["base_sample_generator"] = "counter"
lib = sg._lib.load()
for mode in (0, 1):
    lib.gpmpc_debug_joint_real_kernel(mode)
    for sharded in (False, True):
        agent = make_sharded_agent(sg.Agent, p, sg.make_env(p)) if sharded else sg.Agent(p, sg.make_env(p))
        x0 = np.asarray(p["env"]["start"], dtype=np.float64)[:agent.nx]
        u_h, x_h = wl.synthetic_u_ff(agent.nu, H), np.tile(x0, (H, agent.ns))
        agent.mpc_iteration(0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for k in range(3):
                try:
                    agent.train_hallucinated_dynGP(k)
                    jac = agent.dyn_fg_jacobians_device(agent.get_batch_x_hat(x_h, u_h), k)
                    post = agent.model_i_call
                    print(f"real={mode} sharded={sharded} k={k}: bits {post.last_bits:#x} path {lib.gpmpc_joint_last_path()} n_h={agent.model_i.n_h} nslots={agent.model_i.h_slots.numel()} "
                          f"mean {float(post.mean.abs().sum()):.12e}", flush=True)
                except Exception as e:
                    print(f"real={mode} sharded={sharded} k={k}: {type(e).__name__} {e}", flush=True)
                    break
        del agent
dist.destroy_process_group()
