"""CPU: host-side logic of the product package (no GPU, no oracle in the product path) against the goldens, the
C-ABI surface of the built library, and the loud-failure contract."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import sampling_gpmpc_amd as sg
from sampling_gpmpc_amd import _lib
from tests.helpers import GOLDEN, REPO, load_params


def g(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.mark.parametrize("tag,pname", [("pendulum1D", "params_pendulum1D_samples"), ("car_residual", "params_car_residual")])
def test_env_plugins_match_reference(tag, pname):
    d = g(f"env_{tag}.npz")
    p = load_params(pname)
    p["common"]["use_cuda"] = False
    env = sg.make_env(p)
    X, Y = env.initial_training_data()
    xu, dg = torch.tensor(d["xu"]), torch.tensor(d["dg"])
    checks = [(X, d["X_train"]), (Y, d["Y_train"]), (env.get_prior_data(X), d["prior_data"]),
              (env.unknown_dyn(X), d["unknown_dyn"]), (env.known_dyn(xu), d["known_dyn"]),
              (env.get_f_known_jacobian(xu), d["f_jac"]), (env.get_g_xu_hat(xu), d["g_xu_hat"]),
              (env.transform_sensitivity(dg, xu), d["transform"]), (env.B_d, d["B_d"]),
              (env.discrete_dyn(torch.tensor(d["one_xu"])), d["discrete_dyn"])]
    for a, b in checks:
        np.testing.assert_allclose(a.numpy(), b, rtol=1e-13, atol=1e-14, equal_nan=True)
    assert env.pad_g == list(d["pad_g"]) and env.g_idx_inputs == list(d["g_idx"])


def test_tightenings_match_reference():
    d = g("tightenings.npz")
    for tag, pname, H in [("P17", "params_pendulum1D_samples", 17), ("P30", "params_pendulum1D_samples", 30),
                          ("C50", "params_car_residual", 50)]:
        p = load_params(pname)
        p["optimizer"]["H"] = H
        te, ci = sg.get_reachable_set_ball(p, np.ones(H + 1))
        np.testing.assert_allclose(np.stack(te), d[f"{tag}_tilde_eps"], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(ci, d[f"{tag}_ci"], rtol=1e-13)


@pytest.mark.parametrize("tag,pname", [("pendulum1D", "params_pendulum1D_samples"), ("car_residual", "params_car_residual")])
def test_agent_host_plumbing_matches_reference(tag, pname):
    """Seeded base samples (bit-exact stream), real-data batch views, x_hat reshapes, hallucinated-set update."""
    d = g(f"agent_plumbing_{tag}.npz")
    p = load_params(pname)
    p["common"]["use_cuda"] = False
    p["agent"]["num_dyn_samples"] = int(d["Ns"])
    p["agent"]["true_dyn_as_sample"] = False
    p["optimizer"]["H"] = int(d["H"])
    p["common"]["num_MPC_itrs"] = int(d["n_mpc"])
    p["optimizer"]["SEMPC"]["max_sqp_iter"] = int(d["n_itr"])
    torch.manual_seed(123456)
    env = sg.make_env(p)
    agent = sg.Agent(p, env)
    np.testing.assert_array_equal(agent.epistimic_random_vector.numpy(), d["epistimic_random_vector"])
    np.testing.assert_array_equal(agent.Dyn_gp_X_train_batch.numpy(), d["X_train_batch"])
    np.testing.assert_array_equal(agent.Dyn_gp_Y_train_batch.numpy(), d["Y_train_batch"])
    np.testing.assert_array_equal(agent.get_batch_x_hat(d["x_h"], d["u_h"]).numpy(), d["batch_x_hat"])
    bxd = agent.get_batch_x_hat_u_diff(d["x_h"], d["u_diff"])
    np.testing.assert_array_equal(bxd.numpy(), d["batch_x_hat_u_diff"])
    g_xu = env.get_g_xu_hat(bxd)
    y = torch.tensor(d["y_inj"])
    agent.update_hallucinated_Dyn_dataset(g_xu, y)
    np.testing.assert_array_equal(agent.Hallcinated_X_train.numpy(), d["hall_X_0"])
    np.testing.assert_array_equal(agent.Hallcinated_Y_train.numpy(), d["hall_Y_0"])
    p["agent"]["Dyn_gp_min_data_dist"] = float(d["min_dist_1"])
    agent.update_hallucinated_Dyn_dataset(g_xu + 0.2, y * 2)
    np.testing.assert_array_equal(agent.Hallcinated_X_train.numpy(), d["hall_X_1"])
    np.testing.assert_array_equal(agent.Hallcinated_Y_train.numpy(), d["hall_Y_1"])
    assert len(agent.tilde_eps_list) == int(d["H"]) + 1 and len(agent.ci_list) == int(d["H"])
    assert agent.get_next_to_go_loc().tolist() == [2]


def test_vectorized_base_samples_respect_the_truncation():
    p = load_params("params_pendulum1D_samples")
    p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = 500, 30
    p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = 2, 1
    torch.manual_seed(0)
    z = sg.random_vector_within_bounds(p, 1, 3, mode="vectorized")
    assert tuple(z.shape) == (2, 1, 500, 1, 30, 3)
    assert float(z.abs().max()) <= p["agent"]["Dyn_gp_beta"]
    assert abs(float(z.mean())) < 0.02 and 0.85 < float(z.std()) < 1.0


def test_library_exports_every_declared_symbol():
    """Every function include/gpmpc_hip.h declares is exported by the built .so and bound by the ctypes layer."""
    hdr = open(os.path.join(REPO, "include", "gpmpc_hip.h")).read()
    declared = set(re.findall(r"\b(gpmpc_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} not exported"
    assert lib.gpmpc_abi_version() == _lib.ABI_VERSION
    assert lib.gpmpc_plin_len(4, 2, 10) == 10 * (16 + 8 + 8) + 2 + 2 + 7


def test_argument_validation_without_a_gpu():
    """Host-side argument checks run before any device work and report through gpmpc_last_error_string."""
    lib = _lib.load()
    bad = _lib.make_gp_desc(1, 2, 2, 36, False, [[1.0, 1.0]], [1.0], [1e-6, 1e-6], 1e-6)   # T must be 1 or 1+D
    assert lib.gpmpc_plan_bytes(bad) == 0
    assert b"T must be" in lib.gpmpc_last_error_string()
    ok = _lib.make_gp_desc(1, 2, 3, 36, False, [[1.84, 1.92]], [0.03], [4.8e-6, 2.27e-6, 4.8e-6], 1e-6)
    assert lib.gpmpc_plan_bytes(ok) >= (2 * 36 * 36 + 72) * 8
    assert lib.gpmpc_rollout_workspace_bytes(ok, _lib.MODE_RECONDITIONED, 3, 1024, 30) >= 1024 * (36 * 87 + 87 * 44) * 8
    assert lib.gpmpc_joint_workspace_bytes(ok, 16, 0, 30) > 0
    env = _lib.make_env_desc(_lib.ENV_CAR_RESIDUAL, 4, 2, True, 0.06, 1.1, 1.7, [[0] * 4] * 2, [0] * 4)
    rc = lib.gpmpc_rollout(ok, env, 1, 1, _lib.MODE_RECONDITIONED, 3, -1.0, 2.5, 4, 5, 1, 0, 1, 1, 0, 1, None, None, 1,
                           None, 0, None)
    assert rc == -1 and b"car_residual needs" in lib.gpmpc_last_error_string()
    # the entry points of rounds 6 (ABI 8 / 9) check their arguments on the host as well
    assert lib.gpmpc_joint_sample_pending(ok, None, None, 4, 0, None, None, None, 0, 5, None, None, -1.0, 2.5, 1, None, None, None, None,
                                          None, 0, None, None, 0, None, None, 0, 0, _lib.PENDING_USE | _lib.PENDING_WRITE) == -1
    assert b"NULL pointer" in lib.gpmpc_last_error_string() and lib.gpmpc_joint_pending_written() == 0
    assert lib.gpmpc_build_x_hat(2, 1, 4, 5, None, None, 0, None, None) == -1 and b"gpmpc_build_x_hat" in lib.gpmpc_last_error_string()
    assert lib.gpmpc_assemble_jacobians_plin(ok, env, 4, 5, None, None, None, None, None, None, None, None, None, None, None, None) != 0


def test_product_path_fails_loudly_without_hip_device():
    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible")
    p = load_params("params_pendulum1D_samples")
    p["agent"]["num_dyn_samples"], p["optimizer"]["H"] = 4, 3
    p["common"]["num_MPC_itrs"] = 1
    agent = sg.Agent(p, sg.make_env(p))
    with pytest.raises(_lib.GpmpcError):
        agent.train_hallucinated_dynGP(0)
    from sampling_gpmpc_amd.rollout import forward_sampling_rollout
    p["optimizer"]["H"], p["common"]["num_MPC_itrs"], p["optimizer"]["SEMPC"]["max_sqp_iter"] = 1, 3, 2
    agent = sg.Agent(p, sg.make_env(p))
    with pytest.raises(_lib.GpmpcError):
        forward_sampling_rollout(agent, np.zeros((3, 1)))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, "sampling_gpmpc_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, os.path.join(root, f)


def test_tensor_grid_detection():
    from sampling_gpmpc_amd.gp_model import _detect_tensor_grid
    for pname, want in [("params_pendulum1D_samples", (4, 9)), ("params_car_residual", (5, 9))]:
        p = load_params(pname)
        p["common"]["use_cuda"] = False
        X, _ = sg.make_env(p).initial_training_data()
        assert _detect_tensor_grid(X) == want
        Xb = X.clone()
        Xb[7, 1] += 1e-9                         # not a tensor grid any more
        assert _detect_tensor_grid(Xb) == (0, 0)
    assert _detect_tensor_grid(torch.rand(10, 2, dtype=torch.float64)) in [(0, 0), (10, 1)]


def test_gp_desc_layout_matches_header():
    """ctypes mirror of gpmpc_gp_desc_t / gpmpc_env_desc_t: sizes implied by the header's field list."""
    assert ctypes.sizeof(_lib.GpDesc) == 8 * 4 + 8 * (4 * 4 + 4 + 5 + 2)
    assert ctypes.sizeof(_lib.EnvDesc) == 4 * 4 + 8 * (3 + 4 * 8 + 8)
    d = _lib.make_gp_desc(3, 2, 1, 45, False, [[2.0, 1.1]] * 3, [0.05] * 3, [2e-7], 1e-20, grid=(5, 9))
    assert (d.grid_n0, d.grid_n1, d.N_r, d.T) == (5, 9, 45, 1)
    lib = _lib.load()
    bad = _lib.make_gp_desc(3, 2, 1, 45, False, [[2.0, 1.1]] * 3, [0.05] * 3, [2e-7], 1e-20, grid=(5, 8))
    assert lib.gpmpc_plan_bytes(bad) == 0 and b"grid_n0" in lib.gpmpc_last_error_string()


def test_on_disk_formats_roundtrip(tmp_path):
    """data_X_traj_<idx>.pkl as the reference writes it: float64 ndarray (Ns, nx, H+1), readable with plain pickle."""
    import pickle
    from sampling_gpmpc_amd import io_formats as io
    rng = np.random.default_rng(0)
    a, b = rng.normal(size=(5, 4, 9)), rng.normal(size=(3, 4, 9))
    pa = io.save_x_traj(str(tmp_path), 400, torch.tensor(a))
    io.save_x_traj(str(tmp_path), 401, b)
    assert os.path.basename(pa) == "data_X_traj_400.pkl"
    with open(pa, "rb") as f:
        got = pickle.load(f)                      # what generate_convex_hull.py does
    assert isinstance(got, np.ndarray) and got.dtype == np.float64
    np.testing.assert_array_equal(got, a)
    np.testing.assert_array_equal(io.merge_x_traj(str(tmp_path), [400, 401]), np.concatenate([a, b], axis=0))
    erv = torch.randn(2, 2, 3, 1, 1, 3, dtype=torch.float64)
    pe = io.save_epistemic_vector(str(tmp_path), 7, erv)
    assert torch.equal(io.load_epistemic_vector(pe), erv)
    with pytest.raises(ValueError):
        io.save_x_traj(str(tmp_path), 1, np.zeros((3, 4)))


def test_inline_asm_dpp_table_reads_have_no_valu_write_hazard():
    """rollout_indep_grid_kernel reads its register-resident tables as the DPP source of inline-asm v_fmac_f64_dpp; the
    compiler cannot insert the two wait states a VALU-written DPP source needs into inline asm, so the ISA is scanned
    for a write of a table register right in front of such a read (tools/check_dpp_hazard.py)."""
    import subprocess, sys, shutil
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "check_dpp_hazard.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_on_disk_formats_are_read_by_the_reference_loaders(tmp_path):
    """data.pkl, X_traj_list_<k>.pkl and data_X_traj_<idx>.pkl written by sampling_gpmpc_amd.io_formats: the fixture holds
    what the REFERENCE's own loader lines produced from them (tests/golden/make_goldens.py --only formats executes
    simulate_forward_sampling_car.py:91-98, extra/cdc_plt.py:169-176, generate_convex_hull.py:76-83 from the reference
    sources); the files are rewritten here from the same inputs and read with the package's loaders."""
    import torch
    from sampling_gpmpc_amd import io_formats as io
    d = np.load(os.path.join(GOLDEN, "formats_check.npz"))
    X_traj, U = d["X_traj"], d["U"]
    data = {k: [] for k in io.DATA_PKL_KEYS}
    data["input_traj"] = [d["input_traj_0"], d["input_traj_1"]]
    io.save_data_pkl(str(tmp_path), data)
    back = io.load_data_pkl(os.path.join(str(tmp_path), "data.pkl"))
    assert set(back) == set(io.DATA_PKL_KEYS)
    np.testing.assert_array_equal(np.asarray(back["input_traj"][-1]), d["ref_input_traj_last"])
    pth = io.save_x_traj_list(str(tmp_path), 3, X_traj, U, g_ny=1)
    lst = io.load_x_traj_list(pth)
    arr = np.array([np.asarray(t) for t in lst])
    assert list(arr.shape) == d["ref_list_shape"].tolist() and lst[0].dtype == torch.float64
    np.testing.assert_array_equal(arr[:, :, 0, 0, 0:2], d["ref_state_traj"])
    np.testing.assert_array_equal(arr[:-1, 0, 0, 0, 2:], U)                       # applied input rides in the last columns
    for idx in (1, 2):
        io.save_x_traj(str(tmp_path), idx, X_traj + idx)
    np.testing.assert_array_equal(io.merge_x_traj(str(tmp_path), (1, 2)), d["ref_merged"])


def test_joint_factor_cache_bookkeeping():
    """Host logic of gp_model.JointFactorCache (no GPU): which rows it vouches for.  Append-only growth of the slot list
    keeps the old rows, a changed point / a shorter or different slot list / a failed factorisation drops them, the
    buffer grows with the set."""
    from types import SimpleNamespace
    from sampling_gpmpc_amd import _lib
    from sampling_gpmpc_amd.gp_model import JointFactorCache
    _lib.load()
    Ns, g_ny, T, D, n_r = 4, 3, 3, 2, 45
    desc = _lib.make_gp_desc(g_ny, D, T, n_r, False, [[2.0, 1.1]] * 3, [0.05] * 3, [2e-7] * 3, 1e-20)
    g = torch.Generator().manual_seed(0)
    X = torch.randn(Ns, g_ny, 40, D, generator=g, dtype=torch.float64)

    def model(n_pts):
        return SimpleNamespace(hyper=SimpleNamespace(g_ny=g_ny, T=T, D=D), n_h=n_pts, hall_X=X[:, :, :n_pts].clone(),
                               plan=SimpleNamespace(n_r=n_r, desc=desc, X_r=torch.zeros(1), version=1),
                               h_slots=torch.arange(n_pts * T, dtype=torch.int32))

    plan = model(0).plan

    def mk(n_pts):
        m = model(n_pts)
        m.plan = plan                                   # the cache is keyed by the plan's version
        return m

    c = JointFactorCache()
    assert c.prepare(mk(2), Ns, 6)[0] is None           # fewer than 16 slots: not worth a cache
    buf, rows, n_c = c.prepare(mk(10), Ns, 30)
    assert buf is not None and rows >= 4 * 30 and n_c == 0 and buf.numel() * 8 == _lib.load().gpmpc_joint_cache_bytes(desc, Ns, rows)
    c.commit(mk(10), 30, ok=True)
    assert c.n_valid == 30
    assert c.prepare(mk(20), Ns, 60)[2] == 30           # grown: the old slots' rows are kept
    assert c.prepare(mk(10), Ns, 30)[2] == 30           # the same set again: everything cached
    m = mk(20)
    m.hall_X[1, 2, 3, 0] += 1e-9
    assert c.prepare(m, Ns, 60)[2] == 0                 # a cached point moved
    m = mk(20)
    m.h_slots = torch.cat([m.h_slots[:5], m.h_slots[6:]])
    assert c.prepare(m, Ns, 59)[2] == 0                 # a slot disappeared from the prefix (NaN-masked label)
    assert c.prepare(mk(5), Ns, 15)[2] == 0             # reset: fewer slots than cached
    c.rewind(12)
    assert c.n_valid == 12 and c.prepare(mk(20), Ns, 60)[2] == 12
    # the Agent's bound on the conditioning set (max_sqp_iter * H points) sizes the buffer once: no regrowth up to it
    ch = JointFactorCache()
    mh = mk(10)
    mh._ws_cache = {"joint_points_hint": 160}
    bufh, rows_h, _ = ch.prepare(mh, Ns, 30)
    assert rows_h == 512 and ch.prepare(mk(20), Ns, 60)[0] is bufh and ch.prepare(mk(40), Ns, 120)[0] is bufh
    # ... but only a REACHABLE bound (within gpmpc_joint_sample's 2048-row limit) is honoured: the shipped car's
    # max_sqp_iter * H = 150 * 50 points is not, and the size then follows the 4x rule
    cu = JointFactorCache()
    mu = mk(10)
    mu._ws_cache = {"joint_points_hint": 150 * 50}
    assert cu.prepare(mu, Ns, 30)[1] == 256
    # the entry points take an EVEN row count only (16-byte row units of the matrix-pipe path)
    assert _lib.load().gpmpc_joint_cache_bytes(desc, Ns, 257) == 0 and _lib.load().gpmpc_joint_cache_bytes(desc, Ns, 256) > 0
    c.commit(mk(20), 60, ok=False)
    assert c.n_valid == 0 and c.prepare(mk(20), Ns, 60)[2] == 0      # a failed factorisation is not kept
    c.commit(mk(20), 60, ok=True)
    old_rows = c.rows
    big = SimpleNamespace(**vars(mk(20)))
    big.h_slots = torch.arange(old_rows + 16, dtype=torch.int32)
    big.n_h = (old_rows + 16 + T - 1) // T
    big.hall_X = torch.zeros(Ns, g_ny, big.n_h, D, dtype=torch.float64)
    buf2, rows2, n_c2 = c.prepare(big, Ns, old_rows + 16)
    assert rows2 > old_rows and n_c2 == 0               # a set beyond the capacity: new buffer, nothing carried over
    # a rebuilt plan (new version) never inherits the rows, even if CPython reuses the old object's id
    c.commit(big, old_rows + 16, ok=True)
    assert c.prepare(big, Ns, old_rows + 16)[2] == old_rows + 16
    big.plan = SimpleNamespace(**vars(plan))
    big.plan.version = 2
    assert c.prepare(big, Ns, old_rows + 16)[2] == 0
    # commit copies only the appended points and skips calls that computed nothing new
    c2 = JointFactorCache()
    c2.prepare(mk(10), Ns, 30)
    c2.commit(mk(10), 30, ok=True)
    snap = c2.X.clone()
    assert c2.prepare(mk(10), Ns, 30)[2] == 30
    c2.commit(mk(10), 30, ok=True, n_cached=30)          # mean-only repeat: nothing to copy
    assert torch.equal(c2.X, snap)
    assert c2.prepare(mk(20), Ns, 60)[2] == 30
    c2.commit(mk(20), 60, ok=True, n_cached=30)
    assert c2.n_valid == 60 and torch.equal(c2.X, X[:, :, :20])
    # validity by the Agent's lineage counter (no tensor comparison): same generation + append-only growth; a rewind (the
    # benchmarks' "forget this draw") keeps whole points and stays valid - bench.py's closed-loop legs depend on it
    c3 = JointFactorCache()
    def lm(n_pts, gen):
        m = mk(n_pts)
        m.lineage = (gen, n_pts)
        return m
    c3.prepare(lm(10, 7), Ns, 30)
    c3.commit(lm(10, 7), 30, ok=True)
    assert c3.prepare(lm(20, 7), Ns, 60)[2] == 30
    c3.commit(lm(20, 7), 60, ok=True, n_cached=30)
    assert c3.prepare(lm(30, 7), Ns, 90)[2] == 60
    c3.rewind(30)
    assert c3.n_valid == 30 and c3.prepare(lm(20, 7), Ns, 60)[2] == 30
    c3.rewind(31)                                        # not a whole point: rounded down
    assert c3.n_valid == 30
    assert c3.prepare(lm(20, 8), Ns, 60)[2] == 0        # another generation of the hallucinated set
    c.enabled = False
    assert c.prepare(mk(20), Ns, 60) == (None, 0, 0)


def test_host_thread_budget_follows_the_cgroup_quota(tmp_path):
    """Thread pools are lowered only when they exceed the CPUs the cgroup grants (profiles/r3_closed_loop_trace.md)."""
    from sampling_gpmpc_amd import _host_threads as ht
    assert ht.parse_cpu_max("1600000 100000\n") == 16.0 and ht.parse_cpu_max("max 100000") is None
    assert ht.parse_cpu_max("garbage") is None
    (tmp_path / "cpu.max").write_text("250000 100000\n")
    assert ht.cgroup_cpu_quota(str(tmp_path)) == 2.5
    assert ht.cgroup_cpu_quota(str(tmp_path / "missing")) is None
    assert ht.host_thread_budget(visible=256, quota=16.0, current=128) == 8      # the GPU boxes of this project
    assert ht.host_thread_budget(visible=256, quota=2.5, current=256) == 1
    assert ht.host_thread_budget(visible=8, quota=None, current=8) is None       # nothing oversubscribed: hands off
    assert ht.host_thread_budget(visible=64, quota=16.0, current=4) is None


def test_host_thread_env_below_one_is_ignored(monkeypatch):
    """GPMPC_HOST_THREADS=0 used to reach torch.set_num_threads(0), which raises - at package import."""
    from sampling_gpmpc_amd import _host_threads as ht
    monkeypatch.setenv("GPMPC_HOST_THREADS", "0")
    monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
    before = torch.get_num_threads()
    got = ht.limit_host_threads()
    assert got is None or 1 <= got <= before
    assert torch.get_num_threads() >= 1


def test_joint_workspace_covers_the_split_launches():
    """gpmpc_joint_workspace_bytes (host logic, no GPU): between 417 and 544 conditioning slots the test rows of the matrix-pipe path
    run as TOP + BOTTOM launches, whose X tiles (416 KB per chain, batches of <= 3072 chains) live in the workspace.  With 45 real
    slots the form ends behind 499 hallucinated ones: the size drops by the X tiles there (at its lower end the temporary factor
    cache of the one-launch form leaves as the X tiles come: no step to see)."""
    from sampling_gpmpc_amd import _lib
    lib = _lib.load()
    desc = _lib.make_gp_desc(3, 2, 3, 45, False, [[2.0, 1.1]] * 3, [0.05] * 3, [2e-7] * 3, 1e-20)
    m = 40
    for Ns, chains in ((1024, 3072), (8, 24), (4096, 3072)):
        xt = chains * 26 * 8 * 2048                   # chains of a batch x tiles x waves x 2 KB
        b = {n: lib.gpmpc_joint_workspace_bytes(desc, Ns, n, m) for n in (498, 499, 500, 501)}
        grow = b[499] - b[498]
        assert grow > 0 and b[501] - b[500] > 0
        assert abs((b[499] + grow - b[500]) - xt) < 0.1 * xt + (64 << 20), (Ns, b)      # (the per-chain slots grow in steps)
