"""Common scaffolding of the environment plug-ins (device choice, training grid)."""
from __future__ import annotations

import torch

F64 = torch.float64


def pick_device(params, device=None) -> torch.device:
    """Same rule as the reference (agent.py:43-50): use the accelerator when common.use_cuda and one is visible."""
    if device is not None:
        return torch.device(device)
    if params["common"]["use_cuda"] and torch.cuda.is_available():
        return torch.device("cuda")
    return torch.device("cpu")


class ResidualEnv:
    """Plug-in interface consumed by Agent: training grid, known part f (+Jacobian), residual g, B_d, selectors."""

    #: filled by subclasses
    pad_g: list
    g_idx_inputs: list
    env_id: int

    def __init__(self, params, device=None):
        self.params = params
        ag = params["agent"]
        self.nx, self.nu = ag["dim"]["nx"], ag["dim"]["nu"]
        self.g_ny, self.g_nx, self.g_nu = ag["g_dim"]["ny"], ag["g_dim"]["nx"], ag["g_dim"]["nu"]
        self.torch_device = pick_device(params, device)
        self.use_cuda = self.torch_device.type == "cuda"
        self.dt = float(params["optimizer"]["dt"])

    # -- training data ------------------------------------------------------------------------------------
    def _grid_axes(self):
        raise NotImplementedError

    def initial_training_data(self):
        """(X (N_r, D), Y (g_ny, N_r, 1+D)); gradient labels are NaN unless env.train_data_has_derivatives."""
        if not self.params["env"]["prior_dyn_meas"]:
            raise NotImplementedError("only prior_dyn_meas: True is supported (all runnable reference configs)")
        a0, a1 = self._grid_axes()
        G0, G1 = torch.meshgrid(a0, a1, indexing="ij")           # first axis major, like the reference
        X = torch.stack([G0.reshape(-1), G1.reshape(-1)], dim=1)
        Y = self.get_prior_data(X)
        if not self.params["env"]["train_data_has_derivatives"]:
            Y[:, :, 1:] = float("nan")
        return X, Y

    def get_g_xu_hat(self, xu_hat):
        """GP-input columns of the first g_ny replicated rows: (Ns, nx, H, nx+nu) -> (Ns, g_ny, H, D)."""
        return xu_hat[:, 0:self.g_ny, :, self.g_idx_inputs]

    def discrete_dyn(self, xu):
        """True plant step for ONE state-input row (1, nx+nu) -> (nx, 1); host-side (used by the MPC loop)."""
        xu = torch.as_tensor(xu, dtype=F64).cpu()
        assert xu.shape[1] == self.nx + self.nu
        f = self.known_dyn(xu.reshape(1, 1, 1, -1).expand(1, self.nx, 1, -1))[0, :, :]
        g = self.unknown_dyn(xu[:, self.g_idx_inputs]).transpose(0, 1)
        return f + self._B_d_of(xu) @ g
