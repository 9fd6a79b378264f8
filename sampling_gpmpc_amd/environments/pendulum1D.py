"""Pendulum with a 1-output residual GP (reference src/environments/pendulum1D.py, class Pendulum).

State x = (theta, omega), input u.  Known part f = (theta + omega dt, omega); residual g(theta, u) =
-g sin(theta) dt / l + u dt enters through B_d = [0, 1]^T.  GP input = (theta, u) = xu[[0, 2]].
"""
from __future__ import annotations

import torch

from .base import F64, ResidualEnv


class Pendulum(ResidualEnv):
    env_id = 0                                   # GPMPC_ENV_PENDULUM1D
    pad_g = [0, 1, 3]                            # value, d/dtheta, d/du  inside [val | dx (2) | du (1)]
    g_idx_inputs = [0, 2]

    def __init__(self, params, device=None):
        super().__init__(params, device)
        self.l = float(params["env"]["params"]["l"])
        self.g = float(params["env"]["params"]["g"])
        self.B_d = torch.tensor([[0.0], [1.0]], dtype=F64, device=self.torch_device)
        self.env_params = (self.l, self.g)

    def _grid_axes(self):
        o, e = self.params["optimizer"], self.params["env"]
        return (torch.linspace(o["x_min"][0], o["x_max"][0], e["n_data_x"], dtype=F64),
                torch.linspace(o["u_min"][0], o["u_max"][0], e["n_data_u"], dtype=F64))

    def unknown_dyn(self, xu):
        """residual increment of omega for rows (theta, u): (n, 2) -> (n, 1)"""
        assert xu.shape[1] == 2
        return -self.g * torch.sin(xu[:, [0]]) * self.dt / self.l + xu[:, [1]] * self.dt

    def get_prior_data(self, x_hat):
        """labels [g, dg/dtheta, dg/du] per training input: (n, 2) -> (1, n, 3)"""
        n = x_hat.shape[0]
        y = torch.zeros((self.g_ny, n, 1 + self.g_nx + self.g_nu), dtype=F64, device=x_hat.device)
        y[0, :, 0] = self.unknown_dyn(x_hat)[:, 0]
        y[0, :, 1] = (-self.g * torch.cos(x_hat[:, 0]) / self.l) * self.dt
        y[0, :, 2] = self.dt
        return y

    def known_dyn(self, xu):
        """(Ns, nx, H, nx+nu) -> (Ns, nx, H): theta + omega dt, omega (row 0 of dim 1 is read)"""
        assert xu.dim() == 4 and xu.shape[1] == self.nx and xu.shape[3] == self.nx + self.nu
        th, om = xu[:, [0], :, 0], xu[:, [0], :, 1]
        return torch.cat([th + om * self.dt, om], dim=1)

    def get_f_known_jacobian(self, xu):
        """[f | df/dx | df/du] of the known part: (Ns, nx, H, 1+nx+nu)"""
        ns, nH = xu.shape[0], xu.shape[2]
        J = torch.zeros((ns, self.nx, nH, 1 + self.nx + self.nu), dtype=F64, device=xu.device)
        J[..., 0] = self.known_dyn(xu)
        J[:, 0, :, 1] = 1.0
        J[:, 0, :, 2] = self.dt
        J[:, 1, :, 2] = 1.0
        return J

    def transform_sensitivity(self, dg_dxu_grad, xu_hat):
        return dg_dxu_grad                        # constant B_d: nothing to transform

    def _B_d_of(self, xu):
        return torch.tensor([[0.0], [1.0]], dtype=F64)
