"""Kinematic bicycle with a 3-output residual GP (reference src/environments/car_model_residual.py, alias bicycle_Bdx).

State (X, Y, phi, v), input (delta, a).  Known part f = (X, Y, phi, v + a dt); residual
g(phi, delta) = (cos(phi+beta) dt, sin(phi+beta) dt, sin(beta) dt / lr), beta = atan(tan(delta) lr / (lr+lf)),
enters as v * I_{4x3} g.  GP input = (phi, delta) = xu[[2, 4]].
"""
from __future__ import annotations

import torch

from .base import F64, ResidualEnv


class CarKinematicsModel(ResidualEnv):
    env_id = 1                                   # GPMPC_ENV_CAR_RESIDUAL
    pad_vg = [0, 1, 3]
    pad_g = [0, 3, 4, 5]                         # value, d/dphi, d/dv, d/ddelta inside [val | dx (4) | du (2)]
    g_idx_inputs = [2, 4]
    has_nominal_model = True

    def __init__(self, params, device=None):
        super().__init__(params, device)
        self.H = params["optimizer"]["H"]
        self.lf = float(params["env"]["params"]["lf"])
        self.lr = float(params["env"]["params"]["lr"])
        self.B_d = torch.eye(self.nx, self.g_ny, dtype=F64, device=self.torch_device)
        self.env_params = (self.lf, self.lr)

    def _grid_axes(self):
        o, e = self.params["optimizer"], self.params["env"]
        return (torch.linspace(o["x_min"][2], o["x_max"][2], e["n_data_x"], dtype=F64),
                torch.linspace(o["u_min"][0], o["u_max"][0], e["n_data_u"], dtype=F64))

    def _beta(self, delta):
        return torch.atan(torch.tan(delta) * self.lr / (self.lr + self.lf))

    def unknown_dyn(self, xu):
        """rows (phi, delta): (n, 2) -> (n, 3) residual per unit speed"""
        assert xu.shape[1] == 2
        phi, beta = xu[:, [0]], self._beta(xu[:, [1]])
        return torch.hstack([torch.cos(phi + beta) * self.dt, torch.sin(phi + beta) * self.dt,
                             torch.sin(beta) * self.dt / self.lr])

    def get_prior_data(self, xu):
        """labels [g, dg/dphi, dg/ddelta] for the three outputs: (n, 2) -> (3, n, 3)"""
        assert xu.shape[1] == self.g_nx + self.g_nu
        phi, delta = xu[:, 0], xu[:, 1]
        g = self.unknown_dyn(xu)
        y = torch.zeros((self.g_ny, xu.shape[0], 1 + self.g_nx + self.g_nu), dtype=F64, device=xu.device)
        y[:, :, 0] = g.transpose(0, 1)
        b_in = (self.lr * torch.tan(delta)) / (self.lf + self.lr)
        beta = torch.atan(b_in)
        dbeta = ((self.lr / (torch.cos(delta) ** 2)) / (self.lf + self.lr)) / (1 + b_in ** 2)
        s, c = torch.sin(phi + beta) * self.dt, torch.cos(phi + beta) * self.dt
        y[0, :, 1], y[0, :, 2] = -s, -s * dbeta
        y[1, :, 1], y[1, :, 2] = c, c * dbeta
        y[2, :, 2] = torch.cos(beta) * self.dt * dbeta / self.lr
        return y

    def known_dyn(self, xu):
        assert xu.dim() == 4 and xu.shape[1] == self.nx and xu.shape[3] == self.nx + self.nu
        r = xu[:, [0], :, :]
        return torch.cat([r[..., 0], r[..., 1], r[..., 2], r[..., 3] + r[..., 5] * self.dt], dim=1)

    def known_dyn_xu(self, xu):
        return self.known_dyn(xu.reshape(1, 1, 1, -1).expand(1, self.nx, 1, -1))[0, :, :]

    def get_f_known_jacobian(self, xu):
        ns, nH = xu.shape[0], xu.shape[2]
        J = torch.zeros((ns, self.nx, nH, 1 + self.nx + self.nu), dtype=F64, device=xu.device)
        J[..., 0] = self.known_dyn(xu)
        for i in range(self.nx):
            J[:, i, :, 1 + i] = 1.0
        J[:, 3, :, 6] = self.dt
        return J

    def transform_sensitivity(self, dg_dxu_grad, xu_hat):
        """[g, dg/dphi, dg/ddelta] -> [v g, v dg/dphi, g, v dg/ddelta]  (B_d(x) = v I folded into the sample)"""
        ns, nH = dg_dxu_grad.shape[0], dg_dxu_grad.shape[2]
        out = torch.zeros((ns, self.g_ny, nH, 4), dtype=F64, device=dg_dxu_grad.device)
        out[..., self.pad_vg] = xu_hat[:, 0:3, :, [3]] * dg_dxu_grad
        out[..., 2] = dg_dxu_grad[..., 0]
        return out

    def unknown_dyn_Bd_fun(self, xu):
        return xu[:, [3]] * torch.eye(self.nx, self.g_ny, dtype=xu.dtype, device=xu.device)

    def _B_d_of(self, xu):
        return self.unknown_dyn_Bd_fun(xu)
