// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "gpmpc_device.hpp"
#include "gpmpc_hip.h"

namespace gpmpc {

std::string& last_error();   // thread-local, defined in capi.hip

inline int fail(int code, const std::string& msg) {
    last_error() = msg;
    return code;
}

#define GPMPC_HIP_CHECK(expr)                                                                            \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            return ::gpmpc::fail(GPMPC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
    } while (0)

inline int observed_real_slots(const gpmpc_gp_desc_t* gp) { return gp->real_has_grad ? gp->N_r * gp->T : gp->N_r; }

inline int check_gp(const gpmpc_gp_desc_t* gp) {
    if (!gp) return fail(GPMPC_E_ARG, "gp descriptor is NULL");
    if (gp->g_ny < 1 || gp->g_ny > GPMPC_MAX_NY) return fail(GPMPC_E_ARG, "g_ny out of range");
    if (gp->D < 1 || gp->D > GPMPC_MAX_D) return fail(GPMPC_E_ARG, "D out of range");
    if (!(gp->T == 1 || gp->T == gp->D + 1)) return fail(GPMPC_E_ARG, "T must be 1 or 1+D");
    if (gp->N_r < 1) return fail(GPMPC_E_ARG, "N_r must be >= 1");
    if (gp->real_has_grad && gp->T == 1) return fail(GPMPC_E_ARG, "real_has_grad needs T = 1+D");
    if ((gp->grid_n0 || gp->grid_n1) && ((long)gp->grid_n0 * gp->grid_n1 != gp->N_r || gp->D != 2))
        return fail(GPMPC_E_ARG, "grid_n0 * grid_n1 must equal N_r (D = 2)");
    return GPMPC_OK;
}

inline GpParams make_gp_params(const gpmpc_gp_desc_t* gp) {
    GpParams p;
    std::memset(&p, 0, sizeof(p));
    p.g_ny = gp->g_ny;
    p.D = gp->D;
    p.T = gp->T;
    p.N_r = gp->N_r;
    p.real_has_grad = gp->real_has_grad;
    p.n_r = observed_real_slots(gp);
    p.grid_n0 = gp->grid_n0;
    p.grid_n1 = gp->grid_n1;
    for (int o = 0; o < gp->g_ny; ++o) {
        for (int d = 0; d < gp->D; ++d) p.inv_l2[o][d] = 1.0 / (gp->ell[o][d] * gp->ell[o][d]);
        p.os[o] = gp->outputscale[o];
    }
    for (int t = 0; t < gp->T; ++t) p.noise[t] = gp->noise[t];
    p.jitter = gp->jitter;
    p.var_floor = gp->var_floor;
    p.plan_stride = plan_doubles_per_output(p.n_r, p.grid_n0, p.grid_n1);
    return p;
}

inline EnvParams make_env_params(const gpmpc_env_desc_t* env) {
    EnvParams e;
    std::memset(&e, 0, sizeof(e));
    e.env_id = env->env_id;
    e.nx = env->nx;
    e.nu = env->nu;
    e.use_feedback = env->use_feedback;
    e.dt = env->dt;
    e.p0 = env->p0;
    e.p1 = env->p1;
    for (int i = 0; i < GPMPC_MAX_NU; ++i)
        for (int j = 0; j < GPMPC_MAX_NX; ++j) e.K[i][j] = env->K[i][j];
    for (int j = 0; j < GPMPC_MAX_NX; ++j) e.x_goal[j] = env->x_goal[j];
    return e;
}

inline int check_env(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env) {
    if (!env) return fail(GPMPC_E_ARG, "env descriptor is NULL");
    if (env->env_id == GPMPC_ENV_PENDULUM1D) {
        if (env->nx != 2 || env->nu != 1 || gp->g_ny != 1 || gp->D != 2)
            return fail(GPMPC_E_ARG, "pendulum1D needs nx=2 nu=1 g_ny=1 D=2");
    } else if (env->env_id == GPMPC_ENV_CAR_RESIDUAL) {
        if (env->nx != 4 || env->nu != 2 || gp->g_ny != 3 || gp->D != 2)
            return fail(GPMPC_E_ARG, "car_residual needs nx=4 nu=2 g_ny=3 D=2");
    } else {
        return fail(GPMPC_E_ARG, "unknown env_id");
    }
    return GPMPC_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace gpmpc
