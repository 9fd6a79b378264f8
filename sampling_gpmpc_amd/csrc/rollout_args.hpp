// Kernel-argument block shared by the rollout kernels (generic: rollout.hip, tuned: rollout_fast.hip).
#pragma once
#include "gpmpc_device.hpp"

namespace gpmpc {

struct RolloutArgs {
    GpParams gp;
    EnvParams env;
    const double* plan;
    const double* X_r;
    int mode, hall_tasks;
    double var_zero_thr, beta;
    long Ns;
    int H;
    const double* x0;
    int x0_per_sample;
    const double* u_ff;
    const double* z;
    long z_step_stride;
    double* X_traj;
    double* Y;
    double* Xi;
    int* info;
    double* ws;
    long ws_chain_stride;   // doubles per chain in the HBM workspace
    int nh_max;             // hallucinated slots allocated per chain
    int lds_shared;         // doubles of block-shared LDS
    int lds_per_wave;       // doubles of per-wave LDS
    int linv_in_lds;        // generic kernel: L_rr^-1 staged per wave in LDS (else read from the plan in HBM/L2)
    // generic kernel only: conditioning points the chains start from, and the exported / resumed factor state
    const double* X_h0;     // (Ns, g_ny, n_h0, D) seed points, all T tasks observed (NULL: none)
    const double* Y_h0;     // (Ns, g_ny, n_h0, T)
    int n_h0;
    const double* X_v0;     // (Ns, g_ny, n_v0, D) further seed points observed with hall_tasks tasks (value-only when 1)
    const double* Y_v0;     // (Ns, g_ny, n_v0, T): the first hall_tasks entries of a row are used
    int n_v0;
    double* state;          // per-sample factor state (gpmpc_rollout_state_bytes), NULL: not kept
    long state_stride;      // doubles per sample
    int state_points;       // point capacity of the state
    int resume;             // 1: the chains continue from `state` (no seeds)
    int max_points;         // capacity of the LDS point list (seed + resumed + H)
};

// factor state of one sample: [header 4 | points state_points x D | per chain: LhrT n_r x slots | L_hh packed | w | 1/diag]
__host__ __device__ __forceinline__ long state_chain_doubles(int n_r, int slots) {
    return (long)n_r * slots + ((long)slots * (slots + 1)) / 2 + 2L * slots;
}
__host__ __device__ __forceinline__ long state_sample_doubles(int g_ny, int n_r, int slots, int points, int D) {
    return 4 + (long)points * D + (long)g_ny * state_chain_doubles(n_r, slots);
}

// packed lower-triangular, column-major: element (row, col) at col_ofs(col) + row - col, rows col..nh_max-1
__host__ __device__ __forceinline__ long col_ofs(int p, int nh_max) { return (long)p * nh_max - ((long)p * (p - 1)) / 2; }

// gpmpc_rollout_pin_kernel (rollout.hip): GPMPC_KERNEL_AUTO or the kernel every launch must take where its shape allows
extern int g_rollout_pin;

// tuned path (rollout_fast.hip)
bool rollout_fast_eligible(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int mode, int hall_tasks, int H);
size_t rollout_fast_workspace_bytes(const gpmpc_gp_desc_t* gp, int64_t Ns, int H);
int rollout_fast_launch(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, RolloutArgs& args, void* ws,
                        size_t ws_bytes, hipStream_t st);

// throughput path: four chains per wave, forward substitution on the FP64 matrix pipe (rollout_tiles.hip)
bool rollout_tiles_eligible(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int mode, int hall_tasks, int H, int64_t Ns,
                            int n_h0 = 0, int n_v0 = 0);
size_t rollout_tiles_workspace_bytes(const gpmpc_gp_desc_t* gp, int64_t Ns, int H, int n_pre = 0);
int rollout_tiles_launch(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, RolloutArgs& args, void* ws, size_t ws_bytes,
                         hipStream_t st);

// latency path: one chain per wave, the chain's factor in AGPR panels, forward substitution on the FP64 matrix pipe (rollout_one.hip)
bool rollout_one_eligible(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int mode, int hall_tasks, int H, int64_t Ns);
int rollout_one_launch(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, RolloutArgs& args, hipStream_t st);

// mode-I thread-per-sample path (rollout_indep.hip)
bool rollout_indep_eligible(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int mode);
int rollout_indep_launch(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, RolloutArgs& args, hipStream_t st);

}  // namespace gpmpc
