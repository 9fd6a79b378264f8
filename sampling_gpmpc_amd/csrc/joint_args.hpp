// Kernel arguments shared by the translation units of gpmpc_joint_sample (joint.hip: factor extension, VALU test rows,
// root + sample; joint_mfma.hip: the test rows on the FP64 matrix pipe).
#pragma once
#include "gpmpc_host.hpp"

namespace gpmpc {

// phases of joint_kernel (JointArgs::phase)
enum : int {
    JOINT_PHASE_ALL = 0,      // one launch: factor rows, w row, test rows, mean, S, root, sample (the VALU path)
    JOINT_PHASE_FACTOR = 1,   // the hallucinated rows n_c .. n_ho - 1 of the factor only (into M and the factor cache)
    JOINT_PHASE_TAIL = 2,     // root + sample only: S is in Sall, the mean in `mean` (written by joint_test_mfma_kernel)
    JOINT_PHASE_HEAD = 4,     // everything but the tail: factor rows, test rows, mean (into `mean`) and S (into Sall); the root and the
                              // sample follow in joint_tail_mfma_kernel (joint_chol.hip)
    JOINT_PHASE_CHOL = 3      // the new hallucinated rows against the NEW columns only: blocked Cholesky of the Schur complement
                              // joint_test_mfma_kernel (JOINT_MFMA_FACTOR) left in Sall (leading dimension n_ho - n_c); the rows'
                              // entries against the old columns are already in the cache
};
// JOINT_MFMA_TEST_TOP / _BOTTOM: the test rows of a conditioning set of more than JOINT_MFMA_SPLIT slots in two launches - TOP is
// the test mode over the first JOINT_MFMA_SPLIT slots (a leading block of the factor is the factor of the leading block); it leaves
// the partial S and mean in their buffers and its X tiles in `xbuf`; BOTTOM conditions on the remaining slots: its right-hand side is
// K_bottom - L_21 X_top (the factor's rows of the remaining slots against the first ones, streamed like every other tile), its Gram
// phase starts from TOP's partial results.
enum : int { JOINT_MFMA_TEST = 0, JOINT_MFMA_FACTOR = 1, JOINT_MFMA_TEST_TOP = 2, JOINT_MFMA_TEST_BOTTOM = 3 };
constexpr int JOINT_MFMA_SPLIT = 416;           // conditioning slots of one launch of joint_test_mfma_kernel (26 tiles of 16)
constexpr int JOINT_MFMA_BOTTOM_MAX = 128;      // slots of the BOTTOM launch at most (its stream table holds 26 x 8 + 28 tiles)
constexpr long JOINT_MFMA_XBUF_DOUBLES = (long)(JOINT_MFMA_SPLIT / 16) * 8 * 256;   // TOP's X tiles of one chain: [tile][wave][4][64]

struct JointArgs {
    GpParams gp;
    const double* plan;
    const double* X_r;
    long Ns;
    int n_h;
    const double* X_h;
    const double* Y_h;
    const int* h_slots;
    int n_ho;
    int m;
    const double* X_s;
    const double* z;
    double var_zero_thr, beta;
    int apply_clip;
    double* mean;
    double* var;
    double* y;
    double* covar;
    int* info;
    double* ws;
    long ws_chain_stride;   // doubles
    int ld;                 // rows of M (padded)
    double* Sall;           // [chains][mT*mT] posterior covariance, column-major, both triangles written
    // where the READERS of the test rows' covariance (joint_tail_mfma_kernel, joint_eigh_kernel) find it, and where joint_test_mfma_kernel's
    // test mode writes it: chain c at Sv + (c - Sv_chain_base) * Sv_cs, element (r, c2) at r * Sv_ld + c2.  Default: Sall, mT, mT * mT, 0.  With
    // pending rows written (pend_write) it is the diagonal block of the cache rows the next call's new slots will occupy - S is written
    // ONCE (115 KB per chain at the configs[4] shard) instead of into Sall and into the cache
    double* Sv;
    long Sv_cs, Sv_chain_base;
    int Sv_ld;
    int* any_fail;          // set when a chain's jitter chain failed (read by joint_eigh_kernel)
    // factor cache (caller-owned, persists between calls): per chain the hallucinated rows of the factor, row-major
    // [rows_cap][fc_cs] (columns: real slots, then hallucinated slots; the diagonal blocks as block_factor left them) and
    // 1/diag [rows_cap].  The first n_c rows (any count: the column blocks restart at slot n_c) are valid on entry and are
    // not recomputed.  fc_cs is EVEN and so is rows_cap: every row starts on a 16-byte boundary (joint_test_mfma_kernel
    // moves 16-byte pieces of the rows straight into LDS).
    double* fcache;
    long fc_stride;         // doubles per chain
    int fc_cs;              // row stride = even(n_r + rows_cap)
    int fc_cap;             // rows_cap
    int n_c;
    long fc_chain_base;     // the cache entry of chain c is fcache + (c - fc_chain_base) * fc_stride (a temporary cache inside
                            // the workspace holds one batch of chains; the caller's cache all of them: 0)
    int abandon_root;       // 1 (GPMPC_ROOT_AUTO): the eigh kernel redraws the WHOLE batch once a chain has failed every retry -
                            // a chain that sees the flag stops its own Cholesky attempts (their result would be overwritten)
    int phase;              // JOINT_PHASE_*
    int info_in;            // JOINT_PHASE_TAIL: info[chain] already holds the factor phase's bits (OR into it)
    long chain0, chain1;    // the chains of this launch: [chain0, chain1)
    int mfma_mode;          // joint_test_mfma_kernel: JOINT_MFMA_TEST / _FACTOR / _TEST_TOP / _TEST_BOTTOM
    double* xbuf;           // TOP writes, BOTTOM reads: X tiles of the chains [chain0, chain1), JOINT_MFMA_XBUF_DOUBLES each
    // PENDING ROWS (round 6).  In the SQP loop the points a draw is made at become the next call's new hallucinated points
    // (reference src/agent.py:629-641 then :164-202), so the draw's own X = L^-1 K_o* IS the new rows' block against the old columns
    // and its S = K** - X^T X the Schur complement (up to the likelihood noise on the diagonal).  pend_write: the test-mode launch
    // also writes X^T into the cache rows n_ho .. n_ho + m T - 1 and S into their diagonal block.  pend_use: the caller vouches
    // that the cache rows n_c .. n_ho - 1 hold exactly that from the previous call - the factor extension is then only the
    // Cholesky of (block + noise), in place (joint_chol_mfma_kernel), no JOINT_MFMA_FACTOR launch.
    int pend_write, pend_use;
};

// joint_mfma.hip ---------------------------------------------------------------------------------------------------------
// true when joint_test_mfma_kernel is instantiated for these sizes: n_r observed real slots + n_hc hallucinated slots to condition
// on, ncols columns (test mode: m T + 1; factor mode: the new hallucinated rows), T tasks
bool joint_mfma_eligible(int n_r, int n_hc, int ncols, int T);
// true when the test rows of this conditioning set run as TOP + BOTTOM launches (JOINT_MFMA_SPLIT < n_r + n_hc <= + _BOTTOM_MAX)
bool joint_mfma_split_eligible(int n_r, int n_hc, int ncols, int T);
// launches joint_test_mfma_kernel for the chains [a.chain0, a.chain1): V^T = L^-1 K_o* (w column included), then
// mean = V^T w into a.mean and S = K** - V^T V into a.Sall.  Needs a.fcache with every hallucinated row filled.
int joint_mfma_launch(const JointArgs& a, hipStream_t st);

// joint_chol.hip ---------------------------------------------------------------------------------------------------------
// the Cholesky of the Schur complement JOINT_MFMA_FACTOR left in a.Sall (the new rows against the new columns: into the factor
// cache, 1 / diag, the chain's info word) on the matrix pipe, one wave per chain; instantiated for 1..128 new rows
bool joint_chol_mfma_eligible(int n_new);
int joint_chol_mfma_launch(const JointArgs& a, hipStream_t st);
// the TAIL of a draw (root with the jitter chain, sample, post-processing; S in a.Sall, the mean in a.mean) one wave per chain
bool joint_tail_mfma_eligible(int mT, int T);
int joint_tail_mfma_launch(const JointArgs& a, hipStream_t st);

// columns conditioned on the real data alone (joint_real_mfma_kernel): a.mfma_mode = JOINT_MFMA_FACTOR - the n_ho new hallucinated slots with
// nothing cached: X^T and the Schur complement before the noise into the cache, as a draw with GPMPC_PENDING_WRITE leaves them -, or
// JOINT_MFMA_TEST - a draw without hallucinated slots: mean into a.mean, S into a.Sall, info[chain] = 0
bool joint_real_mfma_eligible(int n_r, int N_r, int ncols, int P, int T, int D);      // P: points behind the columns (m / n_h)
int joint_real_mfma_launch(const JointArgs& a, hipStream_t st);

}  // namespace gpmpc
