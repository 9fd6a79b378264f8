// Storage layout of a chain's appended-slot factor L_hh shared by the tuned rollout kernels
// (rollout_fast.hip: one wave per chain; rollout_split.hip: one wave per right-hand side).
//
// L_hh is stored ROW-major (row r = its r strictly-lower entries, 16-byte aligned) and COLUMN-SCALED
// (L''[r][p] = L[r][p] / L[p][p]); 1/L_pp lives with the lane that owns row p.  The storage is zero-initialised.
#pragma once
#include <type_traits>
#include <utility>

#include "gpmpc_device.hpp"

namespace gpmpc {

typedef double double2_t __attribute__((ext_vector_type(2)));

// row-major, every row start 16-byte aligned: row r holds r entries in a slot of r rounded up to even
__host__ __device__ __forceinline__ int lhh_rowofs(int r) {
    const int h = r >> 1;
    return (r & 1) ? 2 * h * (h + 1) : 2 * h * h;
}
// + slack for the look-ahead reads of the last row (8 pivots ahead of a pivot index rounded up to 8)
__host__ __device__ __forceinline__ long lhh_doubles(int nh_max) {
    const int r = nh_max - 1, cap = r + (r & 1);
    int slack = ((nh_max + 7) & ~7) + 8 - cap;
    slack = (slack < 0) ? 0 : ((slack + 1) & ~1);
    return lhh_rowofs(nh_max) + slack;
}

// ---------------------------------------------------------------------------------------------------------------
// DPP forward substitution (rollout_fast.hip).  A wave64 is four DPP rows of 16 lanes.  In the "row layout" lane r of
// register b holds right-hand side b at matrix row r; in the "DPP layout" lane (g, i) of register k holds right-hand
// side g at matrix row 16 k + i.  The two are a 4x4 transpose of 16-lane blocks, done with the gfx950 lane swaps:
//   v_permlane32_swap a, b : a's rows 2,3 <-> b's rows 0,1        v_permlane16_swap a, b : a's odd rows <-> b's even rows
// In the DPP layout the pivot broadcast of the substitution is a DPP operand modifier (row_newbcast:i = lane i of every
// row, i.e. each right-hand side broadcasts its own pivot value), so the per-pivot dependency chain is one
// v_fmac_f64_dpp (17.5 cycles measured, tools/ubench/dpp64.hip) instead of 2 v_readlane_b32 per right-hand side
// feeding the FMAs through SGPRs (~117 cycles per pivot for 3 right-hand sides).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lane_swap32(double& a, double& b) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void lane_swap16(double& a, double& b) {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}
// r[b] (row layout, 64 matrix rows) <-> r[k] (DPP layout, banks k = 0..3); an involution
__device__ __forceinline__ void transpose_rows_dpp(double (&r)[4]) {
    lane_swap32(r[0], r[2]);
    lane_swap32(r[1], r[3]);
    lane_swap16(r[0], r[1]);
    lane_swap16(r[2], r[3]);
}

// acc -= l * (lane PI of every DPP row of piv).  The s_nop pair covers the "VALU writes a VGPR -> DPP reads it" hazard
// (2 wait states) on both sides, because the compiler's hazard recogniser does not look inside inline assembly:
// subst_diag is the chain (acc is its own pivot source), subst_off needs piv written at least one subst_diag earlier.
#ifdef GPMPC_PHASE_TIMERS
#define GPMPC_ASM asm volatile          // keeps the statements between the s_memtime reads of the phase timers
#else
#define GPMPC_ASM asm
#endif
template <int PI>
__device__ __forceinline__ void subst_diag(double& acc, double l) {
    GPMPC_ASM("s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%1 row_newbcast:%2 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
        : "+v"(acc)
        : "v"(l), "n"(PI));
}
template <int PI>
__device__ __forceinline__ void subst_off(double& acc, double piv, double l) {
    GPMPC_ASM("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(piv), "v"(l), "n"(PI));
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace gpmpc
