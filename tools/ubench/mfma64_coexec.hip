// Do FP64 MFMAs and FP64 VALU instructions co-execute on a gfx950 SIMD?  One workgroup of 8 waves on one CU = two waves per SIMD (launch
// bounds force 256 registers each).  Modes: (m) every wave issues MFMAs only, (v) every wave issues v_fma_f64 only, (x) of each SIMD's
// two waves one issues MFMAs and the other v_fma_f64 (wave index parity), (i) every wave interleaves one MFMA with VPM v_fma_f64.
// If the two kinds shared nothing, (x) would take max(m-half, v-half); if they share the pipe, the sum.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma64_coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NM = 64;        // MFMAs (16x16x4 FP64: 2048 FLOP each) per wave and iteration
constexpr int NV = 1024;      // v_fma_f64 (128 FLOP each) per wave and iteration: 64 x 16 independent chains... same FLOP as NM MFMAs
template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, long long* cyc, int iters) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = 1.0 + l * 1e-3 + i;
    const double a = 1.0 + 1e-9 * l, b = 1e-9;
    // waves of a workgroup go to the SIMDs in a cyclic order: waves w and w + 4 share a SIMD -> the kind by (w >> 2)
    // (mode 4: the kind by the wave's parity instead - whichever way the waves are dealt to the SIMDs, one of the two mixes them on a SIMD)
    const bool do_m = MODE == 0 || (MODE == 2 && (w >> 2) == 0) || (MODE == 4 && (w & 1) == 0);
    const bool do_v = MODE == 1 || (MODE == 2 && (w >> 2) == 1) || (MODE == 4 && (w & 1) == 1);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV / NM; ++j) f[j] = __builtin_fma(f[j], a, b);
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 3], 0, 0, 0);
            }
            if (do_v) {
#pragma unroll
                for (int i = 0; i < NV; ++i) f[i & 15] = __builtin_fma(f[i & 15], a, b);
            }
        }
    }
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += f[i];
    out[threadIdx.x] = s + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
int main() {
    double* o; long long* c;
    hipMalloc(&o, 512 * 8); hipMalloc(&c, 8);
    const int iters = 200;
    const char* name[5] = {"all 8 waves: MFMA only            ", "all 8 waves: v_fma_f64 only        ", "waves 0-3 MFMA, waves 4-7 FMA     ",
                           "all 8 waves: 1 MFMA : 16 FMA interleaved", "even waves MFMA, odd waves FMA    "};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 5; ++mode) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(512), 0, 0, o, c, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(512), 0, 0, o, c, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(512), 0, 0, o, c, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(512), 0, 0, o, c, iters);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(512), 0, 0, o, c, iters);
            hipDeviceSynchronize();
            long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            if (rep == 1) {
                // per SIMD and iteration: modes 0 / 1: two waves x (NM MFMAs | NV FMAs); mode 2: NM MFMAs + NV FMAs; mode 3: two waves x both
                printf("%s  %8.1f cycles per iteration (per SIMD: %s)\n", name[mode], (double)cy / iters,
                       mode == 0 ? "128 MFMAs" : mode == 1 ? "2048 FMAs" : mode == 3 ? "128 MFMAs + 2048 FMAs" : "64 MFMAs + 1024 FMAs if the kinds meet on a SIMD");
            }
        }
    return 0;
}
