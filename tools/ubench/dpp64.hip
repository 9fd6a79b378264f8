// Semantics + latency check of the gfx950 pieces behind the DPP forward substitution:
//   v_permlane32_swap / v_permlane16_swap as a 4x4 transpose of 16-lane blocks, and v_fmac_f64_dpp row_newbcast.
// build: hipcc --offload-arch=gfx950 -O3 -w tools/ubench/dpp64.hip -o /tmp/dpp64 ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void swap32(double& a, double& b) {
    auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void swap16(double& a, double& b) {
    auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void transpose4(double (&r)[4]) {
    swap32(r[0], r[2]);
    swap32(r[1], r[3]);
    swap16(r[0], r[1]);
    swap16(r[2], r[3]);
}

__global__ void layout_kernel(double* out) {
    const int lane = threadIdx.x;
    double r[4];
    for (int b = 0; b < 4; ++b) r[b] = b * 1000 + lane;
    transpose4(r);
    for (int k = 0; k < 4; ++k) out[k * 64 + lane] = r[k];
    double acc = 0.0, v = 100 + lane, l = 1.0;
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(l));
    out[256 + lane] = acc;
}

// dependent chain of DPP fmacs: cycles per link
__global__ void chain_kernel(double* out, long long* cyc, int iters) {
    double v = 1.0 + threadIdx.x * 1e-3, l = 1e-3;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define LINK(n) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(v) : "v"(l));
        LINK(0) LINK(1) LINK(2) LINK(3) LINK(4) LINK(5) LINK(6) LINK(7)
        LINK(8) LINK(9) LINK(10) LINK(11) LINK(12) LINK(13) LINK(14) LINK(15)
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double* d;
    long long* c;
    hipMalloc(&d, 512 * sizeof(double));
    hipMalloc(&c, 8);
    layout_kernel<<<1, 64>>>(d);
    std::vector<double> h(512);
    hipMemcpy(h.data(), d, 320 * sizeof(double), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int k = 0; k < 4; ++k)
        for (int lane = 0; lane < 64; ++lane) {
            const int g = lane / 16, i = lane % 16;
            if (h[k * 64 + lane] != g * 1000 + 16 * k + i) ++bad;
        }
    printf("transpose4: %d mismatches (V_k[g][i] == R_g[16k+i])   sample V_1 lanes 0,17,35: %g %g %g\n", bad, h[64], h[64 + 17], h[64 + 35]);
    int bad2 = 0;
    for (int lane = 0; lane < 64; ++lane)
        if (h[256 + lane] != 100 + (lane / 16) * 16 + 5) ++bad2;
    printf("row_newbcast:5: %d mismatches (acc == v[16*row + 5])   lanes 0,20,63: %g %g %g\n", bad2, h[256], h[256 + 20], h[256 + 63]);
    chain_kernel<<<1, 64>>>(d, c, 1000);
    chain_kernel<<<1, 64>>>(d, c, 1000);
    long long cy;
    hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("dependent v_fmac_f64_dpp chain (with s_nop 1): %.1f cycles per link\n", cy / 16000.0);
    return 0;
}
