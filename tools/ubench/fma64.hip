// micro-benchmark: v_fma_f64 issue rate / dependent latency, v_readlane -> fma, ds_read_b128 broadcast; 1 wave per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double2_t __attribute__((ext_vector_type(2)));
template <int CH>
__global__ void k_fma(double* out, long long* cyc, int iters, double a, double b) {
    double acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = threadIdx.x * 1e-3 + c;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = fma(acc[c], a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int c = 0; c < CH; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_readlane(double* out, long long* cyc, int iters, double a) {
    double v0 = threadIdx.x * 1e-3, v1 = v0 + 1, v2 = v0 + 2;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int p = (i * 8 + r) & 63;
            int lo0 = __builtin_amdgcn_readlane(__double2loint(v0), p), hi0 = __builtin_amdgcn_readlane(__double2hiint(v0), p);
            int lo1 = __builtin_amdgcn_readlane(__double2loint(v1), p), hi1 = __builtin_amdgcn_readlane(__double2hiint(v1), p);
            int lo2 = __builtin_amdgcn_readlane(__double2loint(v2), p), hi2 = __builtin_amdgcn_readlane(__double2hiint(v2), p);
            v0 = fma(-a, __hiloint2double(hi0, lo0), v0);
            v1 = fma(-a, __hiloint2double(hi1, lo1), v1);
            v2 = fma(-a, __hiloint2double(hi2, lo2), v2);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds(double* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) double buf[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = i * 1e-3;
    __syncthreads();
    double a0 = 0, a1 = 0, a2 = 0;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const double2_t x = *reinterpret_cast<const double2_t*>(buf + ((i * 8 + r) * 6 & 1000));
            const double2_t y = *reinterpret_cast<const double2_t*>(buf + ((i * 8 + r) * 6 & 1000) + 2);
            const double2_t z = *reinterpret_cast<const double2_t*>(buf + ((i * 8 + r) * 6 & 1000) + 4);
            a0 = fma(x.x, 1.0001, a0); a0 = fma(x.y, 1.0002, a0);
            a1 = fma(y.x, 1.0001, a1); a1 = fma(y.y, 1.0002, a1);
            a2 = fma(z.x, 1.0001, a2); a2 = fma(z.y, 1.0002, a2);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    long long h; const int iters = 2000;
#define RUN(K, name, per)  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(K, dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.0000001, 1e-9); hipDeviceSynchronize(); } hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-28s %8.2f cycles per %s\n", name, (double)h / (iters * 8.0), per);
    RUN(k_fma<1>, "fma64 1 chain", "fma (latency)");
    RUN(k_fma<2>, "fma64 2 chains", "2 fma");
    RUN(k_fma<3>, "fma64 3 chains", "3 fma");
    RUN(k_fma<4>, "fma64 4 chains", "4 fma");
    RUN(k_fma<8>, "fma64 8 chains", "8 fma");
    RUN(k_fma<16>, "fma64 16 chains", "16 fma");
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_readlane, dim3(256), dim3(256), 0, 0, out, cyc, iters, 1e-9); hipDeviceSynchronize(); }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-28s %8.2f cycles per pivot (6 readlane + 3 fma)\n", "readlane->fma x3", (double)h / (iters * 8.0));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-28s %8.2f cycles per (3 b128 bcast + 6 fma)\n", "lds b128 bcast + fma", (double)h / (iters * 8.0));
    return 0;
}
