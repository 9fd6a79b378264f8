// What does the memory system give the tiled rollout's streaming pattern?  One wave per workgroup, 40 KB of LDS per workgroup
// (four workgroups per CU, one wave per SIMD, as rollout_tiles_kernel), every wave re-reads ITS OWN lower-triangular tile
// matrix once per "step" (tile = 512 B, chain-major; the car leaves one 128-byte line of every tile untouched), 117 rows
// growing by three per step, the first 78 tiles resident (not read).  No arithmetic besides one add per tile, an optional
// pause per step stands in for the phases that do not stream.
//   hipcc --offload-arch=gfx950 -O3 -o stream_tiles.bin stream_tiles.hip && ./stream_tiles.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__host__ __device__ constexpr int tri(int r) { return r * (r + 1) / 2; }
typedef unsigned u32x2_v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_v __attribute__((ext_vector_type(4)));

// MODE bit 0: sc1 (device scope: misses the CU's L1), bit 1: 16 bytes per lane (two tiles per instruction)
template <int MODE, int UNROLL>
__global__ __launch_bounds__(64, 1) void stream_kernel(double* ws, long stride, int H, int resident_tiles, int live_chains, int pause,
                                                       int stagger, double* out, long long* cyc) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x, bm = (lane >> 2) & 3, kq = lane >> 4, jq = lane & 3;
    constexpr int AUX = (MODE & 1) ? 16 : 0;
    constexpr bool X4 = (MODE & 2) != 0;
    const bool live = bm < live_chains;
    char* base = reinterpret_cast<char*>(ws + blockIdx.x * stride);
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(stride * 8), 0x00020000);
    // x2: tile e at 512 e, lane's double at (bm*16 + kq*4 + jq)*8;  x4: tile pair at 1024 p, lane's two doubles at (bm*16 + kq*4 + jq)*16
    const unsigned voff = live ? (unsigned)(bm * 16 + kq * 4 + jq) * (X4 ? 16u : 8u) : 0x7ffff000u;
    double acc0 = 0.0, acc1 = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    if (stagger > 0 && blockIdx.x < 1024) {                       // first round of waves only: later rounds inherit the offsets
        const long long d = (long long)((blockIdx.x >> 8) & 3) * stagger;   // blocks b, b + 256, b + 512, b + 768 share a CU
        while (__builtin_readcyclecounter() - t0 < d) __builtin_amdgcn_s_sleep(8);
    }
    for (int t = 1; t < H; ++t) {
        const int n_h = 3 * t, nt = (n_h + 3) >> 2;
        const int hi = tri(nt), lo = resident_tiles < hi ? resident_tiles : hi;
        if constexpr (!X4) {
            int e = lo;
            for (; e + UNROLL <= hi; e += UNROLL) {
                double v[UNROLL];
#pragma unroll
                for (int k = 0; k < UNROLL; ++k) v[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, (e + k) * 512, AUX));
#pragma unroll
                for (int k = 0; k < UNROLL; ++k) ((k & 1) ? acc1 : acc0) += v[k];
            }
            for (; e < hi; ++e) acc0 += __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, e * 512, AUX));
        } else {
            int e = lo >> 1;
            const int hp = hi >> 1;
            for (; e + UNROLL / 2 <= hp; e += UNROLL / 2) {
                u32x4_v v[UNROLL / 2];
#pragma unroll
                for (int k = 0; k < UNROLL / 2; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, (e + k) * 1024, AUX);
#pragma unroll
                for (int k = 0; k < UNROLL / 2; ++k) {
                    acc0 += __builtin_bit_cast(double, u32x2_v{v[k].x, v[k].y});
                    acc1 += __builtin_bit_cast(double, u32x2_v{v[k].z, v[k].w});
                }
            }
            for (; e < hp; ++e) {
                const u32x4_v v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, e * 1024, AUX);
                acc0 += __builtin_bit_cast(double, u32x2_v{v.x, v.y});
                acc1 += __builtin_bit_cast(double, u32x2_v{v.z, v.w});
            }
        }
        if (pause > 0) {
            const long long p0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - p0 < pause) __builtin_amdgcn_s_sleep(8);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 64 + lane] = acc0 + acc1 + smem[lane];
}

template <int MODE, int UNROLL>
static void run(const char* name, double* ws, long stride, int waves, int H, int resident, int live, int pause, int stagger, double* out, long long* cyc) {
    auto k = stream_kernel<MODE, UNROLL>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 40000);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(waves), dim3(64), 40000, 0, ws, stride, H, resident, live, pause, stagger, out, cyc);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(waves), dim3(64), 40000, 0, ws, stride, H, resident, live, pause, stagger, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    double tiles = 0;
    for (int t = 1; t < H; ++t) {
        const int nt = (3 * t + 3) >> 2;
        tiles += tri(nt) - (resident < tri(nt) ? resident : tri(nt));
    }
    const double bytes = tiles * 128.0 * live * waves;
    std::printf("%-28s waves %5d live %d pause %5d stagger %5d: %7.3f ms  %6.2f TB/s useful  (%.2f GB, %.0f tiles per wave)\n", name, waves, live, pause, stagger, ms,
                bytes / ms * 1e-9, bytes * 1e-9, tiles);
}

int main(int argc, char** argv) {
    const int H = 40, waves_max = 4096;
    const long stride = (long)tri(32) * 64;                       // doubles per wave
    double *ws, *out;
    long long* cyc;
    hipMalloc(&ws, (size_t)waves_max * stride * 8);
    hipMemset(ws, 0, (size_t)waves_max * stride * 8);
    hipMalloc(&out, (size_t)waves_max * 64 * 8);
    hipMalloc(&cyc, 64);
    std::printf("workspace %.2f GB\n", waves_max * stride * 8e-9);
    const bool full = argc > 1;
    for (int waves : {256, 512, 1024, 2048, 4096}) {
        run<1, 16>("x2 sc1 unroll 16", ws, stride, waves, H, 78, 3, 0, 0, out, cyc);
        run<1, 32>("x2 sc1 unroll 32", ws, stride, waves, H, 78, 3, 0, 0, out, cyc);
        run<3, 16>("x4 sc1 unroll 16", ws, stride, waves, H, 78, 3, 0, 0, out, cyc);
    }
    // staggered starts: do the pauses of some waves overlap with the streaming of the others?
    for (int stagger : {0, 3000, 6000, 12000, 24000, 48000}) {
        run<1, 32>("x2 sc1 unroll 32", ws, stride, 4096, H, 78, 3, 14000, stagger, out, cyc);
        run<3, 32>("x4 sc1 unroll 32", ws, stride, 4096, H, 78, 3, 14000, stagger, out, cyc);
    }
    // how the rate depends on what is resident (0: stream everything, 78: the kernel, 210: 20 tile rows resident)
    for (int res : {0, 78, 210, 300})
        run<1, 32>(res == 0 ? "x2 sc1, nothing resident" : (res == 78 ? "x2 sc1, 78 resident" : (res == 210 ? "x2 sc1, 210 resident" : "x2 sc1, 300 resident")), ws,
                   stride, 4096, H, res, 3, 14000, 0, out, cyc);
    return 0;
}
