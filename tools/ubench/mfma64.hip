// Micro-benchmarks behind the design of csrc/rollout_rows.hip (gfx950): the FP64 4x4x4 MFMA (4 blocks = the four DPP rows of
// a wave) as the engine of a 16-lanes-per-chain triangular solve.
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/mfma64.hip -o tools/ubench/mfma64.bin && tools/ubench/mfma64.bin
// Reports: the lane maps of A / B / D; cycles per instruction (s_memtime, one wave alone on its SIMD) of independent and
// dependent MFMAs, of MFMAs with the A operand in an AGPR, of v_fma_f64 / v_fmac_f64_dpp / v_accvgpr_read / v_mov_b64_dpp /
// ds_bpermute, and whether FP64 VALU work issued between MFMAs overlaps with them.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));                    \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

__global__ void layout_kernel(const double* A, const double* B, double* D, double* D2, double* bc) {
    const int l = threadIdx.x;
    double acc = 0.0;
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], acc, 0, 0, 0);
    D[l] = acc;
    // D = C - A B through the neg modifier on A
    double c = 1000.0 + l, d;
    const double a = A[l], b = B[l];
    asm volatile("s_nop 4\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %3 neg:[1,0,0]\n\ts_nop 7\n\ts_nop 7" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    D2[l] = d;
    // v_mov_b64_dpp row_newbcast
    double m = 0.0;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "=v"(m) : "v"(a));
    bc[l] = m;
}

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int MODE>
__global__ void timing_kernel(double* out, long long* cyc, int iters) {
    const int l = threadIdx.x;
    double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
    double c0 = 0.0, c1 = 0.1, c2 = 0.2, c3 = 0.3, c4 = 0.4, c5 = 0.5, c6 = 0.6, c7 = 0.7;
    double f0 = 0.0, f1 = 0.1, f2 = 0.2, f3 = 0.3, f4 = 0.4, f5 = 0.5;
    int p0 = l, p1 = l + 1;
    const int addr = ((l * 5) & 63) << 2;
    // AGPR operand for MODE 5
    asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\ts_nop 4" ::"v"(__double2loint(a)), "v"(__double2hiint(a)) : "a0", "a1");
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {          // 16 independent-ish MFMAs (8 accumulators round robin)
            asm volatile(REP4("v_mfma_f64_4x4x4_4b_f64 %0, %8, %9, %0\n\tv_mfma_f64_4x4x4_4b_f64 %1, %8, %9, %1\n\t"
                              "v_mfma_f64_4x4x4_4b_f64 %2, %8, %9, %2\n\tv_mfma_f64_4x4x4_4b_f64 %3, %8, %9, %3\n\t")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
        } else if constexpr (MODE == 1) {   // 16 dependent MFMAs (one accumulator)
            asm volatile(REP16("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\t") : "+v"(c0) : "v"(a), "v"(b));
        } else if constexpr (MODE == 2) {   // 16 independent v_fma_f64 (4 accumulators)
            asm volatile(REP4("v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3\n\t")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a), "v"(b));
        } else if constexpr (MODE == 3) {   // 16 x (1 MFMA + 3 independent v_fma_f64), 4 MFMA accumulators
            asm volatile(REP4("v_mfma_f64_4x4x4_4b_f64 %0, %7, %8, %0\n\tv_fma_f64 %4, %7, %8, %4\n\tv_fma_f64 %5, %7, %8, %5\n\tv_fma_f64 %6, %7, %8, %6\n\t"
                              "v_mfma_f64_4x4x4_4b_f64 %1, %7, %8, %1\n\tv_fma_f64 %4, %7, %8, %4\n\tv_fma_f64 %5, %7, %8, %5\n\tv_fma_f64 %6, %7, %8, %6\n\t"
                              "v_mfma_f64_4x4x4_4b_f64 %2, %7, %8, %2\n\tv_fma_f64 %4, %7, %8, %4\n\tv_fma_f64 %5, %7, %8, %5\n\tv_fma_f64 %6, %7, %8, %6\n\t"
                              "v_mfma_f64_4x4x4_4b_f64 %3, %7, %8, %3\n\tv_fma_f64 %4, %7, %8, %4\n\tv_fma_f64 %5, %7, %8, %5\n\tv_fma_f64 %6, %7, %8, %6\n\t")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(f0), "+v"(f1), "+v"(f2) : "v"(a), "v"(b));
        } else if constexpr (MODE == 4) {   // 16 v_fmac_f64_dpp row_newbcast (4 accumulators)
            asm volatile(REP4("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a), "v"(b));
        } else if constexpr (MODE == 5) {   // 16 MFMAs with the A operand in an AGPR (4 accumulators)
            asm volatile(REP4("v_mfma_f64_4x4x4_4b_f64 %0, a[0:1], %4, %0\n\tv_mfma_f64_4x4x4_4b_f64 %1, a[0:1], %4, %1\n\t"
                              "v_mfma_f64_4x4x4_4b_f64 %2, a[0:1], %4, %2\n\tv_mfma_f64_4x4x4_4b_f64 %3, a[0:1], %4, %3\n\t")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b));
        } else if constexpr (MODE == 6) {   // 16 v_accvgpr_read_b32
            asm volatile(REP4("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %0, a1\n\tv_accvgpr_read_b32 %1, a0\n\t") : "+v"(p0), "+v"(p1));
        } else if constexpr (MODE == 7) {   // 16 v_mov_b64_dpp row_newbcast
            asm volatile(REP4("v_mov_b64_dpp %0, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mov_b64_dpp %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a));
        } else if constexpr (MODE == 8) {   // 16 ds_bpermute_b32 (2 chains)
            asm volatile(REP4("ds_bpermute_b32 %0, %2, %0\n\tds_bpermute_b32 %1, %2, %1\n\tds_bpermute_b32 %0, %2, %0\n\tds_bpermute_b32 %1, %2, %1\n\t") "s_waitcnt lgkmcnt(0)"
                         : "+v"(p0), "+v"(p1) : "v"(addr));
        } else if constexpr (MODE == 9) {   // 16 x (1 MFMA + 3 v_fmac_f64_dpp)
            asm volatile(REP4("v_mfma_f64_4x4x4_4b_f64 %0, %7, %8, %0\n\tv_fmac_f64_dpp %4, %7, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %7, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mfma_f64_4x4x4_4b_f64 %1, %7, %8, %1\n\tv_fmac_f64_dpp %4, %7, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %7, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mfma_f64_4x4x4_4b_f64 %2, %7, %8, %2\n\tv_fmac_f64_dpp %4, %7, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %7, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mfma_f64_4x4x4_4b_f64 %3, %7, %8, %3\n\tv_fmac_f64_dpp %4, %7, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %7, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(f0), "+v"(f1), "+v"(f2) : "v"(a), "v"(b));
        } else if constexpr (MODE == 10) {  // 16 x (1 MFMA + 2 ds_read_b128), waits every 16
            asm volatile(REP4("v_mfma_f64_4x4x4_4b_f64 %0, %8, %9, %0\n\tds_read_b128 %4, %10\n\tds_read_b128 %5, %10 offset:1024\n\t"
                              "v_mfma_f64_4x4x4_4b_f64 %1, %8, %9, %1\n\tds_read_b128 %6, %10 offset:2048\n\tds_read_b128 %7, %10 offset:3072\n\t"
                              "v_mfma_f64_4x4x4_4b_f64 %2, %8, %9, %2\n\tds_read_b128 %4, %10 offset:4096\n\tds_read_b128 %5, %10 offset:5120\n\t"
                              "v_mfma_f64_4x4x4_4b_f64 %3, %8, %9, %3\n\tds_read_b128 %6, %10 offset:6144\n\tds_read_b128 %7, %10 offset:7168\n\t") "s_waitcnt lgkmcnt(0)"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(*(double2*)&f0), "=&v"(*(double2*)&f2), "=&v"(*(double2*)&f4), "=&v"(*(double2*)&c6)
                         : "v"(a), "v"(b), "v"(l * 16));
        } else if constexpr (MODE == 11) {  // 2-deep dependent chains: 16 MFMAs alternating between two accumulators
            asm volatile(REP4("v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, %0\n\tv_mfma_f64_4x4x4_4b_f64 %1, %2, %3, %1\n\t"
                              "v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, %0\n\tv_mfma_f64_4x4x4_4b_f64 %1, %2, %3, %1\n\t")
                         : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
        } else if constexpr (MODE == 12) {  // 16 v_fma_f64 in ONE dependent chain
            asm volatile(REP16("v_fma_f64 %0, %1, %2, %0\n\t") : "+v"(f0) : "v"(a), "v"(b));
        } else if constexpr (MODE == 13) {  // 16 dependent v_fmac_f64_dpp reading their own accumulator through DPP (with the hazard nop)
            asm volatile(REP16("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t") : "+v"(f0) : "v"(b));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 64 + l] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + f0 + f1 + f2 + f3 + f4 + f5 + p0 + p1;
}

template <int MODE>
static void run(const char* name, double* out, long long* cyc, int blocks) {
    const int iters = 2000;
    hipLaunchKernelGGL(timing_kernel<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 100);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(timing_kernel<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long h = 0;
    CHECK(hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    std::printf("%-58s blocks %5d: %7.2f cycles per instruction group (s_memtime), %8.3f ms\n", name, blocks, (double)h / (iters * 16.0), ms);
}

int main() {
    double *dA, *dB, *dD, *dD2, *dbc;
    std::vector<double> A(64), B(64), D(64), D2(64), bc(64);
    for (int l = 0; l < 64; ++l) {
        A[l] = 1.0 + l;
        B[l] = 100.0 + 3.0 * l + (l % 5) * 0.25;
    }
    CHECK(hipMalloc(&dA, 512));
    CHECK(hipMalloc(&dB, 512));
    CHECK(hipMalloc(&dD, 512));
    CHECK(hipMalloc(&dD2, 512));
    CHECK(hipMalloc(&dbc, 512));
    CHECK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, dD2, dbc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(D2.data(), dD2, 512, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(bc.data(), dbc, 512, hipMemcpyDeviceToHost));
    // candidate lane maps inside a block of 16 lanes: element (x, y) at lane 4x + y ("xy") or 4y + x ("yx")
    auto at = [](int mode, int x, int y) { return mode ? 4 * y + x : 4 * x + y; };
    for (int ma = 0; ma < 2; ++ma)
        for (int mb = 0; mb < 2; ++mb)
            for (int md = 0; md < 2; ++md) {
                bool ok = true;
                for (int blk = 0; blk < 4 && ok; ++blk)
                    for (int i = 0; i < 4 && ok; ++i)
                        for (int j = 0; j < 4 && ok; ++j) {
                            double s = 0;
                            for (int k = 0; k < 4; ++k) s += A[16 * blk + at(ma, i, k)] * B[16 * blk + at(mb, k, j)];
                            if (s != D[16 * blk + at(md, i, j)]) ok = false;
                        }
                if (ok)
                    std::printf("layout: block = lane/16; A[i][k] at lane %s, B[k][j] at lane %s, D[i][j] at lane %s\n",
                                ma ? "4k+i" : "4i+k", mb ? "4j+k" : "4k+j", md ? "4j+i" : "4i+j");
            }
    {   // the neg modifier: D2 = C - A B ?
        bool ok = true;
        for (int l = 0; l < 64; ++l) ok = ok && (D2[l] == 1000.0 + l - D[l]);
        std::printf("neg:[1,0,0] gives C - A B: %s\n", ok ? "yes" : "NO");
        bool okb = true;
        for (int l = 0; l < 64; ++l) okb = okb && (bc[l] == A[(l & ~15) + 5]);
        std::printf("v_mov_b64_dpp row_newbcast:5 broadcasts lane 5 of each row: %s (lane 20 got %g)\n", okb ? "yes" : "NO", bc[20]);
    }
    double* out;
    long long* cyc;
    CHECK(hipMalloc(&out, 4096 * 64 * 8));
    CHECK(hipMalloc(&cyc, 64));
    for (int blocks : {1, 1024, 2048}) {
        run<0>("MFMA f64 4x4x4_4b, 8 accumulators", out, cyc, blocks);
        run<1>("MFMA f64 4x4x4_4b, dependent chain", out, cyc, blocks);
        run<11>("MFMA f64 4x4x4_4b, two alternating accumulators", out, cyc, blocks);
        run<5>("MFMA f64 4x4x4_4b, A operand in AGPR", out, cyc, blocks);
        run<2>("v_fma_f64, 4 accumulators", out, cyc, blocks);
        run<12>("v_fma_f64, dependent chain", out, cyc, blocks);
        run<4>("v_fmac_f64_dpp row_newbcast, 4 accumulators", out, cyc, blocks);
        run<13>("s_nop 1 + v_fmac_f64_dpp on its own accumulator (chain)", out, cyc, blocks);
        run<3>("1 MFMA + 3 v_fma_f64 (group)", out, cyc, blocks);
        run<9>("1 MFMA + 3 v_fmac_f64_dpp (group)", out, cyc, blocks);
        run<10>("1 MFMA + 2 ds_read_b128 (group)", out, cyc, blocks);
        run<6>("v_accvgpr_read_b32", out, cyc, blocks);
        run<7>("v_mov_b64_dpp row_newbcast", out, cyc, blocks);
        run<8>("ds_bpermute_b32", out, cyc, blocks);
    }
    return 0;
}
