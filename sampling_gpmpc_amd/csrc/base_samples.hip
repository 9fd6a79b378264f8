// gpmpc_base_samples: the counter-based base samples of the Agent (reference src/agent.py:76-104, the epistemic random vector:
// i.i.d. N(0, 1) vectors of V = g_ny H T entries, the WHOLE vector redrawn until every entry lies in [-beta, beta]), one launch for
// all (MPC step, SQP iteration, sample) vectors of a rank.
//
// The stream (sampling_gpmpc_amd.agent.counter_base_samples states it in torch ops; tests/ keeps that form as the checker): vector
// (j, i, s) has the key mix64(s * C1 + seed * C2 + (j n_itrs + i) * C3), s the GLOBAL sample id; attempt a draws entry e from the
// hashed counters c = key + (2 e + 2 a V) * C4 and c + C4: u1, u2 = (top 53 bits + 0.5) 2^-53, w = sqrt(-2 log u1) cos(2 pi u2)
// (Box-Muller); the first attempt with max |w| <= beta is kept.  Every vector is a pure function of (seed, j, i, s): a rank of a
// sample-sharded run generates exactly its own shard and the assembled run is the same for every GPU count.
//
// Mapping: one wave per vector, lanes stride the V entries, the attempts of a vector loop inside the wave (the torch form was ~15
// launches per attempt and (j, i) slab - 60 k launches per bench process, all Agent construction); the accepted attempt is
// recomputed into the output (an attempt is rejected with probability 1 - P(|w| <= beta)^V: ~70 % at V = 90, beta = 2.5 - storing
// every attempt would write three times as much).  Entry arithmetic is contraction-free and uses the device library's log / cos /
// sqrt - operation for operation what the torch form evaluates on the same device: the two agree bit for bit (GPU test).
#include "gpmpc_host.hpp"

namespace gpmpc {

__device__ __forceinline__ unsigned long long bs_mix64(unsigned long long x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

#pragma clang fp contract(off)
__device__ __forceinline__ double bs_entry(unsigned long long key, long e, long a, long V) {
    const unsigned long long C4 = 0xDA942042E4DD58B5ull;
    const unsigned long long c = key + (unsigned long long)(2 * e + 2 * a * V) * C4;
    const double u1 = ((double)(long long)(bs_mix64(c) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(long long)(bs_mix64(c + C4) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double two_pi = 2.0 * 3.141592653589793;
    return sqrt(-2.0 * log(u1)) * cos(two_pi * u2);
}

__global__ __launch_bounds__(256) void base_samples_kernel(unsigned long long seed, int n_mpc, int n_itrs, long offset, long Ns, int V, double beta,
                                                           double* __restrict__ out, int* __restrict__ attempts) {
    const int lane = threadIdx.x & 63;
    const long nvec = (long)n_mpc * n_itrs * Ns;
    for (long vec = (long)blockIdx.x * 4 + (threadIdx.x >> 6); vec < nvec; vec += (long)gridDim.x * 4) {
        const long ji = vec / Ns, s = vec - ji * Ns;
        const unsigned long long key = bs_mix64((unsigned long long)(offset + s) * 0x9E3779B97F4A7C15ull +
                                                (seed * 0xD1B54A32D192ED03ull + (unsigned long long)ji * 0x8CB92BA72F3D8DD7ull));
        long a = 0;
        for (;; ++a) {
            bool ok = true;
            for (int e = lane; e < V; e += 64) ok = ok && (fabs(bs_entry(key, e, a, V)) <= beta);
            if (__all(ok)) break;
        }
        double* o = out + vec * V;
        for (int e = lane; e < V; e += 64) o[e] = bs_entry(key, e, a, V);
        if (attempts && lane == 0) attempts[vec] = (int)a;
    }
}

}  // namespace gpmpc

using namespace gpmpc;

extern "C" int gpmpc_base_samples(uint64_t seed, int32_t n_mpc, int32_t n_itrs, int64_t offset, int64_t Ns, int32_t V, double beta,
                                  double* out, int32_t* attempts, void* stream) {
    if (!out || n_mpc < 1 || n_itrs < 1 || Ns < 1 || V < 1 || offset < 0 || !(beta > 0.0))
        return fail(GPMPC_E_ARG, "gpmpc_base_samples: bad arguments");
    // P(|w| <= beta)^V is the acceptance probability of an attempt: refuse parameter sets that would never finish
    const double p1 = erf(beta / 1.4142135623730951);
    if (pow(p1, (double)V) < 1e-6) return fail(GPMPC_E_ARG, "gpmpc_base_samples: beta too small for vectors of this length (acceptance < 1e-6)");
    const long nvec = (long)n_mpc * n_itrs * Ns;
    long g = (nvec + 3) / 4;
    if (g > 256L * 32) g = 256L * 32;
    hipLaunchKernelGGL(base_samples_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (unsigned long long)seed, n_mpc, n_itrs,
                       (long)offset, (long)Ns, V, beta, out, (int*)attempts);
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}
