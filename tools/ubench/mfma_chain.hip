// Micro-benchmark (round 2): issue / dependent-accumulator cost of v_mfma_f32_32x32x16_bf16 and of the exact three-way bf16
// split on one SIMD of an MI355X.  One workgroup; waves_per_simd waves share every SIMD.  Prints cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain tools/ubench/mfma_chain.hip && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mm(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int CHAINS>
__global__ __launch_bounds__(512) void k_mfma(long long* out, float* sink, int iters) {
    u32x4 a = {threadIdx.x + 1u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, b = {0x3f803f80u, threadIdx.x * 3u, 0x3f803f80u, 0x3f803f80u};
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 24 / CHAINS; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = mm(a, b, acc[c]);
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}

// the split of ffno_device.h: 8 floats -> three packed planes
__device__ __forceinline__ unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float u2f(unsigned x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = f2u(x0), u1 = f2u(x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - u2f(u0 & 0xffff0000u), r1 = x1 - u2f(u1 & 0xffff0000u);
    const unsigned v0 = f2u(r0), v1 = f2u(r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float q0 = r0 - u2f(v0 & 0xffff0000u), q1 = r1 - u2f(v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(f2u(q1), f2u(q0), 0x07060302u);
}

// VALU only: 8 split pairs (= two split3_8) per iteration
__global__ __launch_bounds__(512) void k_split(long long* out, float* sink, int iters) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.37f + i;
    unsigned acc = 0;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned h, m, l;
            split_pair(x[2 * i], x[2 * i + 1], h, m, l);
            acc ^= h + m + l;
            x[2 * i] += 1.f;
        }
    }
    long long t1 = clock64();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}

// mixed: per iteration 24 MFMAs (CHAINS accumulators) on waves [0, nm), 16 split pairs on the others
template <int CHAINS>
__global__ __launch_bounds__(512) void k_mixed(long long* out, float* sink, int iters, int nm) {
    const int wave = threadIdx.x >> 6;
    long long t0, t1;
    float s = 0.f;
    if (wave < nm) {
        u32x4 a = {threadIdx.x + 1u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, b = {0x3f803f80u, threadIdx.x * 3u, 0x3f803f80u, 0x3f803f80u};
        f32x16 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c)
            for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
        __syncthreads();
        t0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 24 / CHAINS; ++r) {
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) acc[c] = mm(a, b, acc[c]);
            }
        }
        t1 = clock64();
        for (int c = 0; c < CHAINS; ++c)
            for (int i = 0; i < 16; ++i) s += acc[c][i];
    } else {
        float x[16];
        for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.37f + i;
        unsigned acc = 0;
        __syncthreads();
        t0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsigned h, m, l;
                split_pair(x[2 * i], x[2 * i + 1], h, m, l);
                acc ^= h + m + l;
                x[2 * i] += 1.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsigned h, m, l;
                split_pair(x[2 * i], x[2 * i + 1], h, m, l);
                acc ^= h + m + l;
                x[2 * i + 1] += 1.f;
            }
        }
        t1 = clock64();
        s = (float)acc;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

int main() {
    long long* d_out;
    float* d_sink;
    hipMalloc(&d_out, 64 * sizeof(long long));
    hipMalloc(&d_sink, 4096 * sizeof(float));
    const int iters = 2000;
    std::vector<long long> h(8);
    auto report = [&](const char* what, int waves, double per_iter_units, const char* unit) {
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_out, 8 * sizeof(long long), hipMemcpyDeviceToHost);
        double mx = 0;
        for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
        // clock64 = s_memtime: counts at a fixed 100 MHz on gfx9 (wall clock), so report ns and cycles at 2.4 GHz
        printf("%-58s waves/WG %d: %9.1f ticks/iter  -> %7.2f ticks per %s\n", what, waves, mx / iters, mx / iters / per_iter_units, unit);
    };
    for (int waves : {4, 8}) {
        hipLaunchKernelGGL(k_mfma<1>, dim3(1), dim3(64 * waves), 0, 0, d_out, d_sink, iters);
        report("24 MFMA, 1 dependent chain", waves, 24, "MFMA");
        hipLaunchKernelGGL(k_mfma<2>, dim3(1), dim3(64 * waves), 0, 0, d_out, d_sink, iters);
        report("24 MFMA, 2 interleaved chains", waves, 24, "MFMA");
        hipLaunchKernelGGL(k_mfma<4>, dim3(1), dim3(64 * waves), 0, 0, d_out, d_sink, iters);
        report("24 MFMA, 4 interleaved chains", waves, 24, "MFMA");
        hipLaunchKernelGGL(k_split, dim3(1), dim3(64 * waves), 0, 0, d_out, d_sink, iters);
        report("8 split pairs (88 VALU)", waves, 8, "pair");
    }
    hipLaunchKernelGGL(k_mixed<1>, dim3(1), dim3(512), 0, 0, d_out, d_sink, iters, 4);
    report("mixed: waves 0-3 24 MFMA (1 chain) | waves 4-7 16 pairs", 8, 1, "iter");
    hipLaunchKernelGGL(k_mixed<2>, dim3(1), dim3(512), 0, 0, d_out, d_sink, iters, 4);
    report("mixed: waves 0-3 24 MFMA (2 chains) | waves 4-7 16 pairs", 8, 1, "iter");
    // wall-clock calibration of the tick: a known-length s_sleep loop is not needed -- time the whole launch instead
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<4>, dim3(1), dim3(256), 0, 0, d_out, d_sink, 20 * iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_out, 8 * sizeof(long long), hipMemcpyDeviceToHost);
    printf("calibration: %lld ticks = %.3f ms  -> %.1f ns per tick; 24 MFMA x %d iters: %.1f ns per MFMA\n", h[0], ms, 1e6 * ms / h[0],
           20 * iters, 1e6 * ms / (24.0 * 20 * iters));
    return 0;
}
