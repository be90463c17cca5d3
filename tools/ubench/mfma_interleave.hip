// Micro-benchmark (round 2): one wave stream of {1 MFMA, K independent VALU} repeated, 1 or 2 waves per SIMD, and the
// variant where the matrix wave pads with s_nop instead of stalling on the busy pipe while a partner wave runs VALU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/mfma_interleave tools/ubench/mfma_interleave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(x[(i) + 8]))

template <int K>
__global__ __launch_bounds__(512) void k_inter(long long* out, float* sink, int iters) {
    u32x4 a = {threadIdx.x + 1u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, b = {0x3f803f80u, threadIdx.x * 3u, 0x3f803f80u, 0x3f803f80u};
    f32x16 acc0, acc1;
    float x[16];
    const float c = 1.0001f;
    for (int i = 0; i < 16; ++i) acc0[i] = 0.f, acc1[i] = 0.f, x[i] = threadIdx.x * 0.37f + i;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r & 1) MFMA(acc1); else MFMA(acc0);
#pragma unroll
            for (int k = 0; k < K; ++k) FMA((r * K + k) & 7);
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + x[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}

// waves 0-3: {MFMA, s_nop padding to ~32 cycles} so the wave never waits on the busy pipe; waves 4-7: independent fmas
template <int PAD>
__global__ __launch_bounds__(512) void k_padded(long long* out, float* sink, int it_m, int it_v) {
    const int wave = threadIdx.x >> 6;
    u32x4 a = {threadIdx.x + 1u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, b = {0x3f803f80u, threadIdx.x * 3u, 0x3f803f80u, 0x3f803f80u};
    f32x16 acc0, acc1;
    float x[16];
    const float c = 1.0001f;
    for (int i = 0; i < 16; ++i) acc0[i] = 0.f, acc1[i] = 0.f, x[i] = threadIdx.x * 0.37f + i;
    __syncthreads();
    long long t0 = clock64();
    if (wave < 4) {
        for (int it = 0; it < it_m; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (r & 1) MFMA(acc1); else MFMA(acc0);
                if (PAD == 1) { asm volatile("s_nop 7\n s_nop 7\n s_nop 7"); }
                if (PAD == 2) { asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 3"); }
                if (PAD == 3) { asm volatile("s_sleep 1"); }
            }
        }
    } else {
        for (int it = 0; it < it_v; ++it) {
#pragma unroll
            for (int r = 0; r < 11; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) FMA(i);
            }
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + x[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

int main() {
    long long* d_out;
    float* d_sink;
    hipMalloc(&d_out, 64 * sizeof(long long));
    hipMalloc(&d_sink, 4096 * sizeof(float));
    std::vector<long long> h(8);
    const int iters = 2000;
    auto inter = [&](auto kern, int K, int waves) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), 0, 0, d_out, d_sink, iters);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_out, 8 * sizeof(long long), hipMemcpyDeviceToHost);
        double mx = 0;
        for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
        printf("{1 MFMA + %2d VALU} x8, %d wave(s)/SIMD: %7.1f cycles per MFMA slot per wave\n", K, waves / 4, mx / iters / 8);
    };
    for (int waves : {4, 8}) {
        inter(k_inter<0>, 0, waves);
        inter(k_inter<2>, 2, waves);
        inter(k_inter<4>, 4, waves);
        inter(k_inter<6>, 6, waves);
        inter(k_inter<8>, 8, waves);
        inter(k_inter<10>, 10, waves);
        inter(k_inter<12>, 12, waves);
        inter(k_inter<16>, 16, waves);
        inter(k_inter<24>, 24, waves);
    }
    auto padded = [&](auto kern, const char* what, int im, int iv) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(512), 0, 0, d_out, d_sink, im, iv);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_out, 8 * sizeof(long long), hipMemcpyDeviceToHost);
        double tm = 0, tv = 0;
        for (int w = 0; w < 4; ++w) tm = h[w] > tm ? h[w] : tm, tv = h[w + 4] > tv ? h[w + 4] : tv;
        printf("%-28s iters %5d/%5d: matrix waves %7.1f cycles per MFMA | vector waves %7.2f cycles per VALU\n", what, im, iv,
               im ? tm / im / 8 : 0.0, iv ? tv / iv / 88 : 0.0);
    };
    padded(k_padded<0>, "no padding", 3000, 0);
    padded(k_padded<0>, "no padding", 0, 3000);
    padded(k_padded<0>, "no padding", 12000, 1000);
    padded(k_padded<1>, "s_nop x24", 3000, 0);
    padded(k_padded<1>, "s_nop x24", 12000, 1000);
    padded(k_padded<2>, "s_nop x28", 3000, 0);
    padded(k_padded<2>, "s_nop x28", 12000, 1000);
    padded(k_padded<3>, "s_sleep 1", 3000, 0);
    padded(k_padded<3>, "s_sleep 1", 12000, 1000);
    return 0;
}
