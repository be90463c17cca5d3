// Micro-benchmark (round 2): shader clock an MI355X sustains while every SIMD of the chip issues MFMAs back to back
// (s_memtime ticks are shader cycles: 24 dependent v_mfma_f32_32x32x16_bf16 = 768 ticks), against the wall clock of HIP events.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/sustained_clock tools/ubench/sustained_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k_mfma(long long* out, float* sink, int iters) {
    u32x4 a = {threadIdx.x + 1u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, b = {0x3f803f80u, threadIdx.x * 3u, 0x3f803f80u, 0x3f803f80u};
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) acc0[i] = 0.f, acc1[i] = 0.f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc1, 0, 0, 0);
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    sink[(blockIdx.x * blockDim.x + threadIdx.x) & 4095] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
    long long* d_out;
    float* d_sink;
    hipMalloc(&d_out, 1024 * sizeof(long long));
    hipMalloc(&d_sink, 4096 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<long long> h(1024);
    for (int wgs : {1, 64, 256}) {
        for (int waves : {4, 8}) {
            const int iters = 40000;   // x 24 MFMAs x 32 cycles = 30.7 M cycles per wave (x 2 with two waves per SIMD)
            hipLaunchKernelGGL(k_mfma, dim3(wgs), dim3(64 * waves), 0, 0, d_out, d_sink, 1000);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma, dim3(wgs), dim3(64 * waves), 0, 0, d_out, d_sink, iters);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), d_out, wgs * sizeof(long long), hipMemcpyDeviceToHost);
            double mx = 0;
            for (int w = 0; w < wgs; ++w) mx = h[w] > mx ? h[w] : mx;
            const double flops = 2.0 * 32 * 32 * 16 * 24.0 * iters * waves * wgs;
            printf("%3d workgroups x %d waves: %.0f ticks in %.3f ms -> %.2f GHz; %.1f ticks per MFMA per wave; %.0f TFLOP/s bf16\n", wgs, waves,
                   mx, ms, mx / (ms * 1e6), mx / (24.0 * iters), flops / (ms * 1e9));
        }
    }
    return 0;
}
