// Micro-benchmark (round 2): how much vector-ALU work issues beside a busy matrix pipe on one SIMD of an MI355X, per
// register class of the accumulator (arch VGPR / AccVGPR) and with s_setprio.  One workgroup of 8 waves: wave w and w + 4
// share SIMD w % 4.  Waves 0-3 run MFMAs, waves 4-7 the exact bf16 split (ffno_device.h) -- each class is timed twice:
// once running LONGER than the other class (its time then contains a stretch alone) and once SHORTER (fully beside it).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/mfma_valu_overlap tools/ubench/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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

// MODE 0: accumulators in arch VGPRs (compiler's choice in ffx.hip), 1: in AccVGPRs, 2: arch VGPRs + s_setprio 3 on the
// matrix waves, 3: AccVGPRs + s_setprio 3, 4: arch VGPRs, the VECTOR waves at s_setprio 3
template <int MODE>
__global__ __launch_bounds__(512) void k_overlap(long long* out, float* sink, int it_mfma, int it_valu, int valu_kind) {
    const int wave = threadIdx.x >> 6;
    long long t0, t1;
    float s = 0.f;
    if (wave < 4) {
        u32x4 a = {threadIdx.x + 1u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, b = {0x3f803f80u, threadIdx.x * 3u, 0x3f803f80u, 0x3f803f80u};
        f32x16 acc0, acc1;
        for (int i = 0; i < 16; ++i) acc0[i] = 0.f, acc1[i] = 0.f;
        if (MODE == 2 || MODE == 3) __builtin_amdgcn_s_setprio(3);
        __syncthreads();
        t0 = clock64();
        for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                if (MODE == 1 || MODE == 3) {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
                } else {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
                }
            }
        }
        t1 = clock64();
        for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    } else {
        float x[16];
        for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.37f + i;
        unsigned acc = 0;
        if (MODE == 4) __builtin_amdgcn_s_setprio(3);
        __syncthreads();
        t0 = clock64();
        if (valu_kind == 0) {
            for (int it = 0; it < it_valu; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    unsigned h, m, l;
                    split_pair(x[2 * i], x[2 * i + 1], h, m, l);
                    acc ^= h + m + l;
                    x[2 * i] += 1.f;
                }
            }
        } else {
            // 8 independent fma chains: no dependent-issue stalls, pure issue rate
            for (int it = 0; it < it_valu; ++it) {
#pragma unroll
                for (int r = 0; r < 11; ++r) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, x[i + 8]);
                }
            }
        }
        t1 = clock64();
        for (int i = 0; i < 16; ++i) s += x[i];
        s += (float)acc;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

int main() {
    long long* d_out;
    float* d_sink;
    hipMalloc(&d_out, 64 * sizeof(long long));
    hipMalloc(&d_sink, 4096 * sizeof(float));
    std::vector<long long> h(8);
    // tick calibration: 24 MFMAs back to back = 768 cycles
    auto run = [&](auto kern, const char* what, int im, int iv, int kind) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(512), 0, 0, d_out, d_sink, im, iv, kind);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_out, 8 * sizeof(long long), hipMemcpyDeviceToHost);
        double tm = 0, tv = 0;
        for (int w = 0; w < 4; ++w) tm = h[w] > tm ? h[w] : tm, tv = h[w + 4] > tv ? h[w + 4] : tv;
        printf("%-34s kind %d  iters %5d/%5d: matrix waves %8.1f ticks per 24 MFMA | vector waves %8.1f ticks per 88 VALU\n", what, kind,
               im, iv, im ? tm / im : 0.0, iv ? tv / iv : 0.0);
    };
    for (int kind : {0, 1}) {
        run(k_overlap<0>, "alone", 2000, 0, kind);
        run(k_overlap<0>, "alone", 0, 2000, kind);
        // matrix waves fully covered by vector work (vector waves run ~3x longer), then the converse
        run(k_overlap<0>, "acc in VGPR", 1000, 8000, kind);
        run(k_overlap<0>, "acc in VGPR", 4000, 1000, kind);
        run(k_overlap<1>, "acc in AccVGPR", 1000, 8000, kind);
        run(k_overlap<1>, "acc in AccVGPR", 4000, 1000, kind);
        run(k_overlap<2>, "VGPR, matrix waves prio 3", 1000, 8000, kind);
        run(k_overlap<2>, "VGPR, matrix waves prio 3", 4000, 1000, kind);
        run(k_overlap<3>, "AccVGPR, matrix waves prio 3", 1000, 8000, kind);
        run(k_overlap<3>, "AccVGPR, matrix waves prio 3", 4000, 1000, kind);
        run(k_overlap<4>, "VGPR, vector waves prio 3", 1000, 8000, kind);
        run(k_overlap<4>, "VGPR, vector waves prio 3", 4000, 1000, kind);
    }
    return 0;
}
