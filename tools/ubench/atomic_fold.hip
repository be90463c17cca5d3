// Micro-benchmark (round 6): what the one-atomic-per-workgroup fold of a range word (ffno_device.h range_fold: atomicMax on ONE
// 32-bit word from every workgroup of a launch) costs on MI355X as the number of workgroups grows, and what a test-before-atomic
// (read the word, skip the atomic when the workgroup's maximum does not exceed it) changes.  Each workgroup streams `bytes` of a
// buffer (so the launch has a body), then folds.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/atomic_fold tools/ubench/atomic_fold.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// MODE 0: no fold; 1: atomicMax per workgroup; 2: load + compare, atomic only when larger; 3: atomicMax per WAVE (4 per workgroup)
template <int MODE>
__global__ __launch_bounds__(256) void fold_kernel(const float4* __restrict__ buf, size_t n4_per_wg, unsigned* word, int pattern) {
    __shared__ float red[4];
    float m = 0.f;
    const float4* p = buf + (size_t)blockIdx.x * n4_per_wg;
    for (size_t i = threadIdx.x; i < n4_per_wg; i += 256) {
        const float4 v = p[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    // pattern 0: the data's own maximum (all ~equal); 1: ascending with the workgroup index (every workgroup raises the word)
    if (pattern == 1) m += (float)blockIdx.x;
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if (MODE == 3) {
        if ((threadIdx.x & 63) == 0) atomicMax(word, __float_as_uint(m));
        return;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0 && MODE != 0) {
        for (int w = 1; w < 4; ++w) m = fmaxf(m, red[w]);
        const unsigned u = __float_as_uint(m);
        if (MODE == 1) atomicMax(word, u);
        if (MODE == 2) {
            if (u > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, u);
        }
    }
}

template <int MODE>
static float run(const float4* buf, size_t n4_per_wg, unsigned* word, int wgs, int pattern, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fold_kernel<MODE>, dim3(wgs), dim3(256), 0, 0, buf, n4_per_wg, word, pattern);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) {
        // (the word is NOT reset between launches for pattern 0: like a range word that is zeroed once per step; pattern 1 launches
        //  raise it only the first time -- so reset it for both to see the worst case)
        hipMemsetAsync(word, 0, 4);
        hipLaunchKernelGGL(fold_kernel<MODE>, dim3(wgs), dim3(256), 0, 0, buf, n4_per_wg, word, pattern);
    }
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return 1e3f * ms / reps;
}

int main() {
    const size_t total = (size_t)64 << 20;      // 64 MiB
    float4* buf;
    unsigned* word;
    hipMalloc(&buf, total);
    hipMalloc(&word, 256);
    std::vector<float> h(total / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * 1e-4f;
    hipMemcpy(buf, h.data(), total, hipMemcpyHostToDevice);
    // the memset alone
    {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 200; ++i) hipMemsetAsync(word, 0, 4);
        hipEventRecord(b);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("hipMemsetAsync(4 bytes) alone: %.2f us\n", 1e3f * ms / 200);
    }
    printf("us per (memset + launch); body = each workgroup streams its share of `MiB`\n");
    printf("%6s %6s %8s | %9s %9s %9s %9s\n", "wgs", "MiB", "pattern", "no fold", "atomic", "test+atom", "per-wave");
    const int wgl[] = {64, 256, 1024, 2048, 8192};
    const size_t mibs[] = {1, 32};
    for (size_t mib : mibs)
        for (int wgs : wgl)
            for (int pattern = 0; pattern < 2; ++pattern) {
                const size_t n4 = (mib << 20) / 16 / wgs;
                const float t0 = run<0>(buf, n4, word, wgs, pattern, 200);
                const float t1 = run<1>(buf, n4, word, wgs, pattern, 200);
                const float t2 = run<2>(buf, n4, word, wgs, pattern, 200);
                const float t3 = run<3>(buf, n4, word, wgs, pattern, 200);
                printf("%6d %6zu %8s | %9.2f %9.2f %9.2f %9.2f\n", wgs, mib, pattern ? "ascend" : "equal", t0, t1, t2, t3);
            }
    return 0;
}
