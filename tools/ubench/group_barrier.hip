// Micro-benchmark (round 6): what a barrier among the 8 workgroups that own one image costs on MI355X when it goes through a global
// counter with agent-scope release / acquire -- the primitive a persistent per-image pipeline would
// put where the layer stack has chip-wide launch boundaries today.  256 workgroups of 512 threads (one per CU, cooperative launch:
// all resident or the launch fails), groups of 8 = the workgroups with the same (id mod 8) residue class ... or 8 consecutive ids
// (both maps timed: workgroup w lands on XCD w % 8, so "consecutive" groups span all eight XCDs and "strided" groups sit on one).
// Every iteration each workgroup writes a 64 KiB slice, passes the barrier, reads the slice of its neighbour in the group and checks it.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/group_barrier tools/ubench/group_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__device__ __forceinline__ bool group_barrier(unsigned* cnt, unsigned target, unsigned* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        // MODE 0: agent-scope fences (what the memory model asks for when the group may span XCDs); 1: no fences (raw atomic + poll
        // latency, NOT correct); 2: same-XCD coherence by hand -- stores are write-through to the XCD's L2 (wait for them), readers
        // invalidate their CU's vector L1 (buffer_inv sc0); 3: as 2 with buffer_inv sc1
        if (MODE == 0) __threadfence();
        if (MODE >= 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) {                    // (a lost workgroup must not hang the device)
                atomicAdd(err, 1u);
                ok = false;
                break;
            }
        }
        if (MODE == 0) __threadfence();
        if (MODE == 2) asm volatile("buffer_inv sc0" ::: "memory");
        if (MODE == 3) asm volatile("buffer_inv sc1" ::: "memory");
    }
    __syncthreads();
    return ok;
}

template <int MODE>
__global__ __launch_bounds__(512) void k_barrier(unsigned* cnt, float* buf, unsigned* err, long long* ticks, int iters, int strided,
                                                 int slice_floats, int do_io) {
    const int w = blockIdx.x, ngroups = gridDim.x / 8;
    const int group = strided ? (w % 8) * (ngroups / 8) + (w / 8) / 8 : w / 8;      // strided: the 8 members share w % 8 (one XCD)
    const int member = strided ? (w / 8) % 8 : w % 8;
    float* mine = buf + ((long)group * 8 + member) * slice_floats;
    const float* next = buf + ((long)group * 8 + (member + 1) % 8) * slice_floats;
    long long t0 = clock64();
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        if (do_io)
            for (int i = threadIdx.x; i < slice_floats; i += blockDim.x) mine[i] = (float)(it * 8 + member) + i * 1e-3f;
        if (!group_barrier<MODE>(cnt + group, 8u * (2 * it + 1), err)) return;
        if (do_io)
            for (int i = threadIdx.x; i < slice_floats; i += blockDim.x) {
                const float v = next[i];
                bad += v != (float)(it * 8 + (member + 1) % 8) + i * 1e-3f;
            }
        if (!group_barrier<MODE>(cnt + group, 8u * (2 * it + 2), err)) return;      // nobody overwrites a slice that is still being read
    }
    if (bad) atomicAdd(err + 1, bad);
    if (threadIdx.x == 0) ticks[w] = clock64() - t0;
}

int main() {
    const int WG = 256, slice = 16384;
    unsigned *cnt, *err;
    float* buf;
    long long* ticks;
    hipMalloc(&cnt, 64 * sizeof(unsigned));
    hipMalloc(&err, 2 * sizeof(unsigned));
    hipMalloc(&buf, (size_t)WG * slice * sizeof(float));
    hipMalloc(&ticks, WG * sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int dev = 0, coop = 0;
    hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
    printf("cooperative launch supported: %d\n", coop);
    for (int mode : {0, 1, 2, 3}) {
      for (int do_io : {0, 1}) {
        for (int strided : {1, 0}) {
            int iters = 2000;
            hipMemset(cnt, 0, 64 * sizeof(unsigned));
            hipMemset(err, 0, 2 * sizeof(unsigned));
            int sl = slice;
            void* args[] = {&cnt, &buf, &err, &ticks, &iters, &strided, &sl, &do_io};
            const void* fn = mode == 0 ? (const void*)k_barrier<0> : mode == 1 ? (const void*)k_barrier<1> : mode == 2 ? (const void*)k_barrier<2> : (const void*)k_barrier<3>;
            hipEventRecord(e0);
            hipError_t rc = hipLaunchCooperativeKernel(fn, dim3(WG), dim3(512), args, 0, 0);
            hipEventRecord(e1);
            hipError_t rs = hipDeviceSynchronize();
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned herr[2];
            hipMemcpy(herr, err, sizeof(herr), hipMemcpyDeviceToHost);
            printf("mode %d, %s groups, %s: rc %d/%d, %.2f us per barrier%s; timeouts %u, stale reads %u\n", mode,
                   strided ? "same-XCD (strided)" : "cross-XCD (consecutive)", do_io ? "64 KiB out + 64 KiB in per iteration" : "no data",
                   (int)rc, (int)rs, 1e3 * ms / (2.0 * iters), do_io ? " (incl. the I/O)" : "", herr[0], herr[1]);
        }
      }
    }
    return 0;
}
