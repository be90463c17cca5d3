// Micro-benchmark (round 4): what the memory side of the paired spectral launch costs by ACCESS WIDTH and pattern, without any
// of its arithmetic.  One launch = 256 workgroups x 8 waves (one per CU, like spectral_x3_pair at batch 32); a wave owns two
// lines of 64 samples x 64 channels (fp32); even workgroups take contiguous lines (last-axis branch: 16 KiB per line), odd ones
// strided lines (first-axis branch: 256-byte samples at a 16 KiB stride).  Per line the wave requests the whole line
// (VEC = 2: 32 x 8-byte loads per lane -- a half-wave covers one 256-byte sample, what the kernel does today;
//  VEC = 4: 16 x 16-byte loads per lane -- 16 lanes cover a sample, four samples per instruction), reduces it trivially and
// writes `nout` output images with the same pattern plus, optionally, the 8 KiB per line of saved spectrum in mode-major
// layout (256-byte rows at a stride of R x 512 bytes, always 8-byte stores as today).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/stream_patterns tools/ubench/stream_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct Args {
    const float* in;
    float* out0;
    float* out1;
    float* spec;
    int nout, save, B, M, N, R, lines_per_wg;   // M = N = L = 64; R = B * 64 lines per axis
    int mode;                                    // 0: read + write; 1: read only; 2: write only
};

template <int VEC>
__global__ __launch_bounds__(512) void k_stream(Args A) {
    constexpr int C = 64, L = 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // image-local map of the real launch: workgroup w runs on XCD w % 8; the 8 workgroups of an image (4 tiles per branch) share
    // an XCD, so the image crosses HBM once and the second branch finds it in that XCD's L2
    const int wg = blockIdx.x;
    const int xcd = wg & 7, slot = wg >> 3;
    const int image = xcd * (A.B / 8) + (slot >> 3), sub = slot & 7;
    const int branch = sub & 1, tile = image * 4 + (sub >> 1);
    const int NLW = A.lines_per_wg / 8;
    float sink = 0.f;
    for (int ln = 0; ln < NLW; ++ln) {
        const int line = tile * A.lines_per_wg + wave * NLW + ln;
        if (line >= A.R) continue;
        long base, es;
        if (branch == 0) {
            base = (long)line * L * C, es = C;              // (b, m): contiguous
        } else {
            const int b = line / A.N, n = line % A.N;       // (b, n): strided by N * C
            base = (long)b * A.M * A.N * C + (long)n * C, es = (long)A.N * C;
        }
        if (VEC == 2) {
            const int j = lane & 31, half = lane >> 5;
            f2 v[32];
            if (A.mode != 2) {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int s = 16 * (i >> 3) + 8 * half + (i & 7);
                    v[i] = *reinterpret_cast<const f2*>(A.in + base + s * es + 2 * j);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i) v[i] = f2{(float)lane, (float)i};
            }
            if (A.mode == 1) {
#pragma unroll
                for (int i = 0; i < 32; ++i) sink += v[i].x + v[i].y;
                continue;
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int s = 16 * (i >> 3) + 8 * half + (i & 7);
                *reinterpret_cast<f2*>(A.out0 + base + s * es + 2 * j) = v[i];
                if (A.nout > 1) *reinterpret_cast<f2*>(A.out1 + base + s * es + 2 * j) = v[i] * 2.f;
            }
            if (A.save) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {      // 32 rows (k, re/im) of 64 channels per line: half-wave per row
                    const int row = 2 * r + half;
                    *reinterpret_cast<f2*>(A.spec + ((long)row * A.R * 2 + (long)(branch * A.R + line)) * C + 2 * j) = v[r];
                }
            }
        } else {
            const int q = lane & 15, g = lane >> 4;
            f4 v[16];
            if (A.mode != 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int s = 4 * i + g;
                    v[i] = *reinterpret_cast<const f4*>(A.in + base + s * es + 4 * q);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = f4{(float)lane, (float)i, 0.f, 1.f};
            }
            if (A.mode == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sink += v[i].x + v[i].y + v[i].z + v[i].w;
                continue;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int s = 4 * i + g;
                *reinterpret_cast<f4*>(A.out0 + base + s * es + 4 * q) = v[i];
                if (A.nout > 1) *reinterpret_cast<f4*>(A.out1 + base + s * es + 4 * q) = v[i] * 2.f;
            }
            if (A.save) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {       // 16-byte stores: four rows per instruction
                    const int row = 4 * r + g;
                    *reinterpret_cast<f4*>(A.spec + ((long)row * A.R * 2 + (long)(branch * A.R + line)) * C + 4 * q) = v[r];
                }
            }
        }
    }
    if (A.mode == 1 && sink == 12345.678f) A.out0[0] = sink;
}

// 256 x 256 x 64 channels, batch 2 (BASELINE configs[3]): 512 lines of 256 samples per axis, 4 lines per workgroup, two waves
// per line (spectral_x3k).  DUP = 2: both waves of a line request the whole line (what the kernel does: each wave transforms all
// samples into its own rows of the spectrum); DUP = 1: each wave requests its half of the samples.  Outputs: each wave writes
// its half of the line's samples into `nout` images; spectrum save: 2 K = 64 rows of 256 bytes per line.
struct Args256 {
    const float* in;
    float* out0;
    float* out1;
    float* spec;
    int nout, save, lines_per_wg, mode;
};

template <int VEC, int DUP>
__global__ __launch_bounds__(512) void k_stream256(Args256 A) {
    constexpr int C = 64, L = 256, N = 256, Bn = 2, R = Bn * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wg = blockIdx.x;
    const int branch = wg & 1, tile = wg >> 1;
    const int wpl = 8 / A.lines_per_wg;                    // waves per line (2 at 4 lines, 1 at 8 lines)
    const int line = tile * A.lines_per_wg + wave / wpl, part = wave % wpl;
    long base, es;
    if (branch == 0) {
        base = (long)line * L * C, es = C;
    } else {
        const int b = line / N, n = line % N;
        base = (long)b * L * N * C + (long)n * C, es = (long)N * C;
    }
    float sink = 0.f;
    constexpr int PER = 64;                                // samples per chunk
    const int s_lo = (DUP == 2 || wpl == 1) ? 0 : part * (L / wpl), s_hi = (DUP == 2 || wpl == 1) ? L : (part + 1) * (L / wpl);
    const int w_lo = part * (L / wpl), w_hi = (part + 1) * (L / wpl);
    for (int s0 = s_lo; s0 < s_hi; s0 += PER) {
        const bool wr = s0 >= w_lo && s0 < w_hi;           // this wave writes the chunk (each sample is written once)
        if (VEC == 2) {
            const int j = lane & 31, half = lane >> 5;
            f2 v[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = *reinterpret_cast<const f2*>(A.in + base + (s0 + 2 * i + half) * es + 2 * j);
            if (A.mode == 1 || !wr) {
#pragma unroll
                for (int i = 0; i < 32; ++i) sink += v[i].x + v[i].y;
                continue;
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                *reinterpret_cast<f2*>(A.out0 + base + (s0 + 2 * i + half) * es + 2 * j) = v[i];
                if (A.nout > 1) *reinterpret_cast<f2*>(A.out1 + base + (s0 + 2 * i + half) * es + 2 * j) = v[i] * 2.f;
            }
            if (A.save) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {       // 64 rows per line over its four written chunks: 16 rows per chunk
                    const int row = (s0 / PER) * 16 + 2 * r + half;
                    *reinterpret_cast<f2*>(A.spec + ((long)row * R * 2 + (long)(branch * R + line)) * C + 2 * j) = v[r];
                }
            }
        } else {
            const int q = lane & 15, g = lane >> 4;
            f4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const f4*>(A.in + base + (s0 + 4 * i + g) * es + 4 * q);
            if (A.mode == 1 || !wr) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sink += v[i].x + v[i].y + v[i].z + v[i].w;
                continue;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                *reinterpret_cast<f4*>(A.out0 + base + (s0 + 4 * i + g) * es + 4 * q) = v[i];
                if (A.nout > 1) *reinterpret_cast<f4*>(A.out1 + base + (s0 + 4 * i + g) * es + 4 * q) = v[i] * 2.f;
            }
            if (A.save) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (s0 / PER) * 16 + 4 * r + g;
                    *reinterpret_cast<f4*>(A.spec + ((long)row * R * 2 + (long)(branch * R + line)) * C + 4 * q) = v[r];
                }
            }
        }
    }
    if (sink == 12345.678f) A.out0[0] = sink;
}

// reference: a plain grid-stride float4 copy (read n bytes, write n bytes) with many workgroups
__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ in, f4* __restrict__ out, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) out[i] = in[i];
}

int main() {
    const int B = 32, M = 64, N = 64, C = 64, R = B * 64;
    const size_t img = (size_t)B * M * N * C * 4;              // 33.5 MB
    const size_t spec = (size_t)32 * R * 2 * C * 4;            // both branches' saved spectra: 33.5 MB
    const int NBUF = 10;                                       // rotate buffers: 10 x (in + 2 out + spec) = 1.3 GB >> 256 MB of MALL
    std::vector<float*> in(NBUF), o0(NBUF), o1(NBUF), sp(NBUF);
    for (int i = 0; i < NBUF; ++i) {
        hipMalloc(&in[i], img), hipMalloc(&o0[i], img), hipMalloc(&o1[i], img), hipMalloc(&sp[i], spec);
        hipMemset(in[i], 0, img);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int reps = 40;
    printf("%-44s %10s %10s %10s\n", "pattern (256 WGs x 512 threads, batch 32)", "us cold", "us warm", "MB moved");
    for (int vec : {2, 4})
        for (int cfg = 0; cfg < 6; ++cfg) {
            Args A{};
            A.B = B, A.M = M, A.N = N, A.R = R, A.lines_per_wg = 16;
            const char* name;
            double mb;
            switch (cfg) {
                case 0: A.mode = 1, A.nout = 0, A.save = 0, name = "read x", mb = img / 1e6; break;
                case 1: A.mode = 2, A.nout = 1, A.save = 0, name = "write 1 image", mb = img / 1e6; break;
                case 2: A.mode = 0, A.nout = 1, A.save = 0, name = "read x + write 1 image", mb = 2 * img / 1e6; break;
                case 3: A.mode = 0, A.nout = 2, A.save = 0, name = "read x + write 2 images", mb = 3 * img / 1e6; break;
                case 4: A.mode = 0, A.nout = 2, A.save = 1, name = "read x + write 2 images + save spectra", mb = (3 * img + spec) / 1e6; break;
                default: A.mode = 0, A.nout = 1, A.save = 1, name = "read x + write 1 image + save spectra", mb = (2 * img + spec) / 1e6; break;
            }
            float us[2];
            for (int warm = 0; warm < 2; ++warm) {
                auto launch = [&](int it) {
                    const int b = warm ? 0 : it % NBUF;
                    A.in = in[b], A.out0 = o0[b], A.out1 = o1[b], A.spec = sp[b];
                    if (vec == 2)
                        hipLaunchKernelGGL(k_stream<2>, dim3(256), dim3(512), 0, 0, A);
                    else
                        hipLaunchKernelGGL(k_stream<4>, dim3(256), dim3(512), 0, 0, A);
                };
                for (int it = 0; it < 5; ++it) launch(it);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int it = 0; it < reps; ++it) launch(it);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                us[warm] = 1e3f * ms / reps;
            }
            char label[96];
            snprintf(label, sizeof label, "%2d-byte: %s", 4 * vec, name);
            printf("%-44s %10.1f %10.1f %10.1f   (%.2f / %.2f TB/s)\n", label, us[0], us[1], mb, mb / us[0], mb / us[1]);
        }
    printf("\n%-52s %10s %10s %10s\n", "256 x 256, batch 2, 32 modes (256 WGs x 512 threads)", "us cold", "us warm", "MB moved");
    for (int v = 0; v < 6; ++v) {
        Args256 A{};
        const int vec = (v & 1) ? 4 : 2, dup = v < 4 ? (v < 2 ? 2 : 1) : 1, lpw = v < 4 ? 4 : 8;
        A.nout = 2, A.save = 1, A.lines_per_wg = lpw, A.mode = 0;
        const double mb = (3.0 * img + 2.0 * 64 * 512 * 256) / 1e6;
        float us[2];
        for (int warm = 0; warm < 2; ++warm) {
            auto launch = [&](int it) {
                const int b = warm ? 0 : it % NBUF;
                A.in = in[b], A.out0 = o0[b], A.out1 = o1[b], A.spec = sp[b];
                const int wgs = 1024 / lpw;
                if (vec == 2 && dup == 2) hipLaunchKernelGGL((k_stream256<2, 2>), dim3(wgs), dim3(512), 0, 0, A);
                if (vec == 4 && dup == 2) hipLaunchKernelGGL((k_stream256<4, 2>), dim3(wgs), dim3(512), 0, 0, A);
                if (vec == 2 && dup == 1) hipLaunchKernelGGL((k_stream256<2, 1>), dim3(wgs), dim3(512), 0, 0, A);
                if (vec == 4 && dup == 1) hipLaunchKernelGGL((k_stream256<4, 1>), dim3(wgs), dim3(512), 0, 0, A);
            };
            for (int it = 0; it < 5; ++it) launch(it);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int it = 0; it < reps; ++it) launch(it);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            us[warm] = 1e3f * ms / reps;
        }
        char label[96];
        snprintf(label, sizeof label, "%2d-byte, %d lines/WG, line requested %dx", 4 * vec, lpw, dup);
        printf("%-52s %10.1f %10.1f %10.1f   (%.2f / %.2f TB/s)\n", label, us[0], us[1], mb, mb / us[0], mb / us[1]);
    }
    for (int wgs : {256, 1024, 4096}) {
        float us[2];
        for (int warm = 0; warm < 2; ++warm) {
            for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, 0, (const f4*)in[warm ? 0 : it % NBUF], (f4*)o0[warm ? 0 : it % NBUF], (long)(img / 16));
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int it = 0; it < reps; ++it)
                hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, 0, (const f4*)in[warm ? 0 : it % NBUF], (f4*)o0[warm ? 0 : it % NBUF], (long)(img / 16));
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            us[warm] = 1e3f * ms / reps;
        }
        printf("plain float4 copy of one image, %4d WGs x 256 %10.1f %10.1f %10.1f   (%.2f / %.2f TB/s)\n", wgs, us[0], us[1], 2 * img / 1e6,
               2 * img / 1e6 / us[0], 2 * img / 1e6 / us[1]);
    }
    return 0;
}
