// Non-factorized 2-D spectral convolution (FNOPlus2DBlock, the "no factorization" F-FNO ablation):
//     out = irfft2( pad( corner_mix( rfft2(x, norm='ortho') ) ), norm='ortho' )
// reference fourierflow/modules/zongyi_fno/grid_plus_2d.py:52-83 -- two K x K corner blocks of the half spectrum
// (rows [0,K) and [M-K,M), columns [0,K)) mixed with their own [I,O,K,K,2] weight tensors.
//
// The transform along the LAST axis, the per-mode channel mix and the weight-gradient contraction are the kernels of
// spectral.hip (the 2K*K retained (kx,ky) modes are just "modes" to them, with the B samples as rows).  This file
// adds what the factorized operator never needs:
//   * the complex DFT along the FIRST axis of the already y-transformed spectrum, restricted to the 2K retained rows
//     (forward) / zero-padded back to M rows (inverse);   kx' in [0,K) -> kx = kx',  kx' in [K,2K) -> kx = M - 2K + kx'
//   * the weight repack  [I][O][Kx][Ky][2] x 2  ->  planes[mode = ky*2K + kx'][re/im][I][O]  (+ transposed copy)
//   * the gradient scatter back into the two parameter tensors.
// The 2-D mix streams 2*K*K*C*C*8 B of weights (16.8 MB at K = 16, C = 64) for B rows per mode: weight-bandwidth-bound;
// the DFT along x is 0.5 GFLOP at the markov shape and runs on the vector ALUs.
#include "ffno_device.h"
#include "ffno.h"

#include <cmath>

namespace ffno {

__device__ __forceinline__ void sincos_2pi_frac2(int k, int n, float& s, float& c) {   // angle = 2 pi k / n, 0 <= k < n
    plat::sincos_pi(2.f * (float)k / (float)n, s, c);
}

__device__ __forceinline__ int kx_of(int kxp, int K, int M) { return kxp < K ? kxp : M - 2 * K + kxp; }

// forward:  Z[ky][kx'][b][ri][c] = 1/sqrt(M) sum_m e^{-2 pi i kx m / M} S[ky][b*M + m][ri][c]
// block = (ky, b); the [M][2C] slab of the sample is staged in LDS (coalesced), thread = (channel c, group g of output
// rows); twiddle table in LDS, index advanced incrementally (kx m mod M)
__global__ __launch_bounds__(256) void cdft_fwd_kernel(const float* __restrict__ S, float* __restrict__ Z, int B, int M,
                                                       int C, int K) {
    FFNO_DYN_SMEM(smem);
    float* ct = reinterpret_cast<float*>(smem);
    float* st = ct + M;
    float* slab = st + M;                          // [M][2C]
    const int ky = blockIdx.x / B, b = blockIdx.x % B;
    const float* src = S + ((long)ky * B * M + (long)b * M) * 2 * C;
    for (int m = threadIdx.x; m < M; m += blockDim.x) sincos_2pi_frac2(m, M, st[m], ct[m]);
    for (int e = threadIdx.x * 4; e < M * 2 * C; e += blockDim.x * 4)
        *reinterpret_cast<float4*>(slab + e) = *reinterpret_cast<const float4*>(src + e);
    __syncthreads();
    const int G = blockDim.x / C;                 // row groups
    const int c = threadIdx.x % C, g = threadIdx.x / C;
    const float scale = rsqrtf((float)M);
    // four output rows per pass: every slab element read from LDS feeds four complex multiply-adds
    for (int k0 = 4 * g; k0 < 2 * K; k0 += 4 * G) {
        int kx[4], idx[4];
        float re[4], im[4];
        FFNO_UNROLL
        for (int u = 0; u < 4; ++u) {
            kx[u] = kx_of(min(k0 + u, 2 * K - 1), K, M);
            idx[u] = 0;
            re[u] = im[u] = 0.f;
        }
        for (int m = 0; m < M; ++m) {             // (a + i bb)(cos - i sin)
            const float a = slab[m * 2 * C + c], bb = slab[m * 2 * C + C + c];
            FFNO_UNROLL
            for (int u = 0; u < 4; ++u) {
                const float cs = ct[idx[u]], sn = st[idx[u]];
                re[u] += a * cs + bb * sn;
                im[u] += bb * cs - a * sn;
                idx[u] += kx[u];
                if (idx[u] >= M) idx[u] -= M;
            }
        }
        FFNO_UNROLL
        for (int u = 0; u < 4; ++u) {
            if (k0 + u < 2 * K) {
                float* dst = Z + (((long)ky * 2 * K + k0 + u) * B + b) * 2 * C;
                dst[c] = re[u] * scale;
                dst[C + c] = im[u] * scale;
            }
        }
    }
}

// inverse:  S[ky][b*M + m][ri][c] = 1/sqrt(M) sum_{kx'} e^{+2 pi i kx m / M} Z[ky][kx'][b][ri][c]
// the sample's [2K][2C] retained rows are staged in LDS; the two runs of kx (0..K-1 and M-K..M-1) advance the twiddle
// index by m per step
__global__ __launch_bounds__(256) void cdft_inv_kernel(const float* __restrict__ Z, float* __restrict__ S, int B, int M,
                                                       int C, int K) {
    FFNO_DYN_SMEM(smem);
    float* ct = reinterpret_cast<float*>(smem);
    float* st = ct + M;
    float* slab = st + M;                          // [2K][2C]
    const int ky = blockIdx.x / B, b = blockIdx.x % B;
    for (int m = threadIdx.x; m < M; m += blockDim.x) sincos_2pi_frac2(m, M, st[m], ct[m]);
    for (int e = threadIdx.x * 4; e < 2 * K * 2 * C; e += blockDim.x * 4) {
        const int kxp = e / (2 * C), off = e % (2 * C);
        *reinterpret_cast<float4*>(slab + e) =
            *reinterpret_cast<const float4*>(Z + (((long)ky * 2 * K + kxp) * B + b) * 2 * C + off);
    }
    __syncthreads();
    const int G = blockDim.x / C;
    const int c = threadIdx.x % C, g = threadIdx.x / C;
    const float scale = rsqrtf((float)M);
    float* dst = S + ((long)ky * B * M + (long)b * M) * 2 * C;
    // four output positions per pass: every retained row read from LDS feeds four complex multiply-adds
    for (int m0 = 4 * g; m0 < M; m0 += 4 * G) {
        float re[4], im[4];
        FFNO_UNROLL
        for (int u = 0; u < 4; ++u) re[u] = im[u] = 0.f;
        FFNO_UNROLL
        for (int run = 0; run < 2; ++run) {       // kx = 0..K-1, then kx = M-K..M-1
            int idx[4], stp[4];
            FFNO_UNROLL
            for (int u = 0; u < 4; ++u) {
                stp[u] = (m0 + u) % M;
                idx[u] = run == 0 ? 0 : (int)(((long)(M - K) * (m0 + u)) % M);
            }
            for (int t = 0; t < K; ++t) {         // (a + i bb)(cos + i sin)
                const float* z = slab + (run * K + t) * 2 * C;
                const float a = z[c], bb = z[C + c];
                FFNO_UNROLL
                for (int u = 0; u < 4; ++u) {
                    const float cs = ct[idx[u]], sn = st[idx[u]];
                    re[u] += a * cs - bb * sn;
                    im[u] += a * sn + bb * cs;
                    idx[u] += stp[u];
                    if (idx[u] >= M) idx[u] -= M;
                }
            }
        }
        FFNO_UNROLL
        for (int u = 0; u < 4; ++u) {
            if (m0 + u < M) {
                dst[(long)(m0 + u) * 2 * C + c] = re[u] * scale;
                dst[(long)(m0 + u) * 2 * C + C + c] = im[u] * scale;
            }
        }
    }
}

// planes[mode = ky*2K + kx'][ri][i][o]  <-  w_{kx' / K}[i][o][kx' % K][ky][ri]   (+ wpt[mode][ri][o][i]);  K = Kx retained
// rows per corner block, Ky retained columns (FNOMesh2D keeps modes1 != modes2, zongyi_fno/mesh_2d.py:46-49)
__global__ void fw2d_pack_kernel(const float* __restrict__ w0, const float* __restrict__ w1, float* __restrict__ wp,
                                 float* __restrict__ wpt, int C, int K, int Ky) {
    const long total = (long)2 * K * Ky * 2 * C * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int o = e % C;
        const int i = (e / C) % C;
        const int ri = (e / ((long)C * C)) % 2;
        const int mode = e / ((long)C * C * 2);
        const int kxp = mode % (2 * K), ky = mode / (2 * K);
        const float* w = kxp < K ? w0 : w1;
        const float v = w[((((long)i * C + o) * K + (kxp % K)) * Ky + ky) * 2 + ri];
        wp[e] = v;
        wpt[(((long)mode * 2 + ri) * C + o) * C + i] = v;
    }
}

// partial[split][mode][ri][i][o] summed over splits -> gw_{kx'/K}[i][o][kx' % K][ky][ri]
__global__ void fw2d_grad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw0,
                                        float* __restrict__ gw1, int C, int K, int Ky, int nsplit, int accumulate) {
    const long total = (long)2 * K * Ky * 2 * C * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int o = e % C;
        const int i = (e / C) % C;
        const int ri = (e / ((long)C * C)) % 2;
        const int mode = e / ((long)C * C * 2);
        const int kxp = mode % (2 * K), ky = mode / (2 * K);
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += partial[(long)sp * total + e];
        float* g = (kxp < K ? gw0 : gw1) + ((((long)i * C + o) * K + (kxp % K)) * Ky + ky) * 2 + ri;
        *g = accumulate ? (*g + s) : s;
    }
}

// ---- 3-D corner blocks (FNOMesh3D, zongyi_fno/mesh_3d.py:38-57): four weight tensors [I][O][K1][K2][K3][2], one per sign
// pair of (kx, ky):  w[0] = (+x, +y), w[1] = (-x, +y), w[2] = (+x, -y), w[3] = (-x, -y).
// planes[mode = (kz * 2K2 + ky') * 2K1 + kx'][ri][i][o],  kx' >= K1 / ky' >= K2 = the negative-frequency rows.
struct Fw3dPtrs {
    float* w[4];
};

__global__ void fw3d_pack_kernel(Fw3dPtrs W, float* __restrict__ wp, float* __restrict__ wpt, int C, int K1, int K2, int K3) {
    const long total = (long)K3 * 2 * K2 * 2 * K1 * 2 * C * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int o = e % C;
        const int i = (e / C) % C;
        const int ri = (e / ((long)C * C)) % 2;
        const long mode = e / ((long)C * C * 2);
        const int kxp = mode % (2 * K1), kyp = (mode / (2 * K1)) % (2 * K2), kz = (int)(mode / ((long)4 * K1 * K2));
        const float* w = W.w[(kxp >= K1 ? 1 : 0) + (kyp >= K2 ? 2 : 0)];
        const float v = w[(((((long)i * C + o) * K1 + (kxp % K1)) * K2 + (kyp % K2)) * K3 + kz) * 2 + ri];
        wp[e] = v;
        wpt[((mode * 2 + ri) * C + o) * C + i] = v;
    }
}

__global__ void fw3d_grad_reduce_kernel(const float* __restrict__ partial, Fw3dPtrs G, int C, int K1, int K2, int K3, int nsplit,
                                        int accumulate) {
    const long total = (long)K3 * 2 * K2 * 2 * K1 * 2 * C * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int o = e % C;
        const int i = (e / C) % C;
        const int ri = (e / ((long)C * C)) % 2;
        const long mode = e / ((long)C * C * 2);
        const int kxp = mode % (2 * K1), kyp = (mode / (2 * K1)) % (2 * K2), kz = (int)(mode / ((long)4 * K1 * K2));
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += partial[(long)sp * total + e];
        float* g = G.w[(kxp >= K1 ? 1 : 0) + (kyp >= K2 ? 2 : 0)] +
                   (((((long)i * C + o) * K1 + (kxp % K1)) * K2 + (kyp % K2)) * K3 + kz) * 2 + ri;
        *g = accumulate ? (*g + s) : s;
    }
}

static inline int s2d_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFNO_OK : (int)e;
}

}  // namespace ffno

using namespace ffno;

// Kx = retained rows per corner block (2 Kx rows in all), Ky = retained columns (the K of the preceding ffno_dft_fwd)
extern "C" int ffno_cdft_rows2(const float* in, float* out, int B, int M, int C, int Kx, int Ky, int inverse, void* stream) {
    if (!in || !out || B <= 0 || M <= 0 || Kx <= 0 || Ky <= 0) return FFNO_EINVAL;
    if (C != 64 && C != 32) return FFNO_EUNSUPPORTED;
    if (2 * Kx > M) return FFNO_EMODES;
    const dim3 grid((unsigned)((long)Ky * B)), block(256);
    const size_t smem = sizeof(float) * (2 * (size_t)M + (inverse ? (size_t)2 * Kx * 2 * C : (size_t)M * 2 * C));
    if (smem > 150 * 1024) return FFNO_EUNSUPPORTED;    // the staged slab must fit LDS (M <= 288 at C = 64)
    {     // dynamic LDS beyond the default window needs an explicit opt-in
        const int e = inverse ? allow_dynamic_lds(cdft_inv_kernel, smem) : allow_dynamic_lds(cdft_fwd_kernel, smem);
        if (e) return e;
    }
    if (inverse)
        FFNO_LAUNCH(cdft_inv_kernel, grid, block, smem, (hipStream_t)stream, in, out, B, M, C, Kx);
    else
        FFNO_LAUNCH(cdft_fwd_kernel, grid, block, smem, (hipStream_t)stream, in, out, B, M, C, Kx);
    return s2d_status();
}

extern "C" int ffno_cdft_rows(const float* in, float* out, int B, int M, int C, int K, int inverse, void* stream) {
    return ffno_cdft_rows2(in, out, B, M, C, K, K, inverse, stream);
}

extern "C" int ffno_fw2d_pack2(const float* w0, const float* w1, float* wp, float* wpt, int C, int Kx, int Ky, void* stream) {
    if (!w0 || !w1 || !wp || !wpt || C <= 0 || Kx <= 0 || Ky <= 0) return FFNO_EINVAL;
    const long total = (long)2 * Kx * Ky * 2 * C * C;
    FFNO_LAUNCH(fw2d_pack_kernel, dim3((unsigned)min((total + 255) / 256, 4096L)), dim3(256), 0, (hipStream_t)stream, w0,
                w1, wp, wpt, C, Kx, Ky);
    return s2d_status();
}

extern "C" int ffno_fw2d_pack(const float* w0, const float* w1, float* wp, float* wpt, int C, int K, void* stream) {
    return ffno_fw2d_pack2(w0, w1, wp, wpt, C, K, K, stream);
}

extern "C" int ffno_fw2d_grad_reduce2(const float* partial, float* gw0, float* gw1, int C, int Kx, int Ky, int nsplit,
                                      int accumulate, void* stream) {
    if (!partial || !gw0 || !gw1 || C <= 0 || Kx <= 0 || Ky <= 0 || nsplit <= 0) return FFNO_EINVAL;
    const long total = (long)2 * Kx * Ky * 2 * C * C;
    FFNO_LAUNCH(fw2d_grad_reduce_kernel, dim3((unsigned)min((total + 255) / 256, 4096L)), dim3(256), 0,
                (hipStream_t)stream, partial, gw0, gw1, C, Kx, Ky, nsplit, accumulate);
    return s2d_status();
}

extern "C" int ffno_fw2d_grad_reduce(const float* partial, float* gw0, float* gw1, int C, int K, int nsplit,
                                     int accumulate, void* stream) {
    return ffno_fw2d_grad_reduce2(partial, gw0, gw1, C, K, K, nsplit, accumulate, stream);
}

extern "C" int ffno_fw3d_pack(const float* w1, const float* w2, const float* w3, const float* w4, float* wp, float* wpt, int C,
                              int K1, int K2, int K3, void* stream) {
    if (!w1 || !w2 || !w3 || !w4 || !wp || !wpt || C <= 0 || K1 <= 0 || K2 <= 0 || K3 <= 0) return FFNO_EINVAL;
    const long total = (long)K3 * 2 * K2 * 2 * K1 * 2 * C * C;
    Fw3dPtrs W{{const_cast<float*>(w1), const_cast<float*>(w2), const_cast<float*>(w3), const_cast<float*>(w4)}};
    FFNO_LAUNCH(fw3d_pack_kernel, dim3((unsigned)min((total + 255) / 256, 8192L)), dim3(256), 0, (hipStream_t)stream, W, wp, wpt,
                C, K1, K2, K3);
    return s2d_status();
}

extern "C" int ffno_fw3d_grad_reduce(const float* partial, float* g1, float* g2, float* g3, float* g4, int C, int K1, int K2,
                                     int K3, int nsplit, int accumulate, void* stream) {
    if (!partial || !g1 || !g2 || !g3 || !g4 || C <= 0 || K1 <= 0 || K2 <= 0 || K3 <= 0 || nsplit <= 0) return FFNO_EINVAL;
    const long total = (long)K3 * 2 * K2 * 2 * K1 * 2 * C * C;
    Fw3dPtrs G{{g1, g2, g3, g4}};
    FFNO_LAUNCH(fw3d_grad_reduce_kernel, dim3((unsigned)min((total + 255) / 256, 8192L)), dim3(256), 0, (hipStream_t)stream,
                partial, G, C, K1, K2, K3, nsplit, accumulate);
    return s2d_status();
}
