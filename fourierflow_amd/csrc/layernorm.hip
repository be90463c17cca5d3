// LayerNorm over the channel axis of channels-last activations, as the last stage of FeedForward(layer_norm=True)
// (reference fourierflow/modules/feedforward.py:18-19: nn.LayerNorm(out_dim) after the last linear; torch semantics: biased
// variance, eps inside the square root, elementwise affine), fused with the layer's residual add (grid_2d.py:169), and its
// backward fused with the sum of the two gradient buffers of the paired spectral launch.
//   fwd:  y[p][c] = (t[p][c] - mean_p) * rstd_p * gamma[c] + beta[c]  (+ resid[p][c]);   stats[p] = {mean_p, rstd_p}
//   bwd:  gy = g (+ g2);  a = gy * gamma;  dt = rstd (a - mean_c(a) - xhat mean_c(a xhat));
//         dgamma = sum_p gy xhat,  dbeta = sum_p gy   (deterministic: ordered slices, then an ordered reduction)
// One wave per pixel (C = 64: a lane per channel; C = 32: two pixels per wave); nothing here is matrix work.
#include "ffno_device.h"
#include "ffno.h"

namespace ffno {

template <int C>
__device__ __forceinline__ float group_sum(float v) {      // sum over the C lanes that hold one pixel
    FFNO_UNROLL
    for (int m = C / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

template <int C>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ t, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* resid, float* out,
                                                            float* __restrict__ stats, long P, float eps) {
    constexpr int PPB = 256 / C;                     // pixels per block and step
    const int c = threadIdx.x % C, q = threadIdx.x / C;
    const float ga = gamma[c], be = beta[c];
    for (long p = (long)blockIdx.x * PPB + q; p < P; p += (long)gridDim.x * PPB) {
        const float v = t[p * C + c];
        const float mean = group_sum<C>(v) * (1.f / C);
        const float d = v - mean;
        const float var = group_sum<C>(d * d) * (1.f / C);
        const float rstd = 1.f / sqrtf(var + eps);
        float y = d * rstd * ga + be;
        if (resid) y += resid[p * C + c];
        out[p * C + c] = y;
        if (c == 0) stats[2 * p] = mean, stats[2 * p + 1] = rstd;
    }
}

// partial[block][2][C]: the block's pixels are a contiguous slice, accumulated in order
template <int C>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ t, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* g, const float* g2,
                                                            float* g_sum, float* __restrict__ dt, float* __restrict__ partial,
                                                            long P) {
    constexpr int PPB = 256 / C;
    __shared__ float red[2][256];
    const int c = threadIdx.x % C, q = threadIdx.x / C;
    const float ga = gamma[c];
    const long per = (P + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per, p1 = min(P, p0 + per);
    float sg = 0.f, sb = 0.f;
    for (long p = p0 + q; p < p1; p += PPB) {
        float gy = g[p * C + c];
        if (g2) {
            gy += g2[p * C + c];
            if (g_sum) g_sum[p * C + c] = gy;
        }
        const float mean = stats[2 * p], rstd = stats[2 * p + 1];
        const float xhat = (t[p * C + c] - mean) * rstd;
        const float a = gy * ga;
        const float m1 = group_sum<C>(a) * (1.f / C);
        const float m2 = group_sum<C>(a * xhat) * (1.f / C);
        dt[p * C + c] = rstd * (a - m1 - xhat * m2);
        sg += gy * xhat;
        sb += gy;
    }
    red[0][threadIdx.x] = sg;
    red[1][threadIdx.x] = sb;
    __syncthreads();
    if (threadIdx.x < C) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < PPB; ++k) a += red[0][k * C + threadIdx.x], b += red[1][k * C + threadIdx.x];
        partial[((long)blockIdx.x * 2 + 0) * C + threadIdx.x] = a;
        partial[((long)blockIdx.x * 2 + 1) * C + threadIdx.x] = b;
    }
}

__global__ __launch_bounds__(64) void layernorm_bwd_reduce_kernel(const float* __restrict__ partial, float* dgamma, float* dbeta,
                                                                  int C, int nsplit, int accumulate) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int s = 0; s < nsplit; ++s) a += partial[((long)s * 2 + 0) * C + c], b += partial[((long)s * 2 + 1) * C + c];
    dgamma[c] = accumulate ? dgamma[c] + a : a;
    dbeta[c] = accumulate ? dbeta[c] + b : b;
}

static inline int ln_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFNO_OK : (int)e;
}

}  // namespace ffno

using namespace ffno;

extern "C" int ffno_layernorm_fwd(const float* t, const float* gamma, const float* beta, const float* resid, float* out,
                                  float* stats, long P, int C, float eps, void* stream) {
    if (!t || !gamma || !beta || !out || !stats || P <= 0 || !(eps > 0.f)) return FFNO_EINVAL;
    const int blocks = (int)min((long)2048, (P + 3) / 4);
    if (C == 64)
        FFNO_LAUNCH(layernorm_fwd_kernel<64>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, gamma, beta, resid, out, stats, P, eps);
    else if (C == 32)
        FFNO_LAUNCH(layernorm_fwd_kernel<32>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, gamma, beta, resid, out, stats, P, eps);
    else
        return FFNO_EUNSUPPORTED;
    return ln_status();
}

extern "C" int ffno_layernorm_nsplit(long P) { return (int)max((long)1, min((long)512, (P + 255) / 256)); }

extern "C" int ffno_layernorm_bwd(const float* t, const float* stats, const float* gamma, const float* g, const float* g2,
                                  float* g_sum, float* dt, float* partial, float* dgamma, float* dbeta, long P, int C,
                                  int accumulate, void* stream) {
    if (!t || !stats || !gamma || !g || !dt || !partial || !dgamma || !dbeta || P <= 0) return FFNO_EINVAL;
    const int ns = ffno_layernorm_nsplit(P);
    hipStream_t st = (hipStream_t)stream;
    if (C == 64)
        FFNO_LAUNCH(layernorm_bwd_kernel<64>, dim3(ns), dim3(256), 0, st, t, stats, gamma, g, g2, g_sum, dt, partial, P);
    else if (C == 32)
        FFNO_LAUNCH(layernorm_bwd_kernel<32>, dim3(ns), dim3(256), 0, st, t, stats, gamma, g, g2, g_sum, dt, partial, P);
    else
        return FFNO_EUNSUPPORTED;
    FFNO_LAUNCH(layernorm_bwd_reduce_kernel, dim3((C + 63) / 64), dim3(64), 0, st, partial, dgamma, dbeta, C, ns, accumulate);
    return ln_status();
}
