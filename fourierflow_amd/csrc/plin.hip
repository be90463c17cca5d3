// Pointwise ("1x1") linear layers of the FNOZongyi2DBlock baseline (BASELINE config 0).
//
// Replaces, in reference fourierflow/modules/zongyi_fno/grid_2d.py:
//     SpectralConv2d.linear + residual + ReLU   act(x_spec + linear(x))          :22,45,74-77
//     FNOZongyi2DBlock.in_proj                  nn.Linear(input_dim, width)      :106
//     FNOZongyi2DBlock.feedforward              Linear(width,128) ReLU Linear(128,1)   :119-122
// and their backward passes.  The baseline is 20 channels wide; its activations live in channels-last buffers padded to
// the 32-channel tiles of the spectral kernels (pad channels are exact zeros end to end), so every routine here takes a
// leading dimension next to the logical width.  All of this is a few FLOPs per byte: plain VALU code on LDS tiles of 64
// points, one fused pass per layer (bias, residual add, ReLU and the ReLU mask of the backward pass are folded in).
#include "ffno_device.h"
#include "ffno.h"

namespace ffno {

static constexpr int kPlinTile = 64;    // points per LDS tile

// activation modes (include/ffno.h FFNO_ACT_*): 0 none, 1 ReLU, 2 GELU (exact, erf: torch.nn.functional.gelu default)
__device__ __forceinline__ float plin_act(float v, int mode) {
    if (mode == 1) return v > 0.f ? v : 0.f;
    if (mode == 2) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    return v;
}
// d act / d pre.  `a` is what the forward pass kept: the OUTPUT for ReLU (sign is enough), the PRE-activation for GELU.
__device__ __forceinline__ float plin_dact(float a, int mode) {
    if (mode == 1) return a > 0.f ? 1.f : 0.f;
    if (mode == 2) return 0.5f * (1.f + erff(a * 0.70710678118654752f)) + a * 0.3989422804014327f * expf(-0.5f * a * a);
    return 1.f;
}
static constexpr int kPlinOB = 8;       // outputs per thread in the weight-gradient kernel
static constexpr int kPlinMaxJ = 5;     // (output group, input column) items per thread: 16 groups * 65 columns / 256

// out[p][o] = act(b[o] + sum_i W[o][i] x[p][i] + add[p][o])  for o < Cout,  0 for Cout <= o < ldo;
// optional second output out2 = out + res (the block-level residual x + layer(x), which must not disturb the ReLU mask)
// VEC (ldo % 4 == 0): a thread owns four consecutive outputs of a point -- one LDS read of x[p][i] and one 16-byte read of
// the transposed weights per four FMAs (the scalar form spends two LDS reads per FMA and is LDS-bound).
template <bool VEC>
__global__ __launch_bounds__(256) void plin_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ W,
                                                       const float* __restrict__ b, const float* __restrict__ add,
                                                       float* __restrict__ out, int ldo, const float* __restrict__ res,
                                                       float* __restrict__ out2, float* __restrict__ pre_out, long P,
                                                       int Cin, int Cout, int act_mode) {
    FFNO_DYN_SMEM(smem);
    const int t = threadIdx.x, sx = Cin + 1;
    const int cp = VEC ? (Cout + 3) / 4 * 4 : Cout;       // VEC: weights transposed to [Cin][cp], zero-padded columns
    float* xs = reinterpret_cast<float*>(smem);          // [tile][Cin + 1]
    float* ws = xs + (kPlinTile * sx + 3) / 4 * 4;       // VEC: [Cin][cp]   scalar: [Cout][Cin + 1]
    float* bs = ws + (VEC ? Cin * cp : Cout * sx);       // [cp]
    if (VEC) {
        for (int e = t; e < Cin * cp; e += 256) {
            const int i = e / cp, o = e - i * cp;
            ws[e] = o < Cout ? W[o * Cin + i] : 0.f;
        }
        for (int e = t; e < cp; e += 256) bs[e] = (b && e < Cout) ? b[e] : 0.f;
    } else {
        for (int e = t; e < Cout * Cin; e += 256) ws[(e / Cin) * sx + e % Cin] = W[e];
        for (int e = t; e < Cout; e += 256) bs[e] = b ? b[e] : 0.f;
    }
    const long ntile = (P + kPlinTile - 1) / kPlinTile;
    for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long p0 = tile * kPlinTile;
        const int np = (int)((P - p0) < kPlinTile ? (P - p0) : kPlinTile);
        __syncthreads();
        for (int e = t; e < np * Cin; e += 256) {
            const int p = e / Cin, i = e - p * Cin;
            xs[p * sx + i] = x[(p0 + p) * ldx + i];
        }
        __syncthreads();
        if (VEC) {
            const int l4 = ldo / 4;
            for (int e = t; e < np * l4; e += 256) {
                const int p = e / l4, o = 4 * (e - p * l4);
                const long idx = (p0 + p) * ldo + o;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (o < Cout) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(bs + o);
                    v[0] = bb[0], v[1] = bb[1], v[2] = bb[2], v[3] = bb[3];
                    const float* xr = xs + p * sx;
                    for (int i = 0; i < Cin; ++i) {
                        const float xv = xr[i];
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(ws + i * cp + o);
                        v[0] = fmaf(w4[0], xv, v[0]);
                        v[1] = fmaf(w4[1], xv, v[1]);
                        v[2] = fmaf(w4[2], xv, v[2]);
                        v[3] = fmaf(w4[3], xv, v[3]);
                    }
                }
                float prev[4], outv[4];
                FFNO_UNROLL
                for (int u = 0; u < 4; ++u) {
                    const bool live = o + u < Cout;
                    float a = live ? v[u] : 0.f;
                    if (live && add) a += add[idx + u];
                    prev[u] = a;
                    outv[u] = live ? plin_act(a, act_mode) : 0.f;
                }
                *reinterpret_cast<f32x4*>(out + idx) = f32x4{outv[0], outv[1], outv[2], outv[3]};
                if (pre_out) *reinterpret_cast<f32x4*>(pre_out + idx) = f32x4{prev[0], prev[1], prev[2], prev[3]};
                if (out2) {
                    const f32x4 r4 = *reinterpret_cast<const f32x4*>(res + idx);
                    *reinterpret_cast<f32x4*>(out2 + idx) = f32x4{outv[0] + r4[0], outv[1] + r4[1], outv[2] + r4[2], outv[3] + r4[3]};
                }
            }
        } else {
            for (int e = t; e < np * ldo; e += 256) {
                const int p = e / ldo, o = e - p * ldo;
                float v = 0.f;
                if (o < Cout) {
                    v = bs[o];
                    const float* wr = ws + o * sx;
                    const float* xr = xs + p * sx;
                    for (int i = 0; i < Cin; ++i) v = fmaf(wr[i], xr[i], v);
                    if (add) v += add[(p0 + p) * ldo + o];
                    if (pre_out) pre_out[(p0 + p) * ldo + o] = v;
                    v = plin_act(v, act_mode);
                } else if (pre_out) {
                    pre_out[(p0 + p) * ldo + o] = 0.f;
                }
                out[(p0 + p) * ldo + o] = v;
                if (out2) out2[(p0 + p) * ldo + o] = v + res[(p0 + p) * ldo + o];
            }
        }
    }
}

// dpre[p][o] = g[p][o] * act'(.);   dx[p][i] (+)= sum_o dpre[p][o] W[o][i];   optional copy of dpre
// VEC (ldx % 4 == 0, Cin % 4 == 0): a thread owns four consecutive inputs of a point (one broadcast read of dpre[p][o] and one
// 16-byte read of a weight row per four FMAs).
template <bool VEC>
__global__ __launch_bounds__(256) void plin_bwd_data_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ act,
                                                            const float* __restrict__ W, float* __restrict__ dx, int ldx,
                                                            float* __restrict__ dpre_out, long P, int Cin, int Cout,
                                                            int accumulate, int act_mode) {
    FFNO_DYN_SMEM(smem);
    const int t = threadIdx.x, sd = Cout + 1;
    float* ws = reinterpret_cast<float*>(smem);          // [Cout][Cin]   (first: 16-byte aligned rows when Cin % 4 == 0)
    float* ds = ws + (Cout * Cin + 3) / 4 * 4;           // [tile][Cout + 1]
    for (int e = t; e < Cout * Cin; e += 256) ws[e] = W[e];
    const long ntile = (P + kPlinTile - 1) / kPlinTile;
    for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long p0 = tile * kPlinTile;
        const int np = (int)((P - p0) < kPlinTile ? (P - p0) : kPlinTile);
        __syncthreads();
        for (int e = t; e < np * ldg; e += 256) {
            const int p = e / ldg, o = e - p * ldg;
            const long idx = (p0 + p) * ldg + o;
            float v = 0.f;
            if (o < Cout) {
                v = g[idx];
                if (act) v *= plin_dact(act[idx], act_mode);
                ds[p * sd + o] = v;
            }
            if (dpre_out) dpre_out[idx] = v;
        }
        __syncthreads();
        if (VEC) {
            const int l4 = ldx / 4;
            for (int e = t; e < np * l4; e += 256) {
                const int p = e / l4, i = 4 * (e - p * l4);
                const long idx = (p0 + p) * ldx + i;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (i < Cin) {
                    const float* dr = ds + p * sd;
                    for (int o = 0; o < Cout; ++o) {
                        const float d = dr[o];
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(ws + o * Cin + i);
                        v[0] = fmaf(d, w4[0], v[0]);
                        v[1] = fmaf(d, w4[1], v[1]);
                        v[2] = fmaf(d, w4[2], v[2]);
                        v[3] = fmaf(d, w4[3], v[3]);
                    }
                }
                f32x4 r = f32x4{v[0], v[1], v[2], v[3]};
                if (accumulate) {
                    const f32x4 old = *reinterpret_cast<const f32x4*>(dx + idx);
                    r = f32x4{old[0] + v[0], old[1] + v[1], old[2] + v[2], old[3] + v[3]};
                }
                *reinterpret_cast<f32x4*>(dx + idx) = r;
            }
        } else {
            for (int e = t; e < np * ldx; e += 256) {
                const int p = e / ldx, i = e - p * ldx;
                const long idx = (p0 + p) * ldx + i;
                float v = 0.f;
                if (i < Cin) {
                    const float* dr = ds + p * sd;
                    for (int o = 0; o < Cout; ++o) v = fmaf(dr[o], ws[o * Cin + i], v);
                }
                dx[idx] = accumulate ? dx[idx] + v : v;
            }
        }
    }
}

// part[s][o][i] = sum over the points of slice s of dpre[p][o] * x[p][i],  i = Cin is the bias column (x = 1).
// A thread owns one input column i and kPlinOB consecutive outputs: per point one LDS read of x and kPlinOB / 4 vector
// reads of dpre (a broadcast: the lanes of a wave share o) feed kPlinOB FMAs.
__global__ __launch_bounds__(256) void plin_wgrad_partial_kernel(const float* __restrict__ g, int ldg,
                                                                 const float* __restrict__ act, const float* __restrict__ x,
                                                                 int ldx, float* __restrict__ part, long P, int Cin, int Cout,
                                                                 long per, int act_mode) {
    FFNO_DYN_SMEM(smem);
    const int t = threadIdx.x, sx = Cin + 1, npairs = Cout * sx;
    const int sdp = (Cout + kPlinOB - 1) / kPlinOB * kPlinOB;       // dpre row, padded with zeros to whole output groups
    float* ds = reinterpret_cast<float*>(smem);          // [tile][sdp]
    float* xs = ds + kPlinTile * sdp;                    // [tile][Cin + 1]
    const int nog = sdp / kPlinOB, nitems = sx * nog;    // item = (output group, input column), column fastest
    float acc[kPlinMaxJ][kPlinOB];
    FFNO_UNROLL
    for (int j = 0; j < kPlinMaxJ; ++j) {
        FFNO_UNROLL
        for (int u = 0; u < kPlinOB; ++u) acc[j][u] = 0.f;
    }
    const long pbeg = (long)blockIdx.x * per;
    const long pend = (pbeg + per) < P ? (pbeg + per) : P;
    for (long p0 = pbeg; p0 < pend; p0 += kPlinTile) {
        const int np = (int)((pend - p0) < kPlinTile ? (pend - p0) : kPlinTile);
        __syncthreads();
        for (int e = t; e < np * sdp; e += 256) {
            const int p = e / sdp, o = e - p * sdp;
            float v = 0.f;
            if (o < Cout) {
                const long idx = (p0 + p) * ldg + o;
                v = g[idx];
                if (act) v *= plin_dact(act[idx], act_mode);
            }
            ds[e] = v;
        }
        for (int e = t; e < np * sx; e += 256) {
            const int p = e / sx, i = e - p * sx;
            xs[e] = i < Cin ? x[(p0 + p) * ldx + i] : 1.f;
        }
        __syncthreads();
        FFNO_UNROLL
        for (int j = 0; j < kPlinMaxJ; ++j) {
            const int item = t + 256 * j;
            if (item < nitems) {
                const int og = item / sx, i = item - og * sx;
                const float* dr = ds + og * kPlinOB;
                const float* xr = xs + i;
                for (int p = 0; p < np; ++p) {
                    const float xv = xr[p * sx];
                    FFNO_UNROLL
                    for (int u = 0; u < kPlinOB; u += 4) {
                        const f32x4 d = *reinterpret_cast<const f32x4*>(dr + p * sdp + u);
                        acc[j][u] = fmaf(d[0], xv, acc[j][u]);
                        acc[j][u + 1] = fmaf(d[1], xv, acc[j][u + 1]);
                        acc[j][u + 2] = fmaf(d[2], xv, acc[j][u + 2]);
                        acc[j][u + 3] = fmaf(d[3], xv, acc[j][u + 3]);
                    }
                }
            }
        }
    }
    FFNO_UNROLL
    for (int j = 0; j < kPlinMaxJ; ++j) {
        const int item = t + 256 * j;
        if (item < nitems) {
            const int og = item / sx, i = item - og * sx;
            FFNO_UNROLL
            for (int u = 0; u < kPlinOB; ++u) {
                const int o = og * kPlinOB + u;
                if (o < Cout) part[(long)blockIdx.x * npairs + o * sx + i] = acc[j][u];
            }
        }
    }
}

// dW[o][i] (+)= sum_s part[s][o][i]: one 64-lane wave per (o, i) pair, the slices spread over the lanes
__global__ __launch_bounds__(256) void plin_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW,
                                                                float* __restrict__ db, int Cin, int Cout, int nsplit,
                                                                int accumulate) {
    const int sx = Cin + 1, npairs = Cout * sx;
    __shared__ float red[256];
    // block = 4 pairs x 64 slices-lanes; consecutive threads take consecutive pairs so the loads stay coalesced
    const int q = blockIdx.x * 4 + (threadIdx.x & 3);
    const int lane = threadIdx.x >> 2;
    float a = 0.f;
    if (q < npairs)
        for (int s = lane; s < nsplit; s += 64) a += part[(long)s * npairs + q];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w >= 4; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < 4 && q < npairs) {
        const int o = q / sx, i = q - o * sx;
        float* dst = i < Cin ? dW + o * Cin + i : (db ? db + o : nullptr);
        if (dst) *dst = accumulate ? *dst + red[threadIdx.x] : red[threadIdx.x];
    }
}

// plain [R][Cc][inner]  <->  padded [.][Cp][inner] (rows r < R, columns c < Cc of the padded tensor)
__global__ __launch_bounds__(256) void pad_copy_kernel(const ffno_pad_desc* __restrict__ descs, int to_padded) {
    const ffno_pad_desc d = descs[blockIdx.y];
    const long n = (long)d.R * d.Cc * d.inner;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const long k = e % d.inner;
        const long rc = e / d.inner;
        const long c = rc % d.Cc, r = rc / d.Cc;
        const long pidx = (r * d.Cp + c) * d.inner + k;
        if (to_padded)
            d.padded[pidx] = d.plain[e];
        else
            d.plain[e] = d.padded[pidx];
    }
}

static inline int plin_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFNO_OK : (int)e;
}

static inline unsigned plin_grid(long P) {
    const long ntile = (P + kPlinTile - 1) / kPlinTile;
    return (unsigned)(ntile < 2048 ? ntile : 2048);
}

}  // namespace ffno

using namespace ffno;

extern "C" {

int ffno_plin_supported(int Cin, int Cout) {
    return (Cin >= 1 && Cout >= 1 && Cin <= 128 && Cout <= 128 && (Cin + 1) * ((Cout + kPlinOB - 1) / kPlinOB) <= 256 * kPlinMaxJ) ? 1 : 0;
}

int ffno_plin_fwd(const float* x, int ldx, const float* W, const float* b, const float* add, float* out, int ldo,
                  const float* res, float* out2, float* pre_out, long P, int Cin, int Cout, int act_mode, void* stream) {
    if (!x || !W || !out || P <= 0 || ldx < Cin || ldo < Cout || (out2 && !res) || act_mode < 0 || act_mode > 2)
        return FFNO_EINVAL;
    if (!ffno_plin_supported(Cin, Cout)) return FFNO_EUNSUPPORTED;
    const bool vec = ldo % 4 == 0 && ((uintptr_t)out % 16 == 0) && (!add || (uintptr_t)add % 16 == 0) &&
                     (!res || (uintptr_t)res % 16 == 0) && (!out2 || (uintptr_t)out2 % 16 == 0) &&
                     (!pre_out || (uintptr_t)pre_out % 16 == 0);
    const int cp = (Cout + 3) / 4 * 4;
    const size_t xsz = ((size_t)kPlinTile * (Cin + 1) + 3) / 4 * 4;
    const size_t lds = sizeof(float) * (xsz + (vec ? (size_t)Cin * cp + cp : (size_t)Cout * (Cin + 1) + Cout));
    if (vec)
        FFNO_LAUNCH(plin_fwd_kernel<true>, dim3(plin_grid(P)), dim3(256), lds, (hipStream_t)stream, x, ldx, W, b, add, out, ldo,
                    res, out2, pre_out, P, Cin, Cout, act_mode);
    else
        FFNO_LAUNCH(plin_fwd_kernel<false>, dim3(plin_grid(P)), dim3(256), lds, (hipStream_t)stream, x, ldx, W, b, add, out, ldo,
                    res, out2, pre_out, P, Cin, Cout, act_mode);
    return plin_status();
}

int ffno_plin_bwd_data(const float* g, int ldg, const float* act, const float* W, float* dx, int ldx, float* dpre_out, long P,
                       int Cin, int Cout, int accumulate, int act_mode, void* stream) {
    if (!g || !W || !dx || P <= 0 || ldg < Cout || ldx < Cin || act_mode < 0 || act_mode > 2) return FFNO_EINVAL;
    if (!ffno_plin_supported(Cin, Cout)) return FFNO_EUNSUPPORTED;
    const bool vec = ldx % 4 == 0 && Cin % 4 == 0 && (uintptr_t)dx % 16 == 0;
    const size_t lds = sizeof(float) * (((size_t)Cout * Cin + 3) / 4 * 4 + (size_t)kPlinTile * (Cout + 1));
    if (vec)
        FFNO_LAUNCH(plin_bwd_data_kernel<true>, dim3(plin_grid(P)), dim3(256), lds, (hipStream_t)stream, g, ldg, act, W, dx, ldx,
                    dpre_out, P, Cin, Cout, accumulate, act_mode);
    else
        FFNO_LAUNCH(plin_bwd_data_kernel<false>, dim3(plin_grid(P)), dim3(256), lds, (hipStream_t)stream, g, ldg, act, W, dx, ldx,
                    dpre_out, P, Cin, Cout, accumulate, act_mode);
    return plin_status();
}

int ffno_plin_wgrad_nsplit(long P) {      // two 64-point tiles per slice, all slices co-resident (41 KB of LDS each)
    const long n = (P + 127) / 128;
    return (int)(n < 1 ? 1 : (n > 1024 ? 1024 : n));
}

size_t ffno_plin_wgrad_partial_floats(long P, int Cin, int Cout) {
    return (size_t)ffno_plin_wgrad_nsplit(P) * Cout * (Cin + 1);
}

int ffno_plin_bwd_weights(const float* g, int ldg, const float* act, const float* x, int ldx, float* part, float* dW, float* db,
                          long P, int Cin, int Cout, int accumulate, int act_mode, void* stream) {
    if (!g || !x || !part || !dW || P <= 0 || ldg < Cout || ldx < Cin || act_mode < 0 || act_mode > 2) return FFNO_EINVAL;
    if (!ffno_plin_supported(Cin, Cout)) return FFNO_EUNSUPPORTED;
    const int nsplit = ffno_plin_wgrad_nsplit(P);
    const long per = ((P + nsplit - 1) / nsplit + kPlinTile - 1) / kPlinTile * kPlinTile;
    const int sdp = (Cout + kPlinOB - 1) / kPlinOB * kPlinOB;
    const size_t lds = sizeof(float) * ((size_t)kPlinTile * sdp + (size_t)kPlinTile * (Cin + 1));
    FFNO_LAUNCH(plin_wgrad_partial_kernel, dim3(nsplit), dim3(256), lds, (hipStream_t)stream, g, ldg, act, x, ldx, part, P, Cin,
                Cout, per, act_mode);
    int rc = plin_status();
    if (rc) return rc;
    const int npairs = Cout * (Cin + 1);
    FFNO_LAUNCH(plin_wgrad_reduce_kernel, dim3((npairs + 3) / 4), dim3(256), 0, (hipStream_t)stream, part, dW, db, Cin, Cout,
                nsplit, accumulate);
    return plin_status();
}

int ffno_pad_copy(const ffno_pad_desc* descs, int n, int to_padded, void* stream) {
    if (!descs || n <= 0) return FFNO_EINVAL;
    FFNO_LAUNCH(pad_copy_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, descs, to_padded);
    return plin_status();
}

}  // extern "C"
