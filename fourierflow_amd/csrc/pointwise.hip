// Small bandwidth-bound kernels around the spectral/FF hot loop (fp32, VALU): weight-norm,
// lift (in_proj), output head, relative-L2 loss, fused flat AdamW.  Reference citations per entry
// point are in include/ffno.h.
#include "ffno_device.h"
#include "ffno.h"

#include <math.h>

namespace ffno {

static inline int pw_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFNO_OK : (int)e;
}

// ---- weight norm ------------------------------------------------------------------------------------
// one wave per weight row; descriptor table selects the matrix (blockIdx.y)
__global__ __launch_bounds__(256) void wn_fwd_kernel(const ffno_wn_desc* __restrict__ descs) {
    const ffno_wn_desc d = descs[blockIdx.y];
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.rows) return;
    const float* v = d.v + (long)row * d.cols;
    float ss = 0.f;
    for (int c = lane; c < d.cols; c += 64) ss += v[c] * v[c];
    ss = wave_sum(ss);
    const float scale = d.g[row] / sqrtf(ss);
    float* w = d.w + (long)row * d.cols;
    for (int c = lane; c < d.cols; c += 64) w[c] = v[c] * scale;
}

__global__ __launch_bounds__(256) void wn_bwd_kernel(const ffno_wn_desc* __restrict__ descs) {
    const ffno_wn_desc d = descs[blockIdx.y];
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.rows) return;
    const float* v = d.v + (long)row * d.cols;
    const float* dw = d.dw + (long)row * d.cols;
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < d.cols; c += 64) {
        ss += v[c] * v[c];
        dot += dw[c] * v[c];
    }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    const float inv = 1.f / sqrtf(ss);
    const float dg = dot * inv;
    const float g = d.g[row];
    if (lane == 0) d.dg[row] = dg;
    float* dv = d.dv + (long)row * d.cols;
    for (int c = lane; c < d.cols; c += 64) dv[c] = g * inv * (dw[c] - dg * inv * v[c]);
}

// ---- batched transpose (effective weights -> transposed copies for the backward chain kernel) --------
__global__ __launch_bounds__(256) void transpose_batched_kernel(const ffno_tr_desc* __restrict__ descs) {
    __shared__ float tile[32][33];
    const ffno_tr_desc d = descs[blockIdx.z];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    if (c0 >= d.cols || r0 >= d.rows) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int yy = ty; yy < 32; yy += 8)
        if (r0 + yy < d.rows && c0 + tx < d.cols) tile[yy][tx] = d.src[(long)(r0 + yy) * d.cols + c0 + tx];
    __syncthreads();
    for (int yy = ty; yy < 32; yy += 8)
        if (c0 + yy < d.cols && r0 + tx < d.rows) d.dst[(long)(c0 + yy) * d.rows + r0 + tx] = tile[tx][yy];
}

// ---- pad / crop index map -------------------------------------------------------------------------------
// The 3-D mesh operator zero-pads the lifted features by 8 cells at the END of every spatial axis and crops
// the last layer's output again (reference mesh_3d.py:165,173).  Instead of materialising pad / crop copies,
// the lift and head kernels address the padded activation buffer directly: unpadded pixel p -> padded pixel q.
struct PadMapDev {
    int s1, s2;          // unpadded sizes of the two inner spatial axes (row-major [b][s0][s1][s2])
    int p0, p1, p2;      // padded sizes of the three spatial axes
    int s0;
    int enabled;
    __device__ __forceinline__ long map(long p) const {
        if (!enabled) return p;
        const long z = p % s2, t = p / s2;
        const long y = t % s1, t2 = t / s1;
        const long x = t2 % s0, b = t2 / s0;
        return ((b * p0 + x) * p1 + y) * p2 + z;
    }
};

static inline PadMapDev make_padmap(const ffno_padmap* pm) {
    PadMapDev d;
    d.enabled = 0;
    d.s0 = d.s1 = d.s2 = d.p0 = d.p1 = d.p2 = 1;
    if (pm) {
        d.s0 = pm->size[0];
        d.s1 = pm->size[1];
        d.s2 = pm->size[2];
        d.p0 = pm->padded[0];
        d.p1 = pm->padded[1];
        d.p2 = pm->padded[2];
        d.enabled = (d.s0 != d.p0 || d.s1 != d.p1 || d.s2 != d.p2) ? 1 : 0;
    }
    return d;
}

// ---- lift (in_proj) -----------------------------------------------------------------------------------
// (ST: storage format of the activation tensor -- `out` here; ffno_device.h)
// One workgroup per CU at most (grid <= 256 x 512 threads; 1024 threads measured equal: 12.4 vs 12.7 us): the range word of the lifted features costs one same-address atomic per
// workgroup, and those serialise at ~10 ns each (tools/ubench/atomic_fold.hip: 2048 workgroups = 21 us of tail behind a 10-us body --
// the round-5 form of this kernel; 256 = 0.6-2 us).  Two pixels per thread and pass: both pixels' inputs are requested before either
// is used.
template <int C, class ST = StF32>
__global__ __launch_bounds__(512) void lift_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                       const float* __restrict__ b, typename ST::T* __restrict__ out, int P,
                                                       int Cin, PadMapDev pm, unsigned* out_amax) {
    FFNO_DYN_SMEM(smem);
    __shared__ float rfold[16];
    float omax = 0.f;
    float* Wt = reinterpret_cast<float*>(smem);  // [Cin + 1][C], last row = bias
    for (int e = threadIdx.x; e < Cin * C; e += blockDim.x) Wt[(e % Cin) * C + (e / Cin)] = W[e];
    for (int e = threadIdx.x; e < C; e += blockDim.x) Wt[Cin * C + e] = b[e];
    __syncthreads();
    constexpr int LPP = C / 4;        // lanes per pixel (one float4 of outputs each)
    const int PPB = (int)blockDim.x / LPP;    // pixels per block pass
    const int c4 = (threadIdx.x % LPP) * 4, pl = threadIdx.x / LPP;
    const float4 bias = *reinterpret_cast<const float4*>(Wt + Cin * C + c4);
    const long stride = (long)gridDim.x * PPB;
    for (long p = (long)blockIdx.x * PPB + pl; p < P; p += 2 * stride) {
        const long p1 = p + stride;
        const bool two = p1 < P;
        float4 acc0 = bias, acc1 = bias;
        const float* xp0 = x + p * Cin;
        const float* xp1 = x + (two ? p1 : p) * Cin;
        for (int i = 0; i < Cin; ++i) {
            const float xv0 = xp0[i], xv1 = xp1[i];
            const float4 w = *reinterpret_cast<const float4*>(Wt + i * C + c4);
            acc0.x = fmaf(xv0, w.x, acc0.x);
            acc0.y = fmaf(xv0, w.y, acc0.y);
            acc0.z = fmaf(xv0, w.z, acc0.z);
            acc0.w = fmaf(xv0, w.w, acc0.w);
            acc1.x = fmaf(xv1, w.x, acc1.x);
            acc1.y = fmaf(xv1, w.y, acc1.y);
            acc1.z = fmaf(xv1, w.z, acc1.z);
            acc1.w = fmaf(xv1, w.w, acc1.w);
        }
        ST::st4(out + pm.map(p) * C + c4, acc0);
        acc0 = st_rnd4<ST>(acc0);
        omax = fmaxf(fmaxf(omax, fmaxf(fabsf(acc0.x), fabsf(acc0.y))), fmaxf(fabsf(acc0.z), fabsf(acc0.w)));
        if (two) {
            ST::st4(out + pm.map(p1) * C + c4, acc1);
            acc1 = st_rnd4<ST>(acc1);
            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(acc1.x), fabsf(acc1.y))), fmaxf(fabsf(acc1.z), fabsf(acc1.w)));
        }
    }
    if (out_amax) range_fold(omax, rfold, (int)blockDim.x >> 6, out_amax);      // (optional range word of the lifted features)
}

// partial[split][i][c] = sum_{p in slice} gout[q(p)][c] * (i < Cin ? x[p][i] : 1)
template <int C, class ST = StF32>
__global__ __launch_bounds__(256) void lift_bwd_partial_kernel(const float* __restrict__ x,
                                                               const typename ST::T* __restrict__ gout,
                                                               const typename ST::T* __restrict__ gout2,
                                                               float* __restrict__ partial, int P, int Cin,
                                                               int chunk, PadMapDev pm) {
    constexpr int TP = 32;                     // pixels staged per pass
    constexpr int MAXU = (C * 64) / 256;       // pairs per thread for Cin + 1 <= 64
    __shared__ __attribute__((aligned(16))) float gs[TP * C];
    __shared__ float xs[TP * 64];
    const int npairs = C * (Cin + 1);
    float acc[MAXU];
    FFNO_UNROLL
    for (int u = 0; u < MAXU; ++u) acc[u] = 0.f;
    const long pbeg = (long)blockIdx.x * chunk, pend = min((long)P, pbeg + chunk);
    for (long p0 = pbeg; p0 < pend; p0 += TP) {
        const int np = (int)min((long)TP, pend - p0);
        __syncthreads();
        // (gout2: the second gradient buffer of a paired adjoint launch, added while the rows are staged -- the sum the engine used to
        //  form with an axpy pass over both images before this kernel)
        // (four channels per thread and load: 16-byte requests instead of the 4-byte ones of rounds 1-5)
        for (int e4 = threadIdx.x; e4 < TP * C / 4; e4 += 256) {
            const int e = 4 * e4;
            float4 gvv = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((e / C) < np) {
                const long off = pm.map(p0 + e / C) * C + (e % C);
                gvv = ST::ld4(gout + off);
                if (gout2) {
                    const float4 g2 = ST::ld4(gout2 + off);
                    gvv = make_float4(ST::rnd(gvv.x + g2.x), ST::rnd(gvv.y + g2.y), ST::rnd(gvv.z + g2.z), ST::rnd(gvv.w + g2.w));
                }
            }
            *reinterpret_cast<float4*>(gs + e) = gvv;
        }
        for (int e = threadIdx.x; e < TP * (Cin + 1); e += 256) {
            const int pp = e / (Cin + 1), i = e % (Cin + 1);
            xs[pp * 64 + i] = (pp < np) ? (i < Cin ? x[(p0 + pp) * Cin + i] : 1.f) : 0.f;
        }
        __syncthreads();
        FFNO_UNROLL
        for (int u = 0; u < MAXU; ++u) {
            const int e = u * 256 + threadIdx.x;
            if (e < npairs) {
                const int c = e % C, i = e / C;
                float a = acc[u];
                for (int pp = 0; pp < TP; ++pp) a = fmaf(gs[pp * C + c], xs[pp * 64 + i], a);
                acc[u] = a;
            }
        }
    }
    FFNO_UNROLL
    for (int u = 0; u < MAXU; ++u) {
        const int e = u * 256 + threadIdx.x;
        if (e < npairs) partial[(long)blockIdx.x * npairs + e] = acc[u];
    }
}

// dx[p][i] = sum_c gout[q(p)][c] W[c][i]: the data gradient of the lift (what autograd hands to whatever produced the block's
// input).  One thread per pixel; W^T staged in LDS ([Cin][C], read as float4 broadcasts).
template <int C>
__global__ __launch_bounds__(256) void lift_bwd_data_kernel(const float* __restrict__ gout, const float* __restrict__ W,
                                                            float* __restrict__ dx, int P, int Cin, PadMapDev pm) {
    FFNO_DYN_SMEM(smem);
    float* Wt = reinterpret_cast<float*>(smem);  // [Cin][C]
    for (int e = threadIdx.x; e < Cin * C; e += blockDim.x) Wt[(e % Cin) * C + (e / Cin)] = W[e];
    __syncthreads();
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
        const float* g = gout + pm.map(p) * C;
        float4 gv[C / 4];
        FFNO_UNROLL
        for (int q = 0; q < C / 4; ++q) gv[q] = *reinterpret_cast<const float4*>(g + 4 * q);
        for (int i = 0; i < Cin; ++i) {
            float a = 0.f;
            FFNO_UNROLL
            for (int q = 0; q < C / 4; ++q) {
                const float4 w = *reinterpret_cast<const float4*>(Wt + i * C + 4 * q);
                a = fmaf(gv[q].x, w.x, a), a = fmaf(gv[q].y, w.y, a), a = fmaf(gv[q].z, w.z, a), a = fmaf(gv[q].w, w.w, a);
            }
            dx[p * Cin + i] = a;
        }
    }
}

// 4 (input, channel) pairs per block, the slices spread over 64 lanes each (the serial version spent 60 us on 256 slices)
__global__ __launch_bounds__(256) void lift_bwd_reduce_kernel(const float* __restrict__ partial, float* dW, float* db,
                                                              int Cin, int C, int nsplit, int accumulate) {
    __shared__ float red[256];
    const int npairs = C * (Cin + 1);
    const int e = blockIdx.x * 4 + (threadIdx.x & 3), lane = threadIdx.x >> 2;
    float a = 0.f;
    if (e < npairs)
        for (int sp = lane; sp < nsplit; sp += 64) a += partial[(long)sp * npairs + e];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w >= 4; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < 4 && e < npairs) {
        const int c = e % C, i = e / C;
        float* dst = (i < Cin) ? (dW + c * Cin + i) : (db + c);
        *dst = accumulate ? (*dst + red[threadIdx.x]) : red[threadIdx.x];
    }
}

// ---- output head ----------------------------------------------------------------------------------------
// y[p][o] = (b[p] Wa^T + ca) Wb[o]^T + cb[o]  folded to  y[p][o] = b[p] . weff[o] + beff[o];  fold[o][C+1].
static constexpr int kHeadMaxOut = 8;

// one 64-lane wave per folded entry: the D-long dot product is spread over the lanes
__global__ __launch_bounds__(256) void head_fold_kernel(const float* __restrict__ Wa, const float* __restrict__ ca,
                                                        const float* __restrict__ Wb, const float* __restrict__ cb,
                                                        float* fold, int C, int D, int O) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= O * (C + 1)) return;
    const int o = e / (C + 1), c = e % (C + 1);
    float s = 0.f;
    for (int jd = lane; jd < D; jd += 64) s = fmaf(Wb[o * D + jd], (c == C) ? ca[jd] : Wa[jd * C + c], s);
    FFNO_UNROLL
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) fold[e] = s + ((c == C) ? cb[o] : 0.f);
}

template <int C, class ST = StF32>
__global__ __launch_bounds__(256) void head_fwd_kernel(const typename ST::T* __restrict__ b, const float* __restrict__ fold,
                                                       float* y, int P, int O, int accumulate, PadMapDev pm) {
    constexpr int LPP = C / 4, PPB = 256 / LPP;
    const int c4 = (threadIdx.x % LPP) * 4, pl = threadIdx.x / LPP;
    const long npass = ((long)P + PPB - 1) / PPB;
    for (long pass = blockIdx.x; pass < npass; pass += gridDim.x) {  // uniform trip count per wave (shuffles)
        const long p = pass * PPB + pl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < P) v = ST::ld4(b + pm.map(p) * C + c4);
        for (int o = 0; o < O; ++o) {
            const float4 w = *reinterpret_cast<const float4*>(fold + o * (C + 1) + c4);
            float d = v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w;
            FFNO_UNROLL
            for (int m = LPP / 2; m >= 1; m >>= 1) d += __shfl_xor(d, m);
            if (p < P && c4 == 0) y[p * O + o] = (accumulate ? y[p * O + o] : 0.f) + d + fold[o * (C + 1) + C];
        }
    }
}

// gb[q(p)][:] = sum_o gy[p][o] * weff[o] ;  partial[block][o][0..C) = sum_p gy[p][o] b[q(p)][:],  [o][C] = sum_p gy[p][o]
// (256 or 1024 threads per workgroup: the slice count -- one same-address atomic each for the range word, ~10 ns apiece -- stays at
//  <= 256, and a launch over many pixels gets its memory-level parallelism from 16 waves per CU instead of 4: every pass of a wave
//  is one dependent load -> store round)
template <int C, class ST = StF32>
__global__ __launch_bounds__(1024) void head_bwd_kernel(const typename ST::T* __restrict__ b, const float* __restrict__ gy,
                                                        const float* __restrict__ fold, typename ST::T* __restrict__ gb,
                                                        float* __restrict__ partial, int P, int O, PadMapDev pm,
                                                        unsigned* gb_amax) {
    constexpr int LPP = C / 4;
    const int PPB = (int)blockDim.x / LPP;
    __shared__ float red[(1024 / LPP) * (C + 4)];
    __shared__ float rfold[16];
    float omax = 0.f;
    const int l = threadIdx.x % LPP, c4 = l * 4, pl = threadIdx.x / LPP;
    float4 acc[kHeadMaxOut];
    float sg[kHeadMaxOut];
    FFNO_UNROLL
    for (int o = 0; o < kHeadMaxOut; ++o) {
        acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
        sg[o] = 0.f;
    }
    for (long p = (long)blockIdx.x * PPB + pl; p < P; p += (long)gridDim.x * PPB) {
        const long q = pm.map(p);
        const float4 v = ST::ld4(b + q * C + c4);
        float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);
        FFNO_UNROLL
        for (int o = 0; o < kHeadMaxOut; ++o) {
            if (o < O) {
                const float g = gy[p * O + o];
                const float4 w = *reinterpret_cast<const float4*>(fold + o * (C + 1) + c4);
                acc[o].x = fmaf(g, v.x, acc[o].x);
                acc[o].y = fmaf(g, v.y, acc[o].y);
                acc[o].z = fmaf(g, v.z, acc[o].z);
                acc[o].w = fmaf(g, v.w, acc[o].w);
                sg[o] += g;
                gsum.x = fmaf(g, w.x, gsum.x);
                gsum.y = fmaf(g, w.y, gsum.y);
                gsum.z = fmaf(g, w.z, gsum.z);
                gsum.w = fmaf(g, w.w, gsum.w);
            }
        }
        if (gb) ST::st4(gb + q * C + c4, gsum);
        gsum = st_rnd4<ST>(gsum);
        omax = fmaxf(fmaxf(omax, fmaxf(fabsf(gsum.x), fabsf(gsum.y))), fmaxf(fabsf(gsum.z), fabsf(gsum.w)));
    }
    if (gb && gb_amax) range_fold(omax, rfold, (int)blockDim.x >> 6, gb_amax);      // (optional range word of the gradient handed to the layers)
    for (int o = 0; o < O; ++o) {
        __syncthreads();
        float* r = red + pl * (C + 4);
        FFNO_UNROLL
        for (int oo = 0; oo < kHeadMaxOut; ++oo) {
            if (oo == o) {
                r[c4] = acc[oo].x;
                r[c4 + 1] = acc[oo].y;
                r[c4 + 2] = acc[oo].z;
                r[c4 + 3] = acc[oo].w;
                if (l == 0) r[C] = sg[oo];
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e <= C; e += blockDim.x) {
            float s = 0.f;
            for (int qq = 0; qq < PPB; ++qq) s += red[qq * (C + 4) + e];
            partial[((long)blockIdx.x * O + o) * (C + 1) + e] = s;
        }
    }
}

__global__ __launch_bounds__(256) void head_bwd_reduce_kernel(const float* __restrict__ partial, float* red, int C, int O,
                                                              int nsplit) {
    __shared__ float sm[256];
    const int n = O * (C + 1);
    const int e = blockIdx.x * 4 + (threadIdx.x & 3), lane = threadIdx.x >> 2;
    float a = 0.f;
    if (e < n)
        for (int sp = lane; sp < nsplit; sp += 64) a += partial[(long)sp * n + e];
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w >= 4; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < 4 && e < n) red[e] = sm[threadIdx.x];
}

// y = Wb (Wa b + ca) + cb, red[o] = { G_o = sum_p gy[p][o] b[p][:], S_o = sum_p gy[p][o] }:
//   dWb[o][j] = (Wa G_o)[j] + ca[j] S_o ; dWa[j][c] = sum_o Wb[o][j] G_o[c] ; dca[j] = sum_o Wb[o][j] S_o ; dcb[o] = S_o
// (one flat index space over the four outputs, grid-stride: the single-workgroup form of rounds 1-5 walked its 8k entries of dWa
//  in 32 dependent rounds -- 13.5 us for 40 kB of results)
__global__ __launch_bounds__(256) void head_param_grads_kernel(const float* __restrict__ red, const float* __restrict__ Wa,
                                                               const float* __restrict__ ca, const float* __restrict__ Wb,
                                                               float* dWa, float* dca, float* dWb, float* dcb, int C, int D,
                                                               int O, int accumulate) {
    const int n0 = O * D, n1 = n0 + D, n2 = n1 + D * C, n3 = n2 + O;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n3; idx += gridDim.x * blockDim.x) {
        if (idx < n0) {
            const int e = idx, o = e / D, jd = e % D;
            float s = ca[jd] * red[o * (C + 1) + C];
            for (int c = 0; c < C; ++c) s = fmaf(Wa[jd * C + c], red[o * (C + 1) + c], s);
            dWb[e] = accumulate ? dWb[e] + s : s;
        } else if (idx < n1) {
            const int jd = idx - n0;
            float t = 0.f;
            for (int o = 0; o < O; ++o) t = fmaf(Wb[o * D + jd], red[o * (C + 1) + C], t);
            dca[jd] = accumulate ? dca[jd] + t : t;
        } else if (idx < n2) {
            const int e = idx - n1;
            float t = 0.f;
            for (int o = 0; o < O; ++o) t = fmaf(Wb[o * D + e / C], red[o * (C + 1) + e % C], t);
            dWa[e] = accumulate ? dWa[e] + t : t;
        } else {
            const int o = idx - n2;
            const float S = red[o * (C + 1) + C];
            dcb[o] = accumulate ? dcb[o] + S : S;
        }
    }
}

// ---- markov feature build + running normaliser ---------------------------------------------------------
// Replaces Grid2DMarkovExperiment._build_features (routines/grid_2d_markov.py:124-170, use_position=True,
// no velocity/force/mu) + Normalizer (modules/normalizer.py:18-77):
//   raw[p] = [ x[p][0..Cx) , low + (high-low)*m/(M-1) , low + (high-low)*n/(N-1) ]      (linspace, ij meshgrid)
//   accumulate: sum[c] += sum_p raw, sum_squared[c] += sum_p raw^2, count += P          (deterministic 2-stage)
//   finalize  : mean = sum/max(count,1) ; std = max(sqrt(sum_squared/max(count,1) - mean^2), eps)
//   apply     : feat[p][c] = (raw[p][c] - mean[c]) / std[c] + noise[p][c] * noise_std
// channel order of grid_2d_markov.py:146-163: x (and velocity) | position (use_position) | force map | viscosity mu
struct MarkovSrc {
    const float* x;
    const float* force;   // [B][M][N] or null   (append_force, :156-158)
    const float* mu;      // [B] or null         (append_mu, :160-162)
    int Cx, use_pos;
};

__device__ __forceinline__ float markov_raw(const MarkovSrc& src, long p, int c, int M, int N, float low, float high) {
    if (c < src.Cx) return src.x[p * src.Cx + c];
    c -= src.Cx;
    if (src.use_pos) {
        if (c < 2) {
            const int n = (int)(p % N), m = (int)((p / N) % M);
            const int idx = c == 0 ? m : n, size = c == 0 ? M : N;
            return size > 1 ? low + (high - low) * (float)idx / (float)(size - 1) : low;
        }
        c -= 2;
    }
    if (src.force) {
        if (c == 0) return src.force[p];
        c -= 1;
    }
    return src.mu[p / ((long)M * N)];
}

__global__ __launch_bounds__(256) void markov_stats_partial_kernel(MarkovSrc src, float* __restrict__ partial,
                                                                   long P, int D, int M, int N, float low,
                                                                   float high, int chunk) {
    __shared__ float red[2][4][16];
    const long pbeg = (long)blockIdx.x * chunk, pend = min(P, pbeg + chunk);
    float s[16], q[16];
    FFNO_UNROLL
    for (int c = 0; c < 16; ++c) s[c] = q[c] = 0.f;
    for (long p = pbeg + threadIdx.x; p < pend; p += 256) {
        FFNO_UNROLL
        for (int c = 0; c < 16; ++c) {
            if (c < D) {
                const float v = markov_raw(src, p, c, M, N, low, high);
                s[c] += v;
                q[c] = fmaf(v, v, q[c]);
            }
        }
    }
    FFNO_UNROLL
    for (int c = 0; c < 16; ++c) {
        s[c] = wave_sum(s[c]);
        q[c] = wave_sum(q[c]);
        if ((threadIdx.x & 63) == 0) {
            red[0][threadIdx.x >> 6][c] = s[c];
            red[1][threadIdx.x >> 6][c] = q[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int which = threadIdx.x >> 4, c = threadIdx.x & 15;
        partial[((long)blockIdx.x * 2 + which) * 16 + c] = red[which][0][c] + red[which][1][c] + red[which][2][c] + red[which][3][c];
    }
}

// state[0..D) sum, [D..2D) sum_squared, [2D] count, [2D+1] n_accumulations ; derived[0..D) mean, [D..2D) std
__global__ void markov_stats_finalize_kernel(const float* __restrict__ partial, float* state, float* derived, int D,
                                             int nsplit, float count_add, float eps, int accumulate) {
    const int c = threadIdx.x;
    if (c < D) {
        if (accumulate) {
            float s = 0.f, q = 0.f;
            for (int sp = 0; sp < nsplit; ++sp) {
                s += partial[((long)sp * 2 + 0) * 16 + c];
                q += partial[((long)sp * 2 + 1) * 16 + c];
            }
            state[c] += s;
            state[D + c] += q;
        }
        const float cnt = fmaxf(state[2 * D] + (accumulate ? count_add : 0.f), 1.f);
        const float mean = state[c] / cnt;
        derived[c] = mean;
        derived[D + c] = fmaxf(sqrtf(state[D + c] / cnt - mean * mean), eps);
    }
    __syncthreads();
    if (c == 0 && accumulate) {
        state[2 * D] += count_add;
        state[2 * D + 1] += 1.f;
    }
}

__global__ __launch_bounds__(256) void markov_features_kernel(MarkovSrc src, const float* __restrict__ derived,
                                                              const float* __restrict__ noise, float* __restrict__ out,
                                                              long P, int D, int M, int N, float low, float high,
                                                              float noise_std, int normalize) {
    const long total = P * D;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long p = e / D;
        const int c = (int)(e % D);
        float v = markov_raw(src, p, c, M, N, low, high);
        if (normalize) v = (v - derived[c]) / derived[D + c];
        if (noise) v = fmaf(noise[e], noise_std, v);
        out[e] = v;
    }
}

// ---- relative L2 loss -----------------------------------------------------------------------------------
// Two deterministic stages: block (slice, sample) -> partial sums tmp[(sample*S + slice)*2 + {0,1}]; every consumer adds
// the S partials of a sample in the same order.
__global__ __launch_bounds__(256) void lploss_reduce_kernel(const float* __restrict__ pred,
                                                            const float* __restrict__ target, float* tmp, int n,
                                                            const float* __restrict__ affine) {
    __shared__ float sd[4], sy[4];
    const int bidx = blockIdx.y, S = gridDim.x;
    const int chunk = (n + S - 1) / S;
    const int beg = blockIdx.x * chunk, end = min(n, beg + chunk);
    const float* p = pred + (long)bidx * n;
    const float* t = target + (long)bidx * n;
    const float sc = affine ? affine[0] : 1.f, sh = affine ? affine[1] : 0.f;   // pred * std + mean (Normalizer.inverse)
    float d2 = 0.f, y2 = 0.f;
    for (int i = beg + threadIdx.x; i < end; i += 256) {
        const float d = fmaf(p[i], sc, sh) - t[i];
        d2 = fmaf(d, d, d2);
        y2 = fmaf(t[i], t[i], y2);
    }
    d2 = wave_sum(d2);
    y2 = wave_sum(y2);
    if ((threadIdx.x & 63) == 0) {
        sd[threadIdx.x >> 6] = d2;
        sy[threadIdx.x >> 6] = y2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        tmp[2 * (bidx * S + blockIdx.x)] = sd[0] + sd[1] + sd[2] + sd[3];
        tmp[2 * (bidx * S + blockIdx.x) + 1] = sy[0] + sy[1] + sy[2] + sy[3];
    }
}

__global__ __launch_bounds__(256) void lploss_grad_kernel(const float* __restrict__ pred,
                                                          const float* __restrict__ target,
                                                          const float* __restrict__ tmp, float* loss_out,
                                                          float* gpred, int B, int n, int S, float gscale,
                                                          const float* __restrict__ affine) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && loss_out) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            float d2 = 0.f, y2 = 0.f;
            for (int k = 0; k < S; ++k) d2 += tmp[2 * (b * S + k)], y2 += tmp[2 * (b * S + k) + 1];
            s += sqrtf(d2) / sqrtf(y2);
        }
        loss_out[0] = s / (float)B;
    }
    if (!gpred) return;
    const int bidx = blockIdx.y;
    float d2 = 0.f, y2 = 0.f;
    for (int k = 0; k < S; ++k) d2 += tmp[2 * (bidx * S + k)], y2 += tmp[2 * (bidx * S + k) + 1];
    const float dn = sqrtf(d2), yn = sqrtf(y2);
    const float sc = affine ? affine[0] : 1.f, sh = affine ? affine[1] : 0.f;
    const float coef = dn > 0.f ? gscale * sc / ((float)B * dn * yn) : 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const long e = (long)bidx * n + i;
        gpred[e] = coef * (fmaf(pred[e], sc, sh) - target[e]);
    }
}

// ---- fused flat AdamW ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1,
                                                    float bc2_sqrt, float gscale, int decoupled) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        // decoupled = 1: torch.optim.AdamW (p *= 1 - lr wd);  0: torch.optim.Adam (L2: the decay joins the gradient)
        const float gi = decoupled ? g[i] * gscale : fmaf(wd, p[i], g[i] * gscale);
        float pi = decoupled ? p[i] * (1.f - lr * wd) : p[i];
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi;
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha,
                                                   size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = fmaf(alpha, x[i], y[i]);
}

}  // namespace ffno

using namespace ffno;

// one tensor -> its range word (ffno_device.h "range words"): grid-stride maximum, one atomic per workgroup
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, size_t n, unsigned* word) {
    __shared__ float red[4];
    float m = 0.f;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    range_fold(m, red, 4, word);
}
extern "C" int ffno_amax(const float* x, size_t n, uint32_t* word, void* stream) {
    if (!x || !word || n == 0 || (reinterpret_cast<uintptr_t>(x) & 15u)) return FFNO_EINVAL;
    const int blocks = (int)std::min<size_t>((size_t)4 * device_cu_count(), (n / 4 + 255) / 256 + 1);
    FFNO_LAUNCH(amax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, word);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFNO_OK : (int)e;
}

// the same for n tensors in ONE launch (descs is a DEVICE array): blockIdx.y = tensor, all folded into one word
struct AmaxDesc {
    const float* x;
    size_t n;
};
// 16-byte loads where the tensor allows them; a workgroup whose first element lies past the end of ITS tensor leaves at once (the
// grid is sized for the longest tensor), and a zero maximum is not folded: every fold is a same-address atomic that costs ~10 ns
// behind all the others (tools/ubench/atomic_fold.hip; the 37 tensors x 64 workgroups of the 3-D model were 30 us of atomics)
__global__ __launch_bounds__(256) void amax_batched_kernel(const AmaxDesc* __restrict__ descs, unsigned* word) {
    __shared__ float red[4];
    const AmaxDesc d = descs[blockIdx.y];
    // this tensor is walked by its first `act` workgroups only (>= 1024 elements each per pass)
    const size_t act = min((size_t)gridDim.x, (d.n + 1023) / 1024);
    if (blockIdx.x >= act) return;
    float m = 0.f;
    const bool al = (reinterpret_cast<uintptr_t>(d.x) & 15u) == 0;
    const size_t n4 = al ? d.n / 4 : 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += act * 256) {
        const float4 v = reinterpret_cast<const float4*>(d.x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < d.n; i += act * 256) m = fmaxf(m, fabsf(d.x[i]));
    range_fold(m, red, 4, word);
}
extern "C" int ffno_amax_batched(const ffno_amax_desc* descs_dev, int n, size_t max_n, uint32_t* word, void* stream) {
    static_assert(sizeof(ffno_amax_desc) == sizeof(AmaxDesc), "descriptor layout");
    if (!descs_dev || !word || n <= 0 || max_n == 0) return FFNO_EINVAL;
    const int blocks = (int)std::min<size_t>(128, (max_n + 2047) / 2048);
    FFNO_LAUNCH(amax_batched_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream,
                reinterpret_cast<const AmaxDesc*>(descs_dev), word);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFNO_OK : (int)e;
}

extern "C" const char* ffno_build_target(void) {
    return FFNO_BUILD_TARGET;
}
extern "C" int ffno_abi_version(void) { return FFNO_ABI_VERSION; }

extern "C" int ffno_twiddle_fill_host(float* tw_host, int L) {
    if (!tw_host || L <= 0) return FFNO_EINVAL;
    const double inv = 1.0 / sqrt((double)L);
    for (int jx = 0; jx < L; ++jx) {
        const double th = 2.0 * M_PI * (double)jx / (double)L;
        double c = cos(th), s = sin(th);
        // exact zeros / +-1 at the quarter points so DC / Nyquist imaginary parts vanish identically
        if (4 * jx == L || 4 * jx == 3 * L) c = 0.0;
        if (jx == 0 || 2 * jx == L) s = 0.0;
        tw_host[jx] = (float)(c * inv);
        tw_host[L + jx] = (float)(s * inv);
    }
    return FFNO_OK;
}

extern "C" int ffno_weightnorm_fwd(const ffno_wn_desc* descs_dev, int n, int max_rows, void* stream) {
    if (!descs_dev || n <= 0 || max_rows <= 0) return FFNO_EINVAL;
    FFNO_LAUNCH(wn_fwd_kernel, dim3((max_rows + 3) / 4, n), dim3(256), 0, (hipStream_t)stream, descs_dev);
    return pw_status();
}
extern "C" int ffno_weightnorm_bwd(const ffno_wn_desc* descs_dev, int n, int max_rows, void* stream) {
    if (!descs_dev || n <= 0 || max_rows <= 0) return FFNO_EINVAL;
    FFNO_LAUNCH(wn_bwd_kernel, dim3((max_rows + 3) / 4, n), dim3(256), 0, (hipStream_t)stream, descs_dev);
    return pw_status();
}

extern "C" int ffno_transpose_batched(const ffno_tr_desc* descs_dev, int n, int max_rows, int max_cols, void* stream) {
    if (!descs_dev || n <= 0 || max_rows <= 0 || max_cols <= 0) return FFNO_EINVAL;
    FFNO_LAUNCH(transpose_batched_kernel, dim3((max_cols + 31) / 32, (max_rows + 31) / 32, n), dim3(256), 0,
                (hipStream_t)stream, descs_dev);
    return pw_status();
}

template <class ST>
static int lift_fwd_impl(const float* x, const float* W, const float* b, typename ST::T* out, int P, int Cin, int C,
                         const ffno_padmap* pad, uint32_t* out_amax, void* stream) {
    if (!x || !W || !b || !out || P <= 0 || Cin <= 0) return FFNO_EINVAL;
    if (Cin > 63) return FFNO_EUNSUPPORTED;
    const PadMapDev pm = make_padmap(pad);
    const size_t smem = sizeof(float) * (size_t)(Cin + 1) * C;
    const int ppb = 512 / (C / 4);
    const dim3 grid((unsigned)min(((long)P + ppb - 1) / ppb, (long)device_cu_count())), block(512);
    hipStream_t s = (hipStream_t)stream;
    if (C == 64)
        FFNO_LAUNCH((lift_fwd_kernel<64, ST>), grid, block, smem, s, x, W, b, out, P, Cin, pm, out_amax);
    else if (C == 32)
        FFNO_LAUNCH((lift_fwd_kernel<32, ST>), grid, block, smem, s, x, W, b, out, P, Cin, pm, out_amax);
    else
        return FFNO_EUNSUPPORTED;
    return pw_status();
}
extern "C" int ffno_lift_fwd(const float* x, const float* W, const float* b, float* out, int P, int Cin, int C,
                             const ffno_padmap* pad, uint32_t* out_amax, void* stream) {
    return lift_fwd_impl<StF32>(x, W, b, out, P, Cin, C, pad, out_amax, stream);
}
// bf16 storage twins of the four pointwise kernels that touch the layer stack's activations (C = 64): `out` / `gout` / `b` / `gb`
// are bf16, everything else (inputs, targets, parameters and their gradients) stays fp32
extern "C" int ffno_lift_fwd_bf16(const float* x, const float* W, const float* b, uint16_t* out, int P, int Cin, int C,
                                  const ffno_padmap* pad, uint32_t* out_amax, void* stream) {
    return lift_fwd_impl<StBf16>(x, W, b, out, P, Cin, C, pad, out_amax, stream);
}

template <class ST>
static int lift_bwd_impl(const float* x, const typename ST::T* gout, const typename ST::T* gout2, float* partial, float* dW, float* db,
                         int P, int Cin, int C, int nsplit, int accumulate, const ffno_padmap* pad, void* stream) {
    if (!x || !gout || !partial || !dW || !db || P <= 0 || Cin <= 0 || nsplit <= 0) return FFNO_EINVAL;
    if (Cin > 63) return FFNO_EUNSUPPORTED;
    const PadMapDev pm = make_padmap(pad);
    const int chunk = (P + nsplit - 1) / nsplit;
    hipStream_t s = (hipStream_t)stream;
    if (C == 64)
        FFNO_LAUNCH((lift_bwd_partial_kernel<64, ST>), dim3(nsplit), dim3(256), 0, s, x, gout, gout2, partial, P, Cin, chunk, pm);
    else if (C == 32)
        FFNO_LAUNCH((lift_bwd_partial_kernel<32, ST>), dim3(nsplit), dim3(256), 0, s, x, gout, gout2, partial, P, Cin, chunk, pm);
    else
        return FFNO_EUNSUPPORTED;
    int rc = pw_status();
    if (rc) return rc;
    const int npairs = C * (Cin + 1);
    FFNO_LAUNCH(lift_bwd_reduce_kernel, dim3((npairs + 3) / 4), dim3(256), 0, s, partial, dW, db, Cin, C, nsplit,
                accumulate);
    return pw_status();
}
extern "C" int ffno_lift_bwd(const float* x, const float* gout, float* partial, float* dW, float* db, int P,
                             int Cin, int C, int nsplit, int accumulate, const ffno_padmap* pad, void* stream) {
    return lift_bwd_impl<StF32>(x, gout, nullptr, partial, dW, db, P, Cin, C, nsplit, accumulate, pad, stream);
}
extern "C" int ffno_lift_bwd2(const float* x, const float* gout, const float* gout2, float* partial, float* dW, float* db, int P,
                              int Cin, int C, int nsplit, int accumulate, const ffno_padmap* pad, void* stream) {
    return lift_bwd_impl<StF32>(x, gout, gout2, partial, dW, db, P, Cin, C, nsplit, accumulate, pad, stream);
}
extern "C" int ffno_lift_bwd_bf16(const float* x, const uint16_t* gout, float* partial, float* dW, float* db, int P,
                                  int Cin, int C, int nsplit, int accumulate, const ffno_padmap* pad, void* stream) {
    return lift_bwd_impl<StBf16>(x, gout, nullptr, partial, dW, db, P, Cin, C, nsplit, accumulate, pad, stream);
}

extern "C" int ffno_lift_bwd_data(const float* gout, const float* W, float* dx, int P, int Cin, int C,
                                  const ffno_padmap* pad, void* stream) {
    if (!gout || !W || !dx || P <= 0 || Cin <= 0) return FFNO_EINVAL;
    if (Cin > 63) return FFNO_EUNSUPPORTED;
    const PadMapDev pm = make_padmap(pad);
    const dim3 grid((unsigned)std::min<long>(((long)P + 255) / 256, 4L * device_cu_count()));
    const size_t smem = sizeof(float) * (size_t)Cin * C;
    hipStream_t s = (hipStream_t)stream;
    if (C == 64)
        FFNO_LAUNCH((lift_bwd_data_kernel<64>), grid, dim3(256), smem, s, gout, W, dx, P, Cin, pm);
    else if (C == 32)
        FFNO_LAUNCH((lift_bwd_data_kernel<32>), grid, dim3(256), smem, s, gout, W, dx, P, Cin, pm);
    else
        return FFNO_EUNSUPPORTED;
    return pw_status();
}

extern "C" int ffno_head_fold(const float* Wa, const float* ca, const float* Wb, const float* cb, float* fold,
                              int C, int D, int O, void* stream) {
    if (!Wa || !ca || !Wb || !cb || !fold || C <= 0 || D <= 0 || O <= 0) return FFNO_EINVAL;
    if (O > kHeadMaxOut) return FFNO_EUNSUPPORTED;
    FFNO_LAUNCH(head_fold_kernel, dim3((O * (C + 1) + 3) / 4), dim3(256), 0, (hipStream_t)stream, Wa, ca, Wb, cb, fold, C,
                D, O);
    return pw_status();
}

template <class ST>
static int head_fwd_impl(const typename ST::T* b, const float* fold, float* y, int P, int C, int O, int accumulate,
                         const ffno_padmap* pad, void* stream) {
    if (!b || !fold || !y || P <= 0 || O <= 0) return FFNO_EINVAL;
    if (O > kHeadMaxOut) return FFNO_EUNSUPPORTED;
    const PadMapDev pm = make_padmap(pad);
    const int ppb = 256 / (C / 4);
    const dim3 grid((unsigned)min(((long)P + ppb - 1) / ppb, 2048L)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (C == 64)
        FFNO_LAUNCH((head_fwd_kernel<64, ST>), grid, block, 0, s, b, fold, y, P, O, accumulate, pm);
    else if (C == 32)
        FFNO_LAUNCH((head_fwd_kernel<32, ST>), grid, block, 0, s, b, fold, y, P, O, accumulate, pm);
    else
        return FFNO_EUNSUPPORTED;
    return pw_status();
}
extern "C" int ffno_head_fwd(const float* b, const float* fold, float* y, int P, int C, int O, int accumulate,
                             const ffno_padmap* pad, void* stream) {
    return head_fwd_impl<StF32>(b, fold, y, P, C, O, accumulate, pad, stream);
}
extern "C" int ffno_head_fwd_bf16(const uint16_t* b, const float* fold, float* y, int P, int C, int O, int accumulate,
                                  const ffno_padmap* pad, void* stream) {
    return head_fwd_impl<StBf16>(b, fold, y, P, C, O, accumulate, pad, stream);
}

template <class ST>
static int head_bwd_impl(const typename ST::T* b, const float* gy, const float* fold, typename ST::T* gb, float* partial,
                         float* red, int P, int C, int O, int nsplit, const ffno_padmap* pad, uint32_t* gb_amax,
                         void* stream) {
    if (!b || !gy || !fold || !partial || !red || P <= 0 || nsplit <= 0 || O <= 0) return FFNO_EINVAL;
    if (O > kHeadMaxOut) return FFNO_EUNSUPPORTED;
    const PadMapDev pm = make_padmap(pad);
    hipStream_t s = (hipStream_t)stream;
    // 1024 threads while every workgroup still has several passes of 1024 / (C / 4) pixels
    const int threads = (long)P >= (long)nsplit * 4 * (1024 / (C / 4)) ? 1024 : 256;
    if (C == 64)
        FFNO_LAUNCH((head_bwd_kernel<64, ST>), dim3(nsplit), dim3(threads), 0, s, b, gy, fold, gb, partial, P, O, pm, gb_amax);
    else if (C == 32)
        FFNO_LAUNCH((head_bwd_kernel<32, ST>), dim3(nsplit), dim3(threads), 0, s, b, gy, fold, gb, partial, P, O, pm, gb_amax);
    else
        return FFNO_EUNSUPPORTED;
    int rc = pw_status();
    if (rc) return rc;
    FFNO_LAUNCH(head_bwd_reduce_kernel, dim3((O * (C + 1) + 3) / 4), dim3(256), 0, s, partial, red, C, O, nsplit);
    return pw_status();
}
extern "C" int ffno_head_bwd(const float* b, const float* gy, const float* fold, float* gb, float* partial,
                             float* red, int P, int C, int O, int nsplit, const ffno_padmap* pad, uint32_t* gb_amax,
                             void* stream) {
    return head_bwd_impl<StF32>(b, gy, fold, gb, partial, red, P, C, O, nsplit, pad, gb_amax, stream);
}
extern "C" int ffno_head_bwd_bf16(const uint16_t* b, const float* gy, const float* fold, uint16_t* gb, float* partial,
                                  float* red, int P, int C, int O, int nsplit, const ffno_padmap* pad, uint32_t* gb_amax,
                                  void* stream) {
    return head_bwd_impl<StBf16>(b, gy, fold, gb, partial, red, P, C, O, nsplit, pad, gb_amax, stream);
}

extern "C" int ffno_head_param_grads(const float* red, const float* Wa, const float* ca, const float* Wb,
                                     float* dWa, float* dca, float* dWb, float* dcb, int C, int D, int O,
                                     int accumulate, void* stream) {
    if (!red || !Wa || !ca || !Wb || !dWa || !dca || !dWb || !dcb || O <= 0) return FFNO_EINVAL;
    const int total = O * D + D + D * C + O;
    FFNO_LAUNCH(head_param_grads_kernel, dim3((unsigned)std::min((total + 255) / 256, 256)), dim3(256), 0, (hipStream_t)stream,
                red, Wa, ca, Wb, dWa, dca, dWb, dcb, C, D, O, accumulate);
    return pw_status();
}

// (1024 elements per slice: at 4096 outputs per sample the one-slice form ran 16 dependent load rounds on 32 workgroups)
static inline int lploss_slices(int n) { return max(1, min(64, (n + 1023) / 1024)); }

extern "C" size_t ffno_lploss_tmp_floats(int B, int n_per_sample) {
    return (size_t)2 * (size_t)B * (size_t)lploss_slices(n_per_sample);
}

extern "C" int ffno_lploss_fwd_bwd(const float* pred, const float* target, float* loss_out, float* gpred,
                                   float* tmp, int B, int n_per_sample, float gscale, const float* affine,
                                   void* stream) {
    if (!pred || !target || !tmp || B <= 0 || n_per_sample <= 0) return FFNO_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int S = lploss_slices(n_per_sample);
    FFNO_LAUNCH(lploss_reduce_kernel, dim3(S, B), dim3(256), 0, s, pred, target, tmp, n_per_sample, affine);
    int rc = pw_status();
    if (rc) return rc;
    const int gx = max(1, min((n_per_sample + 255) / 256, 64));
    FFNO_LAUNCH(lploss_grad_kernel, dim3(gx, B), dim3(256), 0, s, pred, target, tmp, loss_out, gpred, B,
                       n_per_sample, S, gscale, affine);
    return pw_status();
}

extern "C" int ffno_markov_features(const float* x, float* state, float* derived, const float* noise, float* out,
                                    float* partial, int B, int M, int N, int Cx, float low, float high,
                                    float noise_std, float eps, int accumulate, int normalize,
                                    const ffno_markov_extra* extra, void* stream) {
    if (!x || !out || !state || !derived || !partial || B <= 0 || M <= 0 || N <= 0 || Cx <= 0) return FFNO_EINVAL;
    MarkovSrc src{x, extra ? extra->force : nullptr, extra ? extra->mu : nullptr, Cx, extra ? extra->use_position : 1};
    const int D = Cx + (src.use_pos ? 2 : 0) + (src.force ? 1 : 0) + (src.mu ? 1 : 0);
    if (D > 16) return FFNO_EUNSUPPORTED;
    const long P = (long)B * M * N;
    hipStream_t s = (hipStream_t)stream;
    const int nsplit = (int)max(1L, min(256L, (P + 1023) / 1024));
    const int chunk = (int)((P + nsplit - 1) / nsplit);
    if (normalize) {
        if (accumulate) {
            FFNO_LAUNCH(markov_stats_partial_kernel, dim3(nsplit), dim3(256), 0, s, src, partial, P, D, M, N, low, high, chunk);
            int rc = pw_status();
            if (rc) return rc;
        }
        FFNO_LAUNCH(markov_stats_finalize_kernel, dim3(1), dim3(64), 0, s, partial, state, derived, D, nsplit, (float)P, eps,
                    accumulate);
        int rc = pw_status();
        if (rc) return rc;
    }
    const unsigned blocks = (unsigned)min((P * D + 255) / 256, 2048L);
    FFNO_LAUNCH(markov_features_kernel, dim3(blocks), dim3(256), 0, s, src, derived, noise, out, P, D, M, N, low, high,
                noise_std, normalize);
    return pw_status();
}

static int adam_launch(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, int step, float grad_scale, int decoupled, void* stream) {
    if (!p || !g || !m || !v || n == 0 || step <= 0) return FFNO_EINVAL;
    const float bc1 = 1.f - (float)pow((double)beta1, (double)step);
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    const unsigned blocks = (unsigned)min((n + 255) / 256, (size_t)2048);
    FFNO_LAUNCH(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2,
                       eps, weight_decay, bc1, bc2s, grad_scale, decoupled);
    return pw_status();
}

extern "C" int ffno_adamw_flat(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, float grad_scale,
                               void* stream) {
    return adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, 1, stream);
}

extern "C" int ffno_adam_flat(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float grad_scale,
                              void* stream) {
    return adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, 0, stream);
}

extern "C" int ffno_axpy(float* y, const float* x, float alpha, size_t n, void* stream) {
    if (!y || !x || n == 0) return FFNO_EINVAL;
    const unsigned blocks = (unsigned)min((n + 255) / 256, (size_t)2048);
    FFNO_LAUNCH(axpy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, x, alpha, n);
    return pw_status();
}
