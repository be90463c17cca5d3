// General pointwise linear layers on the fp32 matrix cores: the feed-forward shapes the fused chain kernels do not take.
//
// The fused kernels of ffx.hip / ff.hip implement FeedForward (reference fourierflow/modules/feedforward.py:6-24) for the
// configuration every shipped experiment uses: n_layers = 2, dropout = 0.  The reference class is more general --
//     for i in range(n_layers):  Linear(dim | dim*factor -> dim*factor | dim), Dropout(p), ReLU (all but the last), [LayerNorm]
// -- and FNOFactorized2DBlock adds Dropout(in_dropout) after in_proj (grid_2d.py:113,158).  These kernels are that general path:
// one layer per launch, hidden activations kept in HBM for the backward pass, any (Cin, Cout) that is a multiple of 32 up to
// 256, exact fp32 arithmetic (v_mfma_f32_32x32x2_f32 is an fmaf chain), and dropout as a counter-based mask that the backward
// pass REGENERATES from (seed, element index) instead of storing:
//     fwd          out[p][o] = keep(p, o) / (1 - p_drop) * act(b[o] + sum_i W[o][i] x[p][i])  (+ resid[p][o])
//                  (Linear -> Dropout -> ReLU of the reference: the positive dropout scale commutes with the ReLU)
//     bwd_data     dpre[p][o] = g[p][o] * scale * (y given ? [y[p][o] > 0] : keep(p, o));   dx[p][i] = sum_o dpre[p][o] W[o][i]
//                  (y = the kept OUTPUT of a ReLU layer: its zeros already contain the dropped units)
//     bwd_weights  dW[o][i] = sum_p dpre[p][o] x[p][i],  db[o] = sum_p dpre[p][o]   (deterministic two-stage reduction)
// One GEMM core serves the three: C[M][N] = A[M][K] B[K][N] on 64 x 64 block tiles (four waves, one 32 x 32 MFMA tile each),
// K staged through LDS in chunks of 32, A / B read through (row, column) strides so that no operand is ever transposed in HBM.
#include <algorithm>

#include "ffno_device.h"
#include "ffno.h"

namespace ffno {

// keep(seed, idx): the dropout mask bit of element idx: fmix32(fmix32(idx + k1) ^ k2) with k1 = fmix32(seed), k2 = fmix32(~seed)
// (fmix32: the 32-bit murmur3 finaliser, a bijection; uniform to ~2^-32).  The seed enters twice, on either side of a
// non-linear round: `idx * G + seed` made the masks of two sites shifted copies of one bit sequence, and a single
// `fmix32(idx ^ fmix32(seed))` made them XOR-permuted copies (mask2[i] = mask1[i ^ d]: ADVICE r04) -- neither is an independent draw.
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ bool drop_keep(uint32_t seed, uint32_t idx, uint32_t thr) {
    return fmix32(fmix32(idx + fmix32(seed)) ^ fmix32(~seed)) >= thr;
}

struct GlinDrop {
    uint32_t seed, thr;   // keep iff hash >= thr (thr = p * 2^32; 0 = no dropout)
    float scale;          // 1 / (1 - p)
};

struct GlinArgs {
    const float* A;
    long a_rs, a_cs;      // A[m][k] = A[m * a_rs + k * a_cs]
    const float* B;
    long b_rs, b_cs;      // B[k][n] = B[k * b_rs + n * b_cs]
    int M, N, K;
    // transform of A while it is staged (the backward kernels read g and make dpre on the fly): a = g * scale * gate, with
    // gate = [y > 0] when y is given, keep(idx) when drop.thr != 0, else 1; idx / y are addressed like g ([P][O] row-major)
    const float* y;
    GlinDrop drop;
    int a_is_g;           // 0: A is used as it is (forward)
    int g_cols;           // O: row length of g / y (index = p * O + o)
    int a_m_is_col;       // bwd_weights: A's m index is the column o of g (k = p); else m = p, k = o
};

constexpr int kGT = 64, kGK = 32;

// stage one 64 x 32 A chunk and one 32 x 64 B chunk (zero padded), then 16 k-steps of v_mfma_f32_32x32x2_f32 per wave
template <bool AKM, bool BNM>
__device__ __forceinline__ void glin_chunk(const GlinArgs& g, int m0, int n0, int k0, float (*As)[kGK + 1], float (*Bs)[kGT + 1],
                                           f32x16& acc, float* rowsum) {
    const int tid = threadIdx.x;
    FFNO_UNROLL
    for (int t = 0; t < kGT * kGK / 256; ++t) {
        const int e = tid + 256 * t;
        // AKM: consecutive threads walk m (A is k-major in memory: a_rs == 1), else they walk k
        const int m = AKM ? e % kGT : e / kGK, k = AKM ? e / kGT : e % kGK;
        float v = 0.f;
        if (m0 + m < g.M && k0 + k < g.K) {
            v = g.A[(long)(m0 + m) * g.a_rs + (long)(k0 + k) * g.a_cs];
            if (g.a_is_g) {
                const long p = g.a_m_is_col ? k0 + k : m0 + m, o = g.a_m_is_col ? m0 + m : k0 + k;
                const long idx = p * g.g_cols + o;
                bool on = true;
                if (g.y)
                    on = g.y[idx] > 0.f;
                else if (g.drop.thr)
                    on = drop_keep(g.drop.seed, (uint32_t)idx, g.drop.thr);
                v = on ? v * g.drop.scale : 0.f;
            }
        }
        As[m][k] = v;
    }
    FFNO_UNROLL
    for (int t = 0; t < kGK * kGT / 256; ++t) {
        const int e = tid + 256 * t;
        const int k = BNM ? e % kGK : e / kGT, n = BNM ? e / kGK : e % kGT;
        float v = 0.f;
        if (k0 + k < g.K && n0 + n < g.N) v = g.B[(long)(k0 + k) * g.b_rs + (long)(n0 + n) * g.b_cs];
        Bs[k][n] = v;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int j = lane & 31, half = lane >> 5;
    FFNO_UNROLL
    for (int kk = 0; kk < kGK; kk += 2) acc = mfma32(As[wr * 32 + j][kk + half], Bs[kk + half][wc * 32 + j], acc);
    if (rowsum && tid < kGT) {          // bias gradient: row sums of the staged dpre^T chunk
        float s = 0.f;
        for (int k = 0; k < kGK; ++k) s += As[tid][k];
        *rowsum += s;
    }
    __syncthreads();
}

// forward / backward-data: one block per 64 x 64 output tile
template <bool BNM>
__global__ __launch_bounds__(256) void glin_rowmajor_kernel(GlinArgs g, const float* __restrict__ bias, const float* resid,
                                                            float* out, int relu, GlinDrop odrop, int accumulate) {
    __shared__ float As[kGT][kGK + 1];
    __shared__ float Bs[kGK][kGT + 1];
    const int m0 = blockIdx.x * kGT, n0 = blockIdx.y * kGT;
    f32x16 acc = zero16();
    for (int k0 = 0; k0 < g.K; k0 += kGK) glin_chunk<false, BNM>(g, m0, n0, k0, As, Bs, acc, nullptr);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
    const int j = lane & 31, half = lane >> 5;
    const int n = n0 + wc * 32 + j;
    if (n >= g.N) return;
    const float b = bias ? bias[n] : 0.f;
    FFNO_UNROLL
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 32 + drow(r, half);
        if (m < g.M) {
            const long idx = (long)m * g.N + n;
            float v = acc[r] + b;
            if (relu) v = fmaxf(v, 0.f);
            if (odrop.thr) v = drop_keep(odrop.seed, (uint32_t)idx, odrop.thr) ? v * odrop.scale : 0.f;
            if (resid) v += resid[idx];
            if (accumulate) v += out[idx];
            out[idx] = v;
        }
    }
}

// weight gradients: block (tile, split) accumulates its slice of the pixels; partial[split][M * N + M] (dW then db)
__global__ __launch_bounds__(256) void glin_wgrad_kernel(GlinArgs g, float* __restrict__ partial, int kchunk) {
    __shared__ float As[kGT][kGK + 1];
    __shared__ float Bs[kGK][kGT + 1];
    const int tiles_n = (g.N + kGT - 1) / kGT;
    const int m0 = (blockIdx.x / tiles_n) * kGT, n0 = (blockIdx.x % tiles_n) * kGT;
    const int kbeg = blockIdx.y * kchunk, kend = min(g.K, kbeg + kchunk);
    f32x16 acc = zero16();
    float rs = 0.f;
    GlinArgs s = g;
    s.K = kend;
    for (int k0 = kbeg; k0 < kend; k0 += kGK) glin_chunk<true, false>(s, m0, n0, k0, As, Bs, acc, n0 == 0 ? &rs : nullptr);
    float* part = partial + (long)blockIdx.y * ((long)g.M * g.N + g.M);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
    const int j = lane & 31, half = lane >> 5;
    const int n = n0 + wc * 32 + j;
    if (n < g.N) {
        FFNO_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wr * 32 + drow(r, half);
            if (m < g.M) part[(long)m * g.N + n] = acc[r];
        }
    }
    if (n0 == 0 && (int)threadIdx.x < kGT && m0 + (int)threadIdx.x < g.M) part[(long)g.M * g.N + m0 + threadIdx.x] = rs;
}

__global__ __launch_bounds__(256) void glin_reduce_kernel(const float* __restrict__ partial, float* dW, float* db, long nW, long nb,
                                                          int nsplit, int accumulate) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= nW + nb) return;
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += partial[(long)sp * (nW + nb) + e];
    float* dst = e < nW ? dW + e : db + (e - nW);
    *dst = accumulate ? *dst + s : s;
}

// x[i] = keep(seed, i) ? x[i] / (1 - p) : 0   (in place; the same call on a gradient is its backward)
__global__ __launch_bounds__(256) void dropout_kernel(float* __restrict__ x, size_t n, GlinDrop d) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        x[i] = drop_keep(d.seed, (uint32_t)i, d.thr) ? x[i] * d.scale : 0.f;
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ keep, size_t n, GlinDrop d) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        keep[i] = drop_keep(d.seed, (uint32_t)i, d.thr) ? 1 : 0;
}

static inline int glin_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFNO_OK : (int)e;
}
static inline bool glin_dims_ok(int Cin, int Cout) {
    return Cin >= 1 && Cout >= 1 && Cin <= 256 && Cout <= 256;
}
static inline int glin_drop(GlinDrop& d, float p, uint32_t seed) {
    if (!(p >= 0.f) || p >= 1.f) return FFNO_EINVAL;
    d.seed = seed;
    d.thr = p > 0.f ? (uint32_t)std::min(4294967295.0, (double)p * 4294967296.0) : 0u;
    d.scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    return FFNO_OK;
}

}  // namespace ffno

using namespace ffno;

extern "C" int ffno_glin_supported(int Cin, int Cout) { return glin_dims_ok(Cin, Cout) ? 1 : 0; }

extern "C" int ffno_glin_fwd(const float* x, const float* W, const float* b, const float* resid, float* out, long P, int Cin,
                             int Cout, int relu, float drop_p, uint32_t drop_seed, void* stream) {
    if (!x || !W || !out || P <= 0 || P * (long)std::max(Cin, Cout) >= (1L << 32)) return FFNO_EINVAL;
    if (!glin_dims_ok(Cin, Cout)) return FFNO_EUNSUPPORTED;
    GlinDrop od;
    const int rc = glin_drop(od, drop_p, drop_seed);
    if (rc) return rc;
    GlinArgs g{x, Cin, 1, W, 1, Cin, (int)P, Cout, Cin, nullptr, GlinDrop{0, 0, 1.f}, 0, Cout, 0};
    const dim3 grid((unsigned)((P + kGT - 1) / kGT), (Cout + kGT - 1) / kGT);
    FFNO_LAUNCH((glin_rowmajor_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, g, b, resid, out, relu, od, 0);
    return glin_status();
}

extern "C" int ffno_glin_bwd_data(const float* g_, const float* y, const float* W, float* dx, long P, int Cin, int Cout,
                                  float drop_p, uint32_t drop_seed, int accumulate, void* stream) {
    if (!g_ || !W || !dx || P <= 0 || P * (long)std::max(Cin, Cout) >= (1L << 32)) return FFNO_EINVAL;
    if (!glin_dims_ok(Cin, Cout)) return FFNO_EUNSUPPORTED;
    GlinDrop d;
    const int rc = glin_drop(d, drop_p, drop_seed);
    if (rc) return rc;
    GlinArgs g{g_, Cout, 1, W, Cin, 1, (int)P, Cin, Cout, y, d, 1, Cout, 0};
    const dim3 grid((unsigned)((P + kGT - 1) / kGT), (Cin + kGT - 1) / kGT);
    FFNO_LAUNCH((glin_rowmajor_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, g, (const float*)nullptr,
                (const float*)nullptr, dx, 0, GlinDrop{0, 0, 1.f}, accumulate);
    return glin_status();
}

extern "C" int ffno_glin_wgrad_nsplit(long P) { return (int)std::max<long>(1, std::min<long>(64, (P + 2047) / 2048)); }
extern "C" size_t ffno_glin_wgrad_partial_floats(long P, int Cin, int Cout) {
    return (size_t)ffno_glin_wgrad_nsplit(P) * ((size_t)Cin * Cout + Cout);
}

extern "C" int ffno_glin_bwd_weights(const float* g_, const float* y, const float* x, float* partial, float* dW, float* db, long P,
                                     int Cin, int Cout, float drop_p, uint32_t drop_seed, int accumulate, void* stream) {
    if (!g_ || !x || !partial || !dW || !db || P <= 0 || P * (long)std::max(Cin, Cout) >= (1L << 32)) return FFNO_EINVAL;
    if (!glin_dims_ok(Cin, Cout)) return FFNO_EUNSUPPORTED;
    GlinDrop d;
    int rc = glin_drop(d, drop_p, drop_seed);
    if (rc) return rc;
    const int nsplit = ffno_glin_wgrad_nsplit(P);
    const int kchunk = (int)(((P + nsplit - 1) / nsplit + kGK - 1) / kGK * kGK);
    // dW[o][i] = sum_p dpre[p][o] x[p][i]:  A[m = o][k = p] = g[p * Cout + o],  B[k = p][n = i] = x[p * Cin + i]
    GlinArgs g{g_, 1, Cout, x, Cin, 1, Cout, Cin, (int)P, y, d, 1, Cout, 1};
    const int tiles = ((Cout + kGT - 1) / kGT) * ((Cin + kGT - 1) / kGT);
    hipStream_t st = (hipStream_t)stream;
    FFNO_LAUNCH(glin_wgrad_kernel, dim3(tiles, nsplit), dim3(256), 0, st, g, partial, kchunk);
    rc = glin_status();
    if (rc) return rc;
    const long nW = (long)Cin * Cout, nb = Cout;
    FFNO_LAUNCH(glin_reduce_kernel, dim3((unsigned)((nW + nb + 255) / 256)), dim3(256), 0, st, partial, dW, db, nW, nb, nsplit,
                accumulate);
    return glin_status();
}

extern "C" int ffno_dropout(float* x, size_t n, float p, uint32_t seed, void* stream) {
    if (!x || n == 0 || n >= (1UL << 32)) return FFNO_EINVAL;
    GlinDrop d;
    const int rc = glin_drop(d, p, seed);
    if (rc) return rc;
    if (!d.thr) return FFNO_OK;
    const int blocks = (int)std::min<size_t>((size_t)4 * device_cu_count(), (n + 255) / 256);
    FFNO_LAUNCH(dropout_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, d);
    return glin_status();
}

extern "C" int ffno_dropout_mask(uint8_t* keep, size_t n, float p, uint32_t seed, void* stream) {
    if (!keep || n == 0 || n >= (1UL << 32)) return FFNO_EINVAL;
    GlinDrop d;
    const int rc = glin_drop(d, p, seed);
    if (rc) return rc;
    const int blocks = (int)std::min<size_t>((size_t)4 * device_cu_count(), (n + 255) / 256);
    FFNO_LAUNCH(dropout_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, keep, n, d);
    return glin_status();
}
