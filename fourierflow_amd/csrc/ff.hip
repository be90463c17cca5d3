// Pointwise feed-forward of the F-FNO layer for gfx950 (fp32, v_mfma_f32_32x32x2_f32).
//
// Replaces FeedForward.forward (reference fourierflow/modules/feedforward.py:13-19, n_layers = 2,
// dropout 0, no LayerNorm) + the residual add of FNOFactorized2DBlock.forward (grid_2d.py:169),
// and their autograd:
//     h = relu(s W1^T + b1)        out = resid + h W2^T + b2
//
// Formulation: pixels are the MFMA *column* index (lane & 31), so both GEMMs are evaluated
// transposed -- h^T[hid][px] = W1[hid][:] . s^T[:][px] and out^T[c][px] = W2[c][:] . h^T[:][px].
// The D-fragment of the first GEMM (lane = pixel, registers = hidden rows) is then exactly a valid
// B-fragment of the second one (k index = hidden row), so the [P][H] hidden activations never touch
// LDS or HBM between the two GEMMs; the k-ordering of the second GEMM simply follows the D layout:
// accumulator register r of half h holds hidden row (r&3) + 8(r>>2) + 4h, and the W2 A-operand is
// fetched with the same permutation (4 consecutive hidden rows = one 16-B LDS read).
// The weights (W1 [H][C], W2 [C][H]; 128 KiB at C=64,H=256) live in LDS for the lifetime of a
// persistent workgroup; rows are padded by 4 floats so the 16-B A-operand reads are conflict-free.
#include "ffno_device.h"
#include "ffno.h"

namespace ffno {

template <int C, int H>
struct FFSmem {
    static constexpr int LD1 = C + 4;  // W1s row stride (floats)
    static constexpr int LD2 = H + 4;  // W2s row stride
    static constexpr int W1_OFF = 0;
    static constexpr int W2_OFF = H * LD1;
    static constexpr int B1_OFF = W2_OFF + C * LD2;
    static constexpr int B2_OFF = B1_OFF + H;
    static constexpr int FLOATS = B2_OFF + C;
};

template <int C, int H>
__device__ __forceinline__ void ff_load_weights(float* sm, const float* __restrict__ W1,
                                                const float* __restrict__ W2, const float* b1, const float* b2) {
    using S = FFSmem<C, H>;
    for (int e = threadIdx.x * 4; e < H * C; e += blockDim.x * 4) {
        const float4 v = *reinterpret_cast<const float4*>(W1 + e);
        *reinterpret_cast<float4*>(sm + S::W1_OFF + (e / C) * S::LD1 + (e % C)) = v;
    }
    for (int e = threadIdx.x * 4; e < H * C; e += blockDim.x * 4) {
        const float4 v = *reinterpret_cast<const float4*>(W2 + e);
        *reinterpret_cast<float4*>(sm + S::W2_OFF + (e / H) * S::LD2 + (e % H)) = v;
    }
    if (b1)
        for (int e = threadIdx.x; e < H; e += blockDim.x) sm[S::B1_OFF + e] = b1[e];
    if (b2)
        for (int e = threadIdx.x; e < C; e += blockDim.x) sm[S::B2_OFF + e] = b2[e];
}

// ---- forward / backward-data chain kernel ---------------------------------------------------------
// One code path for both directions (BWD selects the epilogues):
//   forward : in = s,  A1 = W1 [H][C],   A2 = W2 [C][H]    h^T = relu(A1 in^T + b1) ; out^T = A2 h^T + b2 (+resid)
//   backward: in = db, A1 = W2^T [H][C], A2 = W1^T [C][H]   dh^T = mask * (A1 in^T)  ; ds^T  = A2 dh^T
// (the backward takes the TRANSPOSED effective weights so both A operands are row-contiguous 16-B LDS reads).
template <int C, int H, int NW, bool BWD>
__global__ __launch_bounds__(NW * 64) void ff_chain_kernel(const float* __restrict__ in, const float* resid,
                                                           const float* __restrict__ A1g,
                                                           const float* __restrict__ bias1,
                                                           const float* __restrict__ A2g,
                                                           const float* __restrict__ bias2, float* out,
                                                           float* hid_out, uint32_t* mask, int P) {
    using S = FFSmem<C, H>;
    constexpr int KS = C / 2;    // k-steps of GEMM1 (two channels per MFMA)
    constexpr int CTO = C / 32;  // output row tiles of GEMM2
    constexpr int NQ = H / 64;   // hidden chunks of 64 (two 32-row tiles each)
    __shared__ __attribute__((aligned(16))) float sm[S::FLOATS];
    ff_load_weights<C, H>(sm, A1g, A2g, bias1, bias2);
    __syncthreads();
    const float* A1s = sm + S::W1_OFF;
    const float* A2s = sm + S::W2_OFF;
    const float* b1s = sm + S::B1_OFF;
    const float* b2s = sm + S::B2_OFF;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ntiles = (P + 31) >> 5;
    const int tstride = gridDim.x * NW;

    // B operand of GEMM1 for the first tile: this lane's pixel, channels [KS*half, KS*half + KS)
    float nB[KS];
    {
        // tile order is wave-major (tile = wave*gridDim + block): a partial last round (e.g. the reference batch 19:
        // 2432 tiles on 2048 waves) then spreads over all CUs instead of piling 8 extra tiles on a few of them
        const long px0 = (long)(wave * gridDim.x + blockIdx.x) * 32 + j;
        FFNO_UNROLL
        for (int u = 0; u < KS / 4; ++u) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (px0 < P) v = *reinterpret_cast<const float4*>(in + px0 * C + KS * half + 4 * u);
            nB[4 * u + 0] = v.x;
            nB[4 * u + 1] = v.y;
            nB[4 * u + 2] = v.z;
            nB[4 * u + 3] = v.w;
        }
    }
    for (int tile = wave * gridDim.x + blockIdx.x; tile < ntiles; tile += tstride) {
        const long px = (long)tile * 32 + j;
        const bool valid = px < P;
        float sB[KS];
        FFNO_UNROLL
        for (int t = 0; t < KS; ++t) sB[t] = nB[t];
        {   // prefetch the next tile's operand while this tile computes
            const long pxn = px + (long)tstride * 32;
            FFNO_UNROLL
            for (int u = 0; u < KS / 4; ++u) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pxn < P) v = *reinterpret_cast<const float4*>(in + pxn * C + KS * half + 4 * u);
                nB[4 * u + 0] = v.x;
                nB[4 * u + 1] = v.y;
                nB[4 * u + 2] = v.z;
                nB[4 * u + 3] = v.w;
            }
        }
        f32x16 o[CTO];
        FFNO_UNROLL
        for (int to = 0; to < CTO; ++to) o[to] = zero16();
        uint32_t* mp = mask ? mask + ((long)tile * 64 + lane) * NQ : nullptr;
        uint32_t nbits = BWD ? mp[0] : 0u;

        FFNO_NOUNROLL
        for (int q = 0; q < NQ; ++q) {
            const uint32_t mbits = nbits;
            if (BWD && q + 1 < NQ) nbits = mp[q + 1];
            f32x16 hT[2];
            FFNO_UNROLL
            for (int T = 0; T < 2; ++T) {
                FFNO_UNROLL
                for (int r = 0; r < 16; ++r) hT[T][r] = BWD ? 0.f : b1s[64 * q + 32 * T + drow(r, half)];
            }
            FFNO_UNROLL
            for (int u = 0; u < KS / 4; ++u) {
                FFNO_UNROLL
                for (int T = 0; T < 2; ++T) {
                    const float4 a =
                        *reinterpret_cast<const float4*>(A1s + (64 * q + 32 * T + j) * S::LD1 + KS * half + 4 * u);
                    hT[T] = mfma32(a.x, sB[4 * u + 0], hT[T]);
                    hT[T] = mfma32(a.y, sB[4 * u + 1], hT[T]);
                    hT[T] = mfma32(a.z, sB[4 * u + 2], hT[T]);
                    hT[T] = mfma32(a.w, sB[4 * u + 3], hT[T]);
                }
            }
            if (BWD) {
                FFNO_UNROLL
                for (int T = 0; T < 2; ++T) {
                    FFNO_UNROLL
                    for (int r = 0; r < 16; ++r) hT[T][r] = ((mbits >> (16 * T + r)) & 1u) ? hT[T][r] : 0.f;
                }
            } else {
                uint32_t bits = 0;
                FFNO_UNROLL
                for (int T = 0; T < 2; ++T) {
                    FFNO_UNROLL
                    for (int r = 0; r < 16; ++r) {
                        const bool pos = hT[T][r] > 0.f;
                        hT[T][r] = pos ? hT[T][r] : 0.f;
                        bits |= (pos ? 1u : 0u) << (16 * T + r);
                    }
                }
                if (mp) mp[q] = bits;
            }
            if (hid_out && valid) {
                FFNO_UNROLL
                for (int T = 0; T < 2; ++T) {
                    FFNO_UNROLL
                    for (int g = 0; g < 4; ++g) {
                        *reinterpret_cast<float4*>(hid_out + px * H + 64 * q + 32 * T + 8 * g + 4 * half) =
                            make_float4(hT[T][4 * g], hT[T][4 * g + 1], hT[T][4 * g + 2], hT[T][4 * g + 3]);
                    }
                }
            }
            // GEMM2 partial over this hidden chunk, k order = D-fragment order of hT
            FFNO_UNROLL
            for (int T = 0; T < 2; ++T) {
                FFNO_UNROLL
                for (int g = 0; g < 4; ++g) {
                    FFNO_UNROLL
                    for (int to = 0; to < CTO; ++to) {
                        const float4 a = *reinterpret_cast<const float4*>(A2s + (32 * to + j) * S::LD2 + 64 * q +
                                                                           32 * T + 8 * g + 4 * half);
                        o[to] = mfma32(a.x, hT[T][4 * g + 0], o[to]);
                        o[to] = mfma32(a.y, hT[T][4 * g + 1], o[to]);
                        o[to] = mfma32(a.z, hT[T][4 * g + 2], o[to]);
                        o[to] = mfma32(a.w, hT[T][4 * g + 3], o[to]);
                    }
                }
            }
        }
        if (valid) {
            FFNO_UNROLL
            for (int to = 0; to < CTO; ++to) {
                FFNO_UNROLL
                for (int g = 0; g < 4; ++g) {
                    const int c0 = 32 * to + 8 * g + 4 * half;
                    float4 v = make_float4(o[to][4 * g], o[to][4 * g + 1], o[to][4 * g + 2], o[to][4 * g + 3]);
                    if (!BWD) {
                        v.x += b2s[c0];
                        v.y += b2s[c0 + 1];
                        v.z += b2s[c0 + 2];
                        v.w += b2s[c0 + 3];
                        if (resid) {
                            const float4 rv = *reinterpret_cast<const float4*>(resid + px * C + c0);
                            v.x += rv.x;
                            v.y += rv.y;
                            v.z += rv.z;
                            v.w += rv.w;
                        }
                    }
                    *reinterpret_cast<float4*>(out + px * C + c0) = v;
                }
            }
        }
    }
}

// ---- backward, weight path ------------------------------------------------------------------------
// Pixel-sliced "TN" GEMMs (contraction over pixels), partial sums per slice:
//   dW1[hid][c] = sum_p dh[p][hid] s[p][c]     dW2[c][hid] = sum_p db[p][c] h[p][hid]
//   db1[hid]    = sum_p dh[p][hid]             db2[c]      = sum_p db[p][c]
// Both operands are read straight from global memory: for a fixed pixel the lanes read consecutive
// channels (128-B segments), and the big operands (h, dh) are exclusive to one wave.
template <int C, int H>
struct FFWgCfg {
    static constexpr int HT = H / 32;
    static constexpr int NW = HT < 4 ? HT : 4;   // waves per k-group (each owns TPW hidden tiles)
    static constexpr int TPW = HT / NW;
    static constexpr int CT = C / 32;
    static constexpr int UNR = 4;                // k-steps (pixel pairs) per trip, all loads issued up front
    static constexpr int PART = 2 * H * C + H + C;  // floats per slice
    static constexpr int ACC = TPW * CT * 16;    // accumulator floats per lane for one of {dW1, dW2}
};

// Two k-groups of NW waves split the pixel slice (interleaved trips) so every SIMD hosts two waves
// whose load latencies overlap; the groups' accumulators are combined through LDS at the end.
template <int C, int H>
__global__ __launch_bounds__((FFWgCfg<C, H>::NW * 128)) void ff_bwd_weights_partial_kernel(
    const float* __restrict__ s, const float* __restrict__ db, const float* __restrict__ h,
    const float* __restrict__ dh, float* __restrict__ partial, int P, int chunk) {
    using G = FFWgCfg<C, H>;
    constexpr int TPW = G::TPW, CT = G::CT, NW = G::NW, UNR = G::UNR;
    __shared__ float comb[NW * 64 * (G::ACC + 2)];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = wv / NW, wave = wv % NW;
    const int j = lane & 31, half = lane >> 5;
    const long pbeg = (long)blockIdx.x * chunk;
    const long pend = min((long)P, pbeg + chunk);

    f32x16 acc1[TPW][CT], acc2[CT][TPW];
    float bs1[TPW], bs2[CT];
    FFNO_UNROLL
    for (int a = 0; a < TPW; ++a) {
        bs1[a] = 0.f;
        FFNO_UNROLL
        for (int b = 0; b < CT; ++b) {
            acc1[a][b] = zero16();
            acc2[b][a] = zero16();
        }
    }
    FFNO_UNROLL
    for (int b = 0; b < CT; ++b) bs2[b] = 0.f;

    const int nsteps = (int)((max(pend - pbeg, 0L) + 1) >> 1);
    for (int t0 = grp * UNR; t0 < nsteps; t0 += 2 * UNR) {
        float dhA[UNR][TPW], hB[UNR][TPW], sB[UNR][CT], dbA[UNR][CT];
        FFNO_UNROLL
        for (int u = 0; u < UNR; ++u) {
            const long p = pbeg + 2 * (t0 + u) + half;
            const bool valid = p < pend;
            FFNO_UNROLL
            for (int a = 0; a < TPW; ++a) {
                dhA[u][a] = valid ? dh[p * H + 32 * (TPW * wave + a) + j] : 0.f;
                hB[u][a] = valid ? h[p * H + 32 * (TPW * wave + a) + j] : 0.f;
            }
            FFNO_UNROLL
            for (int b = 0; b < CT; ++b) {
                sB[u][b] = valid ? s[p * C + 32 * b + j] : 0.f;
                dbA[u][b] = valid ? db[p * C + 32 * b + j] : 0.f;
            }
        }
        FFNO_UNROLL
        for (int u = 0; u < UNR; ++u) {
            FFNO_UNROLL
            for (int a = 0; a < TPW; ++a) {
                bs1[a] += dhA[u][a];
                FFNO_UNROLL
                for (int b = 0; b < CT; ++b) {
                    acc1[a][b] = mfma32(dhA[u][a], sB[u][b], acc1[a][b]);
                    acc2[b][a] = mfma32(dbA[u][b], hB[u][a], acc2[b][a]);
                }
            }
            FFNO_UNROLL
            for (int b = 0; b < CT; ++b) bs2[b] += dbA[u][b];
        }
    }
    // ---- combine the two k-groups (group 1 -> LDS -> group 0), one accumulator set at a time ----
    const int slot = wave * 64 + lane;  // consecutive lanes -> consecutive floats (conflict-free)
    FFNO_UNROLL
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        if (grp == 1) {
            FFNO_UNROLL
            for (int a = 0; a < TPW; ++a) {
                FFNO_UNROLL
                for (int b = 0; b < CT; ++b) {
                    FFNO_UNROLL
                    for (int r = 0; r < 16; ++r)
                        comb[((a * CT + b) * 16 + r) * (NW * 64) + slot] = pass == 0 ? acc1[a][b][r] : acc2[b][a][r];
                }
            }
            if (pass == 0) {
                FFNO_UNROLL
                for (int a = 0; a < TPW; ++a) comb[(G::ACC) * (NW * 64) + a * (NW * 64) + slot] = bs1[a];
            } else {
                FFNO_UNROLL
                for (int b = 0; b < CT; ++b) comb[(G::ACC) * (NW * 64) + b * (NW * 64) + slot] = bs2[b];
            }
        }
        __syncthreads();
        if (grp == 0) {
            FFNO_UNROLL
            for (int a = 0; a < TPW; ++a) {
                FFNO_UNROLL
                for (int b = 0; b < CT; ++b) {
                    FFNO_UNROLL
                    for (int r = 0; r < 16; ++r) {
                        const float v = comb[((a * CT + b) * 16 + r) * (NW * 64) + slot];
                        if (pass == 0)
                            acc1[a][b][r] += v;
                        else
                            acc2[b][a][r] += v;
                    }
                }
            }
            if (pass == 0) {
                FFNO_UNROLL
                for (int a = 0; a < TPW; ++a) bs1[a] += comb[(G::ACC) * (NW * 64) + a * (NW * 64) + slot];
            } else {
                FFNO_UNROLL
                for (int b = 0; b < CT; ++b) bs2[b] += comb[(G::ACC) * (NW * 64) + b * (NW * 64) + slot];
            }
        }
    }
    // the shuffles below need whole waves: every wave executes them, only group 0 stores
    float* part = partial + (long)blockIdx.x * G::PART;
    float* pW1 = part;
    float* pW2 = part + H * C;
    float* pb1 = part + 2 * H * C;
    float* pb2 = pb1 + H;
    FFNO_UNROLL
    for (int a = 0; a < TPW; ++a) {
        if (grp == 0) {
            FFNO_UNROLL
            for (int b = 0; b < CT; ++b) {
                FFNO_UNROLL
                for (int r = 0; r < 16; ++r) {
                    pW1[(32 * (TPW * wave + a) + drow(r, half)) * C + 32 * b + j] = acc1[a][b][r];
                    pW2[(32 * b + drow(r, half)) * H + 32 * (TPW * wave + a) + j] = acc2[b][a][r];
                }
            }
        }
        const float v = bs1[a] + __shfl_xor(bs1[a], 32);
        if (grp == 0 && half == 0) pb1[32 * (TPW * wave + a) + j] = v;
    }
    FFNO_UNROLL
    for (int b = 0; b < CT; ++b) {
        const float v = bs2[b] + __shfl_xor(bs2[b], 32);
        if (grp == 0 && wave == 0 && half == 0) pb2[32 * b + j] = v;
    }
}

// block = 64 consecutive elements x 4 slice-groups; 8 independent loads in flight per thread
__global__ __launch_bounds__(256) void ff_bwd_weights_reduce_kernel(const float* __restrict__ partial, float* dW1,
                                                                    float* dW2, float* db1, float* db2, int C, int H,
                                                                    int nsplit, int accumulate) {
    __shared__ float red[4][64];
    const int part = 2 * H * C + H + C;
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), sg = threadIdx.x >> 6;
    float sum = 0.f;
    if (e < part) {
        float t[8];
        int sp = sg;
        for (; sp + 28 < nsplit; sp += 32) {
            FFNO_UNROLL
            for (int u = 0; u < 8; ++u) t[u] = partial[(long)(sp + 4 * u) * part + e];
            FFNO_UNROLL
            for (int u = 0; u < 8; ++u) sum += t[u];
        }
        for (; sp < nsplit; sp += 4) sum += partial[(long)sp * part + e];
    }
    red[sg][threadIdx.x & 63] = sum;
    __syncthreads();
    if (sg == 0 && e < part) {
        sum = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        float* dst;
        if (e < H * C)
            dst = dW1 + e;
        else if (e < 2 * H * C)
            dst = dW2 + (e - H * C);
        else if (e < 2 * H * C + H)
            dst = db1 + (e - 2 * H * C);
        else
            dst = db2 + (e - 2 * H * C - H);
        *dst = accumulate ? (*dst + sum) : sum;
    }
}

static inline int ff_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFNO_OK : (int)e;
}

// number of persistent workgroups: one per CU (the 128-KiB weight image allows one resident block)
static const int kFFBlocks = 256;

}  // namespace ffno

using namespace ffno;

extern "C" size_t ffno_ff_mask_words(int P, int H) { return (size_t)((P + 31) / 32) * 64 * (size_t)(H / 64); }

#define FFNO_FF_DISPATCH(MACRO) \
    MACRO(64, 256)              \
    MACRO(64, 128)              \
    MACRO(32, 128)              \
    MACRO(32, 64)

extern "C" int ffno_ff_fwd(const float* s, const float* resid, const float* W1, const float* b1, const float* W2,
                           const float* b2, float* out, float* h, uint32_t* mask, int P, int C, int H,
                           void* stream) {
    if (!s || !W1 || !b1 || !W2 || !b2 || !out || P <= 0) return FFNO_EINVAL;
    constexpr int NW = 8;
    const int ntiles = (P + 31) / 32;
    const dim3 grid(max(1, min(kFFBlocks, (ntiles + NW - 1) / NW))), block(NW * 64);
    hipStream_t st = (hipStream_t)stream;
#define CASE(CC, HH)                                                                                         \
    if (C == CC && H == HH) {                                                                                \
        FFNO_LAUNCH((ff_chain_kernel<CC, HH, NW, false>), grid, block, 0,                                      \
                           st, s, resid, W1, b1, W2, b2, out, h, mask, P);                                   \
        return ff_launch_status();                                                                           \
    }
    FFNO_FF_DISPATCH(CASE)
#undef CASE
    return FFNO_EUNSUPPORTED;
}

extern "C" int ffno_ff_bwd_data(const float* db, const uint32_t* mask, const float* W1t, const float* W2t, float* dh,
                                float* ds, int P, int C, int H, void* stream) {
    if (!db || !mask || !W1t || !W2t || !ds || P <= 0) return FFNO_EINVAL;
    constexpr int NW = 8;
    const int ntiles = (P + 31) / 32;
    const dim3 grid(max(1, min(kFFBlocks, (ntiles + NW - 1) / NW))), block(NW * 64);
    hipStream_t st = (hipStream_t)stream;
#define CASE(CC, HH)                                                                                              \
    if (C == CC && H == HH) {                                                                                     \
        FFNO_LAUNCH((ff_chain_kernel<CC, HH, NW, true>), grid, block, 0,                                      \
                           st, db, nullptr, W2t, nullptr, W1t, nullptr, ds, dh, const_cast<uint32_t*>(mask), P);                                                      \
        return ff_launch_status();                                                                                \
    }
    FFNO_FF_DISPATCH(CASE)
#undef CASE
    return FFNO_EUNSUPPORTED;
}

extern "C" size_t ffno_ff_wgrad_partial_floats(int C, int H, int nsplit) {
    return (size_t)nsplit * (size_t)(2 * H * C + H + C);
}

extern "C" int ffno_ff_bwd_weights_partial(const float* s, const float* db, const float* h, const float* dh,
                                           float* partial, int P, int C, int H, int nsplit, void* stream) {
    if (!s || !db || !h || !dh || !partial || P <= 0 || nsplit <= 0) return FFNO_EINVAL;
    int chunk = (P + nsplit - 1) / nsplit;
    chunk += chunk & 1;
    hipStream_t st = (hipStream_t)stream;
#define CASE(CC, HH)                                                                                               \
    if (C == CC && H == HH) {                                                                                      \
        FFNO_LAUNCH((ff_bwd_weights_partial_kernel<CC, HH>), dim3(nsplit), dim3(FFWgCfg<CC, HH>::NW * 128), 0, \
                           st, s, db, h, dh, partial, P, chunk);                                                   \
        return ff_launch_status();                                                                                 \
    }
    FFNO_FF_DISPATCH(CASE)
#undef CASE
    return FFNO_EUNSUPPORTED;
}

extern "C" int ffno_ff_bwd_weights_reduce(const float* partial, float* dW1, float* dW2, float* db1, float* db2,
                                          int C, int H, int nsplit, int accumulate, void* stream) {
    if (!partial || !dW1 || !dW2 || !db1 || !db2 || nsplit <= 0) return FFNO_EINVAL;
    const int part = 2 * H * C + H + C;
    FFNO_LAUNCH(ff_bwd_weights_reduce_kernel, dim3((part + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       partial, dW1, dW2, db1, db2, C, H, nsplit, accumulate);
    return ff_launch_status();
}
