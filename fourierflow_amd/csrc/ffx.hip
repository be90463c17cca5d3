// Pointwise feed-forward of the F-FNO layer on the bf16 matrix cores at fp32 accuracy ("bf16x3").
//
// Same operator as ff.hip (FeedForward.forward, reference fourierflow/modules/feedforward.py:13-19 with n_layers = 2,
// + the residual add of grid_2d.py:169, and their autograd):
//     h = relu(s W1^T + b1)        out = resid + h W2^T + b2
// but every fp32 operand is cut EXACTLY into three bf16 planes and each product is evaluated with six
// v_mfma_f32_32x32x16_bf16 (ffno_device.h "split-bf16"): fp32-grade results at 8/3 of the fp32 MFMA rate.  With the
// matrix work that cheap the kernels are organised around HBM traffic instead:
//
//   * the [P][H] hidden activations are never written: the forward keeps only the ReLU sign bits (P*H/8 bytes) and the
//     weight-gradient kernel RECOMPUTES h and dh from s and db (two extra GEMMs that cost less than reading them back);
//   * a wave owns a 32-row chunk of the hidden layer for the lifetime of a persistent workgroup and keeps its slices of
//     both weight matrices (pre-split, pre-permuted by ffx_pack) in registers -- no LDS bandwidth for the A operands;
//     all waves of a workgroup work on the same 32-pixel tile;
//   * forward / backward-data: GEMM1 gives the wave h^T[chunk][px]; as in ff.hip the D fragment is re-used directly as
//     the B operand of GEMM2 (k order = D-fragment order, matched by the packed A2 operand), which yields a PARTIAL
//     out^T[c][px] over the wave's hidden chunk; the partials meet in LDS (double-buffered, one barrier per tile) and
//     each wave reduces + stores 1/NW of the output tile;
//   * weight gradients: h^T / dh^T are recomputed TRANSPOSED (pixels = D rows) so that the D fragment is again a valid
//     B operand, now of the pixel-contraction GEMMs dW2[c][hid] = sum_px db[px][c] h[px][hid] and
//     dW1^T[c][hid] = sum_px s[px][c] dh[px][hid]; every wave accumulates its own [C x 32] slice of both for all the
//     pixels of the workgroup -- no cross-wave reduction at all.
//
// Activation tiles are staged through LDS already split (each element is split once per workgroup, not once per wave).
#include <algorithm>

#include "ffno_device.h"
#include "ffno.h"

#include <string.h>

namespace ffno {

// magnitude bound (2^kFfRangeTarget) the split-fp16 kernels bring their staged rows to: 2^12 of head-room up to the half
// format's 65504 for the growth through the first linear map of the chain (|h| <= ||W row||_1 |s| + |b|)
constexpr int kFfRangeTarget = 4;

template <int C, int H, int CPW_ = 1>
struct FxCfg {
    static constexpr int CPW = CPW_;              // hidden chunks (of 32 rows) per wave
    static constexpr int NW = H / (32 * CPW);     // waves per workgroup (two per SIMD at H = 256)
    static constexpr int NT = NW * 64;
    static constexpr int KS = C / 16;             // k16 steps over the channels
    static constexpr int CTO = C / 32;            // 32-row tiles over the channels
    static constexpr int NV = (32 * C / 4) / NT;  // float4 per thread per staged [32 px][C] tile
    static constexpr int G = CTO * 4;             // float4 groups per lane of an output tile
    static constexpr int GPW = G / NW;            // ... reduced by each wave
    // pixel-major planes (B operand of GEMM1 / A operand of the recompute): [32 px][C] bf16, 16 B row pad
    static constexpr int PROW = 2 * C + 16;
    static constexpr int PPLANE = 32 * PROW;
    // channel-major planes (A operand of the pixel-contraction GEMMs): [C][32 px in k order] bf16, 16 B row pad
    static constexpr int TROW = 64 + 16;
    static constexpr int TPLANE = C * TROW;
    static constexpr int PART = 2 * H * C + H + C;
    static_assert(NV >= 1 && NV * NT * 4 == 32 * C, "tile staging map");
    static_assert(GPW >= 1 && GPW * NW == G, "output reduction map");
};

// ---- weight packing -----------------------------------------------------------------------------------------------
// type 1 ("hidden on lanes"):  frag (w, st)      lane (j, half) slot e  <-  W(hid = 32w + j, c = 16 st + 8 half + e)
// type 2 ("channel on lanes"): frag (w, mt, s2)  lane (j, half) slot e  <-  W(hid = 32w + (e&3) + 8(2 s2 + (e>>2)) + 4 half,
//                                                                           c = 32 mt + j)
// with W(hid, c) = src[hid * sh + c * sc]; each fragment is three planes of 64 lanes x 16 B.
struct FxPackDesc {
    const float* src;
    u32x4* dst;
    int sh, sc, type, pad;
};

template <class S>
__global__ __launch_bounds__(256) void ffx_pack_kernel(const FxPackDesc* __restrict__ descs, int C, int H) {
    const FxPackDesc d = descs[blockIdx.y];
    const int KS = C / 16, CTO = C / 32, NW = H / 32;
    const int nfrag = d.type == 1 ? NW * KS : NW * CTO * 2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nfrag * 64) return;
    const int frag = t >> 6, lane = t & 63, j = lane & 31, half = lane >> 5;
    float v[8];
    if (d.type == 1) {
        const int w = frag / KS, st = frag % KS;
        for (int e = 0; e < 8; ++e) v[e] = d.src[(long)(32 * w + j) * d.sh + (long)(16 * st + 8 * half + e) * d.sc];
    } else {
        const int s2 = frag & 1, mt = (frag >> 1) % CTO, w = (frag >> 1) / CTO;
        for (int e = 0; e < 8; ++e)
            v[e] = d.src[(long)(32 * w + (e & 3) + 8 * (2 * s2 + (e >> 2)) + 4 * half) * d.sh + (long)(32 * mt + j) * d.sc];
    }
    const typename S::Frag f = S::split8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    FFNO_UNROLL
    for (int p = 0; p < S::NP; ++p) d.dst[(frag * S::NP + p) * 64 + lane] = S::plane(f, p);
}

__device__ __forceinline__ Bf3 load_frag(const u32x4* __restrict__ pk, int frag, int lane) {
    Bf3 f;
    f.hi = pk[(frag * 3 + 0) * 64 + lane];
    f.mid = pk[(frag * 3 + 1) * 64 + lane];
    f.lo = pk[(frag * 3 + 2) * 64 + lane];
    return f;
}

// (acc << 1) | msb(x)
__device__ __forceinline__ uint32_t push_sign(uint32_t acc, uint32_t x) {
    return plat::shift_in_msb(acc, x);
}
// all-ones if bit b of x is set, else 0
__device__ __forceinline__ uint32_t bit_mask(uint32_t x, int b) {
    return plat::bit_to_mask(x, b);
}

// three planes at the same offset of an LDS tile
__device__ __forceinline__ Bf3 lds_frag(const char* base, int plane_bytes, int off) {
    Bf3 f;
    f.hi = *reinterpret_cast<const u32x4*>(base + off);
    f.mid = *reinterpret_cast<const u32x4*>(base + plane_bytes + off);
    f.lo = *reinterpret_cast<const u32x4*>(base + 2 * plane_bytes + off);
    return f;
}

// split 4 consecutive values and store them as 8 B per plane
__device__ __forceinline__ void stage4(char* base, int plane_bytes, int off, float x, float y, float z, float w) {
    unsigned h0, m0, l0, h1, m1, l1;
    split3_pair(x, y, h0, m0, l0);
    split3_pair(z, w, h1, m1, l1);
    *reinterpret_cast<uint2*>(base + off) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(base + plane_bytes + off) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(base + 2 * plane_bytes + off) = make_uint2(l0, l1);
}

template <class S>
__device__ __forceinline__ typename S::Frag load_frag_s(const u32x4* __restrict__ pk, int frag, int lane) {
    typename S::Frag f;
    FFNO_UNROLL
    for (int p = 0; p < S::NP; ++p) S::set_plane(f, p, pk[(frag * S::NP + p) * 64 + lane]);
    return f;
}
// NP planes at the same offset of an LDS tile
template <class S>
__device__ __forceinline__ typename S::Frag lds_frag_s(const char* base, int plane_bytes, int off) {
    typename S::Frag f;
    FFNO_UNROLL
    for (int p = 0; p < S::NP; ++p) S::set_plane(f, p, *reinterpret_cast<const u32x4*>(base + p * plane_bytes + off));
    return f;
}
// split 4 consecutive values and store them as 8 B per plane
template <class S>
__device__ __forceinline__ void stage4_s(char* base, int plane_bytes, int off, float x, float y, float z, float w) {
    uint2 pl[S::NP];
    S::split4(x, y, z, w, pl);
    FFNO_UNROLL
    for (int p = 0; p < S::NP; ++p) *reinterpret_cast<uint2*>(base + p * plane_bytes + off) = pl[p];
}

// ---- forward / backward-data -----------------------------------------------------------------------------------------
//   forward : in = s,  A1 = pack1(W1),   A2 = pack2(W2)     h = relu(A1 in + b1), sign bits -> mask ; out = A2 h + b2 (+ resid)
//   backward: in = db, A1 = pack1(W2^T), A2 = pack2(W1^T)   dh = mask ? A1 in : 0                    ; ds  = A2 dh
// ST = storage format of in / in2 / sum_out / resid / out (ffno_device.h "activation storage formats"); with StBf16 the input
// sum is rounded to the stored format BEFORE it is used, so the kernels that read sum_out later see the operand of this one.
template <int C, int H, bool BWD, class S, class ST = StF32>
__global__ __launch_bounds__((FxCfg<C, H>::NT)) void ffx_chain_kernel(const typename ST::T* __restrict__ in,
                                                                     const typename ST::T* __restrict__ in2,
                                                                     typename ST::T* sum_out,
                                                                     const typename ST::T* resid,
                                                                     const u32x4* __restrict__ pk1,
                                                                     const float* __restrict__ bias1,
                                                                     const u32x4* __restrict__ pk2,
                                                                     const float* __restrict__ bias2, typename ST::T* out,
                                                                     uint32_t* mask, int P, const unsigned* in_amax, unsigned* out_amax) {
    using F = FxCfg<C, H>;
    constexpr int NW = F::NW, KS = F::KS, CTO = F::CTO, NV = F::NV, G = F::G, GPW = F::GPW, CPW = F::CPW;
    __shared__ __attribute__((aligned(16))) char sp[2][S::NP * F::PPLANE];
    __shared__ __attribute__((aligned(16))) float part[2][NW * G * 64 * 4];
    __shared__ __attribute__((aligned(16))) float b1s[H];
    __shared__ float b2s[C];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ntiles = (P + 31) >> 5;
    // range scale of the split-fp16 operands (ffno_device.h "range words"): the staged rows are multiplied by a power of two
    // derived on the device from the range word of the input (bound = 2 max|in|: the sum of two addends -> 2^kFfRangeTarget,
    // which leaves 2^(16 - kFfRangeTarget) for the growth through the first product chain), the outputs divided again (exact)
    const float gscale = (S::SCALED && in_amax) ? range_scale(*in_amax, 1, kFfRangeTarget) : 1.f;
    const float rgscale = 1.f / gscale;
    float omax = 0.f;                                   // max |out| over the rows this thread stores (-> out_amax)
    __shared__ float rfold[NW];

    using Frag = typename S::Frag;
    Frag A1[CPW][KS], A2[CPW][CTO][2];
    FFNO_UNROLL
    for (int ch = 0; ch < CPW; ++ch) {
        const int q = wave * CPW + ch;
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) A1[ch][st] = load_frag_s<S>(pk1, q * KS + st, lane);
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            FFNO_UNROLL
            for (int s2 = 0; s2 < 2; ++s2) A2[ch][mt][s2] = load_frag_s<S>(pk2, (q * CTO + mt) * 2 + s2, lane);
        }
    }
    if (!BWD) {
        for (int e = tid; e < H; e += F::NT) b1s[e] = bias1[e] * gscale;
        for (int e = tid; e < C; e += F::NT) b2s[e] = bias2[e];
    }

    // staging map: float4 number f = tid + v * NT of the [32][C] tile.  Loads run TWO tiles ahead of the matrix work: the
    // raw rows of tile t + 2 are requested at the top of iteration t (pA / pB), summed and handed to the staging registers
    // (nS) at the top of iteration t + 1, and split into LDS at its end -- a full iteration (~3 us) between request and use,
    // so an HBM round trip under load (~2 us) is never on the critical path of a tile (profiles/r01_v4_ablation.md: 8.5 of
    // 51 us were exposed staging latency with one tile of distance).
    float4 nS[NV];
    typename ST::Raw4 pA[NV], pB[NV];      // (raw words: widened in consume(), an iteration after the request)
    auto gload_raw = [&](int tile) {
        FFNO_UNROLL
        for (int v = 0; v < NV; ++v) {
            const int f = tid + v * F::NT;
            const long px = (long)tile * 32 + f / (C / 4);
            pA[v] = pB[v] = ST::zero4();
            if (px < P) {
                const long off = px * C + 4 * (f % (C / 4));
                pA[v] = ST::ldr4(in + off);
                if (in2) pB[v] = ST::ldr4(in2 + off);
            }
        }
    };
    auto consume = [&](int tile) {       // raw rows -> staging registers (the input may be the sum of two tensors: the two
        FFNO_UNROLL                      // spectral branches ran side by side; the sum is optionally written back)
        for (int v = 0; v < NV; ++v) {
            nS[v] = ST::w4(pA[v]);
            if (in2) {
                const float4 b = ST::w4(pB[v]);
                nS[v].x += b.x, nS[v].y += b.y, nS[v].z += b.z, nS[v].w += b.w;
                nS[v] = st_rnd4<ST>(nS[v]);
                if (sum_out) {
                    const int f = tid + v * F::NT;
                    const long px = (long)tile * 32 + f / (C / 4);
                    if (px < P) ST::st4(sum_out + px * C + 4 * (f % (C / 4)), nS[v]);
                }
            }
            if (S::SCALED) nS[v].x *= gscale, nS[v].y *= gscale, nS[v].z *= gscale, nS[v].w *= gscale;
        }
    };
    auto stage = [&](int buf) {
        FFNO_UNROLL
        for (int v = 0; v < NV; ++v) {
            const int f = tid + v * F::NT;
            stage4_s<S>(sp[buf], F::PPLANE, (f / (C / 4)) * F::PROW + (f % (C / 4)) * 8, nS[v].x, nS[v].y, nS[v].z, nS[v].w);
        }
    };
    // reduce the NW partial tiles of `tile` (this wave owns float4 groups [wave*GPW, +GPW)) and store the output rows
    typename ST::Raw4 rres[GPW], rnext[GPW];
    auto rload = [&](int tile) {   // residual rows of the tile, requested a whole iteration before its reduction
        const long px = (long)tile * 32 + j;
        FFNO_UNROLL
        for (int u = 0; u < GPW; ++u) {
            const int gi = wave * GPW + u;
            rnext[u] = ST::zero4();
            if (!BWD && resid && px < P)
                rnext[u] = ST::ldr4(resid + px * C + 32 * (gi >> 2) + 8 * (gi & 3) + 4 * half);
        }
    };
    auto reduce = [&](int tile, int buf) {
        const long px = (long)tile * 32 + j;
        FFNO_UNROLL
        for (int u = 0; u < GPW; ++u) {
            const int gi = wave * GPW + u;
            float4 acc = *reinterpret_cast<const float4*>(&part[buf][((0 * G + gi) * 64 + lane) * 4]);
            FFNO_UNROLL
            for (int w = 1; w < NW; ++w) {
                const float4 t = *reinterpret_cast<const float4*>(&part[buf][((w * G + gi) * 64 + lane) * 4]);
                acc.x += t.x;
                acc.y += t.y;
                acc.z += t.z;
                acc.w += t.w;
            }
            const int c0 = 32 * (gi >> 2) + 8 * (gi & 3) + 4 * half;
            if (S::SCALED) acc.x *= rgscale, acc.y *= rgscale, acc.z *= rgscale, acc.w *= rgscale;
            if (!BWD) {
                const float4 rr = ST::w4(rres[u]);
                acc.x += b2s[c0] + rr.x;
                acc.y += b2s[c0 + 1] + rr.y;
                acc.z += b2s[c0 + 2] + rr.z;
                acc.w += b2s[c0 + 3] + rr.w;
            }
            if (px < P) {
                ST::st4(out + px * C + c0, acc);
                acc = st_rnd4<ST>(acc);
                omax = fmaxf(fmaxf(omax, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
            }
        }
    };

    if ((int)blockIdx.x < ntiles) {
        gload_raw(blockIdx.x);
        consume(blockIdx.x);
        stage(0);
        if ((int)(blockIdx.x + gridDim.x) < ntiles) gload_raw(blockIdx.x + gridDim.x);
    }
    uint32_t bits_next = 0;
    if (BWD && (int)blockIdx.x < ntiles)
        bits_next = reinterpret_cast<const uint16_t*>(mask)[((long)blockIdx.x * NW + wave) * 64 + lane];
    __syncthreads();

    // Software pipeline: iteration t multiplies tile t and reduces + stores tile t-1.  The two waves that share a SIMD
    // (w and w + NW/2) place that reduction at opposite ends of the iteration, so between two barriers one of them
    // runs  [reduce | GEMM1 | epilogue | GEMM2]  and the other  [GEMM1 | epilogue | GEMM2 | reduce]: the VALU/LDS
    // phases of one fall on the MFMA phases of the other instead of both queueing for the same pipe in lockstep.
    const bool early = wave < NW / 2 || NW == 1;
    int buf = 0, prev = -1;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const int nt = tile + gridDim.x, nt2 = nt + gridDim.x;
        if (nt < ntiles) consume(nt);            // requested one iteration ago
        rload(tile);                             // consumed by this tile's reduction, one iteration from now
        static_assert(CPW == 1, "one 16-bit sign word per (tile, wave, lane)");
        uint16_t* mp = mask ? reinterpret_cast<uint16_t*>(mask) + ((long)tile * NW + wave) * 64 + lane : nullptr;
        uint32_t bits = 0;
        if (BWD) {                               // this tile's sign word arrived during the previous iteration
            bits = bits_next;
            if (nt < ntiles) bits_next = reinterpret_cast<const uint16_t*>(mask)[((long)nt * NW + wave) * 64 + lane];
        }
        // youngest requests of the iteration (memory operations retire in order: whoever waits for the residual rows or the
        // sign word at the end of this iteration must not wait for these)
        if (nt2 < ntiles) gload_raw(nt2);

        // GEMM1: this wave's hidden chunks for the 32 pixels of the tile
        f32x16 d[CPW];
        {
            f32x16 dc[CPW];
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) d[ch] = zero16(), dc[ch] = zero16();
            FFNO_UNROLL
            for (int st = 0; st < KS; ++st) {
                const Frag b = lds_frag_s<S>(sp[buf], F::PPLANE, j * F::PROW + 32 * st + 16 * half);
                FFNO_UNROLL
                for (int ch = 0; ch < CPW; ++ch) S::mma(A1[ch][st], b, d[ch], dc[ch]);
            }
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) S::fold(d[ch], dc[ch]);
        }
        if (early && prev >= 0) reduce(prev, buf ^ 1);
        // epilogue + GEMM2 partial over the wave's hidden rows, k order = D-fragment order
        f32x16 o[CTO], oc[CTO];
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) o[mt] = zero16(), oc[mt] = zero16();
        FFNO_UNROLL
        for (int ch = 0; ch < CPW; ++ch) {
            if (BWD) {
                FFNO_UNROLL
                for (int r = 0; r < 16; ++r) d[ch][r] = u2f(f2u(d[ch][r]) & bit_mask(bits, 15 - r));
            } else {
                FFNO_UNROLL
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(&b1s[32 * (wave * CPW + ch) + 8 * g + 4 * half]);
                    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
                    FFNO_UNROLL
                    for (int i = 0; i < 4; ++i) {
                        const float v = fmaxf(d[ch][4 * g + i] + bb[i], 0.f);
                        d[ch][4 * g + i] = v;
                        bits = push_sign(bits, 0u - f2u(v));   // msb(-bits(v)) = [v > 0]; element r ends up at bit 15 - r
                    }
                }
            }
            Frag hb[2];
            hb[0] = S::split8(d[ch][0], d[ch][1], d[ch][2], d[ch][3], d[ch][4], d[ch][5], d[ch][6], d[ch][7]);
            hb[1] = S::split8(d[ch][8], d[ch][9], d[ch][10], d[ch][11], d[ch][12], d[ch][13], d[ch][14], d[ch][15]);
            FFNO_UNROLL
            for (int mt = 0; mt < CTO; ++mt) {
                S::mma(A2[ch][mt][0], hb[0], o[mt], oc[mt]);
                S::mma(A2[ch][mt][1], hb[1], o[mt], oc[mt]);
            }
        }
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) S::fold(o[mt], oc[mt]);
        if (!BWD && mp) *mp = (uint16_t)bits;
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            FFNO_UNROLL
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(&part[buf][(((wave * G) + mt * 4 + g) * 64 + lane) * 4]) =
                    make_float4(o[mt][4 * g], o[mt][4 * g + 1], o[mt][4 * g + 2], o[mt][4 * g + 3]);
        }
        if (!early && prev >= 0) reduce(prev, buf ^ 1);
        FFNO_UNROLL
        for (int u = 0; u < GPW; ++u) rres[u] = rnext[u];
        if (nt < ntiles) stage(buf ^ 1);
        prev = tile;
        __syncthreads();
    }
    if (prev >= 0) reduce(prev, buf ^ 1);
    if (out_amax) range_fold(omax, rfold, NW, out_amax);
}

// ---- forward / backward-data, ONE WAVE PER PIXEL TILE ("wave tiles", split-fp16) ---------------------------------------------
// Same operators, packs, sign-bit layout and range handling as ffx_chain_kernel.  What differs is who owns what.  There the
// eight waves of a workgroup share ONE 32-pixel tile, each holding the weights of a 32-row hidden chunk in registers: per tile a
// wave runs 12 + 12 MFMAs between an LDS staging pass, a 64 KiB exchange of partial outputs and barriers -- the tile is too
// small for eight waves, the kernels run at a fifth of the matrix rate and their time follows the synchronisation, not the
// instruction count (round 2-4 measurements).  Here a wave owns a WHOLE tile: it walks all H / 32 hidden chunks
// itself (GEMM1 chunk -> epilogue -> GEMM2 accumulation, 24 MFMAs per chunk, 192 per tile at 64 / 256), the weights of BOTH
// linear maps sit in LDS for the whole launch (128 KiB of fp16x2 fragments, copied once per workgroup, read conflict-free as
// ds_read_b128).  The MFMA fragments want a pixel per lane; read that way from global memory a load touches 32 rows with 16
// bytes each (measured: a launch that ONLY moves the tiles like that takes 29-40 us, the whole shared-tile kernel 41-45), so
// the rows travel in 64-byte segments (four lanes per row of a 16-channel block) and a 2.5 KiB staging area per wave turns a
// block into the fragment layout and back.  No exchange between waves and no barrier after the weight copy: the eight waves of
// a CU drift apart and fill each other's memory waits.
//   GEMM1 (per chunk): main + correction accumulators, folded per chunk -- product for product the h of ffx_chain_kernel and
//   of the weight-gradient kernel's recomputation (same ReLU decisions).  GEMM2: ONE main + correction pair per output tile
//   over all 16 k-steps, folded once (ffx_chain_kernel folds per chunk and adds the eight partial tiles: same products, other
//   rounding of the sum -- results agree to fp32 rounding, not bit for bit; bf16 twin == bf16(this kernel at fp32 storage)).
template <int C, int H, bool BWD, class ST = StF32>
__global__ __launch_bounds__(512) FFNO_WAVES_PER_SIMD(2) void ffw_chain_kernel(const typename ST::T* __restrict__ in,
                                                                              const typename ST::T* __restrict__ in2,
                                                                              typename ST::T* sum_out,
                                                                              const typename ST::T* resid,
                                                                              const u32x4* __restrict__ pk1,
                                                                              const float* __restrict__ bias1,
                                                                              const u32x4* __restrict__ pk2,
                                                                              const float* __restrict__ bias2,
                                                                              typename ST::T* out, uint32_t* mask, int P,
                                                                              const unsigned* in_amax, unsigned* out_amax) {
    using S = SplitHf2;
    constexpr int NCH = H / 32, KS = C / 16, CTO = C / 32, NWV = 8;
    constexpr int NF1 = NCH * KS, NF2 = NCH * CTO * 2;      // fragments of the two packs: two planes of 64 x 16 B each
    constexpr int SROW = 20;                                // floats per staged row: 16 channels + 4 of padding (bank spread)
    FFNO_DYN_SMEM(smem);
    u32x4* w1 = reinterpret_cast<u32x4*>(smem);
    u32x4* w2 = w1 + NF1 * 2 * 64;
    float* b1s = reinterpret_cast<float*>(w2 + NF2 * 2 * 64);
    float* b2s = b1s + H;
    float* stg_all = b2s + C;                               // NWV x [32 rows][SROW]: one 16-channel block of a tile per wave
    __shared__ float rfold[NWV];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ntiles = (P + 31) >> 5;
    const float gscale = in_amax ? range_scale(*in_amax, 1, kFfRangeTarget) : 1.f;
    const float rgscale = 1.f / gscale;
    float omax = 0.f;

    for (int i = tid; i < NF1 * 2 * 64; i += NWV * 64) w1[i] = pk1[i];
    for (int i = tid; i < NF2 * 2 * 64; i += NWV * 64) w2[i] = pk2[i];
    if (!BWD) {
        for (int e = tid; e < H; e += NWV * 64) b1s[e] = bias1[e] * gscale;
        for (int e = tid; e < C; e += NWV * 64) b2s[e] = bias2[e];
    }
    __syncthreads();

    // Two lane maps of a 16-channel block of the tile ([32 rows][16 channels], 64 contiguous bytes per row in global memory):
    //   memory map : lane l <-> row 16 i + (l >> 2), channels 4 (l & 3) + (0..3), i = 0, 1   (four lanes per 64-byte row segment:
    //                what the global loads / stores use)
    //   operand map: lane (j, half) <-> row j, channels 8 half + (0..7)                       (the MFMA B / D fragments)
    // The wave's staging rows in LDS convert one into the other, block by block (LDS operations of a wave execute in order, so
    // a block may be overwritten as soon as the reads of the previous one are issued; wave_sync is a compiler fence there).
    float* stg = stg_all + wave * (32 * SROW);
    const int mrow = lane >> 2, mq = lane & 3;
    float* stg_m0 = stg + mrow * SROW + 4 * mq;             // memory map, i = 0 (i = 1: + 16 rows)
    float* stg_o = stg + j * SROW + 8 * half;               // operand map
    const uint16_t* mrd = reinterpret_cast<const uint16_t*>(mask);
    uint16_t* mwr = reinterpret_cast<uint16_t*>(mask);
    FFNO_NOUNROLL
    for (int tile = (int)blockIdx.x * NWV + wave; tile < ntiles; tile += (int)gridDim.x * NWV) {
        // rows of this lane in the memory map; rows past the end re-read the last row and are zeroed / not stored
        long mpx[2];
        bool mok[2];
        long moff[2];
        FFNO_UNROLL
        for (int i = 0; i < 2; ++i) {
            mpx[i] = (long)tile * 32 + 16 * i + mrow;
            mok[i] = mpx[i] < P;
            moff[i] = (mok[i] ? mpx[i] : (long)P - 1) * C + 4 * mq;
        }
        typename ST::Raw4 ra[KS][2], rb[KS][2];
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {
            FFNO_UNROLL
            for (int i = 0; i < 2; ++i) ra[st][i] = ST::ldr4(in + moff[i] + 16 * st);
        }
        if (in2) {
            FFNO_UNROLL
            for (int st = 0; st < KS; ++st) {
                FFNO_UNROLL
                for (int i = 0; i < 2; ++i) rb[st][i] = ST::ldr4(in2 + moff[i] + 16 * st);
            }
        }
        // requested now, used after the last chunk (forward: the residual rows, memory map) / chunk by chunk (backward: the
        // sign words of ALL chunks, two per register): nothing inside the chunk loop waits for global memory
        typename ST::Raw4 rres[KS][2];
        uint32_t mw[NCH / 2];
        if (!BWD) {
            if (resid) {
                FFNO_UNROLL
                for (int st = 0; st < KS; ++st) {
                    FFNO_UNROLL
                    for (int i = 0; i < 2; ++i) rres[st][i] = ST::ldr4(resid + moff[i] + 16 * st);
                }
            }
        } else {
            static_assert(NCH % 2 == 0, "sign words are held in pairs");
            FFNO_UNROLL
            for (int q = 0; q < NCH / 2; ++q)
                mw[q] = (uint32_t)mrd[((long)tile * NCH + 2 * q) * 64 + lane] |
                        ((uint32_t)mrd[((long)tile * NCH + 2 * q + 1) * 64 + lane] << 16);
        }
        // the tile's rows as B operands: k-step st <-> channels 16 st + 8 half + (0..7) of pixel j
        Hf2 b[KS];
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {
            FFNO_UNROLL
            for (int i = 0; i < 2; ++i) {
                float4 v = ST::w4(ra[st][i]);
                if (in2) {
                    const float4 t = ST::w4(rb[st][i]);
                    v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
                    v = st_rnd4<ST>(v);
                    if (sum_out && mok[i]) ST::st4(sum_out + moff[i] + 16 * st, v);
                }
                const float sc = mok[i] ? gscale : 0.f;
                v.x *= sc, v.y *= sc, v.z *= sc, v.w *= sc;
                *reinterpret_cast<float4*>(stg_m0 + 16 * i * SROW) = v;
            }
            plat::wave_sync();
            const float4 v0 = *reinterpret_cast<const float4*>(stg_o), v1 = *reinterpret_cast<const float4*>(stg_o + 4);
            plat::wave_sync();
            b[st] = split2_8(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w);
        }
        f32x16 o[CTO], oc[CTO];
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) o[mt] = zero16(), oc[mt] = zero16();
        FFNO_NOUNROLL
        for (int ch = 0; ch < NCH; ++ch) {
            uint32_t bits = 0;
            if (BWD) {
                uint32_t w = mw[0];
                FFNO_UNROLL
                for (int q = 1; q < NCH / 2; ++q) w = (ch >> 1) == q ? mw[q] : w;
                bits = (ch & 1) ? (w >> 16) : (w & 0xffffu);
            }
            // GEMM1: the chunk's 32 hidden rows for the tile's 32 pixels
            f32x16 d = zero16(), dc = zero16();
            FFNO_UNROLL
            for (int st = 0; st < KS; ++st) {
                Hf2 a;
                a.hi = w1[((ch * KS + st) * 2 + 0) * 64 + lane];
                a.lo = w1[((ch * KS + st) * 2 + 1) * 64 + lane];
                mfma_h2(a, b[st], d, dc);
            }
            S::fold(d, dc);
            if (BWD) {
                FFNO_UNROLL
                for (int r = 0; r < 16; ++r) d[r] = u2f(f2u(d[r]) & bit_mask(bits, 15 - r));
            } else {
                FFNO_UNROLL
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(&b1s[32 * ch + 8 * g + 4 * half]);
                    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
                    FFNO_UNROLL
                    for (int i = 0; i < 4; ++i) {
                        const float v = fmaxf(d[4 * g + i] + bb[i], 0.f);
                        d[4 * g + i] = v;
                        bits = push_sign(bits, 0u - f2u(v));   // element r ends up at bit 15 - r (ffx_chain_kernel's sign word)
                    }
                }
                if (mwr) mwr[((long)tile * NCH + ch) * 64 + lane] = (uint16_t)bits;
            }
            // GEMM2: the chunk's contribution to the output tile, k order = D-fragment order
            Hf2 hb[2];
            hb[0] = split2_8(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]);
            hb[1] = split2_8(d[8], d[9], d[10], d[11], d[12], d[13], d[14], d[15]);
            FFNO_UNROLL
            for (int mt = 0; mt < CTO; ++mt) {
                FFNO_UNROLL
                for (int s2 = 0; s2 < 2; ++s2) {
                    Hf2 a;
                    a.hi = w2[(((ch * CTO + mt) * 2 + s2) * 2 + 0) * 64 + lane];
                    a.lo = w2[(((ch * CTO + mt) * 2 + s2) * 2 + 1) * 64 + lane];
                    mfma_h2(a, hb[s2], o[mt], oc[mt]);
                }
            }
        }
        // output rows: accumulator group g of tile mt = channels 32 mt + 8 g + 4 half + (0..3) of pixel j; block (mt, gp) =
        // channels 32 mt + 16 gp + (0..15): groups 2 gp and 2 gp + 1 of both halves, back through the staging rows
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            S::fold(o[mt], oc[mt]);
            FFNO_UNROLL
            for (int gp = 0; gp < 2; ++gp) {
                const int blk = 2 * mt + gp;      // = the k-step index of the same 16 channels
                FFNO_UNROLL
                for (int u = 0; u < 2; ++u) {
                    const int g = 2 * gp + u;
                    *reinterpret_cast<float4*>(stg + j * SROW + 8 * u + 4 * half) =
                        make_float4(o[mt][4 * g], o[mt][4 * g + 1], o[mt][4 * g + 2], o[mt][4 * g + 3]);
                }
                plat::wave_sync();
                float4 acc[2];
                FFNO_UNROLL
                for (int i = 0; i < 2; ++i) acc[i] = *reinterpret_cast<const float4*>(stg_m0 + 16 * i * SROW);
                plat::wave_sync();
                FFNO_UNROLL
                for (int i = 0; i < 2; ++i) {
                    const int c0 = 16 * blk + 4 * mq;
                    acc[i].x *= rgscale, acc[i].y *= rgscale, acc[i].z *= rgscale, acc[i].w *= rgscale;
                    if (!BWD) {
                        float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (resid) rr = ST::w4(rres[blk][i]);
                        acc[i].x += b2s[c0] + rr.x;
                        acc[i].y += b2s[c0 + 1] + rr.y;
                        acc[i].z += b2s[c0 + 2] + rr.z;
                        acc[i].w += b2s[c0 + 3] + rr.w;
                    }
                    if (mok[i]) {
                        ST::st4(out + moff[i] + 16 * blk, acc[i]);
                        const float4 t = st_rnd4<ST>(acc[i]);
                        omax = fmaxf(fmaxf(omax, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
                    }
                }
            }
        }
    }
    if (out_amax) range_fold(omax, rfold, NWV, out_amax);
}

// ---- forward / backward-data, role-split schedule ---------------------------------------------------------------------
// Same arithmetic, operands and results as ffx_chain_kernel (bit-identical: same products, same summation order), other
// schedule.  In ffx_chain_kernel the two waves that share a SIMD (w and w + NW/2) run the same code in phase: after every
// barrier both want the matrix pipe (GEMM1), then both want the vector ALU (bias / ReLU / three-way split), then both the
// matrix pipe again -- the pipes are used one after the other and the phases ADD (measured: 7600 cycles per tile against
// 3072 cycles of MFMA per SIMD; the matrix pipe is busy 40 % of the launch).  Here the halves take ROLES that keep them one
// slot apart, with a barrier after every slot so that they stay apart:
//
//     slot            1                2                  3                    4
//     waves 0..NW/2   GEMM1(t)         epilogue(t)        GEMM2(t) -> part     stage tile t+1 (split -> LDS)
//     waves NW/2..    reduce(t-1)      GEMM1(t)           epilogue(t)          GEMM2(t) -> part
//
// so every slot pairs a matrix segment on one wave with a vector / LDS segment on its SIMD partner.  The first half
// stages all input tiles, the second half reduces and stores all output tiles; the partial-output exchange needs ONE
// buffer (reduce(t-1) is over before the first half writes part(t)), i.e. 64 KiB of LDS less.
template <int C, int H, bool BWD, class S, class ST = StF32>
__global__ __launch_bounds__((FxCfg<C, H>::NT)) void ffx_chain_rs_kernel(const typename ST::T* __restrict__ in,
                                                                        const typename ST::T* __restrict__ in2,
                                                                        typename ST::T* sum_out,
                                                                        const typename ST::T* resid,
                                                                        const u32x4* __restrict__ pk1,
                                                                        const float* __restrict__ bias1,
                                                                        const u32x4* __restrict__ pk2,
                                                                        const float* __restrict__ bias2, typename ST::T* out,
                                                                        uint32_t* mask, int P, const unsigned* in_amax, unsigned* out_amax) {
    using F = FxCfg<C, H>;
    constexpr int NW = F::NW, KS = F::KS, CTO = F::CTO, G = F::G;
    constexpr int NWA = NW / 2;                      // waves per role
    constexpr int NTA = NWA * 64;                    // threads that stage
    constexpr int NVA = (32 * C / 4) / NTA;          // float4 per staging thread and tile
    constexpr int GPB = G / NWA;                     // float4 groups per reducing wave
    static_assert(NW >= 2 && NWA * 2 == NW && NVA * NTA * 4 == 32 * C && GPB * NWA == G, "role split");
    __shared__ __attribute__((aligned(16))) char sp[2][S::NP * F::PPLANE];
    __shared__ __attribute__((aligned(16))) float part[NW * G * 64 * 4];
    __shared__ __attribute__((aligned(16))) float b1s[H];
    __shared__ float b2s[C];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ntiles = (P + 31) >> 5;
    // range scale of the split-fp16 operands (ffno_device.h "range words"): the staged rows are multiplied by a power of two
    // derived on the device from the range word of the input (bound = 2 max|in|: the sum of two addends -> 2^kFfRangeTarget,
    // which leaves 2^(16 - kFfRangeTarget) for the growth through the first product chain), the outputs divided again (exact)
    const float gscale = (S::SCALED && in_amax) ? range_scale(*in_amax, 1, kFfRangeTarget) : 1.f;
    const float rgscale = 1.f / gscale;
    float omax = 0.f;                                   // max |out| over the rows this thread stores (-> out_amax)
    __shared__ float rfold[NW];
    const bool first = wave < NWA;                   // role: first half stages, second half reduces
    const int rw = wave - NWA;                       // index inside the reducing half

    using Frag = typename S::Frag;
    Frag A1[KS], A2[CTO][2];
    FFNO_UNROLL
    for (int st = 0; st < KS; ++st) A1[st] = load_frag_s<S>(pk1, wave * KS + st, lane);
    FFNO_UNROLL
    for (int mt = 0; mt < CTO; ++mt) {
        FFNO_UNROLL
        for (int s2 = 0; s2 < 2; ++s2) A2[mt][s2] = load_frag_s<S>(pk2, (wave * CTO + mt) * 2 + s2, lane);
    }
    if (!BWD) {
        for (int e = tid; e < H; e += F::NT) b1s[e] = bias1[e] * gscale;
        for (int e = tid; e < C; e += F::NT) b2s[e] = bias2[e];
    }

    // ---- staging half: raw rows requested two tiles ahead (qA / qB), summed into pS one tile ahead, split into LDS in slot 4 ----
    float4 pS[NVA];
    typename ST::Raw4 qA[NVA], qB[NVA];      // (raw words: widened in consume(), an iteration after the request)
    // float4 number (within the [32][C] tile) of this thread's staging slot v.  bf16 storage: the two slots are neighbours, so
    // that one 16-byte access covers both (8-byte accesses run at 0.5-0.7 of the 16-byte rate, MI355X_MICROARCH.md)
    static_assert(!ST::BF16 || (NVA == 2 && GPB == 2), "bf16 storage: 16-byte accesses pair two float4 groups");
    auto fmap = [&](int v) { return ST::BF16 ? (tid / (C / 8)) * (C / 4) + 2 * (tid % (C / 8)) + v : tid + v * NTA; };
    auto gload_raw = [&](int tile) {
        if constexpr (ST::BF16) {
            const int f = fmap(0);
            const long px = (long)tile * 32 + f / (C / 4);
            qA[0] = qA[1] = qB[0] = qB[1] = ST::zero4();
            if (px < P) {
                const long off = px * C + 4 * (f % (C / 4));
                const u32x4 a = *reinterpret_cast<const u32x4*>(in + off);
                qA[0] = make_uint2(a[0], a[1]), qA[1] = make_uint2(a[2], a[3]);
                if (in2) {
                    const u32x4 b = *reinterpret_cast<const u32x4*>(in2 + off);
                    qB[0] = make_uint2(b[0], b[1]), qB[1] = make_uint2(b[2], b[3]);
                }
            }
        } else {
            FFNO_UNROLL
            for (int v = 0; v < NVA; ++v) {
                const int f = fmap(v);
                const long px = (long)tile * 32 + f / (C / 4);
                qA[v] = qB[v] = ST::zero4();
                if (px < P) {
                    const long off = px * C + 4 * (f % (C / 4));
                    qA[v] = ST::ldr4(in + off);
                    if (in2) qB[v] = ST::ldr4(in2 + off);
                }
            }
        }
    };
    auto consume = [&](int tile) {      // raw rows -> pS (the input may be the sum of two tensors; the sum is optionally stored)
        FFNO_UNROLL
        for (int v = 0; v < NVA; ++v) {
            pS[v] = ST::w4(qA[v]);
            if (in2) {
                const float4 b = ST::w4(qB[v]);
                pS[v].x += b.x, pS[v].y += b.y, pS[v].z += b.z, pS[v].w += b.w;
                pS[v] = st_rnd4<ST>(pS[v]);
                const int f = fmap(v);
                const long px = (long)tile * 32 + f / (C / 4);
                if constexpr (!ST::BF16)
                    if (sum_out && px < P) ST::st4(sum_out + px * C + 4 * (f % (C / 4)), pS[v]);
            }
        }
        if constexpr (ST::BF16) {
            const int f = fmap(0);
            const long px = (long)tile * 32 + f / (C / 4);
            if (in2 && sum_out && px < P) {
                u32x4 w;
                w[0] = plat::pack_bf16(pS[0].x, pS[0].y), w[1] = plat::pack_bf16(pS[0].z, pS[0].w);
                w[2] = plat::pack_bf16(pS[1].x, pS[1].y), w[3] = plat::pack_bf16(pS[1].z, pS[1].w);
                *reinterpret_cast<u32x4*>(sum_out + px * C + 4 * (f % (C / 4))) = w;
            }
        }
        if (S::SCALED) {
            FFNO_UNROLL
            for (int v = 0; v < NVA; ++v) pS[v].x *= gscale, pS[v].y *= gscale, pS[v].z *= gscale, pS[v].w *= gscale;
        }
    };
    auto stage = [&](int buf) {
        FFNO_UNROLL
        for (int v = 0; v < NVA; ++v) {
            const int f = fmap(v);
            stage4_s<S>(sp[buf], F::PPLANE, (f / (C / 4)) * F::PROW + (f % (C / 4)) * 8, pS[v].x, pS[v].y, pS[v].z, pS[v].w);
        }
    };
    // ---- reducing half: residual rows one tile ahead; reduce the NW partial tiles and store the output rows ----
    typename ST::Raw4 rres[GPB], rnext[GPB];
    // bf16 storage: the wave's two float4 groups are the halves of two 8-channel blocks; a lane of the lower half-wave reads and
    // writes the WHOLE first block of its pixel (16 bytes), a lane of the upper half the second, and the two halves exchange
    // the quads they own through v_permlane32_swap (rres / rnext then hold the raw 16 bytes: [0] = words 0, 1, [1] = words 2, 3)
    const int c16 = 32 * ((rw * GPB) >> 2) + 8 * (((rw * GPB) & 3) + half);
    auto rload = [&](int tile) {
        const long px = (long)tile * 32 + j;
        if constexpr (ST::BF16) {
            rnext[0] = rnext[1] = ST::zero4();
            if (!BWD && resid && px < P) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(resid + px * C + c16);
                rnext[0] = make_uint2(a[0], a[1]), rnext[1] = make_uint2(a[2], a[3]);
            }
            return;
        }
        FFNO_UNROLL
        for (int u = 0; u < GPB; ++u) {
            const int gi = rw * GPB + u;
            rnext[u] = ST::zero4();
            if (!BWD && resid && px < P)
                rnext[u] = ST::ldr4(resid + px * C + 32 * (gi >> 2) + 8 * (gi & 3) + 4 * half);
        }
    };
    auto reduce = [&](int tile) {
        const long px = (long)tile * 32 + j;
        if constexpr (ST::BF16) {      // words 0, 1 of both halves -> the lower quads' owners; words 2, 3 -> the upper quads'
            plat::swap_halves(rres[0].x, rres[1].x);
            plat::swap_halves(rres[0].y, rres[1].y);
        }
        uint2 pk16[2];
        FFNO_UNROLL
        for (int u = 0; u < GPB; ++u) {
            const int gi = rw * GPB + u;
            float4 acc = *reinterpret_cast<const float4*>(&part[((0 * G + gi) * 64 + lane) * 4]);
            FFNO_UNROLL
            for (int w = 1; w < NW; ++w) {
                const float4 t = *reinterpret_cast<const float4*>(&part[((w * G + gi) * 64 + lane) * 4]);
                acc.x += t.x;
                acc.y += t.y;
                acc.z += t.z;
                acc.w += t.w;
            }
            const int c0 = 32 * (gi >> 2) + 8 * (gi & 3) + 4 * half;
            if (S::SCALED) acc.x *= rgscale, acc.y *= rgscale, acc.z *= rgscale, acc.w *= rgscale;
            if (!BWD) {
                const float4 rr = ST::w4(rres[u]);
                acc.x += b2s[c0] + rr.x;
                acc.y += b2s[c0 + 1] + rr.y;
                acc.z += b2s[c0 + 2] + rr.z;
                acc.w += b2s[c0 + 3] + rr.w;
            }
            if constexpr (ST::BF16) pk16[u & 1] = make_uint2(plat::pack_bf16(acc.x, acc.y), plat::pack_bf16(acc.z, acc.w));
            if (px < P) {
                if constexpr (!ST::BF16) ST::st4(out + px * C + c0, acc);
                acc = st_rnd4<ST>(acc);
                omax = fmaxf(fmaxf(omax, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
            }
        }
        if constexpr (ST::BF16) {
            plat::swap_halves(pk16[0].x, pk16[1].x);
            plat::swap_halves(pk16[0].y, pk16[1].y);
            if (px < P) {
                u32x4 w;
                w[0] = pk16[0].x, w[1] = pk16[0].y, w[2] = pk16[1].x, w[3] = pk16[1].y;
                *reinterpret_cast<u32x4*>(out + px * C + c16) = w;
            }
        }
    };

    // ---- the three compute segments of a wave (its hidden chunk of the current tile) ----
    f32x16 d;
    Frag hb[2];
    uint32_t bits = 0, bits_next = 0;
    auto gemm1 = [&](int buf) {
        f32x16 dc = zero16();
        d = zero16();
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {
            const Frag b = lds_frag_s<S>(sp[buf], F::PPLANE, j * F::PROW + 32 * st + 16 * half);
            S::mma(A1[st], b, d, dc);
        }
        S::fold(d, dc);
    };
    auto epilogue = [&](int tile) {
        uint16_t* mp = mask ? reinterpret_cast<uint16_t*>(mask) + ((long)tile * NW + wave) * 64 + lane : nullptr;
        if (BWD) {
            bits = bits_next;                          // this tile's sign word was requested during the previous tile
            const int nt = tile + gridDim.x;
            if (nt < ntiles) bits_next = reinterpret_cast<const uint16_t*>(mask)[((long)nt * NW + wave) * 64 + lane];
            FFNO_UNROLL
            for (int r = 0; r < 16; ++r) d[r] = u2f(f2u(d[r]) & bit_mask(bits, 15 - r));
        } else {
            bits = 0;
            FFNO_UNROLL
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(&b1s[32 * wave + 8 * g + 4 * half]);
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
                FFNO_UNROLL
                for (int i = 0; i < 4; ++i) {
                    const float v = fmaxf(d[4 * g + i] + bb[i], 0.f);
                    d[4 * g + i] = v;
                    bits = push_sign(bits, 0u - f2u(v));   // msb(-bits(v)) = [v > 0]; element r ends up at bit 15 - r
                }
            }
            if (mp) *mp = (uint16_t)bits;
        }
        hb[0] = S::split8(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]);
        hb[1] = S::split8(d[8], d[9], d[10], d[11], d[12], d[13], d[14], d[15]);
    };
    auto gemm2 = [&]() {
        f32x16 o[CTO];
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            f32x16 oc = zero16();
            o[mt] = zero16();
            S::mma(A2[mt][0], hb[0], o[mt], oc);
            S::mma(A2[mt][1], hb[1], o[mt], oc);
            S::fold(o[mt], oc);
        }
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            FFNO_UNROLL
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(&part[(((wave * G) + mt * 4 + g) * 64 + lane) * 4]) =
                    make_float4(o[mt][4 * g], o[mt][4 * g + 1], o[mt][4 * g + 2], o[mt][4 * g + 3]);
        }
    };

    const int t0 = blockIdx.x, gs = gridDim.x;
    if (first) {
        if (t0 < ntiles) {
            gload_raw(t0);
            consume(t0);
            stage(0);
        }
        if (t0 + gs < ntiles) {
            gload_raw(t0 + gs);
            consume(t0 + gs);
        }
        if (t0 + 2 * gs < ntiles) gload_raw(t0 + 2 * gs);
    } else if (t0 < ntiles) {
        rload(t0);
    }
    if (BWD && t0 < ntiles) bits_next = reinterpret_cast<const uint16_t*>(mask)[((long)t0 * NW + wave) * 64 + lane];
    __syncthreads();

    int buf = 0, prev = -1;
    for (int tile = t0; tile < ntiles; tile += gs, buf ^= 1) {
        const int nt = tile + gs;
        // slot 1
        if (first) {
            gemm1(buf);
        } else {
            if (prev >= 0) reduce(prev);
            FFNO_UNROLL
            for (int u = 0; u < GPB; ++u) rres[u] = rnext[u];
            if (nt < ntiles) rload(nt);
        }
        __syncthreads();
        // slot 2
        if (first)
            epilogue(tile);
        else
            gemm1(buf);
        __syncthreads();
        // slot 3
        if (first)
            gemm2();
        else
            epilogue(tile);
        __syncthreads();
        // slot 4
        if (first) {
            if (nt < ntiles) stage(buf ^ 1);                       // pS holds tile nt
            if (nt + gs < ntiles) consume(nt + gs);                // requested one iteration ago
            if (nt + 2 * gs < ntiles) gload_raw(nt + 2 * gs);
        } else {
            gemm2();
        }
        prev = tile;
        __syncthreads();
    }
    if (!first && prev >= 0) reduce(prev);
    if (out_amax) range_fold(omax, rfold, NW, out_amax);
}

// ---- weight gradients with recomputed hidden activations --------------------------------------------------------------
// Per workgroup slice:  dW1^T[c][hid], dW2[c][hid], db1[hid], db2[c]  (dW1 is stored transposed; the reduce kernel
// un-transposes).  pk1 = pack1(W1), pk2t = pack1(W2^T) -- the A1 operands of the forward and backward chain kernels.
template <int C, int H, class S>
__global__ __launch_bounds__((FxCfg<C, H>::NT)) void ffx_wgrad_kernel(const float* __restrict__ s,
                                                                     const float* __restrict__ db,
                                                                     const u32x4* __restrict__ pk1,
                                                                     const float* __restrict__ bias1,
                                                                     const u32x4* __restrict__ pk2t,
                                                                     float* __restrict__ partial, int P, const unsigned* s_amax,
                                                                     const unsigned* db_amax) {
    using F = FxCfg<C, H>;
    constexpr int KS = F::KS, CTO = F::CTO, NV = F::NV, CPW = F::CPW;
    constexpr int NP = S::NP;
    // range scales of the two staged tensors (ffno_device.h "range words"): s like the forward chain kernel (the recomputed h
    // then carries fscale), db like the backward one (dh carries gscale); every output is divided by the scales of its factors
    const float fscale = (S::SCALED && s_amax) ? range_scale(*s_amax, 1, kFfRangeTarget) : 1.f;
    const float gscale = (S::SCALED && db_amax) ? range_scale(*db_amax, 1, kFfRangeTarget) : 1.f;
    constexpr int BUF = 2 * NP * F::PPLANE + 2 * NP * F::TPLANE;   // [sP x NP][dbP x NP][sT x NP][dbT x NP]
    constexpr int OFF_SP = 0, OFF_DP = NP * F::PPLANE, OFF_ST = 2 * NP * F::PPLANE, OFF_DT = 2 * NP * F::PPLANE + NP * F::TPLANE;
    using Frag = typename S::Frag;
    __shared__ __attribute__((aligned(16))) char lds[2][BUF];
    __shared__ float red[F::NT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ntiles = (P + 31) >> 5;

    Frag W1f[CPW][KS], W2f[CPW][KS];
    float b1v[CPW];
    FFNO_UNROLL
    for (int ch = 0; ch < CPW; ++ch) {
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {
            W1f[ch][st] = load_frag_s<S>(pk1, (wave * CPW + ch) * KS + st, lane);
            W2f[ch][st] = load_frag_s<S>(pk2t, (wave * CPW + ch) * KS + st, lane);
        }
        b1v[ch] = bias1[32 * (wave * CPW + ch) + j] * fscale;
    }

    // staging: pixel-major map as in the chain kernel; channel-major map: channel tc, pixel group tg (+ v * NT/C)
    const int tc = tid % C, tg = tid / C;
    float4 nSP[NV], nDP[NV], nST[NV], nDT[NV];
    float bs2 = 0.f;
    // addresses = uniform tile base (scalar registers) + constant 32-bit lane offsets (pixel-major float4 f; channel-major
    // gathers of 4 pixels with stride C: immediate offsets)
    auto gload = [&](int tile) {
        const float* st = s + (long)tile * (32 * C);
        const float* dt = db + (long)tile * (32 * C);
        const int rows = P - tile * 32;            // pixels of this tile that exist (>= 32: all)
        FFNO_UNROLL
        for (int v = 0; v < NV; ++v) {
            const int f = tid + v * F::NT;
            const int row = f / (C / 4);
            const unsigned offp = (unsigned)(row * C + 4 * (f % (C / 4)));
            nSP[v] = nDP[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rows) {
                nSP[v] = *reinterpret_cast<const float4*>(st + offp);
                nDP[v] = *reinterpret_cast<const float4*>(dt + offp);
            }
            const int r0 = 4 * (tg + v * (F::NT / C));
            const unsigned offt = (unsigned)(r0 * C + tc);
            float a[4], b[4];
            FFNO_UNROLL
            for (int i = 0; i < 4; ++i) {
                const bool ok = r0 + i < rows;
                a[i] = ok ? st[offt + i * C] : 0.f;
                b[i] = ok ? dt[offt + i * C] : 0.f;
            }
            nST[v] = make_float4(a[0], a[1], a[2], a[3]);
            nDT[v] = make_float4(b[0], b[1], b[2], b[3]);
            if (S::SCALED) {
                nSP[v].x *= fscale, nSP[v].y *= fscale, nSP[v].z *= fscale, nSP[v].w *= fscale;
                nST[v].x *= fscale, nST[v].y *= fscale, nST[v].z *= fscale, nST[v].w *= fscale;
                nDP[v].x *= gscale, nDP[v].y *= gscale, nDP[v].z *= gscale, nDP[v].w *= gscale;
                nDT[v].x *= gscale, nDT[v].y *= gscale, nDT[v].z *= gscale, nDT[v].w *= gscale;
            }
        }
    };
    auto stage = [&](int buf) {
        FFNO_UNROLL
        for (int v = 0; v < NV; ++v) {
            const int f = tid + v * F::NT;
            const int offp = (f / (C / 4)) * F::PROW + (f % (C / 4)) * 8;
            stage4_s<S>(lds[buf] + OFF_SP, F::PPLANE, offp, nSP[v].x, nSP[v].y, nSP[v].z, nSP[v].w);
            stage4_s<S>(lds[buf] + OFF_DP, F::PPLANE, offp, nDP[v].x, nDP[v].y, nDP[v].z, nDP[v].w);
            // pixel group grp = (s2 << 2) | (q << 1) | half  <->  local pixels 16 s2 + 8 q + 4 half + i  <->  k slot 4 q + i
            const int grp = tg + v * (F::NT / C);
            const int pos = 16 * (grp & 1) + 8 * (grp >> 2) + 4 * ((grp >> 1) & 1);
            const int offt = tc * F::TROW + 2 * pos;
            stage4_s<S>(lds[buf] + OFF_ST, F::TPLANE, offt, nST[v].x, nST[v].y, nST[v].z, nST[v].w);
            stage4_s<S>(lds[buf] + OFF_DT, F::TPLANE, offt, nDT[v].x, nDT[v].y, nDT[v].z, nDT[v].w);
            bs2 += (nDT[v].x + nDT[v].y) + (nDT[v].z + nDT[v].w);
        }
    };

    f32x16 acc1[CPW][CTO], acc2[CPW][CTO];
    float bs1[CPW];
    FFNO_UNROLL
    for (int ch = 0; ch < CPW; ++ch) {
        bs1[ch] = 0.f;
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) acc1[ch][mt] = zero16(), acc2[ch][mt] = zero16();
    }

    if ((int)blockIdx.x < ntiles) {
        gload(blockIdx.x);
        stage(0);
    }
    // The two waves that share a SIMD (w and w + NW/2) stage the next tile at opposite ends of an iteration: the "early"
    // half runs one tile ahead with its global loads and converts + writes the next tile BEFORE its MFMA phases, the other
    // half after them, so the split / LDS-write work of one wave falls on the matrix phases of the other instead of both
    // queueing for the same pipe in lockstep (same idea as the reduce placement in the chain kernel).
    const bool early = F::NW > 1 && wave < F::NW / 2;
    if (early && (int)(blockIdx.x + gridDim.x) < ntiles) gload(blockIdx.x + gridDim.x);
    __syncthreads();
    int buf = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const int nt = tile + gridDim.x;
        if (early) {
            if (nt < ntiles) stage(buf ^ 1);            // requested during the previous iteration
            if (nt + (int)gridDim.x < ntiles) gload(nt + gridDim.x);
        } else if (nt < ntiles) {
            gload(nt);
        }
        const char* L = lds[buf];
        // h^T[px][hid] = relu(s W1^T + b1), pixels on the D rows
        f32x16 d[CPW];
        uint32_t bits = 0;
        {
            f32x16 dc[CPW];
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) d[ch] = zero16(), dc[ch] = zero16();
            FFNO_UNROLL
            for (int st = 0; st < KS; ++st) {
                const Frag a = lds_frag_s<S>(L + OFF_SP, F::PPLANE, j * F::PROW + 32 * st + 16 * half);
                FFNO_UNROLL
                for (int ch = 0; ch < CPW; ++ch) S::mma(a, W1f[ch][st], d[ch], dc[ch]);
            }
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) S::fold(d[ch], dc[ch]);
        }
        Frag hb[CPW][2];
        FFNO_UNROLL
        for (int ch = 0; ch < CPW; ++ch) {
            FFNO_UNROLL
            for (int r = 0; r < 16; ++r) {
                const float v = d[ch][r] + b1v[ch];
                const bool pos = v > 0.f;
                d[ch][r] = pos ? v : 0.f;
                bits |= (pos ? 1u : 0u) << (16 * ch + r);
            }
            hb[ch][0] = S::split8(d[ch][0], d[ch][1], d[ch][2], d[ch][3], d[ch][4], d[ch][5], d[ch][6], d[ch][7]);
            hb[ch][1] = S::split8(d[ch][8], d[ch][9], d[ch][10], d[ch][11], d[ch][12], d[ch][13], d[ch][14], d[ch][15]);
        }
        // (the accumulators stay plain fp32 tiles between tiles: a product chain starts from them and is folded back)
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            f32x16 tc[CPW];
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) tc[ch] = zero16();
            FFNO_UNROLL
            for (int s2 = 0; s2 < 2; ++s2) {
                const Frag a = lds_frag_s<S>(L + OFF_DT, F::TPLANE, (32 * mt + j) * F::TROW + 32 * half + 16 * s2);
                FFNO_UNROLL
                for (int ch = 0; ch < CPW; ++ch) S::mma(a, hb[ch][s2], acc2[ch][mt], tc[ch]);
            }
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) S::fold(acc2[ch][mt], tc[ch]);
        }
        // dh^T[px][hid] = (db W2) * [h > 0]
        {
            f32x16 dc[CPW];
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) d[ch] = zero16(), dc[ch] = zero16();
            FFNO_UNROLL
            for (int st = 0; st < KS; ++st) {
                const Frag a = lds_frag_s<S>(L + OFF_DP, F::PPLANE, j * F::PROW + 32 * st + 16 * half);
                FFNO_UNROLL
                for (int ch = 0; ch < CPW; ++ch) S::mma(a, W2f[ch][st], d[ch], dc[ch]);
            }
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) S::fold(d[ch], dc[ch]);
        }
        FFNO_UNROLL
        for (int ch = 0; ch < CPW; ++ch) {
            FFNO_UNROLL
            for (int r = 0; r < 16; ++r) {
                d[ch][r] = ((bits >> (16 * ch + r)) & 1u) ? d[ch][r] : 0.f;
                bs1[ch] += d[ch][r];
            }
            hb[ch][0] = S::split8(d[ch][0], d[ch][1], d[ch][2], d[ch][3], d[ch][4], d[ch][5], d[ch][6], d[ch][7]);
            hb[ch][1] = S::split8(d[ch][8], d[ch][9], d[ch][10], d[ch][11], d[ch][12], d[ch][13], d[ch][14], d[ch][15]);
        }
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            f32x16 tc[CPW];
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) tc[ch] = zero16();
            FFNO_UNROLL
            for (int s2 = 0; s2 < 2; ++s2) {
                const Frag a = lds_frag_s<S>(L + OFF_ST, F::TPLANE, (32 * mt + j) * F::TROW + 32 * half + 16 * s2);
                FFNO_UNROLL
                for (int ch = 0; ch < CPW; ++ch) S::mma(a, hb[ch][s2], acc1[ch][mt], tc[ch]);
            }
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) S::fold(acc1[ch][mt], tc[ch]);
        }
        if (!early && nt < ntiles) stage(buf ^ 1);
        __syncthreads();
    }

    const float rg = S::SCALED ? 1.f / gscale : 1.f;     // (powers of two: exact; applied one after the other so that their
    const float rf = S::SCALED ? 1.f / fscale : 1.f;     //  product never leaves the float range)
    float* part = partial + (long)blockIdx.x * F::PART;
    float* pW1t = part;              // [c][hid]
    float* pW2 = part + H * C;       // [c][hid]
    float* pb1 = part + 2 * H * C;
    float* pb2 = pb1 + H;
    FFNO_UNROLL
    for (int ch = 0; ch < CPW; ++ch) {
        const int hid = 32 * (wave * CPW + ch) + j;
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            FFNO_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * mt + drow(r, half);
                pW1t[c * H + hid] = acc1[ch][mt][r] * rf * rg;
                pW2[c * H + hid] = acc2[ch][mt][r] * rf * rg;
            }
        }
        const float v1 = bs1[ch] + __shfl_xor(bs1[ch], 32);
        if (half == 0) pb1[hid] = v1 * rg;
    }
    red[tid] = bs2;
    __syncthreads();
    if (tid < C) {
        float v = 0.f;
        for (int k = tid; k < F::NT; k += C) v += red[k];
        pb2[tid] = v * rg;
    }
}

// ---- weight gradients, split-fp16 with SINGLE-accumulator products ("m") ------------------------------------------------------
// Same operator and slice layout as ffx_wgrad_kernel<SplitHf2>.  Measured on that kernel (profiles/r02_*): per tile and SIMD
// 3072 cycles of MFMA, ~2000 of LDS operand reads and ~2000 of vector work that ADD UP (7100 cycles); 256 registers per wave,
// 16 of them spilled, every correction tile bounced through v_accvgpr moves and folded after every product chain (64 vector
// operations per tile and accumulator).  What changes here:
//   * NO correction accumulators for the products without a decision.  A staged operand x is range-scaled (|x| <= 2^4, see
//     kFfRangeTarget), so 2^11 hi_x <= 2^15 is a valid half, and
//         acc += (2^11 hi_x) hi_y + hi_x (2^11 lo_y) + (2^11 lo_x) hi_y  =  2^11 (hi_x hi_y + hi_x lo_y + lo_x hi_y)
//     lands in ONE fp32 accumulator: the same three exact products as mfma_h2, the factor 2^-11 applied once (when dh is masked,
//     when the slice is written).  The 2^11 hi plane is made from the hi plane after the LDS read (one v_pk_mul_f16 per word).
//     Accumulator tiles per hidden chunk: 4 + 2 instead of 4 + 4 + 2 + 2, no folds, no spills.
//   * the h GEMM keeps the main + correction form of the forward kernel, product for product: its ReLU decisions must be the
//     forward's (a unit whose pre-activation rounds to the other side of zero would move a whole row of dW1);
//   * NWV = 8 waves (one hidden chunk each, two waves per SIMD): what ships.  The source also instantiates with 4 waves (two
//     chunks each, one wave per SIMD with 512 registers: every LDS fragment feeds both chunks' MFMAs and the two chunks are
//     independent instruction streams) -- measured 64 us against 56: a lone wave per SIMD exposes every LDS / HBM round trip
//     that its partner hides in the eight-wave form (round 3, DESIGN.md "Negative results"), so it is not compiled in;
//   * branch-free tile loop (one basic block: the scheduler interleaves staging / epilogue vector work with the MFMAs).
//   * the body takes its slice index and slice count as arguments (bid of nb): the per-layer launch passes its block index and
//     grid size, the all-layers launch (ffh_wgrad_m_multi_kernel below) the position inside its layer's slices.
//   * TWO = 1: s is the SUM of two tensors (the two branch outputs of the paired spectral launch), added -- and rounded to the
//     storage format, as the chain kernel does -- while the rows are staged: the forward chain kernel then need not write the sum
//     back (one image write less per launch: 46.1 -> 38.2 us with the wave-tile kernels, +2.5 % on the step, MI355X round 5);
//     s is such a sum (db is single: a second gradient addend, round 4's TWO = 2, was measured slower and removed).
// Round 5: the CHANNEL-MAJOR operands come from the pixel-major tile through the LDS transpose read.  Until round 4 every tile was
// staged twice -- pixel-major (float4 rows) and channel-major (four scalar loads per thread and tensor, a second split, a second
// set of LDS planes).  gfx950's ds_read_b64_tr_b16 hands a 16-lane
// group the TRANSPOSE of a [4 rows][16 halves] block: lane i of the group receives element i of each of the four rows, and the
// rows are wherever lanes 4 r .. 4 r + 3 of the group point (8 bytes each).  Four pixel rows x 16 channels of a pixel-major fp16
// plane are therefore one read away from the A operand of the pixel-contraction GEMMs (rows = channels, k = pixels), in exactly
// the k order of the D-fragment operand on the other side (slot e <-> pixel 16 s2 + (e & 3) + 8 (e >> 2) + 4 half).  Same fp16
// planes, same products in the same order: the W1 / W2 / b1 slices are bit-identical to the twice-staged kernel (b2 is summed in
// another order).  1387 -> 1186 us per all-layers launch under rocprofv3; the launch is power-limited.
//   LDS image of one plane: row R (pixel) at  R * PROW + 16 ((R >> 2) & 3) + WRAP (R >> 4),  PROW = 192 (C = 64) / 64 (C = 32):
//   * a transpose read touches rows 4 a .. 4 a + 3 over 64 bytes each: R * PROW mod 256 are four different multiples of 64;
//   * a pixel-major ds_read_b128 touches 16 rows with different R mod 16 at one column: their 16-byte slots mod 256 are
//     4 ((-R) & 3) + ((R >> 2) & 3), all different  -- both conflict-free (tools/lds_bank_check.py enumerates the lane groups).
template <int C>
struct WgT {
    static constexpr int PROW = C == 64 ? 192 : 64;
    static constexpr int WRAP = C == 64 ? 0 : 64;
    static constexpr int PLANE = 32 * PROW + 48 + WRAP + 16;      // (+16: keeps the next plane 64-byte aligned)
    static_assert(PLANE % 64 == 0, "plane alignment");
    static __device__ __forceinline__ int row(int R) { return R * PROW + 16 * ((R >> 2) & 3) + WRAP * (R >> 4); }
};

template <int C, int H, int NWV, class ST = StF32, int TWO = 0>
__device__ __forceinline__ void ffh_wgrad_m_body(const typename ST::T* __restrict__ s, const typename ST::T* __restrict__ db,
                                                 const u32x4* __restrict__ pk1, const float* __restrict__ bias1,
                                                 const u32x4* __restrict__ pk2t, float* __restrict__ partial, int P,
                                                 const unsigned* s_amax, const unsigned* db_amax, const int bid, const int nb,
                                                 const typename ST::T* __restrict__ s2 = nullptr) {
    constexpr int CPW = H / (32 * NWV);            // hidden chunks per wave
    using F = FxCfg<C, H, CPW>;
    using T = WgT<C>;
    constexpr int KS = F::KS, CTO = F::CTO, NV = F::NV;
    constexpr int GS = KS;                         // operand fragments per staged tile and orientation (C / 16)
    static_assert(F::NW == NWV && F::NT == NWV * 64 && CPW >= 1 && F::NT % C == 0, "wave / chunk map");
    static_assert(TWO == 0 || TWO == 1, "TWO = 1: s is the sum of two tensors");
    const float fscale = range_scale(*s_amax, 1, kFfRangeTarget);       // (the host side only takes this kernel WITH range words:
    const float gscale = range_scale(*db_amax, 1, kFfRangeTarget);      //  the 2^11 hi plane needs the 2^4 bound)
    // [s hi][s lo][db hi][db lo], pixel-major.  (The third plane of the single-accumulator product, hs = 2^11 hi, is made from hi
    // after the LDS read: staged as a plane of its own it saves 44 of 280 vector instructions per wave and tile and costs 20 more
    // LDS reads -- measured equal or slower, MI355X round 5: the launch is power-limited, see DESIGN.md.)
    constexpr int BUF = 4 * T::PLANE;
    constexpr int OFF_SP = 0, OFF_DP = 2 * T::PLANE;
    __shared__ __attribute__((aligned(64))) char lds[2][BUF];
    __shared__ float4 red[F::NT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ntiles = (P + 31) >> 5;

    Hf2 W1f[CPW][KS], W2f[CPW][KS];
    float b1v[CPW];
    FFNO_UNROLL
    for (int ch = 0; ch < CPW; ++ch) {
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {
            W1f[ch][st] = load_frag_s<SplitHf2>(pk1, (wave * CPW + ch) * KS + st, lane);
            W2f[ch][st] = load_frag_s<SplitHf2>(pk2t, (wave * CPW + ch) * KS + st, lane);
        }
        b1v[ch] = bias1[32 * (wave * CPW + ch) + j] * fscale;
    }
    // one product block of a staged (bounded) operand a with b on ONE accumulator: 2^11 x the fp32-grade product
    auto mma3 = [&](const Hf3& a, const Hf2& b, f32x16& acc) { acc = mfma_h2s(a, b, acc); };

    // Staging: thread f owns 4 consecutive channels of pixel row f / (C / 4).  Branch-free: tiles past the end re-read the last
    // tile (their staged copy is never used); rows past the end of a ragged last tile re-read its last valid row and are zeroed
    // -- by stage(), an iteration later: a select placed next to the load would make the wave wait for every load where it is issued.
    // (raw words of the next tile: widened in stage(), an iteration after the request)
    typename ST::Raw4 nSP[NV], nDP[NV], mSP[NV];      // (m*: the second addend of s, TWO only)
    float4 bs2 = make_float4(0.f, 0.f, 0.f, 0.f);      // this thread's part of db2: its 4 channels, its pixel row of every tile
    auto gload = [&](int tile_) {
        const int tile = min(tile_, ntiles - 1);
        const typename ST::T* st = s + (long)tile * (32 * C);
        const typename ST::T* dt = db + (long)tile * (32 * C);
        const typename ST::T* st2 = TWO ? s2 + (long)tile * (32 * C) : nullptr;
        const int rows = min(P - tile * 32, 32);
        FFNO_UNROLL
        for (int v = 0; v < NV; ++v) {
            const int f = tid + v * F::NT;
            const unsigned offp = (unsigned)(min(f / (C / 4), rows - 1) * C + 4 * (f % (C / 4)));
            nSP[v] = ST::ldr4(st + offp);
            nDP[v] = ST::ldr4(dt + offp);
            if constexpr (TWO >= 1) mSP[v] = ST::ldr4(st2 + offp);
        }
    };
    // tile_ = the tile whose rows are in the staging registers (may lie past the end: its copy is staged but never used and
    // does not count in the bias sums: zero rows)
    auto stage = [&](int buf, int tile_) {
        const int rows = tile_ < ntiles ? min(P - tile_ * 32, 32) : 0;
        FFNO_UNROLL
        for (int v = 0; v < NV; ++v) {
            const int f = tid + v * F::NT;
            const int R = f / (C / 4);
            const float mp = R < rows ? 1.f : 0.f;
            const float fs = fscale * mp, gs = gscale * mp;
            float4 sp = ST::w4(nSP[v]), dp = ST::w4(nDP[v]);
            if constexpr (TWO >= 1) {
                const float4 s2v = ST::w4(mSP[v]);
                sp.x += s2v.x, sp.y += s2v.y, sp.z += s2v.z, sp.w += s2v.w;
                sp = st_rnd4<ST>(sp);
            }
            sp.x *= fs, sp.y *= fs, sp.z *= fs, sp.w *= fs;
            dp.x *= gs, dp.y *= gs, dp.z *= gs, dp.w *= gs;
            const int offp = T::row(R) + (f % (C / 4)) * 8;
            uint2 ph, pl;
            plat::split2_pair_mix(sp.x, sp.y, ph.x, pl.x);
            plat::split2_pair_mix(sp.z, sp.w, ph.y, pl.y);
            *reinterpret_cast<uint2*>(lds[buf] + OFF_SP + offp) = ph;
            *reinterpret_cast<uint2*>(lds[buf] + OFF_SP + T::PLANE + offp) = pl;
            plat::split2_pair_mix(dp.x, dp.y, ph.x, pl.x);
            plat::split2_pair_mix(dp.z, dp.w, ph.y, pl.y);
            *reinterpret_cast<uint2*>(lds[buf] + OFF_DP + offp) = ph;
            *reinterpret_cast<uint2*>(lds[buf] + OFF_DP + T::PLANE + offp) = pl;
            bs2.x += dp.x, bs2.y += dp.y, bs2.z += dp.z, bs2.w += dp.w;
        }
    };

    f32x16 acc1[CPW][CTO], acc2[CPW][CTO];          // 2^11 x the two weight-gradient slices
    float bs1[CPW];
    FFNO_UNROLL
    for (int ch = 0; ch < CPW; ++ch) {
        bs1[ch] = 0.f;
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) acc1[ch][mt] = zero16(), acc2[ch][mt] = zero16();
    }
    // per-lane parts of the two LDS address patterns (the rest are compile-time offsets)
    const int prow = T::row(j) + 16 * half;                                             // pixel-major: row j, +32 q
    const int trow = (4 * half + ((lane & 15) >> 2)) * T::PROW + 16 * half              // transposed: row 4 half + jj of its group,
                     + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);                         //  channels 16 g + 4 q'

    gload(bid);
    stage(0, bid);
    gload(bid + nb);
    __syncthreads();
    int buf = 0;
    for (int tile = bid; tile < ntiles; tile += nb, buf ^= 1) {
        const int nt = tile + nb;
        const char* L = lds[buf];
        // channel-major fragment (mt, s2) of a tensor: rows = channels 32 mt + (lane & 31), k slots = pixels of k-step s2 in the
        // D-fragment order; two transpose reads per plane (k slots 0..3 / 4..7 = pixel rows +0 / +8)
        auto tfrag = [&](int off_tensor, int mt, int s2_) {
            Hf3 f;
            FFNO_UNROLL
            for (int p = 0; p < 2; ++p) {
                const char* base = L + off_tensor + p * T::PLANE + trow + s2_ * (16 * T::PROW + T::WRAP) + 64 * mt;
                const uint2 lo4 = plat::lds_read_tr16_b64(base);
                const uint2 hi4 = plat::lds_read_tr16_b64(base + 8 * T::PROW + 32);
                u32x4 w;
                w[0] = lo4.x, w[1] = lo4.y, w[2] = hi4.x, w[3] = hi4.y;
                if (p == 0) f.hi = w;
                if (p == 1) f.lo = w;
            }
            FFNO_UNROLL
            for (int q = 0; q < 4; ++q) f.hs[q] = plat::pk_mul_f16(f.hi[q], kHf2Scale);
            return f;
        };
        // pixel-major fragment q of a tensor (rows = the tile's pixels, k = channels 16 q ..): hi / lo for the h GEMM, + hs for dh
        auto pfrag = [&](int off_tensor, int q, bool with_hs) {
            Hf3 f;
            const char* base = L + off_tensor + prow + 32 * q;
            f.hi = *reinterpret_cast<const u32x4*>(base);
            f.lo = *reinterpret_cast<const u32x4*>(base + T::PLANE);
            if (with_hs) {
                FFNO_UNROLL
                for (int q2 = 0; q2 < 4; ++q2) f.hs[q2] = plat::pk_mul_f16(f.hi[q2], kHf2Scale);
            } else f.hs = f.hi;      // (never used: mfma_h2 takes hi / lo)
            return f;
        };
        // The 16 operand fragments of a tile, in the order the products consume them (s, db^T, db, s^T), come through a ring of
        // two: fragment i + 2 is requested when fragment i has been handed to its MFMAs, and the request is pinned there (a
        // scheduling barrier that only LDS reads may not cross) -- left alone the scheduler puts every read right in front of
        // its use and the wave waits out each LDS round trip.
        // Order of a tile: both recompute GEMMs first (h from s, dh from db: pixel-major operands), then BOTH epilogues, then both
        // pixel-contraction GEMMs (channel-major operands through the transpose read).  The MFMAs of the dh product have no
        // dependence on the h epilogue, those of dW2 none on the dh epilogue, those of dW1 none on the next tile's staging: the
        // scheduler has independent matrix work to put between the vector instructions of every phase.
        auto frag = [&](int i) {
            const int g = i / GS, q = i % GS;
            if (g == 0) return pfrag(OFF_SP, q, false);
            if (g == 1) return pfrag(OFF_DP, q, true);
            if (g == 2) return tfrag(OFF_DP, q >> 1, q & 1);
            return tfrag(OFF_SP, q >> 1, q & 1);
        };
        Hf3 ring[2];
        ring[0] = frag(0), ring[1] = frag(1);
        FFNO_SCHED_PIN_DSREAD();
        stage(buf ^ 1, nt);
        if constexpr (NWV == 8) {
            gload(nt + nb);
            FFNO_SCHED_PIN_VMEM();
        }
        f32x16 d[CPW], dc[CPW], e[CPW];
        FFNO_UNROLL
        for (int ch = 0; ch < CPW; ++ch) d[ch] = zero16(), dc[ch] = zero16(), e[ch] = zero16();
        static_assert(2 * CTO == KS, "fragment ring: C / 16 fragments per operand, pixel-major and channel-major alike");
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {      // h^T[px][hid] = s W1^T: main + correction tile, as in the forward kernel
            const Hf2 a = {ring[st & 1].hi, ring[st & 1].lo};
            ring[st & 1] = frag(0 * GS + st + 2);
            FFNO_SCHED_PIN_DSREAD();
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) mfma_h2(a, W1f[ch][st], d[ch], dc[ch]);
        }
        if constexpr (NWV != 8) {
            gload(nt + nb);
            FFNO_SCHED_FENCE();
        }
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {      // 2^11 db W2 (unmasked dh)
            const Hf3 a = ring[st & 1];
            ring[st & 1] = frag(1 * GS + st + 2);
            FFNO_SCHED_PIN_DSREAD();
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) mma3(a, W2f[ch][st], e[ch]);
        }
        Hf2 hb[CPW][2], gb[CPW][2];
        FFNO_UNROLL
        for (int ch = 0; ch < CPW; ++ch) {
            SplitHf2::fold(d[ch], dc[ch]);
            FFNO_UNROLL
            for (int r = 0; r < 16; ++r) {
                const float v = d[ch][r] + b1v[ch];
                const bool pos = v > 0.f;
                d[ch][r] = pos ? v : 0.f;
                e[ch][r] = pos ? e[ch][r] * kHf2Unscale : 0.f;
                bs1[ch] += e[ch][r];
            }
            FFNO_UNROLL
            for (int k2 = 0; k2 < 2; ++k2) {
                FFNO_UNROLL
                for (int w2 = 0; w2 < 4; ++w2) {
                    unsigned hh, ll;
                    plat::split2_pair_mix(d[ch][8 * k2 + 2 * w2], d[ch][8 * k2 + 2 * w2 + 1], hh, ll);
                    hb[ch][k2].hi[w2] = hh, hb[ch][k2].lo[w2] = ll;
                    plat::split2_pair_mix(e[ch][8 * k2 + 2 * w2], e[ch][8 * k2 + 2 * w2 + 1], hh, ll);
                    gb[ch][k2].hi[w2] = hh, gb[ch][k2].lo[w2] = ll;
                }
            }
        }
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {     // dW2[c][hid] += db^T h
            FFNO_UNROLL
            for (int s2_ = 0; s2_ < 2; ++s2_) {
                const int q = 2 * mt + s2_;
                const Hf3 a = ring[q & 1];
                ring[q & 1] = frag(2 * GS + q + 2);
                FFNO_SCHED_PIN_DSREAD();
                FFNO_UNROLL
                for (int ch = 0; ch < CPW; ++ch) mma3(a, hb[ch][s2_], acc2[ch][mt]);
            }
        }
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {     // dW1^T[c][hid] += s^T dh
            FFNO_UNROLL
            for (int s2_ = 0; s2_ < 2; ++s2_) {
                const int q = 2 * mt + s2_;
                const Hf3 a = ring[q & 1];
                if (q + 2 < GS) {
                    ring[q & 1] = frag(3 * GS + q + 2);
                    FFNO_SCHED_PIN_DSREAD();
                }
                FFNO_UNROLL
                for (int ch = 0; ch < CPW; ++ch) mma3(a, gb[ch][s2_], acc1[ch][mt]);
            }
        }
        __syncthreads();
    }

    // (powers of two: exact; applied one after the other so that their product never leaves the float range)
    const float rg = 1.f / gscale, rf = 1.f / fscale;
    float* part = partial + (long)bid * F::PART;
    float* pW1t = part;              // [c][hid]
    float* pW2 = part + H * C;       // [c][hid]
    float* pb1 = part + 2 * H * C;
    float* pb2 = pb1 + H;
    FFNO_UNROLL
    for (int ch = 0; ch < CPW; ++ch) {
        const int hid = 32 * (wave * CPW + ch) + j;
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            FFNO_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * mt + drow(r, half);
                pW1t[c * H + hid] = acc1[ch][mt][r] * kHf2Unscale * rf * rg;
                pW2[c * H + hid] = acc2[ch][mt][r] * kHf2Unscale * rf * rg;
            }
        }
        const float v1 = bs1[ch] + __shfl_xor(bs1[ch], 32);
        if (half == 0) pb1[hid] = v1 * rg;
    }
    // db2[c] = sum over the slice's pixels: thread f holds channels 4 (f % (C/4)) .. +3 of pixel row f / (C/4); rows in order
    red[tid] = bs2;
    __syncthreads();
    if (tid < C) {
        float v = 0.f;
        for (int r = 0; r < F::NT / (C / 4); ++r) {
            const float4 t = red[r * (C / 4) + (tid >> 2)];
            v += (tid & 3) == 0 ? t.x : (tid & 3) == 1 ? t.y : (tid & 3) == 2 ? t.z : t.w;
        }
        pb2[tid] = v * rg;
    }
}

template <int C, int H, int NWV, class ST = StF32>
__global__ __launch_bounds__(NWV * 64) void ffh_wgrad_m_kernel(const typename ST::T* __restrict__ s,
                                                               const typename ST::T* __restrict__ db,
                                                               const u32x4* __restrict__ pk1, const float* __restrict__ bias1,
                                                               const u32x4* __restrict__ pk2t, float* __restrict__ partial,
                                                               int P, const unsigned* s_amax, const unsigned* db_amax) {
    ffh_wgrad_m_body<C, H, NWV, ST>(s, db, pk1, bias1, pk2t, partial, P, s_amax, db_amax, (int)blockIdx.x, (int)gridDim.x);
}

// The weight-gradient launches of ALL layers of a backward pass as one (ffno_ffh_bwd_weights_partial_multi).  Nothing downstream
// of a layer's weight gradient is on the backward's critical path, so the engine keeps every layer's (s, summed gradient) and
// runs this once at the end: `nsplit` workgroups per layer, each walking ntiles / nsplit tiles of its layer.  Measured on the
// per-layer launch (MI355X, round 4, batch 8..128): 7.3 us + 1.5 us per image -- the constant is the fragment loads at the
// start and the burst of 256 x 132 KB slices at the end, plus 7 us per layer in the reduce kernel that reads them back.  With
// 3 x CUs / L slices per layer (768 workgroups for 24 layers, three full rounds) a workgroup walks 8 x as many tiles per slice.
struct FfWgDesc {
    const void* s;
    const void* g;
    const u32x4* pk1;
    const float* b1;
    const u32x4* pk2t;
    float* partial;
    const unsigned* s_amax;
    const unsigned* g_amax;
    const void* s2;      // second addend of s (TWO launches)
    const void* g2;      // (reserved: round 4's second gradient addend, measured slower and removed)
};

// The table travels BY VALUE in the kernel-argument segment (up to FFWG_MAX_BLOCKS blocks per launch): pointers inside a by-value
// argument are known to be global, pointers loaded from a device-resident table are generic to the compiler -- every access
// through them becomes a flat_load / flat_store, which also counts on the LDS counter, so each wait for an LDS read waited for the
// wave's outstanding global prefetch as well (the first version of this launch did that: 115-145 flat accesses per kernel).
constexpr int FFWG_MAX_BLOCKS = 32;
struct FfWgTable {
    FfWgDesc d[FFWG_MAX_BLOCKS];
};

template <int C, int H, int NWV, class ST = StF32, int TWO = 0>
__global__ __launch_bounds__(NWV * 64) void ffh_wgrad_m_multi_kernel(const FfWgTable tab, int P, int nsplit) {
    const int layer = (int)blockIdx.x / nsplit;
    const FfWgDesc& d = tab.d[layer];
    typedef const typename ST::T* cp;
    ffh_wgrad_m_body<C, H, NWV, ST, TWO>((cp)d.s, (cp)d.g, d.pk1, d.b1, d.pk2t, d.partial, P, d.s_amax, d.g_amax,
                                         (int)blockIdx.x - layer * nsplit, nsplit, (cp)d.s2);
}

// partial slices -> gradients; the dW1 block of a slice is [c][hid] (transposed)
__global__ __launch_bounds__(256) void ffx_wgrad_reduce_kernel(const float* __restrict__ partial, float* dW1, float* dW2,
                                                               float* db1, float* db2, int C, int H, int nsplit,
                                                               int accumulate) {
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
        sum = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        float* dst;
        if (e < H * C)
            dst = dW1 + (e % H) * C + e / H;
        else if (e < 2 * H * C)
            dst = dW2 + (e - H * C);
        else if (e < 2 * H * C + H)
            dst = db1 + (e - 2 * H * C);
        else
            dst = db2 + (e - 2 * H * C - H);
        *dst = accumulate ? (*dst + sum) : sum;
    }
}

// the same reduction for a table of feed-forward blocks in ONE launch (blockIdx.y = block): every layer of a backward pass
// writes its slices to its own partial buffer and the 24 small reductions become one kernel boundary
struct FxRedDesc {
    const float* partial;
    float *dW1, *dW2, *db1, *db2;
};

// (round 6: a 16-byte-load form of this kernel -- a lane summing four consecutive elements of every slice, a quarter of the
//  workgroups -- was measured and not kept: 20.3 us per launch against 17.3 for this one at the headline shape; 101 MB in 17 us is
//  5.8 TB/s already)
__global__ __launch_bounds__(256) void ffx_wgrad_reduce_batched_kernel(const FxRedDesc* __restrict__ descs, int C, int H,
                                                                       int nsplit) {
    __shared__ float red[4][64];
    const FxRedDesc d = descs[blockIdx.y];
    const int part = 2 * H * C + H + C;
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), sg = threadIdx.x >> 6;
    float sum = 0.f;
    if (e < part) {
        float t[8];
        int sp = sg;
        for (; sp + 28 < nsplit; sp += 32) {
            FFNO_UNROLL
            for (int u = 0; u < 8; ++u) t[u] = d.partial[(long)(sp + 4 * u) * part + e];
            FFNO_UNROLL
            for (int u = 0; u < 8; ++u) sum += t[u];
        }
        for (; sp < nsplit; sp += 4) sum += d.partial[(long)sp * part + e];
    }
    red[sg][threadIdx.x & 63] = sum;
    __syncthreads();
    if (sg == 0 && e < part) {
        sum = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (e < H * C)
            d.dW1[(e % H) * C + e / H] = sum;
        else if (e < 2 * H * C)
            d.dW2[e - H * C] = sum;
        else if (e < 2 * H * C + H)
            d.db1[e - 2 * H * C] = sum;
        else
            d.db2[e - 2 * H * C - H] = sum;
    }
}

// ReLU sign bits of the chain kernel -> one byte per (pixel, hidden unit): word (tile, wave, lane = (j, half)) holds the 16
// accumulator rows of pixel 32 tile + j, hidden units 32 wave + drow(r, half) at bit 15 - r.
__global__ __launch_bounds__(256) void ffx_mask_unpack_kernel(const uint16_t* __restrict__ mask, uint8_t* __restrict__ out, int P,
                                                              int H) {
    const int NW = H / 32;
    const long nwords = (long)((P + 31) >> 5) * NW * 64;
    for (long w = (long)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(w & 63), j = lane & 31, half = lane >> 5;
        const int wave = (int)((w >> 6) % NW);
        const long px = (w >> 6) / NW * 32 + j;
        if (px >= P) continue;
        const uint32_t bits = mask[w];
        for (int r = 0; r < 16; ++r) out[px * H + 32 * wave + drow(r, half)] = (uint8_t)((bits >> (15 - r)) & 1u);
    }
}

static inline int ffx_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFNO_OK : (int)e;
}

// persistent workgroups of the chain kernels: one per compute unit of the current device unless the caller says otherwise
// (max_workgroups > 0).  The CU count is an immutable per-device fact, cached after the first query.
// persistent workgroups of a chain launch: one per CU at H = 256 (512 threads, 85-150 KiB of LDS: one fits); THREE per CU for the
// 256-thread instances (H = 128: width 32 -- four waves and 27-43 KiB each, so a lone workgroup leaves three quarters of the
// CU's wave slots empty; measured at 72^3 x 32, round 4: 157 -> 165 steps/s with 768 workgroups and 512 weight-gradient slices)
static inline int fx_workgroups(int max_workgroups, int ntiles, int H = 256) {
    const int n = max_workgroups > 0 ? max_workgroups : device_cu_count() * (H <= 128 ? 3 : 1);
    return n < ntiles ? n : ntiles;
}

}  // namespace ffno

using namespace ffno;

#define FFNO_FX_DISPATCH(MACRO) \
    MACRO(64, 256)              \
    MACRO(64, 128)              \
    MACRO(32, 128)              \
    MACRO(32, 64)

extern "C" int ffno_ffx_supported(int C, int H) {
#define CASE(CC, HH) \
    if (C == CC && H == HH) return 1;
    FFNO_FX_DISPATCH(CASE)
#undef CASE
    return 0;
}

extern "C" size_t ffno_ffx_pack_bytes(int C, int H) { return (size_t)C * H * 6; }

extern "C" int ffno_ffx_mask_unpack(const void* mask, uint8_t* active, int P, int C, int H, void* stream) {
    if (!mask || !active || P <= 0) return FFNO_EINVAL;
    if (!ffno_ffx_supported(C, H)) return FFNO_EUNSUPPORTED;
    FFNO_LAUNCH(ffx_mask_unpack_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)mask, active, P, H);
    return ffx_launch_status();
}

// ---- host side, written once over the split policy --------------------------------------------------------------------------
// Schedules (measured on MI355X at markov/24, batch 32, profiles/r02_ffx_schedules.md): the forward runs the role-split kernel
// (53 vs 57 us), backward-data and the weight gradients the in-phase kernels (50 vs 56 us, 89 vs 108 us).  The schedules that
// lost are in tools/experiments/ffx_negative_results.hip, not in this library.
template <class S>
static int fx_pack(const ffno_fxpack_desc* descs_dev, int n, int C, int H, void* stream) {
    if (!descs_dev || n <= 0) return FFNO_EINVAL;
    if (!ffno_ffx_supported(C, H)) return FFNO_EUNSUPPORTED;
    static_assert(sizeof(ffno_fxpack_desc) == sizeof(FxPackDesc), "descriptor layout");
    const int threads = (C * H / 8);   // one thread per (fragment, lane): 8 weights each
    FFNO_LAUNCH((ffx_pack_kernel<S>), dim3((threads + 255) / 256, n), dim3(256), 0, (hipStream_t)stream,
                reinterpret_cast<const FxPackDesc*>(descs_dev), C, H);
    return ffx_launch_status();
}

// wave-tile chain kernels (ffw_chain_kernel): split-fp16 at 64 / 256; schedule 0 picks them, FFNO_FF_SCHED_WAVE_TILES demands them
static inline bool fw_wave_tiles(int np, int C, int H, int schedule, int ntiles) {
    if (np != 2 || !(C == 64 && H == 256)) return false;
    if (schedule == FFNO_FF_SCHED_WAVE_TILES) return true;
    // by default only where every CU gets whole workgroups of tiles (a workgroup copies 128 KiB of weights and walks eight tiles
    // at a time: a batch-1 rollout step of 128 tiles would run on 16 CUs)
    return schedule == 0 && ntiles >= 8 * device_cu_count();
}
template <int C, int H, bool BWD, class ST>
static int fw_launch(const void* in, const void* in2, void* sum_out, const void* resid, const void* pk1, const float* b1,
                     const void* pk2, const float* b2, void* out, void* mask, int P, const unsigned* ia, unsigned* oa,
                     int max_workgroups, hipStream_t st) {
    typedef typename ST::T T;
    constexpr int NCH = H / 32, KS = C / 16, CTO = C / 32;
    const size_t smem = (size_t)(NCH * KS + NCH * CTO * 2) * 2 * 64 * sizeof(u32x4) + (size_t)(H + C + 8 * 32 * 20) * sizeof(float);
    const int ntiles = (P + 31) / 32;
    const int want = (ntiles + 7) / 8;
    const int cap = max_workgroups > 0 ? max_workgroups : device_cu_count();
    const dim3 grid(want < cap ? want : cap);
    const int rc = allow_dynamic_lds(ffw_chain_kernel<C, H, BWD, ST>, smem);
    if (rc) return rc;
    FFNO_LAUNCH((ffw_chain_kernel<C, H, BWD, ST>), grid, dim3(512), smem, st, (const T*)in, (const T*)in2, (T*)sum_out,
                (const T*)resid, (const u32x4*)pk1, b1, (const u32x4*)pk2, b2, (T*)out, (uint32_t*)mask, P, ia, oa);
    return ffx_launch_status();
}

template <class S>
static int fx_fwd2(const float* s, const float* s2, float* s_sum, const float* resid, const void* pk1, const float* b1,
                   const void* pk2, const float* b2, float* out, void* mask, int P, int C, int H, const ffno_ff_opts* o,
                   void* stream) {
    if (!s || !pk1 || !b1 || !pk2 || !b2 || !out || P <= 0 || (s_sum && !s2)) return FFNO_EINVAL;
    const int ntiles = (P + 31) / 32;
    const dim3 grid(fx_workgroups(o ? o->max_workgroups : 0, ntiles, H));
    const unsigned* ia = o ? o->in_amax : nullptr;
    unsigned* oa = o ? o->out_amax : nullptr;
    const int sched = o ? o->schedule : 0;
    if (sched < 0 || sched > FFNO_FF_SCHED_ROLE_SPLIT) return FFNO_EINVAL;
    const bool in_phase = sched == FFNO_FF_SCHED_IN_PHASE;
    hipStream_t st = (hipStream_t)stream;
    const bool b16 = o && o->storage == FFNO_STORE_BF16;
    if (o && o->storage != FFNO_STORE_F32 && !b16) return FFNO_EINVAL;
    if (fw_wave_tiles(S::NP, C, H, sched, ntiles)) {
        const int mw = o ? o->max_workgroups : 0;
        return b16 ? fw_launch<64, 256, false, StBf16>(s, s2, s_sum, resid, pk1, b1, pk2, b2, out, mask, P, ia, oa, mw, st)
                   : fw_launch<64, 256, false, StF32>(s, s2, s_sum, resid, pk1, b1, pk2, b2, out, mask, P, ia, oa, mw, st);
    }
    if (sched == FFNO_FF_SCHED_WAVE_TILES) return FFNO_EUNSUPPORTED;
    if (o && o->storage == FFNO_STORE_BF16) {      // bf16 storage twins: the split-fp16 kernels of widths 64 and 32 (factor 4)
        typedef const uint16_t* cp;
        if (S::NP != 2 || !((C == 64 && H == 256) || (C == 32 && H == 128))) return FFNO_EUNSUPPORTED;
        if (C == 64)
            FFNO_LAUNCH((ffx_chain_rs_kernel<64, 256, false, SplitHf2, StBf16>), grid, dim3(FxCfg<64, 256>::NT), 0, st, (cp)s,
                        (cp)s2, (uint16_t*)s_sum, (cp)resid, (const u32x4*)pk1, b1, (const u32x4*)pk2, b2, (uint16_t*)out,
                        (uint32_t*)mask, P, ia, oa);
        else
            FFNO_LAUNCH((ffx_chain_rs_kernel<32, 128, false, SplitHf2, StBf16>), grid, dim3(FxCfg<32, 128>::NT), 0, st, (cp)s,
                        (cp)s2, (uint16_t*)s_sum, (cp)resid, (const u32x4*)pk1, b1, (const u32x4*)pk2, b2, (uint16_t*)out,
                        (uint32_t*)mask, P, ia, oa);
        return ffx_launch_status();
    }
    if (o && o->storage != FFNO_STORE_F32) return FFNO_EINVAL;
#define CASE(CC, HH)                                                                                                  \
    if (C == CC && H == HH) {                                                                                         \
        if (!in_phase && HH >= 64)                                                                                    \
            FFNO_LAUNCH((ffx_chain_rs_kernel<CC, HH, false, S>), grid, dim3(FxCfg<CC, HH>::NT), 0, st, s, s2, s_sum,  \
                        resid, (const u32x4*)pk1, b1, (const u32x4*)pk2, b2, out, (uint32_t*)mask, P, ia, oa);        \
        else                                                                                                          \
            FFNO_LAUNCH((ffx_chain_kernel<CC, HH, false, S>), grid, dim3(FxCfg<CC, HH>::NT), 0, st, s, s2, s_sum,     \
                        resid, (const u32x4*)pk1, b1, (const u32x4*)pk2, b2, out, (uint32_t*)mask, P, ia, oa);        \
        return ffx_launch_status();                                                                                   \
    }
    FFNO_FX_DISPATCH(CASE)
#undef CASE
    return FFNO_EUNSUPPORTED;
}

template <class S>
static int fx_bwd_data2(const float* db, const float* db2, float* db_sum, const void* mask, const void* pk1b,
                        const void* pk2b, float* ds, int P, int C, int H, const ffno_ff_opts* o, void* stream) {
    if (!db || !mask || !pk1b || !pk2b || !ds || P <= 0 || (db_sum && !db2)) return FFNO_EINVAL;
    const int ntiles = (P + 31) / 32;
    const dim3 grid(fx_workgroups(o ? o->max_workgroups : 0, ntiles, H));
    const unsigned* ia = o ? o->in_amax : nullptr;
    unsigned* oa = o ? o->out_amax : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const int sched = o ? o->schedule : 0;
    if (sched < 0 || sched > FFNO_FF_SCHED_ROLE_SPLIT) return FFNO_EINVAL;
    if (o && o->storage != FFNO_STORE_F32 && o->storage != FFNO_STORE_BF16) return FFNO_EINVAL;
    if (fw_wave_tiles(S::NP, C, H, sched, ntiles)) {
        const int mw = o ? o->max_workgroups : 0;
        void* m = const_cast<void*>(mask);
        return (o && o->storage == FFNO_STORE_BF16)
                   ? fw_launch<64, 256, true, StBf16>(db, db2, db_sum, nullptr, pk1b, nullptr, pk2b, nullptr, ds, m, P, ia, oa, mw, st)
                   : fw_launch<64, 256, true, StF32>(db, db2, db_sum, nullptr, pk1b, nullptr, pk2b, nullptr, ds, m, P, ia, oa, mw, st);
    }
    if (sched == FFNO_FF_SCHED_WAVE_TILES) return FFNO_EUNSUPPORTED;
    if (o && o->storage == FFNO_STORE_BF16) {
        typedef const uint16_t* cp;
        if (S::NP != 2 || !((C == 64 && H == 256) || (C == 32 && H == 128))) return FFNO_EUNSUPPORTED;
        if (C == 64)
            FFNO_LAUNCH((ffx_chain_kernel<64, 256, true, SplitHf2, StBf16>), grid, dim3(FxCfg<64, 256>::NT), 0, st, (cp)db,
                        (cp)db2, (uint16_t*)db_sum, (cp) nullptr, (const u32x4*)pk1b, nullptr, (const u32x4*)pk2b, nullptr,
                        (uint16_t*)ds, (uint32_t*)const_cast<void*>(mask), P, ia, oa);
        else
            FFNO_LAUNCH((ffx_chain_kernel<32, 128, true, SplitHf2, StBf16>), grid, dim3(FxCfg<32, 128>::NT), 0, st, (cp)db,
                        (cp)db2, (uint16_t*)db_sum, (cp) nullptr, (const u32x4*)pk1b, nullptr, (const u32x4*)pk2b, nullptr,
                        (uint16_t*)ds, (uint32_t*)const_cast<void*>(mask), P, ia, oa);
        return ffx_launch_status();
    }
    if (o && o->storage != FFNO_STORE_F32) return FFNO_EINVAL;
#define CASE(CC, HH)                                                                                                  \
    if (C == CC && H == HH) {                                                                                         \
        FFNO_LAUNCH((ffx_chain_kernel<CC, HH, true, S>), grid, dim3(FxCfg<CC, HH>::NT), 0, st, db, db2, db_sum,       \
                    nullptr, (const u32x4*)pk1b, nullptr, (const u32x4*)pk2b, nullptr, ds,                            \
                    (uint32_t*)const_cast<void*>(mask), P, ia, oa);                                                   \
        return ffx_launch_status();                                                                                   \
    }
    FFNO_FX_DISPATCH(CASE)
#undef CASE
    return FFNO_EUNSUPPORTED;
}

template <class S>
static int fx_bwd_weights_partial(const float* s, const float* db, const void* pk1, const float* b1, const void* pk1b,
                                  float* partial, int P, int C, int H, int nsplit, const unsigned* s_amax,
                                  const unsigned* db_amax, int storage, void* stream) {
    if (!s || !db || !pk1 || !b1 || !pk1b || !partial || P <= 0 || nsplit <= 0) return FFNO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (storage == FFNO_STORE_BF16) {      // bf16 storage twin: the single-accumulator kernel (needs both range words)
        if (S::NP != 2 || !((C == 64 && H == 256) || (C == 32 && H == 128))) return FFNO_EUNSUPPORTED;
        if (!s_amax || !db_amax) return FFNO_EINVAL;
        if (C == 64)
            FFNO_LAUNCH((ffh_wgrad_m_kernel<64, 256, 8, StBf16>), dim3(nsplit), dim3(512), 0, st, (const uint16_t*)s,
                        (const uint16_t*)db, (const u32x4*)pk1, b1, (const u32x4*)pk1b, partial, P, s_amax, db_amax);
        else
            FFNO_LAUNCH((ffh_wgrad_m_kernel<32, 128, 4, StBf16>), dim3(nsplit), dim3(256), 0, st, (const uint16_t*)s,
                        (const uint16_t*)db, (const u32x4*)pk1, b1, (const u32x4*)pk1b, partial, P, s_amax, db_amax);
        return ffx_launch_status();
    }
    if (storage != FFNO_STORE_F32) return FFNO_EINVAL;
    // split-fp16 with range words at the headline shape: the single-accumulator kernel (measured 56 vs 66 us per layer)
    const bool merged = S::NP == 2 && s_amax && db_amax;
#define CASE(CC, HH)                                                                                               \
    if (C == CC && H == HH) {                                                                                      \
        if (merged && CC == 64 && HH == 256)                                                                       \
            FFNO_LAUNCH((ffh_wgrad_m_kernel<64, 256, 8>), dim3(nsplit), dim3(512), 0, st, s, db,                   \
                        (const u32x4*)pk1, b1, (const u32x4*)pk1b, partial, P, s_amax, db_amax);                   \
        else if (merged && CC == 32 && HH == 128)      /* width 32 (3-D mesh operators): four waves */              \
            FFNO_LAUNCH((ffh_wgrad_m_kernel<32, 128, 4>), dim3(nsplit), dim3(256), 0, st, s, db,                   \
                        (const u32x4*)pk1, b1, (const u32x4*)pk1b, partial, P, s_amax, db_amax);                   \
        else                                                                                                       \
            FFNO_LAUNCH((ffx_wgrad_kernel<CC, HH, S>), dim3(nsplit), dim3(FxCfg<CC, HH>::NT), 0, st, s, db,        \
                        (const u32x4*)pk1, b1, (const u32x4*)pk1b, partial, P, s_amax, db_amax);                   \
        return ffx_launch_status();                                                                                \
    }
    FFNO_FX_DISPATCH(CASE)
#undef CASE
    return FFNO_EUNSUPPORTED;
}

// ---- split-bf16 entry points (any fp32 range: the range words of `opts` are only PRODUCED, never needed) ----
extern "C" int ffno_ffx_pack(const ffno_fxpack_desc* descs_dev, int n, int C, int H, void* stream) {
    return fx_pack<SplitBf3>(descs_dev, n, C, H, stream);
}
extern "C" int ffno_ffx_fwd(const float* s, const float* resid, const void* pk1, const float* b1, const void* pk2,
                            const float* b2, float* out, void* mask, int P, int C, int H, void* stream) {
    return fx_fwd2<SplitBf3>(s, nullptr, nullptr, resid, pk1, b1, pk2, b2, out, mask, P, C, H, nullptr, stream);
}
extern "C" int ffno_ffx_fwd2(const float* s, const float* s2, float* s_sum, const float* resid, const void* pk1,
                             const float* b1, const void* pk2, const float* b2, float* out, void* mask, int P, int C,
                             int H, const ffno_ff_opts* opts, void* stream) {
    return fx_fwd2<SplitBf3>(s, s2, s_sum, resid, pk1, b1, pk2, b2, out, mask, P, C, H, opts, stream);
}
extern "C" int ffno_ffx_bwd_data(const float* db, const void* mask, const void* pk1b, const void* pk2b, float* ds, int P,
                                 int C, int H, void* stream) {
    return fx_bwd_data2<SplitBf3>(db, nullptr, nullptr, mask, pk1b, pk2b, ds, P, C, H, nullptr, stream);
}
extern "C" int ffno_ffx_bwd_data2(const float* db, const float* db2, float* db_sum, const void* mask, const void* pk1b,
                                  const void* pk2b, float* ds, int P, int C, int H, const ffno_ff_opts* opts, void* stream) {
    return fx_bwd_data2<SplitBf3>(db, db2, db_sum, mask, pk1b, pk2b, ds, P, C, H, opts, stream);
}
extern "C" int ffno_ffx_bwd_weights_partial(const float* s, const float* db, const void* pk1, const float* b1,
                                            const void* pk1b, float* partial, int P, int C, int H, int nsplit,
                                            void* stream) {
    return fx_bwd_weights_partial<SplitBf3>(s, db, pk1, b1, pk1b, partial, P, C, H, nsplit, nullptr, nullptr, FFNO_STORE_F32,
                                            stream);
}

// ---- split-fp16 entry points (same operators, masks, partial-slice layout and reduce kernels; own weight packs) ----
extern "C" size_t ffno_ffh_pack_bytes(int C, int H) { return (size_t)C * H * 4; }
extern "C" int ffno_ffh_pack(const ffno_fxpack_desc* descs_dev, int n, int C, int H, void* stream) {
    return fx_pack<SplitHf2>(descs_dev, n, C, H, stream);
}
extern "C" int ffno_ffh_fwd2(const float* s, const float* s2, float* s_sum, const float* resid, const void* pk1,
                             const float* b1, const void* pk2, const float* b2, float* out, void* mask, int P, int C,
                             int H, const ffno_ff_opts* opts, void* stream) {
    return fx_fwd2<SplitHf2>(s, s2, s_sum, resid, pk1, b1, pk2, b2, out, mask, P, C, H, opts, stream);
}
extern "C" int ffno_ffh_bwd_data2(const float* db, const float* db2, float* db_sum, const void* mask, const void* pk1b,
                                  const void* pk2b, float* ds, int P, int C, int H, const ffno_ff_opts* opts, void* stream) {
    return fx_bwd_data2<SplitHf2>(db, db2, db_sum, mask, pk1b, pk2b, ds, P, C, H, opts, stream);
}
extern "C" int ffno_ffh_bwd_weights_partial(const float* s, const float* db, const void* pk1, const float* b1,
                                            const void* pk1b, float* partial, int P, int C, int H, int nsplit,
                                            const uint32_t* s_amax, const uint32_t* db_amax, int storage, void* stream) {
    return fx_bwd_weights_partial<SplitHf2>(s, db, pk1, b1, pk1b, partial, P, C, H, nsplit, s_amax, db_amax, storage, stream);
}

extern "C" int ffno_ffh_bwd_weights_partial_multi(const ffno_ffwg_desc* descs, int n, int P, int C, int H, int nsplit,
                                                  int storage, int two_addends, void* stream) {
    static_assert(sizeof(ffno_ffwg_desc) == sizeof(FfWgDesc), "descriptor layout");
    if (!descs || n <= 0 || P <= 0 || nsplit <= 0) return FFNO_EINVAL;
    if (storage != FFNO_STORE_F32 && storage != FFNO_STORE_BF16) return FFNO_EINVAL;
    if (two_addends < 0 || two_addends > 1) return FFNO_EINVAL;
    if (!((C == 64 && H == 256) || (C == 32 && H == 128))) return FFNO_EUNSUPPORTED;
    if (two_addends && C != 64) return FFNO_EUNSUPPORTED;      // (the wave-tile chain kernels' shape: they leave the sums unwritten)
    for (int i = 0; i < n; ++i) {
        const ffno_ffwg_desc& d = descs[i];
        if (!d.s || !d.g || !d.pk1 || !d.b1 || !d.pk1b || !d.partial || !d.s_amax || !d.g_amax) return FFNO_EINVAL;
        if (two_addends == 1 && !d.s2) return FFNO_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool b16 = storage == FFNO_STORE_BF16;
    // the table is copied into the kernel arguments (host memory, read at enqueue): FFWG_MAX_BLOCKS blocks per launch
    for (int i0 = 0; i0 < n; i0 += FFWG_MAX_BLOCKS) {
        const int nb = n - i0 < FFWG_MAX_BLOCKS ? n - i0 : FFWG_MAX_BLOCKS;
        FfWgTable tab;
        memcpy(tab.d, descs + i0, sizeof(FfWgDesc) * nb);
        for (int i = nb; i < FFWG_MAX_BLOCKS; ++i) tab.d[i] = tab.d[0];
        const dim3 grid((unsigned)nb * (unsigned)nsplit);
#define WG_LAUNCH(CC, HH, NW, STT, TW) FFNO_LAUNCH((ffh_wgrad_m_multi_kernel<CC, HH, NW, STT, TW>), grid, dim3(NW * 64), 0, st, tab, P, nsplit)
        if (C == 64) {
            if (two_addends == 0) { if (b16) WG_LAUNCH(64, 256, 8, StBf16, 0); else WG_LAUNCH(64, 256, 8, StF32, 0); }
            else { if (b16) WG_LAUNCH(64, 256, 8, StBf16, 1); else WG_LAUNCH(64, 256, 8, StF32, 1); }
        } else {
            if (b16) WG_LAUNCH(32, 128, 4, StBf16, 0); else WG_LAUNCH(32, 128, 4, StF32, 0);
        }
#undef WG_LAUNCH
        const int rc = ffx_launch_status();
        if (rc) return rc;
    }
    return FFNO_OK;
}

// Hardware self-check of the LDS transpose read the weight-gradient kernel is built on: one wave copies `image` (n16 halves) into
// LDS, every lane reads 8 bytes through ds_read_b64_tr_b16 at ITS byte offset, and writes the four halves it received.
__global__ __launch_bounds__(64) void lds_tr16_probe_kernel(const uint16_t* __restrict__ image, int n16,
                                                            const int32_t* __restrict__ byte_off, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(64))) uint16_t img[8192];
    for (int i = threadIdx.x; i < n16; i += 64) img[i] = image[i];
    __syncthreads();
    const uint2 v = plat::lds_read_tr16_b64(reinterpret_cast<const char*>(img) + byte_off[threadIdx.x]);
    out[4 * threadIdx.x + 0] = (uint16_t)(v.x & 0xffffu), out[4 * threadIdx.x + 1] = (uint16_t)(v.x >> 16);
    out[4 * threadIdx.x + 2] = (uint16_t)(v.y & 0xffffu), out[4 * threadIdx.x + 3] = (uint16_t)(v.y >> 16);
}
extern "C" int ffno_lds_tr16_probe(const uint16_t* image, int n16, const int32_t* byte_off, uint16_t* out, void* stream) {
    if (!image || !byte_off || !out || n16 <= 0 || n16 > 8192) return FFNO_EINVAL;
    FFNO_LAUNCH(lds_tr16_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, image, n16, byte_off, out);
    return ffx_launch_status();
}

extern "C" int ffno_ffx_bwd_weights_reduce(const float* partial, float* dW1, float* dW2, float* db1, float* db2, int C,
                                           int H, int nsplit, int accumulate, void* stream) {
    if (!partial || !dW1 || !dW2 || !db1 || !db2 || nsplit <= 0) return FFNO_EINVAL;
    const int part = 2 * H * C + H + C;
    FFNO_LAUNCH(ffx_wgrad_reduce_kernel, dim3((part + 63) / 64), dim3(256), 0, (hipStream_t)stream, partial, dW1, dW2,
                db1, db2, C, H, nsplit, accumulate);
    return ffx_launch_status();
}

extern "C" int ffno_ffx_bwd_weights_reduce_batched(const ffno_fxred_desc* descs_dev, int n, int C, int H, int nsplit,
                                                   void* stream) {
    if (!descs_dev || n <= 0 || nsplit <= 0) return FFNO_EINVAL;
    static_assert(sizeof(ffno_fxred_desc) == sizeof(FxRedDesc), "descriptor layout");
    const int part = 2 * H * C + H + C;
    FFNO_LAUNCH(ffx_wgrad_reduce_batched_kernel, dim3((part + 63) / 64, n), dim3(256), 0, (hipStream_t)stream,
                reinterpret_cast<const FxRedDesc*>(descs_dev), C, H, nsplit);
    return ffx_launch_status();
}
