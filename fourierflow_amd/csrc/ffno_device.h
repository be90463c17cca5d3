// Device-side helpers shared by the F-FNO HIP kernels (gfx950 / CDNA4, wave64).
//
// fp32 matrix work uses v_mfma_f32_32x32x2_f32: exact fp32 (an fmaf chain), 64 cycles/instr/SIMD,
// 157 TFLOP/s chip peak (MI355X_MICROARCH.md "Matrix cores").  Fragment maps (cdna_hip_programming.md s3):
//   A operand: one float per lane, lane l holds A[i = l & 31][k = l >> 5]
//   B operand: one float per lane, lane l holds B[k = l >> 5][j = l & 31]
//   C/D      : 16 floats per lane, reg r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l & 31]
//
// Compiler / ISA specifics live in <ffno_platform.h> (csrc/ffno_platform.h: gfx950).
#pragma once

#include <ffno_platform.h>

// Every launch first clears any stale (non-sticky) error another HIP user of the process left behind
// (e.g. PyTorch probing a capability), so the status a C-ABI entry point returns belongs to OUR launch.
#define FFNO_LAUNCH(...)              \
    do {                              \
        (void)hipGetLastError();      \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

namespace ffno {

static constexpr int kWave = 64;

static inline int device_cu_count() { return plat::cu_count(); }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return plat::mfma_f32_32x32x2(a, b, c);
}

// v_mfma_f32_16x16x4_f32: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D reg r = D[4*(l>>4) + r][l & 15]
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return plat::mfma_f32_16x16x4(a, b, c);
}

// v_mfma_f32_16x16x32_bf16 (4 passes): A[i = l & 15][k = 8*(l>>4) + e], B[k = 8*(l>>4) + e][j = l & 15], D as mfma16
__device__ __forceinline__ f32x4 mfma16_bf16(u32x4 a, u32x4 b, f32x4 c) { return plat::mfma_bf16_16x16x32(a, b, c); }

// ---- split-bf16 ("bf16x3") matrix arithmetic ---------------------------------------------------------------
// An fp32 value is the EXACT sum of three bf16 numbers (its 24 significant bits cut into 8+8+8 by truncation), so an
// fp32 product a*b is recovered from bf16 products: of the nine partial products the six with weight >= 2^-16
// (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) are kept; the dropped ones are <= 3 * 2^-24 |a b| -- the size of one
// fp32 rounding.  v_mfma_f32_32x32x16_bf16 (8 passes, 16x the fp32 MFMA rate; MI355X_MICROARCH.md "Matrix cores")
// accumulates them in fp32, so six of them do the work of eight v_mfma_f32_32x32x2_f32 in 3/8 of the time.
//   A operand: 8 bf16 per lane, lane l holds A[i = l & 31][k = 8*(l>>5) + e];  B likewise B[k = 8*(l>>5) + e][j = l & 31];
//   C/D as the fp32 forms.  A and B use the same (lane-half, slot) -> k map, so any k permutation applied to both cancels.
struct Bf3 {
    u32x4 hi, mid, lo;
};

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return plat::mfma_bf16_32x32x16(a, b, c);
}

__device__ __forceinline__ f32x16 mfma_x3(const Bf3& a, const Bf3& b, f32x16 c) {
    c = mfma_bf16(a.lo, b.hi, c);
    c = mfma_bf16(a.hi, b.lo, c);
    c = mfma_bf16(a.mid, b.mid, c);
    c = mfma_bf16(a.mid, b.hi, c);
    c = mfma_bf16(a.hi, b.mid, c);
    c = mfma_bf16(a.hi, b.hi, c);
    return c;
}

// the same product when b is ONE bf16 plane (data stored as bf16): the three products of mfma_x3 that are not with a zero plane,
// in its order -- bit for bit what mfma_x3(a, {b, 0, 0}, c) gives
__device__ __forceinline__ f32x16 mfma_x3_plane(const Bf3& a, u32x4 b, f32x16 c) {
    c = mfma_bf16(a.lo, b, c);
    c = mfma_bf16(a.mid, b, c);
    c = mfma_bf16(a.hi, b, c);
    return c;
}

__device__ __forceinline__ unsigned f2u(float x) {
    unsigned u;
    __builtin_memcpy(&u, &x, 4);
    return u;
}
__device__ __forceinline__ float u2f(unsigned u) {
    float x;
    __builtin_memcpy(&x, &u, 4);
    return x;
}
// upper halves of two words -> one word (element 0 = x0 in the low half)
__device__ __forceinline__ unsigned pack_hi16(unsigned u0, unsigned u1) {
    return plat::pack_hi16(u0, u1);
}
// exact 3-way truncation split of a pair of floats into packed bf16 pairs
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = f2u(x0), u1 = f2u(x1);
    h = pack_hi16(u0, u1);
    const float r0 = x0 - u2f(u0 & 0xffff0000u), r1 = x1 - u2f(u1 & 0xffff0000u);
    const unsigned v0 = f2u(r0), v1 = f2u(r1);
    m = pack_hi16(v0, v1);
    const float q0 = r0 - u2f(v0 & 0xffff0000u), q1 = r1 - u2f(v1 & 0xffff0000u);
    l = pack_hi16(f2u(q0), f2u(q1));
}
// 8 floats (slots e = 0..7 of one k16 step) -> the three operand planes
__device__ __forceinline__ Bf3 split3_8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
    Bf3 f;
    unsigned h, m, l;
    split3_pair(v0, v1, h, m, l);
    f.hi[0] = h, f.mid[0] = m, f.lo[0] = l;
    split3_pair(v2, v3, h, m, l);
    f.hi[1] = h, f.mid[1] = m, f.lo[1] = l;
    split3_pair(v4, v5, h, m, l);
    f.hi[2] = h, f.mid[2] = m, f.lo[2] = l;
    split3_pair(v6, v7, h, m, l);
    f.hi[3] = h, f.mid[3] = m, f.lo[3] = l;
    return f;
}

// ---- "split-fp16" (fp16x2): fp32-accurate products from THREE half MFMAs --------------------------------------------
// x = hi + lo / 2^11 with hi = fp16(x) (round to nearest: |x - hi| <= 2^-12 |x|) and lo = fp16((x - hi) * 2^11), so that
// |x - hi - lo / 2^11| <= max(2^-24 |x|, 2^-36): as good as fp32 while both planes are normal halves (2.4e-4 <= |x| < 65504),
// a fixed absolute error below; at 65504 the half format overflows -- the caller keeps its data in range (ffno_ffh_*, ffno.h).  Of the four
// partial products hi*hi, hi*lo, lo*hi are kept (lo*lo <= 2^-24 |a b|); half products are exact in fp32 and
// v_mfma_f32_32x32x16_f16 accumulates in fp32.  The two cross products carry the factor 2^11 and go to their own accumulator:
//     main += a.hi b.hi        corr += a.hi b.lo + a.lo b.hi        a b = main + corr / 2^11
// Half the matrix work of split-bf16 (3 instead of 6 MFMAs), two operand planes instead of three, 6 instead of 11 vector
// instructions per split pair -- at the price of the half format's exponent range.
struct Hf2 {
    u32x4 hi, lo;
};
constexpr float kHf2Scale = 2048.f, kHf2Unscale = 1.f / 2048.f;

__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& h, unsigned& l) {
    plat::split2_pair(x0, x1, h, l);      // (h = fp16(x), l = fp16((x - h) 2^11): on the packed instructions)
}
__device__ __forceinline__ Hf2 split2_8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
    Hf2 f;
    unsigned h, l;
    split2_pair(v0, v1, h, l);
    f.hi[0] = h, f.lo[0] = l;
    split2_pair(v2, v3, h, l);
    f.hi[1] = h, f.lo[1] = l;
    split2_pair(v4, v5, h, l);
    f.hi[2] = h, f.lo[2] = l;
    split2_pair(v6, v7, h, l);
    f.hi[3] = h, f.lo[3] = l;
    return f;
}
// one product block on the main / correction accumulator tiles of a 32x32 output tile
__device__ __forceinline__ void mfma_h2(const Hf2& a, const Hf2& b, f32x16& m, f32x16& c) {
    c = plat::mfma_f16_32x32x16(a.lo, b.hi, c);
    c = plat::mfma_f16_32x32x16(a.hi, b.lo, c);
    m = plat::mfma_f16_32x32x16(a.hi, b.hi, m);
}

// ---- single-accumulator split-fp16 products against a BOUNDED operand ------------------------------------------------------
// When one operand of a product block is known to be small (|a| < 2^4: DFT-matrix entries are <= 2 / sqrt(L)), a third plane
// hs = hi * 2^11 (exact: one v_pk_mul_f16 per pair) makes every kept partial product carry the same factor:
//     acc += a.lo b.hi + a.hi b.lo + a.hs b.hi  =  2^11 (a b)        (lo lo / 2^11 dropped: <= 2^-24 |a b|)
// -- three MFMAs on ONE accumulator tile, no correction tile and no fold; the caller divides by 2^11 where it scales anyway.
struct Hf3 {
    u32x4 hi, lo, hs;
};
__device__ __forceinline__ Hf3 split2s_8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
    const Hf2 h = split2_8(v0, v1, v2, v3, v4, v5, v6, v7);
    Hf3 f;
    f.hi = h.hi, f.lo = h.lo;
    FFNO_UNROLL
    for (int w = 0; w < 4; ++w) f.hs[w] = plat::pk_mul_f16(h.hi[w], kHf2Scale);
    return f;
}
// acc += 2^11 (a b), a bounded (Hf3), b any in-range split pair
__device__ __forceinline__ f32x16 mfma_h2s(const Hf3& a, const Hf2& b, f32x16 c) {
    c = plat::mfma_f16_32x32x16(a.lo, b.hi, c);
    c = plat::mfma_f16_32x32x16(a.hi, b.lo, c);
    c = plat::mfma_f16_32x32x16(a.hs, b.hi, c);
    return c;
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
    FFNO_UNROLL
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// ---- operand-split policies -------------------------------------------------------------------------------------------------
// The kernels below are written once over the way an fp32 operand is cut into low-precision MFMA operands:
//   SplitBf3  three bf16 planes, six MFMAs per product block (ffno_device.h "split-bf16"): any fp32 range;
//   SplitHf2  two fp16 planes, three MFMAs ("split-fp16"): half the matrix work, two thirds of the LDS / register operand
//             footprint, 6 instead of 11 vector instructions per split pair -- for data inside the half format's exponent range
//             (2.4e-4 <= |x| < 65504 at full accuracy; gradients are brought there by a power-of-two scale, see ffno_ffh_*).
// Frag = one MFMA operand fragment (NP planes of 16 B per lane).  A product chain runs on a main tile m (any fp32 start value)
// and a correction tile c (zero at the start): mma(a, b, m, c) adds one product block, fold(m, c) leaves the result in m.
struct SplitBf3 {
    using Frag = Bf3;
    static constexpr int NP = 3;
    static constexpr bool SCALED = false;          // gradients need no range scale
    static __device__ __forceinline__ Frag split8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
        return split3_8(v0, v1, v2, v3, v4, v5, v6, v7);
    }
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& m, f32x16&) { m = mfma_x3(a, b, m); }
    static __device__ __forceinline__ void fold(f32x16&, const f32x16&) {}
    static __device__ __forceinline__ u32x4 plane(const Frag& f, int p) { return p == 0 ? f.hi : (p == 1 ? f.mid : f.lo); }
    static __device__ __forceinline__ void set_plane(Frag& f, int p, u32x4 v) {
        if (p == 0) f.hi = v;
        if (p == 1) f.mid = v;
        if (p == 2) f.lo = v;
    }
    // 4 consecutive values -> two words per plane
    static __device__ __forceinline__ void split4(float x, float y, float z, float w, uint2* planes) {
        unsigned h0, m0, l0, h1, m1, l1;
        split3_pair(x, y, h0, m0, l0);
        split3_pair(z, w, h1, m1, l1);
        planes[0] = make_uint2(h0, h1), planes[1] = make_uint2(m0, m1), planes[2] = make_uint2(l0, l1);
    }
};
struct SplitHf2 {
    using Frag = Hf2;
    static constexpr int NP = 2;
    static constexpr bool SCALED = true;
    static __device__ __forceinline__ Frag split8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
        return split2_8(v0, v1, v2, v3, v4, v5, v6, v7);
    }
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& m, f32x16& c) { mfma_h2(a, b, m, c); }
    static __device__ __forceinline__ void fold(f32x16& m, const f32x16& c) {
        FFNO_UNROLL
        for (int i = 0; i < 16; ++i) m[i] = __builtin_fmaf(c[i], kHf2Unscale, m[i]);
    }
    static __device__ __forceinline__ u32x4 plane(const Frag& f, int p) { return p == 0 ? f.hi : f.lo; }
    static __device__ __forceinline__ void set_plane(Frag& f, int p, u32x4 v) {
        if (p == 0) f.hi = v;
        if (p == 1) f.lo = v;
    }
    static __device__ __forceinline__ void split4(float x, float y, float z, float w, uint2* planes) {
        unsigned h0, l0, h1, l1;
        split2_pair(x, y, h0, l0);
        split2_pair(z, w, h1, l1);
        planes[0] = make_uint2(h0, h1), planes[1] = make_uint2(l0, l1);
    }
};

// ---- range words: how the fp16x2 kernels stay inside the half format's exponent range ---------------------------------------
// A RANGE WORD is a caller-owned 32-bit device word holding the bit pattern of max |x| over a tensor (non-negative floats order
// like their bit patterns, so producers fold their maxima with one atomicMax per workgroup; zero it before the first producer).
// A kernel that cuts an operand into fp16 planes reads the range word of that operand and derives, on the device, the power of
// two that brings the operand's magnitude BOUND to 2^target: the data is multiplied by it while it is staged and the results are
// divided by it again -- both exact.  No host round trip, no assumption about the data: any finite fp32 input is in range.
//   bound = max|x| * 2^extra_log2   (extra_log2: what the kernel may add before the split -- e.g. +1 for a sum of two tensors,
//                                    + log2(2 sqrt(L)) for an orthonormal DFT of length L with the c_k = 2 weights)
//   scale * bound in (2^(target-1), 2^target]
// A zero word (all-zero tensor) or a non-finite maximum gives scale 1 (non-finite data stays non-finite, as in fp32 arithmetic).
__device__ __forceinline__ float range_scale(unsigned amax_bits, int extra_log2, int target) {
    const int e = (int)((amax_bits >> 23) & 0xffu);          // max|x| < 2^(e - 126)
    if (amax_bits == 0u || e == 255) return 1.f;
    int sexp = target - (e - 126) - extra_log2;
    sexp = sexp < -120 ? -120 : (sexp > 120 ? 120 : sexp);   // the scale and its reciprocal stay normal floats
    return u2f((unsigned)(sexp + 127) << 23);
}
__device__ __forceinline__ int ceil_log2_int(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
// fold a per-lane maximum (>= 0) of a workgroup of NWAVES waves into a range word: wave butterflies, one LDS word per wave, one
// atomic per workgroup.  Every thread of the workgroup must call it (it contains a barrier); `red` = NWAVES floats of LDS.
__device__ __forceinline__ void range_fold(float m, float* red, int nwaves, unsigned* word) {
    FFNO_UNROLL
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nwaves; ++w) m = fmaxf(m, red[w]);
        if (f2u(m) != 0u) atomicMax(word, f2u(m));      // (a zero maximum never raises the word: no atomic for it)
    }
}

// ---- activation storage formats ------------------------------------------------------------------------------------------
// The hot kernels are written once over the format their ACTIVATION tensors have in HBM (layer inputs / outputs, branch outputs,
// saved feed-forward inputs, gradients of those): StF32 (the parity path: the reference is `precision: 32`) or StBf16 (the
// "bf16 storage twins": half the activation bytes; values are widened as they are loaded and rounded to nearest even where they
// are stored, everything between -- operand splits, accumulation, spectra, weights, reductions -- is the fp32-grade arithmetic
// of the fp32 path).  A bf16 kernel run on inputs x gives bit for bit bf16(fp32 kernel run on float(x)).
// Loads that are requested long before their use keep the RAW words (ldr* -> Raw*) and widen them (w*) where they are consumed:
// arithmetic placed next to a load makes the wave wait for it there (s_waitcnt vmcnt), i.e. turns the prefetch into a stall.
struct StF32 {
    using T = float;
    using Raw4 = float4;
    using Raw2 = float2;
    using Raw1 = float;
    static constexpr bool BF16 = false;
    static constexpr int BYTES = 4;
    static __device__ __forceinline__ Raw4 ldr4(const T* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ Raw2 ldr2(const void* p) { return *reinterpret_cast<const float2*>(p); }
    static __device__ __forceinline__ Raw1 ldr1(const T* p) { return *p; }
    static __device__ __forceinline__ Raw4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ float4 w4(Raw4 r) { return r; }
    static __device__ __forceinline__ float2 w2(Raw2 r) { return r; }
    static __device__ __forceinline__ float w1(Raw1 r) { return r; }
    static __device__ __forceinline__ float4 ld4(const T* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ float2 ld2(const void* p) { return *reinterpret_cast<const float2*>(p); }
    static __device__ __forceinline__ float ld1(const T* p) { return *p; }
    static __device__ __forceinline__ void st4(T* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
    static __device__ __forceinline__ void st2(void* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
    static __device__ __forceinline__ void st1(void* p, float v) { *reinterpret_cast<float*>(p) = v; }
    static __device__ __forceinline__ float rnd(float x) { return x; }      // what a stored value reads back as
};
struct StBf16 {
    using T = uint16_t;
    using Raw4 = uint2;
    using Raw2 = unsigned;
    using Raw1 = uint16_t;
    static constexpr bool BF16 = true;
    static constexpr int BYTES = 2;
    static __device__ __forceinline__ Raw4 ldr4(const T* p) { return *reinterpret_cast<const uint2*>(p); }
    static __device__ __forceinline__ Raw2 ldr2(const void* p) { return *reinterpret_cast<const unsigned*>(p); }
    static __device__ __forceinline__ Raw1 ldr1(const T* p) { return *p; }
    static __device__ __forceinline__ Raw4 zero4() { return make_uint2(0u, 0u); }
    static __device__ __forceinline__ float4 w4(Raw4 w) {
        return make_float4(u2f(w.x << 16), u2f(w.x & 0xffff0000u), u2f(w.y << 16), u2f(w.y & 0xffff0000u));
    }
    static __device__ __forceinline__ float2 w2(Raw2 w) { return make_float2(u2f(w << 16), u2f(w & 0xffff0000u)); }
    static __device__ __forceinline__ float w1(Raw1 r) { return u2f((unsigned)r << 16); }
    static __device__ __forceinline__ float4 ld4(const T* p) {
        const uint2 w = *reinterpret_cast<const uint2*>(p);
        return make_float4(u2f(w.x << 16), u2f(w.x & 0xffff0000u), u2f(w.y << 16), u2f(w.y & 0xffff0000u));
    }
    static __device__ __forceinline__ float2 ld2(const void* p) {
        const unsigned w = *reinterpret_cast<const unsigned*>(p);
        return make_float2(u2f(w << 16), u2f(w & 0xffff0000u));
    }
    static __device__ __forceinline__ float ld1(const T* p) { return u2f((unsigned)*p << 16); }
    static __device__ __forceinline__ void st4(T* p, float4 v) {
        *reinterpret_cast<uint2*>(p) = make_uint2(plat::pack_bf16(v.x, v.y), plat::pack_bf16(v.z, v.w));
    }
    static __device__ __forceinline__ void st2(void* p, float2 v) { *reinterpret_cast<unsigned*>(p) = plat::pack_bf16(v.x, v.y); }
    static __device__ __forceinline__ void st1(void* p, float v) { *reinterpret_cast<uint16_t*>(p) = (uint16_t)plat::pack_bf16(v, 0.f); }
    static __device__ __forceinline__ float rnd(float x) { return u2f(plat::pack_bf16(x, 0.f) << 16); }
};
template <class ST>
__device__ __forceinline__ float4 st_rnd4(float4 v) {
    return make_float4(ST::rnd(v.x), ST::rnd(v.y), ST::rnd(v.z), ST::rnd(v.w));
}

// row of the 32x32 D tile held in accumulator register r by a lane in half `half` (= lane >> 5)
__device__ __forceinline__ int drow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ float wave_sum(float v) {
    FFNO_UNROLL
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

}  // namespace ffno
