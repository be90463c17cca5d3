// Fused spectral branch of the factorized Fourier layer on the bf16 matrix cores at fp32 accuracy ("bf16x3").
//
// Same operator as spectral_fused_kernel (spectral.hip): SpectralConv2d.forward_fourier along ONE axis
// (reference fourierflow/modules/factorized_fno/grid_2d.py:58-72 / :76-90: rfft(norm='ortho') -> [:K] -> complex einsum with
// fourier_weight -> zero-filled irfft) and its adjoint, with the spectrum tile resident in LDS -- rebuilt around three
// measurements of round 1 (profiles/r01_v8_kernel_stats.md):
//   * the kernel is MATRIX-bound on the fp32 MFMA (27 us of v_mfma_f32_32x32x2_f32 per paired launch at batch 32): every
//     product here is six v_mfma_f32_32x32x16_bf16 over exact three-way bf16 splits of both operands (ffno_device.h),
//     3/8 of the matrix time at unchanged accuracy.  The DFT matrices are split once per wave and kept in registers; the
//     activations are split as they arrive from HBM / LDS;
//   * every 8-line workgroup streamed all 512 KiB of weight planes from L2 (134 MB of L2 -> CU traffic per launch):
//     a workgroup now owns 16 lines, so the per-mode mix is a full 32-row tile ((line, re/im) pairs) and the weight stream
//     per line halves; the weights arrive pre-split and pre-permuted in MFMA fragment order (ffno_spectral_x3_pack), one
//     coalesced 16-byte load per lane, plane and fragment, double-buffered one k-step ahead;
//   * a wave's input line was requested in four dependent trips: the whole line (16 KiB) is now requested at once, and the
//     second line of the wave is requested while the first is being transformed.
//
// Workgroup = 8 waves = 16 lines of one axis.  LDS tile XS[line][kk = 2k + ri][c] (row stride 68 floats, line stride
// 32 * 68 + 8: the 16-byte row reads of phase 2 are bank-conflict free).
//   phase 1  wave w: truncated DFT of lines 2w, 2w+1:   X[kk][c] = sum_n F[kk][n] x[n][c]        (A = F in registers)
//   phase 2  wave w: modes k = w, w + 8: in-place channel mix, rows = (line, re/im) of all 16 lines:
//            P1 = X Wr, P2 = X Wi;  Yr = P1[re] - P2[im], Yi = P2[re] + P1[im]   (adjoint: planes of W^T, signs flipped)
//   phase 3  wave w: zero-padded inverse DFT of lines 2w, 2w+1 with the accumulate / residual epilogue.
// Column tile ct of every 32-column MFMA tile holds channels c = 2 j + ct (j = lane & 31), so a lane owns two adjacent
// channels: 8-byte global loads / stores and LDS accesses throughout.
#include <type_traits>

#include "ffno_device.h"
#include "ffno_lines.h"
#include "ffno_x3_dft.h"
#include "ffno_infer_body.h"
#include "ffno.h"

namespace ffno {

struct X3Cfg {
    static constexpr int C = 64;
    static constexpr int NW = 8;               // waves per workgroup (one or two lines each: 8- or 16-line tiles)
    static constexpr int KK = 32;              // (mode, re/im) rows per line: K <= 16
    static constexpr int RS = 68;              // row stride (floats): +4 shifts the re / im rows of a line by 4 banks
    static constexpr int LSF = KK * RS + 8;    // line stride (floats): +8 shifts consecutive lines by 8 banks
    static constexpr int FRAG = 3 * 64;        // u32x4 per packed fragment (three planes x 64 lanes)
    static constexpr int MODE_FRAGS = 16;      // per mode: 4 k-steps x (2 planes x 2 column tiles)
};

struct X3Args {
    const float* in;
    float* out;
    const float* resid;
    float* spec_save;
    const u32x4* wpk;     // packed weights (ffno_spectral_x3_pack), NULL = no mix ('low-pass')
    const float* tw;
    int R, L, K;
    LineMap lm;
    int fwd_ck, inv_ck, conj_t, accumulate;
    // range words (ffno_device.h): in_amax = max |in| (NULL: the data is known to fit the half format unscaled) -- with fp16x2
    // packs the spectrum tile is held multiplied by the power of two derived from it; out_amax (optional) receives max |out|
    const unsigned* in_amax;
    unsigned* out_amax;
    // many-mode kernel only: precomputed DFT-matrix fragments (x3k_dft_frags_kernel), NULL = built from the twiddle table
    const u32x4* dft;
    // MIXOUT instances (the inference layer's first kernel, infer.hip): the MIXED spectrum of every line leaves the launch instead
    // of its inverse transform -- as the split-fp16 operand fragments the second kernel multiplies (mix_out: 8 fragments of 64 lanes
    // x 16 B per line) beside the factor that turns its accumulators into samples (mix_scale: one float per line)
    u32x4* mix_out;
    float* mix_scale;
    // MIXOUT instances, lines of <= 64 samples (the wave holds the whole line before its first product): every line is scaled from
    // ITS OWN maximum instead of the range word of the tensor (FFNO_BRANCH_SELF_RANGE) -- no range word, no atomics, no amax pass
    int self_range;
};

// ---- weight packing --------------------------------------------------------------------------------------------------
// planes[k][p][i][o] (p = re | im; ffno_fw_pack: forward planes, or the transposed planes for the adjoint) ->
// fragment (k, st, p, t), lane (j, half), slot e  <-  planes[k][p][i = 16 st + 8 half + e][o = 2 j + t], split in three.
// format 1 (FFNO_PLANES_FP16X2): two fp16 planes per fragment (SplitHf2), fragments of a mode ordered (p, t, st) instead of
// (st, p, t) so that the k-steps of one output tile are consecutive (the mix then needs one correction accumulator at a time).
struct X3PackDesc {
    const float* planes;
    u32x4* dst;
    int K, format;
};

__global__ __launch_bounds__(256) void x3_pack_kernel(const X3PackDesc* __restrict__ descs) {
    constexpr int C = X3Cfg::C;
    const X3PackDesc d = descs[blockIdx.y];
    const int nfrag = d.K * X3Cfg::MODE_FRAGS;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nfrag * 64) return;
    const int frag = t >> 6, lane = t & 63, j = lane & 31, half = lane >> 5;
    const int pt = frag & 3, st = (frag >> 2) & 3, k = frag >> 4;
    const int p = pt >> 1, tt = pt & 1;
    const float* src = d.planes + ((long)(k * 2 + p) * C + (16 * st + 8 * half)) * C + 2 * j + tt;
    float v[8];
    FFNO_UNROLL
    for (int e = 0; e < 8; ++e) v[e] = src[(long)e * C];
    if (d.format == 1) {
        const Hf2 f = split2_8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        const int fo = (k * 4 + pt) * 4 + st;
        d.dst[(fo * 2 + 0) * 64 + lane] = f.hi;
        d.dst[(fo * 2 + 1) * 64 + lane] = f.lo;
        return;
    }
    if (d.format == 2) {
        // format 2 (FFNO_PLANES_FP16X2_M16): B fragments of v_mfma_f32_16x16x32_f16 -- the mix of the 4-line tiles (8 live rows) runs
        // 16-row products.  fragment (k, p, t16, s2): lane (j16 = lane & 15, g = lane >> 4), slot e  <-
        //     planes[k][p][i = 32 s2 + 8 g + e][o = 4 j16 + t16]        (p = re | im; t16 = 0..3; s2 = 0..1: two k32 steps)
        // two fp16 planes per fragment, fragments of a mode ordered ((p, t16), s2): 16 per mode, the size of format 1.
        const int fm = frag & 15, s2 = fm & 1, t16 = (fm >> 1) & 3, pp = fm >> 3;
        const int j16 = lane & 15, g = lane >> 4;
        const float* src2 = d.planes + ((long)(k * 2 + pp) * C + (32 * s2 + 8 * g)) * C + 4 * j16 + t16;
        float w[8];
        FFNO_UNROLL
        for (int e = 0; e < 8; ++e) w[e] = src2[(long)e * C];
        const Hf2 f = split2_8(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
        d.dst[(frag * 2 + 0) * 64 + lane] = f.hi;
        d.dst[(frag * 2 + 1) * 64 + lane] = f.lo;
        return;
    }
    const Bf3 f = split3_8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    d.dst[(frag * 3 + 0) * 64 + lane] = f.hi;
    d.dst[(frag * 3 + 1) * 64 + lane] = f.mid;
    d.dst[(frag * 3 + 2) * 64 + lane] = f.lo;
}

__device__ __forceinline__ Bf3 x3_load_frag(const u32x4* __restrict__ pk, int frag, int lane) {
    Bf3 f;
    f.hi = pk[(frag * 3 + 0) * 64 + lane];
    f.mid = pk[(frag * 3 + 1) * 64 + lane];
    f.lo = pk[(frag * 3 + 2) * 64 + lane];
    return f;
}

// ---- DFT-matrix fragment table of the many-mode kernel ------------------------------------------------------------------------
// The many-mode kernel gives a wave ONE line, so nothing amortises the construction of its DFT-matrix fragments: per product
// block ~90 vector instructions (table lookups, index bookkeeping, the fp16 split) against 6 MFMAs -- measured 20 vector
// instructions per MFMA over the whole launch (profiles/r04_x3k_sq_counters.md).  The fragments only depend on (L, K, flags), so
// they are built ONCE into a table in MFMA lane order (two fp16 planes per fragment: hi, lo; the 2^11 hi plane is one
// v_pk_mul_f16 per word in the kernel) and every wave loads them (L1 / L2 hits: all waves of a CU walk the same table):
//   forward  part: fragment ((rt * nchunks + chunk) * 4 + u)      rt < RT = KKT / 32, chunk < nchunks, u < 4
//   inverse  part: fragment (nfwd + tile * NST + st)               tile < ceil(L / 32), st < NST = KKT / 16
// Values = exactly the expressions of spectral_x3k_body's on-the-fly path (bit-identical results with and without a table).
__global__ __launch_bounds__(64) void x3k_dft_frags_kernel(const float* __restrict__ tw, int L, int K, int fwd_ck, int inv_ck,
                                                           X3kDft d, u32x4* __restrict__ out) {
    const int frag = blockIdx.x, lane = threadIdx.x, j = lane & 31, half = lane >> 5;
    float f[8];
    if (frag < d.nfwd) {
        const int u = frag & 3, chunk = (frag >> 2) % d.nchunks, rt = (frag >> 2) / d.nchunks;
        const int kk = 32 * rt + j, k = kk >> 1, ri = kk & 1;
        const bool rowok = kk < 2 * K;
        const float ck = (fwd_ck && !(k == 0 || 2 * k == L)) ? 2.f : 1.f;
        const float amul = rowok ? (ri ? -ck : ck) : 0.f;
        const int km = rowok ? k : 0;
        FFNO_UNROLL
        for (int e = 0; e < 8; ++e) {
            const int n = 16 * (4 * chunk + u) + 8 * half + e;
            f[e] = n < L ? amul * tw[(ri ? L : 0) + (int)(((long)km * n) % L)] : 0.f;
        }
    } else {
        const int g = frag - d.nfwd, st = g % d.NST, tile = g / d.NST;
        const int n = 32 * tile + j;
        FFNO_UNROLL
        for (int e = 0; e < 8; ++e) {
            const int kk = 16 * st + 8 * half + e, t = kk >> 1, part = kk & 1;
            const float ck = (inv_ck && !(t == 0 || 2 * t == L)) ? 2.f : 1.f;
            f[e] = (kk < 2 * K && n < L) ? (part ? -ck : ck) * tw[(part ? L : 0) + (int)(((long)n * t) % L)] : 0.f;
        }
    }
    const Hf2 h = split2_8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
    out[(frag * 2 + 0) * 64 + lane] = h.hi;
    out[(frag * 2 + 1) * 64 + lane] = h.lo;
}
// ---- the branch --------------------------------------------------------------------------------------------------------
// NL = lines per workgroup: 16 (two per wave; the per-mode mix fills its 32-row tile) or 8 (one per wave: launches with few
// lines -- batch-1 rollout: 64 lines per axis -- spread over twice as many CUs; the mix then uses rows 0..15 of the tile,
// rows 16..31 repeat them and are dropped).
// ST = storage format of the activation tensors in / out / resid (ffno_device.h): with StBf16 a loaded sample IS one bf16 plane,
// so the forward DFT needs three MFMAs per product block instead of six; the saved spectra stay fp32.
// DFTH2 (= MIXH2: the launches whose operands carry range words): the two DFT phases on split-fp16 as well -- the DFT matrices
// are the bounded operand of single-accumulator products (ffno_device.h: Hf3, three MFMAs per product block instead of the
// six of the bf16 split, 6 instead of 11 vector instructions per split pair), the samples are multiplied by the power of two
// that brings max |in| (range word) to 2^10 while they are split, and each line's mixed spectrum by one derived from ITS OWN
// maximum (the wave reads the whole line from LDS anyway) -- both exact, both undone where the result is scaled anyway.
// MIXOUT: phases 1 and 2 only -- each line's mixed spectrum is written out as MFMA operand fragments (X3Args::mix_out) for the
// kernel that runs both inverse transforms and the feed-forward of an inference layer (infer.hip); nothing else is stored.
// EXTLDS: the spectrum tile and the twiddle table live in caller-provided LDS (`ext_lds`: NL LSF + 2 L floats, 16-byte aligned) instead
// of this function's own static tile + the launch's dynamic window: the persistent inference kernel (infer_stack_kernel) runs this
// body and the second inference kernel's by turns in ONE allocation.
// DFTTAB (split-fp16 DFT): the DFT-matrix fragments of both transforms come from the precomputed table A.dft (x3k_dft_frags_kernel:
// the same values, bit for bit) instead of being built from the twiddle table by every wave of every launch -- 32 LDS reads, the
// index arithmetic and four three-way splits cost a wave 2.2 us per transform (tools/trace_stack.py: the persistent inference
// kernel), and the twiddle staging with its workgroup barrier goes too.
template <int NL, bool MIXH2, class ST = StF32, bool MIXOUT = false, bool EXTLDS = false, bool DFTTAB = false>
__device__ __forceinline__ void spectral_x3_body(const X3Args A, int bidx, int skew_cycles, float* ext_lds = nullptr, int tid_in = 0,
                                                 unsigned long long* tr = nullptr) {      // tr: diagnostic stamps (infer_stack_kernel<.., TRACE>)
    using F = X3Cfg;
    constexpr int C = F::C, RS = F::RS, LSF = F::LSF;
    constexpr int NLW = NL / F::NW;                 // lines per wave
    constexpr bool DFTH2 = MIXH2;
    using DftFrag = typename std::conditional<DFTH2, Hf3, Bf3>::type;      // DFT-matrix fragments
    static_assert(NL == 16 || NL == 8, "16 or 8 lines per workgroup");
    __shared__ __attribute__((aligned(16))) float XS_own[EXTLDS ? 4 : NL * F::LSF];
    float* XS = XS_own;
    float* tws;
    if constexpr (EXTLDS) {
        XS = ext_lds;
        tws = ext_lds + NL * F::LSF;
    } else {
        FFNO_DYN_SMEM(smem);
        tws = reinterpret_cast<float*>(smem);
    }

    const float* __restrict__ in = A.in;
    const int R = A.R, L = A.L, K = A.K;
    const LineMap lm = A.lm;
    // range scale of the spectrum tile (fp16x2 mix): applied where phase 1 writes the tile, removed where phase 3 stores; the
    // saved spectrum stays unscaled.  |X[k]| <= 2 sqrt(L) max|x| (orthonormal DFT, c_k <= 2): that bound goes to 2^15, so no
    // finite input can push a split operand of the mix past the half format's 65504.
    const float rs_w = (MIXH2 && A.wpk && A.in_amax) ? range_scale(*A.in_amax, 1 + (ceil_log2_int(L) + 1) / 2, 15) : 1.f;
    const float rrs = 1.f / rs_w;
    // fp16x2 DFT: samples x sx (max |in| -> 2^10); the accumulators then hold 2^11 sx X
    const float sx_w = (DFTH2 && A.in_amax) ? range_scale(*A.in_amax, 0, 10) : 1.f;
    const float unx_w = DFTH2 ? kHf2Unscale / sx_w : 1.f;      // accumulator -> spectrum
    float omax = 0.f;                  // max |out| over what this thread stores
    __shared__ float rfold[F::NW];
    __shared__ float lrrs[MIXOUT ? NL : 1];      // self-ranged lines: 1 / (tile scale) of every line, phase 1 -> phase 3' (not in registers)

    // (EXTLDS: the thread index comes from the caller -- the persistent kernel hands it over through an opaque move per phase, so that
    //  nothing derived from it is hoisted out of its phase loop and kept in registers across the other kernel's body)
    const int tidx = EXTLDS ? tid_in : (int)threadIdx.x;
    const int lane = tidx & 63, wave = tidx >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int lw = NLW * wave;                     // first of this wave's lines inside the tile
    const int line0 = bidx * NL + lw;
    const bool live0 = line0 < R, live1 = NLW > 1 && line0 + 1 < R;
    // Addresses = wave-uniform part (tensor base + sample index x element stride: scalar registers) + a 32-bit per-lane byte
    // offset (line base + the lane's channel pair + its half-wave's sample / row offset): one VGPR per line instead of a
    // 64-bit address per access.  (The host side refuses tensors of 4 GiB or more.)
    const long es = lm.elem_stride;
    // Lines past the end of the axis read (and transform) line R - 1 again: rows of the tile are independent, and nothing of
    // a dead line is ever stored.
    const unsigned lo0 = (unsigned)((lm.base(min(line0, R - 1)) + 2 * j) * ST::BYTES);
    const unsigned lo1 = (unsigned)((lm.base(min(line0 + 1, R - 1)) + 2 * j) * ST::BYTES);
    if (skew_cycles > 0) plat::sleep_cycles(skew_cycles);      // optional start skew (see spectral_x3_pair_kernel)

    // ---------------- phase 1: truncated forward DFT of the wave's two lines ----------------
    {
        const int kk = j, k = kk >> 1, ri = kk & 1;              // A-operand row of this lane
        const bool rowok = kk < 2 * K;
        const float ck = (A.fwd_ck && !(k == 0 || 2 * k == L)) ? 2.f : 1.f;
        const float amul = rowok ? (ri ? -ck : ck) : 0.f;
        const int tbase = ri ? L : 0;
        const int km = rowok ? k : 0;

        // one 64-sample chunk of one line: k-step u, slot e -> sample 16 (4 chunk + u) + 8 half + e (bf16 storage: the word
        // holding the lane's two channels, as it came)
        using Raw = typename std::conditional<ST::BF16, unsigned, float2>::type;
        Raw raw[4][8];
        // samples past the end of the line (L not a multiple of 16) re-read sample L - 1: finite data under a zero of F
        const unsigned esb = (unsigned)(es * ST::BYTES);
        auto load_rows = [&](int chunk, int u, unsigned lo) {
            FFNO_UNROLL
            for (int e = 0; e < 8; ++e) {
                const int n = min(16 * (4 * chunk + u) + 8 * half + e, L - 1);
                raw[u][e] = *reinterpret_cast<const Raw*>(reinterpret_cast<const char*>(in) + (lo + (unsigned)n * esb));
            }
        };
        // DFT-matrix fragments of one chunk (exact three-way splits; built while the loads are in flight).  The table index
        // k n mod L advances incrementally: +k per sample, +8 k across the other half-wave's samples.
        DftFrag Ff[4];
        const int k8 = (km * 8) % L;
        static_assert(!DFTTAB || DFTH2, "the table holds split-fp16 fragments");
        auto build_F = [&](int chunk) {
            if constexpr (DFTTAB) {
                FFNO_UNROLL
                for (int u = 0; u < 4; ++u) Ff[u] = x3k_load_dft(A.dft, chunk * 4 + u, lane);
                return;
            }
            int idx = (km * (64 * chunk + 8 * half)) % L;
            FFNO_UNROLL
            for (int u = 0; u < 4; ++u) {
                float f[8];
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) {
                    const int n = 16 * (4 * chunk + u) + 8 * half + e;
                    f[e] = n < L ? amul * tws[tbase + idx] : 0.f;
                    idx += km;
                    if (idx >= L) idx -= L;
                }
                idx += k8;
                if (idx >= L) idx -= L;
                if constexpr (DFTH2)
                    Ff[u] = split2s_8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
                else
                    Ff[u] = split3_8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
            }
        };
        const int nchunks = (L + 63) >> 6;
        // the first line is requested before anything else: the twiddle table is staged (and the DFT-matrix fragments are
        // built from it) while those 16 KiB are on their way
        if constexpr (DFTTAB) build_F(0);      // (requested first: they are back before the line)
        FFNO_UNROLL
        for (int u = 0; u < 4; ++u) load_rows(0, u, lo0);
        if constexpr (!DFTTAB) {
            for (int i = tidx; i < 2 * L; i += blockDim.x) tws[i] = A.tw[i];
            __syncthreads();
        }
        if (tr && tidx == 0) tr[3] = plat::realtime();
        if constexpr (!DFTTAB) build_F(0);
        if (tr && tidx == 0) tr[6] = plat::realtime();
        FFNO_UNROLL
        for (int ln = 0; ln < NLW; ++ln) {
            f32x16 acc0 = zero16(), acc1 = zero16();
            float sx = sx_w, unx = unx_w, rs = rs_w;      // this line's scales (the tensor's, or the line's own)
            if constexpr (MIXOUT && !ST::BF16) {
                if (A.self_range) {
                    // the whole line (L <= 64: one chunk) sits in `raw`: its maximum over samples and channels takes 63 maxima per lane
                    // and six butterflies, and replaces the tensor's range word -- scales are per line anyway from here on (rs goes
                    // into the tile rows of this line, the mix is linear per row, phase 3' folds 1 / rs into the line's mix_scale)
                    float lmx = 0.f;
                    FFNO_UNROLL
                    for (int u = 0; u < 4; ++u) {
                        FFNO_UNROLL
                        for (int e = 0; e < 8; ++e) lmx = fmaxf(lmx, fmaxf(fabsf(raw[u][e].x), fabsf(raw[u][e].y)));
                    }
                    FFNO_UNROLL
                    for (int sh = 32; sh >= 1; sh >>= 1) lmx = fmaxf(lmx, __shfl_xor(lmx, sh));
                    const unsigned lw_ = f2u(lmx);
                    sx = range_scale(lw_, 0, 10);
                    unx = kHf2Unscale / sx;
                    rs = A.wpk ? range_scale(lw_, 1 + (ceil_log2_int(L) + 1) / 2, 15) : 1.f;
                    if (lane == 0) lrrs[lw + ln] = 1.f / rs;
                }
            }
            if (tr && tidx == 0) tr[ln ? 9 : 7] = plat::realtime();
            FFNO_NOUNROLL
            for (int chunk = 0; chunk < nchunks; ++chunk) {
                if (nchunks > 1 && (ln | chunk)) build_F(chunk);      // one chunk (L <= 64): the fragments serve both lines
                // as each k-step's rows are consumed, the rows of the next (chunk, line) are requested into the same registers
                const bool more = chunk + 1 < nchunks;
                FFNO_UNROLL
                for (int u = 0; u < 4; ++u) {
                    if constexpr (DFTH2) {
                        // split-fp16 samples (bf16 storage: widened first -- the same arithmetic as the fp32 kernel on float(x))
                        float2 w[8];
                        FFNO_UNROLL
                        for (int e = 0; e < 8; ++e) {
                            w[e] = ST::w2(raw[u][e]);
                            w[e].x *= sx, w[e].y *= sx;
                        }
                        const Hf2 b0 = split2_8(w[0].x, w[1].x, w[2].x, w[3].x, w[4].x, w[5].x, w[6].x, w[7].x);
                        const Hf2 b1 = split2_8(w[0].y, w[1].y, w[2].y, w[3].y, w[4].y, w[5].y, w[6].y, w[7].y);
                        if (more)
                            load_rows(chunk + 1, u, ln ? lo1 : lo0);
                        else if (ln == 0 && NLW > 1)
                            load_rows(0, u, lo1);
                        acc0 = mfma_h2s(Ff[u], b0, acc0);
                        acc1 = mfma_h2s(Ff[u], b1, acc1);
                    } else if constexpr (ST::BF16) {
                        // the samples are bf16: one operand plane (low halves = even channel, high halves = odd channel)
                        u32x4 b0, b1;
                        FFNO_UNROLL
                        for (int q = 0; q < 4; ++q) {
                            b0[q] = plat::pack_lo16(raw[u][2 * q], raw[u][2 * q + 1]);
                            b1[q] = plat::pack_hi16(raw[u][2 * q], raw[u][2 * q + 1]);
                        }
                        if (more)
                            load_rows(chunk + 1, u, ln ? lo1 : lo0);
                        else if (ln == 0 && NLW > 1)
                            load_rows(0, u, lo1);
                        acc0 = mfma_x3_plane(Ff[u], b0, acc0);
                        acc1 = mfma_x3_plane(Ff[u], b1, acc1);
                    } else {
                        const Bf3 b0 = split3_8(raw[u][0].x, raw[u][1].x, raw[u][2].x, raw[u][3].x, raw[u][4].x, raw[u][5].x,
                                                raw[u][6].x, raw[u][7].x);
                        const Bf3 b1 = split3_8(raw[u][0].y, raw[u][1].y, raw[u][2].y, raw[u][3].y, raw[u][4].y, raw[u][5].y,
                                                raw[u][6].y, raw[u][7].y);
                        if (more)
                            load_rows(chunk + 1, u, ln ? lo1 : lo0);
                        else if (ln == 0 && NLW > 1)
                            load_rows(0, u, lo1);
                        acc0 = mfma_x3(Ff[u], b0, acc0);
                        acc1 = mfma_x3(Ff[u], b1, acc1);
                    }
                }
            }
            float* xs = XS + (lw + ln) * LSF + 2 * j;
            const bool live = ln ? live1 : live0;
            FFNO_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row = drow(r, half);
                if (row < 2 * K) {
                    const float2 v = make_float2(acc0[r] * unx, acc1[r] * unx);      // (unx = 1 on the bf16 split)
                    *reinterpret_cast<float2*>(xs + row * RS) = make_float2(v.x * rs, v.y * rs);
                    if (A.spec_save && live)
                        *reinterpret_cast<float2*>(A.spec_save + (((long)(row >> 1) * R + line0 + ln) * 2 + (row & 1)) * C + 2 * j) = v;
                }
            }
            if (tr && tidx == 0) tr[ln ? 10 : 8] = plat::realtime();
        }
    }
    // weight fragments of phase 2: a ring of RING, requested RING products (6 RING MFMAs ~ 0.6 us at RING = 8) before they are
    // used -- a wave streams 96 KiB of fragments from L2 and each round trip costs ~0.5-1 us, so the depth of the ring is what
    // sets its streaming rate (measured: the batch-1 launch, 16 workgroups, is paced by this stream, not by the MFMAs).  The
    // first RING fragments of the wave's first mode are requested on this side of the barrier.
    constexpr int RING = 8;
    using MixFrag = typename std::conditional<MIXH2, Hf2, Bf3>::type;
    constexpr int MNP = MIXH2 ? 2 : 3;                       // planes per packed weight fragment
    auto load_w = [&](const u32x4* __restrict__ wk, int f) {
        MixFrag w;
        if constexpr (MIXH2) {
            w.hi = wk[(f * 2 + 0) * 64 + lane];
            w.lo = wk[(f * 2 + 1) * 64 + lane];
        } else {
            w = x3_load_frag(wk, f, lane);
        }
        return w;
    };
    MixFrag ring[RING];
    if (A.wpk && wave < K) {
        FFNO_UNROLL
        for (int f = 0; f < RING; ++f) ring[f] = load_w(A.wpk + (long)wave * F::MODE_FRAGS * 64 * MNP, f);
    }
    __syncthreads();
    if (tr && tidx == 0) tr[4] = plat::realtime();

    // ---------------- phase 2: per-mode channel mix of all 16 lines, in place ----------------
    if (A.wpk) {
        // MFMA row j = (line (j mod 2 NL) >> 1, part j & 1); with 8 lines rows 16..31 repeat rows 0..15
        const float* arow = XS + ((j & (2 * NL - 1)) >> 1) * LSF + (j & 1) * RS + 8 * half;
        for (int k = wave; k < K; k += F::NW) {
            const u32x4* __restrict__ wk = A.wpk + (long)k * F::MODE_FRAGS * 64 * MNP;
            if (k != wave) {
                FFNO_UNROLL
                for (int f = 0; f < RING; ++f) ring[f] = load_w(wk, f);
            }
            MixFrag a[4];
            FFNO_UNROLL
            for (int st = 0; st < 4; ++st) {
                const float4 v0 = *reinterpret_cast<const float4*>(arow + 2 * k * RS + 16 * st);
                const float4 v1 = *reinterpret_cast<const float4*>(arow + 2 * k * RS + 16 * st + 4);
                if constexpr (MIXH2)
                    a[st] = split2_8(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w);
                else
                    a[st] = split3_8(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w);
            }
            f32x16 p[4];
            FFNO_UNROLL
            for (int pt = 0; pt < 4; ++pt) p[pt] = zero16();
            if constexpr (MIXH2) {
                // fragment order (pt, st): the four k-steps of an output tile run back to back on one correction accumulator
                FFNO_UNROLL
                for (int pt = 0; pt < 4; ++pt) {
                    f32x16 pc = zero16();
                    FFNO_UNROLL
                    for (int st = 0; st < 4; ++st) {
                        const int f = pt * 4 + st;
                        const Hf2 b = ring[f % RING];
                        if (f + RING < F::MODE_FRAGS) ring[f % RING] = load_w(wk, f + RING);
                        mfma_h2(a[st], b, p[pt], pc);
                    }
                    SplitHf2::fold(p[pt], pc);
                }
            } else {
                FFNO_UNROLL
                for (int st = 0; st < 4; ++st) {
                    FFNO_UNROLL
                    for (int pt = 0; pt < 4; ++pt) {
                        const int f = st * 4 + pt;                       // fragment f lives in ring slot f mod RING
                        const Bf3 b = ring[f % RING];
                        if (f + RING < F::MODE_FRAGS) ring[f % RING] = load_w(wk, f + RING);
                        p[pt] = mfma_x3(a[st], b, p[pt]);
                    }
                }
            }
            // D rows 2q, 2q+1 of this lane = (re, im) of line (q & 1) + 4 (q >> 1) + 2 half
            FFNO_UNROLL
            for (int q = 0; q < NL / 2; ++q) {           // (8 lines: accumulator registers 0..7 hold the 16 live rows)
                const int line = (q & 1) + 4 * (q >> 1) + 2 * half;
                float yr[2], yi[2];
                FFNO_UNROLL
                for (int t = 0; t < 2; ++t) {
                    const float p1r = p[t][2 * q], p1i = p[t][2 * q + 1];
                    const float p2r = p[2 + t][2 * q], p2i = p[2 + t][2 * q + 1];
                    if (A.conj_t == 0) {
                        yr[t] = p1r - p2i;
                        yi[t] = p2r + p1i;
                    } else {
                        yr[t] = p1r + p2i;
                        yi[t] = p1i - p2r;
                    }
                }
                float* dst = XS + line * LSF + 2 * k * RS + 2 * j;
                *reinterpret_cast<float2*>(dst) = make_float2(yr[0], yr[1]);
                *reinterpret_cast<float2*>(dst + RS) = make_float2(yi[0], yi[1]);
            }
        }
        __syncthreads();
    }
    if (tr && tidx == 0) tr[5] = plat::realtime();

    if constexpr (MIXOUT) {
        // ---------------- phase 3': the lines' mixed spectra leave as operand fragments ----------------
        // fragment (st, ct) of a line: lane (j, half), slot e <-> Y[kk = 16 st + 8 half + e][c = 2 j + ct] -- the B operand of this
        // kernel's own phase 3 and, read as rows, the A operand Y^T[c][kk] of the second kernel's transposed inverse transform.
        // ONE power of two per line brings its maximum to 2^10 (the line went through the weights: nothing bounds it a priori).
        static_assert(MIXH2, "the fragments are split-fp16");
        FFNO_UNROLL
        for (int ln = 0; ln < NLW; ++ln) {
            if (!(ln ? live1 : live0)) continue;
            const float* xs = XS + (lw + ln) * LSF + 2 * j;
            float2 v[2][8];
            float ym = 0.f;
            FFNO_UNROLL
            for (int st = 0; st < 2; ++st) {
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) {
                    const int kk = 16 * st + 8 * half + e;
                    v[st][e] = make_float2(0.f, 0.f);
                    if (kk < 2 * K) v[st][e] = *reinterpret_cast<const float2*>(xs + kk * RS);
                    ym = fmaxf(ym, fmaxf(fabsf(v[st][e].x), fabsf(v[st][e].y)));
                }
            }
            FFNO_UNROLL
            for (int sh = 32; sh >= 1; sh >>= 1) ym = fmaxf(ym, __shfl_xor(ym, sh));
            const float sy = range_scale(f2u(ym), 0, 10);
            u32x4* dst = A.mix_out + (long)(line0 + ln) * (8 * 64) + lane;
            FFNO_UNROLL
            for (int st = 0; st < 2; ++st) {
                const Hf2 y0 = split2_8(v[st][0].x * sy, v[st][1].x * sy, v[st][2].x * sy, v[st][3].x * sy, v[st][4].x * sy,
                                        v[st][5].x * sy, v[st][6].x * sy, v[st][7].x * sy);
                const Hf2 y1 = split2_8(v[st][0].y * sy, v[st][1].y * sy, v[st][2].y * sy, v[st][3].y * sy, v[st][4].y * sy,
                                        v[st][5].y * sy, v[st][6].y * sy, v[st][7].y * sy);
                dst[((st * 2 + 0) * 2 + 0) * 64] = y0.hi;
                dst[((st * 2 + 0) * 2 + 1) * 64] = y0.lo;
                dst[((st * 2 + 1) * 2 + 0) * 64] = y1.hi;
                dst[((st * 2 + 1) * 2 + 1) * 64] = y1.lo;
            }
            // accumulator of the second kernel (2^11 sy rs Y) -> samples
            if (lane == 0) A.mix_scale[line0 + ln] = (A.self_range ? lrrs[lw + ln] : rrs) * kHf2Unscale / sy;
        }
        return;
    }
    // ---------------- phase 3: zero-padded inverse DFT of the wave's two lines ----------------
    {
        const int RTtot = (L + 31) >> 5;
        const unsigned hoff = (unsigned)(4 * half * es * ST::BYTES);
        FFNO_NOUNROLL
        for (int rt0 = 0; rt0 < RTtot; rt0 += 2) {
            // inverse-DFT matrix fragments of two 32-row output tiles: row n, slot e of k-step st <-> kk = (mode t, part)
            DftFrag G[2][2];
            FFNO_UNROLL
            for (int q = 0; q < 2; ++q) {
                if constexpr (DFTTAB) {
                    // inverse part of the table (ffno_x3_dft.h; <= 16 modes: 4 forward fragments per 64-sample chunk, then 2 per
                    // 32-row tile); a tile past the end (odd tile count) reads the last one -- its rows are never stored
                    FFNO_UNROLL
                    for (int st = 0; st < 2; ++st)
                        G[q][st] = x3k_load_dft(A.dft, ((L + 63) >> 6) * 4 + min(rt0 + q, RTtot - 1) * 2 + st, lane);
                    continue;
                }
                const int n = 32 * (rt0 + q) + j;
                FFNO_UNROLL
                for (int st = 0; st < 2; ++st) {
                    float g[8];
                    const int nm = n < L ? n : 0;
                    int idx = (nm * (8 * st + 4 * half)) % L;          // n t mod L, advanced by n per mode
                    FFNO_UNROLL
                    for (int e = 0; e < 8; ++e) {
                        const int kk = 16 * st + 8 * half + e, t = kk >> 1, part = kk & 1;
                        const float ck = (A.inv_ck && !(t == 0 || 2 * t == L)) ? 2.f : 1.f;
                        g[e] = (kk < 2 * K && n < L) ? (part ? -ck : ck) * tws[(part ? L : 0) + idx] : 0.f;
                        if (part) {
                            idx += nm;
                            if (idx >= L) idx -= L;
                        }
                    }
                    if constexpr (DFTH2)
                        G[q][st] = split2s_8(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
                    else
                        G[q][st] = split3_8(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
                }
            }
            FFNO_UNROLL
            for (int ln = 0; ln < NLW; ++ln) {
                if (!(ln ? live1 : live0)) continue;
                const unsigned lo = (ln ? lo1 : lo0) + hoff;
                // B operands: the line's spectrum, split: slot e of k-step st <-> row kk = 16 st + 8 half + e
                using SpecFrag = typename std::conditional<DFTH2, Hf2, Bf3>::type;
                SpecFrag y[2][2];
                const float* xs = XS + (lw + ln) * LSF + 2 * j;
                float2 v[2][8];
                FFNO_UNROLL
                for (int st = 0; st < 2; ++st) {
                    FFNO_UNROLL
                    for (int e = 0; e < 8; ++e) {
                        const int kk = 16 * st + 8 * half + e;
                        v[st][e] = make_float2(0.f, 0.f);
                        if (kk < 2 * K) v[st][e] = *reinterpret_cast<const float2*>(xs + kk * RS);
                    }
                }
                // fp16x2: the line's mixed spectrum (nobody bounds it: it went through the weights) is brought to 2^10 by the power
                // of two of ITS OWN maximum -- the wave holds the whole line -- and the outputs are divided again (exact)
                // (one scale per COLUMN TILE -- the even / the odd channels: what one wave of the latency kernel sees of a line, so
                // that the two kernels stay bit-identical)
                float osc0 = rrs, osc1 = rrs;                 // accumulator -> output, per column tile
                if constexpr (DFTH2) {
                    float ym0 = 0.f, ym1 = 0.f;
                    FFNO_UNROLL
                    for (int st = 0; st < 2; ++st) {
                        FFNO_UNROLL
                        for (int e = 0; e < 8; ++e) ym0 = fmaxf(ym0, fabsf(v[st][e].x)), ym1 = fmaxf(ym1, fabsf(v[st][e].y));
                    }
                    FFNO_UNROLL
                    for (int sh = 32; sh >= 1; sh >>= 1) ym0 = fmaxf(ym0, __shfl_xor(ym0, sh)), ym1 = fmaxf(ym1, __shfl_xor(ym1, sh));
                    const float sy0 = range_scale(f2u(ym0), 0, 10), sy1 = range_scale(f2u(ym1), 0, 10);
                    osc0 = rrs * kHf2Unscale / sy0, osc1 = rrs * kHf2Unscale / sy1;
                    FFNO_UNROLL
                    for (int st = 0; st < 2; ++st) {
                        FFNO_UNROLL
                        for (int e = 0; e < 8; ++e) v[st][e].x *= sy0, v[st][e].y *= sy1;
                    }
                }
                FFNO_UNROLL
                for (int st = 0; st < 2; ++st) {
                    if constexpr (DFTH2) {
                        y[st][0] = split2_8(v[st][0].x, v[st][1].x, v[st][2].x, v[st][3].x, v[st][4].x, v[st][5].x, v[st][6].x, v[st][7].x);
                        y[st][1] = split2_8(v[st][0].y, v[st][1].y, v[st][2].y, v[st][3].y, v[st][4].y, v[st][5].y, v[st][6].y, v[st][7].y);
                    } else {
                        y[st][0] = split3_8(v[st][0].x, v[st][1].x, v[st][2].x, v[st][3].x, v[st][4].x, v[st][5].x, v[st][6].x, v[st][7].x);
                        y[st][1] = split3_8(v[st][0].y, v[st][1].y, v[st][2].y, v[st][3].y, v[st][4].y, v[st][5].y, v[st][6].y, v[st][7].y);
                    }
                }
                FFNO_UNROLL
                for (int q = 0; q < 2; ++q) {
                    if (rt0 + q >= RTtot) continue;
                    // rows the epilogue adds (residual / accumulate) are requested before the tile's products
                    typename ST::Raw2 pre[16];      // (raw words: widened in the epilogue, after the tile's products)
                    const char* addsrc = A.resid ? reinterpret_cast<const char*>(A.resid)
                                                 : (A.accumulate ? reinterpret_cast<const char*>(A.out) : nullptr);
                    if (addsrc) {
                        FFNO_UNROLL
                        for (int r = 0; r < 16; ++r) {
                            const int nu = min(32 * (rt0 + q) + (r & 3) + 8 * (r >> 2) + 4 * half, L - 1);
                            pre[r] = ST::ldr2(addsrc + (lo - hoff) + (unsigned)nu * (unsigned)(es * ST::BYTES));
                        }
                    }
                    f32x16 o0 = zero16(), o1 = zero16();
                    FFNO_UNROLL
                    for (int st = 0; st < 2; ++st) {
                        if constexpr (DFTH2) {
                            o0 = mfma_h2s(G[q][st], y[st][0], o0);
                            o1 = mfma_h2s(G[q][st], y[st][1], o1);
                        } else {
                            o0 = mfma_x3(G[q][st], y[st][0], o0);
                            o1 = mfma_x3(G[q][st], y[st][1], o1);
                        }
                    }
                    FFNO_UNROLL
                    for (int r = 0; r < 16; ++r) {
                        const int nu = 32 * (rt0 + q) + (r & 3) + 8 * (r >> 2);     // uniform part of the output sample index
                        if (nu + 4 * half < L) {
                            const long uo = (long)nu * es * ST::BYTES;
                            float2 o = make_float2(o0[r] * osc0, o1[r] * osc1);
                            if (addsrc) {
                                const float2 pw = ST::w2(pre[r]);
                                o.x += pw.x, o.y += pw.y;
                            }
                            if (A.accumulate && A.resid) {      // both at once (rare): the second addend is read in place
                                const float2 pv = ST::ld2(reinterpret_cast<const char*>(A.out) + uo + lo);
                                o.x += pv.x, o.y += pv.y;
                            }
                            ST::st2(reinterpret_cast<char*>(A.out) + uo + lo, o);
                            omax = fmaxf(omax, fmaxf(fabsf(ST::rnd(o.x)), fabsf(ST::rnd(o.y))));
                        }
                    }
                }
            }
        }
    }
    if (A.out_amax) range_fold(omax, rfold, F::NW, A.out_amax);
}

// ---- width 32 (the 3-D mesh operators of the reference run width 32: mesh_3d.py / BASELINE config 5) -------------------------
// The same three phases for C = 32 channels.  A wave still owns two lines, but transforms them TOGETHER: the two 32-column MFMA
// tiles that were the even / odd channels of one 64-channel line are now the 32 channels of line 2w and of line 2w + 1 (the DFT
// acts on every column independently, so nothing else changes in phases 1 and 3 -- a lane reads / writes one float of each
// line instead of a float2 of one).  The per-mode channel mix has rows = (line, re/im) of the 16 lines (a full 32-row tile), 32
// input channels (two k16 steps) and 32 output channels (one column tile): four weight fragments per mode.
struct X3Cfg32 {
    static constexpr int C = 32;
    static constexpr int NW = 8;
    static constexpr int NL = 16;              // lines per workgroup (two per wave)
    static constexpr int KK = 32;              // (mode, re/im) rows per line: K <= 16
    static constexpr int RS = 36;              // row stride (floats): +4 shifts the re / im rows of a line by 4 banks
    static constexpr int LSF = KK * RS + 8;    // line stride (floats)
    static constexpr int MODE_FRAGS = 4;       // per mode: 2 k-steps x 2 planes (re, im)
};

// planes[k][p][i][o] (C = 32) -> fragment (k, st, p), lane (j, half), slot e  <-  planes[k][p][i = 16 st + 8 half + e][o = j];
// format 1: two fp16 planes per fragment, fragments of a mode ordered (p, st)
__global__ __launch_bounds__(256) void x3_pack32_kernel(const X3PackDesc* __restrict__ descs) {
    constexpr int C = X3Cfg32::C;
    const X3PackDesc d = descs[blockIdx.y];
    const int nfrag = d.K * X3Cfg32::MODE_FRAGS;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nfrag * 64) return;
    const int frag = t >> 6, lane = t & 63, j = lane & 31, half = lane >> 5;
    const int p = frag & 1, st = (frag >> 1) & 1, k = frag >> 2;
    const float* src = d.planes + ((long)(k * 2 + p) * C + (16 * st + 8 * half)) * C + j;
    float v[8];
    FFNO_UNROLL
    for (int e = 0; e < 8; ++e) v[e] = src[(long)e * C];
    if (d.format == 1) {
        const Hf2 f = split2_8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        const int fo = (k * 2 + p) * 2 + st;
        d.dst[(fo * 2 + 0) * 64 + lane] = f.hi;
        d.dst[(fo * 2 + 1) * 64 + lane] = f.lo;
        return;
    }
    const Bf3 f = split3_8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    d.dst[(frag * 3 + 0) * 64 + lane] = f.hi;
    d.dst[(frag * 3 + 1) * 64 + lane] = f.mid;
    d.dst[(frag * 3 + 2) * 64 + lane] = f.lo;
}

template <bool MIXH2, class ST = StF32, bool TAB = false>
__device__ __forceinline__ void spectral_x3c32_body(const X3Args A, int bidx) {
    static_assert(!ST::BF16 || MIXH2, "bf16 storage runs the split-fp16 path");
    static_assert(!TAB || MIXH2, "the fragment table holds fp16 planes");
    using F = X3Cfg32;
    constexpr int C = F::C, RS = F::RS, LSF = F::LSF, NL = F::NL;
    __shared__ __attribute__((aligned(16))) float XS[NL * LSF];
    __shared__ float rfold[F::NW];
    FFNO_DYN_SMEM(smem);
    float* tws = reinterpret_cast<float*>(smem);

    const float* __restrict__ in = A.in;
    const int R = A.R, L = A.L, K = A.K;
    const LineMap lm = A.lm;
    const float rs = (MIXH2 && A.wpk && A.in_amax) ? range_scale(*A.in_amax, 1 + (ceil_log2_int(L) + 1) / 2, 15) : 1.f;
    const float rrs = 1.f / rs;
    // the DFT phases on split-fp16 with the fp16x2 packs (spectral_x3_body "DFTH2"); here a column tile IS a line
    constexpr bool DFTH2 = MIXH2;
    using DftFrag = typename std::conditional<DFTH2, Hf3, Bf3>::type;
    const float sx = (DFTH2 && A.in_amax) ? range_scale(*A.in_amax, 0, 10) : 1.f;
    const float unx = DFTH2 ? kHf2Unscale / sx : 1.f;
    float omax = 0.f;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int lw = 2 * wave;                       // this wave's two lines inside the tile: lw (column tile 0), lw + 1 (tile 1)
    const int line0 = bidx * NL + lw;
    const bool live0 = line0 < R, live1 = line0 + 1 < R;
    const long es = lm.elem_stride;
    const unsigned esb = (unsigned)(es * ST::BYTES);
    // lines past the end of the axis read (and transform) line R - 1 again; nothing of a dead line is ever stored
    const unsigned lo0 = (unsigned)((lm.base(min(line0, R - 1)) + j) * ST::BYTES);
    const unsigned lo1 = (unsigned)((lm.base(min(line0 + 1, R - 1)) + j) * ST::BYTES);

    // ---------------- phase 1: truncated forward DFT of the wave's two lines, side by side ----------------
    {
        const int kk = j, k = kk >> 1, ri = kk & 1;
        const bool rowok = kk < 2 * K;
        const float ck = (A.fwd_ck && !(k == 0 || 2 * k == L)) ? 2.f : 1.f;
        const float amul = rowok ? (ri ? -ck : ck) : 0.f;
        const int tbase = ri ? L : 0;
        const int km = rowok ? k : 0;
        float2 raw[4][8];      // .x = line lw, .y = line lw + 1
        auto load_rows = [&](int chunk, int u) {
            FFNO_UNROLL
            for (int e = 0; e < 8; ++e) {
                const unsigned no = (unsigned)min(16 * (4 * chunk + u) + 8 * half + e, L - 1) * esb;
                raw[u][e].x = ST::ld1(reinterpret_cast<const typename ST::T*>(reinterpret_cast<const char*>(in) + (lo0 + no)));
                raw[u][e].y = ST::ld1(reinterpret_cast<const typename ST::T*>(reinterpret_cast<const char*>(in) + (lo1 + no)));
            }
        };
        const int nchunks = (L + 63) >> 6;
        // TAB: a 16-sample k-step that lies wholly past the end of the line (the 72-sample lines of the padded 64^3 mesh: three of
        // the second chunk's four) multiplies zeros of the DFT matrix -- its loads, splits and products are skipped (wave-uniform;
        // exact: the products it drops are +0) -- and nothing reads the twiddle table, so it is not staged
        auto kstep = [&](int chunk, int u) { return !TAB || 16 * (4 * chunk + u) < L; };
        FFNO_UNROLL
        for (int u = 0; u < 4; ++u)
            if (kstep(0, u)) load_rows(0, u);
        if constexpr (!TAB) {
            for (int i = threadIdx.x; i < 2 * L; i += blockDim.x) tws[i] = A.tw[i];
            __syncthreads();
        }
        const int k8 = (km * 8) % L;
        f32x16 acc0 = zero16(), acc1 = zero16();
        FFNO_NOUNROLL
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            DftFrag Ff[4];
            int idx = (km * (64 * chunk + 8 * half)) % L;
            if constexpr (TAB) {      // the chunk's fragments from the table (the table layout of <= 16 modes: one row tile)
                FFNO_UNROLL
                for (int u = 0; u < 4; ++u)
                    if (kstep(chunk, u)) Ff[u] = x3k_load_dft(A.dft, chunk * 4 + u, lane);
            } else
            FFNO_UNROLL
            for (int u = 0; u < 4; ++u) {
                float f[8];
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) {
                    const int n = 16 * (4 * chunk + u) + 8 * half + e;
                    f[e] = n < L ? amul * tws[tbase + idx] : 0.f;
                    idx += km;
                    if (idx >= L) idx -= L;
                }
                idx += k8;
                if (idx >= L) idx -= L;
                if constexpr (DFTH2)
                    Ff[u] = split2s_8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
                else
                    Ff[u] = split3_8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
            }
            FFNO_UNROLL
            for (int u = 0; u < 4; ++u) {
                if (!kstep(chunk, u)) continue;      // (then the same k-step of every later chunk is past the end too)
                if constexpr (DFTH2) {
                    float2 w[8];
                    FFNO_UNROLL
                    for (int e = 0; e < 8; ++e) w[e] = make_float2(raw[u][e].x * sx, raw[u][e].y * sx);
                    const Hf2 b0 = split2_8(w[0].x, w[1].x, w[2].x, w[3].x, w[4].x, w[5].x, w[6].x, w[7].x);
                    const Hf2 b1 = split2_8(w[0].y, w[1].y, w[2].y, w[3].y, w[4].y, w[5].y, w[6].y, w[7].y);
                    if (chunk + 1 < nchunks && kstep(chunk + 1, u)) load_rows(chunk + 1, u);
                    acc0 = mfma_h2s(Ff[u], b0, acc0);
                    acc1 = mfma_h2s(Ff[u], b1, acc1);
                } else {
                    const Bf3 b0 = split3_8(raw[u][0].x, raw[u][1].x, raw[u][2].x, raw[u][3].x, raw[u][4].x, raw[u][5].x,
                                            raw[u][6].x, raw[u][7].x);
                    const Bf3 b1 = split3_8(raw[u][0].y, raw[u][1].y, raw[u][2].y, raw[u][3].y, raw[u][4].y, raw[u][5].y,
                                            raw[u][6].y, raw[u][7].y);
                    if (chunk + 1 < nchunks) load_rows(chunk + 1, u);
                    acc0 = mfma_x3(Ff[u], b0, acc0);
                    acc1 = mfma_x3(Ff[u], b1, acc1);
                }
            }
        }
        float* xs0 = XS + lw * LSF + j;
        float* xs1 = xs0 + LSF;
        FFNO_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int row = drow(r, half);
            if (row < 2 * K) {
                const float v0 = acc0[r] * unx, v1 = acc1[r] * unx;
                xs0[row * RS] = v0 * rs;
                xs1[row * RS] = v1 * rs;
                if (A.spec_save) {
                    float* sp = A.spec_save + (((long)(row >> 1) * R + line0) * 2 + (row & 1)) * C + j;
                    if (live0) sp[0] = v0;
                    if (live1) sp[2 * C] = v1;
                }
            }
        }
    }
    // weight fragments of phase 2 through a ring (the fragments of all this wave's modes are one stream)
    constexpr int RING = 4;
    using MixFrag = typename std::conditional<MIXH2, Hf2, Bf3>::type;
    constexpr int MNP = MIXH2 ? 2 : 3;
    auto load_w = [&](const u32x4* __restrict__ wk, int f) {
        MixFrag w;
        if constexpr (MIXH2) {
            w.hi = wk[(f * 2 + 0) * 64 + lane];
            w.lo = wk[(f * 2 + 1) * 64 + lane];
        } else {
            w = x3_load_frag(wk, f, lane);
        }
        return w;
    };
    MixFrag ring[RING];
    if (A.wpk && wave < K) {
        FFNO_UNROLL
        for (int f = 0; f < RING; ++f) ring[f] = load_w(A.wpk + (long)wave * F::MODE_FRAGS * 64 * MNP, f);
    }
    __syncthreads();

    // ---------------- phase 2: per-mode channel mix of the 16 lines, in place ----------------
    if (A.wpk) {
        // MFMA row j = (line j >> 1, part j & 1)
        const float* arow = XS + (j >> 1) * LSF + (j & 1) * RS + 8 * half;
        for (int k = wave; k < K; k += F::NW) {
            const bool more = k + F::NW < K;
            const u32x4* __restrict__ wn = A.wpk + (long)(more ? k + F::NW : k) * F::MODE_FRAGS * 64 * MNP;
            MixFrag a[2];
            FFNO_UNROLL
            for (int st = 0; st < 2; ++st) {
                const float4 v0 = *reinterpret_cast<const float4*>(arow + 2 * k * RS + 16 * st);
                const float4 v1 = *reinterpret_cast<const float4*>(arow + 2 * k * RS + 16 * st + 4);
                if constexpr (MIXH2)
                    a[st] = split2_8(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w);
                else
                    a[st] = split3_8(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w);
            }
            f32x16 p[2];
            p[0] = zero16(), p[1] = zero16();
            if constexpr (MIXH2) {      // fragment order (p, st)
                FFNO_UNROLL
                for (int pp = 0; pp < 2; ++pp) {
                    f32x16 pc = zero16();
                    FFNO_UNROLL
                    for (int st = 0; st < 2; ++st) {
                        const int f = pp * 2 + st;
                        const Hf2 b = ring[f];
                        if (more) ring[f] = load_w(wn, f);
                        mfma_h2(a[st], b, p[pp], pc);
                    }
                    SplitHf2::fold(p[pp], pc);
                }
            } else {                    // fragment order (st, p)
                FFNO_UNROLL
                for (int st = 0; st < 2; ++st) {
                    FFNO_UNROLL
                    for (int pp = 0; pp < 2; ++pp) {
                        const int f = st * 2 + pp;
                        const Bf3 b = ring[f];
                        if (more) ring[f] = load_w(wn, f);
                        p[pp] = mfma_x3(a[st], b, p[pp]);
                    }
                }
            }
            // D rows 2q, 2q+1 of this lane = (re, im) of line (q & 1) + 4 (q >> 1) + 2 half;  P1 = X Wr (p[0]), P2 = X Wi (p[1])
            FFNO_UNROLL
            for (int q = 0; q < 8; ++q) {
                const int ln = (q & 1) + 4 * (q >> 1) + 2 * half;
                const float p1r = p[0][2 * q], p1i = p[0][2 * q + 1];
                const float p2r = p[1][2 * q], p2i = p[1][2 * q + 1];
                float yr, yi;
                if (A.conj_t == 0) {
                    yr = p1r - p2i;
                    yi = p2r + p1i;
                } else {
                    yr = p1r + p2i;
                    yi = p1i - p2r;
                }
                float* dst = XS + ln * LSF + 2 * k * RS + j;
                dst[0] = yr;
                dst[RS] = yi;
            }
        }
        __syncthreads();
    }

    // ---------------- phase 3: zero-padded inverse DFT of the wave's two lines, side by side ----------------
    {
        const int RTtot = (L + 31) >> 5;
        const unsigned hoff = (unsigned)(4 * half * es * ST::BYTES);
        const float* xs0 = XS + lw * LSF + j;
        const float* xs1 = xs0 + LSF;
        // B operands: the two lines' spectra (tile 0 = line lw, tile 1 = line lw + 1): slot e of k-step st <-> row 16 st + 8 half + e
        using SpecFrag = typename std::conditional<DFTH2, Hf2, Bf3>::type;
        SpecFrag y[2][2];
        float v0[2][8], v1[2][8];
        FFNO_UNROLL
        for (int st = 0; st < 2; ++st) {
            FFNO_UNROLL
            for (int e = 0; e < 8; ++e) {
                const int kk = 16 * st + 8 * half + e;
                v0[st][e] = v1[st][e] = 0.f;
                if (kk < 2 * K) v0[st][e] = xs0[kk * RS], v1[st][e] = xs1[kk * RS];
            }
        }
        float osc0 = rrs, osc1 = rrs;
        if constexpr (DFTH2) {      // each line's mixed spectrum scaled from its own maximum (the wave holds both lines whole)
            float ym0 = 0.f, ym1 = 0.f;
            FFNO_UNROLL
            for (int st = 0; st < 2; ++st) {
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) ym0 = fmaxf(ym0, fabsf(v0[st][e])), ym1 = fmaxf(ym1, fabsf(v1[st][e]));
            }
            FFNO_UNROLL
            for (int sh = 32; sh >= 1; sh >>= 1) ym0 = fmaxf(ym0, __shfl_xor(ym0, sh)), ym1 = fmaxf(ym1, __shfl_xor(ym1, sh));
            const float sy0 = range_scale(f2u(ym0), 0, 10), sy1 = range_scale(f2u(ym1), 0, 10);
            osc0 = rrs * kHf2Unscale / sy0, osc1 = rrs * kHf2Unscale / sy1;
            FFNO_UNROLL
            for (int st = 0; st < 2; ++st) {
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) v0[st][e] *= sy0, v1[st][e] *= sy1;
            }
        }
        FFNO_UNROLL
        for (int st = 0; st < 2; ++st) {
            if constexpr (DFTH2) {
                y[st][0] = split2_8(v0[st][0], v0[st][1], v0[st][2], v0[st][3], v0[st][4], v0[st][5], v0[st][6], v0[st][7]);
                y[st][1] = split2_8(v1[st][0], v1[st][1], v1[st][2], v1[st][3], v1[st][4], v1[st][5], v1[st][6], v1[st][7]);
            } else {
                y[st][0] = split3_8(v0[st][0], v0[st][1], v0[st][2], v0[st][3], v0[st][4], v0[st][5], v0[st][6], v0[st][7]);
                y[st][1] = split3_8(v1[st][0], v1[st][1], v1[st][2], v1[st][3], v1[st][4], v1[st][5], v1[st][6], v1[st][7]);
            }
        }
        const char* addsrc = A.resid ? reinterpret_cast<const char*>(A.resid)
                                     : (A.accumulate ? reinterpret_cast<const char*>(A.out) : nullptr);
        FFNO_NOUNROLL
        for (int rt = 0; rt < RTtot; ++rt) {
            const int n = 32 * rt + j;
            DftFrag G[2];
            if constexpr (TAB) {
                FFNO_UNROLL
                for (int st = 0; st < 2; ++st) G[st] = x3k_load_dft(A.dft, ((L + 63) >> 6) * 4 + rt * 2 + st, lane);
            } else
            FFNO_UNROLL
            for (int st = 0; st < 2; ++st) {
                float g[8];
                const int nm = n < L ? n : 0;
                int idx = (nm * (8 * st + 4 * half)) % L;
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) {
                    const int kk = 16 * st + 8 * half + e, t = kk >> 1, part = kk & 1;
                    const float ck = (A.inv_ck && !(t == 0 || 2 * t == L)) ? 2.f : 1.f;
                    g[e] = (kk < 2 * K && n < L) ? (part ? -ck : ck) * tws[(part ? L : 0) + idx] : 0.f;
                    if (part) {
                        idx += nm;
                        if (idx >= L) idx -= L;
                    }
                }
                if constexpr (DFTH2)
                    G[st] = split2s_8(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
                else
                    G[st] = split3_8(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
            }
            float2 pre[16];
            if (addsrc) {
                FFNO_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const unsigned no = (unsigned)min(32 * rt + (r & 3) + 8 * (r >> 2) + 4 * half, L - 1) * esb;
                    pre[r].x = ST::ld1(reinterpret_cast<const typename ST::T*>(addsrc + (lo0 + no)));
                    pre[r].y = ST::ld1(reinterpret_cast<const typename ST::T*>(addsrc + (lo1 + no)));
                }
            }
            f32x16 o0 = zero16(), o1 = zero16();
            FFNO_UNROLL
            for (int st = 0; st < 2; ++st) {
                if constexpr (DFTH2) {
                    o0 = mfma_h2s(G[st], y[st][0], o0);
                    o1 = mfma_h2s(G[st], y[st][1], o1);
                } else {
                    o0 = mfma_x3(G[st], y[st][0], o0);
                    o1 = mfma_x3(G[st], y[st][1], o1);
                }
            }
            FFNO_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int nu = 32 * rt + (r & 3) + 8 * (r >> 2);
                if (nu + 4 * half < L) {
                    const unsigned uo = (unsigned)nu * esb + hoff;
                    float2 o = make_float2(o0[r] * osc0, o1[r] * osc1);
                    if (addsrc) o.x += pre[r].x, o.y += pre[r].y;
                    char* ob = reinterpret_cast<char*>(A.out);
                    if (A.accumulate && A.resid) {
                        o.x += ST::ld1(reinterpret_cast<const typename ST::T*>(ob + (lo0 + uo)));
                        o.y += ST::ld1(reinterpret_cast<const typename ST::T*>(ob + (lo1 + uo)));
                    }
                    if (live0) {
                        ST::st1(ob + (lo0 + uo), o.x);
                        omax = fmaxf(omax, fabsf(ST::rnd(o.x)));
                    }
                    if (live1) {
                        ST::st1(ob + (lo1 + uo), o.y);
                        omax = fmaxf(omax, fabsf(ST::rnd(o.y)));
                    }
                }
            }
        }
    }
    if (A.out_amax) range_fold(omax, rfold, F::NW, A.out_amax);
}

template <int NL, bool MIXH2, class ST = StF32>
__global__ __launch_bounds__(512) void spectral_x3_kernel(X3Args a) {
    spectral_x3_body<NL, MIXH2, ST>(a, blockIdx.x, 0);
}

// Two branches (the two axes of a layer) in ONE launch of n0 + n1 workgroups, one per CU at batch 32.  interleave: even
// workgroups run branch a, odd ones branch b -- workgroup w lands on XCD w % 8, so every XCD's L2 then holds the packed
// weights of ONE branch only; otherwise [0, n0) run a and the rest b.
// DFTTAB (both branches carry a DFT-fragment table): see spectral_x3_body
template <int NL, bool MIXH2, class ST = StF32, bool MIXOUT = false, bool DFTTAB = false>
__global__ __launch_bounds__(512) void spectral_x3_pair_kernel(X3Args a, X3Args b, int n0, int interleave, int skew) {
    const int w = blockIdx.x;
    bool second;
    int idx;
    if ((interleave & 3) == 2) {
        // image-local map (square images, batch a multiple of 8, whole tiles per image; T = interleave >> 8 tiles per image and
        // branch): workgroup w lands on XCD w % 8, so the 2 T workgroups that read one image -- T row tiles of one branch, T
        // column tiles of the other -- are given the same XCD and consecutive slots: the image crosses HBM once and the second
        // branch finds it in that XCD's L2.
        const int T = interleave >> 8, G = 2 * T, per_xcd = n0 / (8 * T);
        const int xcd = w & 7, slot = w >> 3;
        const int image = xcd * per_xcd + slot / G, t = slot % G;
        second = t >= T;
        idx = image * T + (second ? t - T : t);
    } else {
        second = interleave ? (w & 1) : (w >= n0);
        idx = interleave ? (w >> 1) : (second ? w - n0 : w);
    }
    // skew > 0: every other workgroup of each branch starts `skew` cycles late, so that its HBM-bound phases (line loads,
    // output stores) fall on the L2 / matrix phase of its neighbours instead of all 256 CUs hitting HBM in lockstep
    X3Args s;
    s.in = second ? b.in : a.in;
    s.out = second ? b.out : a.out;
    s.resid = second ? b.resid : a.resid;
    s.spec_save = second ? b.spec_save : a.spec_save;
    s.wpk = second ? b.wpk : a.wpk;
    s.tw = second ? b.tw : a.tw;
    s.R = second ? b.R : a.R;
    s.L = second ? b.L : a.L;
    s.K = second ? b.K : a.K;
    s.lm.lines_per_group = second ? b.lm.lines_per_group : a.lm.lines_per_group;
    s.lm.group_stride = second ? b.lm.group_stride : a.lm.group_stride;
    s.lm.line_stride = second ? b.lm.line_stride : a.lm.line_stride;
    s.lm.elem_stride = second ? b.lm.elem_stride : a.lm.elem_stride;
    s.fwd_ck = a.fwd_ck, s.inv_ck = a.inv_ck, s.conj_t = a.conj_t;      // common to both branches
    s.accumulate = second ? b.accumulate : a.accumulate;
    s.in_amax = second ? b.in_amax : a.in_amax;
    s.out_amax = second ? b.out_amax : a.out_amax;
    s.dft = DFTTAB ? (second ? b.dft : a.dft) : nullptr;
    s.mix_out = second ? b.mix_out : a.mix_out;
    s.mix_scale = second ? b.mix_scale : a.mix_scale;
    s.self_range = a.self_range;      // (common to both branches: ffno_spectral_x3_mix_pair checks it)
    spectral_x3_body<NL, MIXH2, ST, MIXOUT, false, DFTTAB>(s, idx, (idx & 1) ? skew : 0);
}

// ---- the three STAGE kernels on the same arithmetic (shapes outside the fused tile: 17..32 modes, e.g. 256 x 256 grids) ----
// Same operators and spectrum layout (spec[k][line][re|im][c]) as dft_fwd / mode_mix / dft_inv of spectral.hip; every
// product is the exact three-way bf16 split.  One wave per work item, four waves per workgroup, no LDS tile:
//   stage A  item = (line, 32-row tile of the (mode, part) rows)          -> spectrum in HBM
//   stage B  item = (mode, 16-line tile): rows = (line, part), weights streamed from the packed sets through a ring
//   stage C  item = (line, pair of 32-sample output tiles)
struct X3Stage {
    const float* in;
    float* out;
    const float* resid;
    const u32x4* wpk;
    const float* tw;
    int R, L, K;
    LineMap lm;
    int accumulate;
    unsigned* out_amax;   // optional range word of `out` (stage C only)
};

__device__ __forceinline__ void x3_dft_fwd_body(const X3Stage S, int scale_ck, int bidx, int nblk) {
    constexpr int C = X3Cfg::C;
    FFNO_DYN_SMEM(smem);
    float* tws = reinterpret_cast<float*>(smem);
    const int R = S.R, L = S.L, K = S.K;
    for (int i = threadIdx.x; i < 2 * L; i += blockDim.x) tws[i] = S.tw[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int RT = (2 * K + 31) >> 5;
    const long es = S.lm.elem_stride;
    const unsigned esb = (unsigned)(es * 4);
    const int nchunks = (L + 63) >> 6;
    const int nitems = R * RT;
    for (int item = bidx * 4 + wave; item < nitems; item += nblk * 4) {
        const int line = item / RT, rt = item - line * RT;
        const int kk = 32 * rt + j, k = kk >> 1, ri = kk & 1;
        const bool rowok = kk < 2 * K;
        const float ck = (scale_ck && !(k == 0 || 2 * k == L)) ? 2.f : 1.f;
        const float amul = rowok ? (ri ? -ck : ck) : 0.f;
        const int tbase = ri ? L : 0;
        const int km = rowok ? k : 0;
        const int k8 = (km * 8) % L;
        const unsigned lo = (unsigned)((S.lm.base(line) + 2 * j) * 4);
        float2 raw[4][8];
        auto load_rows = [&](int chunk, int u) {
            FFNO_UNROLL
            for (int e = 0; e < 8; ++e) {
                const int n = min(16 * (4 * chunk + u) + 8 * half + e, L - 1);
                raw[u][e] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(S.in) + (lo + (unsigned)n * esb));
            }
        };
        FFNO_UNROLL
        for (int u = 0; u < 4; ++u) load_rows(0, u);
        f32x16 acc0 = zero16(), acc1 = zero16();
        FFNO_NOUNROLL
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            Bf3 Ff[4];
            int idx = (km * (64 * chunk + 8 * half)) % L;
            FFNO_UNROLL
            for (int u = 0; u < 4; ++u) {
                float f[8];
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) {
                    const int n = 16 * (4 * chunk + u) + 8 * half + e;
                    f[e] = n < L ? amul * tws[tbase + idx] : 0.f;
                    idx += km;
                    if (idx >= L) idx -= L;
                }
                idx += k8;
                if (idx >= L) idx -= L;
                Ff[u] = split3_8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
            }
            FFNO_UNROLL
            for (int u = 0; u < 4; ++u) {
                const Bf3 b0 = split3_8(raw[u][0].x, raw[u][1].x, raw[u][2].x, raw[u][3].x, raw[u][4].x, raw[u][5].x,
                                        raw[u][6].x, raw[u][7].x);
                const Bf3 b1 = split3_8(raw[u][0].y, raw[u][1].y, raw[u][2].y, raw[u][3].y, raw[u][4].y, raw[u][5].y,
                                        raw[u][6].y, raw[u][7].y);
                if (chunk + 1 < nchunks) load_rows(chunk + 1, u);
                acc0 = mfma_x3(Ff[u], b0, acc0);
                acc1 = mfma_x3(Ff[u], b1, acc1);
            }
        }
        FFNO_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int kr = 32 * rt + drow(r, half);
            if (kr < 2 * K)
                *reinterpret_cast<float2*>(S.out + (((long)(kr >> 1) * R + line) * 2 + (kr & 1)) * C + 2 * j) =
                    make_float2(acc0[r], acc1[r]);
        }
    }
}

__device__ __forceinline__ void x3_mode_mix_body(const X3Stage S, int conj_t, int k, int bx, int nbx) {
    using F = X3Cfg;
    constexpr int C = F::C;
    const int R = S.R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const u32x4* __restrict__ wk = S.wpk + (long)k * F::MODE_FRAGS * F::FRAG;
    const float* xin = S.in + (long)k * R * 2 * C;
    float* yout = S.out + (long)k * R * 2 * C;
    const int ntiles = (R + 15) >> 4;
    for (int tile = bx * 4 + wave; tile < ntiles; tile += nbx * 4) {
        Bf3 ring[4];
        FFNO_UNROLL
        for (int f = 0; f < 4; ++f) ring[f] = x3_load_frag(wk, f, lane);
        const int line = min(16 * tile + (j >> 1), R - 1);           // MFMA row j = (line, part j & 1); dead rows re-read R - 1
        const float* arow = xin + ((long)line * 2 + (j & 1)) * C + 8 * half;
        Bf3 a[4];
        FFNO_UNROLL
        for (int st = 0; st < 4; ++st) {
            const float4 v0 = *reinterpret_cast<const float4*>(arow + 16 * st);
            const float4 v1 = *reinterpret_cast<const float4*>(arow + 16 * st + 4);
            a[st] = split3_8(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w);
        }
        f32x16 p[4];
        FFNO_UNROLL
        for (int pt = 0; pt < 4; ++pt) p[pt] = zero16();
        FFNO_UNROLL
        for (int st = 0; st < 4; ++st) {
            FFNO_UNROLL
            for (int pt = 0; pt < 4; ++pt) {
                const Bf3 b = ring[pt];
                if (st < 3) ring[pt] = x3_load_frag(wk, (st + 1) * 4 + pt, lane);
                p[pt] = mfma_x3(a[st], b, p[pt]);
            }
        }
        FFNO_UNROLL
        for (int q = 0; q < 8; ++q) {
            const int ol = 16 * tile + (q & 1) + 4 * (q >> 1) + 2 * half;
            float yr[2], yi[2];
            FFNO_UNROLL
            for (int t = 0; t < 2; ++t) {
                const float p1r = p[t][2 * q], p1i = p[t][2 * q + 1];
                const float p2r = p[2 + t][2 * q], p2i = p[2 + t][2 * q + 1];
                if (conj_t == 0) {
                    yr[t] = p1r - p2i;
                    yi[t] = p2r + p1i;
                } else {
                    yr[t] = p1r + p2i;
                    yi[t] = p1i - p2r;
                }
            }
            if (ol < R) {
                float* dst = yout + (long)ol * 2 * C + 2 * j;
                *reinterpret_cast<float2*>(dst) = make_float2(yr[0], yr[1]);
                *reinterpret_cast<float2*>(dst + C) = make_float2(yi[0], yi[1]);
            }
        }
    }
}

__device__ __forceinline__ void x3_dft_inv_body(const X3Stage S, int apply_ck, int bidx, int nblk) {
    constexpr int C = X3Cfg::C;
    constexpr int NST = 4;                    // k-steps over the (mode, part) rows: K <= 32
    FFNO_DYN_SMEM(smem);
    float* tws = reinterpret_cast<float*>(smem);
    const int R = S.R, L = S.L, K = S.K;
    for (int i = threadIdx.x; i < 2 * L; i += blockDim.x) tws[i] = S.tw[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const long es = S.lm.elem_stride;
    const int RTtot = (L + 31) >> 5, NP = (RTtot + 1) >> 1;
    const unsigned hoff = (unsigned)(4 * half * es * 4);
    const int nitems = R * NP;
    float omax = 0.f;
    __shared__ float rfold[4];
    for (int item = bidx * 4 + wave; item < nitems; item += nblk * 4) {
        const int line = item / NP, rt0 = 2 * (item - line * NP);
        const unsigned lo = (unsigned)((S.lm.base(line) + 2 * j) * 4) + hoff;
        // k-step outer, output tile inner: the spectrum rows of a k-step are split once and serve both 32-sample tiles; only
        // the four accumulators stay live across k-steps (<= 128 VGPRs: four waves per SIMD hide the epilogue's reads)
        f32x16 o[2][2];
        FFNO_UNROLL
        for (int q = 0; q < 2; ++q) o[q][0] = zero16(), o[q][1] = zero16();
        FFNO_UNROLL
        for (int st = 0; st < NST; ++st) {
            if (16 * st >= 2 * K) continue;
            float2 v[8];
            FFNO_UNROLL
            for (int e = 0; e < 8; ++e) {
                const int kk = 16 * st + 8 * half + e;
                v[e] = make_float2(0.f, 0.f);
                if (kk < 2 * K) v[e] = *reinterpret_cast<const float2*>(S.in + (((long)(kk >> 1) * R + line) * 2 + (kk & 1)) * C + 2 * j);
            }
            const Bf3 y0 = split3_8(v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x);
            const Bf3 y1 = split3_8(v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y);
            FFNO_UNROLL
            for (int q = 0; q < 2; ++q) {
                const int n = 32 * (rt0 + q) + j;
                const int nm = n < L ? n : 0;
                float g[8];
                int idx = (nm * (8 * st + 4 * half)) % L;
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) {
                    const int kk = 16 * st + 8 * half + e, t = kk >> 1, part = kk & 1;
                    const float ck = (apply_ck && !(t == 0 || 2 * t == L)) ? 2.f : 1.f;
                    g[e] = (kk < 2 * K && n < L) ? (part ? -ck : ck) * tws[(part ? L : 0) + idx] : 0.f;
                    if (part) {
                        idx += nm;
                        if (idx >= L) idx -= L;
                    }
                }
                const Bf3 G = split3_8(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
                o[q][0] = mfma_x3(G, y0, o[q][0]);
                o[q][1] = mfma_x3(G, y1, o[q][1]);
            }
        }
        const char* addsrc = S.resid ? reinterpret_cast<const char*>(S.resid)
                                     : (S.accumulate ? reinterpret_cast<const char*>(S.out) : nullptr);
        FFNO_UNROLL
        for (int q = 0; q < 2; ++q) {
            const int rt = rt0 + q;
            if (rt >= RTtot) continue;
            FFNO_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int nu = 32 * rt + (r & 3) + 8 * (r >> 2);
                if (nu + 4 * half < L) {
                    const long uo = (long)nu * es * 4;
                    float2 ov = make_float2(o[q][0][r], o[q][1][r]);
                    if (addsrc) {
                        const float2 pv = *reinterpret_cast<const float2*>(addsrc + uo + lo);
                        ov.x += pv.x, ov.y += pv.y;
                    }
                    if (S.accumulate && S.resid) {
                        const float2 pv = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(S.out) + uo + lo);
                        ov.x += pv.x, ov.y += pv.y;
                    }
                    *reinterpret_cast<float2*>(reinterpret_cast<char*>(S.out) + uo + lo) = ov;
                    omax = fmaxf(omax, fmaxf(fabsf(ov.x), fabsf(ov.y)));
                }
            }
        }
    }
    if (S.out_amax) range_fold(omax, rfold, 4, S.out_amax);
}

// one of two kernel-argument structs, selected field by field (a reference to `second ? b : a` would force an addressable
// copy in scratch memory and turn every field access into a scratch load)
__device__ __forceinline__ X3Stage x3_pick(const X3Stage& a, const X3Stage& b, bool second) {
    X3Stage s;
    s.in = second ? b.in : a.in;
    s.out = second ? b.out : a.out;
    s.resid = second ? b.resid : a.resid;
    s.wpk = second ? b.wpk : a.wpk;
    s.tw = second ? b.tw : a.tw;
    s.R = second ? b.R : a.R;
    s.L = second ? b.L : a.L;
    s.K = second ? b.K : a.K;
    s.lm.lines_per_group = second ? b.lm.lines_per_group : a.lm.lines_per_group;
    s.lm.group_stride = second ? b.lm.group_stride : a.lm.group_stride;
    s.lm.line_stride = second ? b.lm.line_stride : a.lm.line_stride;
    s.lm.elem_stride = second ? b.lm.elem_stride : a.lm.elem_stride;
    s.accumulate = second ? b.accumulate : a.accumulate;
    s.out_amax = second ? b.out_amax : a.out_amax;
    return s;
}

__global__ __launch_bounds__(256) void x3_dft_fwd_pair_kernel(X3Stage a, X3Stage b, int n0, int scale_ck) {
    const bool second = (int)blockIdx.x >= n0;
    x3_dft_fwd_body(x3_pick(a, b, second), scale_ck, second ? blockIdx.x - n0 : blockIdx.x, second ? gridDim.x - n0 : n0);
}
// grid.y = Ka + Kb modes: the first Ka rows of workgroups mix branch a, the rest branch b
__global__ __launch_bounds__(256) void x3_mode_mix_pair_kernel(X3Stage a, X3Stage b, int conj_t) {
    const bool second = (int)blockIdx.y >= a.K;
    x3_mode_mix_body(x3_pick(a, b, second), conj_t, second ? blockIdx.y - a.K : blockIdx.y, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void x3_dft_inv_pair_kernel(X3Stage a, X3Stage b, int n0, int apply_ck) {
    const bool second = (int)blockIdx.x >= n0;
    x3_dft_inv_body(x3_pick(a, b, second), apply_ck, second ? blockIdx.x - n0 : blockIdx.x, second ? gridDim.x - n0 : n0);
}

// ---- the fused branch for 17..64 modes (256 x 256 grids: torus_kochkov runs 32 and 64 modes) ----------------------------------
// Same operator as spectral_x3_body with the spectrum tile in LDS, re-tiled for many modes per line: KKT = 64 or 128 (mode,
// re/im) rows per line, FOUR lines per workgroup, two waves per line.  Why four: at 256 x 256 and batch 2 an axis has 512
// lines of 64 KiB -- an 8-line tile would leave half of the CUs idle, and a CU streams HBM at ~10 B/clk whatever it does (DESIGN
// section 4), so the bytes must be spread over all 256 CUs: 4 lines = 256 KiB in and 256 KiB out per CU, exactly the 64 x 64 /
// batch-32 case.  The three stage kernels above moved the spectra through HBM (69.5 us per paired launch, three launches).
//   phase 1  wave (line, sub): row tiles sub, sub + 2, .. of the truncated DFT of its line (the line is read once from HBM; the
//            partner wave's copy and later row tiles come from L2); writes the LDS tile (and the saved spectrum)
//   phase 2  wave w: modes w, w + 8, ..: per-mode channel mix, rows = (line, re/im) of the four lines (8 live rows of the 32-row
//            MFMA tile -- the mix is 5 % of the launch's MFMA budget here, the weight stream from L2 is what it costs)
//   phase 3  wave (line, sub): pairs of 32-sample output tiles sub, sub + 2, .. of the zero-padded inverse DFT, k-step outer /
//            tile inner as in x3_dft_inv_body, with the accumulate / residual epilogue.
// ST = storage format of in / out / resid (ffno_device.h); the bf16 twin exists for the fp16x2 packs (the split-fp16 DFT path)
// TAB: the DFT-matrix fragments come from the precomputed table A.dft (x3k_dft_frags_kernel) instead of the twiddle table
// M16: the per-mode mix on v_mfma_f32_16x16x32_f16 with format-2 packs (the tile has 8 live rows: half a 16-row tile instead of a
// quarter of a 32-row one -- half the matrix time and a third of the vector work of the mix)
template <int KKT, bool MIXH2, class ST = StF32, bool TAB = false, bool M16 = false>
__device__ __forceinline__ void spectral_x3k_body(const X3Args A, int bidx) {
    static_assert(!TAB || MIXH2, "the fragment table holds fp16 planes");
    static_assert(!M16 || MIXH2, "the 16-row mix runs on fp16x2 packs");
    static_assert(!ST::BF16 || MIXH2, "bf16 storage runs the split-fp16 path");
    using F = X3Cfg;
    constexpr int C = F::C, RS = F::RS, NL = 4, LSF = KKT * RS + 8, RT = KKT / 32, NST = KKT / 16;
    __shared__ __attribute__((aligned(16))) float XS[NL * LSF];
    __shared__ float rfold[F::NW];
    FFNO_DYN_SMEM(smem);
    float* tws = reinterpret_cast<float*>(smem);

    const int R = A.R, L = A.L, K = A.K;
    const LineMap lm = A.lm;
    const float rs = (MIXH2 && A.wpk && A.in_amax) ? range_scale(*A.in_amax, 1 + (ceil_log2_int(L) + 1) / 2, 15) : 1.f;
    const float rrs = 1.f / rs;
    // with the fp16x2 packs the DFT phases run on split-fp16 too (spectral_x3_body "DFTH2": DFT matrices as the bounded operand of
    // single-accumulator products, samples scaled from the range word, each line's mixed spectrum from its own maximum)
    constexpr bool DFTH2 = MIXH2;
    const float sx = (DFTH2 && A.in_amax) ? range_scale(*A.in_amax, 0, 10) : 1.f;
    const float unx = DFTH2 ? kHf2Unscale / sx : 1.f;
    float omax = 0.f;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int lw = wave >> 1, sub = wave & 1;        // this wave's line inside the tile / which half of the line's items
    const int line = bidx * NL + lw;
    const bool live = line < R;
    const long es = lm.elem_stride;
    const unsigned esb = (unsigned)(es * ST::BYTES);
    // lines past the end of the axis read (and transform) line R - 1 again; nothing of a dead line is ever stored
    const unsigned lo = (unsigned)((lm.base(min(line, R - 1)) + 2 * j) * ST::BYTES);
    const int nchunks = (L + 63) >> 6;

    if constexpr (!TAB) {      // (the table kernels never read the twiddles: no staging pass, no barrier in front of the first line load)
        for (int i = threadIdx.x; i < 2 * L; i += blockDim.x) tws[i] = A.tw[i];
        __syncthreads();
    }

    // ---------------- phase 1 ----------------
    if constexpr (TAB) {
        // With the fragments coming from the table a wave can afford ALL row tiles of its line, so the two waves of a line split
        // its CHANNELS instead of its row tiles: wave (line, sub) requests, splits and transforms channels [32 sub, 32 sub + 32)
        // only -- the line crosses the load path once (not once per wave), and each wave splits half the samples.
        const unsigned loc = (unsigned)((lm.base(min(line, R - 1)) + 32 * sub + j) * ST::BYTES);
        typename ST::Raw1 raw[4][8];
        auto load_rows = [&](int chunk, int u) {
            FFNO_UNROLL
            for (int e = 0; e < 8; ++e) {
                const int n = min(16 * (4 * chunk + u) + 8 * half + e, L - 1);
                raw[u][e] = ST::ldr1(reinterpret_cast<const typename ST::T*>(reinterpret_cast<const char*>(A.in) + (loc + (unsigned)n * esb)));
            }
        };
        FFNO_UNROLL
        for (int u = 0; u < 4; ++u) load_rows(0, u);
        f32x16 acc[RT];
        Hf3 Fn[RT];
        FFNO_UNROLL
        for (int rt = 0; rt < RT; ++rt) {
            acc[rt] = zero16();
            Fn[rt] = x3k_load_dft(A.dft, (rt * nchunks) * 4, lane);
        }
        FFNO_NOUNROLL
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const bool more = chunk + 1 < nchunks;
            FFNO_UNROLL
            for (int u = 0; u < 4; ++u) {
                float w[8];
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) w[e] = ST::w1(raw[u][e]) * sx;
                const Hf2 b = split2_8(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
                if (more) load_rows(chunk + 1, u);
                const int nu = u == 3 ? 0 : u + 1, nc = u == 3 ? min(chunk + 1, nchunks - 1) : chunk;
                FFNO_UNROLL
                for (int rt = 0; rt < RT; ++rt) {
                    const Hf3 Fc = Fn[rt];
                    Fn[rt] = x3k_load_dft(A.dft, (rt * nchunks + nc) * 4 + nu, lane);      // one k-step ahead
                    if (32 * rt < 2 * K) acc[rt] = mfma_h2s(Fc, b, acc[rt]);
                }
            }
        }
        float* xs = XS + lw * LSF + 32 * sub + j;
        FFNO_UNROLL
        for (int rt = 0; rt < RT; ++rt) {
            if (32 * rt >= 2 * K) continue;
            FFNO_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * rt + drow(r, half);
                if (row < 2 * K) {
                    const float v = acc[rt][r] * unx;
                    xs[row * RS] = v * rs;
                    if (A.spec_save && live) A.spec_save[(((long)(row >> 1) * R + line) * 2 + (row & 1)) * C + 32 * sub + j] = v;
                }
            }
        }
    } else if constexpr (DFTH2) {
        // The line is requested and split ONCE per wave; every k-step's samples then meet the DFT-matrix fragments of ALL the
        // wave's row tiles (rt = sub, sub + 2, ..: one at <= 32 modes, two at 33..64), which are built on the fly from the
        // twiddle table with a table index that is carried from sample to sample and from chunk to chunk (no modulo in the loop).
        constexpr int RTW = RT / 2;
        float amul[RTW];
        int tbase[RTW], km[RTW], k8[RTW], fidx[RTW];
        bool act[RTW];
        FFNO_UNROLL
        for (int i = 0; i < RTW; ++i) {
            const int rt = sub + 2 * i;
            const int kk = 32 * rt + j, k = kk >> 1, ri = kk & 1;
            const bool rowok = kk < 2 * K;
            const float ck = (A.fwd_ck && !(k == 0 || 2 * k == L)) ? 2.f : 1.f;
            act[i] = 32 * rt < 2 * K;
            amul[i] = rowok ? (ri ? -ck : ck) : 0.f;
            tbase[i] = ri ? L : 0;
            km[i] = rowok ? k : 0;
            k8[i] = (km[i] * 8) % L;
            fidx[i] = (km[i] * 8 * half) % L;
        }
        typename ST::Raw2 raw[4][8];      // (raw words: widened where they are split)
        auto load_rows = [&](int chunk, int u) {
            FFNO_UNROLL
            for (int e = 0; e < 8; ++e) {
                const int n = min(16 * (4 * chunk + u) + 8 * half + e, L - 1);
                raw[u][e] = ST::ldr2(reinterpret_cast<const char*>(A.in) + (lo + (unsigned)n * esb));
            }
        };
        FFNO_UNROLL
        for (int u = 0; u < 4; ++u) load_rows(0, u);
        f32x16 acc[RTW][2];
        FFNO_UNROLL
        for (int i = 0; i < RTW; ++i) acc[i][0] = zero16(), acc[i][1] = zero16();
        FFNO_NOUNROLL
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const bool more = chunk + 1 < nchunks;
            FFNO_UNROLL
            for (int u = 0; u < 4; ++u) {
                float2 w[8];
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) {
                    w[e] = ST::w2(raw[u][e]);
                    w[e].x *= sx, w[e].y *= sx;
                }
                const Hf2 b0 = split2_8(w[0].x, w[1].x, w[2].x, w[3].x, w[4].x, w[5].x, w[6].x, w[7].x);
                const Hf2 b1 = split2_8(w[0].y, w[1].y, w[2].y, w[3].y, w[4].y, w[5].y, w[6].y, w[7].y);
                if (more) load_rows(chunk + 1, u);
                FFNO_UNROLL
                for (int i = 0; i < RTW; ++i) {
                    float f[8];
                    FFNO_UNROLL
                    for (int e = 0; e < 8; ++e) {
                        const int n = 16 * (4 * chunk + u) + 8 * half + e;
                        f[e] = n < L ? amul[i] * tws[tbase[i] + fidx[i]] : 0.f;
                        fidx[i] += km[i];
                        if (fidx[i] >= L) fidx[i] -= L;
                    }
                    fidx[i] += k8[i];
                    if (fidx[i] >= L) fidx[i] -= L;
                    if (act[i]) {
                        const Hf3 Ff = split2s_8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
                        acc[i][0] = mfma_h2s(Ff, b0, acc[i][0]);
                        acc[i][1] = mfma_h2s(Ff, b1, acc[i][1]);
                    }
                }
            }
        }
        float* xs = XS + lw * LSF + 2 * j;
        FFNO_UNROLL
        for (int i = 0; i < RTW; ++i) {
            if (!act[i]) continue;
            FFNO_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * (sub + 2 * i) + drow(r, half);
                if (row < 2 * K) {
                    const float2 v = make_float2(acc[i][0][r] * unx, acc[i][1][r] * unx);
                    *reinterpret_cast<float2*>(xs + row * RS) = make_float2(v.x * rs, v.y * rs);
                    if (A.spec_save && live)
                        *reinterpret_cast<float2*>(A.spec_save + (((long)(row >> 1) * R + line) * 2 + (row & 1)) * C + 2 * j) = v;
                }
            }
        }
    } else if constexpr (!ST::BF16)
    for (int rt = sub; rt < RT; rt += 2) {
        if (32 * rt >= 2 * K) break;
        const int kk = 32 * rt + j, k = kk >> 1, ri = kk & 1;
        const bool rowok = kk < 2 * K;
        const float ck = (A.fwd_ck && !(k == 0 || 2 * k == L)) ? 2.f : 1.f;
        const float amul = rowok ? (ri ? -ck : ck) : 0.f;
        const int tbase = ri ? L : 0;
        const int km = rowok ? k : 0;
        const int k8 = (km * 8) % L;
        float2 raw[4][8];
        auto load_rows = [&](int chunk, int u) {
            FFNO_UNROLL
            for (int e = 0; e < 8; ++e) {
                const int n = min(16 * (4 * chunk + u) + 8 * half + e, L - 1);
                raw[u][e] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(A.in) + (lo + (unsigned)n * esb));
            }
        };
        FFNO_UNROLL
        for (int u = 0; u < 4; ++u) load_rows(0, u);
        f32x16 acc0 = zero16(), acc1 = zero16();
        FFNO_NOUNROLL
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            Bf3 Ff[4];
            int idx = (int)(((long)km * (64 * chunk + 8 * half)) % L);
            FFNO_UNROLL
            for (int u = 0; u < 4; ++u) {
                float f[8];
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) {
                    const int n = 16 * (4 * chunk + u) + 8 * half + e;
                    f[e] = n < L ? amul * tws[tbase + idx] : 0.f;
                    idx += km;
                    if (idx >= L) idx -= L;
                }
                idx += k8;
                if (idx >= L) idx -= L;
                Ff[u] = split3_8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
            }
            FFNO_UNROLL
            for (int u = 0; u < 4; ++u) {
                const Bf3 b0 = split3_8(raw[u][0].x, raw[u][1].x, raw[u][2].x, raw[u][3].x, raw[u][4].x, raw[u][5].x,
                                        raw[u][6].x, raw[u][7].x);
                const Bf3 b1 = split3_8(raw[u][0].y, raw[u][1].y, raw[u][2].y, raw[u][3].y, raw[u][4].y, raw[u][5].y,
                                        raw[u][6].y, raw[u][7].y);
                if (chunk + 1 < nchunks) load_rows(chunk + 1, u);
                acc0 = mfma_x3(Ff[u], b0, acc0);
                acc1 = mfma_x3(Ff[u], b1, acc1);
            }
        }
        float* xs = XS + lw * LSF + 2 * j;
        FFNO_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * rt + drow(r, half);
            if (row < 2 * K) {
                *reinterpret_cast<float2*>(xs + row * RS) = make_float2(acc0[r] * rs, acc1[r] * rs);
                if (A.spec_save && live)
                    *reinterpret_cast<float2*>(A.spec_save + (((long)(row >> 1) * R + line) * 2 + (row & 1)) * C + 2 * j) =
                        make_float2(acc0[r], acc1[r]);
            }
        }
    }
    // The weight fragments of ALL this wave's modes are one stream through a ring: the slots a mode's last products free are
    // refilled with the first fragments of the wave's next mode (an L2 round trip per mode would be exposed otherwise), and the
    // first fragments of its first mode are requested on this side of the barrier.
    constexpr int RING = 8;
    using MixFrag = typename std::conditional<MIXH2, Hf2, Bf3>::type;
    constexpr int MNP = MIXH2 ? 2 : 3;
    auto load_w = [&](const u32x4* __restrict__ wk, int f) {
        MixFrag w;
        if constexpr (MIXH2) {
            w.hi = wk[(f * 2 + 0) * 64 + lane];
            w.lo = wk[(f * 2 + 1) * 64 + lane];
        } else {
            w = x3_load_frag(wk, f, lane);
        }
        return w;
    };
    MixFrag ring[RING];
    if (A.wpk && wave < K) {
        FFNO_UNROLL
        for (int f = 0; f < RING; ++f) ring[f] = load_w(A.wpk + (long)wave * F::MODE_FRAGS * 64 * MNP, f);
    }
    __syncthreads();

    // ---------------- phase 2: per-mode channel mix of the four lines, in place ----------------
    if constexpr (M16) {
        if (A.wpk) {
            // MFMA row r16 = (line (r16 mod 8) >> 1, part r16 & 1); rows 8..15 repeat rows 0..7 and are dropped.  k = input channel
            // 32 s2 + 8 g + e; column j16 of tile t16 = output channel 4 j16 + t16 (a lane ends up with four consecutive channels).
            const int r16 = lane & 15, g = lane >> 4;
            const float* arow = XS + ((r16 & 7) >> 1) * LSF + (r16 & 1) * RS + 8 * g;
            for (int k = wave; k < K; k += F::NW) {
                const u32x4* __restrict__ wk = A.wpk + (long)k * F::MODE_FRAGS * 64 * 2;
                const bool more = k + F::NW < K;
                const u32x4* __restrict__ wn = A.wpk + (long)(more ? k + F::NW : k) * F::MODE_FRAGS * 64 * 2;
                Hf2 a[2];
                FFNO_UNROLL
                for (int s2 = 0; s2 < 2; ++s2) {
                    const float4 v0 = *reinterpret_cast<const float4*>(arow + 2 * k * RS + 32 * s2);
                    const float4 v1 = *reinterpret_cast<const float4*>(arow + 2 * k * RS + 32 * s2 + 4);
                    a[s2] = split2_8(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w);
                }
                f32x4 p[8];
                FFNO_UNROLL
                for (int pt = 0; pt < 8; ++pt) {
                    f32x4 pm = {0.f, 0.f, 0.f, 0.f}, pc = {0.f, 0.f, 0.f, 0.f};
                    FFNO_UNROLL
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const int f = pt * 2 + s2;
                        const Hf2 b = ring[f % RING];
                        if (f + RING < F::MODE_FRAGS)
                            ring[f % RING] = load_w(wk, f + RING);
                        else if (more)
                            ring[f % RING] = load_w(wn, f + RING - F::MODE_FRAGS);
                        pc = plat::mfma_f16_16x16x32(a[s2].lo, b.hi, pc);
                        pc = plat::mfma_f16_16x16x32(a[s2].hi, b.lo, pc);
                        pm = plat::mfma_f16_16x16x32(a[s2].hi, b.hi, pm);
                    }
                    FFNO_UNROLL
                    for (int r = 0; r < 4; ++r) p[pt][r] = __builtin_fmaf(pc[r], kHf2Unscale, pm[r]);
                }
                // D rows 4 g + r of this lane: g = 0 -> lines 0, 1; g = 1 -> lines 2, 3 (r = 2 q + part); g >= 2: the repeated rows
                if (g < 2) {
                    FFNO_UNROLL
                    for (int q = 0; q < 2; ++q) {
                        float yr[4], yi[4];
                        FFNO_UNROLL
                        for (int t = 0; t < 4; ++t) {
                            const float p1r = p[t][2 * q], p1i = p[t][2 * q + 1];
                            const float p2r = p[4 + t][2 * q], p2i = p[4 + t][2 * q + 1];
                            if (A.conj_t == 0) {
                                yr[t] = p1r - p2i;
                                yi[t] = p2r + p1i;
                            } else {
                                yr[t] = p1r + p2i;
                                yi[t] = p1i - p2r;
                            }
                        }
                        float* dst = XS + (2 * g + q) * LSF + 2 * k * RS + 4 * r16;
                        *reinterpret_cast<float4*>(dst) = make_float4(yr[0], yr[1], yr[2], yr[3]);
                        *reinterpret_cast<float4*>(dst + RS) = make_float4(yi[0], yi[1], yi[2], yi[3]);
                    }
                }
            }
            __syncthreads();
        }
    } else
    if (A.wpk) {
        // MFMA row j = (line (j mod 8) >> 1, part j & 1); rows 8..31 repeat rows 0..7 and are dropped
        const float* arow = XS + ((j & (2 * NL - 1)) >> 1) * LSF + (j & 1) * RS + 8 * half;
        for (int k = wave; k < K; k += F::NW) {
            const u32x4* __restrict__ wk = A.wpk + (long)k * F::MODE_FRAGS * 64 * MNP;
            const bool more = k + F::NW < K;
            const u32x4* __restrict__ wn = A.wpk + (long)(more ? k + F::NW : k) * F::MODE_FRAGS * 64 * MNP;
            MixFrag a[4];
            FFNO_UNROLL
            for (int st = 0; st < 4; ++st) {
                const float4 v0 = *reinterpret_cast<const float4*>(arow + 2 * k * RS + 16 * st);
                const float4 v1 = *reinterpret_cast<const float4*>(arow + 2 * k * RS + 16 * st + 4);
                if constexpr (MIXH2)
                    a[st] = split2_8(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w);
                else
                    a[st] = split3_8(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w);
            }
            f32x16 p[4];
            FFNO_UNROLL
            for (int pt = 0; pt < 4; ++pt) p[pt] = zero16();
            if constexpr (MIXH2) {
                FFNO_UNROLL
                for (int pt = 0; pt < 4; ++pt) {
                    f32x16 pc = zero16();
                    FFNO_UNROLL
                    for (int st = 0; st < 4; ++st) {
                        const int f = pt * 4 + st;
                        const Hf2 b = ring[f % RING];
                        if (f + RING < F::MODE_FRAGS)
                            ring[f % RING] = load_w(wk, f + RING);
                        else if (more)
                            ring[f % RING] = load_w(wn, f + RING - F::MODE_FRAGS);
                        mfma_h2(a[st], b, p[pt], pc);
                    }
                    SplitHf2::fold(p[pt], pc);
                }
            } else {
                FFNO_UNROLL
                for (int st = 0; st < 4; ++st) {
                    FFNO_UNROLL
                    for (int pt = 0; pt < 4; ++pt) {
                        const int f = st * 4 + pt;
                        const Bf3 b = ring[f % RING];
                        if (f + RING < F::MODE_FRAGS)
                            ring[f % RING] = load_w(wk, f + RING);
                        else if (more)
                            ring[f % RING] = load_w(wn, f + RING - F::MODE_FRAGS);
                        p[pt] = mfma_x3(a[st], b, p[pt]);
                    }
                }
            }
            // D rows 2q, 2q+1 of this lane = (re, im) of line (q & 1) + 2 half  (accumulator registers 0..3 hold the 8 live rows)
            FFNO_UNROLL
            for (int q = 0; q < NL / 2; ++q) {
                const int ln = (q & 1) + 2 * half;
                float yr[2], yi[2];
                FFNO_UNROLL
                for (int t = 0; t < 2; ++t) {
                    const float p1r = p[t][2 * q], p1i = p[t][2 * q + 1];
                    const float p2r = p[2 + t][2 * q], p2i = p[2 + t][2 * q + 1];
                    if (A.conj_t == 0) {
                        yr[t] = p1r - p2i;
                        yi[t] = p2r + p1i;
                    } else {
                        yr[t] = p1r + p2i;
                        yi[t] = p1i - p2r;
                    }
                }
                float* dst = XS + ln * LSF + 2 * k * RS + 2 * j;
                *reinterpret_cast<float2*>(dst) = make_float2(yr[0], yr[1]);
                *reinterpret_cast<float2*>(dst + RS) = make_float2(yi[0], yi[1]);
            }
        }
        __syncthreads();
    }

    // ---------------- phase 3 ----------------
    {
        const int RTtot = (L + 31) >> 5, NPR = (RTtot + 1) >> 1;
        const unsigned hoff = (unsigned)(4 * half * es * ST::BYTES);
        const unsigned lob = lo + hoff;
        const float* xs = XS + lw * LSF + 2 * j;
        const char* addsrc = A.resid ? reinterpret_cast<const char*>(A.resid)
                                     : (A.accumulate ? reinterpret_cast<const char*>(A.out) : nullptr);
        // fp16x2: power-of-two scales of the line's mixed spectrum, one per column tile (even / odd channels), from its own maximum
        float sy0 = 1.f, sy1 = 1.f, osc0 = rrs, osc1 = rrs;
        if constexpr (DFTH2) {
            float ym0 = 0.f, ym1 = 0.f;
            for (int kk = 8 * half; kk < 2 * K; kk += 16) {
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) {
                    if (kk + e < 2 * K) {
                        const float2 v = *reinterpret_cast<const float2*>(xs + (kk + e) * RS);
                        ym0 = fmaxf(ym0, fabsf(v.x)), ym1 = fmaxf(ym1, fabsf(v.y));
                    }
                }
            }
            FFNO_UNROLL
            for (int sh = 32; sh >= 1; sh >>= 1) ym0 = fmaxf(ym0, __shfl_xor(ym0, sh)), ym1 = fmaxf(ym1, __shfl_xor(ym1, sh));
            sy0 = range_scale(f2u(ym0), 0, 10), sy1 = range_scale(f2u(ym1), 0, 10);
            osc0 = rrs * kHf2Unscale / sy0, osc1 = rrs * kHf2Unscale / sy1;
        }
        for (int np = sub; np < NPR && live; np += 2) {
            const int rt0 = 2 * np;
            f32x16 o[2][2];
            FFNO_UNROLL
            for (int q = 0; q < 2; ++q) o[q][0] = zero16(), o[q][1] = zero16();
            // rows the epilogue adds (residual / accumulate) are requested ahead of the products: the first tile's before the
            // k-step loop, the second tile's while the first is stored
            typename ST::Raw2 pre[16];
            auto request_rows = [&](int rt) {
                FFNO_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int nu = min(32 * rt + (r & 3) + 8 * (r >> 2) + 4 * half, L - 1);
                    pre[r] = ST::ldr2(addsrc + lo + (unsigned)nu * esb);
                }
            };
            if (addsrc) request_rows(rt0);
            const int nst = min(NST, (2 * K + 15) >> 4);
            if constexpr (DFTH2) {
                // table index of the inverse-DFT matrix rows of the two output tiles: n t mod L, carried from mode to mode and from
                // k-step to k-step (+ 8 n per k-step: four modes here, four in the other half-wave)
                int nmq[2], nm4[2], gidx[2];
                bool nok[2];
                FFNO_UNROLL
                for (int q = 0; q < 2; ++q) {
                    const int n = 32 * (rt0 + q) + j;
                    nok[q] = n < L;
                    nmq[q] = nok[q] ? n : 0;
                    nm4[q] = (int)(((long)nmq[q] * 4) % L);
                    gidx[q] = half ? nm4[q] : 0;
                }
                const int dftl_nfwd = RT * nchunks * 4;
                Hf3 Gn[2];
                if constexpr (TAB) {
                    FFNO_UNROLL
                    for (int q = 0; q < 2; ++q) Gn[q] = x3k_load_dft(A.dft, dftl_nfwd + min(rt0 + q, RTtot - 1) * NST, lane);
                }
                FFNO_NOUNROLL
                for (int st = 0; st < nst; ++st) {
                    float2 v[8];
                    FFNO_UNROLL
                    for (int e = 0; e < 8; ++e) {
                        const int kk = 16 * st + 8 * half + e;
                        v[e] = make_float2(0.f, 0.f);
                        if (kk < 2 * K) v[e] = *reinterpret_cast<const float2*>(xs + kk * RS);
                        v[e].x *= sy0, v[e].y *= sy1;
                    }
                    const Hf2 y0 = split2_8(v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x);
                    const Hf2 y1 = split2_8(v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y);
                    if constexpr (TAB) {      // inverse-DFT matrix fragments from the table, one k-step ahead
                        Hf3 Gc[2];
                        FFNO_UNROLL
                        for (int q = 0; q < 2; ++q) Gc[q] = Gn[q];
                        const int sn = min(st + 1, nst - 1);
                        FFNO_UNROLL
                        for (int q = 0; q < 2; ++q)
                            Gn[q] = x3k_load_dft(A.dft, dftl_nfwd + min(rt0 + q, RTtot - 1) * NST + sn, lane);
                        FFNO_UNROLL
                        for (int q = 0; q < 2; ++q) {
                            o[q][0] = mfma_h2s(Gc[q], y0, o[q][0]);
                            o[q][1] = mfma_h2s(Gc[q], y1, o[q][1]);
                        }
                    } else
                    FFNO_UNROLL
                    for (int q = 0; q < 2; ++q) {
                        float g[8];
                        FFNO_UNROLL
                        for (int e = 0; e < 8; ++e) {
                            const int kk = 16 * st + 8 * half + e, t = kk >> 1, part = kk & 1;
                            const float ck = (A.inv_ck && !(t == 0 || 2 * t == L)) ? 2.f : 1.f;
                            g[e] = (kk < 2 * K && nok[q]) ? (part ? -ck : ck) * tws[(part ? L : 0) + gidx[q]] : 0.f;
                            if (part) {
                                gidx[q] += nmq[q];
                                if (gidx[q] >= L) gidx[q] -= L;
                            }
                        }
                        gidx[q] += nm4[q];
                        if (gidx[q] >= L) gidx[q] -= L;
                        const Hf3 G = split2s_8(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
                        o[q][0] = mfma_h2s(G, y0, o[q][0]);
                        o[q][1] = mfma_h2s(G, y1, o[q][1]);
                    }
                }
            } else if constexpr (!ST::BF16)
            FFNO_NOUNROLL
            for (int st = 0; st < nst; ++st) {      // (a real loop: only the four accumulators cross its iterations)
                float2 v[8];
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) {
                    const int kk = 16 * st + 8 * half + e;
                    v[e] = make_float2(0.f, 0.f);
                    if (kk < 2 * K) v[e] = *reinterpret_cast<const float2*>(xs + kk * RS);
                }
                const Bf3 y0 = split3_8(v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x);
                const Bf3 y1 = split3_8(v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y);
                FFNO_UNROLL
                for (int q = 0; q < 2; ++q) {
                    const int n = 32 * (rt0 + q) + j;
                    const int nm = n < L ? n : 0;
                    float g[8];
                    int idx = (int)(((long)nm * (8 * st + 4 * half)) % L);
                    FFNO_UNROLL
                    for (int e = 0; e < 8; ++e) {
                        const int kk = 16 * st + 8 * half + e, t = kk >> 1, part = kk & 1;
                        const float ck = (A.inv_ck && !(t == 0 || 2 * t == L)) ? 2.f : 1.f;
                        g[e] = (kk < 2 * K && n < L) ? (part ? -ck : ck) * tws[(part ? L : 0) + idx] : 0.f;
                        if (part) {
                            idx += nm;
                            if (idx >= L) idx -= L;
                        }
                    }
                    const Bf3 G = split3_8(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
                    o[q][0] = mfma_x3(G, y0, o[q][0]);
                    o[q][1] = mfma_x3(G, y1, o[q][1]);
                }
            }
            FFNO_UNROLL
            for (int q = 0; q < 2; ++q) {
                const int rt = rt0 + q;
                if (rt >= RTtot) continue;
                typename ST::Raw2 cur[16];
                FFNO_UNROLL
                for (int r = 0; r < 16; ++r) cur[r] = pre[r];
                if (addsrc && q == 0 && rt + 1 < RTtot) request_rows(rt + 1);
                FFNO_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int nu = 32 * rt + (r & 3) + 8 * (r >> 2);
                    if (nu + 4 * half < L) {
                        const long uo = (long)nu * es * ST::BYTES;
                        float2 ov = make_float2(o[q][0][r] * osc0, o[q][1][r] * osc1);
                        if (addsrc) {
                            const float2 cw = ST::w2(cur[r]);
                            ov.x += cw.x, ov.y += cw.y;
                        }
                        if (A.accumulate && A.resid) {
                            const float2 pv = ST::ld2(reinterpret_cast<const char*>(A.out) + uo + lob);
                            ov.x += pv.x, ov.y += pv.y;
                        }
                        ST::st2(reinterpret_cast<char*>(A.out) + uo + lob, ov);
                        omax = fmaxf(omax, fmaxf(fabsf(ST::rnd(ov.x)), fabsf(ST::rnd(ov.y))));
                    }
                }
            }
        }
    }
    if (A.out_amax) range_fold(omax, rfold, F::NW, A.out_amax);
}

__device__ __forceinline__ X3Args x3_pick_args(const X3Args& a, const X3Args& b, bool second) {
    X3Args s;
    s.in = second ? b.in : a.in;
    s.out = second ? b.out : a.out;
    s.resid = second ? b.resid : a.resid;
    s.spec_save = second ? b.spec_save : a.spec_save;
    s.wpk = second ? b.wpk : a.wpk;
    s.tw = second ? b.tw : a.tw;
    s.R = second ? b.R : a.R;
    s.L = second ? b.L : a.L;
    s.K = second ? b.K : a.K;
    s.lm.lines_per_group = second ? b.lm.lines_per_group : a.lm.lines_per_group;
    s.lm.group_stride = second ? b.lm.group_stride : a.lm.group_stride;
    s.lm.line_stride = second ? b.lm.line_stride : a.lm.line_stride;
    s.lm.elem_stride = second ? b.lm.elem_stride : a.lm.elem_stride;
    s.fwd_ck = a.fwd_ck, s.inv_ck = a.inv_ck, s.conj_t = a.conj_t;      // common to both branches
    s.accumulate = second ? b.accumulate : a.accumulate;
    s.in_amax = second ? b.in_amax : a.in_amax;
    s.out_amax = second ? b.out_amax : a.out_amax;
    s.dft = second ? b.dft : a.dft;
    s.mix_out = second ? b.mix_out : a.mix_out;
    s.mix_scale = second ? b.mix_scale : a.mix_scale;
    s.self_range = 0;
    return s;
}

template <int KKT, bool MIXH2, class ST = StF32, bool TAB = false, bool M16 = false>
__global__ __launch_bounds__(512) void spectral_x3k_kernel(X3Args a) {
    spectral_x3k_body<KKT, MIXH2, ST, TAB, M16>(a, blockIdx.x);
}
// two branches in one launch: even workgroups run branch a, odd ones branch b while both have tiles left (workgroup w lands on
// XCD w % 8: every XCD's L2 then holds the packed weights of ONE branch), the rest in order
template <int KKT, bool MIXH2, class ST = StF32, bool TAB = false, bool M16 = false>
__global__ __launch_bounds__(512) void spectral_x3k_pair_kernel(X3Args a, X3Args b, int n0, int n1) {
    const int w = blockIdx.x, nmin = min(n0, n1);
    bool second;
    int idx;
    if (w < 2 * nmin) {
        second = w & 1;
        idx = w >> 1;
    } else {
        second = n1 > n0;
        idx = w - nmin;
    }
    spectral_x3k_body<KKT, MIXH2, ST, TAB, M16>(x3_pick_args(a, b, second), idx);
}

template <bool MIXH2, class ST = StF32, bool TAB = false>
__global__ __launch_bounds__(512) void spectral_x3c32_kernel(X3Args a) {
    spectral_x3c32_body<MIXH2, ST, TAB>(a, blockIdx.x);
}
template <bool MIXH2, class ST = StF32, bool TAB = false>
__global__ __launch_bounds__(512) void spectral_x3c32_pair_kernel(X3Args a, X3Args b, int n0, int n1) {
    const int w = blockIdx.x, nmin = min(n0, n1);
    bool second;
    int idx;
    if (w < 2 * nmin) {
        second = w & 1;
        idx = w >> 1;
    } else {
        second = n1 > n0;
        idx = w - nmin;
    }
    spectral_x3c32_body<MIXH2, ST, TAB>(x3_pick_args(a, b, second), idx);
}

// ---- the latency variant: FOUR lines per workgroup, two waves per line (rollout at batch 1: 128 lines per launch) --------------
// With few workgroups a launch is one dependent chain per workgroup -- line load, DFT, barrier, per-mode mix against a weight
// stream from L2, barrier, inverse DFT, store -- and nothing overlaps it (profiles/r02_x3_phase_timing.md, 8-line tiles at batch
// 1: 5.8 + 7.6 + 6.4 us).  Same operator and the same products per output element as spectral_x3_body (bit-identical results);
// the chain is cut three ways: twice as many workgroups (4-line tiles); every line gets TWO waves -- wave (line, t) loads,
// transforms and stores column tile t of its line (the even or the odd channels: half of the operand splits and MFMAs of phases
// 1 and 3); and the ring of weight fragments holds a WHOLE mode (16 fragments in flight per wave instead of 8, refilled with the
// wave's next mode as they are consumed), which halves the L2 round trips the mix waits for.  256 registers per wave.
// CS ("channel split", FFNO_X3_TILE_LATENCY_SPLIT): TWO workgroups per 4-line tile, workgroup h of a tile owning the OUTPUT
// channels of parity h.  Both run phase 1 in full (the mix needs every input channel); in phase 2 each streams only the weight
// fragments of its output column tile (half of the 512 KB a workgroup pulls from L2 per launch -- the floor of the phase at
// batch 1, where 32 workgroups per axis each read the whole pack), and in phase 3 the two waves of a line split the OUTPUT ROW
// tiles of that one column tile instead of the column tiles.  Same products per output element in the same order: bit-identical
// to every other tile choice.  Twice the workgroups (64 per axis at batch 1), each with a shorter dependent chain.
template <bool MIXH2, bool TAB = false, bool CS = false>
__device__ __forceinline__ void spectral_x3s_body(const X3Args A, int bidx_) {
    static_assert(!TAB || MIXH2, "the fragment table holds fp16 planes");
    static_assert(!CS || MIXH2, "the channel-split schedule is written for the fp16x2 packs");
    const int bidx = CS ? bidx_ >> 1 : bidx_;
    const int hc = CS ? (bidx_ & 1) : 0;      // CS: the output channel parity this workgroup owns
    using F = X3Cfg;
    constexpr int C = F::C, RS = F::RS, LSF = F::LSF, NL = 4, NWS = 8;
    __shared__ __attribute__((aligned(16))) float XS[NL * F::LSF];
    __shared__ float rfold[NWS];
    FFNO_DYN_SMEM(smem);
    float* tws = reinterpret_cast<float*>(smem);

    const float* __restrict__ in = A.in;
    const int R = A.R, L = A.L, K = A.K;
    const LineMap lm = A.lm;
    const float rs = (MIXH2 && A.wpk && A.in_amax) ? range_scale(*A.in_amax, 1 + (ceil_log2_int(L) + 1) / 2, 15) : 1.f;
    const float rrs = 1.f / rs;
    // the DFT phases on split-fp16 with the fp16x2 packs, exactly as spectral_x3_body does them (same products, same scales)
    constexpr bool DFTH2 = MIXH2;
    using DftFrag = typename std::conditional<DFTH2, Hf3, Bf3>::type;
    const float sx = (DFTH2 && A.in_amax) ? range_scale(*A.in_amax, 0, 10) : 1.f;
    const float unx = DFTH2 ? kHf2Unscale / sx : 1.f;
    float omax = 0.f;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int lw = wave >> 1, t = wave & 1;          // line inside the tile, column tile (channel parity) of this wave
    const int line0 = bidx * NL + lw;
    const bool live = line0 < R;
    const long es = lm.elem_stride;
    const unsigned esb = (unsigned)(es * 4);
    const unsigned lo = (unsigned)((lm.base(min(line0, R - 1)) + 2 * j + t) * 4);      // this lane's channel 2 j + t

    // ---------------- phase 1: column tile t of the truncated forward DFT of the wave's line ----------------
    {
        const int kk = j, k = kk >> 1, ri = kk & 1;
        const bool rowok = kk < 2 * K;
        const float ck = (A.fwd_ck && !(k == 0 || 2 * k == L)) ? 2.f : 1.f;
        const float amul = rowok ? (ri ? -ck : ck) : 0.f;
        const int tbase = ri ? L : 0;
        const int km = rowok ? k : 0;
        float raw[4][8];
        auto load_rows = [&](int chunk, int u) {
            FFNO_UNROLL
            for (int e = 0; e < 8; ++e) {
                const int n = min(16 * (4 * chunk + u) + 8 * half + e, L - 1);
                raw[u][e] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(in) + (lo + (unsigned)n * esb));
            }
        };
        DftFrag Ff[4];
        const int k8 = (km * 8) % L;
        auto build_F = [&](int chunk) {
            int idx = (km * (64 * chunk + 8 * half)) % L;
            FFNO_UNROLL
            for (int u = 0; u < 4; ++u) {
                float f[8];
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) {
                    const int n = 16 * (4 * chunk + u) + 8 * half + e;
                    f[e] = n < L ? amul * tws[tbase + idx] : 0.f;
                    idx += km;
                    if (idx >= L) idx -= L;
                }
                idx += k8;
                if (idx >= L) idx -= L;
                if constexpr (DFTH2)
                    Ff[u] = split2s_8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
                else
                    Ff[u] = split3_8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
            }
        };
        const int nchunks = (L + 63) >> 6;
        FFNO_UNROLL
        for (int u = 0; u < 4; ++u) load_rows(0, u);
        if constexpr (!TAB) {      // (the table kernels never read the twiddles)
            for (int i = threadIdx.x; i < 2 * L; i += blockDim.x) tws[i] = A.tw[i];
            __syncthreads();
        }
        f32x16 acc = zero16();
        FFNO_NOUNROLL
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            if constexpr (TAB) {      // the four fragments of the chunk from the table (built once per (L, K, direction))
                FFNO_UNROLL
                for (int u = 0; u < 4; ++u) Ff[u] = x3k_load_dft(A.dft, chunk * 4 + u, lane);
            } else {
                build_F(chunk);
            }
            const bool more = chunk + 1 < nchunks;
            FFNO_UNROLL
            for (int u = 0; u < 4; ++u) {
                if constexpr (DFTH2) {
                    const Hf2 b = split2_8(raw[u][0] * sx, raw[u][1] * sx, raw[u][2] * sx, raw[u][3] * sx, raw[u][4] * sx,
                                           raw[u][5] * sx, raw[u][6] * sx, raw[u][7] * sx);
                    if (more) load_rows(chunk + 1, u);
                    acc = mfma_h2s(Ff[u], b, acc);
                } else {
                    const Bf3 b = split3_8(raw[u][0], raw[u][1], raw[u][2], raw[u][3], raw[u][4], raw[u][5], raw[u][6], raw[u][7]);
                    if (more) load_rows(chunk + 1, u);
                    acc = mfma_x3(Ff[u], b, acc);
                }
            }
        }
        float* xs = XS + lw * LSF + 2 * j + t;
        FFNO_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int row = drow(r, half);
            if (row < 2 * K) {
                const float v = acc[r] * unx;
                xs[row * RS] = v * rs;
                if (A.spec_save && live && (!CS || t == hc))
                    A.spec_save[(((long)(row >> 1) * R + line0) * 2 + (row & 1)) * C + 2 * j + t] = v;
            }
        }
    }
    // weight fragments: a ring over ALL this wave's modes (the slots a mode's last products free are refilled with the first
    // fragments of the wave's next mode); with fp16x2 packs the ring holds a whole mode
    constexpr int RING = MIXH2 ? F::MODE_FRAGS : 8;
    using MixFrag = typename std::conditional<MIXH2, Hf2, Bf3>::type;
    constexpr int MNP = MIXH2 ? 2 : 3;
    auto load_w = [&](const u32x4* __restrict__ wk, int f) {
        MixFrag w;
        if constexpr (MIXH2) {
            w.hi = wk[(f * 2 + 0) * 64 + lane];
            w.lo = wk[(f * 2 + 1) * 64 + lane];
        } else {
            w = x3_load_frag(wk, f, lane);
        }
        return w;
    };
    MixFrag ring[RING];
    // CS: fragment g = 0..7 of a mode <-> pack fragment (g < 4 ? hc : 2 + hc) * 4 + (g & 3): the ring holds TWO modes
    auto cs_frag = [&](int g) { return ((g >> 2) * 2 + hc) * 4 + (g & 3); };
    if (A.wpk && wave < K) {
        if constexpr (CS) {
            static_assert(2 * NWS >= F::KK / 2, "two modes per wave cover the kernel's mode range");
            const bool two = wave + NWS < K;
            FFNO_UNROLL
            for (int g = 0; g < 8; ++g) ring[g] = load_w(A.wpk + (long)wave * F::MODE_FRAGS * 64 * MNP, cs_frag(g));
            FFNO_UNROLL
            for (int g = 0; g < 8; ++g)
                ring[8 + g] = load_w(A.wpk + (long)(two ? wave + NWS : wave) * F::MODE_FRAGS * 64 * MNP, cs_frag(g));
        } else {
            FFNO_UNROLL
            for (int f = 0; f < RING; ++f) ring[f] = load_w(A.wpk + (long)wave * F::MODE_FRAGS * 64 * MNP, f);
        }
    }
    __syncthreads();

    // ---------------- phase 2: per-mode channel mix of the four lines (8 live rows of the 32-row tile; rows 8..31 repeat them) ----
    if (A.wpk) {
        const float* arow = XS + ((j & (2 * NL - 1)) >> 1) * LSF + (j & 1) * RS + 8 * half;
        if constexpr (CS) {
            // K <= 16 on 8 waves: a wave has at most two modes (wave, wave + 8), both already in the ring -- no refill.  Per mode
            // the two products of this workgroup's column tile (real-part and imaginary-part weights).
            auto mode = [&](const int k, const int base) {
                Hf2 a[4];
                FFNO_UNROLL
                for (int st = 0; st < 4; ++st) {
                    const float4 v0 = *reinterpret_cast<const float4*>(arow + 2 * k * RS + 16 * st);
                    const float4 v1 = *reinterpret_cast<const float4*>(arow + 2 * k * RS + 16 * st + 4);
                    a[st] = split2_8(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w);
                }
                f32x16 p1 = zero16(), p2 = zero16();
                FFNO_UNROLL
                for (int pp = 0; pp < 2; ++pp) {
                    f32x16 pc = zero16();
                    FFNO_UNROLL
                    for (int st = 0; st < 4; ++st) mfma_h2(a[st], ring[base + pp * 4 + st], pp ? p2 : p1, pc);
                    SplitHf2::fold(pp ? p2 : p1, pc);
                }
                FFNO_UNROLL
                for (int q = 0; q < NL / 2; ++q) {
                    const int line = (q & 1) + 2 * half;
                    const float p1r = p1[2 * q], p1i = p1[2 * q + 1], p2r = p2[2 * q], p2i = p2[2 * q + 1];
                    float yr, yi;
                    if (A.conj_t == 0) {
                        yr = p1r - p2i;
                        yi = p2r + p1i;
                    } else {
                        yr = p1r + p2i;
                        yi = p1i - p2r;
                    }
                    float* dst = XS + line * LSF + 2 * k * RS + 2 * j + hc;
                    dst[0] = yr;
                    dst[RS] = yi;
                }
            };
            if (wave < K) mode(wave, 0);
            if (wave + NWS < K) mode(wave + NWS, 8);
        }
        for (int k = wave; k < (CS ? 0 : K); k += NWS) {
            const u32x4* __restrict__ wk = A.wpk + (long)k * F::MODE_FRAGS * 64 * MNP;
            const bool more = k + NWS < K;
            const u32x4* __restrict__ wn = A.wpk + (long)(more ? k + NWS : k) * F::MODE_FRAGS * 64 * MNP;
            MixFrag a[4];
            FFNO_UNROLL
            for (int st = 0; st < 4; ++st) {
                const float4 v0 = *reinterpret_cast<const float4*>(arow + 2 * k * RS + 16 * st);
                const float4 v1 = *reinterpret_cast<const float4*>(arow + 2 * k * RS + 16 * st + 4);
                if constexpr (MIXH2)
                    a[st] = split2_8(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w);
                else
                    a[st] = split3_8(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w);
            }
            f32x16 p[4];
            FFNO_UNROLL
            for (int pt = 0; pt < 4; ++pt) p[pt] = zero16();
            auto refill = [&](int f) {      // slot f % RING is free: request what will be read from it next
                if (f + RING < F::MODE_FRAGS)
                    ring[f % RING] = load_w(wk, f + RING);
                else if (more)
                    ring[f % RING] = load_w(wn, f + RING - F::MODE_FRAGS);
            };
            if constexpr (MIXH2) {
                FFNO_UNROLL
                for (int pt = 0; pt < 4; ++pt) {
                    f32x16 pc = zero16();
                    FFNO_UNROLL
                    for (int st = 0; st < 4; ++st) {
                        const int f = pt * 4 + st;
                        const Hf2 b = ring[f % RING];
                        refill(f);
                        mfma_h2(a[st], b, p[pt], pc);
                    }
                    SplitHf2::fold(p[pt], pc);
                }
            } else {
                FFNO_UNROLL
                for (int st = 0; st < 4; ++st) {
                    FFNO_UNROLL
                    for (int pt = 0; pt < 4; ++pt) {
                        const int f = st * 4 + pt;
                        const Bf3 b = ring[f % RING];
                        refill(f);
                        p[pt] = mfma_x3(a[st], b, p[pt]);
                    }
                }
            }
            // D rows 2q, 2q+1 of this lane = (re, im) of line (q & 1) + 2 half  (accumulator registers 0..3 hold the 8 live rows)
            FFNO_UNROLL
            for (int q = 0; q < NL / 2; ++q) {
                const int line = (q & 1) + 2 * half;
                float yr[2], yi[2];
                FFNO_UNROLL
                for (int tt = 0; tt < 2; ++tt) {
                    const float p1r = p[tt][2 * q], p1i = p[tt][2 * q + 1];
                    const float p2r = p[2 + tt][2 * q], p2i = p[2 + tt][2 * q + 1];
                    if (A.conj_t == 0) {
                        yr[tt] = p1r - p2i;
                        yi[tt] = p2r + p1i;
                    } else {
                        yr[tt] = p1r + p2i;
                        yi[tt] = p1i - p2r;
                    }
                }
                float* dst = XS + line * LSF + 2 * k * RS + 2 * j;
                *reinterpret_cast<float2*>(dst) = make_float2(yr[0], yr[1]);
                *reinterpret_cast<float2*>(dst + RS) = make_float2(yi[0], yi[1]);
            }
        }
        __syncthreads();
    }

    // ---------------- phase 3: column tile t of the zero-padded inverse DFT of the wave's line ----------------
    if (live) {
        const int RTtot = (L + 31) >> 5;
        const unsigned hoff = (unsigned)(4 * half * es * 4);
        const int ct = CS ? hc : t;               // the column tile this wave transforms back (CS: the workgroup's, rows split)
        const unsigned lo = (unsigned)((lm.base(line0) + 2 * j + ct) * 4);
        const unsigned lob = lo + hoff;
        const float* xs = XS + lw * LSF + 2 * j + ct;
        const char* addsrc = A.resid ? reinterpret_cast<const char*>(A.resid)
                                     : (A.accumulate ? reinterpret_cast<const char*>(A.out) : nullptr);
        // B operands: the line's spectrum, column tile t, split: slot e of k-step st <-> row kk = 16 st + 8 half + e
        using SpecFrag = typename std::conditional<DFTH2, Hf2, Bf3>::type;
        SpecFrag y[2];
        float v[2][8];
        FFNO_UNROLL
        for (int st = 0; st < 2; ++st) {
            FFNO_UNROLL
            for (int e = 0; e < 8; ++e) {
                const int kk = 16 * st + 8 * half + e;
                v[st][e] = kk < 2 * K ? xs[kk * RS] : 0.f;
            }
        }
        float osc = rrs;
        if constexpr (DFTH2) {      // the scale of this column tile of the line, from its own maximum (spectral_x3_body)
            float ym = 0.f;
            FFNO_UNROLL
            for (int st = 0; st < 2; ++st) {
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) ym = fmaxf(ym, fabsf(v[st][e]));
            }
            FFNO_UNROLL
            for (int sh = 32; sh >= 1; sh >>= 1) ym = fmaxf(ym, __shfl_xor(ym, sh));
            const float sy = range_scale(f2u(ym), 0, 10);
            osc = rrs * kHf2Unscale / sy;
            FFNO_UNROLL
            for (int st = 0; st < 2; ++st) {
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) v[st][e] *= sy;
            }
        }
        FFNO_UNROLL
        for (int st = 0; st < 2; ++st) {
            if constexpr (DFTH2)
                y[st] = split2_8(v[st][0], v[st][1], v[st][2], v[st][3], v[st][4], v[st][5], v[st][6], v[st][7]);
            else
                y[st] = split3_8(v[st][0], v[st][1], v[st][2], v[st][3], v[st][4], v[st][5], v[st][6], v[st][7]);
        }
        FFNO_NOUNROLL
        for (int rt = CS ? t : 0; rt < RTtot; rt += CS ? 2 : 1) {
            float pre[16];
            if (addsrc) {
                FFNO_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int nu = min(32 * rt + (r & 3) + 8 * (r >> 2) + 4 * half, L - 1);
                    pre[r] = *reinterpret_cast<const float*>(addsrc + lo + (unsigned)nu * esb);
                }
            }
            // inverse-DFT matrix fragments of this 32-row output tile: row n, slot e of k-step st <-> kk = (mode, part)
            const int n = 32 * rt + j;
            f32x16 o = zero16();
            Hf3 Gt[2];
            if constexpr (TAB) {
                FFNO_UNROLL
                for (int st = 0; st < 2; ++st) Gt[st] = x3k_load_dft(A.dft, ((L + 63) >> 6) * 4 + rt * 2 + st, lane);
            }
            FFNO_UNROLL
            for (int st = 0; st < 2; ++st) {
                float g[8];
                const int nm = n < L ? n : 0;
                int idx = (nm * (8 * st + 4 * half)) % L;
                FFNO_UNROLL
                for (int e = 0; e < 8; ++e) {
                    const int kk = 16 * st + 8 * half + e, tm = kk >> 1, part = kk & 1;
                    const float ck = (A.inv_ck && !(tm == 0 || 2 * tm == L)) ? 2.f : 1.f;
                    g[e] = (kk < 2 * K && n < L) ? (part ? -ck : ck) * tws[(part ? L : 0) + idx] : 0.f;
                    if (part) {
                        idx += nm;
                        if (idx >= L) idx -= L;
                    }
                }
                if constexpr (TAB)
                    o = mfma_h2s(Gt[st], y[st], o);
                else if constexpr (DFTH2)
                    o = mfma_h2s(split2s_8(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]), y[st], o);
                else
                    o = mfma_x3(split3_8(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]), y[st], o);
            }
            FFNO_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int nu = 32 * rt + (r & 3) + 8 * (r >> 2);
                if (nu + 4 * half < L) {
                    const long uo = (long)nu * es * 4;
                    float ov = o[r] * osc;
                    if (addsrc) ov += pre[r];
                    if (A.accumulate && A.resid) ov += *reinterpret_cast<const float*>(reinterpret_cast<const char*>(A.out) + uo + lob);
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(A.out) + uo + lob) = ov;
                    omax = fmaxf(omax, fabsf(ov));
                }
            }
        }
    }
    if (A.out_amax) range_fold(omax, rfold, NWS, A.out_amax);
}

template <bool MIXH2, bool TAB = false, bool CS = false>
__global__ __launch_bounds__(512) FFNO_WAVES_PER_SIMD(2) void spectral_x3s_kernel(X3Args a) {
    spectral_x3s_body<MIXH2, TAB, CS>(a, blockIdx.x);
}
// (CS: n0 / n1 count WORKGROUPS, two per tile)
template <bool MIXH2, bool TAB = false, bool CS = false>
__global__ __launch_bounds__(512) FFNO_WAVES_PER_SIMD(2) void spectral_x3s_pair_kernel(X3Args a, X3Args b, int n0, int n1) {
    const int w = blockIdx.x, nmin = min(n0, n1);
    bool second;
    int idx;
    if (w < 2 * nmin) {
        second = w & 1;
        idx = w >> 1;
    } else {
        second = n1 > n0;
        idx = w - nmin;
    }
    spectral_x3s_body<MIXH2, TAB, CS>(x3_pick_args(a, b, second), idx);
}

// 8-line tiles while the launch still fits one round of workgroups (one per CU of the device): more CUs busy, same weight
// stream per workgroup; a branch descriptor may force either (tile_lines = 8 / 16; results are bit-identical)
static inline bool x3_small_tiles(int Ra, int Rb, int tile_lines) {
    if (tile_lines == 8) return true;
    if (tile_lines == 16) return false;
    return (Ra + 7) / 8 + (Rb + 7) / 8 <= device_cu_count();
}
// ... and 4-line tiles with two waves per line (spectral_x3s_body) while THOSE fit one round of workgroups: such a launch is a
// latency chain per workgroup, not a bandwidth problem (rollout at batch 1: 32 workgroups per axis; measured on MI355X, markov/24
// forward at batch 1 / 2 / 4 / 8: 0.70 / 0.77 / 0.89 / 1.08 ms against 0.76 / 0.82 / 0.93 / 1.11 on 8-line tiles).  fp32 storage.
static inline bool x3_latency_tiles(int Ra, int Rb, int tile_lines, int storage) {
    if (storage != FFNO_STORE_F32) return false;
    if (tile_lines == FFNO_X3_TILE_LATENCY || tile_lines == FFNO_X3_TILE_LATENCY_SPLIT) return true;
    if (tile_lines != 0) return false;
    return (Ra + 3) / 4 + (Rb + 3) / 4 <= device_cu_count();
}
// ... and TWO workgroups per such tile (channel split, spectral_x3s_body<.., CS>) while the doubled launch still fits one
// round: fp16x2 packs with a mix (the split halves the weight stream of a workgroup; a low-pass launch has none)
static inline bool x3_latency_split(int Ra, int Rb, int tile_lines, bool h2_mix) {
    if (!h2_mix) return false;
    if (tile_lines == FFNO_X3_TILE_LATENCY_SPLIT) return true;
    if (tile_lines != 0) return false;
    return 2 * ((Ra + 3) / 4 + (Rb + 3) / 4) <= device_cu_count();
}

static inline bool x3_many_modes(int K) { return 2 * K > X3Cfg::KK; }

static inline int x3_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFNO_OK : (int)e;
}

}  // namespace ffno

using namespace ffno;

// fused x3 kernels: K <= 16 on the 16 / 8-line tile (spectral_x3_body), 17..64 modes on the 4-line tile (spectral_x3k_body)
extern "C" int ffno_spectral_x3_supported(int C, int K, int L) {
    if (C == X3Cfg32::C) return (K >= 1 && 2 * K <= X3Cfg32::KK && L >= 2 && L <= 2048) ? 1 : 0;      // width 32: K <= 16
    return (C == X3Cfg::C && K >= 1 && K <= 64 && L >= 2 && L <= 2048) ? 1 : 0;
}
extern "C" int ffno_spectral_x3_staged_supported(int C, int K, int L) {
    return (C == X3Cfg::C && K >= 1 && K <= 32 && L >= 2 && L <= 2048) ? 1 : 0;
}

extern "C" size_t ffno_spectral_x3_pack_bytes(int C, int K) {
    if (C == X3Cfg32::C) return (K >= 1 && K <= 16) ? (size_t)K * X3Cfg32::MODE_FRAGS * X3Cfg::FRAG * sizeof(u32x4) : 0;
    return (C == X3Cfg::C && K >= 1 && K <= 64) ? (size_t)K * X3Cfg::MODE_FRAGS * X3Cfg::FRAG * sizeof(u32x4) : 0;
}

extern "C" size_t ffno_spectral_x3_dft_frags_bytes(int L, int K) {
    if (K < 1 || K > 64 || L < 2 || L > 2048) return 0;
    const X3kDft d = x3k_dft_layout(L, K);
    return (size_t)(d.nfwd + d.ntiles * d.NST) * 2 * 64 * sizeof(u32x4);
}
extern "C" int ffno_spectral_x3_dft_frags(const float* tw, int L, int K, int scale_ck_fwd, int apply_ck_inv, void* frags,
                                          void* stream) {
    if (!tw || !frags) return FFNO_EINVAL;
    if (K > L / 2 + 1) return FFNO_EMODES;
    if (!ffno_spectral_x3_dft_frags_bytes(L, K)) return FFNO_EUNSUPPORTED;
    const X3kDft d = x3k_dft_layout(L, K);
    FFNO_LAUNCH(x3k_dft_frags_kernel, dim3(d.nfwd + d.ntiles * d.NST), dim3(64), 0, (hipStream_t)stream, tw, L, K, scale_ck_fwd,
                apply_ck_inv, d, reinterpret_cast<u32x4*>(frags));
    return x3_status();
}

extern "C" int ffno_spectral_x3_pack(const ffno_x3pack_desc* descs_dev, int n, int C, int max_K, void* stream) {
    if (!descs_dev || n <= 0 || max_K <= 0) return FFNO_EINVAL;
    static_assert(sizeof(ffno_x3pack_desc) == sizeof(X3PackDesc), "descriptor layout");
    if (C == X3Cfg32::C) {
        if (max_K > 16) return FFNO_EUNSUPPORTED;
        const int threads = max_K * X3Cfg32::MODE_FRAGS * 64;
        FFNO_LAUNCH(x3_pack32_kernel, dim3((threads + 255) / 256, n), dim3(256), 0, (hipStream_t)stream,
                    reinterpret_cast<const X3PackDesc*>(descs_dev));
        return x3_status();
    }
    if (C != X3Cfg::C || max_K > 64) return FFNO_EUNSUPPORTED;
    const int threads = max_K * X3Cfg::MODE_FRAGS * 64;
    FFNO_LAUNCH(x3_pack_kernel, dim3((threads + 255) / 256, n), dim3(256), 0, (hipStream_t)stream,
                reinterpret_cast<const X3PackDesc*>(descs_dev));
    return x3_status();
}

static int x3_args(X3Args& a, const ffno_fused_branch* b, int C, int scale_ck_fwd, int apply_ck_inv, int conj_transpose) {
    if (!b || !b->in || !b->out || !b->tw || b->B <= 0 || b->M <= 0 || b->N <= 0 || b->K <= 0 || (b->axis != 0 && b->axis != 1))
        return FFNO_EINVAL;
    const int L = b->axis == 0 ? b->N : b->M;
    const int R = b->axis == 0 ? b->B * b->M : b->B * b->N;
    if (b->K > L / 2 + 1) return FFNO_EMODES;
    if (!ffno_spectral_x3_supported(C, b->K, L)) return FFNO_EUNSUPPORTED;
    if (b->planes_format != FFNO_PLANES_BF16X3 && b->planes_format != FFNO_PLANES_FP16X2 && b->planes_format != FFNO_PLANES_FP16X2_M16)
        return FFNO_EINVAL;
    // format 2 = the 16-row mix of the many-mode kernel (width 64, 17..64 modes), which also takes its DFT fragments from a table
    if (b->planes && b->planes_format == FFNO_PLANES_FP16X2_M16 && (C != X3Cfg::C || !x3_many_modes(b->K) || !b->dft_frags))
        return FFNO_EUNSUPPORTED;
    if (b->tile_lines != 0 && b->tile_lines != 8 && b->tile_lines != 16 && b->tile_lines != FFNO_X3_TILE_LATENCY &&
        b->tile_lines != FFNO_X3_TILE_LATENCY_SPLIT)
        return FFNO_EINVAL;
    if (b->storage != FFNO_STORE_F32 && b->storage != FFNO_STORE_BF16) return FFNO_EINVAL;
    if (b->storage == FFNO_STORE_BF16 && (b->tile_lines == FFNO_X3_TILE_LATENCY || b->tile_lines == FFNO_X3_TILE_LATENCY_SPLIT))
        return FFNO_EUNSUPPORTED;
    if (b->tile_lines == FFNO_X3_TILE_LATENCY_SPLIT && !(b->planes && b->planes_format == FFNO_PLANES_FP16X2))
        return FFNO_EUNSUPPORTED;      // (the split divides the fp16x2 weight stream: it needs one)
    // bf16 storage twins: every fused split kernel (width 64 with <= 64 modes, width 32), on the split-fp16 path -- i.e. WITH
    // fp16x2 packs (a launch without planes, mode 'low-pass', runs the bf16x3 DFT, which has twins only on the K <= 16 kernel)
    if (b->storage == FFNO_STORE_BF16) {
        if (b->planes && b->planes_format == FFNO_PLANES_BF16X3) return FFNO_EUNSUPPORTED;
        if (!b->planes && (C != X3Cfg::C || x3_many_modes(b->K))) return FFNO_EUNSUPPORTED;
    }
    a = X3Args{b->in, b->out, b->resid, b->spec_save, reinterpret_cast<const u32x4*>(b->planes), b->tw, R, L, b->K,
               make_linemap(b->axis, b->B, b->M, b->N, C), scale_ck_fwd, apply_ck_inv, conj_transpose, b->accumulate,
               b->in_amax, b->out_amax,
               // (only the split-fp16 DFT path of the many-mode kernel reads the table)
               (b->planes && b->planes_format != FFNO_PLANES_BF16X3) ? reinterpret_cast<const u32x4*>(b->dft_frags) : nullptr};
    return FFNO_OK;
}

extern "C" int ffno_spectral_x3(const ffno_fused_branch* br, int C, int scale_ck_fwd, int apply_ck_inv, int conj_transpose,
                                void* stream) {
    X3Args a;
    const int rc = x3_args(a, br, C, scale_ck_fwd, apply_ck_inv, conj_transpose);
    if (rc) return rc;
    // 8-line tiles when they still fit one round of workgroups (one per CU), else 16-line tiles
    const bool h2 = br->planes && br->planes_format != FFNO_PLANES_BF16X3;
    const bool m16 = br->planes && br->planes_format == FFNO_PLANES_FP16X2_M16;      // (x3_args: many modes + table)
    const size_t smem = sizeof(float) * 2 * a.L;
    hipStream_t st = (hipStream_t)stream;
    const bool b16 = br->storage == FFNO_STORE_BF16;      // (x3_args: only with fp16x2 packs outside the K <= 16 kernel)
    if (C == X3Cfg32::C) {          // width 32: 16 lines per workgroup, two per wave side by side
        const dim3 grid((a.R + 15) / 16);
        if (b16 && a.dft)
            FFNO_LAUNCH((spectral_x3c32_kernel<true, StBf16, true>), grid, dim3(512), smem, st, a);
        else if (b16)
            FFNO_LAUNCH((spectral_x3c32_kernel<true, StBf16>), grid, dim3(512), smem, st, a);
        else if (h2 && a.dft)
            FFNO_LAUNCH((spectral_x3c32_kernel<true, StF32, true>), grid, dim3(512), smem, st, a);
        else if (h2)
            FFNO_LAUNCH((spectral_x3c32_kernel<true>), grid, dim3(512), smem, st, a);
        else
            FFNO_LAUNCH((spectral_x3c32_kernel<false>), grid, dim3(512), smem, st, a);
        return x3_status();
    }
    if (x3_many_modes(a.K)) {      // 17..64 modes: the 4-line tile
        const dim3 grid((a.R + 3) / 4);
#define X3K_LAUNCH(...)                                                                                  \
    do {                                                                                                 \
        const int rc_ = allow_dynamic_lds(spectral_x3k_kernel<__VA_ARGS__>, smem);                       \
        if (rc_) return rc_;                                                                             \
        FFNO_LAUNCH((spectral_x3k_kernel<__VA_ARGS__>), grid, dim3(512), smem, st, a);                   \
    } while (0)
        const bool tab = a.dft != nullptr;       // (x3_args: only with fp16x2 planes)
        if (a.K <= 32) {
            if (m16 && b16) X3K_LAUNCH(64, true, StBf16, true, true); else if (m16) X3K_LAUNCH(64, true, StF32, true, true);
            else if (b16 && tab) X3K_LAUNCH(64, true, StBf16, true); else if (b16) X3K_LAUNCH(64, true, StBf16);
            else if (h2 && tab) X3K_LAUNCH(64, true, StF32, true); else if (h2) X3K_LAUNCH(64, true); else X3K_LAUNCH(64, false);
        } else {
            if (m16 && b16) X3K_LAUNCH(128, true, StBf16, true, true); else if (m16) X3K_LAUNCH(128, true, StF32, true, true);
            else if (b16 && tab) X3K_LAUNCH(128, true, StBf16, true); else if (b16) X3K_LAUNCH(128, true, StBf16);
            else if (h2 && tab) X3K_LAUNCH(128, true, StF32, true); else if (h2) X3K_LAUNCH(128, true); else X3K_LAUNCH(128, false);
        }
#undef X3K_LAUNCH
        return x3_status();
    }
    if (x3_latency_tiles(a.R, 0, br->tile_lines, br->storage)) {
        const bool cs = x3_latency_split(a.R, 0, br->tile_lines, h2 && a.wpk);
        const dim3 grid((cs ? 2 : 1) * ((a.R + 3) / 4));
        if (cs && a.dft)
            FFNO_LAUNCH((spectral_x3s_kernel<true, true, true>), grid, dim3(512), smem, st, a);
        else if (cs)
            FFNO_LAUNCH((spectral_x3s_kernel<true, false, true>), grid, dim3(512), smem, st, a);
        else if (h2 && a.dft)
            FFNO_LAUNCH((spectral_x3s_kernel<true, true>), grid, dim3(512), smem, st, a);
        else if (h2)
            FFNO_LAUNCH((spectral_x3s_kernel<true>), grid, dim3(512), smem, st, a);
        else
            FFNO_LAUNCH((spectral_x3s_kernel<false>), grid, dim3(512), smem, st, a);
        return x3_status();
    }
    if (br->storage == FFNO_STORE_BF16) {
        if (x3_small_tiles(a.R, 0, br->tile_lines))
            FFNO_LAUNCH((spectral_x3_kernel<8, true, StBf16>), dim3((a.R + 7) / 8), dim3(512), smem, st, a);
        else
            FFNO_LAUNCH((spectral_x3_kernel<16, true, StBf16>), dim3((a.R + 15) / 16), dim3(512), smem, st, a);
        return x3_status();
    }
    if (x3_small_tiles(a.R, 0, br->tile_lines)) {
        if (h2)
            FFNO_LAUNCH((spectral_x3_kernel<8, true>), dim3((a.R + 7) / 8), dim3(512), smem, st, a);
        else
            FFNO_LAUNCH((spectral_x3_kernel<8, false>), dim3((a.R + 7) / 8), dim3(512), smem, st, a);
    } else {
        if (h2)
            FFNO_LAUNCH((spectral_x3_kernel<16, true>), dim3((a.R + 15) / 16), dim3(512), smem, st, a);
        else
            FFNO_LAUNCH((spectral_x3_kernel<16, false>), dim3((a.R + 15) / 16), dim3(512), smem, st, a);
    }
    return x3_status();
}

extern "C" int ffno_spectral_x3_pair(const ffno_fused_branch* ba, const ffno_fused_branch* bb, int C, int scale_ck_fwd,
                                     int apply_ck_inv, int conj_transpose, int interleave, void* stream) {
    if (!ba || !bb) return FFNO_EINVAL;
    if (ba->out == bb->out) return FFNO_EINVAL;      // concurrent workgroups: the branches may not share an output
    X3Args a, b;
    int rc = x3_args(a, ba, C, scale_ck_fwd, apply_ck_inv, conj_transpose);
    if (rc) return rc;
    rc = x3_args(b, bb, C, scale_ck_fwd, apply_ck_inv, conj_transpose);
    if (rc) return rc;
    const size_t smem = sizeof(float) * 2 * max(a.L, b.L);
    // both branches of a pair carry the same kind of planes (or none) and the same tile choice
    if ((ba->planes == nullptr) != (bb->planes == nullptr) || ba->planes_format != bb->planes_format ||
        ba->tile_lines != bb->tile_lines || ba->storage != bb->storage)
        return FFNO_EINVAL;
    const bool h2 = ba->planes && ba->planes_format != FFNO_PLANES_BF16X3;
    const bool m16 = ba->planes && ba->planes_format == FFNO_PLANES_FP16X2_M16;
    hipStream_t st = (hipStream_t)stream;
    const bool b16 = ba->storage == FFNO_STORE_BF16;
    if (b16 && !h2 && (C != X3Cfg::C || x3_many_modes(a.K) || x3_many_modes(b.K))) return FFNO_EUNSUPPORTED;
    if (C == X3Cfg32::C) {
        const int n0 = (a.R + 15) / 16, n1 = (b.R + 15) / 16;
        const bool tab = a.dft && b.dft;
        if (b16 && tab)
            FFNO_LAUNCH((spectral_x3c32_pair_kernel<true, StBf16, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, n1);
        else if (b16)
            FFNO_LAUNCH((spectral_x3c32_pair_kernel<true, StBf16>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, n1);
        else if (h2 && tab)
            FFNO_LAUNCH((spectral_x3c32_pair_kernel<true, StF32, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, n1);
        else if (h2)
            FFNO_LAUNCH((spectral_x3c32_pair_kernel<true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, n1);
        else
            FFNO_LAUNCH((spectral_x3c32_pair_kernel<false>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, n1);
        return x3_status();
    }
    if (x3_many_modes(a.K) || x3_many_modes(b.K)) {      // 17..64 modes on either axis: both on the 4-line tile
        const int n0 = (a.R + 3) / 4, n1 = (b.R + 3) / 4;
        const dim3 grid(n0 + n1);
#define X3K_LAUNCH(...)                                                                                  \
    do {                                                                                                 \
        const int rc_ = allow_dynamic_lds(spectral_x3k_pair_kernel<__VA_ARGS__>, smem);                  \
        if (rc_) return rc_;                                                                             \
        FFNO_LAUNCH((spectral_x3k_pair_kernel<__VA_ARGS__>), grid, dim3(512), smem, st, a, b, n0, n1);   \
    } while (0)
        // the table kernel needs BOTH branches' tables, laid out for the launch's tile height (64 / 128 rows): a branch with
        // <= 16 modes dragged onto this kernel, or <= 32 next to > 32, has none that fits -- then both build on the fly
        const int kkt = max(a.K, b.K) <= 32 ? 64 : 128;
        const bool tab = a.dft && b.dft && x3_many_modes(a.K) && x3_many_modes(b.K) && (a.K <= 32 ? 64 : 128) == kkt &&
                         (b.K <= 32 ? 64 : 128) == kkt;
        if (!tab) a.dft = b.dft = nullptr;
        if (m16 && !tab) return FFNO_EUNSUPPORTED;      // (the 16-row mix ships with the table kernels only)
        if (kkt == 64) {
            if (m16 && b16) X3K_LAUNCH(64, true, StBf16, true, true); else if (m16) X3K_LAUNCH(64, true, StF32, true, true);
            else if (b16 && tab) X3K_LAUNCH(64, true, StBf16, true); else if (b16) X3K_LAUNCH(64, true, StBf16);
            else if (h2 && tab) X3K_LAUNCH(64, true, StF32, true); else if (h2) X3K_LAUNCH(64, true); else X3K_LAUNCH(64, false);
        } else {
            if (m16 && b16) X3K_LAUNCH(128, true, StBf16, true, true); else if (m16) X3K_LAUNCH(128, true, StF32, true, true);
            else if (b16 && tab) X3K_LAUNCH(128, true, StBf16, true); else if (b16) X3K_LAUNCH(128, true, StBf16);
            else if (h2 && tab) X3K_LAUNCH(128, true, StF32, true); else if (h2) X3K_LAUNCH(128, true); else X3K_LAUNCH(128, false);
        }
#undef X3K_LAUNCH
        return x3_status();
    }
    // interleave: bit 0 = even / odd workgroup -> branch map; bit 1 = image-local map where the shapes allow it (else bit 0
    // decides); bits 8.. = start skew of every other workgroup in units of 256 cycles
    const int skew = (interleave >> 8) * 256;
    auto wg_map = [&](int NL, int n0, int n1) {
        if (n0 != n1) return 0;
        const bool square = ba->B == bb->B && ba->M == bb->M && ba->N == bb->N && ba->M == ba->N && ba->axis != bb->axis;
        if ((interleave & 2) && square && ba->B % 8 == 0 && ba->M % NL == 0) return 2 | ((ba->M / NL) << 8);
        return (interleave & 1) ? 1 : 0;
    };
    if (x3_latency_tiles(a.R, b.R, ba->tile_lines, ba->storage)) {
        const bool cs = x3_latency_split(a.R, b.R, ba->tile_lines, h2 && a.wpk && b.wpk);
        const int n0 = (cs ? 2 : 1) * ((a.R + 3) / 4), n1 = (cs ? 2 : 1) * ((b.R + 3) / 4);
        if (cs && a.dft && b.dft)
            FFNO_LAUNCH((spectral_x3s_pair_kernel<true, true, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, n1);
        else if (cs)
            FFNO_LAUNCH((spectral_x3s_pair_kernel<true, false, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, n1);
        else if (h2 && a.dft && b.dft)
            FFNO_LAUNCH((spectral_x3s_pair_kernel<true, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, n1);
        else if (h2)
            FFNO_LAUNCH((spectral_x3s_pair_kernel<true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, n1);
        else
            FFNO_LAUNCH((spectral_x3s_pair_kernel<false>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, n1);
        return x3_status();
    }
    // <= 16 modes on the split-fp16 DFT with both branches' fragment tables (ffno_spectral_x3_dft_frags for the launch's direction):
    // no wave rebuilds the DFT matrices from the twiddles (same values: results bit-identical with and without the tables)
    const bool tab16 = h2 && a.wpk && b.wpk && a.dft && b.dft;
    if (ba->storage == FFNO_STORE_BF16) {
        if (x3_small_tiles(a.R, b.R, ba->tile_lines)) {
            const int n0 = (a.R + 7) / 8, n1 = (b.R + 7) / 8, il = wg_map(8, n0, n1);
            if (tab16)
                FFNO_LAUNCH((spectral_x3_pair_kernel<8, true, StBf16, false, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, il, skew);
            else
                FFNO_LAUNCH((spectral_x3_pair_kernel<8, true, StBf16>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, il, skew);
        } else {
            const int n0 = (a.R + 15) / 16, n1 = (b.R + 15) / 16, il = wg_map(16, n0, n1);
            if (tab16)
                FFNO_LAUNCH((spectral_x3_pair_kernel<16, true, StBf16, false, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, il, skew);
            else
                FFNO_LAUNCH((spectral_x3_pair_kernel<16, true, StBf16>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, il, skew);
        }
        return x3_status();
    }
    if (x3_small_tiles(a.R, b.R, ba->tile_lines)) {
        const int n0 = (a.R + 7) / 8, n1 = (b.R + 7) / 8, il = wg_map(8, n0, n1);
        if (tab16)
            FFNO_LAUNCH((spectral_x3_pair_kernel<8, true, StF32, false, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, il, skew);
        else if (h2)
            FFNO_LAUNCH((spectral_x3_pair_kernel<8, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, il, skew);
        else
            FFNO_LAUNCH((spectral_x3_pair_kernel<8, false>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, il, skew);
    } else {
        const int n0 = (a.R + 15) / 16, n1 = (b.R + 15) / 16, il = wg_map(16, n0, n1);
        if (tab16)
            FFNO_LAUNCH((spectral_x3_pair_kernel<16, true, StF32, false, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, il, skew);
        else if (h2)
            FFNO_LAUNCH((spectral_x3_pair_kernel<16, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, il, skew);
        else
            FFNO_LAUNCH((spectral_x3_pair_kernel<16, false>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, il, skew);
    }
    return x3_status();
}

// ---- first kernel of the INFERENCE layer (infer.hip): both forward transforms + both channel mixes -> the mixed spectra -------
// Same branch descriptors as ffno_spectral_x3_pair (forward flags), with `out` = the branch's MIX buffer
// (ffno_infer_mix_bytes): per line 8 split-fp16 operand fragments (8 KiB) and, behind all lines, one float per line.
extern "C" size_t ffno_infer_mix_bytes(int C, int K, int lines) {
    if (C != X3Cfg::C || K < 1 || 2 * K > X3Cfg::KK || lines <= 0) return 0;
    return (size_t)lines * (8 * 64 * sizeof(u32x4)) + (((size_t)lines * sizeof(float) + 15) & ~(size_t)15);
}

// launch arguments of the first inference kernel from the two branch descriptors (validated): shared by ffno_spectral_x3_mix_pair and
// ffno_infer_stack
static int mix_pair_args(X3Args& a, X3Args& b, const ffno_fused_branch* ba, const ffno_fused_branch* bb, int C) {
    if (!ba || !bb) return FFNO_EINVAL;
    if (ba->out == bb->out) return FFNO_EINVAL;
    int rc = x3_args(a, ba, C, 0, 1, 0);
    if (rc) return rc;
    rc = x3_args(b, bb, C, 0, 1, 0);
    if (rc) return rc;
    // what the fragment form covers: width 64, <= 16 modes, fp16x2 mix packs with range words, fp32 activations, nothing saved
    if (C != X3Cfg::C || x3_many_modes(a.K) || x3_many_modes(b.K)) return FFNO_EUNSUPPORTED;
    if (!ba->planes || !bb->planes || ba->planes_format != FFNO_PLANES_FP16X2 || bb->planes_format != FFNO_PLANES_FP16X2)
        return FFNO_EUNSUPPORTED;
    if (ba->storage != FFNO_STORE_F32 || bb->storage != FFNO_STORE_F32) return FFNO_EUNSUPPORTED;
    if (ba->resid || bb->resid || ba->spec_save || bb->spec_save || ba->accumulate || bb->accumulate) return FFNO_EINVAL;
    if (ba->tile_lines != bb->tile_lines || (ba->tile_lines != 0 && ba->tile_lines != 8 && ba->tile_lines != 16)) return FFNO_EINVAL;
    a.mix_out = reinterpret_cast<u32x4*>(ba->out), b.mix_out = reinterpret_cast<u32x4*>(bb->out);
    a.mix_scale = reinterpret_cast<float*>(a.mix_out + (size_t)a.R * 8 * 64);
    b.mix_scale = reinterpret_cast<float*>(b.mix_out + (size_t)b.R * 8 * 64);
    a.out = b.out = nullptr, a.out_amax = b.out_amax = nullptr;
    // FFNO_BRANCH_SELF_RANGE: every line scaled from its own maximum (it sits in the wave's registers when L <= 64) -- in_amax unused
    const int self_range = (ba->flags & FFNO_BRANCH_SELF_RANGE) ? 1 : 0;
    if (self_range != ((bb->flags & FFNO_BRANCH_SELF_RANGE) ? 1 : 0)) return FFNO_EINVAL;
    if (self_range && (a.L > 64 || b.L > 64)) return FFNO_EUNSUPPORTED;
    a.self_range = b.self_range = self_range;
    if (self_range) a.in_amax = b.in_amax = nullptr;
    return FFNO_OK;
}

extern "C" int ffno_spectral_x3_mix_pair(const ffno_fused_branch* ba, const ffno_fused_branch* bb, int C, int interleave,
                                         void* stream) {
    X3Args a, b;
    const int rca = mix_pair_args(a, b, ba, bb, C);
    if (rca) return rca;
    const size_t smem = sizeof(float) * 2 * max(a.L, b.L);
    hipStream_t st = (hipStream_t)stream;
    auto wg_map = [&](int NL, int n0, int n1) {
        if (n0 != n1) return 0;
        const bool square = ba->B == bb->B && ba->M == bb->M && ba->N == bb->N && ba->M == ba->N && ba->axis != bb->axis;
        if ((interleave & 2) && square && ba->B % 8 == 0 && ba->M % NL == 0) return 2 | ((ba->M / NL) << 8);
        return (interleave & 1) ? 1 : 0;
    };
    // both branches with a DFT-fragment table (ffno_spectral_x3_dft_frags; the second inference kernel needs them anyway): the forward
    // fragments come from it instead of being rebuilt from the twiddles by every wave (same values: results bit-identical)
    const bool tab = a.dft && b.dft;
    if (x3_small_tiles(a.R, b.R, ba->tile_lines)) {
        const int n0 = (a.R + 7) / 8, n1 = (b.R + 7) / 8, il = wg_map(8, n0, n1);
        if (tab)
            FFNO_LAUNCH((spectral_x3_pair_kernel<8, true, StF32, true, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, il, 0);
        else
            FFNO_LAUNCH((spectral_x3_pair_kernel<8, true, StF32, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, il, 0);
    } else {
        const int n0 = (a.R + 15) / 16, n1 = (b.R + 15) / 16, il = wg_map(16, n0, n1);
        if (tab)
            FFNO_LAUNCH((spectral_x3_pair_kernel<16, true, StF32, true, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, il, 0);
        else
            FFNO_LAUNCH((spectral_x3_pair_kernel<16, true, StF32, true>), dim3(n0 + n1), dim3(512), smem, st, a, b, n0, il, 0);
    }
    return x3_status();
}

// ---- the whole layer stack of a forward-only pass as ONE persistent launch (round 6) -----------------------------------------------
// ffno_layer_infer issues two launches per layer; their 256 workgroups start and stop together, so the chip alternates between
// "everyone loads", "everyone multiplies", "everyone stores", and every launch pays its ramp and waits for its slowest workgroup.
// Here the 8 workgroups that own one image (64 x 64: four row tiles + four column tiles of the first kernel = the eight 8-row tiles
// of the second) run K1 and K2 of ALL layers as phases of one kernel, separated by a barrier among those 8 only
// (plat::group_sync: 2.3 us); images drift apart, so loads, products and stores of different images overlap on the chip.
//   * The 8 members of a group must share an XCD (the barrier's visibility argument is the XCD's coherent L2, and the image then
//     never leaves that L2 between phases).  No placement rule is assumed: a workgroup asks the hardware which XCD it runs on
//     (plat::xcc_id) and draws a ticket from that XCD's counter; ticket / 8 = its image among the XCD's B / 8, ticket % 8 = its role.
//     One workgroup per CU (LDS) and B x 8 = the CU count make every XCD receive exactly its share; anything else (a ticket beyond
//     the share, a barrier that times out) raises the error word and the caller falls back to the two-launch layers.
//   * LDS: one dynamic window holds the first kernel's spectrum tile + twiddles and, by turns, the second kernel's column image /
//     weight fragments (spectral_x3_body<.., EXTLDS> / infer_ff_body).
//   * phases [lo, hi): phase 2 l = K1 of layer l, 2 l + 1 = K2 of layer l.  The emulator build (workgroups run one after the other) and
//     mode 1 launch one phase at a time -- the same kernel, the end of a launch as the barrier.
struct InferStackLayer {
    const u32x4* wpk_a;
    const u32x4* wpk_b;
    const u32x4* pk1;
    const float* b1;
    const u32x4* pk2;
    const float* b2;
};
constexpr int kStackMaxLayers = 32;
constexpr int kStackMaxGroups = 64;      // group counters in the sync words (a 256-CU device runs 32 groups)
struct InferStackArgs {
    X3Args a, b;           // first kernel, branch a / b (wpk per layer)
    InferArgs f;           // second kernel (packs, biases, resid / out per layer)
    float* x;              // the activations, updated in place layer by layer
    float* last_out;       // the LAST layer's feed-forward output (no residual)
    unsigned* sync;        // [0, 8): tickets per XCD; [8, 8 + kStackMaxGroups): group counters; [8 + kStackMaxGroups]: error count
    int L, T1, phase_lo, phase_hi, use_xcc;
    int groups, groups_per_xcd;      // groups of 8 workgroups in this launch; image i belongs to group i % groups
    InferStackLayer layer[kStackMaxLayers];
};

// TRACE (ffno_infer_stack mode | 2, a diagnostic): the 8 members of group 0 leave the device's constant-rate clock (plat::realtime,
// 100 MHz) behind the sync words for every phase of their first image -- [member][phase][start, body done, barrier passed, three
// marks inside the body (after its first / second / third workgroup barrier)].
// GM = members of an image's group: 8 (16-line tiles in K1, 8-row tiles in K2) or 16 (8-line / 4-row tiles: half the chain of an image
// per layer, for batches that fit the 16 groups a 256-CU device then runs)
template <int RING, bool TRACE = false, int GM = 8>
__global__ __launch_bounds__(512) FFNO_WAVES_PER_SIMD(2) void infer_stack_kernel(const InferStackArgs S) {
    static_assert(GM == 8 || GM == 16, "8 or 16 workgroups per image");
    constexpr int NL1 = GM == 8 ? 16 : 8;      // lines per workgroup in the first kernel's phases
    FFNO_DYN_SMEM(smem);
    __shared__ int who[3];
    const int B = S.f.B;
    unsigned* err = S.sync + 8 + kStackMaxGroups;
    unsigned long long* trace = reinterpret_cast<unsigned long long*>(err + 1 + ((8 + kStackMaxGroups + 1) & 1));
    if (threadIdx.x == 0) {
        if (S.use_xcc) {
            // the persistent launch: one workgroup per CU, so every XCD holds groups_per_xcd x 8 of them whatever the dispatch order
            const int xcc = plat::xcc_id() & 7;
            const int ticket = (int)atomicAdd(S.sync + xcc, 1u);
            if (ticket >= GM * S.groups_per_xcd) {      // this XCD received more workgroups than its share: no group for this one
                atomicAdd(err, 1u);
                who[0] = -1, who[1] = 0;
            } else {
                who[0] = (ticket / GM) * 8 + xcc, who[1] = ticket % GM;      // (groups 0..7 = the first group of every XCD: a small batch spreads over all L2s)
            }
        } else {
            who[0] = (int)(blockIdx.x / GM), who[1] = (int)(blockIdx.x % GM);
        }
    }
    __syncthreads();
    const int group = who[0], member = who[1];
    if (group < 0 || group >= B) return;      // (a batch below the group count: the other groups have nothing to do)
    unsigned* cnt = S.sync + 8 + group;
    unsigned arrivals = 0;
    // a group walks its images one after the other (batch > groups), every image through all its phases: no barrier between two
    // images (other lines of x, other lines of the mixed spectra), only the workgroup's own LDS changes hands
    for (int image = group; image < B; image += S.groups) {
        if (image != group) __syncthreads();
        for (int ph = S.phase_lo; ph < S.phase_hi; ++ph) {
            const int l = ph >> 1;
            int tid = (int)threadIdx.x;
            asm volatile("" : "+v"(tid));      // (opaque: per-lane values of one phase are not kept alive across the other's body)
            unsigned long long* tr = nullptr;
            if constexpr (TRACE) {
                if (group == 0 && image == 0 && member < 8) {      // (the trace words hold 8 members)
                    tr = trace + ((size_t)member * (2 * S.L) + ph) * 12;
                    if (threadIdx.x == 0) tr[0] = plat::realtime();
                }
            }
            if ((ph & 1) == 0) {
                const bool second = member >= S.T1;
                X3Args s = x3_pick_args(S.a, S.b, second);
                s.mix_out = second ? S.b.mix_out : S.a.mix_out;
                s.mix_scale = second ? S.b.mix_scale : S.a.mix_scale;
                s.self_range = S.a.self_range;
                s.dft = second ? S.b.dft : S.a.dft;      // (the forward fragment tables: DFTTAB)
                s.in = S.x;
                s.wpk = second ? S.layer[l].wpk_b : S.layer[l].wpk_a;
                if constexpr (TRACE)
                    spectral_x3_body<NL1, true, StF32, true, true, true>(s, image * S.T1 + (second ? member - S.T1 : member), 0,
                                                                          reinterpret_cast<float*>(smem), tid, tr);
                else
                    spectral_x3_body<NL1, true, StF32, true, true, true>(s, image * S.T1 + (second ? member - S.T1 : member), 0,
                                                                          reinterpret_cast<float*>(smem), tid);
            } else {
                InferArgs f = S.f;
                const bool last = l == S.L - 1;
                f.pk1 = S.layer[l].pk1, f.bias1 = S.layer[l].b1, f.pk2 = S.layer[l].pk2, f.bias2 = S.layer[l].b2;
                f.resid = last ? nullptr : S.x;
                f.out = last ? S.last_out : S.x;
                if constexpr (TRACE)
                    infer_ff_body<RING, true>(f, image, member, smem, tid, tr);
                else
                    infer_ff_body<RING, true>(f, image, member, smem, tid);
            }
            if constexpr (TRACE) {
                if (tr) {
                    __syncthreads();
                    if (threadIdx.x == 0) tr[1] = plat::realtime();
                }
            }
            if (ph + 1 < S.phase_hi) {
                arrivals += GM;
                if (!plat::group_sync(cnt, arrivals, err, &who[2])) return;
            }
            if constexpr (TRACE) {
                if (tr && threadIdx.x == 0) tr[2] = plat::realtime();
            }
        }
    }
}

// shapes the persistent form takes: what ffno_layer_infer takes on 64 x 64 images (16-line tiles of both axes and 8-row tiles: 8
// workgroups per image in either kernel), self-ranged lines, any batch.  The persistent launch always fills the device (one
// workgroup per CU: that is what gives every XCD its share), i.e. CUs / 8 groups; a smaller batch leaves groups idle, a larger one
// makes groups walk several images.
// returns 2: the persistent launch (a device whose CUs come in 8 equal XCDs of whole groups); 1: mode 1 only (one launch per
// phase); 0: not this shape
extern "C" int ffno_infer_stack_supported(int B, int M, int N, int C, int H, int K_rows, int K_cols, int n_layers) {
    if (B < 1 || !ffno_layer_infer_supported(B, M, N, C, H, K_rows, K_cols)) return 0;
    if (M != 64 || N != 64 || n_layers < 1 || n_layers > kStackMaxLayers) return 0;
    const int cus = device_cu_count();
    return (cus > 0 && cus % 64 == 0 && cus / 8 <= kStackMaxGroups) ? 2 : 1;
}
extern "C" size_t ffno_infer_stack_sync_words(int B) { return B > 0 ? (size_t)(8 + kStackMaxGroups + 1) : 0; }
// words the trace of mode | 2 needs BEHIND the sync words (one pad word, then 8 members x 2 n_layers phases x 6 stamps of 64 bits)
extern "C" size_t ffno_infer_stack_trace_words(int n_layers) {
    return n_layers > 0 ? (size_t)1 + (size_t)8 * 2 * n_layers * 12 * 2 : 0;
}

extern "C" int ffno_infer_stack(const ffno_infer_stack_desc* d, void* stream) {
    if (!d || !d->layers || !d->last_out || !d->sync || d->a.in != d->b.in || !d->a.in) return FFNO_EINVAL;
    if (!(d->a.flags & FFNO_BRANCH_SELF_RANGE)) return FFNO_EINVAL;      // (no range words travel between the phases)
    const ffno_fused_branch* row = d->a.axis == 0 ? &d->a : &d->b;
    const ffno_fused_branch* col = d->a.axis == 0 ? &d->b : &d->a;
    const int sup = ffno_infer_stack_supported(d->a.B, d->a.M, d->a.N, d->C, d->H, row->K, col->K, d->n_layers);
    if (!sup || (d->mode & ~7)) return FFNO_EUNSUPPORTED;
    const bool per_phase = (d->mode & 1) != 0, trace = (d->mode & 2) != 0;
    // members per image: 16 (8-line / 4-row tiles) while the batch fits the CUs / 16 groups the device then runs -- an image's chain per
    // layer is then about two thirds of the 8-member form's --, else 8; mode | 4 asks for 8 whatever the batch
    const int cus_ = device_cu_count();
    const int GM = (!(d->mode & 4) && d->a.B * 16 <= cus_ && cus_ % 128 == 0) ? 16 : 8;
    if (!per_phase && plat::kPersistentLaunch && sup != 2) return FFNO_EUNSUPPORTED;
    // the descriptors' `planes` are per layer: validate with the first layer's
    ffno_fused_branch ba = d->a, bb = d->b;
    ba.planes = reinterpret_cast<const float*>(d->layers[0].planes_a), bb.planes = reinterpret_cast<const float*>(d->layers[0].planes_b);
    InferStackArgs S;
    int rc = mix_pair_args(S.a, S.b, &ba, &bb, d->C);
    if (rc) return rc;
    rc = infer_build_args(S.f, &ba, &bb, d->layers[0].pk1, d->layers[0].b1, d->layers[0].pk2, d->layers[0].b2, nullptr, d->last_out,
                          d->C, d->H, nullptr, true);
    if (rc) return rc;
    if (!S.a.dft || !S.b.dft) return FFNO_EINVAL;      // (the first kernel's phases read their DFT-matrix fragments from the tables)
    S.f.R = d->a.M / GM, S.f.T = GM;      // (GM members per image in either kernel, whatever infer_rows would pick for a small batch)
    S.x = const_cast<float*>(d->a.in), S.last_out = d->last_out, S.sync = d->sync;
    const int NL1 = GM == 8 ? 16 : 8;
    S.L = d->n_layers, S.T1 = d->a.M / NL1;
    for (int l = 0; l < d->n_layers; ++l) {
        const ffno_infer_stack_layer& y = d->layers[l];
        if (!y.planes_a || !y.planes_b || !y.pk1 || !y.b1 || !y.pk2 || !y.b2) return FFNO_EINVAL;
        S.layer[l] = InferStackLayer{reinterpret_cast<const u32x4*>(y.planes_a), reinterpret_cast<const u32x4*>(y.planes_b),
                                     reinterpret_cast<const u32x4*>(y.pk1), y.b1, reinterpret_cast<const u32x4*>(y.pk2), y.b2};
    }
    const int B = S.f.B;
    const size_t lds1 = sizeof(float) * ((size_t)NL1 * X3Cfg::LSF + 2 * (size_t)max(S.a.L, S.b.L));
    const size_t smem = max(lds1, infer_lds_bytes(S.f.R, S.f.N, true));
    hipStream_t st = (hipStream_t)stream;
    // the four instances: (trace, members)
    const void* fn = GM == 16 ? (trace ? reinterpret_cast<const void*>(infer_stack_kernel<2, true, 16>)
                                       : reinterpret_cast<const void*>(infer_stack_kernel<2, false, 16>))
                              : (trace ? reinterpret_cast<const void*>(infer_stack_kernel<2, true, 8>)
                                       : reinterpret_cast<const void*>(infer_stack_kernel<2, false, 8>));
    rc = GM == 16 ? (trace ? allow_dynamic_lds(infer_stack_kernel<2, true, 16>, smem) : allow_dynamic_lds(infer_stack_kernel<2, false, 16>, smem))
                  : (trace ? allow_dynamic_lds(infer_stack_kernel<2, true, 8>, smem) : allow_dynamic_lds(infer_stack_kernel<2, false, 8>, smem));
    if (rc) return rc;
    if (hipMemsetAsync(d->sync, 0, sizeof(uint32_t) * (ffno_infer_stack_sync_words(B) + (trace ? ffno_infer_stack_trace_words(S.L) : 0)), st) !=
        hipSuccess)
        return (int)hipGetLastError();
    if constexpr (plat::kPersistentLaunch) {
        if (!per_phase) {
            const int cus = device_cu_count();
            S.phase_lo = 0, S.phase_hi = 2 * S.L, S.use_xcc = 1;
            S.groups = cus / GM, S.groups_per_xcd = cus / 8 / GM;
            (void)hipGetLastError();
            void* args[] = {&S};
            // one workgroup per CU, all resident at once (plat::launch_cooperative: the group barriers rely on it)
            const int e = plat::launch_cooperative(fn, dim3(cus), dim3(512), args, smem, st);
            return e == 0 ? x3_status() : e;
        }
    }
    S.use_xcc = 0, S.groups = B, S.groups_per_xcd = 0;      // one group per image, workgroup w = member w % GM of group w / GM
    for (int ph = 0; ph < 2 * S.L; ++ph) {
        S.phase_lo = ph, S.phase_hi = ph + 1;
        if (GM == 16)
            FFNO_LAUNCH((infer_stack_kernel<2, false, 16>), dim3(B * 16), dim3(512), smem, st, S);
        else
            FFNO_LAUNCH((infer_stack_kernel<2, false, 8>), dim3(B * 8), dim3(512), smem, st, S);
        rc = x3_status();
        if (rc) return rc;
    }
    return FFNO_OK;
}

// The two branches through the three split-bf16 STAGE kernels, as three paired launches (see ffno_spectral_staged_pair).
static int x3_stage_args(X3Stage& s, const ffno_fused_branch* b, int C) {
    if (!b || !b->in || !b->out || !b->spec_save || !b->tw || b->B <= 0 || b->M <= 0 || b->N <= 0 || b->K <= 0 ||
        (b->axis != 0 && b->axis != 1))
        return FFNO_EINVAL;
    const int L = b->axis == 0 ? b->N : b->M;
    const int R = b->axis == 0 ? b->B * b->M : b->B * b->N;
    if (b->K > L / 2 + 1) return FFNO_EMODES;
    if (!ffno_spectral_x3_staged_supported(C, b->K, L)) return FFNO_EUNSUPPORTED;
    if (b->planes && b->planes_format != FFNO_PLANES_BF16X3) return FFNO_EUNSUPPORTED;     // the stage kernels read bf16x3 packs
    s = X3Stage{b->in, b->out, b->resid, reinterpret_cast<const u32x4*>(b->planes), b->tw, R, L, b->K,
                make_linemap(b->axis, b->B, b->M, b->N, C), b->accumulate, nullptr};
    return FFNO_OK;
}

extern "C" int ffno_spectral_x3_staged_pair(const ffno_fused_branch* ba, const ffno_fused_branch* bb, float* mix_a, float* mix_b,
                                            int C, int scale_ck_fwd, int apply_ck_inv, int conj_transpose, void* stream) {
    if (!ba || !bb || !mix_a || !mix_b) return FFNO_EINVAL;
    if (ba->out == bb->out || ba->spec_save == bb->spec_save || mix_a == mix_b) return FFNO_EINVAL;
    if ((ba->planes == nullptr) != (bb->planes == nullptr)) return FFNO_EINVAL;
    X3Stage a, b;
    int rc = x3_stage_args(a, ba, C);
    if (rc) return rc;
    rc = x3_stage_args(b, bb, C);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t smem = sizeof(float) * 2 * max(a.L, b.L);
    // stage A: activations -> spectra (spec_save)
    X3Stage fa = a, fb = b;
    fa.out = ba->spec_save, fb.out = bb->spec_save;
    const int rta = (2 * a.K + 31) / 32, rtb = (2 * b.K + 31) / 32;
    int n0 = (int)(((long)a.R * rta + 3) / 4), n1 = (int)(((long)b.R * rtb + 3) / 4);
    FFNO_LAUNCH(x3_dft_fwd_pair_kernel, dim3(n0 + n1), dim3(256), smem, st, fa, fb, n0, scale_ck_fwd);
    // stage B: spectra -> mixed spectra
    const float *ya = ba->spec_save, *yb = bb->spec_save;
    if (ba->planes) {
        X3Stage ma = a, mb = b;
        ma.in = ba->spec_save, ma.out = mix_a, mb.in = bb->spec_save, mb.out = mix_b;
        const int tiles = (max(a.R, b.R) + 15) / 16;
        FFNO_LAUNCH(x3_mode_mix_pair_kernel, dim3((tiles + 3) / 4, a.K + b.K), dim3(256), 0, st, ma, mb, conj_transpose);
        ya = mix_a, yb = mix_b;
    }
    // stage C: spectra -> activations (+ accumulate / residual)
    X3Stage ia = a, ib = b;
    ia.in = ya, ib.in = yb;
    ia.out_amax = ba->out_amax, ib.out_amax = bb->out_amax;
    const int npa = (((a.L + 31) / 32) + 1) / 2, npb = (((b.L + 31) / 32) + 1) / 2;
    n0 = (int)(((long)a.R * npa + 3) / 4), n1 = (int)(((long)b.R * npb + 3) / 4);
    FFNO_LAUNCH(x3_dft_inv_pair_kernel, dim3(n0 + n1), dim3(256), smem, st, ia, ib, n0, apply_ck_inv);
    return x3_status();
}
