// Body of the second kernel of the INFERENCE layer (see infer.hip for the design): a device function over an explicit LDS window, so
// that infer.hip's launch (infer_ff_kernel) and the persistent whole-forward kernel of spectral_x3.hip (infer_stack_kernel: this body
// and the first kernel's by turns, group barriers between) run the same code.
#pragma once
#include "ffno_device.h"
#include "ffno_x3_dft.h"
#include "ffno.h"

namespace ffno {

// lines of the column branch a wave keeps in flight (phase A of infer_ff_body)
#ifndef FFNO_INFER_COL_DEPTH
#define FFNO_INFER_COL_DEPTH 3
#endif

// magnitude bound the split-fp16 feed-forward brings its input tile to (ffx.hip: kFfRangeTarget)
constexpr int kInferRangeTarget = 4;

struct InferArgs {
    const u32x4* mix_row;      // lines (b, m), transform along n: 8 fragments of 64 x 16 B per line (ffno_spectral_x3_mix_pair)
    const float* sc_row;       // accumulator -> sample factor of every such line
    const u32x4* mix_col;      // lines (b, n), transform along m
    const float* sc_col;
    const u32x4* dft_row;      // INVERSE part of the DFT-fragment table of (N, K_row): fragment (32-sample tile, k-step)
    const u32x4* dft_col;      // ... of (M, K_col)
    const float* resid;        // x (optional)
    float* out;                // x'
    const u32x4* pk1;          // ffno_ffh_pack: W1 (type 1), W2 (type 2)
    const float* bias1;
    const u32x4* pk2;
    const float* bias2;
    int B, M, N;
    int R, T;                  // rows per workgroup, row tiles per image (M = R T)
    int image_local;           // the T workgroups of an image on one XCD (B a multiple of 8)
    unsigned* out_amax;        // optional: max |out|
};

// acc += 2^11 (a b): a any in-range split pair, b bounded (the DFT matrix)
__device__ __forceinline__ f32x16 mfma_h2s_b(const Hf2& a, const Hf3& b, f32x16 c) {
    c = plat::mfma_f16_32x32x16(a.hi, b.lo, c);
    c = plat::mfma_f16_32x32x16(a.lo, b.hi, c);
    c = plat::mfma_f16_32x32x16(a.hi, b.hs, c);
    return c;
}

// the four fragments (k-step st, column tile ct) of one line's mixed spectrum
struct LineFrags {
    Hf2 y[2][2];
};
__device__ __forceinline__ LineFrags load_line(const u32x4* __restrict__ mix, long line, int lane) {
    const u32x4* p = mix + line * (8 * 64) + lane;
    LineFrags f;
    FFNO_UNROLL
    for (int st = 0; st < 2; ++st) {
        FFNO_UNROLL
        for (int ct = 0; ct < 2; ++ct) {
            f.y[st][ct].hi = p[((st * 2 + ct) * 2 + 0) * 64];
            f.y[st][ct].lo = p[((st * 2 + ct) * 2 + 1) * 64];
        }
    }
    return f;
}

// LDS image of the column branch: pixel p (row-major inside the workgroup's R x N block) at p * 256 bytes, its sixteen 16-byte
// channel chunks XOR-swizzled by `key`
__device__ __forceinline__ int swz_key(int mrel, int n) { return (n + mrel) & 15; }

// FF = false: the spectral operator alone -- out = branch_rows(x) + branch_cols(x) (SpectralConv2d.forward_fourier, grid_2d.py:51-99):
// phases A and A', then the tiles leave through the staging rows (ffno_infer_sum; the level-1 entry point ffno_spectral2d_fwd)
// (image, t): the image and the row tile of this workgroup; smem: its LDS window (ffno_infer_lds_bytes), 16-byte aligned
template <int RING = 2, bool FF = true>
__device__ __forceinline__ void infer_ff_body(const InferArgs A, const int image, const int t, char* smem, const int tid,
                                              unsigned long long* tr = nullptr) {      // tr: diagnostic stamps (infer_stack_kernel<.., TRACE>)
    constexpr int C = 64, H = 256, NCH = H / 32, KS = C / 16, CTO = C / 32, NWV = 8;
    constexpr int NF1 = NCH * KS, NF2 = NCH * CTO * 2;
    constexpr int SROW = 20;
    char* scol = smem;                                       // phases A / A'
    u32x4* w1 = reinterpret_cast<u32x4*>(smem);              // phase B (overlays the column-branch image)
    u32x4* w2 = w1 + NF1 * 2 * 64;
    float* b1s = reinterpret_cast<float*>(w2 + NF2 * 2 * 64);
    float* b2s = b1s + H;
    float* stg_all = b2s + C;
    __shared__ float rfold[NWV];
    __shared__ float bfold[NWV];

    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int N = A.N, M = A.M, R = A.R;
    const int m0 = t * R;
    const int ntiles = (R * N) >> 5;                         // <= 16: tiles wave, wave + 8 belong to this wave

    // ---------------- phase A: column branch of the workgroup's R rows, every n, into LDS ----------------
    // v_mfma_f32_16x16x32_f16: A = Y_n^T tile [16 channels x 32 (mode, part)], B = G^T [32 x 16 rows m] (R <= 16 of the 16 columns live: half the matrix work of a
    // 32-column product for the same R), D[16 channels x 16 m]: lane (m, g) holds channels 16 t + 4 g + (0..3) -- one 16-byte chunk
    {
        const int n0c = m0 & ~15, jc0 = m0 & 15;                  // 16-row tile of the inverse DFT that holds m0 .. m0 + R - 1
        const int j16 = lane & 15, g16 = lane >> 4;
        // fragment of the 32x32x16 table (32-sample tile, k-step st): lane (j32, half) <-> G[n = 32 tile + j32][kk = 16 st + 8 half + e];
        // the 16x16x32 operand wants lane (j16, g) <-> G[n = n0c + j16][kk = 8 g + e]: st = g >> 1, half = g & 1
        Hf3 G;
        {
            const int frag = (n0c >> 5) * 2 + (g16 >> 1), src = ((n0c & 31) + j16) + 32 * (g16 & 1);
            G.hi = A.dft_col[(frag * 2 + 0) * 64 + src];
            G.lo = A.dft_col[(frag * 2 + 1) * 64 + src];
            FFNO_UNROLL
            for (int w = 0; w < 4; ++w) G.hs[w] = plat::pk_mul_f16(G.hi[w], kHf2Scale);
        }
        const bool live = j16 >= jc0 && j16 < jc0 + R;
        const int mrel = j16 - jc0;
        const long l0 = (long)image * N;
        // the line's fragments as the first launch wrote them -- (st, ct), lane (j, half): Y[kk = 16 st + 8 half + e][c = 2 j + ct] --
        // read in the lane order of the 16-column product: lane (i, g) of tile t wants Y[kk = 8 g + e][c = 16 t + i], i.e. the whole
        // 16-byte slot vector of fragment (st = g >> 1, ct = i & 1), lane (j = 8 t + (i >> 1), half = g & 1)
        const int src16 = ((((g16 >> 1) * 2 + (j16 & 1)) * 2) * 64) + (j16 >> 1) + 32 * (g16 & 1);
        auto load16 = [&](long line, Hf2* y) {
            const u32x4* pp = A.mix_col + line * (8 * 64) + src16;
            FFNO_UNROLL
            for (int t4 = 0; t4 < 4; ++t4) y[t4].hi = pp[8 * t4], y[t4].lo = pp[8 * t4 + 64];
        };
        // The lines of a wave (n = wave, wave + 8, ...) come from L2, one round trip (~1 us) each, beside 12 short products: DEPTH lines
        // (and their scale factors) are in flight at any time -- with one, every iteration waited out a whole round trip (trace of the
        // persistent kernel, tools/trace_stack.py: 8.5 us for the 8 lines of a wave on an otherwise idle chip)
        constexpr int DEPTH = FFNO_INFER_COL_DEPTH;
        char* dump = scol + (long)R * N * 256 + 16 * lane;      // (infer_lds_bytes leaves 1 KiB behind the column image)
        Hf2 buf[DEPTH][4];
        float oscb[DEPTH];
        FFNO_UNROLL
        for (int d = 0; d < DEPTH; ++d) {
            const int nn = min(wave + d * NWV, N - 1);
            load16(l0 + nn, buf[d]);
            oscb[d] = A.sc_col[l0 + nn];
        }
        // one line through stage d: products, scaled rows into LDS, then (refill) the stage's next line into the same registers
        auto stage = [&](int d, int n, bool refill) {
            // the products of a line past the end run on the (finite) duplicate of line N - 1 and go to a 1-KiB dump behind the image
            // (as do the lanes without a live row): no branch; the refill is requested AFTER everything that reads the stage's
            // registers, with a clamped line index -- no copy of a stage, nothing that waits for a load it just issued
            const float osc = oscb[d];
            f32x4 acc[4];
            FFNO_UNROLL
            for (int t4 = 0; t4 < 4; ++t4) {
                acc[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[t4] = plat::mfma_f16_16x16x32(buf[d][t4].hi, G.lo, acc[t4]);
                acc[t4] = plat::mfma_f16_16x16x32(buf[d][t4].lo, G.hi, acc[t4]);
                acc[t4] = plat::mfma_f16_16x16x32(buf[d][t4].hi, G.hs, acc[t4]);
            }
            const bool on = live && n < N;
            char* base = on ? scol + (long)(mrel * N + n) * 256 : dump;
            const int key = on ? swz_key(mrel, n) : 0;
            FFNO_UNROLL
            for (int t4 = 0; t4 < 4; ++t4)
                *reinterpret_cast<float4*>(base + (on ? (((4 * t4 + g16) ^ key) << 4) : 0)) =
                    make_float4(acc[t4][0] * osc, acc[t4][1] * osc, acc[t4][2] * osc, acc[t4][3] * osc);
            if (refill) {
                const int nn = min(n + DEPTH * NWV, N - 1);      // the line DEPTH rounds ahead travels under the next rounds' products
                load16(l0 + nn, buf[d]);
                oscb[d] = A.sc_col[l0 + nn];
            }
            // (a full fence per stage: left alone the scheduler starts the products of ALL stages together, i.e. waits for every load in
            //  flight there -- one stage in flight instead of DEPTH)
            FFNO_SCHED_FENCE();
        };
        constexpr int LPW = 8;      // lines per wave of the unrolled form
        if (N <= NWV * LPW) {
            // up to 64 columns (every shape the persistent kernel takes): straight-line code, so the wait before a stage's products
            // counts exactly the loads issued after its own (in a loop the compiler waits for ALL loads at the loop header)
            FFNO_UNROLL
            for (int i = 0; i < LPW; ++i) stage(i % DEPTH, wave + i * NWV, i + DEPTH < LPW);
        } else {
            FFNO_NOUNROLL
            for (int nb = wave; nb < N; nb += DEPTH * NWV) {
                FFNO_UNROLL
                for (int d = 0; d < DEPTH; ++d) stage(d, nb + d * NWV, true);
            }
        }
    }
    // the row lines of this wave's tiles are requested on this side of the barrier
    int trow[2], tn0[2];
    bool tlive[2];
    LineFrags yrow[2];
    FFNO_UNROLL
    for (int i = 0; i < 2; ++i) {
        // (tiles w and w + 8, not 2 w and 2 w + 1: with 8 tiles -- four rows per workgroup, what a small batch gets -- every wave
        //  then has ONE tile instead of four waves with two and four with none)
        const int T_i = wave + NWV * i;
        tlive[i] = T_i < ntiles;
        const int pix0 = 32 * (tlive[i] ? T_i : 0);
        trow[i] = pix0 / N, tn0[i] = pix0 - trow[i] * N;
        yrow[i] = load_line(A.mix_row, (long)image * M + m0 + trow[i], lane);
    }
    __syncthreads();
    if (tr && tid == 0) tr[3] = plat::realtime();

    // ---------------- phase A': the tiles of this wave = row branch (registers) + column branch (LDS) ----------------
    float s[2][32];
    FFNO_UNROLL
    for (int i = 0; i < 2; ++i) {
        const int mrel = trow[i], n = tn0[i] + j;
        const char* base = scol + (long)(mrel * N + n) * 256;
        const int key = swz_key(mrel, n);
        FFNO_UNROLL
        for (int g = 0; g < 4; ++g) {
            FFNO_UNROLL
            for (int i2 = 0; i2 < 2; ++i2) {
                const int chunk = 4 * g + 2 * half + i2;
                const float4 v = *reinterpret_cast<const float4*>(base + ((chunk ^ key) << 4));
                s[i][8 * g + 4 * i2 + 0] = v.x, s[i][8 * g + 4 * i2 + 1] = v.y;
                s[i][8 * g + 4 * i2 + 2] = v.z, s[i][8 * g + 4 * i2 + 3] = v.w;
            }
        }
        Hf3 G[2];
        FFNO_UNROLL
        for (int st = 0; st < 2; ++st) G[st] = x3k_load_dft(A.dft_row, (tn0[i] >> 5) * 2 + st, lane);
        f32x16 a0 = zero16(), a1 = zero16();
        FFNO_UNROLL
        for (int st = 0; st < 2; ++st) {
            a0 = mfma_h2s_b(yrow[i].y[st][0], G[st], a0);
            a1 = mfma_h2s_b(yrow[i].y[st][1], G[st], a1);
        }
        const float osc = A.sc_row[(long)image * M + m0 + trow[i]];
        FFNO_UNROLL
        for (int g = 0; g < 4; ++g) {
            FFNO_UNROLL
            for (int q = 0; q < 4; ++q) {
                s[i][8 * g + 2 * q + 0] = __builtin_fmaf(a0[4 * g + q], osc, s[i][8 * g + 2 * q + 0]);
                s[i][8 * g + 2 * q + 1] = __builtin_fmaf(a1[4 * g + q], osc, s[i][8 * g + 2 * q + 1]);
            }
        }
    }
    if constexpr (!FF) {
        __syncthreads();      // (every wave has read its tiles: the column-branch image may be overwritten by the staging rows)
        float omax = 0.f;
        float* stg = reinterpret_cast<float*>(smem) + wave * (32 * SROW);
        const int mrow = lane >> 2, mq = lane & 3;
        float* stg_m0 = stg + mrow * SROW + 4 * mq;
        FFNO_UNROLL
        for (int i = 0; i < 2; ++i) {
            if (!tlive[i]) continue;
            const long px0 = ((long)image * M + m0 + trow[i]) * N + tn0[i];
            FFNO_UNROLL
            for (int g = 0; g < 4; ++g) {      // 16 channels at a time: operand map (pixel per lane) -> memory map (64-byte row segments)
                *reinterpret_cast<float4*>(stg + j * SROW + 8 * half) = make_float4(s[i][8 * g], s[i][8 * g + 1], s[i][8 * g + 2], s[i][8 * g + 3]);
                *reinterpret_cast<float4*>(stg + j * SROW + 8 * half + 4) =
                    make_float4(s[i][8 * g + 4], s[i][8 * g + 5], s[i][8 * g + 6], s[i][8 * g + 7]);
                plat::wave_sync();
                float4 acc[2];
                FFNO_UNROLL
                for (int ii = 0; ii < 2; ++ii) acc[ii] = *reinterpret_cast<const float4*>(stg_m0 + 16 * ii * SROW);
                plat::wave_sync();
                FFNO_UNROLL
                for (int ii = 0; ii < 2; ++ii) {
                    *reinterpret_cast<float4*>(A.out + (px0 + 16 * ii + mrow) * C + 4 * mq + 16 * g) = acc[ii];
                    omax = fmaxf(fmaxf(omax, fmaxf(fabsf(acc[ii].x), fabsf(acc[ii].y))), fmaxf(fabsf(acc[ii].z), fabsf(acc[ii].w)));
                }
            }
        }
        if (A.out_amax) range_fold(omax, rfold, NWV, A.out_amax);
        return;
    }
    // one input scale per WORKGROUP: the maximum of its 512 pixels' feed-forward inputs and of the first linear map's biases (the
    // hidden pre-activations are W1 s + b1: the scale must leave room for both terms) goes to 2^kInferRangeTarget
    {
        float mx = 0.f;
        FFNO_UNROLL
        for (int i = 0; i < 2; ++i) {
            if (!tlive[i]) continue;
            FFNO_UNROLL
            for (int q = 0; q < 32; ++q) mx = fmaxf(mx, fabsf(s[i][q]));
        }
        for (int e = tid; e < H; e += NWV * 64) mx = fmaxf(mx, fabsf(A.bias1[e]));
        FFNO_UNROLL
        for (int sh = 32; sh >= 1; sh >>= 1) mx = fmaxf(mx, __shfl_xor(mx, sh));
        if (lane == 0) bfold[wave] = mx;
    }
    // both weight packs are requested on this side of the barrier (16 x 16 B per thread: the registers the row-branch operands just
    // left): the copy's round trip runs under the reduction and the wait for the slowest wave's tile reads
    constexpr int WPT = (NF1 + NF2) * 2 * 64 / (NWV * 64);
    static_assert(WPT * NWV * 64 == (NF1 + NF2) * 2 * 64 && NF1 == NF2, "weight copy map");
    u32x4 wreg[WPT];
    FFNO_UNROLL
    for (int q = 0; q < WPT; ++q) {
        const int idx = tid + q * (NWV * 64);
        wreg[q] = idx < NF1 * 2 * 64 ? A.pk1[idx] : A.pk2[idx - NF1 * 2 * 64];
    }
    __syncthreads();
    if (tr && tid == 0) tr[4] = plat::realtime();
    float gmax = bfold[0];
    FFNO_UNROLL
    for (int w = 1; w < NWV; ++w) gmax = fmaxf(gmax, bfold[w]);
    const float gscale = range_scale(f2u(gmax), 0, kInferRangeTarget), rgscale = 1.f / gscale;

    // ---------------- the feed-forward weights take the column branch's place ----------------
    FFNO_UNROLL
    for (int q = 0; q < WPT; ++q) w1[tid + q * (NWV * 64)] = wreg[q];      // (w2 follows w1 in LDS)
    for (int e = tid; e < H; e += NWV * 64) b1s[e] = A.bias1[e] * (gscale * kHf2Unscale);      // (as the epilogue adds it: b1 g / 2^11)
    for (int e = tid; e < C; e += NWV * 64) b2s[e] = A.bias2[e];
    __syncthreads();
    if (tr && tid == 0) tr[5] = plat::realtime();

    // ---------------- phase B: Linear + ReLU + Linear + bias + residual of the wave's tiles ----------------
    // Single-accumulator products, software-pipelined over the hidden chunks.  Both GEMMs have an operand that is bounded by
    // construction -- the tile (|s| <= 2^4 after the scale) and the hidden activations divided by 2^11 (|h| < 65504 is the range
    // contract of the split) -- so each product block is three MFMAs on ONE accumulator (ffno_device.h: Hf3), no correction tile,
    // no fold.  Nothing has to agree bit for bit with a backward pass here, so iteration ch runs GEMM1 of chunk ch + 1 and GEMM2
    // of chunk ch - 1 (matrix pipe) beside the bias / ReLU / split of chunk ch (vector pipe): three independent streams instead
    // of the dependent chain GEMM1 -> epilogue -> GEMM2 of the training kernels.  The weight fragments (A operands, LDS) come
    // through a ring that is RING product blocks ahead of its consumer, pinned where it is written: left alone the scheduler puts
    // every ds_read right in front of its MFMA and the wave waits out the LDS round trip 64 times per tile (measured: 11 us of the
    // launch's 46).
    //   flat fragment order = program order, blocks of four: W1[0] | W1[1] | W1[2] W2[0] | W1[3] W2[1] | ... | W1[7] W2[5] | W2[6] | W2[7]
    constexpr int NFR = 2 * NCH * 4;
    auto frag_ptr = [&](int f) -> const u32x4* {
        const int q = f >> 2, r = f & 3;
        if (q <= 1) return w1 + ((q * 4 + r) * 2) * 64 + lane;
        if (q >= 2 * NCH - 2) return w2 + (((q - NCH) * 4 + r) * 2) * 64 + lane;      // blocks 14, 15: W2 chunks 6, 7
        if (q & 1) return w2 + (((((q - 1) >> 1) - 1) * 4 + r) * 2) * 64 + lane;
        return w1 + ((((q >> 1) + 1) * 4 + r) * 2) * 64 + lane;
    };
    float omax = 0.f;
    float* stg = stg_all + wave * (32 * SROW);
    const int mrow = lane >> 2, mq = lane & 3;
    float* stg_m0 = stg + mrow * SROW + 4 * mq;
    FFNO_UNROLL
    for (int i = 0; i < 2; ++i) {
        if (!tlive[i]) continue;
        const long px0 = ((long)image * M + m0 + trow[i]) * N + tn0[i];
        long moff[2];
        FFNO_UNROLL
        for (int ii = 0; ii < 2; ++ii) moff[ii] = (px0 + 16 * ii + mrow) * C + 4 * mq;
        Hf2 ring[RING];
        FFNO_UNROLL
        for (int f = 0; f < RING; ++f) {
            const u32x4* fp = frag_ptr(f);
            ring[f].hi = fp[0], ring[f].lo = fp[64];
        }
        Hf3 b[KS];
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st)
            b[st] = split2s_8(s[i][8 * st] * gscale, s[i][8 * st + 1] * gscale, s[i][8 * st + 2] * gscale, s[i][8 * st + 3] * gscale,
                              s[i][8 * st + 4] * gscale, s[i][8 * st + 5] * gscale, s[i][8 * st + 6] * gscale, s[i][8 * st + 7] * gscale);
        f32x16 o[CTO];
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) o[mt] = zero16();
        FFNO_SCHED_PIN_DSREAD();
        // fragment f of the flat order: out of the ring, its slot refilled with fragment f + RING
        auto take = [&](int f) {
            const Hf2 a = ring[f % RING];
            if (f + RING < NFR) {
                const u32x4* fp = frag_ptr(f + RING);
                ring[f % RING].hi = fp[0], ring[f % RING].lo = fp[64];
            }
            FFNO_SCHED_PIN_DSREAD();
            return a;
        };
        // the residual rows (memory map) are requested late in the chunk loop: early enough to arrive under its last products, late
        // enough not to hold 32 registers through all of it
        float4 rres[KS][2];
        f32x16 dcur = zero16();
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) dcur = mfma_h2s_b(take(st), b[st], dcur);
        Hf3 hprev[2];
        FFNO_UNROLL
        for (int ch = 0; ch < NCH; ++ch) {
            f32x16 dnext = dcur;
            if (ch + 1 < NCH) {
                dnext = zero16();
                FFNO_UNROLL
                for (int st = 0; st < KS; ++st) dnext = mfma_h2s_b(take(4 * (ch == 0 ? 1 : 2 * ch) + st), b[st], dnext);      // 2^11 W1[ch + 1] s
            }
            if (ch > 0) {
                FFNO_UNROLL
                for (int r = 0; r < 4; ++r) o[r >> 1] = mfma_h2s_b(take(4 * (ch == NCH - 1 ? 2 * NCH - 2 : 2 * ch + 1) + r), hprev[r & 1], o[r >> 1]);
            }
            // u = relu(W1 s + b1) / 2^11: the bounded operand of GEMM2
            float u[16];
            FFNO_UNROLL
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(&b1s[32 * ch + 8 * g + 4 * half]);
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
                FFNO_UNROLL
                for (int q = 0; q < 4; ++q) u[4 * g + q] = fmaxf(__builtin_fmaf(dcur[4 * g + q], kHf2Unscale * kHf2Unscale, bb[q]), 0.f);
            }
            hprev[0] = split2s_8(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7]);
            hprev[1] = split2s_8(u[8], u[9], u[10], u[11], u[12], u[13], u[14], u[15]);
            dcur = dnext;
            if (ch == NCH - 3 && A.resid) {
                FFNO_SCHED_PIN_VMEM();
                FFNO_UNROLL
                for (int st = 0; st < KS; ++st) {
                    FFNO_UNROLL
                    for (int ii = 0; ii < 2; ++ii) rres[st][ii] = *reinterpret_cast<const float4*>(A.resid + moff[ii] + 16 * st);
                }
                FFNO_SCHED_PIN_VMEM();
            }
        }
        FFNO_UNROLL
        for (int r = 0; r < 4; ++r) o[r >> 1] = mfma_h2s_b(take(4 * (2 * NCH - 1) + r), hprev[r & 1], o[r >> 1]);
        // output rows back through the wave's staging rows: operand map (pixel per lane) -> memory map (four lanes per 64-byte
        // row segment), 16 channels at a time
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            FFNO_UNROLL
            for (int gp = 0; gp < 2; ++gp) {
                const int blk = 2 * mt + gp;
                FFNO_UNROLL
                for (int uu = 0; uu < 2; ++uu) {
                    const int g = 2 * gp + uu;
                    *reinterpret_cast<float4*>(stg + j * SROW + 8 * uu + 4 * half) =
                        make_float4(o[mt][4 * g], o[mt][4 * g + 1], o[mt][4 * g + 2], o[mt][4 * g + 3]);
                }
                plat::wave_sync();
                float4 acc[2];
                FFNO_UNROLL
                for (int ii = 0; ii < 2; ++ii) acc[ii] = *reinterpret_cast<const float4*>(stg_m0 + 16 * ii * SROW);
                plat::wave_sync();
                FFNO_UNROLL
                for (int ii = 0; ii < 2; ++ii) {
                    const int c0 = 16 * blk + 4 * mq;
                    float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (A.resid) rr = rres[blk][ii];
                    acc[ii].x = acc[ii].x * rgscale + (b2s[c0] + rr.x);
                    acc[ii].y = acc[ii].y * rgscale + (b2s[c0 + 1] + rr.y);
                    acc[ii].z = acc[ii].z * rgscale + (b2s[c0 + 2] + rr.z);
                    acc[ii].w = acc[ii].w * rgscale + (b2s[c0 + 3] + rr.w);
                    *reinterpret_cast<float4*>(A.out + moff[ii] + 16 * blk) = acc[ii];
                    omax = fmaxf(fmaxf(omax, fmaxf(fabsf(acc[ii].x), fabsf(acc[ii].y))), fmaxf(fabsf(acc[ii].z), fabsf(acc[ii].w)));
                }
            }
        }
    }
    if (A.out_amax) range_fold(omax, rfold, NWV, A.out_amax);
}

// rows per workgroup: R N <= 512 pixels (the LDS image of the column branch), R a power of two that divides M (and 16: the live
// columns of a workgroup's column-branch product sit in one 16-column tile); fewer rows while the launch would leave CUs idle
static inline int infer_rows(int B, int M, int N) {
    int R = 1;
    while (2 * R * N <= 512 && 2 * R <= 16 && M % (2 * R) == 0) R *= 2;
    const int cus = device_cu_count();
    while (R > 1 && (long)B * (M / R) * 2 <= cus) R >>= 1;      // halve only while the halved tiles still fit ONE round of workgroups
    return R;
}


// the launch arguments of the second inference kernel from the two branch descriptors (their `out` = the mix buffers of
// ffno_spectral_x3_mix_pair, their dft_frags = the fragment tables) -- shared by ffno_infer_ff / ffno_infer_sum and ffno_infer_stack
static inline int infer_build_args(InferArgs& A, const ffno_fused_branch* ba, const ffno_fused_branch* bb, const void* pk1, const float* b1,
                                   const void* pk2, const float* b2, const float* resid, float* out, int C, int H, uint32_t* out_amax,
                                   bool ff) {
    if (!ba || !bb || !out || !ba->out || !bb->out || !ba->dft_frags || !bb->dft_frags) return FFNO_EINVAL;
    if (ff && (!pk1 || !b1 || !pk2 || !b2)) return FFNO_EINVAL;
    if (ba->axis == bb->axis || ba->B != bb->B || ba->M != bb->M || ba->N != bb->N) return FFNO_EINVAL;
    const ffno_fused_branch* row = ba->axis == 0 ? ba : bb;      // lines (b, m), transform along n
    const ffno_fused_branch* col = ba->axis == 0 ? bb : ba;      // lines (b, n), transform along m
    const int B = row->B, M = row->M, N = row->N;
    if (!ffno_layer_infer_supported(B, M, N, C, H, row->K, col->K)) return FFNO_EUNSUPPORTED;
    A.mix_row = reinterpret_cast<const u32x4*>(row->out);
    A.sc_row = reinterpret_cast<const float*>(A.mix_row + (size_t)B * M * 8 * 64);
    A.mix_col = reinterpret_cast<const u32x4*>(col->out);
    A.sc_col = reinterpret_cast<const float*>(A.mix_col + (size_t)B * N * 8 * 64);
    // inverse parts of the tables: behind the forward fragments (ffno_x3_dft.h)
    A.dft_row = reinterpret_cast<const u32x4*>(row->dft_frags) + (size_t)x3k_dft_layout(N, row->K).nfwd * 2 * 64;
    A.dft_col = reinterpret_cast<const u32x4*>(col->dft_frags) + (size_t)x3k_dft_layout(M, col->K).nfwd * 2 * 64;
    A.resid = resid, A.out = out;
    A.pk1 = reinterpret_cast<const u32x4*>(pk1), A.bias1 = b1, A.pk2 = reinterpret_cast<const u32x4*>(pk2), A.bias2 = b2;
    A.B = B, A.M = M, A.N = N;
    A.R = infer_rows(B, M, N), A.T = M / A.R;
    A.image_local = (B % 8 == 0) ? 1 : 0;
    A.out_amax = out_amax;
    return FFNO_OK;
}

// LDS window of infer_ff_body (ff: with the feed-forward; else the spectral sum alone)
static inline size_t infer_lds_bytes(int R, int N, bool ff) {
    const size_t lds_a = (size_t)R * N * 256 + 1024;      // the column-branch image + the dump of its dead lanes
    const size_t lds_b = (size_t)(2 * 64 * 1024) + (256 + 64) * sizeof(float) + 8 * 32 * 20 * sizeof(float);
    return (!ff || lds_a > lds_b) ? lds_a : lds_b;
}

}  // namespace ffno
