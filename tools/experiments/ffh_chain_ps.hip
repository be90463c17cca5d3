// NOT PART OF THE PRODUCT LIBRARY.  Negative result of round 3 (DESIGN.md "Negative results"): the pixel-split forward /
// backward-data feed-forward kernel.  A wave owns 32 pixels and ALL hidden chunks (no partial outputs, no cross-wave reduction,
// no barrier after the prologue); both weight packs (2 x 64 KiB of split-fp16 fragments) live in LDS and every MFMA's A operand
// is an LDS read through a pinned fragment ring.  Parity-green (tests/test_kernels_ffh.py through the default dispatch), no
// spills (225 VGPRs) -- and 44.9 us per layer against 45.3 for the role-split hidden-split kernel it was meant to replace: the
// launch's resources (MFMA 11.7 us, vector 10.7, per-CU texture addressing of the row-scattered 16-byte accesses ~10, LDS 7.8)
// add up instead of overlapping, exactly as in the kernel it replaces.  The single-accumulator GEMM2 (h / 2^11 as the bounded
// operand) also costs per-pixel accuracy for pixels far below the tensor maximum (test_ffh_accuracy_over_the_half_range: 2.3e-5).
// The text below is what sat in fourierflow_amd/csrc/ffx.hip between ffx_chain_rs_kernel and the weight-gradient kernels.
// ---- forward / backward-data, pixel-split waves ("ps", split-fp16) -----------------------------------------------------------
// Same operator, packs, sign-bit layout and h arithmetic as ffx_chain_kernel<SplitHf2> (GEMM1 + fold + bias + ReLU are the same
// instructions in the same order, so the ReLU decisions are the same bits); what changes is who does what.  The hidden-split
// kernels give every wave a 32-row chunk of the hidden layer: eight waves multiply the SAME 32 pixels, produce eight partial
// output tiles, and meet in LDS (64 KiB written and read per tile, one to four barriers per tile) -- measured 45 / 41 us per
// layer for 12 us of MFMA and 12-22 us of HBM time: the launch is bound by its own synchronisation.  Here a wave owns PIXELS:
// 32 of them at a time, ALL hidden chunks, start to finish:
//     rows of its tile (global -> registers, split once) -> for each hidden chunk: GEMM1 -> bias / ReLU / sign bits (or mask) ->
//     split -> GEMM2 accumulated over the chunks IN REGISTERS -> + bias, residual -> store
// No partial outputs, no cross-wave reduction, no barrier after the prologue.  What the waves share is read-only: both weight
// packs (2 x 64 KiB of split-fp16 fragments at C = 64, H = 256) are copied into LDS once per workgroup and every MFMA's A operand
// is an LDS read (128 KiB per 32-pixel tile: 8 waves x 2 tiles = 2 MiB per workgroup, two thirds of what the LDS pipe delivers
// in the MFMA time of the same tiles).
template <int C, int H, bool BWD>
__global__ __launch_bounds__(512) void ffh_chain_ps_kernel(const float* __restrict__ in, const float* __restrict__ in2,
                                                           float* sum_out, const float* resid, const u32x4* __restrict__ pk1,
                                                           const float* __restrict__ bias1, const u32x4* __restrict__ pk2,
                                                           const float* __restrict__ bias2, float* out, uint32_t* mask, int P,
                                                           const unsigned* in_amax, unsigned* out_amax) {
    using S = SplitHf2;
    constexpr int KS = C / 16, CTO = C / 32, NCH = H / 32, NWV = 8;
    constexpr int F1 = NCH * KS, F2 = NCH * CTO * 2;          // fragments of the two packs
    __shared__ __attribute__((aligned(16))) u32x4 w1l[F1 * 2 * 64];
    __shared__ __attribute__((aligned(16))) u32x4 w2l[F2 * 2 * 64];
    __shared__ __attribute__((aligned(16))) float b1s[H];
    __shared__ __attribute__((aligned(16))) float b2s[C];
    __shared__ float rfold[NWV];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ntiles = (P + 31) >> 5;
    const float gscale = in_amax ? range_scale(*in_amax, 1, kFfRangeTarget) : 1.f;
    const float rgscale = 1.f / gscale;
    float omax = 0.f;

    // prologue: ONE round trip -- every thread requests its share of both packs, the biases (and, below, its first tile) before
    // anything is written to LDS
    constexpr int W1N = F1 * 2 * 64 / 512, W2N = F2 * 2 * 64 / 512;
    static_assert(W1N * 512 == F1 * 2 * 64 && W2N * 512 == F2 * 2 * 64 && H <= 512 && C <= 512, "prologue map");
    u32x4 pw1[W1N], pw2[W2N];
    FFNO_UNROLL
    for (int u = 0; u < W1N; ++u) pw1[u] = pk1[tid + 512 * u];
    FFNO_UNROLL
    for (int u = 0; u < W2N; ++u) pw2[u] = pk2[tid + 512 * u];
    const float pb1 = (!BWD && tid < H) ? bias1[tid] : 0.f, pb2 = (!BWD && tid < C) ? bias2[tid] : 0.f;

    // this wave's tiles: (workgroup, wave) strided over the tiles.  A lane holds pixel 32 tile + j; its half-wave the channel
    // octets 8 half + 16 st of every k16 step (the B-operand slots of GEMM1), as two float4 per step.
    const int tstride = gridDim.x * NWV;
    int tile = blockIdx.x * NWV + wave;
    float4 rA[KS][2], rB[KS][2];
    auto request = [&](int t) {
        const long px = min((long)t * 32 + j, (long)P - 1);       // rows past the end re-read the last pixel (never stored)
        const float* a = in + px * C + 8 * half;
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {
            rA[st][0] = *reinterpret_cast<const float4*>(a + 16 * st);
            rA[st][1] = *reinterpret_cast<const float4*>(a + 16 * st + 4);
        }
        if (in2) {
            const float* b = in2 + px * C + 8 * half;
            FFNO_UNROLL
            for (int st = 0; st < KS; ++st) {
                rB[st][0] = *reinterpret_cast<const float4*>(b + 16 * st);
                rB[st][1] = *reinterpret_cast<const float4*>(b + 16 * st + 4);
            }
        }
    };
    if (tile < ntiles) request(tile);
    FFNO_UNROLL
    for (int u = 0; u < W1N; ++u) w1l[tid + 512 * u] = pw1[u];
    FFNO_UNROLL
    for (int u = 0; u < W2N; ++u) w2l[tid + 512 * u] = pw2[u];
    if (!BWD) {
        if (tid < H) b1s[tid] = pb1 * gscale;
        if (tid < C) b2s[tid] = pb2;
    }
    __syncthreads();

    for (; tile < ntiles; tile += tstride) {
        const long px = (long)tile * 32 + j;
        const bool live = px < P;
        // rows -> (sum of the two addends, optionally stored) -> range scale -> split: the B operands of GEMM1
        Hf2 B[KS];
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {
            float4 v0 = rA[st][0], v1 = rA[st][1];
            if (in2) {
                v0.x += rB[st][0].x, v0.y += rB[st][0].y, v0.z += rB[st][0].z, v0.w += rB[st][0].w;
                v1.x += rB[st][1].x, v1.y += rB[st][1].y, v1.z += rB[st][1].z, v1.w += rB[st][1].w;
                if (sum_out && live) {
                    float* so = sum_out + px * C + 8 * half + 16 * st;
                    *reinterpret_cast<float4*>(so) = v0;
                    *reinterpret_cast<float4*>(so + 4) = v1;
                }
            }
            B[st] = split2_8(v0.x * gscale, v0.y * gscale, v0.z * gscale, v0.w * gscale, v1.x * gscale, v1.y * gscale,
                             v1.z * gscale, v1.w * gscale);
        }
        const int nxt = tile + tstride;
        if (nxt < ntiles) request(nxt);                  // the next tile's rows travel under this tile's products
        // sign words of the tile (backward: read, one per hidden chunk; forward: written)
        uint16_t* mp = mask ? reinterpret_cast<uint16_t*>(mask) + ((long)tile * NCH) * 64 + lane : nullptr;
        uint32_t mbits[NCH / 2];
        if (BWD) {
            FFNO_UNROLL
            for (int q = 0; q < NCH; q += 2) mbits[q / 2] = (uint32_t)mp[q * 64] | ((uint32_t)mp[(q + 1) * 64] << 16);
        }
        // GEMM2 runs on ONE accumulator per output tile (2^11 x the product): its B operand is h / 2^11 -- |h| < 65504 is the
        // range contract of the split (kFfRangeTarget), so |h / 2^11| < 32 and (2^11 hi) is a valid half: the same three exact
        // products as mfma_h2 without a correction tile (32 registers that the fragment ring below needs).
        f32x16 o[CTO];
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) o[mt] = zero16();
        // The A operands (weight fragments) of a chunk come from LDS through a ring of RING fragments: fragment i + RING is
        // requested when fragment i has been handed to its MFMAs, and the request is pinned there -- left alone the scheduler puts
        // every read right in front of its use and the wave waits out each LDS round trip (measured: 46 us for the launch).
        constexpr int FPC = KS + 2 * CTO, RING = 4;      // fragments per chunk: KS of pack 1, then 2 CTO of pack 2
        auto wfrag = [&](int q, int f) {               // fragment f of chunk q (clamped past the end: a harmless re-read)
            const int qq = min(q + f / FPC, NCH - 1), ff = f % FPC;
            Hf2 a;
            if (ff < KS) {
                a.hi = w1l[((qq * KS + ff) * 2 + 0) * 64 + lane];
                a.lo = w1l[((qq * KS + ff) * 2 + 1) * 64 + lane];
            } else {
                a.hi = w2l[((qq * 2 * CTO + ff - KS) * 2 + 0) * 64 + lane];
                a.lo = w2l[((qq * 2 * CTO + ff - KS) * 2 + 1) * 64 + lane];
            }
            return a;
        };
        Hf2 ring[RING];
        FFNO_UNROLL
        for (int f = 0; f < RING; ++f) ring[f] = wfrag(0, f);
        FFNO_SCHED_PIN_DSREAD();
        FFNO_NOUNROLL
        for (int q = 0; q < NCH; ++q) {      // (a real loop: unrolled, the scheduler hoists all 128 KiB of LDS operand reads)
            // GEMM1: hidden chunk q of the 32 pixels (main + correction tile, folded: the arithmetic of ffx_chain_kernel)
            f32x16 d = zero16(), dc = zero16();
            FFNO_UNROLL
            for (int st = 0; st < KS; ++st) {
                const Hf2 a = ring[st % RING];
                ring[st % RING] = wfrag(q, st + RING);
                FFNO_SCHED_PIN_DSREAD();
                mfma_h2(a, B[st], d, dc);
            }
            S::fold(d, dc);
            if (BWD) {
                uint32_t mw = mbits[0];
                FFNO_UNROLL
                for (int u = 1; u < NCH / 2; ++u) mw = (q >> 1) == u ? mbits[u] : mw;      // (selects: no indexed register array)
                const uint32_t bits = (mw >> (16 * (q & 1))) & 0xffffu;
                FFNO_UNROLL
                for (int r = 0; r < 16; ++r) d[r] = u2f(f2u(d[r]) & bit_mask(bits, 15 - r));
            } else {
                uint32_t bits = 0;
                FFNO_UNROLL
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(&b1s[32 * q + 8 * g + 4 * half]);
                    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
                    FFNO_UNROLL
                    for (int i = 0; i < 4; ++i) {
                        const float v = fmaxf(d[4 * g + i] + bb[i], 0.f);
                        d[4 * g + i] = v;
                        bits = push_sign(bits, 0u - f2u(v));   // msb(-bits(v)) = [v > 0]; element r ends up at bit 15 - r
                    }
                }
                if (mp) mp[q * 64] = (uint16_t)bits;
            }
            // GEMM2: this chunk's contribution to the output tile, k order = D-fragment order
            Hf2 hb[2];
            u32x4 hs[2];
            FFNO_UNROLL
            for (int s2 = 0; s2 < 2; ++s2) {
                const int r0 = 8 * s2;
                hb[s2] = split2_8(d[r0] * kHf2Unscale, d[r0 + 1] * kHf2Unscale, d[r0 + 2] * kHf2Unscale, d[r0 + 3] * kHf2Unscale,
                                  d[r0 + 4] * kHf2Unscale, d[r0 + 5] * kHf2Unscale, d[r0 + 6] * kHf2Unscale, d[r0 + 7] * kHf2Unscale);
                FFNO_UNROLL
                for (int w = 0; w < 4; ++w) hs[s2][w] = plat::pk_mul_f16(hb[s2].hi[w], kHf2Scale);
            }
            FFNO_UNROLL
            for (int mt = 0; mt < CTO; ++mt) {
                FFNO_UNROLL
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int f = KS + 2 * mt + s2;
                    const Hf2 a = ring[f % RING];
                    ring[f % RING] = wfrag(q, f + RING);
                    FFNO_SCHED_PIN_DSREAD();
                    o[mt] = plat::mfma_f16_32x32x16(a.lo, hb[s2].hi, o[mt]);
                    o[mt] = plat::mfma_f16_32x32x16(a.hi, hb[s2].lo, o[mt]);
                    o[mt] = plat::mfma_f16_32x32x16(a.hi, hs[s2], o[mt]);
                }
            }
            // (the ring slot of fragment f is f mod RING inside a chunk; FPC mod RING must be 0 for the slots to line up across
            // chunks)
            static_assert(FPC % RING == 0 || RING == 1, "fragment ring period");
        }
        // output rows: lane (j, half) holds channels 32 mt + 8 g + 4 half + {0..3} of pixel j
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            FFNO_UNROLL
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * mt + 8 * g + 4 * half;
                float4 acc = make_float4(o[mt][4 * g] * rgscale, o[mt][4 * g + 1] * rgscale, o[mt][4 * g + 2] * rgscale,
                                         o[mt][4 * g + 3] * rgscale);
                if (!BWD) {
                    acc.x += b2s[c0], acc.y += b2s[c0 + 1], acc.z += b2s[c0 + 2], acc.w += b2s[c0 + 3];
                    if (resid && live) {
                        const float4 rv = *reinterpret_cast<const float4*>(resid + px * C + c0);
                        acc.x += rv.x, acc.y += rv.y, acc.z += rv.z, acc.w += rv.w;
                    }
                }
                if (live) {
                    *reinterpret_cast<float4*>(out + px * C + c0) = acc;
                    omax = fmaxf(fmaxf(omax, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
                }
            }
        }
    }
    if (out_amax) range_fold(omax, rfold, NWV, out_amax);
}

