// EXPERIMENT, not part of the library (round 4, last hours): the weight-gradient body with its channel-major operands made by a
// selector product on the matrix cores instead of a second staging pass.  Drop-in for ffh_wgrad_m_body in
// fourierflow_amd/csrc/ffx.hip (same signature; it was wired in behind an environment switch for the measurement).
// Result on MI355X (markov/24, batch 32, all-layers launch with one extra addend): CORRECT -- W1 / W2 / b1 slices bit-identical to
// the staged body on the emulator and on the GPU (42 kernel / trainer tests green), b2 slice equal to fp32 rounding -- and SLOWER:
// 1870-1882 us against 1413 us.  The compiler spills 29-33 registers (256 either way: 64 of weight fragments + 64 of accumulators
// + the transposition's tiles) and the chain  selector products -> v_cvt_pk -> outer products  is serial per channel tile; scheduling
// fences trade spills for exposed latency (38-58 spilled without them).  A timing-only build WITHOUT any channel-major operand
// runs the launch in 830-900 us, so the headroom is real; what this version lacks is the registers: the W2 fragments (32) fit
// in LDS now that the staging area is half as large (45 + 64 KB), and the selector products of channel tile mt + 1 belong under
// the outer products of tile mt.
// Second version, measured the same evening (not kept as source: this file + the three changes named here reproduce it): W2
// fragments in LDS (read per k-step of the dh product chain), the fold of h BEFORE the selector products, both channel tiles of g
// (then of s) transposed right behind the h (dh) product chain so that they run under its vector epilogue, a scheduling fence
// after each half transposition: 23-27 registers still spilled (7 scratch accesses per tile), the prefetch sits at 16 % of the
// loop body instead of 65-80 % -- 1720 us.  Better than the first version, still behind the staged kernel (1413 us): the 16 extra
// MFMAs + 32 v_cvt_pk per wave and tile and the four extra fragment reads cost more than the second staging pass they replace.
// ---- weight gradients, channel-major operands from the matrix cores ("t") ----------------------------------------------------
// Same operator, slices and products as ffh_wgrad_m_body.  That kernel stages every tile TWICE: pixel-major (the operand of the
// hidden-activation products) and channel-major (the operand of the two outer-product GEMMs) -- a second set of global loads,
// a second split and a second set of LDS planes, measured at a third of the launch (timing-only build without it: -35 %).  Here
// the channel-major operand is MADE from the pixel-major fragment a wave reads from LDS anyway: one product with a 0/1 selector,
//     T[px][c'] = sum_k X[px][16 u + k] Sel_u[k][c'],   Sel_u[k][c'] = [c' == 16 u + k],  u = 0, 1,
// returns the same values in the accumulator layout -- lane = channel c', registers = the pixels (r & 3) + 8 (r >> 2) + 4 half:
// exactly the k order of the outer-product GEMMs (the D-fragment order that `hb` has).  Every sum has ONE non-zero term and the
// values are halves, so the fp32 accumulator holds them exactly and v_cvt_pk_f16_f32 packs them back exactly: the operand is bit
// for bit the one the channel-major staging produced, the slices of W1 / W2 / b1 equal ffh_wgrad_m_body's bit for bit (the b2
// slice is summed in another order: fp32 rounding).  Per wave and tile: 16 more MFMAs (the pipe is 40 % busy), 32
// v_cvt_pk_f16_f32; gone: the scalar loads, splits and LDS writes of the second staging pass, half the staging area.
// (Non-finite inputs: 0 x inf in the selector product turns a row of the operand into NaN where the staged copy kept the inf.)
template <int C, int H, int NWV, class ST = StF32, int TWO = 0>
__device__ __forceinline__ void ffh_wgrad_t_body(const typename ST::T* __restrict__ s, const typename ST::T* __restrict__ db,
                                                 const u32x4* __restrict__ pk1, const float* __restrict__ bias1,
                                                 const u32x4* __restrict__ pk2t, float* __restrict__ partial, int P,
                                                 const unsigned* s_amax, const unsigned* db_amax, const int bid, const int nb,
                                                 const typename ST::T* __restrict__ s2 = nullptr,
                                                 const typename ST::T* __restrict__ db2 = nullptr) {
    using F = FxCfg<C, H, 1>;
    constexpr int KS = F::KS, CTO = F::CTO;
    static_assert(F::NW == NWV && F::NT == NWV * 64 && F::NV == 1 && 2 * CTO == KS, "one hidden chunk per wave, one float4 per thread");
    const float fscale = range_scale(*s_amax, 1, kFfRangeTarget);
    const float gscale = range_scale(*db_amax, 1, kFfRangeTarget);
    constexpr int BUF = 4 * F::PPLANE;      // [sP x2][dbP x2]
    constexpr int OFF_SP = 0, OFF_DP = 2 * F::PPLANE;
    __shared__ __attribute__((aligned(16))) char lds[2][BUF];
    __shared__ float red[F::NT * 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ntiles = (P + 31) >> 5;

    Hf2 W1f[KS], W2f[KS];
    FFNO_UNROLL
    for (int st = 0; st < KS; ++st) {
        W1f[st] = load_frag_s<SplitHf2>(pk1, wave * KS + st, lane);
        W2f[st] = load_frag_s<SplitHf2>(pk2t, wave * KS + st, lane);
    }
    const float b1v = bias1[32 * wave + j] * fscale;
    // selector fragments (B operands): lane (n = j, half), slot e <-> k = 8 half + e; 1.0 (half 0x3C00) where n == 16 u + k
    u32x4 sel[2];
    FFNO_UNROLL
    for (int u = 0; u < 2; ++u) {
        FFNO_UNROLL
        for (int q = 0; q < 4; ++q) {
            const unsigned lo = (j == 16 * u + 8 * half + 2 * q) ? 0x3C00u : 0u;
            const unsigned hi = (j == 16 * u + 8 * half + 2 * q + 1) ? 0x3C00u : 0u;
            sel[u][q] = lo | (hi << 16);
        }
    }
    auto mma3 = [&](const Hf2& a, const Hf2& b, f32x16& acc) {
        u32x4 hs;
        FFNO_UNROLL
        for (int q = 0; q < 4; ++q) hs[q] = plat::pk_mul_f16(a.hi[q], kHf2Scale);
        acc = plat::mfma_f16_32x32x16(a.lo, b.hi, acc);
        acc = plat::mfma_f16_32x32x16(a.hi, b.lo, acc);
        acc = plat::mfma_f16_32x32x16(hs, b.hi, acc);
    };
    // channel tile mt of a staged tensor, as the two k-steps (a0: pixels of accumulator registers 0..7, a1: 8..15) of the
    // channel-major operand, from its pixel-major fragments x0 / x1 (channels 32 mt + 0..15 / + 16..31)
    auto transpose = [&](const Hf2& x0, const Hf2& x1, Hf2& a0, Hf2& a1) {
        f32x16 t = zero16();
        t = plat::mfma_f16_32x32x16(x0.hi, sel[0], t);
        t = plat::mfma_f16_32x32x16(x1.hi, sel[1], t);
        FFNO_UNROLL
        for (int q = 0; q < 4; ++q) a0.hi[q] = plat::pack_f16(t[2 * q], t[2 * q + 1]), a1.hi[q] = plat::pack_f16(t[8 + 2 * q], t[9 + 2 * q]);
        t = zero16();
        t = plat::mfma_f16_32x32x16(x0.lo, sel[0], t);
        t = plat::mfma_f16_32x32x16(x1.lo, sel[1], t);
        FFNO_UNROLL
        for (int q = 0; q < 4; ++q) a0.lo[q] = plat::pack_f16(t[2 * q], t[2 * q + 1]), a1.lo[q] = plat::pack_f16(t[8 + 2 * q], t[9 + 2 * q]);
    };

    // staging: pixel-major only (float4 number tid of the [32][C] tile); branch-free as in ffh_wgrad_m_body
    typename ST::Raw4 nSP, nDP, mSP, mDP;      // (m*: the second addends, TWO only)
    const int prow = tid / (C / 4), pcol = tid % (C / 4);
    float bs2v[4] = {0.f, 0.f, 0.f, 0.f};
    auto gload = [&](int tile_) {
        const int tile = min(tile_, ntiles - 1);
        const int rows = min(P - tile * 32, 32);
        const long off = (long)tile * (32 * C) + min(prow, rows - 1) * C + 4 * pcol;
        nSP = ST::ldr4(s + off);
        nDP = ST::ldr4(db + off);
        if constexpr (TWO >= 1) mSP = ST::ldr4(s2 + off);
        if constexpr (TWO == 2) mDP = ST::ldr4(db2 + off);
    };
    auto stage = [&](int buf, int tile_) {
        const int rows = tile_ < ntiles ? min(P - tile_ * 32, 32) : 0;
        const float mp = prow < rows ? 1.f : 0.f;
        const float fs = fscale * mp, gs = gscale * mp;
        float4 sp = ST::w4(nSP), dp = ST::w4(nDP);
        if constexpr (TWO >= 1) {
            const float4 s2v = ST::w4(mSP);
            sp.x += s2v.x, sp.y += s2v.y, sp.z += s2v.z, sp.w += s2v.w;
            sp = st_rnd4<ST>(sp);
        }
        if constexpr (TWO == 2) {
            const float4 d2v = ST::w4(mDP);
            dp.x += d2v.x, dp.y += d2v.y, dp.z += d2v.z, dp.w += d2v.w;
            dp = st_rnd4<ST>(dp);
        }
        sp.x *= fs, sp.y *= fs, sp.z *= fs, sp.w *= fs;
        dp.x *= gs, dp.y *= gs, dp.z *= gs, dp.w *= gs;
        const int offp = prow * F::PROW + pcol * 8;
        stage4_s<SplitHf2>(lds[buf] + OFF_SP, F::PPLANE, offp, sp.x, sp.y, sp.z, sp.w);
        stage4_s<SplitHf2>(lds[buf] + OFF_DP, F::PPLANE, offp, dp.x, dp.y, dp.z, dp.w);
        bs2v[0] += dp.x, bs2v[1] += dp.y, bs2v[2] += dp.z, bs2v[3] += dp.w;
    };

    f32x16 acc1[CTO], acc2[CTO];          // 2^11 x the two weight-gradient slices
    float bs1 = 0.f;
    FFNO_UNROLL
    for (int mt = 0; mt < CTO; ++mt) acc1[mt] = zero16(), acc2[mt] = zero16();

    gload(bid);
    stage(0, bid);
    gload(bid + nb);
    __syncthreads();
    int buf = 0;
    for (int tile = bid; tile < ntiles; tile += nb, buf ^= 1) {
        const int nt = tile + nb;
        const char* L = lds[buf];
        auto frag = [&](int off, int q) { return lds_frag_s<SplitHf2>(L + off, F::PPLANE, j * F::PROW + 32 * q + 16 * half); };
        Hf2 ring[2];
        ring[0] = frag(OFF_SP, 0), ring[1] = frag(OFF_SP, 1);
        FFNO_SCHED_PIN_DSREAD();
        stage(buf ^ 1, nt);
        gload(nt + nb);
        FFNO_SCHED_PIN_VMEM();
        // h^T[px][hid] = relu(s W1^T + b1): main + correction tile, as in the forward kernel
        f32x16 d = zero16(), dc = zero16();
        uint32_t bits = 0;
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {
            const Hf2 a = ring[st & 1];
            ring[st & 1] = st + 2 < KS ? frag(OFF_SP, st + 2) : frag(OFF_DP, st + 2 - KS);
            FFNO_SCHED_PIN_DSREAD();
            mfma_h2(a, W1f[st], d, dc);
        }
        Hf2 hb[2];
        SplitHf2::fold(d, dc);
        FFNO_UNROLL
        for (int r = 0; r < 16; ++r) {
            const float v = d[r] + b1v;
            const bool pos = v > 0.f;
            d[r] = pos ? v : 0.f;
            bits |= (pos ? 1u : 0u) << r;
        }
        hb[0] = split2_8(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]);
        hb[1] = split2_8(d[8], d[9], d[10], d[11], d[12], d[13], d[14], d[15]);
        // dW2 slice: (g^T)[c][px] x h^T[px][hid], the channel-major operand made from the pixel-major fragments of g
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            // (fragments 2 mt, 2 mt + 1 of g; read where they are used: the registers of a second pair in flight are not there)
            const Hf2 x0 = mt == 0 ? ring[0] : frag(OFF_DP, 2 * mt), x1 = mt == 0 ? ring[1] : frag(OFF_DP, 2 * mt + 1);
            Hf2 a0, a1;
            transpose(x0, x1, a0, a1);
            mma3(a0, hb[0], acc2[mt]);
            mma3(a1, hb[1], acc2[mt]);
            FFNO_SCHED_FENCE();
        }
        ring[0] = frag(OFF_DP, 0), ring[1] = frag(OFF_DP, 1);      // (for the product chain below)
        FFNO_SCHED_PIN_DSREAD();
        // dh^T[px][hid] = (g W2) * [h > 0]   (d = 2^11 x the product)
        d = zero16();
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {
            const Hf2 a = ring[st & 1];
            ring[st & 1] = st + 2 < KS ? frag(OFF_DP, st + 2) : frag(OFF_SP, st + 2 - KS);
            FFNO_SCHED_PIN_DSREAD();
            mma3(a, W2f[st], d);
        }
        FFNO_UNROLL
        for (int r = 0; r < 16; ++r) {
            d[r] = ((bits >> r) & 1u) ? d[r] * kHf2Unscale : 0.f;
            bs1 += d[r];
        }
        hb[0] = split2_8(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]);
        hb[1] = split2_8(d[8], d[9], d[10], d[11], d[12], d[13], d[14], d[15]);
        // dW1^T slice: (s^T)[c][px] x dh^T[px][hid]
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            const Hf2 x0 = mt == 0 ? ring[0] : frag(OFF_SP, 2 * mt), x1 = mt == 0 ? ring[1] : frag(OFF_SP, 2 * mt + 1);
            Hf2 a0, a1;
            transpose(x0, x1, a0, a1);
            mma3(a0, hb[0], acc1[mt]);
            mma3(a1, hb[1], acc1[mt]);
            FFNO_SCHED_FENCE();
        }
        __syncthreads();
    }

    const float rg = 1.f / gscale, rf = 1.f / fscale;
    float* part = partial + (long)bid * F::PART;
    float* pW1t = part;              // [c][hid]
    float* pW2 = part + H * C;       // [c][hid]
    float* pb1 = part + 2 * H * C;
    float* pb2 = pb1 + H;
    const int hid = 32 * wave + j;
    FFNO_UNROLL
    for (int mt = 0; mt < CTO; ++mt) {
        FFNO_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int c = 32 * mt + drow(r, half);
            pW1t[c * H + hid] = acc1[mt][r] * kHf2Unscale * rf * rg;
            pW2[c * H + hid] = acc2[mt][r] * kHf2Unscale * rf * rg;
        }
    }
    const float v1 = bs1 + __shfl_xor(bs1, 32);
    if (half == 0) pb1[hid] = v1 * rg;
    // b2 slice: thread t holds the sums of its pixel row's channels 4 (t % (C / 4)) + (0..3)
    FFNO_UNROLL
    for (int i = 0; i < 4; ++i) red[tid * 4 + i] = bs2v[i];
    __syncthreads();
    if (tid < C) {
        float v = 0.f;
        for (int k = tid / 4; k < F::NT; k += C / 4) v += red[k * 4 + (tid & 3)];
        pb2[tid] = v * rg;
    }
}

