// NOT PART OF THE PRODUCT LIBRARY.  Negative results of round 2, kept for the record (DESIGN.md "Negative results"):
//   * ffx_chain_sp_kernel  -- software-pipelined forward / backward-data chain kernel (bit-identical to ffx_chain_kernel;
//     52 us stand-alone against 54-57, within noise inside a training step; profiles/r02_ffx_sp.md)
//   * ffx_wgrad_rs_kernel  -- role-split weight-gradient kernel (108 vs 89 us; profiles/r02_ffx_schedules.md)
// together with the FFX_STAMP / FFX_ABL profiling hooks they were measured with.  The bodies below are the text that used to
// sit in fourierflow_amd/csrc/ffx.hip; they compile only when pasted back between ffx_chain_kernel and the host section of that
// file (they use its FxCfg / load_frag / stage4 helpers) and nothing in the build references this file.
// profiling hook of tools/scratch (cycle stamps around the iterations of ffx_chain_sp_kernel); empty in the product
#ifndef FFX_STAMP
#define FFX_STAMP(i)
#endif
// ablation mask of the same harness (0 in the product): 1 no GEMM1, 2 no epilogue / split, 4 no GEMM2, 8 no reduction,
// 16 no sum / split / staging of tile t + 2, 32 no global loads, 64 no partial exchange, 128 no scheduling pattern
#ifndef FFX_ABL
#define FFX_ABL 0
#endif

template <bool V>
struct FlagTag {
    static constexpr bool value = V;
};
template <int V>
struct IntTag {
    static constexpr int value = V;
};

// Stores of tiles that do not exist (pipeline prologue / epilogue of ffx_chain_sp_kernel) are redirected here instead of being
// predicated: a predicate is a branch, and a branch would cut the basic block the instruction scheduler interleaves.
__device__ float4 g_fx_sink[512];

// ---- forward / backward-data, software-pipelined schedule ---------------------------------------------------------------
// Same arithmetic as ffx_chain_kernel (bit-identical with one hidden chunk per wave, CPW = 1).
//
// Why another schedule -- measured on one SIMD (tools/ubench/mfma_valu_overlap.hip, mfma_interleave.hip, profiles/r02_ffx_sp.md):
//   * a wave whose next instruction is an MFMA that has to wait for the busy matrix pipe BLOCKS the vector issue of the other
//     wave on its SIMD (the partner gets ~1 VALU per 32-cycle MFMA slot): back-to-back MFMAs never overlap with a partner's
//     vector work, whatever the priorities or the accumulator register class;
//   * inside ONE instruction stream {1 MFMA, k independent VALU} costs max(32, ~5k) cycles (two such streams per SIMD:
//     max(64, ~2.3 * 2k)): the overlap is perfect when the vector work sits BETWEEN the wave's own MFMAs.
// ffx_chain_kernel / _rs_kernel issue 24 MFMAs back to back, so their matrix and vector phases add up (~8400 cycles per tile
// for 3072 cycles of MFMA).  Here the dependent chain GEMM1 -> epilogue -> GEMM2 is cut across tiles so that every MFMA group
// has independent vector work of OTHER tiles next to it in the same wave.  An iteration is a sequence of SLOTS, one
// mfma_x3 group (6 MFMAs per hidden chunk) plus one quantum of vector / memory work, fenced so that the order is the written one:
//
//     GEMM1(t+1), k-step 0     bias + ReLU + sign bits + split of rows 0-7 of tile t      (-> B operand of GEMM2, k-step 0)
//     GEMM1(t+1), k-step 1     ... rows 8-15                                              (-> k-step 1)
//     GEMM1(t+1), k-step 2     reduction of the partial outputs of tile t-1 + residual + store
//     GEMM1(t+1), k-step 3     sum, store and split of the raw rows of tile t+2 into the staging buffer
//     GEMM2(t), ...            row requests for tile t+3, residual rows of t+1, sign words; partial outputs to LDS
//
// One barrier per tile; values that cross iterations live in two register sets selected by the iteration parity (the loop is
// unrolled by two) so nothing is copied at the back-edge; the steady-state iteration has no predicates (loads of tiles past the
// end are clamped, their stores go to g_fx_sink); ragged pixel counts take the predicated flavour.
template <int C, int H, int CPW, bool BWD, bool FULL>
__global__ __launch_bounds__(H / (32 * CPW) * 64) void ffx_chain_sp_kernel(const float* __restrict__ in,
                                                                          const float* __restrict__ in2, float* sum_out,
                                                                          const float* resid,
                                                                          const u32x4* __restrict__ pk1,
                                                                          const float* __restrict__ bias1,
                                                                          const u32x4* __restrict__ pk2,
                                                                          const float* __restrict__ bias2, float* out,
                                                                          uint32_t* mask, int P) {
    using F = FxCfg<C, H>;
    constexpr int NCH = H / 32, NW = NCH / CPW, NT = NW * 64, KS = F::KS, CTO = F::CTO, G = F::G;
    constexpr int NV = (32 * C / 4) / NT, GPW = G / NW;
    static_assert(NW * CPW == NCH && NV >= 1 && NV * NT * 4 == 32 * C && GPW >= 1 && GPW * NW == G && NT <= 512, "maps");
    __shared__ __attribute__((aligned(16))) char sp[2][3 * F::PPLANE];
    __shared__ __attribute__((aligned(16))) float part[2][NW * G * 64 * 4];
    __shared__ __attribute__((aligned(16))) float b1s[H];
    __shared__ __attribute__((aligned(16))) float b2s[C];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ntiles = (P + 31) >> 5;
    FFX_STAMP(10);

    Bf3 A1[CPW][KS], A2[CPW][CTO][2];
    FFNO_UNROLL
    for (int ch = 0; ch < CPW; ++ch) {
        const int q = wave * CPW + ch;
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) A1[ch][st] = load_frag(pk1, q * KS + st, lane);
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            FFNO_UNROLL
            for (int s2 = 0; s2 < 2; ++s2) A2[ch][mt][s2] = load_frag(pk2, (q * CTO + mt) * 2 + s2, lane);
        }
    }
    if (!BWD) {
        for (int e = tid; e < H; e += NT) b1s[e] = bias1[e];
        for (int e = tid; e < C; e += NT) b2s[e] = bias2[e];
    }
    using Safe = FlagTag<true>;
    using Pred = FlagTag<false>;
    const int last_tile = ntiles - 1;
    float* const sink = reinterpret_cast<float*>(g_fx_sink);   // 2048 floats: any lane offset of a [32][C <= 64] tile fits

    // Addresses = uniform tile base (scalar registers) + a 32-bit lane offset that never changes (one VGPR, no 64-bit vector
    // arithmetic in the loop).  soff: float4 number f = tid + v * NT of a staged [32 px][C] tile; ooff: this lane's 4 channels
    // of output group u (D-fragment rows 8 (gi & 3) + 4 half .. + 3 of channel tile gi >> 2, pixel j).
    constexpr long TILE = 32 * C;
    auto soff = [&](int v) { return (unsigned)(((tid + v * NT) / (C / 4)) * C + 4 * ((tid + v * NT) % (C / 4))); };
    auto ooff = [&](int u) {
        const int gi = wave * GPW + u;
        return (unsigned)(j * C + 32 * (gi >> 2) + 8 * (gi & 3) + 4 * half);
    };
    // Helpers in two flavours: SAFE = no per-lane predicate (P % 32 == 0; tile indices are clamped by the caller, stores of
    // non-existent tiles are redirected to the sink by a uniform select); !SAFE = predicated on the pixel index.
    float4 nS[NV], pA[NV], pB[NV];
    auto gload_raw = [&](auto safe, int tile) {
        const float* a = in + tile * TILE;
        const float* b = (FULL || in2) ? in2 + tile * TILE : nullptr;
        FFNO_UNROLL
        for (int v = 0; v < NV; ++v) {
            const long px = (long)tile * 32 + (tid + v * NT) / (C / 4);
            if constexpr (decltype(safe)::value) {
                pA[v] = *reinterpret_cast<const float4*>(a + soff(v));
                if (FULL || in2) pB[v] = *reinterpret_cast<const float4*>(b + soff(v));
            } else {
                pA[v] = pB[v] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (px < P) {
                    pA[v] = *reinterpret_cast<const float4*>(a + soff(v));
                    if (FULL || in2) pB[v] = *reinterpret_cast<const float4*>(b + soff(v));
                }
            }
        }
    };
    // exists: (uniform) the tile is a real one -- only looked at by the SAFE flavour
    auto consume = [&](auto safe, int tile, bool exists) {
        float* dst = (decltype(safe)::value && !exists) ? sink : sum_out + tile * TILE;
        FFNO_UNROLL
        for (int v = 0; v < NV; ++v) {
            nS[v] = pA[v];
            if (FULL || in2) {
                nS[v].x += pB[v].x, nS[v].y += pB[v].y, nS[v].z += pB[v].z, nS[v].w += pB[v].w;
                if (FULL || sum_out) {
                    const long px = (long)tile * 32 + (tid + v * NT) / (C / 4);
                    if (decltype(safe)::value || px < P) *reinterpret_cast<float4*>(dst + soff(v)) = nS[v];
                }
            }
        }
    };
    auto stage = [&](int buf) {
        FFNO_UNROLL
        for (int v = 0; v < NV; ++v) {
            const int f = tid + v * NT;
            stage4(sp[buf], F::PPLANE, (f / (C / 4)) * F::PROW + (f % (C / 4)) * 8, nS[v].x, nS[v].y, nS[v].z, nS[v].w);
        }
    };
    // Values that live across iterations come in two register sets selected by the (compile-time) parity of the iteration -- the
    // loop below is unrolled by two -- so that nothing is copied at the loop back-edge: a copy of a just-requested row would
    // make the wave wait for HBM every iteration.
    //   rr      residual rows of a tile: requested late in its own iteration, added early in the next one (one set is enough)
    //   dd[q]   hidden pre-activations: GEMM1 result written in an iteration of parity 1 - q, consumed in the next one
    //   bw[q]   (backward) sign words of the tile of an iteration of parity q, requested one iteration earlier
    float4 rr[GPW];
    auto rload = [&](auto safe, int tile) {
        const long px = (long)tile * 32 + j;
        const float* r = resid + tile * TILE;
        FFNO_UNROLL
        for (int u = 0; u < GPW; ++u) {
            rr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!BWD && (FULL || resid) && (decltype(safe)::value || px < P)) rr[u] = *reinterpret_cast<const float4*>(r + ooff(u));
        }
    };
    auto reduce = [&](auto safe, auto par, int tile, bool exists) {
        constexpr int q = decltype(par)::value;      // parity of the iteration that reduces: partials in part[q ^ 1]
        const long px = (long)tile * 32 + j;
        float* dst = (decltype(safe)::value && !exists) ? sink : out + tile * TILE;
        FFNO_UNROLL
        for (int u = 0; u < GPW; ++u) {
            const int gi = wave * GPW + u;
            float4 acc = *reinterpret_cast<const float4*>(&part[q ^ 1][((0 * G + gi) * 64 + lane) * 4]);
            FFNO_UNROLL
            for (int w = 1; w < NW; ++w) {
                const float4 t = *reinterpret_cast<const float4*>(&part[q ^ 1][((w * G + gi) * 64 + lane) * 4]);
                acc.x += t.x;
                acc.y += t.y;
                acc.z += t.z;
                acc.w += t.w;
            }
            if (!BWD) {
                const float4 bv = *reinterpret_cast<const float4*>(&b2s[32 * (gi >> 2) + 8 * (gi & 3) + 4 * half]);
                acc.x += bv.x + rr[u].x;
                acc.y += bv.y + rr[u].y;
                acc.z += bv.z + rr[u].z;
                acc.w += bv.w + rr[u].w;
            }
            if (decltype(safe)::value || (px >= 0 && px < P)) *reinterpret_cast<float4*>(dst + ooff(u)) = acc;
        }
    };
    struct Hid {
        f32x16 v[CPW];
    };
    auto gemm1 = [&](int buf) {
        Hid d;
        FFNO_UNROLL
        for (int ch = 0; ch < CPW; ++ch) d.v[ch] = zero16();
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {
            const Bf3 b = lds_frag(sp[buf], F::PPLANE, j * F::PROW + 32 * st + 16 * half);
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) d.v[ch] = mfma_x3(A1[ch][st], b, d.v[ch]);
        }
        return d;
    };
    auto mask_word = [&](int tile, int ch) {
        return reinterpret_cast<uint16_t*>(mask) + (long)tile * (NCH * 64) + (unsigned)((wave * CPW + ch) * 64 + lane);
    };
    using Even = IntTag<0>;
    using Odd = IntTag<1>;

    const int t0 = blockIdx.x, gs = gridDim.x;
    uint32_t bw[2][CPW];
    Hid dd[2];
    FFNO_UNROLL
    for (int ch = 0; ch < CPW; ++ch) bw[0][ch] = bw[1][ch] = 0, dd[0].v[ch] = dd[1].v[ch] = zero16();
    FFNO_UNROLL
    for (int u = 0; u < GPW; ++u) rr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t0 < ntiles) {
        gload_raw(Pred{}, t0);
        consume(Pred{}, t0, true);
        stage(0);
        if (t0 + gs < ntiles) {
            gload_raw(Pred{}, t0 + gs);
            consume(Pred{}, t0 + gs, true);
            stage(1);
        }
        gload_raw(Pred{}, t0 + 2 * gs);
        if (BWD) {
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) bw[0][ch] = *mask_word(t0, ch);
        }
    }
    __syncthreads();
    if (t0 < ntiles) dd[0] = gemm1(0);
    FFNO_DRAIN_MEMORY();

    // one iteration of parity q: tile in sp[q] (its GEMM1 result in dd[q]), previous tile's partials in part[q ^ 1]
    auto iteration = [&](auto safe, auto par, int tile, int prev) {
        constexpr bool SAFE = decltype(safe)::value;
        constexpr int q = decltype(par)::value;
        constexpr int NSLOT = KS + 2 * CTO;          // mfma_x3 groups (per hidden chunk) of an iteration
        static_assert(KS >= 4, "quanta E0 R S E1 ride on four GEMM1 slots");
        int nt = tile + gs, nt2 = nt + gs, nt3 = nt2 + gs;
        const bool have2 = nt2 < ntiles;
        if (SAFE) nt = min(nt, last_tile), nt2 = min(nt2, last_tile), nt3 = min(nt3, last_tile);
        uint32_t bits[CPW];
        Bf3 hb[CPW][2];
        // quantum E(s2): rows 8 s2 .. 8 s2 + 7 of the hidden tile -> activation (+ sign bits) -> B operand of GEMM2's k-step s2
        auto quantum_e = [&](int s2, int ch0, int ch1) {
            FFNO_UNROLL
            for (int ch = ch0; ch < ch1; ++ch) {
                f32x16& dv = dd[q].v[ch];
                if (FFX_ABL & 2) {
                    hb[ch][s2].hi = u32x4{f2u(dv[8 * s2]), f2u(dv[8 * s2 + 3]), f2u(dv[8 * s2 + 6]), f2u(dv[8 * s2 + 7])};
                    hb[ch][s2].mid = u32x4{f2u(dv[8 * s2 + 1]), f2u(dv[8 * s2 + 4]), f2u(dv[8 * s2 + 6]), f2u(dv[8 * s2 + 7])};
                    hb[ch][s2].lo = u32x4{f2u(dv[8 * s2 + 2]), f2u(dv[8 * s2 + 5]), f2u(dv[8 * s2 + 6]), f2u(dv[8 * s2 + 7])};
                    continue;
                }
                if (BWD) {
                    FFNO_UNROLL
                    for (int r = 8 * s2; r < 8 * s2 + 8; ++r) dv[r] = u2f(f2u(dv[r]) & bit_mask(bits[ch], 15 - r));
                } else {
                    FFNO_UNROLL
                    for (int g = 2 * s2; g < 2 * s2 + 2; ++g) {
                        const float4 bv = *reinterpret_cast<const float4*>(&b1s[32 * (wave * CPW + ch) + 8 * g + 4 * half]);
                        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
                        FFNO_UNROLL
                        for (int i = 0; i < 4; ++i) {
                            const float v = fmaxf(dv[4 * g + i] + bb[i], 0.f);
                            dv[4 * g + i] = v;
                            bits[ch] = push_sign(bits[ch], 0u - f2u(v));   // msb(-bits(v)) = [v > 0]; element r ends up at bit 15 - r
                        }
                    }
                }
                hb[ch][s2] = split3_8(dv[8 * s2], dv[8 * s2 + 1], dv[8 * s2 + 2], dv[8 * s2 + 3], dv[8 * s2 + 4], dv[8 * s2 + 5],
                                      dv[8 * s2 + 6], dv[8 * s2 + 7]);
                FFNO_PIN(hb[ch][s2].hi);
                FFNO_PIN(hb[ch][s2].mid);
                FFNO_PIN(hb[ch][s2].lo);
            }
        };
        f32x16 o[CTO];
        auto write_partial = [&](int mt) {
            FFNO_UNROLL
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(&part[q][(((wave * G) + mt * 4 + g) * 64 + lane) * 4]) =
                    make_float4(o[mt][4 * g], o[mt][4 * g + 1], o[mt][4 * g + 2], o[mt][4 * g + 3]);
        };
        // pins the inputs of quantum i in FRONT of the slot's MFMAs (its outputs are pinned behind its code): the quantum's
        // instructions then sit between the two pins, free to be interleaved with the MFMAs of the slot
        // Slot -> quantum.  One chunk per wave (two waves per SIMD): E0 R S E1 L; two chunks per wave (one wave per SIMD, where a
        // vector instruction costs ~5 cycles): the E quanta are cut per chunk and spread, ~80 instructions per 12 MFMAs:
        // E0[0] E0[1] R S E1[0] E1[1] L.  E(s2) must precede slot KS + s2 * CTO, which both orders respect for KS = 4.
        constexpr int QE0 = 0, QE0B = CPW == 2 ? 1 : -1, QR = CPW == 2 ? 2 : 1, QS = QR + 1, QE1 = QS + 1,
                      QE1B = CPW == 2 ? QE1 + 1 : -1, QL = CPW == 2 ? 6 : 4, NQ = QL + 1;
        static_assert(CPW <= 2, "quanta are laid out for one or two hidden chunks per wave");
        // pins the inputs of quantum i in FRONT of the slot's MFMAs (its outputs are pinned behind its code): the quantum's
        // instructions then sit between the two pins, free to be interleaved with the MFMAs of the slot
        auto quantum_inputs = [&](int i) {
            auto& hid = dd;                 // (an asm operand alone does not capture)
            auto& rows = pA;
            if (i == QE0 || i == QE1) FFNO_PIN(hid[q].v[0]);
            if (i == QE0B || i == QE1B) FFNO_PIN(hid[q].v[CPW - 1]);
            if (i == QS) {
                FFNO_UNROLL
                for (int v = 0; v < NV; ++v) FFNO_PIN(rows[v].x);
            }
        };
        auto quantum = [&](int i) {
            if (i == QE0) quantum_e(0, 0, CPW == 2 ? 1 : CPW);
            if (i == QE0B) quantum_e(0, 1, 2);
            if (i == QE1) quantum_e(1, 0, CPW == 2 ? 1 : CPW);
            if (i == QE1B) quantum_e(1, 1, 2);
            if (i == QR && !(FFX_ABL & 8)) reduce(safe, par, prev, prev >= 0);
            if (i == QS && !(FFX_ABL & 16)) {
                consume(safe, nt2, have2);
                stage(q);                   // sp[q] held this tile: its GEMM1 ran one iteration ago
            }
            if (i == QL) {
                if (!(FFX_ABL & 32)) {
                    rload(safe, tile);      // added by the reduction early in the next iteration
                    gload_raw(safe, nt3);
                }
                if (!BWD && (FULL || mask)) {
                    FFNO_UNROLL
                    for (int ch = 0; ch < CPW; ++ch) *mask_word(tile, ch) = (uint16_t)bits[ch];
                }
            }
        };
        // inside a slot: each MFMA followed by its share of the quantum (the groups that cannot be filled are skipped)
        auto slot_pattern = [&]() {
            if (FFX_ABL & 128) return;
            FFNO_UNROLL
            for (int i = 0; i < 6 * CPW; ++i) {
                FFNO_SCHED_GROUP(0x008, 1);
                FFNO_SCHED_GROUP(0x002, CPW == 2 ? 7 : 14);
                FFNO_SCHED_GROUP(0x080, 2);
                FFNO_SCHED_GROUP(0x010, 1);
            }
        };
        FFNO_UNROLL
        for (int ch = 0; ch < CPW; ++ch) {
            bits[ch] = BWD ? bw[q][ch] : 0;
            if (BWD) bw[q ^ 1][ch] = *mask_word(min(nt, last_tile), ch);
            dd[q ^ 1].v[ch] = zero16();
        }
        // ---- GEMM1 of the NEXT tile (whatever sp[q ^ 1] holds when there is none): its B operands are read one slot ahead ----
        Bf3 bcur = lds_frag(sp[q ^ 1], F::PPLANE, j * F::PROW + 16 * half);
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {
            FFNO_SCHED_FENCE();
            Bf3 bnext = bcur;
            if (st + 1 < KS) bnext = lds_frag(sp[q ^ 1], F::PPLANE, j * F::PROW + 32 * (st + 1) + 16 * half);
            quantum_inputs(st);
            if (!(FFX_ABL & 1)) {
                FFNO_UNROLL
                for (int ch = 0; ch < CPW; ++ch) dd[q ^ 1].v[ch] = mfma_x3(A1[ch][st], bcur, dd[q ^ 1].v[ch]);
            }
            quantum(st);
            FFNO_UNROLL
            for (int ch = 0; ch < CPW; ++ch) FFNO_PIN(dd[q ^ 1].v[ch]);
            slot_pattern();
            bcur = bnext;
        }
        // ---- GEMM2 of this tile: k-step s2 needs the B operand of quantum E(s2) ----
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) o[mt] = zero16();
        FFNO_UNROLL
        for (int s2 = 0; s2 < 2; ++s2) {
            FFNO_UNROLL
            for (int mt = 0; mt < CTO; ++mt) {
                const int slot = KS + s2 * CTO + mt;
                FFNO_SCHED_FENCE();
                quantum_inputs(slot);
                FFNO_UNROLL
                for (int ch = 0; ch < CPW; ++ch) {
                    if (FFX_ABL & 4) {
                        FFNO_UNROLL
                        for (int r = 0; r < 4; ++r)
                            o[mt][4 * ch + r] += u2f(hb[ch][s2].hi[r] ^ hb[ch][s2].mid[r] ^ hb[ch][s2].lo[r]);
                        continue;
                    }
                    o[mt] = mfma_x3(A2[ch][mt][s2], hb[ch][s2], o[mt]);
                }
                if (slot < NSLOT - 1) {
                    quantum(slot);
                    FFNO_PIN(o[mt]);
                    slot_pattern();
                } else {
                    FFNO_UNROLL
                    for (int i = NSLOT - 1; i < NQ; ++i) quantum(i);
                    if (!(FFX_ABL & 64)) {
                        FFNO_UNROLL
                        for (int m2 = 0; m2 < CTO - 1; ++m2) write_partial(m2);   // complete since the slot before
                    }
                }
            }
        }
        FFNO_SCHED_FENCE();
        if (!(FFX_ABL & 64)) write_partial(CTO - 1);
    };
    // two iterations per trip (parities 0, 1); `last` = parity of the last iteration run
    int last = 1, prev = -1, tile = t0;
    auto run = [&](auto safe) {
        for (; tile < ntiles; tile += 2 * gs) {
            FFX_STAMP(0);
            iteration(safe, Even{}, tile, prev);
            FFX_STAMP(1);
            __syncthreads();
            FFX_STAMP(2);
            prev = tile;
            last = 0;
            if (tile + gs >= ntiles) break;
            iteration(safe, Odd{}, tile + gs, prev);
            FFX_STAMP(1);
            __syncthreads();
            FFX_STAMP(2);
            prev = tile + gs;
            last = 1;
        }
    };
    if ((P & 31) == 0)
        run(Safe{});
    else
        run(Pred{});
    FFX_STAMP(11);
    // partials of the last tile: written by an iteration of parity `last`, reduced with the rows of the opposite set
    if (prev >= 0) {
        if (last == 0)
            reduce(Pred{}, Odd{}, prev, true);
        else
            reduce(Pred{}, Even{}, prev, true);
    }
    FFX_STAMP(12);
}


// ---- weight gradients, role-split schedule --------------------------------------------------------------------------------
// Same arithmetic and results as ffx_wgrad_kernel (bit-identical partial slices), scheduled like ffx_chain_rs_kernel: the
// two waves of a SIMD (w and w + NW/2) stay one slot apart, a barrier after every slot, so that a matrix segment on one of
// them always runs beside a vector / LDS segment on the other:
//
//     slot            1          2          3                   4                   5          6
//     waves 0..NW/2   h^T GEMM   ReLU+split dW2 += , dh^T GEMM  mask+split          dW1 +=     stage tile t+1
//     waves NW/2..    (idle)     h^T GEMM   ReLU+split          dW2 += , dh^T GEMM  mask+split dW1 +=
//
// The first half stages the whole next tile (both layouts of s and db); its rows are requested in slot 1, five slots
// before they are split into LDS.
template <int C, int H>
__global__ __launch_bounds__((FxCfg<C, H>::NT)) void ffx_wgrad_rs_kernel(const float* __restrict__ s,
                                                                        const float* __restrict__ db,
                                                                        const u32x4* __restrict__ pk1,
                                                                        const float* __restrict__ bias1,
                                                                        const u32x4* __restrict__ pk2t,
                                                                        float* __restrict__ partial, int P) {
    using F = FxCfg<C, H>;
    constexpr int KS = F::KS, CTO = F::CTO, NW = F::NW;
    constexpr int NWA = NW / 2, NTA = NWA * 64, NVA = (32 * C / 4) / NTA;
    static_assert(NW >= 2 && NWA * 2 == NW && NVA * NTA * 4 == 32 * C && NTA % C == 0 && (NTA / C) * NVA == 8, "role split");
    constexpr int BUF = 6 * F::PPLANE + 6 * F::TPLANE;   // [sP x3][dbP x3][sT x3][dbT x3]
    constexpr int OFF_SP = 0, OFF_DP = 3 * F::PPLANE, OFF_ST = 6 * F::PPLANE, OFF_DT = 6 * F::PPLANE + 3 * F::TPLANE;
    __shared__ __attribute__((aligned(16))) char lds[2][BUF];
    __shared__ float red[F::NT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int ntiles = (P + 31) >> 5;
    const bool first = wave < NWA;

    Bf3 W1f[KS], W2f[KS];
    FFNO_UNROLL
    for (int st = 0; st < KS; ++st) {
        W1f[st] = load_frag(pk1, wave * KS + st, lane);
        W2f[st] = load_frag(pk2t, wave * KS + st, lane);
    }
    const float b1v = bias1[32 * wave + j];

    // staging, split by role so that no wave holds more than 2 * NVA rows in registers:
    //   first half : pixel-major planes (sP, dbP; map f = tid + v * NTA), requested in slot 1, split into LDS in slot 6
    //   second half: channel-major planes (sT, dbT; channel tc, pixel group tg + v * NTA / C), requested in slot 2 of the
    //                previous tile, split into LDS in slot 1 -- two slots before the first reader (slot 3)
    const int rt = first ? tid : tid - NTA;
    const int tc = rt % C, tg = rt / C;
    float4 nS[NVA], nD[NVA];
    float bs2 = 0.f;
    auto gload = [&](int tile) {
        FFNO_UNROLL
        for (int v = 0; v < NVA; ++v) {
            if (first) {
                const int f = rt + v * NTA;
                const long px = (long)tile * 32 + f / (C / 4);
                nS[v] = nD[v] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (px < P) {
                    nS[v] = *reinterpret_cast<const float4*>(s + px * C + 4 * (f % (C / 4)));
                    nD[v] = *reinterpret_cast<const float4*>(db + px * C + 4 * (f % (C / 4)));
                }
            } else {
                const long p0 = (long)tile * 32 + 4 * (tg + v * (NTA / C));
                float a[4], b[4];
                FFNO_UNROLL
                for (int i = 0; i < 4; ++i) {
                    const bool ok = p0 + i < P;
                    a[i] = ok ? s[(p0 + i) * C + tc] : 0.f;
                    b[i] = ok ? db[(p0 + i) * C + tc] : 0.f;
                }
                nS[v] = make_float4(a[0], a[1], a[2], a[3]);
                nD[v] = make_float4(b[0], b[1], b[2], b[3]);
            }
        }
    };
    auto stage = [&](int buf) {
        FFNO_UNROLL
        for (int v = 0; v < NVA; ++v) {
            if (first) {
                const int f = rt + v * NTA;
                const int offp = (f / (C / 4)) * F::PROW + (f % (C / 4)) * 8;
                stage4(lds[buf] + OFF_SP, F::PPLANE, offp, nS[v].x, nS[v].y, nS[v].z, nS[v].w);
                stage4(lds[buf] + OFF_DP, F::PPLANE, offp, nD[v].x, nD[v].y, nD[v].z, nD[v].w);
            } else {
                // pixel group grp = (s2 << 2) | (q << 1) | half  <->  local pixels 16 s2 + 8 q + 4 half + i  <->  k slot 4 q + i
                const int grp = tg + v * (NTA / C);
                const int pos = 16 * (grp & 1) + 8 * (grp >> 2) + 4 * ((grp >> 1) & 1);
                const int offt = tc * F::TROW + 2 * pos;
                stage4(lds[buf] + OFF_ST, F::TPLANE, offt, nS[v].x, nS[v].y, nS[v].z, nS[v].w);
                stage4(lds[buf] + OFF_DT, F::TPLANE, offt, nD[v].x, nD[v].y, nD[v].z, nD[v].w);
                bs2 += (nD[v].x + nD[v].y) + (nD[v].z + nD[v].w);
            }
        }
    };

    f32x16 acc1[CTO], acc2[CTO], d;
    Bf3 hb[2];
    float bs1 = 0.f;
    uint32_t bits = 0;
    FFNO_UNROLL
    for (int mt = 0; mt < CTO; ++mt) acc1[mt] = zero16(), acc2[mt] = zero16();

    auto seg_h = [&](const char* L) {          // h^T[px][hid] = s W1^T, pixels on the D rows
        d = zero16();
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {
            const Bf3 a = lds_frag(L + OFF_SP, F::PPLANE, j * F::PROW + 32 * st + 16 * half);
            d = mfma_x3(a, W1f[st], d);
        }
    };
    auto seg_relu = [&]() {
        bits = 0;
        FFNO_UNROLL
        for (int r = 0; r < 16; ++r) {
            const float v = d[r] + b1v;
            const bool pos = v > 0.f;
            d[r] = pos ? v : 0.f;
            bits |= (pos ? 1u : 0u) << r;
        }
        hb[0] = split3_8(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]);
        hb[1] = split3_8(d[8], d[9], d[10], d[11], d[12], d[13], d[14], d[15]);
    };
    auto seg_w2_dh = [&](const char* L) {      // dW2 += db^T h ; then dh^T[px][hid] = db W2
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            FFNO_UNROLL
            for (int s2 = 0; s2 < 2; ++s2) {
                const Bf3 a = lds_frag(L + OFF_DT, F::TPLANE, (32 * mt + j) * F::TROW + 32 * half + 16 * s2);
                acc2[mt] = mfma_x3(a, hb[s2], acc2[mt]);
            }
        }
        FFNO_SCHED_FENCE();
        d = zero16();
        FFNO_UNROLL
        for (int st = 0; st < KS; ++st) {
            const Bf3 a = lds_frag(L + OFF_DP, F::PPLANE, j * F::PROW + 32 * st + 16 * half);
            d = mfma_x3(a, W2f[st], d);
        }
    };
    auto seg_mask = [&]() {
        FFNO_UNROLL
        for (int r = 0; r < 16; ++r) {
            d[r] = ((bits >> r) & 1u) ? d[r] : 0.f;
            bs1 += d[r];
        }
        hb[0] = split3_8(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]);
        hb[1] = split3_8(d[8], d[9], d[10], d[11], d[12], d[13], d[14], d[15]);
    };
    auto seg_w1 = [&](const char* L) {         // dW1^T += s^T dh
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            FFNO_UNROLL
            for (int s2 = 0; s2 < 2; ++s2) {
                const Bf3 a = lds_frag(L + OFF_ST, F::TPLANE, (32 * mt + j) * F::TROW + 32 * half + 16 * s2);
                acc1[mt] = mfma_x3(a, hb[s2], acc1[mt]);
            }
        }
    };

    const int t0 = blockIdx.x, gs = gridDim.x;
    if (t0 < ntiles) {
        gload(t0);
        stage(0);
        if (!first && t0 + gs < ntiles) gload(t0 + gs);
    }
    __syncthreads();
    int buf = 0;
    for (int tile = t0; tile < ntiles; tile += gs, buf ^= 1) {
        const int nt = tile + gs;
        const char* L = lds[buf];
        if (first) {                      // slot 1
            if (nt < ntiles) gload(nt);
            seg_h(L);
        } else if (tile != t0) {
            stage(buf);                   // this tile's channel-major planes (first read in slot 3)
        }
        __syncthreads();
        if (first) {                      // slot 2
            seg_relu();
        } else {
            if (tile != t0 && nt < ntiles) gload(nt);
            seg_h(L);
        }
        __syncthreads();
        if (first) seg_w2_dh(L); else seg_relu();          // slot 3
        __syncthreads();
        if (first) seg_mask(); else seg_w2_dh(L);          // slot 4
        __syncthreads();
        if (first) seg_w1(L); else seg_mask();             // slot 5
        __syncthreads();
        if (first) {                      // slot 6
            if (nt < ntiles) stage(buf ^ 1);
        } else {
            seg_w1(L);
        }
        __syncthreads();
    }

    float* part = partial + (long)blockIdx.x * F::PART;
    float* pW1t = part;              // [c][hid]
    float* pW2 = part + H * C;       // [c][hid]
    float* pb1 = part + 2 * H * C;
    float* pb2 = pb1 + H;
    {
        const int hid = 32 * wave + j;
        FFNO_UNROLL
        for (int mt = 0; mt < CTO; ++mt) {
            FFNO_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * mt + drow(r, half);
                pW1t[c * H + hid] = acc1[mt][r];
                pW2[c * H + hid] = acc2[mt][r];
            }
        }
        const float v1 = bs1 + __shfl_xor(bs1, 32);
        if (half == 0) pb1[hid] = v1;
    }
    red[tid] = bs2;
    __syncthreads();
    if (tid < C) {
        float v = 0.f;
        for (int k = tid; k < F::NT; k += C) v += red[k];
        pb2[tid] = v;
    }
}

