// The INFERENCE layer of the factorized Fourier stack: one layer of FNOFactorized2DBlock.forward with nothing saved for a
// backward pass (reference fourierflow/modules/factorized_fno/grid_2d.py:42-49,51-99,169 + feedforward.py:13-19; what
// `predict`, validation and the autoregressive rollout of routines/grid_2d_markov.py run, and what the reference's own
// latency metric times: commands/train.py:134-148) in TWO launches instead of the training path's paired spectral launch +
// chain launch:
//
//   K1  ffno_spectral_x3_mix_pair (spectral_x3.hip, MIXOUT instances): both truncated forward DFTs and both per-mode channel
//       mixes -> the two MIXED spectra (K x C complex per line: half an image per axis), written as the split-fp16 operand
//       fragments K2 multiplies.  Reads x once, writes one image-equivalent.
//   K2  infer_ff_kernel (this file): both zero-padded inverse DFTs, the branch sum, Linear + ReLU + Linear, bias and residual.
//       Reads the spectra and x, writes x'.
//
// The training layer moves 7 images per layer and direction through its launches (x -> two branch images -> x'); this one 5, and
// no branch image ever exists in memory.  What makes K2 possible is a transposition that never happens:
//
//   * the feed-forward of the wave-tile kernel (ffx.hip: ffw_chain_kernel) wants its 32-pixel tile as the B operand s^T[channel][pixel]
//     with a PIXEL PER LANE.  An inverse DFT written as  s^T[c][n] = sum_kk Y^T[c][kk] G^T[kk][n]  (A = the line's mixed spectrum,
//     rows = channels; B = the DFT matrix) leaves exactly that in its accumulators: lane (n, half) holds, with the channel map of
//     the spectral kernels (row j of column tile ct <-> channel 2 j + ct), channels 16 g + 8 half + (0..7) in registers
//     {4 g .. 4 g + 3} of the two column tiles -- the k-slot order of the feed-forward's own packs.  No LDS round trip, no new
//     weight layout: the row branch (lines along n = the memory order of a tile's pixels) goes from the matrix cores straight
//     into the feed-forward.
//   * the COLUMN branch (lines along m) produces s^T[c][m] at fixed n -- pixels of one column, not of one tile.  A workgroup owns
//     R whole rows of an image (R N = 512 pixels = 16 tiles, two per wave), computes the column branch of those rows for all n
//     as  Y_n^T[c][kk] G^T[kk][m0 .. m0 + R)  on 16-column products (v_mfma_f32_16x16x32_f16)  (the image's column spectra come from L2: the R-row workgroups of an image share an
//     XCD) and passes it through LDS (128 KiB fp32, 16-byte chunks swizzled by pixel so that both the column-order writes and the
//     row-order reads spread over the banks).  Each wave then adds its tiles' row branch in registers.
//   * the 128 KiB of feed-forward weight fragments take the place of the column-branch image in LDS once every wave holds its
//     tiles (2 x 32 registers): phases, not a second buffer.
//   * the range scale of the feed-forward input is the tile's own maximum (the wave holds the whole tile): no range word of s
//     exists, none is needed.
#include "ffno_infer_body.h"

namespace ffno {

template <int RING = 2, bool FF = true>
__global__ __launch_bounds__(512) FFNO_WAVES_PER_SIMD(2) void infer_ff_kernel(const InferArgs A) {
    FFNO_DYN_SMEM(smem);
    int image, t;
    {
        const int wg = blockIdx.x;
        if (A.image_local) {
            const int per_xcd = A.B >> 3, xcd = wg & 7, slot = wg >> 3;
            image = xcd * per_xcd + slot / A.T, t = slot % A.T;
        } else {
            image = wg / A.T, t = wg - image * A.T;
        }
    }
    infer_ff_body<RING, FF>(A, image, t, smem, (int)threadIdx.x);
}

static inline int infer_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFNO_OK : (int)e;
}

}  // namespace ffno

using namespace ffno;

extern "C" int ffno_layer_infer_supported(int B, int M, int N, int C, int H, int K_rows, int K_cols) {
    if (C != 64 || H != 256 || B <= 0 || M <= 0 || N <= 0) return 0;
    if (N % 32 != 0 || N > 512 || M > 2048) return 0;
    if (K_rows < 1 || K_rows > 16 || K_cols < 1 || K_cols > 16 || K_rows > N / 2 + 1 || K_cols > M / 2 + 1) return 0;
    if (!ffno_infer_mix_bytes(C, K_rows, B * M) || !ffno_infer_mix_bytes(C, K_cols, B * N)) return 0;
    return (long)B * M * N * C * 4 < (1l << 32) ? 1 : 0;
}

static int infer_launch(const ffno_fused_branch* ba, const ffno_fused_branch* bb, const void* pk1, const float* b1, const void* pk2,
                        const float* b2, const float* resid, float* out, int C, int H, uint32_t* out_amax, bool ff, void* stream) {
    InferArgs A;
    const int rca = infer_build_args(A, ba, bb, pk1, b1, pk2, b2, resid, out, C, H, out_amax, ff);
    if (rca) return rca;
    const int B = A.B;
    const size_t smem = infer_lds_bytes(A.R, A.N, ff);
    // (ring depth 2: measured 41.6 us per launch at batch 32 against 41.7 / 42.9 / 43.5 for depths 1 / 3 / 4 -- the deeper rings spill)
    if (ff) {
        const int rc = allow_dynamic_lds(infer_ff_kernel<2, true>, smem);
        if (rc) return rc;
        FFNO_LAUNCH((infer_ff_kernel<2, true>), dim3(B * A.T), dim3(512), smem, (hipStream_t)stream, A);
    } else {
        const int rc = allow_dynamic_lds(infer_ff_kernel<2, false>, smem);
        if (rc) return rc;
        FFNO_LAUNCH((infer_ff_kernel<2, false>), dim3(B * A.T), dim3(512), smem, (hipStream_t)stream, A);
    }
    return infer_status();
}

extern "C" int ffno_infer_ff(const ffno_fused_branch* ba, const ffno_fused_branch* bb, const void* pk1, const float* b1,
                             const void* pk2, const float* b2, const float* resid, float* out, int C, int H, uint32_t* out_amax,
                             void* stream) {
    return infer_launch(ba, bb, pk1, b1, pk2, b2, resid, out, C, H, out_amax, true, stream);
}

extern "C" int ffno_infer_sum(const ffno_fused_branch* ba, const ffno_fused_branch* bb, float* out, int C, uint32_t* out_amax,
                              void* stream) {
    return infer_launch(ba, bb, nullptr, nullptr, nullptr, nullptr, nullptr, out, C, 4 * C, out_amax, false, stream);
}

extern "C" int ffno_layer_infer(const ffno_layer_infer_desc* d, void* stream) {
    if (!d) return FFNO_EINVAL;
    if (d->a.in != d->b.in) return FFNO_EINVAL;
    int rc = ffno_spectral_x3_mix_pair(&d->a, &d->b, d->C, d->interleave, stream);
    if (rc) return rc;
    return ffno_infer_ff(&d->a, &d->b, d->pk1, d->b1, d->pk2, d->b2, d->resid, d->out, d->C, d->H, d->out_amax, stream);
}
