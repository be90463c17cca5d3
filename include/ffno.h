/*
 * ffno.h -- C ABI of the MI355X (gfx950) F-FNO hot-path library  (libffno_hip.so)
 *
 * Drop-in boundary for the spectral-layer stack of alasdairtran/fourierflow.  The reference has no
 * FFI of its own (it is pure PyTorch); each entry point below replaces the *sequence of stock torch
 * ops* cited next to it, and the Python host in fourierflow_amd/ re-exposes them behind the
 * reference's `fourierflow.modules` classes (same ctor kwargs, parameter names and forward contract).
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless named *_host.
 *   - the library borrows pointers for the duration of the enqueue, never allocates user-visible
 *     memory, and only enqueues work on `stream` (a hipStream_t passed as void*); no hidden syncs.
 *   - every function returns 0 on success, a negative FFNO_E* code on bad arguments / unsupported
 *     shapes, or a positive hipError_t if a launch failed.  Nothing throws across the ABI.
 *   - activations are channels-last fp32: x[B][M][N][C]; "pixels" P = B*M*N.
 *   - supported widths: C in {32, 64}; hidden H = factor*C in {64, 128, 256} (H % 64 == 0).
 *   - re-entrant: the library keeps no mutable state (only per-device constants such as the CU count, cached
 *     after the first query).  Everything that tunes a launch is an argument of that call; everything a
 *     kernel needs from an earlier kernel travels through caller-owned device memory.
 *
 * Range words (the fp16x2 kernels, ffno_ffh_* and FFNO_PLANES_FP16X2)
 *   A range word is a caller-owned uint32 in DEVICE memory holding the bit pattern of max|x| over a tensor
 *   (non-negative floats order like their bit patterns).  Producers fold their output maximum into the word
 *   they are given (`out_amax`: atomic max, one per workgroup) -- the caller zeroes it before the first
 *   producer of that tensor.  A kernel that cuts an operand into fp16 planes takes the operand's word
 *   (`in_amax`) and derives, on the device, the power of two that brings the operand into the half format's
 *   range; its results are divided by it again (exact).  So any finite fp32 tensor is a valid operand -- no
 *   host round trip, no assumption about magnitudes.  NULL in_amax = "the data is known to lie in
 *   [2^-12, 2^15]" (no scaling); NULL out_amax = nothing is recorded.  ffno_amax() folds any tensor.
 *
 * Storage formats (the "bf16 storage twins" of the hot path; SURVEY 8(b) level 1)
 *   The reference is `precision: 32` and fp32 tensors are the parity path.  The entry points of the markov / torus hot path
 *   (width 64, <= 16 modes, 2-layer feed-forward: ffno_spectral_x3[_pair], ffno_ffh_fwd2 / _bwd_data2 / _bwd_weights_partial,
 *   ffno_layer_fwd / _bwd, ffno_lift_*_bf16, ffno_head_*_bf16) also take their ACTIVATION tensors -- layer inputs and
 *   outputs, branch outputs, the saved feed-forward input, and the gradients of those -- as bf16 (FFNO_STORE_BF16 in the
 *   descriptor / options of the call; the pointers are then read as uint16_t*, same element counts).  Values are widened as
 *   they are loaded and rounded to nearest even where they are stored; weights, spectra, ReLU sign bits, reductions and every
 *   product in between are the fp32-grade arithmetic of the fp32 path, so  twin(x) == bf16(fp32 path(float(x)))  bit for bit.
 *   Half the activation bytes of a training step; the price is the bf16 rounding of the stored tensors (forward ~3e-3
 *   relative against the fp32 path over 24 layers: a throughput variant with its own tolerance, never the parity path).
 *
 * Spectrum layout (internal but part of the ABI because callers own the workspaces):
 *   spec[k][r][ri][c]   k = mode (0..K-1), r = line index, ri = 0 real / 1 imag, c = channel
 *   lines: axis 0 (transform along N, fourier_weight[0]): r = b*M + m  (R = B*M lines of length N)
 *          axis 1 (transform along M, fourier_weight[1]): r = b*N + n  (R = B*N lines of length M)
 */
#ifndef FFNO_H_
#define FFNO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFNO_OK 0
#define FFNO_EINVAL (-1)       /* null pointer / non-positive size */
#define FFNO_EUNSUPPORTED (-2) /* shape outside the compiled template set */
#define FFNO_EMODES (-3)       /* modes > L/2+1 (the reference fails with an einsum size error here) */

#define FFNO_STORE_F32 0       /* activation tensors are float            */
#define FFNO_STORE_BF16 1      /* activation tensors are bf16 (uint16_t)  */

#define FFNO_MODE_FULL 0       /* grid_2d.py:64-68  */
#define FFNO_MODE_LOWPASS 1    /* grid_2d.py:69-70  */
#define FFNO_MODE_NOFOURIER 2  /* grid_2d.py:44-45  */

/* library / build identification ("gfx950", or "emu" for the CPU test build) */
const char* ffno_build_target(void);
/* ABI generation of this header: bumped whenever a struct layout or an argument list below changes incompatibly (round 3 changed
 * ffno_fused_branch, the layer descriptors and inserted range-word arguments before `stream` without bumping it: an older
 * library under newer host code would have read shifted arguments).  ffno_abi_version() returns the value the LIBRARY was built
 * with; a caller compares it with the FFNO_ABI_VERSION it was compiled against before the first compute call (the Python host
 * does: fourierflow_amd/_lib.py refuses a mismatch). */
#define FFNO_ABI_VERSION 7
int ffno_abi_version(void);

/* word[0] = max(word[0], bits(max |x[i]|)): folds a tensor into a range word (see "Range words" above) */
int ffno_amax(const float* x, size_t n, uint32_t* word, void* stream);
/* the same for n tensors in one launch (descs is a DEVICE array; max_n = the largest descs[i].n): every tensor is folded into word[0] */
typedef struct ffno_amax_desc {
    const float* x;
    size_t n;
} ffno_amax_desc;
int ffno_amax_batched(const ffno_amax_desc* descs_dev, int n, size_t max_n, uint32_t* word, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Twiddle table for a transform of length L (host helper, double precision -> fp32):
 *   tw_host[j] = cos(2*pi*j/L)/sqrt(L), tw_host[L + j] = sin(2*pi*j/L)/sqrt(L),  j in [0, L)
 * norm='ortho' of torch.fft.rfft / irfft (grid_2d.py:58,72,76,90) is folded into the table.
 * The caller uploads it once per L and passes the device copy as `tw`.
 * --------------------------------------------------------------------------------------------- */
int ffno_twiddle_fill_host(float* tw_host, int L);

/* ---------------------------------------------------------------------------------------------
 * Stage A -- truncated forward real DFT along one axis.
 * Replaces: rearrange 'b m n i -> b i m n' + torch.fft.rfft(x, dim, norm='ortho')[..., :K]
 *           (grid_2d.py:52,58,67 / :76,85).  Never materialises the L/2+1-bin spectrum.
 * scale_ck = 1 multiplies row k by c_k (1 for DC/Nyquist, 2 otherwise): the adjoint of the
 *           zero-padded irfft, used by the backward pass.
 * --------------------------------------------------------------------------------------------- */
int ffno_dft_fwd(const float* x, float* spec, const float* tw, int B, int M, int N, int C, int K,
                 int axis, int scale_ck, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fourier-weight repack: w[I][O][K][2] (the reference parameter layout, grid_2d.py:26) ->
 * mode-major planes wp[K][2][I][O] and the transposed copy wpt[K][2][O][I] (for the backward mix).
 * --------------------------------------------------------------------------------------------- */
int ffno_fw_pack(const float* w, float* wp, float* wpt, int C, int K, void* stream);
/* the same for n weight tensors in one launch (descs is a DEVICE array); real = 1: real [I][O][K] weights (DCT operators,
 * as ffno_fw_pack_real) */
typedef struct ffno_fwpack_desc {
    const float* w;
    float* wp;
    float* wpt;
    int32_t K, real;
} ffno_fwpack_desc;
int ffno_fw_pack_batched(const ffno_fwpack_desc* descs, int n, int C, int max_K, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stage B -- per-mode complex channel mix on R lines.
 * Replaces: torch.einsum("bixy,ioy->boxy" | "bixy,iox->boxy", X[..:K], view_as_complex(W))
 *           (grid_2d.py:65-68, 83-86).
 *   conj_transpose = 0:  Y[k][r][o] = sum_i X[k][r][i] * W[i][o][k]          (pass planes = wp)
 *   conj_transpose = 1:  dX[k][r][i] = sum_o dY[k][r][o] * conj(W[i][o][k])  (pass planes = wpt)
 * --------------------------------------------------------------------------------------------- */
int ffno_mode_mix(const float* spec_in, const float* planes, float* spec_out, int R, int C, int K,
                  int conj_transpose, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stage C -- zero-padded inverse real DFT along one axis, with fused accumulate / residual.
 * Replaces: new_zeros + slice-assign + torch.fft.irfft(out_ft, n=L, dim, norm='ortho') and the
 *           branch sum `xx + xy` (grid_2d.py:61,65,72,79,83,90,94).  imag(DC) is ignored exactly as
 *           the C2R transform does.
 *   out = (accumulate ? out : 0) + (resid ? resid : 0) + iDFT(spec)
 *   apply_ck = 1: forward semantics (bins k>=1 counted twice); 0: adjoint of the truncated rfft.
 * --------------------------------------------------------------------------------------------- */
int ffno_dft_inv(const float* spec, float* out, const float* resid, const float* tw, int B, int M,
                 int N, int C, int K, int axis, int apply_ck, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fourier-weight gradient:  dW[i][o][k] = sum_r conj(X[k][r][i]) * dY[k][r][o]
 * (autograd of the einsum above).  Two steps so the result is deterministic:
 *   partial[s][k][2][I][O]  (+)=  sum over the s-th slice of the lines        (beta: 0 overwrite, 1 add)
 *   gw[I][O][K][2]          (+)=  sum_s partial[s]                            (accumulate)
 * The contraction can run over `nlayers` layers at once (weights shared by all layers: grid_2d.py:125-141):
 * layer l's spectra start at spec_x + l*layer_stride_x / spec_dy + l*layer_stride_dy (floats).
 * --------------------------------------------------------------------------------------------- */
int ffno_fw_grad_partial(const float* spec_x, const float* spec_dy, float* partial, int R, int C,
                         int K, int nsplit, int beta, int nlayers, size_t layer_stride_x,
                         size_t layer_stride_dy, void* stream);
int ffno_fw_grad_reduce(const float* partial, float* gw, int C, int K, int nsplit, int accumulate,
                        void* stream);
/* The partial step on split-fp16 operands (round 6; three fp16 MFMAs per product instead of six bf16 ones -- the bf16 launch runs on
 * the 1400 W power cap: profiles/r06_power.md).  x_words / d_words: DEVICE arrays with the range words of the tensors whose spectra
 * spec_x / spec_dy are -- the layer inputs and the feed-forward data gradients -- one per layer of the contraction (`nlayers` for the
 * shared launch: ONE power-of-two pair from their maxima scales the whole launch; `n` for the multi launch: one pair per problem);
 * L = the transform length (|spectrum| <= 2 sqrt(L) max |tensor|).  NULL words: the bf16x3 kernels (any range). */
int ffno_fw_grad_partial_h2(const float* spec_x, const float* spec_dy, float* partial, int R, int C, int K, int nsplit, int beta,
                            int nlayers, size_t layer_stride_x, size_t layer_stride_dy, const uint32_t* x_words,
                            const uint32_t* d_words, int L, void* stream);
int ffno_fw_grad_partial_multi_h2(const float* spec_x, const float* spec_dy, float* partial, int R, int C, int K, int nsplit, int n,
                                  size_t stride_x, size_t stride_dy, size_t stride_p, const uint32_t* x_words,
                                  const uint32_t* d_words, int L, void* stream);
/* the two steps for n independent weight tensors in one launch each (per-layer Fourier weights of an unshared model):
 * problem z reads spec_x + z*stride_x / spec_dy + z*stride_dy, its slices live at partial + z*stride_p and are reduced into
 * gws[z] (DEVICE array of n pointers); real = 1: real [I][O][K] outputs (DCT operators). */
int ffno_fw_grad_partial_multi(const float* spec_x, const float* spec_dy, float* partial, int R, int C, int K,
                               int nsplit, int n, size_t stride_x, size_t stride_dy, size_t stride_p,
                               void* stream);
int ffno_fw_grad_reduce_multi(const float* partial, float* const* gws, int n, int C, int K, int nsplit,
                              size_t stride_p, int accumulate, int real, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused branch (stage A -> B -> C in ONE launch, the spectrum tile stays in LDS; fast path):
 *   out = (accumulate ? out : 0) + (resid ? resid : 0) + iDFT_axis( mix( DFT_axis(in) ) )
 *   planes == NULL skips the mix (mode 'low-pass');  spec_save (optional, spec[k][r][ri][c]) receives
 *   the stage-A spectrum, which the Fourier-weight gradient needs.
 *   forward : scale_ck_fwd = 0, apply_ck_inv = 1, conj_transpose = 0, planes = wp
 *   backward: scale_ck_fwd = 1, apply_ck_inv = 0, conj_transpose = 1, planes = wpt
 * ffno_spectral_fused_supported() says whether (C, K, L) fits (8 lines x 2K x C floats of LDS);
 * otherwise use the three stage kernels above.
 * --------------------------------------------------------------------------------------------- */
int ffno_spectral_fused_supported(int C, int K, int L);
int ffno_spectral_fused(const float* in, float* out, const float* resid, float* spec_save,
                        const float* planes, const float* tw, int B, int M, int N, int C, int K,
                        int axis, int scale_ck_fwd, int apply_ck_inv, int conj_transpose,
                        int accumulate, uint32_t* out_amax /* optional range word of out */, void* stream);
/* Two branches in one launch (e.g. the two axes of a layer, or two of the three views of the 3-D operator -- each branch
 * carries its own [B][M][N] view of the buffers): the workgroups of both are resident together (two per CU) and hide each
 * other's memory phases.  The branches must write different `out` buffers (accumulate / resid are per branch); scale /
 * conj flags are common (both forward or both adjoint). */
typedef struct ffno_fused_branch {
    const float* in;
    float* out;
    const float* resid;     /* optional */
    float* spec_save;       /* optional */
    const float* planes;    /* optional (low-pass) */
    const float* tw;        /* twiddle table of this branch's axis length */
    int32_t B, M, N;        /* view of this branch */
    int32_t K, axis, accumulate;
    /* ffno_spectral_x3* only (the fp32 kernels ignore them; zero = the previous behaviour): */
    int32_t planes_format;  /* FFNO_PLANES_BF16X3 / FFNO_PLANES_FP16X2: how ffno_spectral_x3_pack wrote `planes` */
    int32_t tile_lines;     /* lines per workgroup of the fused x3 kernel: 0 = the library chooses (16; 8 while the launch fits one
                               round of workgroups -- one per CU; 4 lines with two waves per line, FFNO_X3_TILE_LATENCY, while those fit
                               one round of workgroups: rollout at batch 1), or 8 / 16 / FFNO_X3_TILE_LATENCY; results are
                               bit-identical */
    const uint32_t* in_amax; /* range word of `in` (FP16X2 planes: the spectrum tile is held scaled by the power of two derived
                                from it -- |X| <= 2 sqrt(L) max|in| goes to 2^15 -- and the outputs are divided again); NULL = 1 */
    uint32_t* out_amax;      /* optional: receives max |out| of what this branch stores (the fused kernels of both families and
                                the x3 stage kernels; the fp32 stage kernels ignore it: fold their output with ffno_amax) */
    int32_t storage;         /* FFNO_STORE_F32 (0) / FFNO_STORE_BF16: format of in / out / resid ("Storage formats" above;
                                ffno_spectral_x3[_pair]: every fused split kernel with FP16X2 planes, the K <= 16 kernel also
                                without planes; spec_save stays fp32) */
    int32_t flags;           /* FFNO_BRANCH_* bits; 0 = none (the field was padding before round 6: zero keeps every earlier behaviour) */
    const void* dft_frags;   /* optional (FP16X2 planes; read by every fused split kernel since round 6: the many-mode kernel,
                                17..64 modes, the 4-line latency kernel and the width-32 kernel as before, and now the 16- / 8-line
                                kernel of the <= 16-mode shapes too -- in a paired launch when BOTH branches carry one): the
                                DFT-matrix fragments of this branch's (L, K, direction = scale_ck_fwd / apply_ck_inv of the call)
                                as ffno_spectral_x3_dft_frags wrote them -- the kernel then loads them instead of rebuilding them
                                from the twiddle table for every line (bit-identical results); NULL = build on the fly */
} ffno_fused_branch;
#define FFNO_PLANES_BF16X3 0
#define FFNO_PLANES_FP16X2 1
#define FFNO_PLANES_FP16X2_M16 2   /* fp16x2 fragments in the order of v_mfma_f32_16x16x32_f16: the many-mode kernel (C = 64, 17..64
                                      modes; its 4-line tile has 8 live mix rows) then runs 16-row products -- half the matrix time and a
                                      third of the vector work of its mix.  Needs dft_frags; same bytes as FFNO_PLANES_FP16X2 */
/* ffno_spectral_x3_mix_pair only (both branches alike, axis lengths <= 64): every LINE is brought into the half format's range by
 * the power of two of its own maximum -- the wave that transforms a line holds all of it before the first product -- instead of
 * the tensor's range word: in_amax is not read, so nobody has to produce it (no ffno_amax pass over x in front of
 * ffno_spectral2d_fwd, no out_amax atomics in the launch that wrote x).  At least as tight as the tensor-wide scale. */
#define FFNO_BRANCH_SELF_RANGE 1
#define FFNO_X3_TILE_LATENCY 1
#define FFNO_X3_TILE_LATENCY_SPLIT 2 /* the latency tile with TWO workgroups per tile, one per output-channel parity (fp16x2 packs
                                      * with a mix only; tile_lines = 0 picks it while the doubled launch fits one round) */
/* DFT-matrix fragment table of the fused kernels that give a wave ONE line (C = 64, FP16X2 planes: the many-mode kernel, 17..64
 * modes, and the latency kernel FFNO_X3_TILE_LATENCY of the <= 16-mode shapes) for one axis length L, mode
 * count K and direction (scale_ck_fwd / apply_ck_inv as the launch will pass them): every (row tile, 64-sample chunk, k-step)
 * fragment of the truncated forward DFT matrix and every (32-sample tile, k-step) fragment of the zero-padded inverse, already
 * split into fp16 planes in MFMA lane order.  Built once per (L, K, direction) -- rfft / irfft twiddles do not change
 * (grid_2d.py:58,72,76,90) -- and shared by all lines, layers and steps.  `tw` = the device twiddle table of length L. */
size_t ffno_spectral_x3_dft_frags_bytes(int L, int K);
int ffno_spectral_x3_dft_frags(const float* tw, int L, int K, int scale_ck_fwd, int apply_ck_inv, void* frags, void* stream);
int ffno_spectral_fused_pair(const ffno_fused_branch* a, const ffno_fused_branch* b, int C, int scale_ck_fwd,
                             int apply_ck_inv, int conj_transpose, void* stream);
/* ---------------------------------------------------------------------------------------------
 * The fused branch on the bf16 matrix cores at fp32 accuracy ("bf16x3": both operands of every product -- DFT matrices,
 * activations, spectra, weights -- cut exactly into three bf16 planes, six v_mfma_f32_32x32x16_bf16 per product block).
 * Same operator and flags as ffno_spectral_fused[_pair] (grid_2d.py:58-72 / :76-90 and the adjoint); a workgroup owns 16
 * lines.  The branch descriptor is ffno_fused_branch with `planes` pointing at PACKED weights:
 *   ffno_spectral_x3_pack: planes[k][re|im][i][o] (forward: wp, adjoint: wpt of ffno_fw_pack) -> fragment order, split;
 *   ffno_spectral_x3_pack_bytes(C, K) bytes per packed set.  descs is a DEVICE array, one launch for all sets.
 * Supported by the fused kernels (ffno_spectral_x3_supported, L <= 2048): C = 64 with K <= 16 (16- / 8-line tiles, or the 4-line
 * latency tiles for small launches), C = 64 with 17..64 modes (4-line tiles), C = 32 with K <= 16 (16 lines, two side by side per
 * wave); the staged variant below takes C = 64, K <= 32; the fp32-MFMA kernels above cover the rest.
 * interleave (pair, equal workgroup counts): bit 0 = even workgroups run branch a, odd ones branch b; bit 1 = image-local map
 * where the shapes allow it (both branches are the two axes of the same square images, batch a multiple of 8, whole tiles per
 * image): the workgroups that read one image run on one XCD, so the second branch finds the image in that XCD's L2.
 * --------------------------------------------------------------------------------------------- */
typedef struct ffno_x3pack_desc {
    const float* planes; /* [K][2][C][C] */
    void* dst;           /* ffno_spectral_x3_pack_bytes(C, K) bytes, 16-B aligned */
    int32_t K;
    int32_t format;      /* FFNO_PLANES_BF16X3, or FFNO_PLANES_FP16X2: the per-mode channel mix of the fused kernels then
                            runs on three fp16 MFMAs per product block instead of six bf16 ones (ffno_ffh_* has the number
                            format; the DFT phases keep the bf16 split).  The staged kernels take BF16X3 packs only. */
} ffno_x3pack_desc;
int ffno_spectral_x3_supported(int C, int K, int L);
size_t ffno_spectral_x3_pack_bytes(int C, int K);
int ffno_spectral_x3_pack(const ffno_x3pack_desc* descs_dev, int n, int C, int max_K, void* stream);
int ffno_spectral_x3(const ffno_fused_branch* br, int C, int scale_ck_fwd, int apply_ck_inv, int conj_transpose,
                     void* stream);
int ffno_spectral_x3_pair(const ffno_fused_branch* a, const ffno_fused_branch* b, int C, int scale_ck_fwd,
                          int apply_ck_inv, int conj_transpose, int interleave, void* stream);
/* The same arithmetic for the shapes outside the fused tile (17..32 modes: 256 x 256 grids with 32 modes, the 32-mode axis
 * of the plasticity / airfoil meshes): the two branches through three paired STAGE launches, spectra through HBM, exactly like
 * ffno_spectral_staged_pair (same arguments; `planes` = packed sets; spec_save must be set; mix_a / mix_b scratch spectra).
 * Supported: C = 64, K <= 32, L <= 2048 (ffno_spectral_x3_staged_supported). */
int ffno_spectral_x3_staged_supported(int C, int K, int L);
int ffno_spectral_x3_staged_pair(const ffno_fused_branch* a, const ffno_fused_branch* b, float* mix_a, float* mix_b, int C,
                                 int scale_ck_fwd, int apply_ck_inv, int conj_transpose, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Layer level (SURVEY 8b "signature level 2"): one call enqueues a whole factorized Fourier layer -- SpectralConv2d.forward
 * plus the residual add (grid_2d.py:42-49,169) -- or its backward, from device-resident operands.
 *   fwd:  a.out <- branch_a(a.in), b.out <- branch_b(b.in)   (one paired launch; a.in == b.in == x)
 *         out = resid + FF(a.out + b.out);  s_sum (optional, may alias a.out) keeps the FF input for the backward pass,
 *         mask (optional) the ReLU sign bits.
 *   bwd:  ds = FF^T(g [+ g2])  (the sum is stored to g_sum when g2 is given);  partial <- feed-forward weight-gradient slices
 *         (ffno_ffx_bwd_weights_partial: reduce them with ffno_ffx_bwd_weights_reduce[_batched]; partial == NULL: skipped --
 *         the caller runs ffno_ffh_bwd_weights_partial_multi over all layers after the pass);  then the adjoint branches
 *         a, b (a.in == b.in == ds; a.resid = the residual gradient), which also save the dY spectra for the Fourier-weight
 *         gradient through a.spec_save / b.spec_save.
 * branch_kernel selects the fused branch kernel: FFNO_BRANCH_X3 (planes = packed split-bf16 sets) or FFNO_BRANCH_FUSED
 * (planes = fp32 planes).  ff_kernel selects the feed-forward family; weight packs as for ffno_ffx_* / ffno_ffh_*: pk1 / pk2
 * forward, pk1b / pk2b backward.
 * --------------------------------------------------------------------------------------------- */
#define FFNO_BRANCH_FUSED 0
#define FFNO_BRANCH_X3 1
#define FFNO_FF_BF16X3 0
#define FFNO_FF_FP16X2 1
/* (a.storage == b.storage is the format of EVERY activation tensor of the layer call: the branches' in / out / resid and the
 *  feed-forward's s_sum / resid / out, g / g2 / g_sum / ds / s) */
typedef struct ffno_layer_fwd_desc {
    ffno_fused_branch a, b;
    int32_t branch_kernel, interleave;
    const void* pk1;
    const float* b1;
    const void* pk2;
    const float* b2;
    float* s_sum;
    const float* resid;
    float* out;
    void* mask;
    int32_t P, C, H;
    int32_t ff_kernel; /* FFNO_FF_BF16X3 (packs of ffno_ffx_pack) or FFNO_FF_FP16X2 (packs of ffno_ffh_pack) */
    int32_t ff_schedule, ff_max_workgroups, pad0_, pad1_; /* ffno_ff_opts.schedule / .max_workgroups of the feed-forward launch (0: the library's choice) */
    /* range words: a.in_amax = b.in_amax = word of x; a.out_amax = b.out_amax = word of the branch outputs, which the
     * feed-forward then reads as its in_amax; out_amax receives max |out| (the next layer's x word).  All optional. */
    uint32_t* out_amax;
} ffno_layer_fwd_desc;
typedef struct ffno_layer_bwd_desc {
    ffno_fused_branch a, b;
    int32_t branch_kernel, interleave;
    const float* g;
    const float* g2;
    float* g_sum;
    const void* mask;
    const void* pk1b;
    const void* pk2b;
    float* ds;
    const float* s;
    const void* pk1;
    const float* b1;
    float* partial;
    int32_t nsplit, P, C, H;
    int32_t ff_kernel, ff_schedule, ff_max_workgroups, pad_; /* ff_schedule / ff_max_workgroups: as in ffno_layer_fwd_desc */
    /* range words (all optional): g_amax = word of g / g2 (both folded into it by their producers) -- read by the data- and
     * the weight-gradient kernel; s_amax = word of the addends of s (the forward's a.out_amax); ds_amax receives max |ds| and
     * is what the caller also passes as a.in_amax / b.in_amax of the adjoint branches; a.out_amax / b.out_amax receive the
     * maxima of the two gradient buffers: the earlier layer's g_amax. */
    const uint32_t* g_amax;
    const uint32_t* s_amax;
    uint32_t* ds_amax;
} ffno_layer_bwd_desc;
int ffno_layer_fwd(const ffno_layer_fwd_desc* d, void* stream);
int ffno_layer_bwd(const ffno_layer_bwd_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * INFERENCE layer (round 6): SpectralConv2d.forward + the residual add (grid_2d.py:42-49,51-99,169; feedforward.py:13-19)
 * with NOTHING saved for a backward pass -- what predict / validation / the autoregressive rollout
 * (routines/grid_2d_markov.py:130-144,195-326) and the reference's latency metric (commands/train.py:134-148) run -- in two
 * launches that never write a branch image:
 *   ffno_spectral_x3_mix_pair  both truncated forward DFTs + both per-mode channel mixes -> the two MIXED spectra (K x C
 *                              complex per line), stored as the split-fp16 operand fragments the second launch multiplies:
 *                              `out` of each branch = its mix buffer, ffno_infer_mix_bytes(C, K, lines) bytes (lines = B M for
 *                              axis 0, B N for axis 1); resid / spec_save / accumulate must be unset; planes = FFNO_PLANES_FP16X2
 *                              packs; in_amax = the range word of x.
 *   ffno_infer_ff              both zero-padded inverse DFTs (transposed: channels x pixels, straight into the feed-forward's
 *                              operand registers), branch sum, Linear + ReLU + Linear, bias, residual -> out; the same branch
 *                              descriptors (their `out` = the mix buffers just written, their dft_frags = the tables of
 *                              ffno_spectral_x3_dft_frags(tw, L, K, 0, 1, ...): REQUIRED here), packs of ffno_ffh_pack (pk1: W1
 *                              type 1, pk2: W2 type 2), resid = x (optional), out_amax (optional) receives max |out|.
 *   ffno_layer_infer           = the two, one call per layer.
 * 5 image passes per layer instead of the 7 of ffno_layer_fwd (x -> 2 half-image spectra -> x'); results agree with
 * ffno_layer_fwd to fp32 rounding (other summation order), not bit for bit.
 * Supported (ffno_layer_infer_supported): C = 64, H = 256, <= 16 modes per axis, N a multiple of 32 up to 512, fp32 storage.
 * --------------------------------------------------------------------------------------------- */
typedef struct ffno_layer_infer_desc {
    ffno_fused_branch a, b;  /* the two axes over the same x (a.in == b.in, a.axis != b.axis, same B / M / N) */
    int32_t interleave, pad_;
    const void* pk1;
    const float* b1;
    const void* pk2;
    const float* b2;
    const float* resid;
    float* out;
    int32_t C, H;
    uint32_t* out_amax;
} ffno_layer_infer_desc;
size_t ffno_infer_mix_bytes(int C, int K, int lines);
int ffno_layer_infer_supported(int B, int M, int N, int C, int H, int K_rows, int K_cols);
int ffno_spectral_x3_mix_pair(const ffno_fused_branch* a, const ffno_fused_branch* b, int C, int interleave, void* stream);
int ffno_infer_ff(const ffno_fused_branch* a, const ffno_fused_branch* b, const void* pk1, const float* b1, const void* pk2,
                  const float* b2, const float* resid, float* out, int C, int H, uint32_t* out_amax, void* stream);
/* the second launch WITHOUT the feed-forward: out = the sum of the two branches' zero-padded inverse DFTs = SpectralConv2d.
 * forward_fourier (grid_2d.py:51-99) of the x that ffno_spectral_x3_mix_pair transformed (ffno_spectral2d_fwd runs this pair for
 * the shapes ffno_layer_infer_supported takes) */
int ffno_infer_sum(const ffno_fused_branch* a, const ffno_fused_branch* b, float* out, int C, uint32_t* out_amax, void* stream);
int ffno_layer_infer(const ffno_layer_infer_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The whole layer stack of a forward-only pass in ONE persistent launch (round 6): the loop
 *     for layer in layers: x = x + FeedForward(SpectralConv2d.forward_fourier(x))        (grid_2d.py:159-177, without the saves)
 * that a host otherwise issues as n_layers x ffno_layer_infer.  The 8 workgroups that own one 64 x 64 image run the two kernels
 * of every layer as phases separated by a barrier among those 8 only (they share an XCD: its coherent L2 carries x and the mixed
 * spectra from phase to phase), so images drift apart and loads, products and stores of different images overlap -- no launch
 * boundary, no chip-wide lock step.  Results are bit-identical to the ffno_layer_infer loop (same code, same order).
 *   a, b      as for ffno_layer_infer, with flags = FFNO_BRANCH_SELF_RANGE (required), in = x: the lifted features, UPDATED IN PLACE
 *             layer by layer; `planes` are ignored (per layer below)
 *   layers    HOST array of n_layers entries (copied into the kernel arguments): the FP16X2 packs of both axes, the ffno_ffh_pack
 *             packs and biases of that layer
 *   last_out  receives the LAST layer's feed-forward output (no residual: what the head reads, grid_2d.py:169-177)
 *   sync      ffno_infer_stack_sync_words(B) device words, zeroed by the call; the LAST word != 0 after the launch = a workgroup found
 *             no group or a barrier timed out (~0.1 s): the result is INVALID -- run the ffno_layer_infer loop instead
 *   mode      0: one persistent launch (one workgroup per CU, all resident); 1: the same kernel, one launch per phase (2 n_layers launches);
 *             | 4: 8 workgroups per image whatever the batch (default: 16 -- 8-line / 4-row tiles -- while the batch fits the CUs / 16
 *             groups the device then runs, 8 -- 16-line / 8-row tiles -- above; the results do not depend on it);
 *             | 2 (with mode 0, a diagnostic): the 8 workgroups of group 0 leave the device's 100 MHz clock for every phase of their
 *             first image -- start, body done, barrier passed, three marks inside the body -- as 64-bit stamps [member][phase][6] behind the sync words (+ one pad
 *             word): `sync` then holds ffno_infer_stack_sync_words(B) + ffno_infer_stack_trace_words(n_layers) words
 * ffno_infer_stack_supported: 0 = not this shape (what ffno_layer_infer takes, 64 x 64 images, any batch, n_layers <= 32);
 * 2 = mode 0 and mode 1 (a device whose CUs come as 8 XCDs of whole groups: 256 on MI355X.  The persistent launch always runs one
 * workgroup per CU -- that is what hands every XCD its share --, i.e. CUs / 8 groups: a batch below that leaves groups idle, a
 * larger one makes a group walk images g, g + CUs / 8, ...); 1 = mode 1 only.
 * --------------------------------------------------------------------------------------------- */
typedef struct ffno_infer_stack_layer {
    const void* planes_a;
    const void* planes_b;
    const void* pk1;
    const float* b1;
    const void* pk2;
    const float* b2;
} ffno_infer_stack_layer;
typedef struct ffno_infer_stack_desc {
    ffno_fused_branch a, b;
    const ffno_infer_stack_layer* layers;
    int32_t n_layers, C, H, mode;
    float* last_out;
    uint32_t* sync;
} ffno_infer_stack_desc;
int ffno_infer_stack_supported(int B, int M, int N, int C, int H, int K_rows, int K_cols, int n_layers);
size_t ffno_infer_stack_sync_words(int B);
size_t ffno_infer_stack_trace_words(int n_layers);
int ffno_infer_stack(const ffno_infer_stack_desc* d, void* stream);

/* The same two branches through the three STAGE kernels, as three paired launches (dft_fwd x2 | mode_mix x2 | dft_inv x2),
 * for the shapes the fused kernel does not take (K > 16 at C = 64: 256 x 256 grids with 32 / 64 modes have only 512 lines
 * per axis at batch 2 -- one launch per axis cannot fill the chip).  spec_save must be set in both branches (scratch when
 * nothing is kept); mix_a / mix_b receive the mixed spectra (same size); planes == NULL in both = mode 'low-pass'. */
int ffno_spectral_staged_pair(const ffno_fused_branch* a, const ffno_fused_branch* b, float* mix_a,
                              float* mix_b, int C, int scale_ck_fwd, int apply_ck_inv,
                              int conj_transpose, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Operator level: SpectralConv2d.forward_fourier (grid_2d.py:51-99) and its backward, composed of
 * the stages above over a caller-provided workspace.
 *   w_y = fourier_weight[0] (last spatial axis), w_x = fourier_weight[1] (first spatial axis)
 *   mode FULL / LOWPASS.   ws needs ffno_spectral2d_ws_floats() floats.
 * Backward: gx (+)= d/dx, gw_y/gw_x (+)= d/dW (pass NULL to skip), given gy = dL/dout.
 * --------------------------------------------------------------------------------------------- */
size_t ffno_spectral2d_ws_floats(int B, int M, int N, int C, int K);
/* Which kernels the two calls below run for a shape (round 6): FFNO_SPECTRAL2D_FUSED_X3 -- the fused split kernels of
 * ffno_spectral_x3 (one launch per axis, spectrum tile in LDS, fp16x2 mix packs, the input's range word folded by one ffno_amax
 * launch; every shape ffno_spectral_x3_supported takes on both axes) -- or FFNO_SPECTRAL2D_STAGES, the stage sequence
 * dft_fwd -> fw_pack -> mode_mix -> dft_inv per axis (everything else).  The packed weight sets, fragment tables and range words of
 * the fused path live in `ws` (ffno_spectral2d_ws_floats covers them).
 * ffno_spectral2d_weights_version (optional): by default every call re-packs the weights (four small launches: the weights may
 * have changed).  A caller that knows better declares a non-zero VERSION of the weights it is about to pass with `ws`; a call
 * whose (w_y, w_x, shape, mode, version) equal those the packs inside `ws` were made from skips the re-pack.  The record is kept
 * per workspace pointer on the host (no device read-back); declare 0 -- or a new number -- after the weights changed. */
#define FFNO_SPECTRAL2D_STAGES 0
#define FFNO_SPECTRAL2D_FUSED_X3 1
int ffno_spectral2d_path(int B, int M, int N, int C, int K);
int ffno_spectral2d_weights_version(const float* ws, unsigned long long version);
int ffno_spectral2d_fwd(const float* x, const float* w_y, const float* w_x, float* out, float* ws,
                        const float* tw_n, const float* tw_m, int B, int M, int N, int C, int K,
                        int mode, void* stream);
int ffno_spectral2d_bwd(const float* x, const float* w_y, const float* w_x, const float* gy,
                        float* gx, float* gw_y, float* gw_x, float* ws, const float* tw_n,
                        const float* tw_m, int B, int M, int N, int C, int K, int mode,
                        int accumulate_gx, int accumulate_gw, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Feed-forward  (feedforward.py:13-19 with n_layers=2, dropout=0, no LayerNorm) + residual
 * (grid_2d.py:169):   out = (resid ? resid : 0) + relu(s W1^T + b1) W2^T + b2
 *   W1[H][C], b1[H], W2[C][H], b2[C] are the EFFECTIVE weights (after weight-norm).
 *   h    (optional, [P][H])  : post-ReLU hidden activations, kept for the weight gradient
 *   mask (optional)          : ReLU sign bits, ffno_ff_mask_words(P,H) uint32 words
 * out may alias resid.
 * --------------------------------------------------------------------------------------------- */
size_t ffno_ff_mask_words(int P, int H);
int ffno_ff_fwd(const float* s, const float* resid, const float* W1, const float* b1,
                const float* W2, const float* b2, float* out, float* h, uint32_t* mask, int P,
                int C, int H, void* stream);
/* data gradient: dh = (db W2) * relu'(.)  [P][H],  ds = dh W1  [P][C].
 * Takes the TRANSPOSED effective weights W1t[C][H] = W1^T and W2t[H][C] = W2^T (see
 * ffno_transpose_batched) so both GEMM operands are row-contiguous in LDS, like the forward. */
int ffno_ff_bwd_data(const float* db, const uint32_t* mask, const float* W1t, const float* W2t,
                     float* dh, float* ds, int P, int C, int H, void* stream);
/* weight gradients (two steps, deterministic):
 *   partial[s] = { dW1[H][C], dW2[C][H], db1[H], db2[C] } over the s-th pixel slice
 *   then reduce over s into dW1, dW2, db1, db2 (accumulate: 0 overwrite / 1 add)          */
size_t ffno_ff_wgrad_partial_floats(int C, int H, int nsplit);
int ffno_ff_bwd_weights_partial(const float* s, const float* db, const float* h, const float* dh,
                                float* partial, int P, int C, int H, int nsplit, void* stream);
int ffno_ff_bwd_weights_reduce(const float* partial, float* dW1, float* dW2, float* db1, float* db2,
                               int C, int H, int nsplit, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The same feed-forward on the bf16 matrix cores at fp32 accuracy ("bf16x3": every fp32 operand cut
 * exactly into three bf16 planes, six v_mfma_f32_32x32x16_bf16 per product block).  Replaces the
 * same reference lines as ffno_ff_* (feedforward.py:13-19 + grid_2d.py:169 and their autograd) and is
 * the path the engine uses when ffno_ffx_supported(C, H); the hidden activations are never stored:
 * the forward keeps the ReLU sign bits only and the weight-gradient kernel recomputes h and dh.
 *
 * Weights are consumed pre-split / pre-permuted ("packed", ffno_ffx_pack_bytes(C,H) bytes each):
 *   type 1 (hidden rows on lanes)  and  type 2 (channel rows on lanes, k = hidden in D-fragment order)
 *   of the matrix element  W(hid, c) = src[hid * sh + c * sc]:
 *     forward      A1 = type1(W1: sh=C, sc=1)    A2 = type2(W2: sh=1, sc=H)
 *     backward     A1 = type1(W2: sh=1, sc=H)    A2 = type2(W1: sh=C, sc=1)
 *     weight grads use the two type-1 packs.
 * ffno_ffx_pack runs a device-resident table of n descriptors in one launch.
 * mask: ffno_ff_mask_words(P,H) uint32 words (layout private to the ffx kernels).
 * --------------------------------------------------------------------------------------------- */
typedef struct ffno_fxpack_desc {
    const float* src; /* effective weight matrix */
    void* dst;        /* ffno_ffx_pack_bytes(C,H) bytes, 16-B aligned */
    int32_t sh, sc;   /* element strides of the hidden / channel index in src */
    int32_t type;     /* 1 or 2 */
    int32_t pad_;
} ffno_fxpack_desc;
int ffno_ffx_supported(int C, int H);
/* Per-call options of the chain kernels (forward / backward-data) of both split families; NULL = all defaults. */
typedef struct ffno_ff_opts {
    const uint32_t* in_amax; /* range word of the addends of the input (s / s2, db / db2); used by ffno_ffh_* only */
    uint32_t* out_amax;      /* optional: receives max |out| (forward: out; backward-data: ds) */
    int32_t max_workgroups;  /* persistent workgroups; 0 = one per compute unit of the device */
    int32_t schedule;        /* 0 = the measured default: ffno_ffh_* at C = 64, H = 256: FFNO_FF_SCHED_WAVE_TILES (one wave per 32-pixel
                                tile, both weight maps in LDS, no staging / exchange / barriers); every other shape and the
                                bf16x3 family: the shared-tile kernels (forward: FFNO_FF_SCHED_ROLE_SPLIT -- a matrix segment on
                                one wave of a SIMD beside a vector / LDS segment on the other; backward-data: both halves in
                                phase).  The shared-tile schedules are bit-identical to each other; WAVE_TILES sums the hidden
                                chunks of an output in one accumulator pair instead of eight partial tiles: equal to fp32
                                rounding (same products), not bit for bit.  An explicit schedule a shape does not have:
                                FFNO_EUNSUPPORTED */
    int32_t storage;         /* FFNO_STORE_F32 (0) / FFNO_STORE_BF16: format of every activation pointer of the call (s, s2,
                                s_sum, resid, out / db, db2, db_sum, ds); bf16: ffno_ffh_* at C = 64, H = 256 */
} ffno_ff_opts;
#define FFNO_FF_SCHED_IN_PHASE 1
#define FFNO_FF_SCHED_WAVE_TILES 2
#define FFNO_FF_SCHED_ROLE_SPLIT 3
size_t ffno_ffx_pack_bytes(int C, int H);
int ffno_ffx_pack(const ffno_fxpack_desc* descs_dev, int n, int C, int H, void* stream);
int ffno_ffx_fwd(const float* s, const float* resid, const void* pk1, const float* b1, const void* pk2,
                 const float* b2, float* out, void* mask, int P, int C, int H, void* stream);
/* decode the private sign-bit layout: active[p * H + h] = 1 iff the forward's hidden unit (p, h) passed the ReLU
 * (= torch's threshold_backward mask of feedforward.py:17); diagnostics and parity tests */
int ffno_ffx_mask_unpack(const void* mask, uint8_t* active, int P, int C, int H, void* stream);
/* ..2 variants: the input is the SUM of two tensors (s + s2, db + db2 -- the two spectral branches of a layer written
 * side by side by concurrent launches); with s_sum / db_sum the sum is also stored for the kernels that follow
 * (s_sum may alias s). */
int ffno_ffx_fwd2(const float* s, const float* s2, float* s_sum, const float* resid, const void* pk1, const float* b1,
                  const void* pk2, const float* b2, float* out, void* mask, int P, int C, int H, const ffno_ff_opts* opts,
                  void* stream);
int ffno_ffx_bwd_data2(const float* db, const float* db2, float* db_sum, const void* mask, const void* pk1b,
                       const void* pk2b, float* ds, int P, int C, int H, const ffno_ff_opts* opts, void* stream);
/* ds = ((db W2) * relu'(.)) W1 ; pk1b / pk2b = the backward packs */
int ffno_ffx_bwd_data(const float* db, const void* mask, const void* pk1b, const void* pk2b, float* ds,
                      int P, int C, int H, void* stream);
/* partial[s] = { dW1^T[C][H], dW2[C][H], db1[H], db2[C] } over the pixels of workgroup s (nsplit
 * workgroups, ffno_ff_wgrad_partial_floats(C,H,nsplit) floats); pk1 = forward A1, pk1b = backward A1 */
int ffno_ffx_bwd_weights_partial(const float* s, const float* db, const void* pk1, const float* b1,
                                 const void* pk1b, float* partial, int P, int C, int H, int nsplit,
                                 void* stream);
int ffno_ffx_bwd_weights_reduce(const float* partial, float* dW1, float* dW2, float* db1, float* db2,
                                int C, int H, int nsplit, int accumulate, void* stream);
/* the same reduction (overwrite) for a device-resident table of n feed-forward blocks in one launch */
typedef struct ffno_fxred_desc {
    const float* partial;
    float* dW1;
    float* dW2;
    float* db1;
    float* db2;
} ffno_fxred_desc;
int ffno_ffx_bwd_weights_reduce_batched(const ffno_fxred_desc* descs_dev, int n, int C, int H, int nsplit,
                                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * The same feed-forward on the fp16 matrix cores at fp32 accuracy ("fp16x2": every fp32 operand x = hi + lo / 2^11 with
 * hi = fp16(x), lo = fp16((x - hi) 2^11), both rounded to nearest, so |x - hi - lo / 2^11| <= 2^-24 |x|; three
 * v_mfma_f32_32x32x16_f16 per product block -- hi hi into a main accumulator, hi lo + lo hi into a correction accumulator that
 * is folded in with the factor 2^-11).  Half the matrix work and two thirds of the operand traffic of ffno_ffx_*, same
 * operators, shapes (ffno_ffx_supported), sign-bit masks, partial-slice layout and reduce kernels
 * (ffno_ffx_bwd_weights_reduce[_batched]); the weight packs are this family's own (ffno_ffh_pack, same descriptors).
 * RANGE: the representation error of an element is max(2^-24 |x|, 2^-36) and the half format ends at 65504, so the kernels
 * scale what they split: every entry point takes the RANGE WORD of its input (see "Range words" at the top: `opts->in_amax`,
 * `s_amax`, `db_amax` -- the word bounds EACH addend; the kernels allow for the sum) and multiplies the rows by
 * 2^e, e chosen on the device so that the bound lands on 2^4, while they are staged; biases are scaled alike, outputs divided
 * again -- exact, no host round trip.  2^12 of head-room remain for the growth through the first linear map (|h| <= ||W
 * row||_1 |s| + |b1|); elements down to 2^-16 of the bound keep fp32-level relative accuracy, smaller ones an absolute error of
 * 2^-40 of the bound.  So activations of 1e6 and gradients of 1e-12 are as good as O(1) data.  The WEIGHTS are split unscaled:
 * |W| < 65504 is a HARD PRECONDITION of every fp16x2 pack (ffno_ffh_pack, ffno_spectral_x3_pack with format 1) that the
 * library does NOT check: a larger weight becomes +-inf in its hi plane and the products are non-finite from then on.  A
 * caller that cannot rule it out folds max |W| of what it packs with ffno_amax (the Python host does so at EVERY rebuild of
 * the packs, reads the word back without synchronising and raises FloatingPointError at the next rebuild -- at most one
 * optimiser step late) or uses the split-bf16 twins (ffno_ffx_*, format 0), which take any fp32 range.  With a NULL range
 * word the data is taken as is (in range: 2^-12 <= |x| < 2^15 for full accuracy).
 * --------------------------------------------------------------------------------------------- */
size_t ffno_ffh_pack_bytes(int C, int H);
int ffno_ffh_pack(const ffno_fxpack_desc* descs_dev, int n, int C, int H, void* stream);
int ffno_ffh_fwd2(const float* s, const float* s2, float* s_sum, const float* resid, const void* pk1, const float* b1,
                  const void* pk2, const float* b2, float* out, void* mask, int P, int C, int H, const ffno_ff_opts* opts,
                  void* stream);
int ffno_ffh_bwd_data2(const float* db, const float* db2, float* db_sum, const void* mask, const void* pk1b,
                       const void* pk2b, float* ds, int P, int C, int H, const ffno_ff_opts* opts, void* stream);
int ffno_ffh_bwd_weights_partial(const float* s, const float* db, const void* pk1, const float* b1, const void* pk1b,
                                 float* partial, int P, int C, int H, int nsplit, const uint32_t* s_amax,
                                 const uint32_t* db_amax, int storage /* FFNO_STORE_*: format of s and db */, void* stream);
/* The weight-gradient slices of n feed-forward blocks (all layers of a backward pass) in ONE launch: nothing downstream of a
 * layer's weight gradient is on the backward's critical path, so a caller that keeps every layer's s and summed gradient runs
 * this once after the last layer's data gradient.  Per block: `nsplit` slices (layout of ffno_ffx_bwd_weights_partial) at
 * descs[i].partial; the grid is n x nsplit workgroups -- 3 x CUs / n slices per block keeps three full rounds of workgroups and
 * makes each one walk many tiles per slice written (the per-layer launch pays 7 us of fragment loads + slice burst and its
 * reduce 7 us per layer at the headline shape, MI355X round 4).  Both range words of every block are required; width 64 / 256
 * and 32 / 128 (the single-accumulator kernel's shapes).  `descs` is a HOST array: it is copied into the kernel arguments at
 * enqueue (32 blocks per launch) -- pointers that arrive as arguments are global memory to the compiler, pointers loaded from a
 * device table are not, and the flat accesses they cause tie a wave's LDS waits to its global prefetch. */
typedef struct ffno_ffwg_desc {
    const void* s;           /* feed-forward input of the block  [P, C], storage format of the call */
    const void* g;           /* gradient w.r.t. its output       [P, C] */
    const void* pk1;         /* forward pack of W1 (ffno_ffh_pack) */
    const float* b1;
    const void* pk1b;        /* backward pack (the pk1b argument of ffno_ffh_bwd_weights_partial) */
    float* partial;          /* nsplit slices */
    const uint32_t* s_amax;
    const uint32_t* g_amax;
    const void* s2;          /* two_addends 1: s = s + s2 -- formed (and rounded to the storage format) while the rows are staged,
                              * for callers whose forward chain launch does not write the sum back (s_sum NULL) */
    const void* g2;          /* reserved (NULL): round 4's second gradient addend was measured slower and removed */
} ffno_ffwg_desc;
int ffno_ffh_bwd_weights_partial_multi(const ffno_ffwg_desc* descs, int n, int P, int C, int H, int nsplit, int storage,
                                       int two_addends /* 0 or 1 (s = s + s2; C = 64, H = 256 only) */, void* stream);

/* Hardware self-check of the LDS transpose read (ds_read_b64_tr_b16) the weight-gradient kernel takes its channel-major operands
 * through: one wave copies image[n16] (halves, n16 <= 8192) into LDS, lane l reads 8 bytes at byte_off[l] (8-byte aligned) with the
 * transpose read and writes the four halves it received to out[4 l .. 4 l + 3].  Expected (csrc/ffno_platform.h): within each
 * 16-lane group, lane i receives element i & 3 of the pieces addressed by lanes 4 r + (i >> 2), r = 0..3. */
int ffno_lds_tr16_probe(const uint16_t* image, int n16, const int32_t* byte_off, uint16_t* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over the channel axis, the last stage of FeedForward(layer_norm=True) (feedforward.py:18-19: nn.LayerNorm(dim),
 * eps 1e-5, biased variance, elementwise affine), fused with the layer's residual add (grid_2d.py:169):
 *   fwd:  out = (t - mean) * rstd * gamma + beta (+ resid);  stats[p] = {mean, rstd} (2 floats per pixel, kept for backward)
 *   bwd:  gy = g (+ g2, the sum optionally stored to g_sum);  dt = dL/dt;  dgamma / dbeta (+)= their gradients
 *         (deterministic two-stage reduction through `partial`: 2 * C * ffno_layernorm_nsplit(P) floats)
 * C = 64 or 32.
 * --------------------------------------------------------------------------------------------- */
int ffno_layernorm_fwd(const float* t, const float* gamma, const float* beta, const float* resid, float* out, float* stats,
                       long P, int C, float eps, void* stream);
int ffno_layernorm_nsplit(long P);
int ffno_layernorm_bwd(const float* t, const float* stats, const float* gamma, const float* g, const float* g2, float* g_sum,
                       float* dt, float* partial, float* dgamma, float* dbeta, long P, int C, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Weight normalisation (linear.py:48-49, torch.nn.utils.weight_norm dim=0), batched over a
 * device-resident descriptor table so one launch covers every linear of the block.
 *   fwd: w = g * v / ||v||_row          bwd: dg = sum_in dw*v/||v|| ; dv = g/||v|| (dw - dg v/||v||)
 * --------------------------------------------------------------------------------------------- */
typedef struct ffno_wn_desc {
    const float* g;  /* [rows]            (weight_g, stored [rows,1]) */
    const float* v;  /* [rows][cols]      (weight_v) */
    float* w;        /* [rows][cols]      effective weight (fwd out) */
    const float* dw; /* [rows][cols]      gradient w.r.t. the effective weight (bwd in) */
    float* dg;       /* [rows]            (bwd out) */
    float* dv;       /* [rows][cols]      (bwd out) */
    int32_t rows;
    int32_t cols;
} ffno_wn_desc;
int ffno_weightnorm_fwd(const ffno_wn_desc* descs_dev, int n, int max_rows, void* stream);
int ffno_weightnorm_bwd(const ffno_wn_desc* descs_dev, int n, int max_rows, void* stream);

/* Batched 2-D transpose dst[cols][rows] = src[rows][cols]^T over a device-resident descriptor table
 * (one launch produces every W^T the backward pass needs). */
typedef struct ffno_tr_desc {
    const float* src;
    float* dst;
    int32_t rows;
    int32_t cols;
} ffno_tr_desc;
int ffno_transpose_batched(const ffno_tr_desc* descs_dev, int n, int max_rows, int max_cols, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pad / crop index map (3-D mesh operator, mesh_3d.py:165,173: F.pad by 8 at the END of each spatial
 * axis after in_proj, crop before the head).  The lift and head kernels address the padded activation
 * buffer directly: unpadded pixel (b, i0, i1, i2) <-> padded pixel (b, i0, i1, i2) of [B][padded...].
 * Pass NULL for "no padding" (2-D grid operator).  Host struct, read at enqueue time.
 * --------------------------------------------------------------------------------------------- */
typedef struct ffno_padmap {
    int32_t size[3];    /* unpadded spatial sizes (use {1, M, N} for 2-D) */
    int32_t padded[3];  /* padded spatial sizes */
} ffno_padmap;

/* ---------------------------------------------------------------------------------------------
 * in_proj (grid_2d.py:112,157 / mesh_3d.py:162): out[q(p)][C] = x[p][Cin] W^T + b, and its parameter
 * gradients (deterministic two-step reduction; partial needs nsplit*C*(Cin+1) floats).
 * --------------------------------------------------------------------------------------------- */
int ffno_lift_fwd(const float* x, const float* W, const float* b, float* out, int P, int Cin, int C,
                  const ffno_padmap* pad, uint32_t* out_amax /* optional range word of out */, void* stream);
int ffno_lift_bwd(const float* x, const float* gout, float* partial, float* dW, float* db, int P,
                  int Cin, int C, int nsplit, int accumulate, const ffno_padmap* pad, void* stream);
/* bf16 storage twins ("Storage formats" at the top; C = 64): the lifted features `out` / their gradient `gout` are bf16 */
int ffno_lift_fwd_bf16(const float* x, const float* W, const float* b, uint16_t* out, int P, int Cin, int C,
                       const ffno_padmap* pad, uint32_t* out_amax, void* stream);
/* the same with the gradient given as the SUM of two buffers (gout + gout2: the two gradient images a paired adjoint launch
 * leaves; gout2 may be NULL) -- added while the rows are staged, no separate pass over the images */
int ffno_lift_bwd2(const float* x, const float* gout, const float* gout2, float* partial, float* dW, float* db, int P,
                   int Cin, int C, int nsplit, int accumulate, const ffno_padmap* pad, void* stream);
int ffno_lift_bwd_bf16(const float* x, const uint16_t* gout, float* partial, float* dW, float* db, int P,
                       int Cin, int C, int nsplit, int accumulate, const ffno_padmap* pad, void* stream);
/* dx[p][Cin] = gout[q(p)][C] W: the gradient with respect to the block's INPUT (the reference modules are ordinary autograd
 * modules: grid_2d.py:154-177 propagates it to whatever produced x) */
int ffno_lift_bwd_data(const float* gout, const float* W, float* dx, int P, int Cin, int C,
                       const ffno_padmap* pad, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Output head (grid_2d.py:150-152,171-172; mesh_3d.py:155-157,174): y[P][O] = (b W_a^T + c_a) W_b^T + c_b
 * with W_a[D][C], W_b[O][D] (O <= 8) and NO activation in between, evaluated as one affine map.
 *   ffno_head_fold : fold[o] = { weff[o][C] = W_b[o] W_a, beff[o] = W_b[o] c_a + c_b[o] }   (fold[O][C+1])
 *   ffno_head_fwd  : y[p][o] (+)= b[q(p)] . weff[o] + beff[o]
 *   ffno_head_bwd  : gb[q(p)][c] = sum_o gy[p][o] weff[o][c];  red[o] = { sum_p gy[p][o] b[q(p)][:], sum_p gy[p][o] }
 *                    (partial needs nsplit*O*(C+1) floats; gb is only written at q(p): pre-zero it when padded)
 *   ffno_head_param_grads : dW_a, dc_a, dW_b, dc_b from red (exact chain rule of the two linears)
 * --------------------------------------------------------------------------------------------- */
int ffno_head_fold(const float* Wa, const float* ca, const float* Wb, const float* cb, float* fold,
                   int C, int D, int O, void* stream);
int ffno_head_fwd(const float* b, const float* fold, float* y, int P, int C, int O, int accumulate,
                  const ffno_padmap* pad, void* stream);
int ffno_head_bwd(const float* b, const float* gy, const float* fold, float* gb, float* partial,
                  float* red, int P, int C, int O, int nsplit, const ffno_padmap* pad,
                  uint32_t* gb_amax /* optional range word of gb */, void* stream);
/* bf16 storage twins (C = 64): the last layer's output `b` and its gradient `gb` are bf16; y, gy, red stay fp32 */
int ffno_head_fwd_bf16(const uint16_t* b, const float* fold, float* y, int P, int C, int O, int accumulate,
                       const ffno_padmap* pad, void* stream);
int ffno_head_bwd_bf16(const uint16_t* b, const float* gy, const float* fold, uint16_t* gb, float* partial,
                       float* red, int P, int C, int O, int nsplit, const ffno_padmap* pad, uint32_t* gb_amax, void* stream);
int ffno_head_param_grads(const float* red, const float* Wa, const float* ca, const float* Wb,
                          float* dWa, float* dca, float* dWb, float* dcb, int C, int D, int O,
                          int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Non-factorized 2-D spectral convolution (FNOPlus2DBlock, zongyi_fno/grid_plus_2d.py:52-83):
 *   rfft2 -> two K x K corner blocks (rows [0,K) and [M-K,M), columns [0,K)) mixed with their own
 *   weights [I][O][K][K][2] -> zero-padded irfft2.  The last-axis transforms, the per-mode mix and the
 *   weight-gradient contraction are ffno_dft_fwd / ffno_mode_mix / ffno_dft_inv / ffno_fw_grad_partial with
 *   the 2K*K retained (kx', ky) pairs as modes (mode = ky*2K + kx') and the B samples as rows; these
 *   entry points add the complex DFT along the first axis and the weight layout conversions.
 *   ffno_cdft_rows: inverse = 0:  Z[ky][kx'][b][re/im][c] = 1/sqrt(M) sum_m e^{-2 pi i kx m/M} S[ky][b*M+m][re/im][c]
 *                   inverse = 1:  the zero-padded inverse (also the adjoint of the forward);  2K <= M.
 * --------------------------------------------------------------------------------------------- */
int ffno_cdft_rows(const float* in, float* out, int B, int M, int C, int K, int inverse, void* stream);
int ffno_fw2d_pack(const float* w0, const float* w1, float* wp, float* wpt, int C, int K, void* stream);
int ffno_fw2d_grad_reduce(const float* partial, float* gw0, float* gw1, int C, int K, int nsplit,
                          int accumulate, void* stream);
/* the same with different numbers of retained rows (Kx per corner block) and columns (Ky): FNOMesh2D keeps
 * modes1 != modes2 (zongyi_fno/mesh_2d.py:46-49); weights [I][O][Kx][Ky][2], modes = ky*2Kx + kx' */
int ffno_cdft_rows2(const float* in, float* out, int B, int M, int C, int Kx, int Ky, int inverse,
                    void* stream);
int ffno_fw2d_pack2(const float* w0, const float* w1, float* wp, float* wpt, int C, int Kx, int Ky,
                    void* stream);
int ffno_fw2d_grad_reduce2(const float* partial, float* gw0, float* gw1, int C, int Kx, int Ky, int nsplit,
                           int accumulate, void* stream);
/* ffno_cdft_rows2 on the matrix cores: a complex line is two real lines, so the retained rows follow from the truncated
 * real-DFT kernels (ffno_dft_fwd / ffno_dft_inv internals) over k = 0..Kx plus an element-wise combination -- same
 * layouts, same results to fp32 rounding, the inverse is still the exact adjoint of the forward.
 * ws: ffno_cdft_rows_ws_floats(B, C, Kx, Ky) floats; tw: twiddle table of length M (ffno_twiddle_fill_host). */
size_t ffno_cdft_rows_ws_floats(int B, int C, int Kx, int Ky);
int ffno_cdft_rows_mfma(const float* in, float* out, float* ws, const float* tw, int B, int M, int C,
                        int Kx, int Ky, int inverse, void* stream);

/* 3-D corner blocks (FNOMesh3D, zongyi_fno/mesh_3d.py:38-57): rfftn -> four blocks (+-x, +-y, low z) x their own complex
 * weights [I][O][K1][K2][K3][2] (w1 = (+x,+y), w2 = (-x,+y), w3 = (+x,-y), w4 = (-x,-y)) -> irfftn.  The transforms are
 * ffno_dft_fwd (z) -> ffno_cdft_rows_mfma (y, with B := B*X lines) -> ffno_cdft_rows_mfma (x, with Ky := K3*2K2) and back;
 * the mix and the weight-gradient contraction are ffno_mode_mix / ffno_fw_grad_partial over the K3*2K2*2K1 retained modes
 * (mode = (kz*2K2 + ky')*2K1 + kx') with the B samples as rows.  These two convert the weight layouts. */
int ffno_fw3d_pack(const float* w1, const float* w2, const float* w3, const float* w4, float* wp, float* wpt,
                   int C, int K1, int K2, int K3, void* stream);
int ffno_fw3d_grad_reduce(const float* partial, float* g1, float* g2, float* g3, float* g4, int C, int K1,
                          int K2, int K3, int nsplit, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CNOFactorized* operators (factorized_cno/grid_2d.py:51-96 with modules/dct.py): the F-FNO layer with an
 * orthonormal DCT-II per axis and REAL per-mode weights [I][O][K].  One branch
 *     out (+)= [resid +] iDCT_zero-padded( W (.) DCT(in)[:K] )          (conj_transpose: W^T, the adjoint branch)
 * runs on the truncated real-DFT kernels: X[k] = a_k Re(e^{-i pi k/2L} F_2L[k]) over the L samples of a length-2L
 * transform, and the transpose for the inverse (the DCT is orthonormal, so forward and adjoint branches use the same two
 * transforms).  tw2 = twiddle table of length 2L.  spec keeps the DCT coefficients (complex layout, imaginary parts 0)
 * for ffno_fw_grad_partial; mix is scratch of the same size (K*R*2*C floats).  planes from ffno_fw_pack_real
 * (complex planes with a zero imaginary plane); NULL = no weights.  K <= min(L, 64).
 * --------------------------------------------------------------------------------------------- */
int ffno_dct_branch(const float* in, float* out, const float* resid, float* spec, float* mix,
                    const float* planes, const float* tw2, int B, int M, int N, int C, int K, int axis,
                    int conj_transpose, int accumulate, void* stream);
int ffno_fw_pack_real(const float* w, float* wp, float* wpt, int C, int K, void* stream);
/* per-mode mix of REAL spectra with real weights (the planes' real parts): only the real part of spec_out is written */
int ffno_mode_mix_real(const float* spec_in, const float* planes, float* spec_out, int R, int C, int K,
                       void* stream);
int ffno_fw_grad_reduce_real(const float* partial, float* gw, int C, int K, int nsplit, int accumulate,
                             void* stream);

/* ---------------------------------------------------------------------------------------------
 * Velocity features of the Markov routine (routines/grid_2d_markov.py:130-144, `use_velocity: true`,
 * wavenumber buffers :82-94): vorticity[B][X][Y] -> out[B][X][Y][3] = (vorticity, u, v) with
 *   psi^ = -rfftn(w)/lap,  u = irfftn(2 pi i ky psi^),  v = irfftn(-2 pi i kx psi^)
 * on the periodic domain [0,len_x) x [0,len_y) (reference default 2 pi).  Y must be even.
 * ws: ffno_velocity_ws_floats(B, X, Y) floats.  `out` feeds ffno_markov_features with Cx = 3.
 * --------------------------------------------------------------------------------------------- */
size_t ffno_velocity_ws_floats(int B, int X, int Y);
int ffno_velocity_features(const float* vorticity, float* out, float* ws, int B, int X, int Y,
                           float len_x, float len_y, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Relative-L2 loss (loss.py:33-46) and its gradient:
 *   loss = mean_b ||pred_b - y_b||_2 / ||y_b||_2 ;  gpred = dloss/dpred * gscale
 * work buffer `tmp`: ffno_lploss_tmp_floats(B, n_per_sample) floats (two-stage deterministic reduction,
 * up to 64 slices per sample).  loss is written to loss_out[0].
 * --------------------------------------------------------------------------------------------- */
size_t ffno_lploss_tmp_floats(int B, int n_per_sample);
/* affine (optional, device float[2] = {scale, shift}): the loss is taken on pred*scale + shift, i.e. the
 * Normalizer.inverse(channel=0) of grid_2d_markov.py:185 fused in (scale = std[0], shift = mean[0]). */
int ffno_lploss_fwd_bwd(const float* pred, const float* target, float* loss_out, float* gpred,
                        float* tmp, int B, int n_per_sample, float gscale, const float* affine,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * Feature build of the Markov routine + running normaliser, fused
 * (routines/grid_2d_markov.py:124-170; modules/normalizer.py:18-77):
 *   raw[p]  = [ x[p][0..Cx), linspace(low,high,M)[m], linspace(low,high,N)[n] ]          D = Cx + 2 <= 16
 *   `extra` (optional) selects the other feature channels of :146-162, in the reference's order:
 *   position (use_position, default 1) | force[B][M][N] (append_force) | mu[B] broadcast (append_mu)
 *   accumulate != 0: state.sum += sum_p raw, state.sum_squared += sum_p raw^2, count += B*M*N, n_accumulations += 1
 *   derived = { mean[D], std[D] = max(sqrt(sum_squared/count - mean^2), eps) }
 *   out[p][c] = (normalize ? (raw - mean)/std : raw) + (noise ? noise[p][c]*noise_std : 0)
 * state = float[2D+2] {sum[D], sum_squared[D], count, n_accumulations}; partial = float[256*32] scratch.
 * --------------------------------------------------------------------------------------------- */
typedef struct ffno_markov_extra {
    const float* force; /* [B][M][N] or NULL */
    const float* mu;    /* [B] or NULL */
    int32_t use_position;
    int32_t pad_;
} ffno_markov_extra;
int ffno_markov_features(const float* x, float* state, float* derived, const float* noise, float* out,
                         float* partial, int B, int M, int N, int Cx, float low, float high,
                         float noise_std, float eps, int accumulate, int normalize,
                         const ffno_markov_extra* extra, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused AdamW over one flat parameter buffer (torch.optim.AdamW semantics, config.yaml:36-40):
 *   p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)        with g := grad * grad_scale
 * --------------------------------------------------------------------------------------------- */
int ffno_adamw_flat(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, float grad_scale,
                    void* stream);
/* torch.optim.Adam instead (the geo-FNO baselines' optimiser): weight decay is L2, g := grad * grad_scale + wd * p, and
 * p is not decayed. */
int ffno_adam_flat(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pointwise linear layers of the FNOZongyi2DBlock baseline (zongyi_fno/grid_2d.py:22,45,74-77 the per-layer
 * `linear` + residual + ReLU; :106 in_proj; :119-122 feedforward head) on channels-last buffers whose
 * leading dimension may exceed the logical width (20 channels live in 32-channel tiles, pad = 0):
 *   fwd:         pre = b[o] + sum_i W[o][i] x[p*ldx+i] + add[p*ldo+o];  out[p][o] = act(pre),  o < Cout; 0 for Cout <= o < ldo
 *                act_mode: FFNO_ACT_NONE | FFNO_ACT_RELU | FFNO_ACT_GELU (exact erf form = torch.nn.functional.gelu)
 *                out2 (optional) = out + res   (block-level residual `layer(x) + x`, grid_2d.py:126)
 *                pre_out (optional) = pre      (what the GELU backward needs; ReLU only needs the sign of out)
 *   bwd_data:    dpre = g * act'(.) with `act` = the kept OUTPUT for ReLU, the kept PRE-activation for GELU (NULL: dpre = g);
 *                dx[p*ldx+i] (+)= sum_o dpre[p][o] W[o][i]  (i >= Cin: 0);  dpre_out (optional, layout of g) receives dpre
 *   bwd_weights: dW[o][i] (+)= sum_p dpre[p][o] x[p][i],  db[o] (+)= sum_p dpre[p][o]   (deterministic two-stage
 *                reduction through `part`, ffno_plin_wgrad_partial_floats() floats)
 * W is [Cout][Cin] row-major (nn.Linear.weight).  Cin, Cout <= 128 and (Cin+1)*ceil(Cout/8) <= 1280.
 * --------------------------------------------------------------------------------------------- */
#define FFNO_ACT_NONE 0
#define FFNO_ACT_RELU 1
#define FFNO_ACT_GELU 2
int ffno_plin_supported(int Cin, int Cout);
int ffno_plin_fwd(const float* x, int ldx, const float* W, const float* b, const float* add, float* out,
                  int ldo, const float* res, float* out2, float* pre_out, long P, int Cin, int Cout,
                  int act_mode, void* stream);
int ffno_plin_bwd_data(const float* g, int ldg, const float* act, const float* W, float* dx, int ldx,
                       float* dpre_out, long P, int Cin, int Cout, int accumulate, int act_mode,
                       void* stream);
int ffno_plin_wgrad_nsplit(long P);
size_t ffno_plin_wgrad_partial_floats(long P, int Cin, int Cout);
int ffno_plin_bwd_weights(const float* g, int ldg, const float* act, const float* x, int ldx, float* part,
                          float* dW, float* db, long P, int Cin, int Cout, int accumulate, int act_mode,
                          void* stream);

/* ---------------------------------------------------------------------------------------------
 * The GENERAL feed-forward path (fourierflow/modules/feedforward.py:6-24 beyond the fused kernels' n_layers = 2 / dropout = 0,
 * and nn.Dropout(in_dropout) of grid_2d.py:113,158): one linear layer per call, any Cin, Cout <= 256, exact fp32 on the matrix
 * cores, hidden activations kept by the caller.  Dropout is a counter-based mask: element idx of the [P][Cout] output is kept
 * iff hash(seed, idx) >= p * 2^32 and scaled by 1 / (1 - p); the backward calls REGENERATE it from (p, seed) -- nothing is stored.
 *   fwd:         out[p][o] = drop(act(b[o] + sum_i W[o][i] x[p][i])) (+ resid[p][o]);  relu = 1: ReLU  (Linear -> Dropout -> ReLU
 *                of the reference: the positive dropout scale commutes with the ReLU)
 *   bwd_data:    dpre = g * (y ? [y > 0] : keep) / (1 - p);  dx[p][i] (+)= sum_o dpre[p][o] W[o][i]
 *                (y = the kept OUTPUT of a ReLU layer -- its zeros already contain the dropped units; y = NULL: the last layer)
 *   bwd_weights: dW[o][i] (+)= sum_p dpre[p][o] x[p][i],  db[o] (+)= sum_p dpre[p][o]  (deterministic two-stage reduction through
 *                `partial`, ffno_glin_wgrad_partial_floats() floats)
 *   ffno_dropout: x[i] <- keep(seed, i) ? x[i] / (1 - p) : 0 in place (in_dropout; the same call on the gradient is its
 *                backward); ffno_dropout_mask writes the keep bits (diagnostics / parity tests).
 * --------------------------------------------------------------------------------------------- */
int ffno_glin_supported(int Cin, int Cout);
int ffno_glin_fwd(const float* x, const float* W, const float* b, const float* resid, float* out, long P, int Cin, int Cout,
                  int relu, float drop_p, uint32_t drop_seed, void* stream);
int ffno_glin_bwd_data(const float* g, const float* y, const float* W, float* dx, long P, int Cin, int Cout, float drop_p,
                       uint32_t drop_seed, int accumulate, void* stream);
int ffno_glin_wgrad_nsplit(long P);
size_t ffno_glin_wgrad_partial_floats(long P, int Cin, int Cout);
int ffno_glin_bwd_weights(const float* g, const float* y, const float* x, float* partial, float* dW, float* db, long P, int Cin,
                          int Cout, float drop_p, uint32_t drop_seed, int accumulate, void* stream);
int ffno_dropout(float* x, size_t n, float p, uint32_t seed, void* stream);
int ffno_dropout_mask(uint8_t* keep, size_t n, float p, uint32_t seed, void* stream);

/* One launch copying n parameter tensors between their reference shapes and channel-padded twins:
 * plain [R][Cc][inner] <-> rows r < R, columns c < Cc of padded [.][Cp][inner]; to_padded = 0 copies back
 * (gradients).  descs is a DEVICE array. */
typedef struct ffno_pad_desc {
    float* plain;
    float* padded;
    int32_t R, Cc, inner, Cp;
} ffno_pad_desc;
int ffno_pad_copy(const ffno_pad_desc* descs, int n, int to_padded, void* stream);

/* small utilities used by the host driver */
int ffno_axpy(float* y, const float* x, float alpha, size_t n, void* stream); /* y += alpha*x */

#ifdef __cplusplus
}
#endif
#endif /* FFNO_H_ */
