// Layer-level entry points (SURVEY 8b "signature level 2"): one C call enqueues a whole factorized Fourier layer
//   forward :  s = branch_a(x) + branch_b(x) ;  x' = resid + W2 relu(W1 s + b1) + b2
//              (reference SpectralConv2d.forward + the residual add: grid_2d.py:42-49,169)
//   backward:  ds = FF^T(g) ; weight-gradient slices of the feed-forward ; g' = resid + branch_a^T(ds) [+] branch_b^T(ds)
// by sequencing the kernels of spectral_x3.hip / spectral.hip and ffx.hip on the caller's stream.  Nothing new is computed
// here: the point is the boundary -- a host that is not Python (or a Python host that wants 3x fewer FFI crossings per
// step) drives a layer with one call and device-resident operands.
#include "ffno_device.h"
#include "ffno.h"

static int layer_branches(const ffno_fused_branch* a, const ffno_fused_branch* b, int kernel, int interleave, int C,
                          int fwd, void* stream) {
    const int ck_f = fwd ? 0 : 1, ck_i = fwd ? 1 : 0, conj = fwd ? 0 : 1;
    if (kernel == FFNO_BRANCH_X3) return ffno_spectral_x3_pair(a, b, C, ck_f, ck_i, conj, interleave, stream);
    if (kernel == FFNO_BRANCH_FUSED) return ffno_spectral_fused_pair(a, b, C, ck_f, ck_i, conj, stream);
    return FFNO_EINVAL;
}

extern "C" int ffno_layer_fwd(const ffno_layer_fwd_desc* d, void* stream) {
    if (!d) return FFNO_EINVAL;
    int rc = layer_branches(&d->a, &d->b, d->branch_kernel, d->interleave, d->C, 1, stream);
    if (rc) return rc;
    // the feed-forward reads the word both branches folded their output maxima into
    // (the storage format of the layer's activation tensors is the one its branches name)
    const ffno_ff_opts o = {d->a.out_amax, d->out_amax, d->ff_max_workgroups, d->ff_schedule, d->a.storage};
    if (d->ff_kernel == FFNO_FF_FP16X2)
        return ffno_ffh_fwd2(d->a.out, d->b.out, d->s_sum, d->resid, d->pk1, d->b1, d->pk2, d->b2, d->out, d->mask, d->P,
                             d->C, d->H, &o, stream);
    if (d->ff_kernel != FFNO_FF_BF16X3) return FFNO_EINVAL;
    return ffno_ffx_fwd2(d->a.out, d->b.out, d->s_sum, d->resid, d->pk1, d->b1, d->pk2, d->b2, d->out, d->mask, d->P, d->C,
                         d->H, &o, stream);
}

extern "C" int ffno_layer_bwd(const ffno_layer_bwd_desc* d, void* stream) {
    if (!d) return FFNO_EINVAL;
    if (d->ff_kernel != FFNO_FF_BF16X3 && d->ff_kernel != FFNO_FF_FP16X2) return FFNO_EINVAL;
    const bool h2 = d->ff_kernel == FFNO_FF_FP16X2;
    float* gsum = d->g2 ? d->g_sum : nullptr;
    const ffno_ff_opts o = {d->g_amax, d->ds_amax, d->ff_max_workgroups, d->ff_schedule, d->a.storage};
    if (d->a.storage != FFNO_STORE_F32 && !h2) return FFNO_EUNSUPPORTED;
    int rc = h2 ? ffno_ffh_bwd_data2(d->g, d->g2, gsum, d->mask, d->pk1b, d->pk2b, d->ds, d->P, d->C, d->H, &o, stream)
                : ffno_ffx_bwd_data2(d->g, d->g2, gsum, d->mask, d->pk1b, d->pk2b, d->ds, d->P, d->C, d->H, &o, stream);
    if (rc) return rc;
    // the weight-gradient kernel reads the summed gradient (g_sum when a second addend was given, else g)
    const float* gw = d->g2 ? d->g_sum : d->g;
    if (!d->partial) return layer_branches(&d->a, &d->b, d->branch_kernel, d->interleave, d->C, 0, stream);      // (deferred)
    rc = h2 ? ffno_ffh_bwd_weights_partial(d->s, gw, d->pk1, d->b1, d->pk1b, d->partial, d->P, d->C, d->H, d->nsplit,
                                           d->s_amax, d->g_amax, d->a.storage, stream)
            : ffno_ffx_bwd_weights_partial(d->s, gw, d->pk1, d->b1, d->pk1b, d->partial, d->P, d->C, d->H, d->nsplit, stream);
    if (rc) return rc;
    return layer_branches(&d->a, &d->b, d->branch_kernel, d->interleave, d->C, 0, stream);
}
