// DFT-matrix fragment tables of the fused split kernels (built by ffno_spectral_x3_dft_frags, spectral_x3.hip): the layout of a
// table for (axis length L, modes K) and the loader of one fragment.  Shared by spectral_x3.hip (many-mode / latency / width-32
// kernels) and infer.hip (the inference layer's inverse transforms).
#pragma once

#include "ffno_device.h"

namespace ffno {

struct X3kDft {
    int KKT, RT, NST, nchunks, ntiles, nfwd;
};
static inline X3kDft x3k_dft_layout(int L, int K) {
    X3kDft d;
    d.KKT = K <= 16 ? 32 : (K <= 32 ? 64 : 128);      // (<= 16 modes: the latency kernel's tile height)
    d.RT = d.KKT / 32, d.NST = d.KKT / 16;
    d.nchunks = (L + 63) >> 6, d.ntiles = (L + 31) >> 5;
    d.nfwd = d.RT * d.nchunks * 4;
    return d;
}

// fragment `frag` of the table as the bounded operand of mfma_h2s
__device__ __forceinline__ Hf3 x3k_load_dft(const u32x4* __restrict__ tab, int frag, int lane) {
    Hf3 f;
    f.hi = tab[(frag * 2 + 0) * 64 + lane];
    f.lo = tab[(frag * 2 + 1) * 64 + lane];
    FFNO_UNROLL
    for (int w = 0; w < 4; ++w) f.hs[w] = plat::pk_mul_f16(f.hi[w], kHf2Scale);
    return f;
}


}  // namespace ffno
