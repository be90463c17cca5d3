// Line addressing shared by the spectral kernels: how the R lines of one axis of a channels-last [B][M][N][C] view are
// laid out (reference grid_2d.py:58,76: rfft along dim -1 / dim -2 of the permuted tensor).
#pragma once

namespace ffno {

// ---- line addressing ---------------------------------------------------------------------------
// axis 0: lines (b,m), element n at stride C.      axis 1: lines (b,n), element m at stride N*C.
struct LineMap {
    int lines_per_group;   // axis0: R (one group)   axis1: N
    long group_stride;     // axis0: 0               axis1: M*N*C
    long line_stride;      // axis0: N*C             axis1: C
    long elem_stride;      // axis0: C               axis1: N*C
    __host__ __device__ long base(int r) const {
        return (long)(r / lines_per_group) * group_stride + (long)(r % lines_per_group) * line_stride;
    }
};

static inline LineMap make_linemap(int axis, int B, int M, int N, int C) {
    LineMap m;
    if (axis == 0) {
        m.lines_per_group = B * M;
        m.group_stride = 0;
        m.line_stride = (long)N * C;
        m.elem_stride = C;
    } else {
        m.lines_per_group = N;
        m.group_stride = (long)M * N * C;
        m.line_stride = C;
        m.elem_stride = (long)N * C;
    }
    return m;
}


}  // namespace ffno
