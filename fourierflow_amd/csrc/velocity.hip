// Velocity features of the Markov routine: vorticity -> (vorticity, u, v) through the stream function.
//
// Replaces Grid2DMarkovExperiment._build_features, `use_velocity` branch (reference
// fourierflow/routines/grid_2d_markov.py:130-144, wavenumber buffers built at :82-94 from jax_cfd's Grid.rfft_mesh):
//     w^ = rfftn(w, dim=[1,2], norm='backward');  psi^ = -w^ / lap,  lap = (2 pi i)^2 (kx^2 + ky^2), lap[0,0] = 1
//     u = irfftn( 2 pi i ky psi^),   v = irfftn(-2 pi i kx psi^)                 kx = fftfreq(X, Lx/X), ky = rfftfreq(Y, Ly/Y)
// One single-channel field per sample, so the work is tiny next to the operator (0.35 GFLOP per 256x256 image as plain
// DFT sums); three small kernels, no matrix cores:
//     rows    A[b][x][n]  = sum_y w[b][x][y] e^{-2 pi i n y / Y}                       n = 0 .. Y/2
//     columns W[m] = sum_x A[x][n] e^{-2 pi i m x / X} -> multipliers -> Zq/Zv[x] = sum_m (.)[m] e^{+2 pi i m x / X}
//     rows^-1 q[b][x][y] = 1/(XY) sum_n c_n Re( Zq[b][x][n] e^{+2 pi i n y / Y} ),  c_n = 1 for n = 0 and n = Y/2, else 2
//             (c2r semantics of irfftn's last axis: imaginary parts of the DC and Nyquist bins are ignored)
// Twiddles come from LDS tables built with an exact integer range reduction (k mod N before the sincos).
#include "ffno_device.h"
#include "ffno.h"

#include <cmath>

namespace ffno {

__device__ __forceinline__ void sincos_2pi_frac(int k, int n, float& s, float& c) {   // angle = 2 pi k / n, 0 <= k < n
    plat::sincos_pi(2.f * (float)k / (float)n, s, c);
}

// A[b][x][n] (complex) = sum_y w[b][x][y] e^{-2 pi i n y / Y};  one block per (b, x) row, LDS: row + tables
__global__ __launch_bounds__(256) void vel_rows_fwd_kernel(const float* __restrict__ w, float2* __restrict__ A, int X,
                                                           int Y, int Yh) {
    FFNO_DYN_SMEM(smem);
    float* row = reinterpret_cast<float*>(smem);
    float* ct = row + Y;
    float* st = ct + Y;
    const long r = blockIdx.x;   // b * X + x
    for (int y = threadIdx.x; y < Y; y += blockDim.x) {
        row[y] = w[r * Y + y];
        sincos_2pi_frac(y, Y, st[y], ct[y]);
    }
    __syncthreads();
    for (int n = threadIdx.x; n < Yh; n += blockDim.x) {
        float re = 0.f, im = 0.f;
        int idx = 0;
        for (int y = 0; y < Y; ++y) {
            re = fmaf(row[y], ct[idx], re);
            im = fmaf(-row[y], st[idx], im);
            idx += n;
            if (idx >= Y) idx -= Y;
        }
        A[r * Yh + n] = make_float2(re, im);
    }
}

// one block per (b, n): forward DFT along x, stream-function multipliers, inverse DFT along x for both components
__global__ __launch_bounds__(256) void vel_cols_kernel(const float2* __restrict__ A, float2* __restrict__ Zq,
                                                       float2* __restrict__ Zv, int X, int Yh, float two_pi_over_lx,
                                                       float two_pi_over_ly) {
    FFNO_DYN_SMEM(smem);
    float2* col = reinterpret_cast<float2*>(smem);
    float2* qh = col + X;
    float2* vh = qh + X;
    float* ct = reinterpret_cast<float*>(vh + X);
    float* st = ct + X;
    const int n = blockIdx.x % Yh;
    const long b = blockIdx.x / Yh;
    for (int x = threadIdx.x; x < X; x += blockDim.x) {
        col[x] = A[(b * X + x) * Yh + n];
        sincos_2pi_frac(x, X, st[x], ct[x]);
    }
    __syncthreads();
    const float wy = two_pi_over_ly * (float)n;                  // 2 pi ky
    for (int m = threadIdx.x; m < X; m += blockDim.x) {
        float re = 0.f, im = 0.f;
        int idx = 0;
        for (int x = 0; x < X; ++x) {                            // (a + i b)(c - i s)
            const float a = col[x].x, bb = col[x].y, c = ct[idx], s = st[idx];
            re += a * c + bb * s;
            im += bb * c - a * s;
            idx += m;
            if (idx >= X) idx -= X;
        }
        const int ms = m < (X + 1) / 2 ? m : m - X;              // numpy fftfreq ordering (index X/2 is -X/2 for even X)
        const float wx = two_pi_over_lx * (float)ms;             // 2 pi kx
        const float k2 = wx * wx + wy * wy;
        const float inv = (m == 0 && n == 0) ? -1.f : 1.f / k2;   // psi^ = -w^/lap, lap = -k2 (lap[0,0] = 1)
        const float pr = re * inv, pi_ = im * inv;
        qh[m] = make_float2(-wy * pi_, wy * pr);                 //  i wy psi^
        vh[m] = make_float2(wx * pi_, -wx * pr);                 // -i wx psi^
    }
    __syncthreads();
    for (int x = threadIdx.x; x < X; x += blockDim.x) {
        float qr = 0.f, qi = 0.f, vr = 0.f, vi = 0.f;
        int idx = 0;
        for (int m = 0; m < X; ++m) {                            // (a + i b)(c + i s)
            const float c = ct[idx], s = st[idx];
            qr += qh[m].x * c - qh[m].y * s;
            qi += qh[m].x * s + qh[m].y * c;
            vr += vh[m].x * c - vh[m].y * s;
            vi += vh[m].x * s + vh[m].y * c;
            idx += x;
            if (idx >= X) idx -= X;
        }
        Zq[(b * X + x) * Yh + n] = make_float2(qr, qi);
        Zv[(b * X + x) * Yh + n] = make_float2(vr, vi);
    }
}

// out[b][x][y][0..2] = (w, u, v); one block per (b, x) row
__global__ __launch_bounds__(256) void vel_rows_inv_kernel(const float* __restrict__ w, const float2* __restrict__ Zq,
                                                           const float2* __restrict__ Zv, float* __restrict__ out,
                                                           int X, int Y, int Yh) {
    FFNO_DYN_SMEM(smem);
    float2* zq = reinterpret_cast<float2*>(smem);
    float2* zv = zq + Yh;
    float* ct = reinterpret_cast<float*>(zv + Yh);
    float* st = ct + Y;
    const long r = blockIdx.x;
    for (int n = threadIdx.x; n < Yh; n += blockDim.x) {
        zq[n] = Zq[r * Yh + n];
        zv[n] = Zv[r * Yh + n];
    }
    for (int y = threadIdx.x; y < Y; y += blockDim.x) sincos_2pi_frac(y, Y, st[y], ct[y]);
    __syncthreads();
    const float scale = 1.f / ((float)X * (float)Y);
    for (int y = threadIdx.x; y < Y; y += blockDim.x) {
        float q = 0.f, v = 0.f;
        int idx = 0;
        for (int n = 0; n < Yh; ++n) {
            const float cn = (n == 0 || 2 * n == Y) ? 1.f : 2.f;
            const float c = ct[idx], s = st[idx];
            q += cn * (zq[n].x * c - zq[n].y * s);
            v += cn * (zv[n].x * c - zv[n].y * s);
            idx += y;
            if (idx >= Y) idx -= Y;
        }
        float* o = out + (r * Y + y) * 3;
        o[0] = w[r * Y + y];
        o[1] = q * scale;
        o[2] = v * scale;
    }
}

static inline int vel_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFNO_OK : (int)e;
}

}  // namespace ffno

using namespace ffno;

extern "C" size_t ffno_velocity_ws_floats(int B, int X, int Y) { return (size_t)3 * 2 * (size_t)B * X * (Y / 2 + 1); }

extern "C" int ffno_velocity_features(const float* vorticity, float* out, float* ws, int B, int X, int Y, float len_x,
                                      float len_y, void* stream) {
    if (!vorticity || !out || !ws || B <= 0 || X <= 0 || Y <= 0 || !(len_x > 0.f) || !(len_y > 0.f)) return FFNO_EINVAL;
    if ((Y & 1) || X > 4096 || Y > 4096) return FFNO_EUNSUPPORTED;   // irfftn of an odd last axis changes the size (reference
                                                                      // configs are 64 / 128 / 256 grids)
    const int Yh = Y / 2 + 1;
    hipStream_t s = (hipStream_t)stream;
    float2* A = reinterpret_cast<float2*>(ws);
    float2* Zq = A + (size_t)B * X * Yh;
    float2* Zv = Zq + (size_t)B * X * Yh;
    const float kPi = 3.14159265358979323846f;
    FFNO_LAUNCH(vel_rows_fwd_kernel, dim3((unsigned)((long)B * X)), dim3(256), sizeof(float) * 3 * Y, s, vorticity, A, X, Y,
                Yh);
    int rc = vel_status();
    if (rc) return rc;
    FFNO_LAUNCH(vel_cols_kernel, dim3((unsigned)((long)B * Yh)), dim3(256), sizeof(float) * 8 * X, s, A, Zq, Zv, X, Yh,
                2.f * kPi / len_x, 2.f * kPi / len_y);
    rc = vel_status();
    if (rc) return rc;
    FFNO_LAUNCH(vel_rows_inv_kernel, dim3((unsigned)((long)B * X)), dim3(256), sizeof(float) * (4 * Yh + 2 * Y), s,
                vorticity, Zq, Zv, out, X, Y, Yh);
    return vel_status();
}
