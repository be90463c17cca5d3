// Device-side helpers shared by the F-FNO HIP kernels (gfx950 / CDNA4, wave64).
//
// fp32 matrix work uses v_mfma_f32_32x32x2_f32: exact fp32 (an fmaf chain), 64 cycles/instr/SIMD,
// 157 TFLOP/s chip peak (MI355X_MICROARCH.md "Matrix cores").  Fragment maps (cdna_hip_programming.md s3):
//   A operand: one float per lane, lane l holds A[i = l & 31][k = l >> 5]
//   B operand: one float per lane, lane l holds B[k = l >> 5][j = l & 31]
//   C/D      : 16 floats per lane, reg r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l & 31]
//
// With -DFFNO_EMU the same sources compile for the CPU wave emulator used by the tests
// (tests/emu/hip_emu.h); that build is test infrastructure and is never loaded by the package.
#pragma once

#ifdef FFNO_EMU
#include "hip_emu.h"
#define FFNO_DYN_SMEM(name) char* name = (char*)(((uintptr_t)emu::S().dyn_smem.data() + 63) & ~(uintptr_t)63)
#define FFNO_UNROLL
#define FFNO_NOUNROLL
#define FFNO_SCHED_FENCE() ((void)0)
#else
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define FFNO_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define FFNO_UNROLL _Pragma("unroll")
#define FFNO_NOUNROLL _Pragma("unroll 1")
// bounds live ranges: stops the scheduler from hoisting a whole unrolled loop's operand loads
#define FFNO_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

#include <stdint.h>

// Every launch first clears any stale (non-sticky) error another HIP user of the process left behind
// (e.g. PyTorch probing a capability), so the status a C-ABI entry point returns belongs to OUR launch.
#define FFNO_LAUNCH(...)              \
    do {                              \
        (void)hipGetLastError();      \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

namespace ffno {

static constexpr int kWave = 64;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
#ifdef FFNO_EMU
    return emu::mfma_32x32x2(a, b, c);
#else
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}

// v_mfma_f32_16x16x4_f32: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D reg r = D[4*(l>>4) + r][l & 15]
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
#ifdef FFNO_EMU
    return emu::mfma_16x16x4(a, b, c);
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
    FFNO_UNROLL
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// row of the 32x32 D tile held in accumulator register r by a lane in half `half` (= lane >> 5)
__device__ __forceinline__ int drow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ float wave_sum(float v) {
    FFNO_UNROLL
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

}  // namespace ffno
