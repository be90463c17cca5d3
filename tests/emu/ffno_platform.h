// TEST INFRASTRUCTURE: the names of fourierflow_amd/csrc/ffno_platform.h implemented on the CPU wave emulator
// (hip_emu.h).  tests/emu/build_emu.py puts this directory first on the include path, so the UNCHANGED kernel sources
// compile against it; the package never sees this file.
#pragma once

#include "hip_emu.h"
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef emu_u32x4 u32x4;

#define FFNO_BUILD_TARGET "emu"
#define FFNO_DYN_SMEM(name) char* name = (char*)(((uintptr_t)emu::S().dyn_smem.data() + 63) & ~(uintptr_t)63)
#define FFNO_UNROLL
#define FFNO_NOUNROLL
#define FFNO_SCHED_FENCE() ((void)0)
#define FFNO_SCHED_PIN_VMEM() ((void)0)
#define FFNO_SCHED_PIN_DSREAD() ((void)0)
#define FFNO_SCHED_GROUP(mask, size) ((void)0)
#define FFNO_DRAIN_MEMORY() ((void)0)
#define FFNO_PIN(x) ((void)0)
#define FFNO_WAVES_PER_SIMD(n)

namespace ffno {
namespace plat {

inline f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) { return emu::mfma_32x32x2(a, b, c); }
inline f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) { return emu::mfma_16x16x4(a, b, c); }
inline f32x16 mfma_bf16_32x32x16(u32x4 a, u32x4 b, f32x16 c) { return emu::mfma_32x32x16_bf16(a, b, c); }
inline f32x4 mfma_bf16_16x16x32(u32x4 a, u32x4 b, f32x4 c) { return emu::mfma_16x16x32_bf16(a, b, c); }
inline f32x16 mfma_f16_32x32x16(u32x4 a, u32x4 b, f32x16 c) { return emu::mfma_32x32x16_f16(a, b, c); }
inline f32x4 mfma_f16_16x16x32(u32x4 a, u32x4 b, f32x4 c) { return emu::mfma_16x16x32_f16(a, b, c); }
inline unsigned pack_f16(float x0, float x1) { return (unsigned)emu::emu_float_to_half(x0) | ((unsigned)emu::emu_float_to_half(x1) << 16); }
inline float f16_lo(unsigned w) { return emu::emu_half_to_float((unsigned short)(w & 0xffffu)); }
inline float f16_hi(unsigned w) { return emu::emu_half_to_float((unsigned short)(w >> 16)); }
inline unsigned pk_mul_f16(unsigned w, float k) { return pack_f16(f16_lo(w) * k, f16_hi(w) * k); }
inline void split2_pair(float x0, float x1, unsigned& h, unsigned& l) {
    h = pack_f16(x0, x1);
    l = pack_f16((x0 - f16_lo(h)) * 2048.f, (x1 - f16_hi(h)) * 2048.f);
}
inline void split2_pair_mix(float x0, float x1, unsigned& h, unsigned& l) { split2_pair(x0, x1, h, l); }
inline unsigned pack_hi16(unsigned u0, unsigned u1) { return (u0 >> 16) | (u1 & 0xffff0000u); }
inline unsigned pack_lo16(unsigned u0, unsigned u1) { return (u0 & 0xffffu) | (u1 << 16); }
inline unsigned bf16_rne(float x) {      // IEEE round-to-nearest-even to bf16 (NaN stays NaN)
    unsigned u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
inline unsigned pack_bf16(float x0, float x1) { return bf16_rne(x0) | (bf16_rne(x1) << 16); }
inline void swap_halves(unsigned& a, unsigned& b) {
    float fa, fb;
    memcpy(&fa, &a, 4), memcpy(&fb, &b, 4);
    const float ta = __shfl_xor(fa, 32), tb = __shfl_xor(fb, 32);     // (plain copies: bit patterns survive)
    const bool hi = (emu::S().cur->linear & 63) >= 32;
    const float na = hi ? tb : fa, nb = hi ? fb : ta;
    memcpy(&a, &na, 4), memcpy(&b, &nb, 4);
}
inline uint2 lds_read_tr16_b64(const void* p) { return emu::lds_read_tr16_b64(p); }
inline uint32_t shift_in_msb(uint32_t acc, uint32_t x) { return (acc << 1) | (x >> 31); }
inline uint32_t bit_to_mask(uint32_t x, int b) { return 0u - ((x >> b) & 1u); }
inline void sleep_cycles(int) {}
inline void sincos_pi(float x, float& s, float& c) {
    const double a = 3.14159265358979323846 * (double)x;
    s = (float)sin(a), c = (float)cos(a);
}

static inline int cu_count() { return 256; }      // (the emulator plays an MI355X)

// rendezvous of the 64 fibres of a wave (an exchange through LDS between the lanes of one wave needs every lane's write done)
__device__ __forceinline__ void wave_sync() { (void)__shfl(0.f, 0); }
// the persistent whole-forward kernel: the emulator runs workgroups one after the other, so its host side launches ONE phase per
// launch (kPersistentLaunch false) and the barrier between phases is the end of a launch -- group_sync is never reached
static constexpr bool kPersistentLaunch = false;
static inline int launch_cooperative(const void*, dim3, dim3, void**, size_t, hipStream_t) { return -2; }
inline int xcc_id() { return (int)(blockIdx.x & 7u); }
inline unsigned long long realtime() { return 0ull; }
inline bool group_sync(unsigned*, unsigned, unsigned* err, int*) {
    *err += 1;      // (reaching this is a host-side bug: more than one phase in an emulated launch)
    return false;
}
}  // namespace plat

template <class K>
static inline int allow_dynamic_lds(K, size_t) { return 0; }

}  // namespace ffno
