// Minimal HIP-on-CPU shim -- TEST INFRASTRUCTURE ONLY (never built into the shipped library).
//
// Lets the *actual* kernel sources under fourierflow_amd/csrc/ be compiled with a host
// compiler (against tests/emu/ffno_platform.h) and executed lane-by-lane on the CPU so that index math, MFMA fragment
// layouts, LDS addressing and barrier placement can be checked against the oracle without a GPU.
//
// Execution model: workgroups run one after another; every thread of a workgroup is a fibre (own stack, hand-written context switch)
// fiber; __syncthreads() and the wave-level exchange ops (MFMA, shuffles) are rendezvous points
// at which a fiber yields to a round-robin scheduler.  Fibers are resumed in a rotating order so
// missing barriers have a chance to show up as wrong results.
//
// MFMA lane maps follow /opt/skills/guides/cdna_hip_programming.md section 3:
//   v_mfma_f32_32x32x2_f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
//                            D col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
//   v_mfma_f32_16x16x4_f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D col=l&15, row=4*(l>>4)+r
// A GPU self-test (tests/test_gpu_mfma_layout.py) checks the hardware against the same maps.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }

typedef int hipError_t;
typedef void* hipStream_t;
static const hipError_t hipSuccess = 0;
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static const int hipMemcpyDeviceToDevice = 3;
static const int hipMemcpyHostToDevice = 1;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned emu_u32x4 __attribute__((ext_vector_type(4)));

namespace emu {

// Context switch of the fibres: callee-saved registers + stack pointer only (x86-64 System V).  glibc's swapcontext also saves
// the signal mask -- one system call per switch, and an MFMA of the emulated wave is 128 switches.
#if !defined(__x86_64__)
#error "tests/emu: the fibre switch is written for x86-64 (the build container and the GPU boxes)"
#endif
__attribute__((naked, noinline)) static void fiber_switch(void** /*save_sp*/, void* /*load_sp*/) {
    asm volatile(
        "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
        "movq %rsp, (%rdi)\n\t"
        "movq %rsi, %rsp\n\t"
        "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\t"
        "ret\n\t");
}

struct Fiber {
    void* sp;
    dim3 tid;
    unsigned linear;
    bool done;
};

struct WaveX {  // per-wave exchange state (double-buffered by parity)
    float a[2][64], b[2][64];
    unsigned xa[2][64][4], xb[2][64][4];   // 8 x bf16 per lane operands of the 32x32x16 forms
    const void* ptr[2][64];                // per-lane addresses of a cross-lane memory operation (ds_read_b64_tr_b16)
    int arrived;
    unsigned gen;
    int parity;
};

struct State {
    dim3 blockIdx_, blockDim_, gridDim_;
    Fiber* cur = nullptr;
    void* sched_sp = nullptr;
    std::vector<Fiber> fibers;
    std::vector<char> stacks;
    std::vector<WaveX> waves;
    std::vector<char> dyn_smem;
    std::function<void()> body;
    int wg_arrived = 0;
    unsigned wg_gen = 0;
    unsigned nthreads = 0;
    unsigned long long n_switch = 0;
};

inline State& S() {
    static State s;
    return s;
}

static const size_t kStack = 128 * 1024;

inline void yield() {
    State& s = S();
    s.n_switch++;
    fiber_switch(&s.cur->sp, s.sched_sp);
}

inline void trampoline() {
    State& s = S();
    s.body();
    s.cur->done = true;
    fiber_switch(&s.cur->sp, s.sched_sp);      // never resumed
    abort();
}

inline void wg_barrier() {
    State& s = S();
    unsigned gen = s.wg_gen;
    if (++s.wg_arrived == (int)s.nthreads) {
        s.wg_arrived = 0;
        s.wg_gen++;
        return;
    }
    while (s.wg_gen == gen) yield();
}

inline WaveX& my_wave() { return S().waves[S().cur->linear >> 6]; }

inline void wave_barrier(WaveX& w) {
    unsigned gen = w.gen;
    if (++w.arrived == 64) {
        w.arrived = 0;
        w.gen++;
        return;
    }
    while (w.gen == gen) yield();
}

// exchange one (a, b) pair across the wave; returns the parity slot that holds this round's data
inline int wave_exchange(float a, float b) {
    WaveX& w = my_wave();
    int lane = S().cur->linear & 63;
    int p = w.parity;
    w.a[p][lane] = a;
    w.b[p][lane] = b;
    wave_barrier(w);
    // the last arriver flipped nothing yet: flip lazily, exactly once per round
    // (every lane sees the same p because parity is only advanced after all have read `p`)
    return p;
}

inline void wave_exchange_done(int p) {
    // every lane calls this after consuming slot p; the first one to do so flips parity.
    WaveX& w = my_wave();
    if (w.parity == p) w.parity = p ^ 1;
}

inline f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    int p = wave_exchange(a, b);
    WaveX& w = my_wave();
    int lane = S().cur->linear & 63;
    int j = lane & 31, half = lane >> 5;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        float acc = c[r];
        acc = fmaf(w.a[p][i], w.b[p][j], acc);            // k = 0
        acc = fmaf(w.a[p][i + 32], w.b[p][j + 32], acc);  // k = 1
        d[r] = acc;
    }
    wave_exchange_done(p);
    return d;
}

// v_mfma_f32_32x32x16_bf16: A[i = l&31][k = 8*(l>>5) + e], B[k = 8*(l>>5) + e][j = l&31] (8 bf16 per lane, element e
// in bits [16*(e&1), +16) of word e>>1), D as the fp32 32x32 forms.  bf16 products are exact in fp32; the sum is
// accumulated in fp32 (the hardware's internal summation order is not modelled -- parity tests carry a tolerance).
inline f32x16 mfma_32x32x16_bf16(emu_u32x4 a, emu_u32x4 b, f32x16 c) {
    WaveX& w = my_wave();
    int lane = S().cur->linear & 63;
    int p = w.parity;
    for (int i = 0; i < 4; ++i) {
        w.xa[p][lane][i] = a[i];
        w.xb[p][lane][i] = b[i];
    }
    wave_barrier(w);
    int j = lane & 31, half = lane >> 5;
    auto bf = [](unsigned word, int e) {
        unsigned u = (e & 1) ? (word & 0xffff0000u) : (word << 16);
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        float acc = c[r];
        for (int kh = 0; kh < 2; ++kh)
            for (int e = 0; e < 8; ++e)
                acc = fmaf(bf(w.xa[p][i + 32 * kh][e >> 1], e), bf(w.xb[p][j + 32 * kh][e >> 1], e), acc);
        d[r] = acc;
    }
    wave_exchange_done(p);
    return d;
}

// v_mfma_f32_32x32x16_f16: same fragment layout as the bf16 form with IEEE half elements.  Half products are exact in fp32.
inline float emu_half_to_float(unsigned short h) {
    _Float16 v;
    memcpy(&v, &h, 2);
    return (float)v;
}
inline unsigned short emu_float_to_half(float x) {      // round to nearest even, like v_cvt_pk_f16_f32
    _Float16 v = (_Float16)x;
    unsigned short h;
    memcpy(&h, &v, 2);
    return h;
}
inline f32x16 mfma_32x32x16_f16(emu_u32x4 a, emu_u32x4 b, f32x16 c) {
    WaveX& w = my_wave();
    int lane = S().cur->linear & 63;
    int p = w.parity;
    for (int i = 0; i < 4; ++i) {
        w.xa[p][lane][i] = a[i];
        w.xb[p][lane][i] = b[i];
    }
    wave_barrier(w);
    int j = lane & 31, half = lane >> 5;
    auto hf = [](unsigned word, int e) { return emu_half_to_float((unsigned short)((e & 1) ? (word >> 16) : (word & 0xffffu))); };
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        float acc = c[r];
        for (int kh = 0; kh < 2; ++kh)
            for (int e = 0; e < 8; ++e)
                acc = fmaf(hf(w.xa[p][i + 32 * kh][e >> 1], e), hf(w.xb[p][j + 32 * kh][e >> 1], e), acc);
        d[r] = acc;
    }
    wave_exchange_done(p);
    return d;
}

// v_mfma_f32_16x16x32_bf16: A[i = l&15][k = 8*(l>>4) + e], B[k = 8*(l>>4) + e][j = l&15], D as the fp32 16x16 form.
inline f32x4 mfma_16x16x32_f16(emu_u32x4 a, emu_u32x4 b, f32x4 c) {
    WaveX& w = my_wave();
    int lane = S().cur->linear & 63;
    int p = w.parity;
    for (int i = 0; i < 4; ++i) {
        w.xa[p][lane][i] = a[i];
        w.xb[p][lane][i] = b[i];
    }
    wave_barrier(w);
    int j = lane & 15, q = lane >> 4;
    auto hf = [](unsigned word, int e) { return emu_half_to_float((unsigned short)((e & 1) ? (word >> 16) : (word & 0xffffu))); };
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * q + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g)
            for (int e = 0; e < 8; ++e)
                acc = fmaf(hf(w.xa[p][i + 16 * g][e >> 1], e), hf(w.xb[p][j + 16 * g][e >> 1], e), acc);
        d[r] = acc;
    }
    wave_exchange_done(p);
    return d;
}

inline f32x4 mfma_16x16x32_bf16(emu_u32x4 a, emu_u32x4 b, f32x4 c) {
    WaveX& w = my_wave();
    int lane = S().cur->linear & 63;
    int p = w.parity;
    for (int i = 0; i < 4; ++i) {
        w.xa[p][lane][i] = a[i];
        w.xb[p][lane][i] = b[i];
    }
    wave_barrier(w);
    int j = lane & 15, q = lane >> 4;
    auto bf = [](unsigned word, int e) {
        unsigned u = (e & 1) ? (word & 0xffff0000u) : (word << 16);
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * q + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g)
            for (int e = 0; e < 8; ++e)
                acc = fmaf(bf(w.xa[p][i + 16 * g][e >> 1], e), bf(w.xb[p][j + 16 * g][e >> 1], e), acc);
        d[r] = acc;
    }
    wave_exchange_done(p);
    return d;
}

inline f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    int p = wave_exchange(a, b);
    WaveX& w = my_wave();
    int lane = S().cur->linear & 63;
    int j = lane & 15, q = lane >> 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * q + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.a[p][i + 16 * k], w.b[p][j + 16 * k], acc);
        d[r] = acc;
    }
    wave_exchange_done(p);
    return d;
}

// ds_read_b64_tr_b16 (gfx950): every lane addresses 8 bytes (4 halves; the hardware ignores address bits 0..2 -- here a
// misaligned address aborts); within each 16-lane group lane i receives element (i & 3) of the pieces addressed by lanes
// 4 r + (i >> 2), r = 0..3: column i of the [4][16] block whose row r the lanes 4 r .. 4 r + 3 address.
inline uint2 lds_read_tr16_b64(const void* addr) {
    if (((uintptr_t)addr & 7u) != 0) {
        fprintf(stderr, "emu: ds_read_b64_tr_b16 at an address that is not 8-byte aligned\n");
        abort();
    }
    WaveX& w = my_wave();
    int lane = S().cur->linear & 63;
    int p = w.parity;
    w.ptr[p][lane] = addr;
    wave_barrier(w);
    const int grp = lane & ~15, i = lane & 15;
    unsigned short v[4];
    for (int r = 0; r < 4; ++r) memcpy(&v[r], (const char*)w.ptr[p][grp + 4 * r + (i >> 2)] + 2 * (i & 3), 2);
    wave_exchange_done(p);
    return make_uint2((unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16));
}

inline float shfl(float v, int src_lane) {
    int p = wave_exchange(v, 0.f);
    float r = my_wave().a[p][src_lane & 63];
    wave_exchange_done(p);
    return r;
}

template <class F>
inline void run_grid(dim3 grid, dim3 block, size_t smem, F&& body) {
    State& s = S();
    s.nthreads = block.x * block.y * block.z;
    if (s.nthreads % 64 != 0 || s.nthreads > 1024) {
        fprintf(stderr, "emu: block size %u unsupported\n", s.nthreads);
        abort();
    }
    s.blockDim_ = block;
    s.gridDim_ = grid;
    s.fibers.resize(s.nthreads);
    if (s.stacks.size() < kStack * s.nthreads) s.stacks.resize(kStack * s.nthreads);
    s.waves.assign(s.nthreads / 64, WaveX());
    s.dyn_smem.assign(smem + 64, 0);
    s.body = body;
    unsigned rot = 0;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                s.blockIdx_ = dim3(bx, by, bz);
                s.wg_arrived = 0;
                for (auto& w : s.waves) { w.arrived = 0; w.parity = 0; }
                for (unsigned t = 0; t < s.nthreads; ++t) {
                    Fiber& f = s.fibers[t];
                    f.linear = t;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.done = false;
                    // initial frame: six zeroed callee-saved slots, then the address fiber_switch "returns" to; the stack pointer at
                    // the trampoline's first instruction is 8 (mod 16), as after a call
                    uintptr_t top = ((uintptr_t)(s.stacks.data() + kStack * (t + 1))) & ~(uintptr_t)15;
                    void** fr = reinterpret_cast<void**>(top - 16 - 6 * sizeof(void*));
                    for (int q = 0; q < 6; ++q) fr[q] = nullptr;
                    fr[6] = reinterpret_cast<void*>(&trampoline);
                    fr[7] = nullptr;
                    f.sp = fr;
                }
                unsigned remaining = s.nthreads;
                while (remaining) {
                    unsigned progressed = 0;
                    // rotate the start wave each sweep so wave ordering assumptions are exposed
                    unsigned nw = s.nthreads / 64;
                    rot = (rot + 1) % nw;
                    for (unsigned wi = 0; wi < nw; ++wi) {
                        unsigned wv = (wi + rot) % nw;
                        for (unsigned l = 0; l < 64; ++l) {
                            Fiber& f = s.fibers[wv * 64 + l];
                            if (f.done) continue;
                            s.cur = &f;
                            fiber_switch(&s.sched_sp, f.sp);
                            progressed++;
                            if (f.done) remaining--;
                        }
                    }
                    if (!progressed) break;
                }
            }
    s.cur = nullptr;
}

}  // namespace emu

#define threadIdx (emu::S().cur->tid)
#define blockIdx (emu::S().blockIdx_)
#define blockDim (emu::S().blockDim_)
#define gridDim (emu::S().gridDim_)

static inline void __syncthreads() { emu::wg_barrier(); }
static inline float __shfl_xor(float v, int mask) { return emu::shfl(v, (emu::S().cur->linear & 63) ^ mask); }
static inline float __shfl_down(float v, int d) {
    int l = emu::S().cur->linear & 63;
    return emu::shfl(v, l + d > 63 ? l : l + d);
}
static inline float __shfl(float v, int src) { return emu::shfl(v, src); }
// (the emulator runs one workgroup at a time and its threads as cooperative fibres: plain read-modify-writes are atomic)
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; *p = o > v ? o : v; return o; }
static inline unsigned atomicExch(unsigned* p, unsigned v) { unsigned o = *p; *p = v; return o; }
static inline void __threadfence() {}
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
using std::max;
using std::min;

// kernel launch: hipLaunchKernelGGL(kernel, grid, block, smem, stream, args...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
    emu::run_grid((grid), (block), (smem), [&]() { kernel(__VA_ARGS__); })
