// Platform layer of the F-FNO kernels: gfx950 (CDNA4) through HIP.  Everything the kernel sources need from the
// compiler / ISA is named here once: vector types, the MFMA builtins, the few bit-field instructions, scheduling
// pragmas, dynamic-LDS plumbing.  The kernel sources include it as <ffno_platform.h> (include-path lookup), so a test
// harness can put a different implementation of the SAME names first on the include path (tests/emu/ does, to run
// the kernels lane by lane on the CPU); no such switch exists in this tree.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define FFNO_BUILD_TARGET "gfx950"
#define FFNO_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define FFNO_UNROLL _Pragma("unroll")
#define FFNO_NOUNROLL _Pragma("unroll 1")
// bounds live ranges: stops the scheduler from hoisting a whole unrolled loop's operand loads
#define FFNO_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// scheduling hint: the next `size` instructions of class `mask` (0x8 MFMA, 0x2 VALU, 0x100 DS read, ...) form one group; a
// sequence of such groups asks the scheduler to emit that pattern (llvm.amdgcn.sched.group.barrier)
#define FFNO_SCHED_GROUP(mask, size) __builtin_amdgcn_sched_group_barrier(mask, size, 0)
// scheduling barriers that only ONE class of instruction may not cross (llvm.amdgcn.sched.barrier mask = the classes that MAY:
// 0x1 other ALU, 0x2 VALU, 0x4 SALU, 0x8 MFMA, 0x10 / 0x20 / 0x40 VMEM all / read / write, 0x80 / 0x100 / 0x200 DS all / read /
// write): keeps a prefetch where it was written without fencing the arithmetic around it
#define FFNO_SCHED_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x38F)
#define FFNO_SCHED_PIN_DSREAD() __builtin_amdgcn_sched_barrier(0x27F)
// pins a register value to this point of the instruction stream: whatever computes x is emitted before, whatever uses it after
// (an empty volatile asm; FFNO_SCHED_FENCE alone orders memory operations but not pure arithmetic, which instruction selection
// is free to emit anywhere in the basic block)
#define FFNO_PIN(x) asm volatile("" : "+v"(x))
// drains every outstanding memory operation of the wave (s_waitcnt 0).  Placed in front of a software-pipelined loop so that the
// compiler's counter analysis starts the loop with nothing pending (else it re-waits for the prologue's loads every iteration)
#define FFNO_DRAIN_MEMORY() __builtin_amdgcn_s_waitcnt(0)
// register budget = 512 / n VGPRs per lane, so that n waves (n/2 workgroups of 512 threads) share a SIMD
#define FFNO_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))

namespace ffno {
namespace plat {

// v_mfma_f32_32x32x2_f32
__device__ __forceinline__ f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x4_f32
__device__ __forceinline__ f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_32x32x16_bf16 (8 bf16 per lane and operand, packed in four dwords)
__device__ __forceinline__ f32x16 mfma_bf16_32x32x16(u32x4 a, u32x4 b, f32x16 c) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_bf16
__device__ __forceinline__ f32x4 mfma_bf16_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// v_mfma_f32_32x32x16_f16 (8 passes, the rate of the bf16 form)
__device__ __forceinline__ f32x16 mfma_f16_32x32x16(u32x4 a, u32x4 b, f32x16 c) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_f16 (4 passes): A[i = l & 15][k = 8 (l >> 4) + e], B[k = 8 (l >> 4) + e][j = l & 15], D reg r = D[4 (l >> 4) + r][l & 15]
__device__ __forceinline__ f32x4 mfma_f16_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// two floats -> one word of IEEE halves, round to nearest even (x0 in the low half): v_cvt_pk_f16_f32
__device__ __forceinline__ unsigned pack_f16(float x0, float x1) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 v = {(_Float16)x0, (_Float16)x1};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float f16_lo(unsigned w) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    return (float)__builtin_bit_cast(f16x2, w)[0];
}
__device__ __forceinline__ float f16_hi(unsigned w) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    return (float)__builtin_bit_cast(f16x2, w)[1];
}
// both halves of a packed-fp16 word times k (a power of two here: exact): v_pk_mul_f16
__device__ __forceinline__ unsigned pk_mul_f16(unsigned w, float k) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 v = __builtin_bit_cast(f16x2, w) * (_Float16)k;
    return __builtin_bit_cast(unsigned, v);
}
// split-fp16 of a pair (ffno_device.h "split-fp16"): h = fp16(x), l = fp16((x - h) 2^11), both rounded to nearest, as packed
// words.  Written over two-element vectors so that the pair stays on the packed instructions wherever it is inlined
// (v_cvt_pk_f16_f32, v_pk_add_f32, v_pk_mul_f32: six instructions per pair; written element by element the same code came out as
// twelve scalar ones inside the feed-forward epilogues -- and those kernels issue ten vector instructions per MFMA).
__device__ __forceinline__ void split2_pair_pk(float x0, float x1, unsigned& h, unsigned& l) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 x = {x0, x1};
    const f16x2 hv = __builtin_convertvector(x, f16x2);
    const f32x2 r = (x - __builtin_convertvector(hv, f32x2)) * 2048.f;
    h = __builtin_bit_cast(unsigned, hv);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
// The same split in FIVE instructions: v_cvt_pk_f16_f32, two v_mul_f32 (2^11 x) and the mixed-precision fma that reads h's halves
// as fp16 operands and writes its fp32 result ROUNDED TO fp16 into one half of the destination (v_fma_mixlo / mixhi_f16):
//     l = fp16(fma(h, -2^11, 2^11 x))  ==  fp16((x - h) 2^11)      (both arguments of the rounding are exact: bit-identical)
// The compiler does not select these by itself (its SLP pass packs the pair into v_pk_fma_f32 first), so they are inline
// assembly -- which the scheduler places as a block: a kernel takes this form only where that was measured to pay (the
// weight-gradient tile loop with both epilogues merged: ffx.hip; MI355X round 5: -3 %; the same loop in its round-4 order: +6 %).
__device__ __forceinline__ void split2_pair_mix(float x0, float x1, unsigned& h, unsigned& l) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 hv = {(_Float16)x0, (_Float16)x1};
    h = __builtin_bit_cast(unsigned, hv);
    const float y0 = x0 * 2048.f, y1 = x1 * 2048.f;
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(h), "s"(-2048.f), "v"(y0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(h), "s"(-2048.f), "v"(y1));
    l = lo;
}
// (what every kernel but that loop uses: the whole step ran 183.6 / 184.1 steps/s with this form everywhere, 183.9 with the other)
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& h, unsigned& l) { split2_pair_pk(x0, x1, h, l); }
// two floats -> one word of bf16, round to nearest even (x0 in the low half): v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16(float x0, float x1) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = {(__bf16)x0, (__bf16)x1};
    return __builtin_bit_cast(unsigned, v);
}
// upper halves of two words -> one word (u0's in the low half): v_perm_b32
__device__ __forceinline__ unsigned pack_hi16(unsigned u0, unsigned u1) { return __builtin_amdgcn_perm(u1, u0, 0x07060302u); }
// lower halves of two words -> one word (u0's in the low half): v_perm_b32
__device__ __forceinline__ unsigned pack_lo16(unsigned u0, unsigned u1) { return __builtin_amdgcn_perm(u1, u0, 0x05040100u); }
// lanes 32..63 of a <-> lanes 0..31 of b (v_permlane32_swap): afterwards a lane of the lower half holds (its a, its partner's a),
// a lane of the upper half (its partner's b, its b) -- every lane of the wave must execute it
__device__ __forceinline__ void swap_halves(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0], b = r[1];
}
// ds_read_b64_tr_b16, the LDS transpose read of gfx950.  Every lane gives the address of 8 bytes (4 halves, 8-byte aligned); a
// 16-lane group's 16 x 4 halves are taken as a [4 rows][16 columns] block -- row r = the 8-byte pieces of lanes 4 r .. 4 r + 3 of
// the group -- and lane i of the group receives COLUMN i: halves (row 0, i), (row 1, i), (row 2, i), (row 3, i), i.e. from the
// piece of lane 4 r + (i >> 2) its element i & 3.  (tests/test_kernels_ffh.py checks the hardware against this map.)
__device__ __forceinline__ uint2 lds_read_tr16_b64(const void* p) {
    typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 f16x4;
    typedef __attribute__((address_space(3))) f16x4* lds_f16x4_p;
    const f16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f16x4_p)(p));
    return __builtin_bit_cast(uint2, v);
}
// (acc << 1) | msb(x): v_alignbit_b32
__device__ __forceinline__ uint32_t shift_in_msb(uint32_t acc, uint32_t x) { return __builtin_amdgcn_alignbit(acc, x, 31); }
// all-ones if bit b of x is set, else 0: v_bfe_i32
__device__ __forceinline__ uint32_t bit_to_mask(uint32_t x, int b) { return (uint32_t)__builtin_amdgcn_sbfe((int)x, b, 1); }
// park the wave for about n cycles (s_sleep: 64-cycle units)
__device__ __forceinline__ void sleep_cycles(int n) {
    for (int i = 0; i < n; i += 64 * 16) __builtin_amdgcn_s_sleep(16);
}
// sin / cos of pi * x
__device__ __forceinline__ void sincos_pi(float x, float& s, float& c) { sincospif(x, &s, &c); }

// compute units of the current device: an immutable fact per device ordinal, queried once (the only state the library keeps
// besides its constant tables)
static inline int cu_count() {
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cached[dev];
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return n;
}

// orders the LDS operations of ONE wave around an exchange through LDS between its own lanes: the hardware executes a wave's
// LDS operations in order, so this only has to stop the compiler from moving them across (no instruction is emitted)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// ---- what the persistent whole-forward kernel (infer_stack_kernel) needs beyond a launch ---------------------------------------
// the XCD (accelerator complex die, 0..7 on MI355X) this wave runs on: workgroups of one XCD share its L2, which is coherent
static constexpr bool kPersistentLaunch = true;
// The persistent kernel's group barriers need all its workgroups resident at once.  Its grid is one workgroup per CU of a kernel
// that fits once per CU (LDS), launched behind the stream's previous kernel: an ordinary launch places them all at once.  (The first
// form went through hipLaunchCooperativeKernel, which checks exactly that -- and makes rocprofv3 of ROCm 7.2 crash at process exit
// after it has traced such a launch, data intact.  What the check bought is covered anyway: the occupancy query below refuses a
// kernel that does not fit, and a workgroup that is late or lost -- a device shared with another process -- ends in the barrier
// time-out, the error word and the caller's fallback to the per-layer launches.)
static inline int launch_cooperative(const void* fn, dim3 grid, dim3 block, void** args, size_t smem, hipStream_t st) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)block.x, smem) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        return -2;
    }
    if ((int)grid.x > per_cu * cu_count()) return -2;
    return (int)hipLaunchKernel(fn, grid, block, args, smem, st);
}
// the device's constant-rate clock (100 MHz on MI355X: s_memrealtime), for the diagnostic trace of the persistent kernel
__device__ __forceinline__ unsigned long long realtime() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ int xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return (int)(v & 15u);
}
// Barrier among the workgroups that share `cnt` -- ALL ON ONE XCD: every thread of every member calls it; the call returns once
// cnt >= target (each member adds 1 per call, so target = members x number of the call).  Same-XCD visibility by hand instead of
// agent-scope fences (which write the whole L2 back: 15 us, tools/ubench/group_barrier.hip; this: 2.3 us): a wave's global stores
// are write-through to the XCD's L2 and complete when vmcnt reaches 0; readers drop what their CU's vector L1 and the scalar
// cache hold.  Returns false if the members did not arrive within ~0.1 s (a lost member must not hang the device): the caller
// leaves the kernel; *err counts such exits.
__device__ __forceinline__ bool group_sync(unsigned* cnt, unsigned target, unsigned* err, int* flag_lds) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 21)) {
                atomicAdd(err, 1u);
                ok = 0;
                break;
            }
        }
        asm volatile("buffer_inv sc0\n\ts_dcache_inv\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        *flag_lds = ok;
    }
    __syncthreads();
    return *flag_lds != 0;
}
}  // namespace plat

// dynamic LDS beyond the default 48 KiB window needs an explicit opt-in per kernel AND device
// (remembered per (device, kernel, size): the runtime call costs tens of microseconds of host time, more than the launch it
//  precedes -- a per-launch call makes a stream of 20-us kernels host-bound.  A process that drives a second device gets its own
//  slots there: ADVICE r04.)
template <class K>
static inline int allow_dynamic_lds(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return 0;
    struct Slot {
        std::atomic<const void*> fn{nullptr};
        std::atomic<size_t> bytes{0};
    };
    constexpr int kDevs = 16, kSlots = 64;
    static Slot slots[kDevs][kSlots];
    int dev = 0;
    const void* p = (const void*)kernel;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDevs)      // unknown device: no cache, always ask the runtime
        return (int)hipFuncSetAttribute(p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    Slot* mine = nullptr;
    for (Slot& s : slots[dev]) {
        const void* f = s.fn.load(std::memory_order_acquire);
        if (f == p) {
            if (s.bytes.load(std::memory_order_acquire) >= bytes) return 0;
            mine = &s;
            break;
        }
        if (!f) {
            const void* expect = nullptr;
            if (s.fn.compare_exchange_strong(expect, p) || expect == p) {
                mine = &s;
                break;
            }
        }
    }
    const int rc = (int)hipFuncSetAttribute(p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (rc == 0 && mine) mine->bytes.store(bytes, std::memory_order_release);
    return rc;
}

}  // namespace ffno
