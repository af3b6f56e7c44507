// Shared device helpers for libadamml_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU6 = 2 };

#define ADAMML_OK 0
#define ADAMML_EINVAL (-1)
#define ADAMML_EUNSUPPORTED (-2)
#define ADAMML_ELAUNCH (-3)

// error plumbing (api.hip)
int adamml_set_error(int code, const char* fmt, ...);
int adamml_check_launch(const char* what);
// dw[perm(i)] += sum_{s<nsplit} ws[s*n + i]; taps > 1: ws is [co][tap][cin], dw [co][cin][tap]   (conv_gemm.hip)
int adamml_launch_split_reduce(const float* ws, float* dw, size_t n, int nsplit, hipStream_t stream, int taps = 1, int cin = 1);
int adamml_launch_split_reduce_grouped(const float* ws, float* out, size_t n, int nsplit, int groups, int cin, hipStream_t stream);

// Activations as a clamp to [lo, hi] with wave-uniform bounds (none: [-inf, inf], ReLU: [0, inf], ReLU6: [0, 6]):
// branch-free per element (the runtime `act` is folded into two scalars once per call site).
__device__ __forceinline__ float act_lo(int act) { return act == ACT_NONE ? -INFINITY : 0.f; }
__device__ __forceinline__ float act_hi(int act) { return act == ACT_RELU6 ? 6.f : INFINITY; }
__device__ __forceinline__ float clamp_act(float v, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(v, lo), hi); }
__device__ __forceinline__ float apply_act(float v, int act) { return clamp_act(v, act_lo(act), act_hi(act)); }
// derivative mask of the activation evaluated at pre-activation value v
__device__ __forceinline__ float mask_act(float v, float lo, float hi) { return (v > lo && v < hi) ? 1.f : 0.f; }
__device__ __forceinline__ float act_mask(float v, int act) { return mask_act(v, act_lo(act), act_hi(act)); }

__device__ __forceinline__ f32x8 bf8_to_f32(bf16x8 v) { return __builtin_convertvector(v, f32x8); }
__device__ __forceinline__ bf16x8 f32_to_bf8(f32x8 v) { return __builtin_convertvector(v, bf16x8); }
__device__ __forceinline__ f32x4 bf4_to_f32(bf16x4 v) { return __builtin_convertvector(v, f32x4); }
__device__ __forceinline__ bf16x4 f32_to_bf4(f32x4 v) { return __builtin_convertvector(v, bf16x4); }

__device__ __forceinline__ f32x8 load_f32x8(const float* p) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    f32x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

// lazily-normalised activation read: a = act(scale*z + shift) (scale == nullptr -> identity)
__device__ __forceinline__ f32x8 transform8(bf16x8 raw, const float* scale, const float* shift, int c, int act) {
    f32x8 v = bf8_to_f32(raw);
    if (scale) {
        f32x8 s = load_f32x8(scale + c), t = load_f32x8(shift + c);
        const float lo = act_lo(act), hi = act_hi(act);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = clamp_act(fmaf(v[i], s[i], t[i]), lo, hi);
    }
    return v;
}

__device__ __forceinline__ f32x4 transform4(bf16x4 raw, const float* scale, const float* shift, int c, int act) {
    f32x4 v = bf4_to_f32(raw);
    if (scale) {
        f32x4 s = *reinterpret_cast<const f32x4*>(scale + c), t = *reinterpret_cast<const f32x4*>(shift + c);
        const float lo = act_lo(act), hi = act_hi(act);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = clamp_act(fmaf(v[i], s[i], t[i]), lo, hi);
    }
    return v;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: raise it once per (kernel, device), thread-safely.
// `done` = the call site's own bit set (one bit per device).  A process-wide `static bool` (rounds 4-5) left the limit at 64 KB on every
// device but the first one a process drives, and was a data race between threads (round-5 advisor finding).
struct AdamLdsOnce {
    unsigned long long done = 0;
    bool test(int dev) const { return dev >= 0 && dev < 64 && (__atomic_load_n(&done, __ATOMIC_ACQUIRE) >> dev) & 1ull; }
    void set(int dev) { if (dev >= 0 && dev < 64) __atomic_fetch_or(&done, 1ull << dev, __ATOMIC_RELEASE); }
};
static inline int adamml_current_device() {
    int dev = -1;
    return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}

// ---- reproducible per-channel reductions (BatchNorm statistics, BatchNorm-backward sums) ---------------------------------------
// A sum over millions of pixels is accumulated in two stages, and both are ORDER-FIXED, so that two runs on the same input produce
// the same bits (a 1-ulp difference of a BatchNorm scale is amplified by the bf16 rounding downstream into visibly different
// activations and gradients):
//  (1) inside a workgroup every (wave or row slot, channel) partial has exactly ONE owner lane that adds into a private LDS entry in
//      tile order; at the end of the workgroup one thread per channel folds those entries in index order (no LDS atomics between
//      waves, whose arrival order varies from run to run);
//  (2) across workgroups the fp32 workgroup partial is added EXACTLY into integer bins: the 32 slots of one accumulator
//      ([ADAMML_STAT_SLOTS][2C] doubles, 8 bytes each) are read as 32 int64 bins; an addend +-m * 2^(E-150) (24-bit m, biased exponent
//      E) goes to bin E >> 3 as +-(m << (E & 7)) with ONE native 64-bit integer atomic.  Integer addition is associative, so the bins
//      -- and the fp64 value decoded from them in a fixed order -- do not depend on the order of arrival.  (|addend| < 2^31 per bin:
//      2^32 addends before an int64 bin can overflow.)  One atomic per channel and WORKGROUP: measured on the benchmark step the same
//      cost as the fp64 slot atomics it replaces (round 3 issued det_add per lane and tile: 1.68x).
// There is no other mode and no switch: the library keeps no mutable state (round 3 had a process-global flag selecting fp64 atomics
// spread over the 32 slots instead; the A/B of the two forms is profiles/r04_bench_deterministic.json: 114.6 ms either way).

__device__ __forceinline__ void det_add(double* acc, size_t slot_stride, float v) {
    const unsigned u = __float_as_uint(v);
    unsigned e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
    if (e == 0xffu) {                                    // inf / nan: poison the accumulator, as a floating-point sum would be --
        // bin 31 (exponents >= 2^121: no finite statistic lives there) decodes to NaN as soon as it is non-zero (det_bin_value), so
        // a diverged run shows up in the BatchNorm vectors and the loss instead of being normalised with the finite remainder
        atomicAdd(reinterpret_cast<unsigned long long*>(acc + (size_t)31 * slot_stride), 1ull);
        return;
    }
    if (e) m |= 0x800000u; else e = 1;                   // denormals: no implicit bit, exponent of the smallest normal
    if (!m) return;
    long long c = (long long)m << (e & 7);
    if (u >> 31) c = -c;
    atomicAdd(reinterpret_cast<unsigned long long*>(acc + (size_t)(e >> 3) * slot_stride), (unsigned long long)c);
}

// stage (2): workgroup partial v of one channel -> its accumulator (acc = entry of slot / bin 0, entries slot_stride doubles apart)
// (`slot`: the publishing workgroup's slot of the pre-round-4 layout; unused -- every workgroup adds into the same 32 bins)
__device__ __forceinline__ void stat_publish(double* acc, size_t slot_stride, unsigned slot, float v) {
    (void)slot;
    det_add(acc, slot_stride, v);
}

// value of bin k of a deterministic accumulator (the 32 bin values are summed in a fixed order by the consumers)
__device__ __forceinline__ double det_bin_value(const double* acc, size_t slot_stride, int k) {
    const long long b = *reinterpret_cast<const long long*>(acc + (size_t)k * slot_stride);
    if (k == 31 && b != 0) return __builtin_nan("");     // poisoned by an inf / nan addend (det_add)
    return scalbn((double)b, 8 * k - 150);
}

__device__ __forceinline__ double det_decode(const double* acc, size_t slot_stride) {
    double s = 0.0;
    for (int k = 0; k < 32; ++k) s += det_bin_value(acc, slot_stride, k);
    return s;
}

// overwrite a deterministic accumulator with the fp64 value r (single thread): r = f1 + f2 + f3 exactly (3 x 24 bits >= 53)
__device__ __forceinline__ void det_encode(double* acc, size_t slot_stride, double r) {
    for (int k = 0; k < 32; ++k) acc[(size_t)k * slot_stride] = 0.0;          // (all-zero bits == integer 0)
    for (int i = 0; i < 3; ++i) {
        const float f = (float)r;
        det_add(acc, slot_stride, f);
        r -= (double)f;
    }
}

// Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with a private L2.  This bijection gives XCD k
// the k-th CONTIGUOUS eighth of the logical work list, in dispatch order, so neighbouring tiles (shared halo rows,
// shared weight tiles) meet in one L2 instead of being fetched from HBM by up to 8 of them.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned lin, unsigned total) {
    const unsigned q = total >> 3, r = total & 7, xcd = lin & 7, idx = lin >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
