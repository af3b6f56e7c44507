// ResNet-50 layer 1: conv3 + bn3 + residual add + ReLU of a bottleneck AND conv1 of the NEXT bottleneck in one barrier-free streaming kernel
// (models/resnet.py:104-112 followed by :94-96 of the next block), gfx950.
//
// Layer by layer the block output X' (256 channels at 56 x 56: 4.6 GB per step at the benchmark shape) is written by the fused conv3 kernel
// (adamml_conv_fwd_bn_add) and read back by the next block's conv1 -- and by nothing else in the forward pass except the next residual add.
// In the streaming form of conv1x1_narrow.hip a wave owns ALL channels of its pixels, so the tile of X' it has just produced is a complete
// operand of the next 1 x 1 conv: it stays in the wave's private LDS area and is multiplied with W1 [64][256] right away.  X' is still
// written once (the next add and the backward pass need it), but never re-read for conv1: one 4.6 GB pass and one launch per block pair less.
//
//   per wave and 16-pixel tile:  a2 (64 ch, lazy bn2 + ReLU) --W3--> z3 (2 blocks of 128 ch, MFMA)            [weights W3 in LDS]
//                                out = relu(scale3 z3 + shift3 + value(identity))  -> global (16-byte stores) + 1-bit mask + LDS tile
//                                LDS tile [16][256] --W1--> z1 (64 ch, MFMA)        -> global + sum / sum of squares    [W1 in LDS]
//
// Same K order, rounding points and epilogue expression as conv_gemm_kernel's FADD instance and forward instance: X', the mask and z1 are
// bit-identical to the two-launch form (tests/test_kernels_gpu.py); the statistics of z1 differ in summation order only.
// One 8-wave workgroup per CU (W3 36 KB + W1 33 KB + 8 x 8.25 KB tiles + vectors = 144 KB of LDS), no barrier in the tile loop.
#include <type_traits>
#include "common.h"
#include "../../include/adamml_hip.h"

namespace {

constexpr int C3IN = 64, CB = 256, C1OUT = 64;          // conv3: 64 -> 256; next conv1: 256 -> 64
constexpr int W3ROW = C3IN * 2 + 16;                   // 144
constexpr int W1ROW = CB * 2 + 16;                     // 528
constexpr int SROW = CB * 2 + 16;                      // staging row (16-byte aligned: the next conv's B fragments are ds_read_b128)
constexpr int ZROW = C1OUT * 2 + 8;                    // z1 staging row
constexpr int TPX = 16;
constexpr int NW = 8;                                  // waves per workgroup
constexpr int LDS_BYTES = CB * W3ROW + C1OUT * W1ROW + 2 * C3IN * 4 + 4 * CB * 4 + NW * 2 * C1OUT * 4 + NW * TPX * SROW;

struct FNP {
    const bf16_t* x;         // [groups * P][64] raw conv2 output
    const float* in_scale;   // bn2 scale / shift (lazy), group stride in_gs
    const float* in_shift;
    const bf16_t* w3;        // [256][64]
    const float* bn_vec;     // [groups][4][256]: scale, shift of bn3
    const bf16_t* idn;       // [groups * P][256] identity operand or null
    const float* id_scale;   // lazy identity (downsample branch) or null, group stride id_gs
    const float* id_shift;
    bf16_t* out;             // [groups * P][256]
    uint8_t* mask_out;       // [groups * P][32] or null
    const bf16_t* w1;        // [64][256] next conv1 (forward pack) or null
    bf16_t* y1;              // [groups * P][64]
    double* stats1;          // [groups][SLOTS][128] or null
    int in_act, in_gs, id_gs, act;
    long P;
};

typedef __attribute__((ext_vector_type(4))) short s16x4_;

// ALLFULL: P % 16 == 0 (the launcher's choice) and MASK: mask_out given -- compile-time, so that the tile loop holds no store under a per-lane
// or run-time condition (behind one, every wait is a conservative one: tpool_bwd_prod.hip went 1.74 -> 1.60 ms on that alone; THIS kernel measured the same either way, 2.32 ms)
template <bool NEXT, bool ALLFULL, bool MASK>
__global__ __launch_bounds__(NW * 64, 1) void conv1x1_fadd_next_kernel(FNP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w3 = smem;
    char* s_w1 = s_w3 + CB * W3ROW;
    float* s_vec = reinterpret_cast<float*>(s_w1 + C1OUT * W1ROW);          // [2][64]
    float* s_bn = s_vec + 2 * C3IN;                                          // [4][256]: scale3, shift3, id scale, id shift
    float* s_sum = s_bn + 4 * CB;                                            // [8 waves][128]
    char* s_stage = reinterpret_cast<char*>(s_sum + NW * 2 * C1OUT);         // [8 waves][16][SROW]
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    p.x += (size_t)g * p.P * C3IN;
    p.out += (size_t)g * p.P * CB;
    if (p.idn) p.idn += (size_t)g * p.P * CB;
    if (MASK) p.mask_out += (size_t)g * p.P * (CB / 8);
    if (NEXT) p.y1 += (size_t)g * p.P * C1OUT;
    for (int i = tid; i < CB * (C3IN / 8); i += NW * 64) {
        const int row = i / (C3IN / 8), ch = i - row * (C3IN / 8);
        *reinterpret_cast<bf16x8*>(s_w3 + row * W3ROW + ch * 16) = *reinterpret_cast<const bf16x8*>(p.w3 + (size_t)row * C3IN + ch * 8);
    }
    if (NEXT)
        for (int i = tid; i < C1OUT * (CB / 8); i += NW * 64) {
            const int row = i / (CB / 8), ch = i - row * (CB / 8);
            *reinterpret_cast<bf16x8*>(s_w1 + row * W1ROW + ch * 16) = *reinterpret_cast<const bf16x8*>(p.w1 + (size_t)row * CB + ch * 8);
        }
    for (int i = tid; i < C3IN; i += NW * 64) {
        s_vec[i] = p.in_scale ? p.in_scale[(size_t)g * p.in_gs + i] : 1.f;
        s_vec[C3IN + i] = p.in_scale ? p.in_shift[(size_t)g * p.in_gs + i] : 0.f;
    }
    for (int i = tid; i < CB; i += NW * 64) {
        s_bn[i] = p.bn_vec[(size_t)g * 4 * CB + i];
        s_bn[CB + i] = p.bn_vec[(size_t)g * 4 * CB + CB + i];
        s_bn[2 * CB + i] = p.id_scale ? p.id_scale[(size_t)g * p.id_gs + i] : 1.f;
        s_bn[3 * CB + i] = p.id_scale ? p.id_shift[(size_t)g * p.id_gs + i] : 0.f;
    }
    __syncthreads();
    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float alo = uniform(p.in_scale ? act_lo(p.in_act) : -INFINITY), ahi = uniform(p.in_scale ? act_hi(p.in_act) : INFINITY);
    const float rlo = uniform(act_lo(p.act)), rhi = uniform(act_hi(p.act));
    const bool lazy = p.in_scale != nullptr, has_idn = p.idn != nullptr;
    char* stg = s_stage + wave * (TPX * SROW);
    const long ntile = (p.P + TPX - 1) / TPX;
    const long wid = (long)blockIdx.x * NW + wave, nw = (long)gridDim.x * NW;
    // epilogue geometry of a 128-channel block: lane -> (pixel sub-index epl 0..3, 8-channel chunk ech 0..15), 4 pixels per pass, 4 passes
    const int epl = lane >> 4, ech = lane & 15;
    float ssum[4][4], ssq[4][4];                          // z1 statistics: channel ct * 16 + lg * 4 + r, over this lane's pixels
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { ssum[a][b] = 0.f; ssq[a][b] = 0.f; }

    bf16x8 rx[2], raux[2][4];
    auto issue_x = [&](long tile) {
        long px = tile * TPX + li;
        px = px < p.P ? px : p.P - 1;
#pragma unroll
        for (int k = 0; k < 2; ++k) rx[k] = *reinterpret_cast<const bf16x8*>(p.x + (size_t)px * C3IN + (k * 4 + lg) * 8);
    };
    auto issue_aux = [&](long tile) {
        if (has_idn) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    long px = tile * TPX + i * 4 + epl;
                    px = px < p.P ? px : p.P - 1;
                    raux[b][i] = *reinterpret_cast<const bf16x8*>(p.idn + (size_t)px * CB + b * 128 + ech * 8);
                }
        }
    };
    issue_x(wid < ntile ? wid : ntile - 1);
    issue_aux(wid < ntile ? wid : ntile - 1);

    for (long t = wid; t < ntile; t += nw) {
        const int npx = ALLFULL ? TPX : (int)(p.P - t * TPX < TPX ? p.P - t * TPX : TPX);
        bf16x8 fb[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            fb[k] = rx[k];
            if (lazy) {
                const f32x8 sc = load_f32x8(s_vec + k * 32 + lg * 8), sh = load_f32x8(s_vec + C3IN + k * 32 + lg * 8);
                f32x8 v = bf8_to_f32(fb[k]);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = clamp_act(fmaf(v[i], sc[i], sh[i]), alo, ahi);
                fb[k] = f32_to_bf8(v);
            }
        }
        {
            const long tn = t + nw;
            issue_x(tn < ntile ? tn : ntile - 1);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x4 acc[8];
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(s_w3 + ((b * 8 + ct) * 16 + li) * W3ROW + k * 64 + lg * 16);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[k], acc[ct], 0, 0, 0);
                }
#pragma unroll
            for (int ct = 0; ct < 8; ++ct)
                *reinterpret_cast<bf16x4*>(stg + li * SROW + (b * 128 + ct * 16 + lg * 4) * 2) = f32_to_bf4(acc[ct]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int c0 = b * 128 + ech * 8;
            const f32x8 sc = load_f32x8(s_bn + c0), sh = load_f32x8(s_bn + CB + c0), isc = load_f32x8(s_bn + 2 * CB + c0), ish = load_f32x8(s_bn + 3 * CB + c0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int px = i * 4 + epl;
                char* sp = stg + px * SROW + c0 * 2;
                f32x8 f = bf8_to_f32(*reinterpret_cast<const bf16x8*>(sp));
                if (has_idn) {
                    const f32x8 w = bf8_to_f32(raux[b][i]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = clamp_act(fmaf(f[j], sc[j], sh[j]) + fmaf(w[j], isc[j], ish[j]), rlo, rhi);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = clamp_act(fmaf(f[j], sc[j], sh[j]), rlo, rhi);
                }
                bf16x8 v = f32_to_bf8(f);
                if (!ALLFULL && px >= npx) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};       // (rows past the end: zero operand of the next conv, never stored)
                if (NEXT) *reinterpret_cast<bf16x8*>(sp) = v;            // the block output tile stays in LDS: B operand of the next conv
                if (ALLFULL || px < npx) {
                    const size_t e = ((size_t)t * TPX + px) * CB + c0;
                    *reinterpret_cast<bf16x8*>(p.out + e) = v;
                    if (MASK) {
                        const f32x8 q = bf8_to_f32(v);
                        unsigned bits = 0;
#pragma unroll
                        for (int j = 0; j < 8; ++j) bits |= (q[j] > rlo && q[j] < rhi) ? (1u << j) : 0u;
                        p.mask_out[e >> 3] = (uint8_t)bits;
                    }
                }
            }
        }
        {
            const long tn = t + nw;
            issue_aux(tn < ntile ? tn : ntile - 1);
        }
        if (NEXT) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (the wave's X' tile is complete in LDS)
            f32x4 a1[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) a1[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bf16x8 fx = *reinterpret_cast<const bf16x8*>(stg + li * SROW + (k * 4 + lg) * 16);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(s_w1 + (ct * 16 + li) * W1ROW + (k * 4 + lg) * 16);
                    a1[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fx, a1[ct], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (every fragment read of the tile is done: the area becomes the z1 tile)
            const bool live = ALLFULL || li < npx;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const bf16x4 v = f32_to_bf4(a1[ct]);
                *reinterpret_cast<bf16x4*>(stg + li * ZROW + (ct * 16 + lg * 4) * 2) = v;
                if (p.stats1) {
                    f32x4 q = bf4_to_f32(v);
                    if (!live) q = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ssum[ct][r] += q[r]; ssq[ct][r] += q[r] * q[r]; }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // z1 tile: 16 pixels x 128 bytes = one contiguous run
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = lane + 64 * i, px = e >> 3, ch = e & 7;
                if (ALLFULL || px < npx) {
                    union { struct { s16x4_ a, b; } s; bf16x8 v; } u;
                    u.s.a = *reinterpret_cast<const s16x4_*>(stg + px * ZROW + ch * 16);
                    u.s.b = *reinterpret_cast<const s16x4_*>(stg + px * ZROW + ch * 16 + 8);
                    *reinterpret_cast<bf16x8*>(p.y1 + ((size_t)t * TPX + px) * C1OUT + ch * 8) = u.v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (NEXT && p.stats1) {
        // the 16 lanes of a row hold the same channels: fold the row, then the waves in order, one exact add per channel and workgroup
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    ssum[ct][r] += __shfl_xor(ssum[ct][r], m, 16);
                    ssq[ct][r] += __shfl_xor(ssq[ct][r], m, 16);
                }
                if (li == 0) {
                    s_sum[wave * 128 + ct * 16 + lg * 4 + r] = ssum[ct][r];
                    s_sum[wave * 128 + 64 + ct * 16 + lg * 4 + r] = ssq[ct][r];
                }
            }
        __syncthreads();
        if (tid < 128) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += s_sum[w * 128 + tid];
            stat_publish(p.stats1 + (size_t)g * ADAMML_STAT_SLOTS * 128 + tid, 128, blockIdx.x & (ADAMML_STAT_SLOTS - 1), v);
        }
    }
}


// ---- the same conv3 + bn3 + add + ReLU when the block output feeds ONLY the temporal max-pool (last block of the stage: models/resnet.py
// :205-209 -> models/common.py:4-33, kernel 3 / stride 2 / pad 1 over the T frames of a clip).  A wave takes a (clip, 16-pixel block) task
// and streams the T frames of that block in order; every lane keeps, for its 8 (pixel, 8-channel chunk) slots, the running maximum of the
// open window and the 2-bit tap of its FIRST maximum in registers: an even frame 2 to is tap 1 of window `to`, the odd frame 2 to + 1 is
// tap 2 of window `to` -- which is then complete and stored, pooled value + code -- and tap 0 of window to + 1.  Values, tie rule and the
// "maximum <= 0 -> code 3" rule are conv_gemm_kernel's TP epilogue: pooled and code are bit-identical; the full-rate block output never
// exists in HBM.  conv_gemm's TP instance ran this launch at 3.9 TB/s (its 128-row tile holds 16 pixels x 8 frames: 16 short runs).
struct FTP {
    const bf16_t* x; const float* in_scale; const float* in_shift; const bf16_t* w3; const float* bn_vec;
    const bf16_t* idn; const float* id_scale; const float* id_shift;
    bf16_t* pooled;          // [groups * clips * T / 2][HW][256]
    uint16_t* code;          // [groups * clips * T / 2][HW][32] or null
    int in_act, in_gs, id_gs, act, T, HW, clips;
};

// ALLFULL (HW % 16 == 0) and CODE (codes wanted) are compile-time and the frame loop is unrolled by frame parity, so that the stores of a
// closing window are unconditional code (see conv1x1_fadd_next_kernel)
template <bool ALLFULL, bool CODE>
__global__ __launch_bounds__(NW * 64, 1) void conv1x1_fadd_tpool_kernel(FTP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w3 = smem;
    float* s_vec = reinterpret_cast<float*>(s_w3 + CB * W3ROW);              // [2][64]
    float* s_bn = s_vec + 2 * C3IN;                                          // [4][256]: scale3, shift3, id scale, id shift
    char* s_stage = reinterpret_cast<char*>(s_bn + 4 * CB);                  // [8 waves][16][SROW]
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int To = p.T >> 1;
    {
        const size_t P = (size_t)p.clips * p.T * p.HW, Pp = (size_t)p.clips * To * p.HW;
        p.x += (size_t)g * P * C3IN;
        p.idn += (size_t)g * P * CB;
        p.pooled += (size_t)g * Pp * CB;
        if (CODE) p.code += (size_t)g * Pp * (CB / 8);
    }
    for (int i = tid; i < CB * (C3IN / 8); i += NW * 64) {
        const int row = i / (C3IN / 8), ch = i - row * (C3IN / 8);
        *reinterpret_cast<bf16x8*>(s_w3 + row * W3ROW + ch * 16) = *reinterpret_cast<const bf16x8*>(p.w3 + (size_t)row * C3IN + ch * 8);
    }
    for (int i = tid; i < C3IN; i += NW * 64) {
        s_vec[i] = p.in_scale ? p.in_scale[(size_t)g * p.in_gs + i] : 1.f;
        s_vec[C3IN + i] = p.in_scale ? p.in_shift[(size_t)g * p.in_gs + i] : 0.f;
    }
    for (int i = tid; i < CB; i += NW * 64) {
        s_bn[i] = p.bn_vec[(size_t)g * 4 * CB + i];
        s_bn[CB + i] = p.bn_vec[(size_t)g * 4 * CB + CB + i];
        s_bn[2 * CB + i] = p.id_scale ? p.id_scale[(size_t)g * p.id_gs + i] : 1.f;
        s_bn[3 * CB + i] = p.id_scale ? p.id_shift[(size_t)g * p.id_gs + i] : 0.f;
    }
    __syncthreads();
    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float alo = uniform(p.in_scale ? act_lo(p.in_act) : -INFINITY), ahi = uniform(p.in_scale ? act_hi(p.in_act) : INFINITY);
    const float rlo = uniform(act_lo(p.act)), rhi = uniform(act_hi(p.act));
    const bool lazy = p.in_scale != nullptr;
    char* stg = s_stage + wave * (TPX * SROW);
    const int tpf = (p.HW + TPX - 1) / TPX;                                  // pixel blocks per frame
    const long ntask = (long)p.clips * tpf;
    const long wid = (long)blockIdx.x * NW + wave, nw = (long)gridDim.x * NW;
    const int epl = lane >> 4, ech = lane & 15;

    bf16x8 rx[2], raux[2][4];
    // frame t of task `task`: first pixel (within the group) and the number of live pixels of the block
    auto frame_base = [&](long task, int t, int& npx) {
        const int clip = (int)(task / tpf), blk = (int)(task - (long)clip * tpf);
        npx = p.HW - blk * TPX < TPX ? p.HW - blk * TPX : TPX;
        return ((long)clip * p.T + t) * p.HW + (long)blk * TPX;
    };
    auto issue = [&](long task, int t) {
        int npx;
        const long base = frame_base(task, t, npx);
        const long px = base + (li < npx ? li : npx - 1);
#pragma unroll
        for (int k = 0; k < 2; ++k) rx[k] = *reinterpret_cast<const bf16x8*>(p.x + (size_t)px * C3IN + (k * 4 + lg) * 8);
    };
    auto issue_aux = [&](long task, int t) {
        int npx;
        const long base = frame_base(task, t, npx);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = i * 4 + epl;
                raux[b][i] = *reinterpret_cast<const bf16x8*>(p.idn + (size_t)(base + (q < npx ? q : npx - 1)) * CB + b * 128 + ech * 8);
            }
    };
    {
        const long t0 = wid < ntask ? wid : ntask - 1;
        issue(t0, 0);
        issue_aux(t0, 0);
    }
    for (long task = wid; task < ntask; task += nw) {
        const int clip = (int)(task / tpf), blk = (int)(task - (long)clip * tpf);
        const int npx = ALLFULL ? TPX : (p.HW - blk * TPX < TPX ? p.HW - blk * TPX : TPX);
        f32x8 best[2][4];
        unsigned code[2][4];
        auto frame = [&](int t, auto odd_c) {
            constexpr bool ODD = decltype(odd_c)::value;
            __builtin_amdgcn_sched_barrier(0);
            // the (task, frame) after this one, clamped to the last: its rows are requested while this frame is computed
            long ntask_ = task;
            int nt = t + 1;
            if (nt == p.T) { nt = 0; ntask_ = task + nw < ntask ? task + nw : task; if (ntask_ == task) nt = p.T - 1; }
            bf16x8 fb[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                fb[k] = rx[k];
                if (lazy) {
                    const f32x8 sc = load_f32x8(s_vec + k * 32 + lg * 8), sh = load_f32x8(s_vec + C3IN + k * 32 + lg * 8);
                    f32x8 v = bf8_to_f32(fb[k]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = clamp_act(fmaf(v[i], sc[i], sh[i]), alo, ahi);
                    fb[k] = f32_to_bf8(v);
                }
            }
            issue(ntask_, nt);
            constexpr unsigned tapbits = ODD ? 0xAAAAu : 0x5555u;            // this frame's tap in the open window: 2 (odd frame) or 1
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x4 acc[8];
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int ct = 0; ct < 8; ++ct) {
                        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(s_w3 + ((b * 8 + ct) * 16 + li) * W3ROW + k * 64 + lg * 16);
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[k], acc[ct], 0, 0, 0);
                    }
#pragma unroll
                for (int ct = 0; ct < 8; ++ct)
                    *reinterpret_cast<bf16x4*>(stg + li * SROW + (b * 128 + ct * 16 + lg * 4) * 2) = f32_to_bf4(acc[ct]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int c0 = b * 128 + ech * 8;
                const f32x8 sc = load_f32x8(s_bn + c0), sh = load_f32x8(s_bn + CB + c0), isc = load_f32x8(s_bn + 2 * CB + c0), ish = load_f32x8(s_bn + 3 * CB + c0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int px = i * 4 + epl;
                    f32x8 f = bf8_to_f32(*reinterpret_cast<const bf16x8*>(stg + px * SROW + c0 * 2));
                    const f32x8 w = bf8_to_f32(raux[b][i]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = clamp_act(fmaf(f[j], sc[j], sh[j]) + fmaf(w[j], isc[j], ish[j]), rlo, rhi);
                    const f32x8 v = bf8_to_f32(f32_to_bf8(f));              // the value the unfused path stores and the pool re-reads
                    if (!ODD && t == 0) {                                   // window 0 has no tap 0: the scan starts at tap 1
                        best[b][i] = v;
                        code[b][i] = 0x5555u;
                    } else {
                        unsigned cd = code[b][i];
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (v[j] > best[b][i][j]) { best[b][i][j] = v[j]; cd = (cd & ~(3u << (2 * j))) | (tapbits & (3u << (2 * j))); }   // first maximum in scan order
                        code[b][i] = cd;
                    }
                    if constexpr (ODD) {
                        // window t >> 1 is complete
                        if (ALLFULL || px < npx) {
                            unsigned cd = code[b][i];
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (!(best[b][i][j] > rlo && best[b][i][j] < rhi)) cd |= 3u << (2 * j);       // act'(maximum) == 0: no gradient through this window
                            const size_t po = (((size_t)clip * To + (t >> 1)) * p.HW + (size_t)blk * TPX + px) * CB + c0;
                            *reinterpret_cast<bf16x8*>(p.pooled + po) = f32_to_bf8(best[b][i]);
                            if (CODE) p.code[po >> 3] = (uint16_t)cd;
                        }
                        best[b][i] = v;                                     // tap 0 of the next window
                        code[b][i] = 0u;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (the staged block is consumed before the next one overwrites it)
            }
            issue_aux(ntask_, nt);
        };
#pragma unroll 1
        for (int t = 0; t < p.T; t += 2) {
            frame(t, std::false_type{});
            frame(t + 1, std::true_type{});
        }
    }
}

}  // namespace

// d: the forward descriptor of conv3 (1x1, 64 -> 256); next_cout: output channels of the next block's conv1 (64) or 0
bool adamml_conv1x1_fadd_next_supported(const adamml_conv_desc_t* d, int next_cout) {
    static const int on = getenv("ADAMML_FADD_NEXT") ? atoi(getenv("ADAMML_FADD_NEXT")) : 1;            // A/B aid
    return on && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->up <= 1 && d->Cin == C3IN && d->Cout == CB &&
           (next_cout == 0 || next_cout == C1OUT);
}

int adamml_conv1x1_fadd_next_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                                    const float* bn_vec, const void* idn, const float* id_scale, const float* id_shift, int id_gstride, int act,
                                    void* out, uint8_t* mask_out, const void* w1_packed, void* y1, double* stats1, hipStream_t stream) {
    FNP p;
    p.x = (const bf16_t*)x; p.in_scale = in_scale; p.in_shift = in_scale ? in_shift : nullptr; p.w3 = (const bf16_t*)w_packed; p.bn_vec = bn_vec;
    p.idn = (const bf16_t*)idn; p.id_scale = idn ? id_scale : nullptr; p.id_shift = idn ? id_shift : nullptr;
    p.out = (bf16_t*)out; p.mask_out = mask_out; p.w1 = (const bf16_t*)w1_packed; p.y1 = (bf16_t*)y1; p.stats1 = stats1;
    p.in_act = d->act; p.in_gs = d->in_gstride; p.id_gs = id_gstride; p.act = act;
    p.P = (long)d->N * d->H * d->W;
    if (p.P <= 0) return ADAMML_OK;
    const int groups = d->groups < 1 ? 1 : d->groups;
    const long ntile = (p.P + TPX - 1) / TPX;
    long nblk = (ntile + NW - 1) / NW;
    long cap = 256 / groups;                                             // one workgroup per CU over all groups
    if (cap < 1) cap = 1;
    if (nblk > cap) nblk = cap;
    const dim3 grid((unsigned)nblk, groups);
    const int attr_dev = adamml_current_device();
    auto launch = [&](auto next_c, auto full_c, auto mask_c) -> int {
        constexpr bool NEXT = decltype(next_c)::value, FULL = decltype(full_c)::value, MASK = decltype(mask_c)::value;
        static AdamLdsOnce attr_once;                // (per device and instance: common.h)
        if (!attr_once.test(attr_dev)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_fadd_next_kernel<NEXT, FULL, MASK>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "conv_fwd_bn_add_next: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
            attr_once.set(attr_dev);
        }
        hipLaunchKernelGGL((conv1x1_fadd_next_kernel<NEXT, FULL, MASK>), grid, dim3(NW * 64), LDS_BYTES, stream, p);
        return 0;
    };
    auto pick_mask = [&](auto next_c, auto full_c) -> int {
        return mask_out ? launch(next_c, full_c, std::true_type{}) : launch(next_c, full_c, std::false_type{});
    };
    auto pick_full = [&](auto next_c) -> int {
        return p.P % TPX == 0 ? pick_mask(next_c, std::true_type{}) : pick_mask(next_c, std::false_type{});
    };
    const int lrc = w1_packed ? pick_full(std::true_type{}) : pick_full(std::false_type{});
    if (lrc) return lrc;
    return adamml_check_launch("conv_fwd_bn_add_next");
}

// d: the forward descriptor of conv3 (1x1, 64 -> 256, N = clips * frames images per group)
bool adamml_conv1x1_fadd_tpool_supported(const adamml_conv_desc_t* d, int frames) {
    static const int on = getenv("ADAMML_FADD_TPOOL_STREAM") ? atoi(getenv("ADAMML_FADD_TPOOL_STREAM")) : 1;      // A/B aid
    return on && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->up <= 1 && d->Cin == C3IN && d->Cout == CB &&
           (frames == 2 || frames == 4 || frames == 8) && d->N % frames == 0;
}

int adamml_conv1x1_fadd_tpool_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                                     const float* bn_vec, const void* idn, const float* id_scale, const float* id_shift, int id_gstride, int act,
                                     int frames, void* pooled, uint16_t* code, hipStream_t stream) {
    FTP p;
    p.x = (const bf16_t*)x; p.in_scale = in_scale; p.in_shift = in_scale ? in_shift : nullptr; p.w3 = (const bf16_t*)w_packed; p.bn_vec = bn_vec;
    p.idn = (const bf16_t*)idn; p.id_scale = id_scale; p.id_shift = id_scale ? id_shift : nullptr;
    p.pooled = (bf16_t*)pooled; p.code = code;
    p.in_act = d->act; p.in_gs = d->in_gstride; p.id_gs = id_gstride; p.act = act;
    p.T = frames; p.HW = d->H * d->W; p.clips = d->N / frames;
    if (p.clips <= 0 || p.HW <= 0) return ADAMML_OK;
    const int groups = d->groups < 1 ? 1 : d->groups;
    constexpr int LDS_TP = CB * W3ROW + 2 * C3IN * 4 + 4 * CB * 4 + NW * TPX * SROW;
    const long ntask = (long)p.clips * ((p.HW + TPX - 1) / TPX);
    long nblk = (ntask + NW - 1) / NW;
    long cap = 256 / groups;                                             // one workgroup per CU over all groups
    if (cap < 1) cap = 1;
    if (nblk > cap) nblk = cap;
    const dim3 grid((unsigned)nblk, groups);
    const int attr_dev = adamml_current_device();
    auto launch = [&](auto full_c, auto code_c) -> int {
        constexpr bool FULL = decltype(full_c)::value, CODE = decltype(code_c)::value;
        static AdamLdsOnce attr_once;                // (per device and instance: common.h)
        if (!attr_once.test(attr_dev)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_fadd_tpool_kernel<FULL, CODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TP);
            if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "conv_fwd_bn_add_tpool: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
            attr_once.set(attr_dev);
        }
        hipLaunchKernelGGL((conv1x1_fadd_tpool_kernel<FULL, CODE>), grid, dim3(NW * 64), LDS_TP, stream, p);
        return 0;
    };
    auto pick = [&](auto full_c) -> int { return code ? launch(full_c, std::true_type{}) : launch(full_c, std::false_type{}); };
    const int lrc = p.HW % TPX == 0 ? pick(std::true_type{}) : pick(std::false_type{});
    if (lrc) return lrc;
    return adamml_check_launch("conv_fwd_bn_add_tpool");
}
