// Barrier-free streaming forward of the NARROW 1x1 convolutions of the MobileNetV2 backbones (models/policy_net.py:63-95,
// models/sound_mobilenet_v2.py:43-69: the expansion convs 16 -> 96, 24 -> 144, 32 -> 192 and the projection convs 32 -> 16, 96 -> 24,
// 144 -> 24 / 32, 192 -> 32 on the 128^2 .. 20^2 maps, where the bytes of these networks are), gfx950.
//
// conv_gemm_kernel walks 128-pixel tiles with two workgroup barriers per K step and an LDS-staged epilogue behind two more: with K = 16..192
// a tile is ONE to six K steps, so the fixed cost of a tile dominates and these layers ran at 3.4-3.8 TB/s (round 5,
// tools/launch_table.py: 7.3 ms of conv_fwd per step over the three MobileNetV2 forward passes), their cout tiles of 64 also splitting a
// 96- / 144- / 192-channel pixel row over two or three workgroups.  Here, as in alg_stream_kernel (conv1x1_stream.hip):
//   * the activation operand never touches LDS: the B fragment "lane (pixel li, K chunk lg) = 8 consecutive channels of one pixel" is one
//     16-byte NHWC load (lazy BatchNorm + activation of the producer applied in registers), prefetched TD tiles ahead in a register ring;
//   * the whole weight matrix [Cout][K] sits in LDS (A operand), every wave streams its own 32-pixel tiles: NO workgroup barrier in the loop;
//   * a wave owns ALL output channels of its pixels: the bf16 tile is staged in a wave-private LDS area (a wave's LDS operations execute in
//     order: no barrier), the statistics of the stored values come off the matrix cores from that tile (ones . F and diag(F^T F), as the
//     other forward kernels), and it leaves as 16-byte stores of one contiguous 32 x Cout x 2-byte run.
// Same K order and rounding points as conv_gemm_kernel: bit-identical outputs; the statistics differ in summation order only.
#include <type_traits>
#include "common.h"
#include "../../include/adamml_hip.h"

namespace {

struct N1P {
    const bf16_t* x;         // [groups * P][Cin]
    const bf16_t* w;         // [Cout][Cin] bf16 forward pack
    const float* in_scale;   // lazy input transform (null: identity), group stride in_gs
    const float* in_shift;
    bf16_t* y;               // [groups * P][Cout]
    double* stats;           // [groups][SLOTS][2 * Cout] or null
    int act, in_gs, Cin;
    long P;                  // pixels per group
};

typedef __attribute__((ext_vector_type(4))) short s16x4_;

// ALLFULL: P % 32 == 0 (the launcher's choice): the tile's run leaves through unconditional stores (behind a store under a per-lane condition
// every wait of the loop is a conservative one: csrc/tpool_bwd_prod.hip went 1.74 -> 1.60 ms on that alone)
template <int KS, int COUT, bool ALLFULL>
__global__ __launch_bounds__(256, 2) void conv1x1_narrow_fwd_kernel(N1P p) {
    constexpr int KP = KS * 32;                      // padded K
    constexpr int NCT = (COUT + 15) / 16;            // 16-wide cout tiles
    constexpr int CW = NCT * 16;
    constexpr int WROW = KP * 2 + 16;                // LDS bytes per weight row (+16: bank skew for the 16-lane row reads)
    constexpr int SROW = CW * 2 + 8;                 // staging row bytes
    constexpr int CPR = COUT / 8;                    // 16-byte chunks per output pixel
    constexpr int TD = KS == 1 ? 4 : 2;              // tiles of loads in flight per wave
    constexpr int TPX = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;                                                        // [CW][WROW]
    float* s_vec = reinterpret_cast<float*>(smem + CW * WROW);               // [2][KP]: scale, shift of the lazy input
    float* s_sum = s_vec + 2 * KP;                                           // [4 waves][2 * CW]
    char* s_stage = reinterpret_cast<char*>(s_sum + 4 * 2 * CW);             // [4 waves][32][SROW]
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    p.x += (size_t)g * p.P * p.Cin;
    p.y += (size_t)g * p.P * COUT;
    if (p.in_scale) { p.in_scale += (size_t)g * p.in_gs; p.in_shift += (size_t)g * p.in_gs; }
    for (int i = tid; i < CW * (KP / 8); i += 256) {
        const int row = i / (KP / 8), ch = i - row * (KP / 8);
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < COUT && ch * 8 < p.Cin) v = *reinterpret_cast<const bf16x8*>(p.w + (size_t)row * p.Cin + ch * 8);
        *reinterpret_cast<bf16x8*>(s_w + row * WROW + ch * 16) = v;
    }
    for (int i = tid; i < KP; i += 256) {                // (channels >= Cin: raw 0 -> act(1 * 0 + 0) = 0)
        s_vec[i] = (p.in_scale && i < p.Cin) ? p.in_scale[i] : 1.f;
        s_vec[KP + i] = (p.in_scale && i < p.Cin) ? p.in_shift[i] : 0.f;
    }
    for (int i = tid; i < 4 * 2 * CW; i += 256) s_sum[i] = 0.f;
    __syncthreads();

    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float alo = uniform(p.in_scale ? act_lo(p.act) : -INFINITY), ahi = uniform(p.in_scale ? act_hi(p.act) : INFINITY);
    const bool lazy = p.in_scale != nullptr;
    float* csw = s_sum + wave * 2 * CW;
    char* stg = s_stage + wave * (TPX * SROW);
    const long ntile = (p.P + TPX - 1) / TPX;
    const long wid = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
    // this lane's K chunks: chunk k * 4 + lg of a pixel row exists while (k * 4 + lg) * 8 < Cin (else the lane re-reads chunk 0 and the
    // value is discarded: the weights of those K positions are zero and so is the transformed value)
    bf16x8 ring[TD][KS][2];
    auto issue = [&](int u, long tile) {
        const long p0 = tile * TPX;
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) {
            long px = p0 + pg * 16 + li;
            if (!ALLFULL) px = px < p.P ? px : p.P - 1;
            const bf16_t* row = p.x + px * p.Cin;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const int c = (k * 4 + lg) * 8;
                ring[u][k][pg] = *reinterpret_cast<const bf16x8*>(row + (c < p.Cin ? c : 0));
            }
        }
    };
#pragma unroll
    for (int u = 0; u < TD; ++u) {
        const long t = wid + (long)u * nw;
        issue(u, t < ntile ? t : ntile - 1);
    }
    union { s16x4_ h[2]; bf16x8 v; } ones;
    ones.h[0] = s16x4_{0x3F80, 0x3F80, 0x3F80, 0x3F80};
    ones.h[1] = ones.h[0];
    const int trow = 8 * lg + (li >> 2);

    for (long tb = wid; tb < ntile; tb += (long)TD * nw) {
#pragma unroll
        for (int u = 0; u < TD; ++u) {
            const long t = tb + (long)u * nw;
            if (t >= ntile) break;                                   // (wave-uniform)
            f32x4 acc[2][NCT];
#pragma unroll
            for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[pg][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                bf16x8 fb[2];
#pragma unroll
                for (int pg = 0; pg < 2; ++pg) {
                    fb[pg] = ring[u][k][pg];
                    if ((k * 4 + lg) * 8 >= p.Cin) fb[pg] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                }
                if (lazy) {                                           // (uniform) act(scale * raw + shift), rounded as the other loaders round it
                    const f32x8 sc = load_f32x8(s_vec + k * 32 + lg * 8), sh = load_f32x8(s_vec + KP + k * 32 + lg * 8);
#pragma unroll
                    for (int pg = 0; pg < 2; ++pg) {
                        f32x8 v = bf8_to_f32(fb[pg]);
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = clamp_act(fmaf(v[i], sc[i], sh[i]), alo, ahi);
                        fb[pg] = f32_to_bf8(v);
                    }
                }
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(s_w + (ct * 16 + li) * WROW + k * 64 + lg * 16);
#pragma unroll
                    for (int pg = 0; pg < 2; ++pg) acc[pg][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[pg], acc[pg][ct], 0, 0, 0);
                }
            }
            // this slot's next tile (TD tiles of this wave ahead; past the end: the last tile again -- unconditional requests)
            {
                const long tn = t + (long)TD * nw;
                issue(u, tn < ntile ? tn : ntile - 1);
            }
            // ---- epilogue: D fragment lane (li, lg) = channels ct * 16 + lg * 4 .. + 3 of pixel pg * 16 + li -> wave-private staging tile
            const int npx = ALLFULL ? TPX : (int)(p.P - t * TPX < TPX ? p.P - t * TPX : TPX);
#pragma unroll
            for (int pg = 0; pg < 2; ++pg) {
                const bool live = ALLFULL || pg * 16 + li < npx;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    bf16x4 v = f32_to_bf4(acc[pg][ct]);
                    if (!live) v = bf16x4{0, 0, 0, 0};
                    *reinterpret_cast<bf16x4*>(stg + (pg * 16 + li) * SROW + (ct * 16 + lg * 4) * 2) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the wave's own LDS writes have landed; no other wave touches this area)
            if (p.stats) {
#pragma unroll
                for (int cb = 0; cb < NCT; ++cb) {
                    const char* fp = stg + trow * SROW + (cb * 16 + 4 * (li & 3)) * 2;
                    union { s16x4_ h[2]; bf16x8 v; } f;
                    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(fp));
                    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(fp + 4 * SROW));
                    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 dsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, f.v, z4, 0, 0, 0);
                    const f32x4 dsq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.v, f.v, z4, 0, 0, 0);
                    if (lg == 0) csw[cb * 16 + li] += dsum[0];                        // (own row, one owner lane per entry, tile order)
                    if ((li >> 2) == lg) {
                        const int r = li & 3;
                        csw[CW + cb * 16 + li] += r == 0 ? dsq[0] : r == 1 ? dsq[1] : r == 2 ? dsq[2] : dsq[3];
                    }
                }
            }
            // ---- the tile leaves as ONE contiguous run of npx * COUT * 2 bytes, 16 bytes per lane
            bf16_t* yb = p.y + (size_t)t * TPX * COUT;
#pragma unroll
            for (int i = 0; i < (TPX * CPR + 63) / 64; ++i) {
                int e = lane + 64 * i;
                // (ALLFULL: a run of 32 CPR chunks that does not fill its last 64-lane pass -- 24 channels: 96 chunks -- lets the idle lanes
                // repeat the chunks 32 below theirs: the same bytes to the same address, no predicate)
                if (ALLFULL && (TPX * CPR) % 64 != 0 && e >= TPX * CPR) e -= 32;
                const int px = e / CPR, ch = e - px * CPR;
                if (ALLFULL || e < npx * CPR) {                      // (two 8-byte LDS reads: the staging rows are 8-byte aligned only)
                    union { struct { s16x4_ a, b; } s; bf16x8 v; } o;
                    o.s.a = *reinterpret_cast<const s16x4_*>(stg + px * SROW + ch * 16);
                    o.s.b = *reinterpret_cast<const s16x4_*>(stg + px * SROW + ch * 16 + 8);
                    *reinterpret_cast<bf16x8*>(yb + (size_t)e * 8) = o.v;
                }
            }
            asm volatile("" ::: "memory");
        }
    }
    if (p.stats) {
        __syncthreads();
        for (int i = tid; i < 2 * COUT; i += 256) {          // wave rows folded in wave order, one exact add per channel and workgroup (common.h)
            const int idx = i < COUT ? i : CW + (i - COUT);
            const float v = ((s_sum[idx] + s_sum[2 * CW + idx]) + s_sum[4 * CW + idx]) + s_sum[6 * CW + idx];
            stat_publish(p.stats + (size_t)g * ADAMML_STAT_SLOTS * 2 * COUT + i, 2 * COUT, blockIdx.x & (ADAMML_STAT_SLOTS - 1), v);
        }
    }
}

template <int KS, int COUT, bool ALLFULL>
int narrow_launch_(const N1P& p, int groups, hipStream_t stream) {
    constexpr int KP = KS * 32, NCT = (COUT + 15) / 16, CW = NCT * 16;
    constexpr size_t lds = (size_t)CW * (KP * 2 + 16) + 2 * KP * 4 + 4 * 2 * CW * 4 + 4 * 32 * (CW * 2 + 8);
    static AdamLdsOnce attr_once;                    // (per device: common.h)
    const int attr_dev = adamml_current_device();
    if (!attr_once.test(attr_dev)) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_narrow_fwd_kernel<KS, COUT, ALLFULL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "conv1x1 (narrow): cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        }
        attr_once.set(attr_dev);
    }
    const long ntile = (p.P + 31) / 32;
    long nblk = (ntile + 3) / 4;
    const long per_cu = lds > 0 ? (160 * 1024) / (long)lds : 1;
    long cap = 256 * (per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu)) / groups;         // persistent workgroups over all groups
    if (cap < 1) cap = 1;
    if (nblk > cap) nblk = cap;
    hipLaunchKernelGGL((conv1x1_narrow_fwd_kernel<KS, COUT, ALLFULL>), dim3((unsigned)nblk, groups), dim3(256), lds, stream, p);
    return adamml_check_launch("conv_fwd (narrow 1x1 stream)");
}

template <int KS, int COUT>
int narrow_launch(const N1P& p, int groups, hipStream_t stream) {
    return p.P % 32 == 0 ? narrow_launch_<KS, COUT, true>(p, groups, stream) : narrow_launch_<KS, COUT, false>(p, groups, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Data gradient of a narrow PROJECTION conv (K = 16 / 24 / 32 gradient channels -> 32 .. 192 channels of the expanded tensor) in the forms
// adamml_conv_bwd_data_dual runs for the MobileNetV2s: DUAL -- the loader reads the masked gradient g and the raw conv output z and forms
// dz = A g + B z + C per channel (the BatchNorm-backward "apply" of the projection's own BatchNorm, never a pass of its own), writing dz
// once on the side for the weight gradient --, and EPI 1 -- the BatchNorm-fused epilogue of the tensor the gradient flows into (the
// depthwise conv's output): g' = dx * act'(scale z_in + shift) stored, sum g' and sum g' zhat accumulated -- or EPI 2 (accumulate into dx) /
// EPI 0 (plain store).  Same streaming structure as the forward kernel above.  The epilogue walks the staged tile with a CONSTANT
// 8-channel chunk per lane (64 / CPR pixels per pass), so the per-channel sums stay in registers over all tiles of a wave; wide outputs are
// produced in blocks of <= 96 channels (6 MFMA tiles) that reuse the B fragments.  conv_gemm_kernel's DUAL instance ran these layers at
// 2.4-2.9 TB/s (round 5, tools/launch_table.py sound).
struct ND1P {
    const bf16_t* g;         // [groups * P][K]
    const bf16_t* z;         // DUAL: [groups * P][K]
    const float* aff;        // DUAL: [groups][3][K]
    bf16_t* side;            // DUAL: dz out or null
    const bf16_t* w;         // [COUT][K] (data-gradient pack)
    bf16_t* dx;              // [groups * P][COUT]
    const bf16_t* z_in;      // EPI 1: [groups * P][COUT]
    const float* bn_vec;     // EPI 1: [groups][4][COUT]
    double* stats;           // EPI 1: [groups][SLOTS][2 * COUT]
    int bn_act, K;
    long P;
};

template <int KS, int COUT, bool DUAL, int EPI>
__global__ __launch_bounds__(256, 2) void conv1x1_narrow_dgrad_kernel(ND1P p) {
    constexpr int KP = KS * 32;
    constexpr int NCT = (COUT + 15) / 16;
    constexpr int NCB = (NCT + 5) / 6;                                   // cout blocks of <= 6 tiles
    constexpr int BT = (NCT + NCB - 1) / NCB;                            // tiles per block (the last block may hold fewer live channels)
    constexpr int BW = BT * 16;                                          // block width (channels)
    constexpr int WROW = KP * 2 + 16;
    constexpr int SROW = BW * 2 + 8;
    constexpr int TPX = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;                                                        // [NCB * BW][WROW]
    float* s_aff = reinterpret_cast<float*>(smem + NCB * BW * WROW);         // [3][KP]
    float* s_bn = s_aff + 3 * KP;                                            // [4][NCB * BW]
    float* s_fold = s_bn + 4 * NCB * BW;                                     // [4 waves][64 lanes][NCB][16]  (end of the kernel only)
    char* s_stage = reinterpret_cast<char*>(s_fold + (EPI == 1 ? 4 * 64 * NCB * 16 : 0));     // [4 waves][32][SROW]
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    p.g += (size_t)g * p.P * p.K;
    if (DUAL) { p.z += (size_t)g * p.P * p.K; p.aff += (size_t)g * 3 * p.K; if (p.side) p.side += (size_t)g * p.P * p.K; }
    p.dx += (size_t)g * p.P * COUT;
    if (EPI == 1) { p.z_in += (size_t)g * p.P * COUT; p.bn_vec += (size_t)g * 4 * COUT; }
    for (int i = tid; i < NCB * BW * (KP / 8); i += 256) {
        const int row = i / (KP / 8), ch = i - row * (KP / 8);
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < COUT && ch * 8 < p.K) v = *reinterpret_cast<const bf16x8*>(p.w + (size_t)row * p.K + ch * 8);
        *reinterpret_cast<bf16x8*>(s_w + row * WROW + ch * 16) = v;
    }
    if (DUAL)
        for (int i = tid; i < 3 * KP; i += 256) {
            const int v = i / KP, c = i - v * KP;
            s_aff[i] = c < p.K ? p.aff[v * p.K + c] : 0.f;               // (K positions beyond the tensor: A = B = C = 0 -> dz = 0)
        }
    if (EPI == 1)
        for (int i = tid; i < 4 * NCB * BW; i += 256) {
            const int v = i / (NCB * BW), c = i - v * (NCB * BW);
            s_bn[i] = c < COUT ? p.bn_vec[v * COUT + c] : 0.f;
        }
    __syncthreads();
    char* stg = s_stage + wave * (TPX * SROW);
    const long ntile = (p.P + TPX - 1) / TPX;
    const long wid = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
    // ---- epilogue geometry of a block: lane -> (pixel sub-index pl, 8-channel chunk ch), PPI pixels per pass
    constexpr int CPRB = BW / 8, PPI = 64 / CPRB, NI = (TPX + PPI - 1) / PPI;
    const int epl = lane / CPRB, ech = lane - epl * CPRB;
    const bool eact = lane < PPI * CPRB;
    const float blo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, act_lo(p.bn_act))));
    const float bhi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, act_hi(p.bn_act))));
    f32x8 esum[NCB], esq[NCB];
#pragma unroll
    for (int b = 0; b < NCB; ++b)
#pragma unroll
        for (int i = 0; i < 8; ++i) { esum[b][i] = 0.f; esq[b][i] = 0.f; }

    bf16x8 rg[KS][2], rzz[DUAL ? KS : 1][2], raux[EPI ? NCB : 1][EPI ? NI : 1];
    auto issue = [&](long tile) {
        const long p0 = tile * TPX;
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) {
            long px = p0 + pg * 16 + li;
            px = px < p.P ? px : p.P - 1;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const int c = (k * 4 + lg) * 8;
                const size_t off = (size_t)px * p.K + (c < p.K ? c : 0);
                rg[k][pg] = *reinterpret_cast<const bf16x8*>(p.g + off);
                if (DUAL) rzz[k][pg] = *reinterpret_cast<const bf16x8*>(p.z + off);
            }
        }
    };
    auto issue_aux = [&](long tile) {                                      // the epilogue's second operand (z_in, or the dx accumulated into), requested
        const long p0 = tile * TPX;                                        // when the previous tile's epilogue has consumed its registers
        if (EPI) {
            const bf16_t* aux = EPI == 1 ? p.z_in : p.dx;
#pragma unroll
            for (int b = 0; b < NCB; ++b)
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    long px = p0 + i * PPI + epl;
                    px = px < p.P ? px : p.P - 1;
                    const int c = b * BW + ech * 8;
                    raux[b][i] = *reinterpret_cast<const bf16x8*>(aux + (size_t)px * COUT + (c < COUT ? c : 0));
                }
        }
    };
    issue(wid < ntile ? wid : ntile - 1);
    issue_aux(wid < ntile ? wid : ntile - 1);

    for (long t = wid; t < ntile; t += nw) {
        const int npx = (int)(p.P - t * TPX < TPX ? p.P - t * TPX : TPX);
        // ---- B fragments (dz formed here in the DUAL form, written once on the side)
        bf16x8 fb[KS][2];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int c = (k * 4 + lg) * 8;
#pragma unroll
            for (int pg = 0; pg < 2; ++pg) {
                bf16x8 v = rg[k][pg];
                if (DUAL) {
                    const f32x8 gv = bf8_to_f32(v), zv = bf8_to_f32(rzz[k][pg]);
                    const f32x8 ca = load_f32x8(s_aff + c), cb = load_f32x8(s_aff + KP + c), cc = load_f32x8(s_aff + 2 * KP + c);
                    f32x8 o;
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = fmaf(ca[i], gv[i], fmaf(cb[i], zv[i], cc[i]));
                    v = f32_to_bf8(o);
                    if (p.side && c < p.K && pg * 16 + li < npx) *reinterpret_cast<bf16x8*>(p.side + ((size_t)t * TPX + pg * 16 + li) * p.K + c) = v;
                }
                if (c >= p.K) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                fb[k][pg] = v;
            }
        }
        {
            const long tn = t + nw;
            issue(tn < ntile ? tn : ntile - 1);                           // (unconditional request of this wave's next tile)
        }
#pragma unroll
        for (int b = 0; b < NCB; ++b) {
            f32x4 acc[2][BT];
#pragma unroll
            for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                for (int ct = 0; ct < BT; ++ct) acc[pg][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KS; ++k)
#pragma unroll
                for (int ct = 0; ct < BT; ++ct) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(s_w + ((b * BT + ct) * 16 + li) * WROW + k * 64 + lg * 16);
#pragma unroll
                    for (int pg = 0; pg < 2; ++pg) acc[pg][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[k][pg], acc[pg][ct], 0, 0, 0);
                }
#pragma unroll
            for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                for (int ct = 0; ct < BT; ++ct)
                    *reinterpret_cast<bf16x4*>(stg + (pg * 16 + li) * SROW + (ct * 16 + lg * 4) * 2) = f32_to_bf4(acc[pg][ct]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int c0 = b * BW + ech * 8;                               // this lane's channels of the block
            const bool cact = eact && c0 < COUT;
            f32x8 sc, sh, mu, is;
            if (EPI == 1) {
                sc = load_f32x8(s_bn + c0); sh = load_f32x8(s_bn + NCB * BW + c0);
                mu = load_f32x8(s_bn + 2 * NCB * BW + c0); is = load_f32x8(s_bn + 3 * NCB * BW + c0);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int px = i * PPI + epl;
                if (cact && px < npx) {
                    union { struct { s16x4_ a, b; } s; bf16x8 v; } u;
                    u.s.a = *reinterpret_cast<const s16x4_*>(stg + px * SROW + ech * 16);
                    u.s.b = *reinterpret_cast<const s16x4_*>(stg + px * SROW + ech * 16 + 8);
                    f32x8 f = bf8_to_f32(u.v);
                    bf16x8 v = u.v;
                    if (EPI == 1) {
                        const f32x8 zv = bf8_to_f32(raux[b][i]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) f[j] *= mask_act(fmaf(zv[j], sc[j], sh[j]), blo, bhi);
                        v = f32_to_bf8(f);
                        f = bf8_to_f32(v);
                        esum[b] += f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) esq[b][j] += f[j] * (zv[j] - mu[j]) * is[j];
                    } else if (EPI == 2) {
                        f += bf8_to_f32(raux[b][i]);
                        v = f32_to_bf8(f);
                    }
                    *reinterpret_cast<bf16x8*>(p.dx + ((size_t)t * TPX + px) * COUT + c0) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (staging reads done before the next block / tile overwrites the area)
        }
        {
            const long tn = t + nw;
            issue_aux(tn < ntile ? tn : ntile - 1);
        }
    }
    if (EPI == 1 && p.stats) {
        // per-lane partial sums of (block, 8 channels): to LDS, then one thread per channel folds the PPI lanes of a wave and the four waves
        // in a fixed order (reproducible), one exact add per channel and workgroup
        float* fo = s_fold + (size_t)(wave * 64 + lane) * NCB * 16;
#pragma unroll
        for (int b = 0; b < NCB; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) { fo[b * 16 + j] = esum[b][j]; fo[b * 16 + 8 + j] = esq[b][j]; }
        __syncthreads();
        for (int i = tid; i < 2 * COUT; i += 256) {
            const int which = i >= COUT, c = which ? i - COUT : i;
            const int b = c / BW, cc = c - b * BW, ch = cc >> 3, j = cc & 7;
            float v = 0.f;
            for (int w = 0; w < 4; ++w)
                for (int pl = 0; pl < PPI; ++pl) v += s_fold[(size_t)(w * 64 + pl * CPRB + ch) * NCB * 16 + b * 16 + which * 8 + j];
            stat_publish(p.stats + (size_t)g * ADAMML_STAT_SLOTS * 2 * COUT + i, 2 * COUT, blockIdx.x & (ADAMML_STAT_SLOTS - 1), v);
        }
    }
}

template <int KS, int COUT, bool DUAL, int EPI>
int narrow_dgrad_launch(const ND1P& p, int groups, hipStream_t stream) {
    constexpr int KP = KS * 32, NCT = (COUT + 15) / 16, NCB = (NCT + 5) / 6, BT = (NCT + NCB - 1) / NCB, BW = BT * 16;
    constexpr size_t lds = (size_t)NCB * BW * (KP * 2 + 16) + 3 * KP * 4 + 4 * NCB * BW * 4 + (EPI == 1 ? 4 * 64 * NCB * 16 * 4 : 0) + 4 * 32 * (BW * 2 + 8);
    static AdamLdsOnce attr_once;                    // (per device: common.h)
    const int attr_dev = adamml_current_device();
    if (!attr_once.test(attr_dev)) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_narrow_dgrad_kernel<KS, COUT, DUAL, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "conv_bwd_data (narrow): cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        }
        attr_once.set(attr_dev);
    }
    const long ntile = (p.P + 31) / 32;
    long nblk = (ntile + 3) / 4;
    long cap = 768 / groups;
    if (cap < 1) cap = 1;
    if (nblk > cap) nblk = cap;
    hipLaunchKernelGGL((conv1x1_narrow_dgrad_kernel<KS, COUT, DUAL, EPI>), dim3((unsigned)nblk, groups), dim3(256), lds, stream, p);
    return adamml_check_launch("conv_bwd_data_dual (narrow 1x1 stream)");
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same narrow 1x1 convs: dW[co][ci] = sum_p dz[p][co] * a[p][ci], a = act(scale x + shift) of the lazily normalised
// input.  Both operands are pixel-major and NARROW (16 .. 192 channels): a 32-pixel tile of either is ONE contiguous run of memory.  A wave
// keeps the whole dW (<= 24 MFMA tiles of 16 x 16) in registers over all its tiles; per tile it loads the two runs with 16-byte coalesced
// loads (the next tile's in flight), writes them to a wave-private LDS area (transform applied on the way) and reads both MFMA operands
// back as hardware transpose reads (pixels = the reduction dimension), no workgroup barrier in the loop.  The four waves of a workgroup fold
// their accumulators in wave order through LDS: one partial [Cout][cin_true] per workgroup for adamml_launch_split_reduce.
// conv_wgrad_kernel ran these layers at 1.7-3.9 TB/s (64 x 64 / 128 x 128 output tiles mostly empty, 32-pixel K steps behind barriers).
struct NW1P {
    const bf16_t* dz;        // [groups * P][COUT]
    const bf16_t* x;         // [groups * P][CIN]
    const float* in_scale;
    const float* in_shift;
    float* ws;               // [groups][nblk][COUT][cin_true]
    int act, in_gs, cin_true;
    long P;
};

template <int COUT, int CIN>
__global__ __launch_bounds__(256, 2) void conv1x1_narrow_wgrad_kernel(NW1P p) {
    constexpr int MT = (COUT + 15) / 16, NT = (CIN + 15) / 16;
    constexpr int ZROW = MT * 32 + 8, XROW = NT * 32 + 8;          // staging row bytes (tiles padded to whole 16-channel blocks)
    constexpr int ZC = COUT / 8, XC = CIN / 8;                     // 16-byte chunks per pixel
    constexpr int NZ = (32 * ZC + 63) / 64, NX = (32 * XC + 63) / 64;   // loads per lane and tile
    constexpr int TPX = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_vec = reinterpret_cast<float*>(smem);                          // [2][NT * 16]
    char* s_stage = smem + 2 * NT * 16 * 4;                                  // [4 waves][32][ZROW + XROW]; later the fold buffer [MT*16][NT*16] fp32
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    p.dz += (size_t)g * p.P * COUT;
    p.x += (size_t)g * p.P * CIN;
    if (p.in_scale) { p.in_scale += (size_t)g * p.in_gs; p.in_shift += (size_t)g * p.in_gs; }
    for (int i = tid; i < NT * 16; i += 256) {
        s_vec[i] = (p.in_scale && i < CIN) ? p.in_scale[i] : 1.f;
        s_vec[NT * 16 + i] = (p.in_scale && i < CIN) ? p.in_shift[i] : 0.f;
    }
    char* zs = s_stage + wave * (TPX * (ZROW + XROW));
    char* xs = zs + TPX * ZROW;
    // the padding columns of the staging tiles (channels COUT .. MT*16, CIN .. NT*16) are zero for the whole kernel
    for (int i = lane; i < TPX * (ZROW + XROW) / 8; i += 64) reinterpret_cast<unsigned long long*>(zs)[i] = 0ull;
    __syncthreads();
    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float alo = uniform(p.in_scale ? act_lo(p.act) : -INFINITY), ahi = uniform(p.in_scale ? act_hi(p.act) : INFINITY);
    const bool lazy = p.in_scale != nullptr;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long ntile = (p.P + TPX - 1) / TPX;
    const long wid = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
    bf16x8 rz[NZ], rx[NX];
    auto issue = [&](long tile) {
        const long e0 = tile * TPX;
        const long zmax = p.P * ZC - 1, xmax = p.P * XC - 1;                 // (clamped chunk indices: rows past the end are zeroed at staging)
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            long e = e0 * ZC + lane + 64 * i;
            rz[i] = *reinterpret_cast<const bf16x8*>(p.dz + (e < zmax ? e : zmax) * 8);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            long e = e0 * XC + lane + 64 * i;
            rx[i] = *reinterpret_cast<const bf16x8*>(p.x + (e < xmax ? e : xmax) * 8);
        }
    };
    if (wid < ntile) issue(wid); else issue(ntile - 1);
    const int trow = 8 * lg + (li >> 2);
    for (long t = wid; t < ntile; t += nw) {
        const int npx = (int)(p.P - t * TPX < TPX ? p.P - t * TPX : TPX);
        // ---- staged tiles: chunk e of the run = (pixel e / C, chunk e % C)
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int e = lane + 64 * i, px = e / ZC, ch = e - px * ZC;
            if (e < TPX * ZC) {
                union { struct { s16x4_ a, b; } s; bf16x8 v; } u;
                u.v = px < npx ? rz[i] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<s16x4_*>(zs + px * ZROW + ch * 16) = u.s.a;
                *reinterpret_cast<s16x4_*>(zs + px * ZROW + ch * 16 + 8) = u.s.b;
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = lane + 64 * i, px = e / XC, ch = e - px * XC;
            if (e < TPX * XC) {
                bf16x8 v = rx[i];
                if (lazy) {
                    const f32x8 sc = load_f32x8(s_vec + ch * 8), sh = load_f32x8(s_vec + NT * 16 + ch * 8);
                    f32x8 f = bf8_to_f32(v);
#pragma unroll
                    for (int k = 0; k < 8; ++k) f[k] = clamp_act(fmaf(f[k], sc[k], sh[k]), alo, ahi);
                    v = f32_to_bf8(f);
                }
                union { struct { s16x4_ a, b; } s; bf16x8 v; } u;
                u.v = px < npx ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<s16x4_*>(xs + px * XROW + ch * 16) = u.s.a;
                *reinterpret_cast<s16x4_*>(xs + px * XROW + ch * 16 + 8) = u.s.b;
            }
        }
        {
            const long tn = t + nw;
            issue(tn < ntile ? tn : ntile - 1);                           // (unconditional request of this wave's next tile)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (own LDS writes landed; no other wave touches this area)
        auto frag = [&](const char* base, int row_bytes, int blk) {
            const char* q = base + trow * row_bytes + (blk * 16 + 4 * (li & 3)) * 2;
            union { s16x4_ h[2]; bf16x8 v; } f;
            f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(q));
            f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(q + 4 * row_bytes));
            return f.v;
        };
        if constexpr (NT <= MT) {                                        // (the fragments of the NARROWER operand are held, the other side streams)
            bf16x8 fb[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[nt] = frag(xs, XROW, nt);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bf16x8 fa = frag(zs, ZROW, mt);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[nt], acc[mt][nt], 0, 0, 0);
            }
        } else {
            bf16x8 fa[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[mt] = frag(zs, ZROW, mt);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bf16x8 fb = frag(xs, XROW, nt);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb, acc[mt][nt], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (the transpose reads are done before the next tile overwrites the area)
    }
    // ---- fold the four waves' accumulators in wave order (fixed order: reproducible), one partial per workgroup
    __syncthreads();
    float* fold = reinterpret_cast<float*>(s_stage);                     // [MT*16][NT*16]
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* q = fold + (mt * 16 + lg * 4 + r) * (NT * 16) + nt * 16 + li;
                        *q = w == 0 ? acc[mt][nt][r] : *q + acc[mt][nt][r];
                    }
        }
        __syncthreads();
    }
    float* out = p.ws + ((size_t)g * gridDim.x + blockIdx.x) * ((size_t)COUT * p.cin_true);
    for (int i = tid; i < COUT * p.cin_true; i += 256) {
        const int co = i / p.cin_true, ci = i - co * p.cin_true;
        out[i] = fold[co * (NT * 16) + ci];
    }
}

template <int COUT, int CIN>
int narrow_wgrad_launch(const NW1P& p, int groups, int nblk, hipStream_t stream) {
    constexpr int MT = (COUT + 15) / 16, NT = (CIN + 15) / 16;
    constexpr size_t stage = (size_t)4 * 32 * ((MT * 32 + 8) + (NT * 32 + 8)), foldb = (size_t)MT * 16 * NT * 16 * 4;
    constexpr size_t lds = 2 * NT * 16 * 4 + (stage > foldb ? stage : foldb);
    static AdamLdsOnce attr_once;                    // (per device: common.h)
    const int attr_dev = adamml_current_device();
    if (!attr_once.test(attr_dev)) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_narrow_wgrad_kernel<COUT, CIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "conv_bwd_weight (narrow): cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        }
        attr_once.set(attr_dev);
    }
    hipLaunchKernelGGL((conv1x1_narrow_wgrad_kernel<COUT, CIN>), dim3((unsigned)nblk, groups), dim3(256), lds, stream, p);
    return adamml_check_launch("conv_bwd_weight (narrow 1x1 stream)");
}

}  // namespace

// Shapes with an instance (K steps of 32 channels x true output channels): the byte-heavy MobileNetV2 layers
static int narrow_ks(int Cin) { return (Cin + 31) / 32; }

bool adamml_conv1x1_narrow_fwd_supported(const adamml_conv_desc_t* d) {
    static const int on = getenv("ADAMML_NARROW_STREAM") ? atoi(getenv("ADAMML_NARROW_STREAM")) : 1;      // A/B aid: 0 = conv_gemm_kernel
    if (!on || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->up > 1 || d->accumulate) return false;
    if (d->Cin % 8 || d->Cout % 8) return false;
    const int ks = narrow_ks(d->Cin), co = d->Cout;
    return (ks == 1 && (co == 16 || co == 96 || co == 144 || co == 192)) || (ks == 3 && co == 24) || (ks == 5 && (co == 24 || co == 32)) ||
           (ks == 6 && co == 32);
}

int adamml_conv1x1_narrow_fwd_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                                     void* y, double* stats, hipStream_t stream) {
    N1P p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w_packed; p.in_scale = in_scale; p.in_shift = in_scale ? in_shift : nullptr;
    p.y = (bf16_t*)y; p.stats = stats; p.act = d->act; p.in_gs = d->in_gstride; p.Cin = d->Cin;
    p.P = (long)d->N * d->H * d->W;
    if (p.P <= 0) return ADAMML_OK;
    const int groups = d->groups < 1 ? 1 : d->groups;
    const int ks = narrow_ks(d->Cin), co = d->Cout;
    if (ks == 1 && co == 16) return narrow_launch<1, 16>(p, groups, stream);
    if (ks == 1 && co == 96) return narrow_launch<1, 96>(p, groups, stream);
    if (ks == 1 && co == 144) return narrow_launch<1, 144>(p, groups, stream);
    if (ks == 1 && co == 192) return narrow_launch<1, 192>(p, groups, stream);
    if (ks == 3 && co == 24) return narrow_launch<3, 24>(p, groups, stream);
    if (ks == 5 && co == 24) return narrow_launch<5, 24>(p, groups, stream);
    if (ks == 5 && co == 32) return narrow_launch<5, 32>(p, groups, stream);
    if (ks == 6 && co == 32) return narrow_launch<6, 32>(p, groups, stream);
    return adamml_set_error(ADAMML_EUNSUPPORTED, "conv1x1 (narrow): no instance for Cin %d -> Cout %d", d->Cin, d->Cout);
}

// ---- weight gradient: (Cout, Cin) pairs with an instance = the same layers
bool adamml_conv1x1_narrow_wgrad_supported(const adamml_conv_desc_t* d, int cin_true) {
    static const int on = getenv("ADAMML_NARROW_STREAM") ? atoi(getenv("ADAMML_NARROW_STREAM")) : 1;
    if (!on || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->up > 1 || cin_true != d->Cin) return false;
    const int co = d->Cout, ci = d->Cin;
    return (co == 96 && ci == 16) || (co == 16 && ci == 32) || (co == 24 && ci == 96) || (co == 144 && ci == 24) || (co == 24 && ci == 144) ||
           (co == 32 && ci == 144) || (co == 192 && ci == 32) || (co == 32 && ci == 192);
}

// one partial [Cout][cin_true] per workgroup into ws [groups][nblk]...; returns the launch status, *nblk_out = partials per group
int adamml_conv1x1_narrow_wgrad_launch(const adamml_conv_desc_t* d, const void* dz, const void* x, const float* in_scale, const float* in_shift,
                                       float* ws, int max_blocks_per_group, int* nblk_out, hipStream_t stream) {
    NW1P p;
    p.dz = (const bf16_t*)dz; p.x = (const bf16_t*)x; p.in_scale = in_scale; p.in_shift = in_scale ? in_shift : nullptr; p.ws = ws;
    p.act = d->act; p.in_gs = d->in_gstride; p.cin_true = d->Cin;
    p.P = (long)d->N * d->H * d->W;
    const int groups = d->groups < 1 ? 1 : d->groups;
    const long ntile = (p.P + 31) / 32;
    long nblk = (ntile + 3) / 4;
    if (nblk > max_blocks_per_group) nblk = max_blocks_per_group;
    if (nblk < 1) nblk = 1;
    *nblk_out = (int)nblk;
    const int co = d->Cout, ci = d->Cin;
    if (co == 96 && ci == 16) return narrow_wgrad_launch<96, 16>(p, groups, (int)nblk, stream);
    if (co == 16 && ci == 32) return narrow_wgrad_launch<16, 32>(p, groups, (int)nblk, stream);
    if (co == 24 && ci == 96) return narrow_wgrad_launch<24, 96>(p, groups, (int)nblk, stream);
    if (co == 144 && ci == 24) return narrow_wgrad_launch<144, 24>(p, groups, (int)nblk, stream);
    if (co == 24 && ci == 144) return narrow_wgrad_launch<24, 144>(p, groups, (int)nblk, stream);
    if (co == 32 && ci == 144) return narrow_wgrad_launch<32, 144>(p, groups, (int)nblk, stream);
    if (co == 192 && ci == 32) return narrow_wgrad_launch<192, 32>(p, groups, (int)nblk, stream);
    if (co == 32 && ci == 192) return narrow_wgrad_launch<32, 192>(p, groups, (int)nblk, stream);
    return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_weight (narrow): no instance for Cout %d, Cin %d", co, ci);
}

// ---- DUAL data gradient (the BatchNorm-backward apply folded into the loader, dz written once on the side):
//   * projection convs: K = d->Cout gradient channels (<= 32) -> d->Cin in {32, 96, 144, 192};
//   * round 6, EXPANSION convs (16 -> 96, 24 -> 144, 32 -> 192; models/sound_mobilenet_v2.py:43-69, models/policy_net.py:63-95): K = d->Cout
//     = 96 / 144 / 192 gradient channels -> d->Cin = 16 / 24 / 32.  Their gradient arrives already masked by the ReLU6 of their own BatchNorm
//     (the depthwise conv's fused backward did that and accumulated the sums), so dz = A g' + B z + C as for a linear BatchNorm: the separate
//     adamml_bn_bwd_apply pass over the 6x-wide tensor (read g', read z, write dz) and the data gradient's re-read of dz become ONE read of
//     (g', z) and one write of dz for the weight gradient -- 4 passes over the wide tensor instead of 5.
static bool narrow_dual_expansion(const adamml_conv_desc_t* d) {
    return (d->Cin == 16 && d->Cout == 96) || (d->Cin == 24 && d->Cout == 144) || (d->Cin == 32 && d->Cout == 192);
}

bool adamml_conv1x1_narrow_dual_supported(const adamml_conv_desc_t* d) {
    static const int on = getenv("ADAMML_NARROW_STREAM") ? atoi(getenv("ADAMML_NARROW_STREAM")) : 1;
    if (!on || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->up > 1) return false;
    // measured (round 6, three alternating pairs on one box): Sound-MobileNetV2 15.5 ms with, 15.6 ms without; step 108.05 / 108.06 / 107.46 ms with,
    // 107.66 / 107.76 / 107.80 ms without -- the pass it saves is paid back by the second operand stream of the K = 96 .. 192 loader.  Off by
    // default; ADAMML_NARROW_DUAL_EXP=1 enables it (the instances are tested either way: tests/test_kernels_gpu.py)
    const char* exp_env = getenv("ADAMML_NARROW_DUAL_EXP");          // (read at every call: a test enables the instances within one process)
    const int exp_on = exp_env ? atoi(exp_env) : 0;
    if (exp_on && narrow_dual_expansion(d)) return true;
    return d->Cout % 8 == 0 && d->Cout <= 32 && (d->Cin == 32 || d->Cin == 96 || d->Cin == 144 || d->Cin == 192);
}

template <int COUT>
static int narrow_dual_dispatch(const ND1P& p, int groups, int epi, hipStream_t stream) {
    if (epi == 1) return narrow_dgrad_launch<1, COUT, true, 1>(p, groups, stream);
    if (epi == 2) return narrow_dgrad_launch<1, COUT, true, 2>(p, groups, stream);
    return narrow_dgrad_launch<1, COUT, true, 0>(p, groups, stream);
}

template <int KS, int COUT>
static int narrow_dual_exp_dispatch(const ND1P& p, int groups, int epi, hipStream_t stream) {
    if (epi == 1) return narrow_dgrad_launch<KS, COUT, true, 1>(p, groups, stream);
    if (epi == 2) return narrow_dgrad_launch<KS, COUT, true, 2>(p, groups, stream);
    return narrow_dgrad_launch<KS, COUT, true, 0>(p, groups, stream);
}

// d: the FORWARD descriptor of the conv (as adamml_conv_bwd_data_dual receives it)
int adamml_conv1x1_narrow_dual_launch(const adamml_conv_desc_t* d, const void* g, const void* z, const float* aff, void* dz_side,
                                      const void* w_dgrad_packed, void* dx, int accumulate, const void* z_in, const float* bn_vec, int act,
                                      double* sums, hipStream_t stream) {
    ND1P p;
    p.g = (const bf16_t*)g; p.z = (const bf16_t*)z; p.aff = aff; p.side = (bf16_t*)dz_side; p.w = (const bf16_t*)w_dgrad_packed; p.dx = (bf16_t*)dx;
    p.z_in = (const bf16_t*)z_in; p.bn_vec = bn_vec; p.stats = sums; p.bn_act = act; p.K = d->Cout;
    p.P = (long)d->N * d->OH * d->OW;
    if (p.P <= 0) return ADAMML_OK;
    const int groups = d->groups < 1 ? 1 : d->groups;
    const int epi = z_in ? 1 : (accumulate ? 2 : 0);
    if (narrow_dual_expansion(d)) {
        if (d->Cin == 16) return narrow_dual_exp_dispatch<3, 16>(p, groups, epi, stream);
        if (d->Cin == 24) return narrow_dual_exp_dispatch<5, 24>(p, groups, epi, stream);
        return narrow_dual_exp_dispatch<6, 32>(p, groups, epi, stream);
    }
    switch (d->Cin) {
        case 32: return narrow_dual_dispatch<32>(p, groups, epi, stream);
        case 96: return narrow_dual_dispatch<96>(p, groups, epi, stream);
        case 144: return narrow_dual_dispatch<144>(p, groups, epi, stream);
        case 192: return narrow_dual_dispatch<192>(p, groups, epi, stream);
    }
    return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_data_dual (narrow): no instance for %d -> %d channels", d->Cout, d->Cin);
}

// ---- plain-loader data gradient of the EXPANSION convs (K = d->Cin = 96 / 144 / 192 gradient channels -> 16 / 24 / 32) with the BatchNorm-fused
// epilogue (adamml_conv_bwd_data_bn) or accumulating into dx.  d: the data-gradient-shaped descriptor conv_launch works with.
bool adamml_conv1x1_narrow_dgrad_epi_supported(const adamml_conv_desc_t* d) {
    static const int on = getenv("ADAMML_NARROW_STREAM") ? atoi(getenv("ADAMML_NARROW_STREAM")) : 1;
    if (!on || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->up > 1) return false;
    const int ks = narrow_ks(d->Cin), co = d->Cout;
    return d->Cin % 8 == 0 && ((ks == 3 && co == 16) || (ks == 5 && (co == 24 || co == 32)) || (ks == 6 && co == 32));
}

int adamml_conv1x1_narrow_dgrad_epi_launch(const adamml_conv_desc_t* d, const void* dz, const void* w_packed, void* dx, const void* z_in,
                                           const float* bn_vec, int act, double* sums, hipStream_t stream) {
    ND1P p;
    p.g = (const bf16_t*)dz; p.z = nullptr; p.aff = nullptr; p.side = nullptr; p.w = (const bf16_t*)w_packed; p.dx = (bf16_t*)dx;
    p.z_in = (const bf16_t*)z_in; p.bn_vec = bn_vec; p.stats = sums; p.bn_act = act; p.K = d->Cin;
    p.P = (long)d->N * d->OH * d->OW;
    if (p.P <= 0) return ADAMML_OK;
    const int groups = d->groups < 1 ? 1 : d->groups;
    const int ks = narrow_ks(d->Cin), co = d->Cout;
    const bool bn = z_in != nullptr;
    if (ks == 3 && co == 16) return bn ? narrow_dgrad_launch<3, 16, false, 1>(p, groups, stream) : narrow_dgrad_launch<3, 16, false, 2>(p, groups, stream);
    if (ks == 5 && co == 24) return bn ? narrow_dgrad_launch<5, 24, false, 1>(p, groups, stream) : narrow_dgrad_launch<5, 24, false, 2>(p, groups, stream);
    if (ks == 5 && co == 32) return bn ? narrow_dgrad_launch<5, 32, false, 1>(p, groups, stream) : narrow_dgrad_launch<5, 32, false, 2>(p, groups, stream);
    if (ks == 6 && co == 32) return bn ? narrow_dgrad_launch<6, 32, false, 1>(p, groups, stream) : narrow_dgrad_launch<6, 32, false, 2>(p, groups, stream);
    return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_data (narrow): no instance for %d -> %d channels", d->Cin, d->Cout);
}
