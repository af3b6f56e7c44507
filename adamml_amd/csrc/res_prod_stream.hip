// Streaming form of adamml_conv_bwd_data_res_prod for the layer-1 bottlenecks (models/resnet.py:103-111 under backward): the data gradient of
// conv1 (64 -> 256 channels seen from the gradient's side, K = 64) accumulated onto the identity-path gradient already in dx, masked with
// the 1-bit ReLU mask of the previous block's output, stored once as g', with sum(g') per channel and the product P = g'^T a [256][64] of
// the algebraic BatchNorm backward of the previous block's conv3 (a = its lazily normalised input).
//
// The tile kernel of conv_gemm.hip serves this launch at 4.1 TB/s (two workgroups of 251 registers per CU, three barriers per tile, the
// product's operand through LDS-DMA).  Here the structure of tpool_bwd_prod.hip: no barrier after the prologue.  Wave q of a four-wave
// workgroup owns the 64-channel slice q of the 256 gradient channels for the workgroup's 32-pixel tiles -- its rows of dx are 128-byte
// runs -- with its 64 x 64 weight slice (8 MFMA A fragments) and its slice P[64 q .. 64 q + 63][:] (16 MFMA tiles) in registers for the
// whole kernel.  Per tile: 16 MFMAs of the conv (dz fragments straight from global memory, one 16-byte load per lane and fragment), the
// raw tile staged as bf16 in the wave's private LDS area (the rounding point of the tile kernel), the epilogue in (pixel, 8-channel
// chunk) lanes -- + identity gradient, mask, store, sum -- which writes g' back to the area, the a rows next to it, and 16 MFMAs of the
// product over hardware transpose reads of both (pixels = the reduction dimension).  All loads of the workgroup's next tile are
// requested, unconditionally and clamped, between the stores and the product.  g' is bit-identical to the tile kernel's.
#include "common.h"
#include "../../include/adamml_hip.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_;

struct RPS {
    const bf16_t* dz;        // [groups][P][64] gradient of conv1's raw output
    const bf16_t* w;         // [256][64] data-gradient pack of conv1 (rows = gradient channels)
    bf16_t* dx;              // [groups][P][256] identity-path gradient in, g' out
    const uint8_t* mask;     // [groups][P][32] 1 bit per element
    double* sums;            // [groups][SLOTS][512]: sum(g') into the first 256 entries
    const bf16_t* a;         // [groups][P][64] raw conv3 input of the previous block
    const float* in_scale;   // its lazy BatchNorm (group stride in_gs) or null
    const float* in_shift;
    float* ws;               // [groups][gridDim.x][256][64] partial products
    const bf16_t* zb;        // SECOND: raw output of the second BatchNorm'd operand of the add (the downsample branch) [groups][P][C]
    const float* vecb;       //         its vectors [groups][4][C] (mean, invstd in rows 2, 3)
    double* sums_b;          //         [groups][SLOTS][2C]: sum(g') and sum(g' zhat_b)
    int in_gs, act, P;
};

constexpr int CIN = 64, TPX = 32;
constexpr int ZROW = 64 * 2 + 8, XROW = CIN * 2 + 8;

// NQ waves = NQ x 64 gradient channels; K = channels of dz; PF: with the product (layer 1: <4, 64, true>; layer 2 without it: <8, 128, false>)
template <int NQ, int K, bool PF, bool SECOND = false>
__global__ __launch_bounds__(NQ * 64, NQ == 4 ? 2 : 1) void res_prod_stream_kernel(RPS p) {
    constexpr int C = NQ * 64, KS = K / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_vec = reinterpret_cast<float*>(smem);                  // [2][CIN]
    char* s_stage = smem + 2 * CIN * 4;                             // [4 waves][32][ZROW + XROW]
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    {
        const size_t pp = (size_t)g * p.P;
        p.dz += pp * K;
        p.dx += pp * C + q * 64;
        p.mask += pp * (C / 8) + q * 8;
        if (PF) p.a += pp * CIN;
        if (SECOND) { p.zb += pp * C + q * 64; p.vecb += (size_t)g * 4 * C; p.sums_b += (size_t)g * ADAMML_STAT_SLOTS * 2 * C; }
        p.sums += (size_t)g * ADAMML_STAT_SLOTS * 2 * C;
    }
    if (PF)
        for (int i = tid; i < CIN; i += NQ * 64) {
            s_vec[i] = p.in_scale ? p.in_scale[(size_t)g * p.in_gs + i] : 1.f;
            s_vec[CIN + i] = p.in_scale ? p.in_shift[(size_t)g * p.in_gs + i] : 0.f;
        }
    constexpr int WAREA = TPX * (ZROW + (PF ? XROW : 0));
    char* zs = s_stage + q * WAREA;
    char* xs = zs + TPX * ZROW;
    for (int i = lane; i < WAREA / 8; i += 64) reinterpret_cast<unsigned long long*>(zs)[i] = 0ull;
    __syncthreads();
    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float alo = uniform(p.in_scale ? act_lo(p.act) : -INFINITY), ahi = uniform(p.in_scale ? act_hi(p.act) : INFINITY);
    const bool lazy = PF && p.in_scale != nullptr;
    // ---- this wave's weight slice: A fragments (row = gradient channel 64 q + 16 ct + li, k = 32 ks + 8 lg ..)
    bf16x8 wr[4][KS];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wr[ct][ks] = *reinterpret_cast<const bf16x8*>(p.w + (size_t)(q * 64 + ct * 16 + li) * K + ks * 32 + lg * 8);
    f32x4 acc[PF ? 4 : 1][4];
#pragma unroll
    for (int a = 0; a < (PF ? 4 : 1); ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float sa[8], sb[SECOND ? 8 : 1];
#pragma unroll
    for (int i = 0; i < 8; ++i) sa[i] = 0.f;
#pragma unroll
    for (int i = 0; i < (SECOND ? 8 : 1); ++i) sb[i] = 0.f;
    const int ntile = (p.P + TPX - 1) / TPX;
    // this lane's four (pixel, 8-channel chunk) slots of a tile: chunk lane % 8 of pixels lane / 8 + 8 i
    const int zch = lane & 7, zpx = lane >> 3;

    bf16x8 dzf[2][KS], old[4], ra[PF ? 4 : 1], rzb[SECOND ? 4 : 1];
    f32x8 mu2, is2;                                                         // SECOND: mean / invstd of this lane's 8 channels
    if constexpr (SECOND) { mu2 = load_f32x8(p.vecb + 2 * C + q * 64 + (lane & 7) * 8); is2 = load_f32x8(p.vecb + 3 * C + q * 64 + (lane & 7) * 8); }
    unsigned mb[4];
    // (uniform 64-bit bases -- the tile is the same for the whole wave -- plus 32-bit lane offsets)
    auto issue = [&](int tile) {
        const int p0 = tile * TPX;
        const int npx = p.P - p0 < TPX ? p.P - p0 : TPX;
        const char* zb = reinterpret_cast<const char*>(p.dz + (size_t)p0 * K);
        const char* ob = reinterpret_cast<const char*>(p.dx + (size_t)p0 * C);
        const char* ab = PF ? reinterpret_cast<const char*>(p.a + (size_t)p0 * CIN) : nullptr;
        const uint8_t* mk = p.mask + (size_t)p0 * (C / 8);
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) {
            const int px = pg * 16 + li, pc = px < npx ? px : npx - 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dzf[pg][ks] = *reinterpret_cast<const bf16x8*>(zb + (unsigned)((pc * K + ks * 32 + lg * 8) * 2));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = zpx + 8 * i, pc = px < npx ? px : npx - 1;
            old[i] = *reinterpret_cast<const bf16x8*>(ob + (unsigned)((pc * C + zch * 8) * 2));
            mb[i] = mk[(unsigned)(pc * (C / 8) + zch)];
            if constexpr (PF) ra[i] = *reinterpret_cast<const bf16x8*>(ab + (unsigned)((pc * CIN + zch * 8) * 2));
            if constexpr (SECOND)
                rzb[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(p.zb + (size_t)p0 * C) + (unsigned)((pc * C + zch * 8) * 2));
        }
    };
    const int trow = 8 * lg + (li >> 2);
    auto frag = [&](const char* base, int row_bytes, int blk) {
        const char* qq = base + trow * row_bytes + (blk * 16 + 4 * (li & 3)) * 2;
        union { s16x4_ h[2]; bf16x8 v; } f;
        f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(qq));
        f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(qq + 4 * row_bytes));
        return f.v;
    };
    issue((int)blockIdx.x < ntile ? (int)blockIdx.x : ntile - 1);
    // one tile; FULL (a compile-time flag: 32 pixels) keeps the stores unconditional -- a store under a per-lane condition makes every
    // wait behind it a conservative one
    auto body = [&](int tile, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const int p0 = tile * TPX;
        const int npx = FULL ? TPX : p.P - p0;
        // ---- conv: [64 channels of the slice] x [32 pixels], K = 64
        f32x4 c[2][4];
#pragma unroll
        for (int pg = 0; pg < 2; ++pg)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                c[pg][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[ct][0], dzf[pg][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < KS; ++ks) c[pg][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[ct][ks], dzf[pg][ks], c[pg][ct], 0, 0, 0);
            }
        // raw tile as bf16 (the tile kernel's rounding point): lane (li, lg) holds channels 16 ct + 4 lg .. + 3 of pixel 16 pg + li
#pragma unroll
        for (int pg = 0; pg < 2; ++pg)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                union { bf16x4 b; s16x4_ s; } u;
                u.b = f32_to_bf4(c[pg][ct]);
                *reinterpret_cast<s16x4_*>(zs + (pg * 16 + li) * ZROW + (ct * 16 + lg * 4) * 2) = u.s;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // (own LDS writes landed; no other wave touches this area)
        // ---- epilogue: + identity gradient, mask, store, sum; g' back into the area; the a rows beside it
        char* ob = reinterpret_cast<char*>(p.dx + (size_t)p0 * C);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = zpx + 8 * i;
            char* zp = zs + px * ZROW + zch * 16;
            union { struct { s16x4_ a, b; } s; bf16x8 v; } u;
            u.s.a = *reinterpret_cast<const s16x4_*>(zp);
            u.s.b = *reinterpret_cast<const s16x4_*>(zp + 8);
            f32x8 f = bf8_to_f32(u.v) + bf8_to_f32(old[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (mb[i] >> j) & 1u ? f[j] : 0.f;
            bf16x8 v = f32_to_bf8(f);
            if (FULL) *reinterpret_cast<bf16x8*>(ob + (unsigned)((px * C + zch * 8) * 2)) = v;
            else if (px >= npx) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            else *reinterpret_cast<bf16x8*>(ob + (unsigned)((px * C + zch * 8) * 2)) = v;
            const f32x8 gq = bf8_to_f32(v);
#pragma unroll
            for (int j = 0; j < 8; ++j) sa[j] += gq[j];
            if constexpr (SECOND) {
                const f32x8 z2 = bf8_to_f32(rzb[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) sb[j] += gq[j] * (z2[j] - mu2[j]) * is2[j];
            }
            if constexpr (PF) {
                u.v = v;
                *reinterpret_cast<s16x4_*>(zp) = u.s.a;
                *reinterpret_cast<s16x4_*>(zp + 8) = u.s.b;
            }
        }
        if constexpr (PF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = zpx + 8 * i;
            bf16x8 v = ra[i];
            if (lazy) {
                const f32x8 sc = load_f32x8(s_vec + zch * 8), sh = load_f32x8(s_vec + CIN + zch * 8);
                f32x8 f = bf8_to_f32(v);
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] = clamp_act(fmaf(f[k], sc[k], sh[k]), alo, ahi);
                v = f32_to_bf8(f);
            }
            union { struct { s16x4_ a, b; } s; bf16x8 v; } u;
            u.v = FULL || px < npx ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<s16x4_*>(xs + px * XROW + zch * 16) = u.s.a;
            *reinterpret_cast<s16x4_*>(xs + px * XROW + zch * 16 + 8) = u.s.b;
        }
        }
        // (unconditional request of the workgroup's next tile; past the end: this tile again, unused)
        issue(tile + (int)gridDim.x < ntile ? tile + (int)gridDim.x : tile);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- product: P[slice][:] += g'^T a over the tile's 32 pixels
        if constexpr (PF) {
            bf16x8 fb[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) fb[nt] = frag(xs, XROW, nt);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bf16x8 fa = frag(zs, ZROW, mt);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // (the transpose reads are done before the next tile overwrites the area)
    };
    {
        const int nfull = p.P / TPX;                                         // (at most one partial tile: the last)
        int tile = blockIdx.x;
#pragma unroll 1
        for (; tile < nfull; tile += gridDim.x) body(tile, std::true_type{});
        if (tile < ntile) body(tile, std::false_type{});
    }
    // ---- this wave's slice of the workgroup's partial product (disjoint slices: no fold)
    if constexpr (PF) {
    float* out = p.ws + ((size_t)g * gridDim.x + blockIdx.x) * (C * CIN) + (size_t)q * 64 * CIN;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(mt * 16 + lg * 4 + r) * CIN + nt * 16 + li] = acc[mt][nt][r];
    }
    // ---- sum(g'): the eight lanes that share a chunk (lane % 8) fold their pixels, one exact publication per channel and wave
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = sa[j];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 8 && v != 0.f) stat_publish(p.sums + q * 64 + lane * 8 + j, 2 * C, blockIdx.x & (ADAMML_STAT_SLOTS - 1), v);
        if constexpr (SECOND) {
            if (lane < 8 && v != 0.f) stat_publish(p.sums_b + q * 64 + lane * 8 + j, 2 * C, blockIdx.x & (ADAMML_STAT_SLOTS - 1), v);
            float v2 = sb[j];
            v2 += __shfl_xor(v2, 8, 64);
            v2 += __shfl_xor(v2, 16, 64);
            v2 += __shfl_xor(v2, 32, 64);
            if (lane < 8 && v2 != 0.f) stat_publish(p.sums_b + C + q * 64 + lane * 8 + j, 2 * C, blockIdx.x & (ADAMML_STAT_SLOTS - 1), v2);
        }
    }
}

int rps_blocks(long P, int groups, int per_cu) {
    const long ntile = (P + TPX - 1) / TPX;
    static const long cap0 = getenv("ADAMML_RPS_CAP") ? atol(getenv("ADAMML_RPS_CAP")) : 256;                    // A/B aid: CUs
    long cap = cap0 * per_cu / (groups < 1 ? 1 : groups);
    if (cap < 1) cap = 1;
    return (int)(ntile < cap ? ntile : cap);
}

bool rps_on() { const char* e = getenv("ADAMML_RES_PROD_STREAM"); return !(e && atoi(e) == 0); }                 // A/B aid, read at every call

bool rps_1x1(const adamml_conv_desc_t* d) {
    return d && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->up <= 1 && (long)d->N * d->OH * d->OW >= 4096;
}

}  // namespace

// (declared in conv_gemm.hip, which owns the C entry points and falls back to its tile kernel)
int adamml_res_prod_stream_supported(const adamml_conv_desc_t* d, int a_channels) {
    return rps_on() && rps_1x1(d) && d->Cin == 256 && d->Cout == 64 && a_channels == CIN ? 1 : 0;
}

size_t adamml_res_prod_stream_workspace(const adamml_conv_desc_t* d) {
    const int groups = d->groups < 1 ? 1 : d->groups;
    return (size_t)groups * rps_blocks((long)d->N * d->OH * d->OW, groups, 2) * 256 * CIN * sizeof(float);
}

int adamml_res_prod_stream_launch(const adamml_conv_desc_t* d, const void* dz, const void* w_dgrad_packed, void* dx, const uint8_t* res_mask,
                                  double* sums_a, const void* a, const float* a_scale, const float* a_shift, int a_act, int a_gstride,
                                  float* prod, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    constexpr int C = 256;
    const int groups = d->groups < 1 ? 1 : d->groups;
    const long P = (long)d->N * d->OH * d->OW;
    const int nblk = rps_blocks(P, groups, 2);
    if (!workspace || workspace_bytes < (size_t)groups * nblk * C * CIN * sizeof(float))
        return adamml_set_error(ADAMML_EINVAL, "conv_bwd_data_res_prod: workspace too small (adamml_conv_bwd_data_res_prod_workspace)");
    RPS p;
    p.dz = (const bf16_t*)dz; p.w = (const bf16_t*)w_dgrad_packed; p.dx = (bf16_t*)dx; p.mask = res_mask; p.sums = sums_a;
    p.a = (const bf16_t*)a; p.in_scale = a_scale; p.in_shift = a_scale ? a_shift : nullptr; p.ws = (float*)workspace;
    p.in_gs = a_gstride; p.act = a_act; p.P = (int)P;
    p.zb = nullptr; p.vecb = nullptr; p.sums_b = nullptr;
    constexpr size_t lds = 2 * CIN * 4 + (size_t)4 * TPX * (ZROW + XROW);
    hipLaunchKernelGGL((res_prod_stream_kernel<4, 64, true>), dim3((unsigned)nblk, groups), dim3(256), lds, stream, p);
    int rc = adamml_check_launch("conv_bwd_data_res_prod(stream)");
    if (rc) return rc;
    return adamml_launch_split_reduce_grouped((const float*)workspace, prod, (size_t)C * CIN, nblk, groups, CIN, stream);
}

// adamml_conv_bwd_data_res in the algebraic backward's form (accumulate onto the identity-path gradient in dx, 1-bit mask, sum(g') only for
// the main branch; optionally the second BatchNorm'd operand of the add -- the downsample branch of the stage's first block -- with
// sum(g') / sum(g' zhat_b) into sums_b) at the layer-2 shape: the data gradient of a bottleneck's conv1, 128 -> 512 channels
int adamml_res_stream_supported(const adamml_conv_desc_t* d) {
    return rps_on() && rps_1x1(d) && d->Cin == 512 && d->Cout == 128 ? 1 : 0;
}

int adamml_res_stream_launch(const adamml_conv_desc_t* d, const void* dz, const void* w_dgrad_packed, void* dx, const uint8_t* res_mask, double* sums_a,
                             const void* z_b, const float* vec_b, double* sums_b, hipStream_t stream) {
    const int groups = d->groups < 1 ? 1 : d->groups;
    const long P = (long)d->N * d->OH * d->OW;
    RPS p;
    p.dz = (const bf16_t*)dz; p.w = (const bf16_t*)w_dgrad_packed; p.dx = (bf16_t*)dx; p.mask = res_mask; p.sums = sums_a;
    p.a = nullptr; p.in_scale = nullptr; p.in_shift = nullptr; p.ws = nullptr; p.in_gs = 0; p.act = 0; p.P = (int)P;
    p.zb = (const bf16_t*)z_b; p.vecb = vec_b; p.sums_b = sums_b;
    constexpr size_t lds = 2 * CIN * 4 + (size_t)8 * TPX * ZROW;
    const dim3 grid((unsigned)rps_blocks(P, groups, 1), groups);
    if (z_b) hipLaunchKernelGGL((res_prod_stream_kernel<8, 128, false, true>), grid, dim3(512), lds, stream, p);
    else hipLaunchKernelGGL((res_prod_stream_kernel<8, 128, false, false>), grid, dim3(512), lds, stream, p);
    return adamml_check_launch("conv_bwd_data_res(stream)");
}
