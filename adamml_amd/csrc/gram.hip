// Gram matrix G = a^T a [C][C] and column sums s = sum_p a [C] of a lazily normalised activation a = act(scale x + shift)
// (rounded to bf16 exactly as the conv loaders stage it), per BatchNorm group, in ONE streaming pass over x.
//
// Feeds (i) the train-mode statistics of the fused conv3 + BatchNorm + add kernel (adamml_gram_stats: sum z = W s,
// sum z^2 = diag(W G W^T) for z = W a, models/resnet.py:103-111 without ever storing z) and (ii) the algebraic BatchNorm
// backward (dW = .. + B (.) (W G) + C (x) s).  Before this kernel the two came from adamml_conv_bwd_weight_grouped with
// dz = x (the generic weight-gradient kernel: both operands fetched and transformed separately, one K step of look-ahead --
// 1.8 TB/s on the layer-1 shape) plus adamml_lazy_colsum (a second pass over x).
//
// Structure: a workgroup streams a contiguous pixel range of one group in K steps of 32 pixels.  Each thread owns one 16-byte
// chunk column (its scale / shift live in registers), loads run D steps ahead in a register ring, the transformed tile is
// staged once in LDS ([32 pixels][C] bf16, the transpose-read image of conv_wgrad_kernel) and BOTH MFMA operands of a Gram
// block are the SAME transposed fragment: A[i][k] = a[k][i] and B[k][j] = a[k][j] have identical lane layouts, so
// C/16 fragments per K step serve all (C/16)^2 blocks.  Column sums ride on the matrix cores as ones * F.  Partials go to
// a workspace with plain stores ([group][split][C*C + C]) and are summed in a fixed order by gram_reduce_kernel, so the
// result does not depend on the arrival order (deterministic mode needs no special path).
#include "common.h"
#include "../../include/adamml_hip.h"
#include <type_traits>

namespace {

constexpr int NTHREADS = 256;

struct GramP {
    const bf16_t* x;
    const float* scale;
    const float* shift;
    float* ws;
    size_t gx;               // elements between groups of x
    int gstride, act;
    int P, ppb, nsplit;
};

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// 8-byte-unit XOR swizzle of the [32 pixels][CH channels] transpose-read image (same as conv_gemm.hip: the two 32-lane
// service groups of ds_read_b64_tr_b16 touch 64 distinct banks)
template <int CH>
__device__ __forceinline__ int tr_swz(int row) {
    if (CH >= 128) return ((row & 3) | (((row >> 3) & 1) << 2)) << 2;
    return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 2;
}

template <int C, int D>
__global__ __launch_bounds__(NTHREADS) void gram_colsum_kernel(GramP p) {
    constexpr int ROWB = C * 2;                // bytes per LDS row (one pixel)
    constexpr int TILE_BYTES = 32 * ROWB;
    constexpr int CH = C / 8;                  // 16-byte chunks per row
    constexpr int NL = (32 * CH) / NTHREADS;   // chunks per thread per K step
    constexpr int RSTEP = NTHREADS / CH;       // rows between a thread's chunks
    constexpr int NB = C / 16;                 // 16-channel blocks
    constexpr int RB = NB / 4;                 // row blocks per wave
    // C = 256: all 16 x 16 Gram blocks would take 256 accumulator registers per lane (measured: 202 of them spilled).  G is symmetric:
    // row block r computes the NT = 9 column blocks r, r + 1, .. r + 8 (mod 16) -- every unordered pair of blocks once (the pairs at
    // distance 8 twice) -- with the column fragment read from LDS per product instead of being held for all row blocks, and the
    // partial's mirror blocks are filled in when it is stored.
    constexpr bool SYM = C >= 256;
    constexpr int NT = SYM ? NB / 2 + 1 : NB;  // column blocks per row block
    static_assert(NL >= 1 && RB >= 1, "C must be 64, 128 or 256");
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES];

    const int grp = blockIdx.y, split = blockIdx.x;
    p.x += (size_t)grp * p.gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = tid % CH, row0 = tid / CH;
    const int ps = split * p.ppb;
    const int pe = min(p.P, ps + p.ppb);
    const int nk = pe > ps ? (pe - ps + 31) / 32 : 0;

    f32x8 sc, sh;
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
    if (p.scale) {
        sc = load_f32x8(p.scale + (size_t)grp * p.gstride + ch * 8);
        sh = load_f32x8(p.shift + (size_t)grp * p.gstride + ch * 8);
    }
    const float lo = p.scale ? act_lo(p.act) : -INFINITY, hi = p.scale ? act_hi(p.act) : INFINITY;

    bf16x8 r[D][NL];
    auto issue = [&](auto slot_c, int kt) {
        constexpr int SL = decltype(slot_c)::value;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int pp = ps + kt * 32 + row0 + l * RSTEP;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (pp < pe) v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p.x + (size_t)pp * C + ch * 8));
            r[SL][l] = v;
        }
    };
    auto store_tile = [&](auto slot_c, int kt) {
        constexpr int SL = decltype(slot_c)::value;
        char* base = smem + (kt & 1) * TILE_BYTES;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int row = row0 + l * RSTEP;
            const bool ok = ps + kt * 32 + row < pe;
            f32x8 f = bf8_to_f32(r[SL][l]);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = ok ? clamp_act(fmaf(f[i], sc[i], sh[i]), lo, hi) : 0.f;      // rows past the range add nothing
            *reinterpret_cast<bf16x8*>(base + row * ROWB + ((ch ^ (tr_swz<C>(row) >> 1)) << 4)) = f32_to_bf8(f);
        }
    };

    f32x4 acc[RB][NT], asum[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        asum[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    union { s16x4 h[2]; bf16x8 v; } ones;
    ones.h[0] = s16x4{0x3F80, 0x3F80, 0x3F80, 0x3F80};
    ones.h[1] = ones.h[0];

    const int li = lane & 15, lg = lane >> 4;
    // transpose-read addressing (conv_wgrad_kernel): lane li of a 16-lane group supplies the 8-byte unit
    // [pixel row 8*lg + (li>>2) (+4)][channels 4*(li&3) ..+3] and receives channel li of pixels 8*lg .. 8*lg+7
    const int trow = 8 * lg + (li >> 2), tq = li & 3;
    const int a_lo = trow * ROWB, a_hi = (trow + 4) * ROWB;
    const int x_lo = tr_swz<C>(trow), x_hi = tr_swz<C>(trow + 4);
    auto compute = [&](int buf) {
        const char* base = smem + buf * TILE_BYTES;
        auto frag = [&](int blk) {
            const int u = blk * 4 + tq;
            union { s16x4 h[2]; bf16x8 v; } cvt;
            cvt.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + a_lo + ((u ^ x_lo) << 3)));
            cvt.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + a_hi + ((u ^ x_hi) << 3)));
            return cvt.v;
        };
        bf16x8 f[SYM ? 1 : NB];
        if constexpr (!SYM) {
#pragma unroll
            for (int t = 0; t < NB; ++t) f[t] = frag(t);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            // the row fragment of this wave is read on its own (a per-wave index into f[] would not be a static register index)
            const int rb = wave * RB + i;
            const bf16x8 fr = frag(rb);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if constexpr (SYM) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr, frag((rb + j) & (NB - 1)), acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr, f[j], acc[i][j], 0, 0, 0);
            }
            asum[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, fr, asum[i], 0, 0, 0);
        }
    };

    static_for<D>([&](auto s) {
        if ((int)decltype(s)::value < nk) issue(s, (int)decltype(s)::value);
    });
    for (int kt0 = 0; kt0 < nk; kt0 += D) {
        static_for<D>([&](auto s) {
            const int kt = kt0 + (int)decltype(s)::value;
            if (kt < nk) {                               // uniform
                store_tile(s, kt);                       // waits (counted vmcnt) only for this slot's loads
                if (kt + D < nk) issue(s, kt + D);
                __syncthreads();                         // tile kt visible; everyone is past compute(kt - 1)
                compute(kt & 1);
            }
        });
    }
    // partial of this workgroup: G rows lg*4 + r of row block (wave*RB + i), column li of block j; sums from row 0 of ones * F
    float* out = p.ws + ((size_t)grp * p.nsplit + split) * (C * C + C);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int rb = wave * RB + i;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int cb = SYM ? (rb + j) & (NB - 1) : j;
#pragma unroll
            for (int q = 0; q < 4; ++q) out[(size_t)(rb * 16 + lg * 4 + q) * C + cb * 16 + li] = acc[i][j][q];
            if (SYM && j != 0 && j != NB / 2) {          // mirror block (the diagonal is its own mirror; distance-8 pairs are computed from both sides)
#pragma unroll
                for (int q = 0; q < 4; ++q) out[(size_t)(cb * 16 + li) * C + rb * 16 + lg * 4 + q] = acc[i][j][q];
            }
        }
        if (lg == 0) out[C * C + rb * 16 + li] = asum[i][0];
    }
}

// G[g][i], s[g][i] = sum over the splits, in split order.  The ~200 partials of a group are summed in fp64 and rounded once: the
// train-mode variance of conv_bn_add is diag(W G W^T) / n - mean^2 (adamml_gram_stats, fp64 from here on), and with all-positive
// post-ReLU inputs that difference cancels -- the split sum should not add its own fp32 rounding to the partials' (ADVICE round 2).
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* ws, float* G, float* s, int n_g, int n_s, int nsplit) {
    const int i = blockIdx.x * 256 + threadIdx.x, grp = blockIdx.y;
    const int n = n_g + n_s;
    if (i >= n) return;
    const float* src = ws + (size_t)grp * nsplit * n + i;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int k = 0;
    for (; k + 4 <= nsplit; k += 4) {
        a0 += src[(size_t)k * n]; a1 += src[(size_t)(k + 1) * n]; a2 += src[(size_t)(k + 2) * n]; a3 += src[(size_t)(k + 3) * n];
    }
    for (; k < nsplit; ++k) a0 += src[(size_t)k * n];
    const float v = (float)((a0 + a1) + (a2 + a3));
    if (i < n_g) G[(size_t)grp * n_g + i] = v;
    else s[(size_t)grp * n_s + (i - n_g)] = v;
}

int gram_splits(size_t P, int groups, int* ppb, int C) {
    // ~4 workgroups per CU over all groups; C = 256: ONE per CU -- its 64 Gram blocks per wave take 256 accumulator registers (one
    // workgroup per CU fits), and a partial is 263 KB: a thousand of them would cost as much traffic as the pass over the input
    int nsplit = ((C >= 256 ? 256 : 1024) + groups - 1) / groups;
    size_t per = ((P + nsplit - 1) / nsplit + 31) / 32 * 32;
    if (per < 256) per = 256;
    *ppb = (int)per;
    return (int)((P + per - 1) / per);
}

}  // namespace

extern "C" int adamml_gram_colsum_supported(int C) { return C == 64 || C == 128 || C == 256; }

extern "C" size_t adamml_gram_colsum_workspace(size_t P, int C, int groups) {
    if (!adamml_gram_colsum_supported(C) || !P) return 0;
    if (groups < 1) groups = 1;
    int ppb;
    const int nsplit = gram_splits(P, groups, &ppb, C);
    return (size_t)groups * nsplit * ((size_t)C * C + C) * sizeof(float);
}

extern "C" int adamml_gram_colsum(const void* x, const float* scale, const float* shift, int gstride, int act, float* G, float* s,
                                  size_t P, int C, int groups, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!x || !G || !s) return adamml_set_error(ADAMML_EINVAL, "gram_colsum: null argument");
    if (!adamml_gram_colsum_supported(C)) return adamml_set_error(ADAMML_EUNSUPPORTED, "gram_colsum: C must be 64, 128 or 256 (got %d)", C);
    if (groups < 1) groups = 1;
    if (P >= ((size_t)1 << 31) / C) return adamml_set_error(ADAMML_EUNSUPPORTED, "gram_colsum: group exceeds 2^31 elements");
    if (!P) {
        (void)hipMemsetAsync(G, 0, (size_t)groups * C * C * sizeof(float), stream);
        (void)hipMemsetAsync(s, 0, (size_t)groups * C * sizeof(float), stream);
        return ADAMML_OK;
    }
    GramP p;
    p.nsplit = gram_splits(P, groups, &p.ppb, C);
    if (!workspace || workspace_bytes < adamml_gram_colsum_workspace(P, C, groups))
        return adamml_set_error(ADAMML_EINVAL, "gram_colsum: workspace too small (need %zu bytes)", adamml_gram_colsum_workspace(P, C, groups));
    p.x = (const bf16_t*)x; p.scale = scale; p.shift = scale ? shift : nullptr; p.ws = (float*)workspace;
    p.gx = P * C; p.gstride = gstride; p.act = act; p.P = (int)P;
    dim3 grid(p.nsplit, groups);
    if (C == 64) hipLaunchKernelGGL((gram_colsum_kernel<64, 8>), grid, dim3(NTHREADS), 0, stream, p);
    else if (C == 256) hipLaunchKernelGGL((gram_colsum_kernel<256, 2>), grid, dim3(NTHREADS), 0, stream, p);
    else hipLaunchKernelGGL((gram_colsum_kernel<128, 4>), grid, dim3(NTHREADS), 0, stream, p);
    int rc = adamml_check_launch("gram_colsum");
    if (rc) return rc;
    const int n = C * C + C;
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((n + 255) / 256, groups), dim3(256), 0, stream, (const float*)workspace, G, s, C * C, C, p.nsplit);
    return adamml_check_launch("gram_reduce");
}
