// Backward of the temporal max-pool that ends a ResNet stage (models/common.py:28-33 behind models/resnet.py:104-112) fused with the first
// product its gradient feeds: the 2-bit codes of adamml_conv_fwd_bn_add_tpool are expanded to the full-rate gradient g2 of the stage's last
// block output (what adamml_temporal_pool_bwd_code does) AND, while a tile of g2 is in the wave's LDS area, multiplied with the matching
// tile of the block's conv3 input a (lazy bn2 + ReLU): P = g2^T a [C][Cin] per group, the product the algebraic BatchNorm backward of
// conv3 needs BEFORE its coefficients exist (adamml_alg_sumfix) -- adamml_conv_bwd_weight_grouped read the 4.6 GB of g2 back for it.
//
// Barrier-free streaming structure of conv1x1_narrow.hip: the expansion is elementwise per channel, so wave q of a workgroup owns the
// 64-channel slice q of C for the workgroup's (clip, 16-pixel block) tasks -- its gy / g2 rows are 128-byte runs -- and the slice
// P[64 q .. 64 q + 63][:] in registers (4 x Cin/16 MFMA tiles) for the whole kernel; per window (two frames x 16 pixels = one K step of 32) it stages its g2 slice and the a rows
// (every wave loads the a tile itself: L1 / L2 hits) in its private LDS area and reads both MFMA operands back as hardware transpose reads
// (pixels = the reduction dimension).  g2 and sum(g2) as adamml_temporal_pool_bwd_code (g2 bit-identical); one partial P per workgroup.
#include <type_traits>
#include "common.h"
#include "../../include/adamml_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_;

struct TPP {
    const bf16_t* gy;        // [groups][clips * T/2][HW][C]
    const uint16_t* code;    // [groups][clips * T/2][HW][C/8]
    bf16_t* g2;              // [groups][clips * T][HW][C]
    double* sums;            // [groups][SLOTS][2C]: sum(g2) into the first C entries
    const bf16_t* a;         // [groups][clips * T][HW][CIN] raw conv3 input
    const float* in_scale;   // its lazy BatchNorm (group stride in_gs) or null
    const float* in_shift;
    float* ws;               // [groups][gridDim.x][C][CIN] partial products
    int in_gs, act, clips, HW, C;
};

template <int T, int CIN, int NQ, bool ALLFULL>
__global__ __launch_bounds__(NQ * 64, NQ == 4 ? 2 : 1) void tpool_bwd_prod_kernel(TPP p) {
    constexpr int To = T / 2, MT = 4, NTL = CIN / 16;
    constexpr int FPX = 16, TPX = 32;                              // pixels of a block per frame; rows of a staged tile = 2 frames x 16 pixels
    constexpr int ZROW = 64 * 2 + 8, XROW = CIN * 2 + 8;           // staging row bytes
    constexpr int XC = CIN / 8, NX = FPX * XC / 64;                // 16-byte chunks per pixel of a / loads per lane and frame
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_vec = reinterpret_cast<float*>(smem);                  // [2][CIN]
    char* s_stage = smem + 2 * CIN * 4;                             // [NQ waves][32][ZROW + XROW]
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);         // channel slice of this wave
    const int li = lane & 15, lg = lane >> 4;
    const size_t C = p.C;
    {
        const size_t pp = (size_t)g * p.clips * To * p.HW;
        p.gy += pp * C + q * 64;
        p.code += pp * (C / 8) + q * 8;
        p.g2 += (size_t)g * p.clips * T * p.HW * C + q * 64;
        p.a += (size_t)g * p.clips * T * p.HW * CIN;
        p.sums += (size_t)g * ADAMML_STAT_SLOTS * 2 * C;
    }
    for (int i = tid; i < CIN; i += NQ * 64) {
        s_vec[i] = p.in_scale ? p.in_scale[(size_t)g * p.in_gs + i] : 1.f;
        s_vec[CIN + i] = p.in_scale ? p.in_shift[(size_t)g * p.in_gs + i] : 0.f;
    }
    char* zs = s_stage + q * (TPX * (ZROW + XROW));
    char* xs = zs + TPX * ZROW;
    for (int i = lane; i < TPX * (ZROW + XROW) / 8; i += 64) reinterpret_cast<unsigned long long*>(zs)[i] = 0ull;
    __syncthreads();
    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float alo = uniform(p.in_scale ? act_lo(p.act) : -INFINITY), ahi = uniform(p.in_scale ? act_hi(p.act) : INFINITY);
    const bool lazy = p.in_scale != nullptr;
    f32x4 acc[MT][NTL];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NTL; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float sa[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sa[i] = 0.f;
    const int nblk_px = (p.HW + FPX - 1) / FPX;
    const long ntask = (long)p.clips * nblk_px;
    // this lane's two (pixel, 8-channel chunk) slots of a 16-pixel x 64-channel frame block: chunk lane % 8 of pixels lane / 8 and lane / 8 + 8
    const int zch = lane & 7, zpx = lane >> 3;

    // (uniform 64-bit bases -- task, clip, block and frame are the same for the whole wave -- plus 32-bit lane offsets: one register per
    // address; 64-bit per-lane addresses were a third of the registers the first form spilled)
    bf16x8 rx[2][NX];                                               // a rows of the two frames of a window
    auto issue_a = [&](long task, int w) {
        const int clip = (int)(task / nblk_px), blk = (int)(task - (long)clip * nblk_px);
        const int npx = p.HW - blk * FPX < FPX ? p.HW - blk * FPX : FPX;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const char* base = reinterpret_cast<const char*>(p.a + (((size_t)clip * T + 2 * w + f) * p.HW + (size_t)blk * FPX) * CIN);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int e = lane + 64 * i, px = e / XC, ch = e - px * XC;
                rx[f][i] = *reinterpret_cast<const bf16x8*>(base + (unsigned)(((px < npx ? px : npx - 1) * CIN + ch * 8) * 2));
            }
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
    {
        const long t0 = blockIdx.x < ntask ? blockIdx.x : ntask - 1;
        issue_a(t0, 0);
    }
    // one task; FULL (a compile-time flag: all 16 pixels of the block live) keeps the stores unconditional -- behind a store under a per-lane
    // condition every wait is a conservative one (the loop waited vmcnt(0) between the two frames' stores)
    auto task_body = [&](long task, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const int clip = (int)(task / nblk_px), blk = (int)(task - (long)clip * nblk_px);
        const int npx = FULL ? FPX : p.HW - blk * FPX;
        // ---- window `to` (cur) and window `to + 1` (nxt) of this lane's two slots: frame 2 to is tap 1 of cur; frame 2 to + 1 is tap 2 of
        // cur plus tap 0 of nxt.  One step of the (runtime) window loop expands both frames into the 32-row staged tile -- rows 0..15 the
        // even frame's pixels, 16..31 the odd frame's -- and multiplies it with the matching rows of a: K = 2 frames x 16 pixels.
        struct Win { bf16x8 g[2]; unsigned c[2]; };
        Win cur, nxt;
        unsigned goff[2];                                                   // byte offsets of this lane's slots within a frame / window
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int px = zpx + 8 * i, pc = px < npx ? px : npx - 1;
            goff[i] = (unsigned)((pc * (int)C + zch * 8) * 2);
        }
        auto load_win = [&](int w, Win& d) {
            const int wc = w < To ? w : To - 1;                             // (clamped: an unused request, not a conditional one)
            const size_t row0 = ((size_t)clip * To + wc) * p.HW + (size_t)blk * FPX;
            const char* gb = reinterpret_cast<const char*>(p.gy + row0 * C);
            const char* cb = reinterpret_cast<const char*>(p.code + row0 * (C / 8));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                d.g[i] = *reinterpret_cast<const bf16x8*>(gb + goff[i]);
                d.c[i] = *reinterpret_cast<const uint16_t*>(cb + (goff[i] >> 3));
            }
        };
        // one frame: expand (k0 = its tap in `cur`; odd frames add tap 0 of `nxt` when that window exists), store, stage into rows r0 ..
        auto frame = [&](int t, unsigned k0, bool add_next, int r0, const bf16x8 (&ra)[NX]) {
            char* ob = reinterpret_cast<char*>(p.g2 + (((size_t)clip * T + t) * p.HW + (size_t)blk * FPX) * C);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int px = zpx + 8 * i;
                const f32x8 g0 = bf8_to_f32(cur.g[i]);
                f32x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ((cur.c[i] >> (2 * j)) & 3u) == k0 ? g0[j] : 0.f;
                if (add_next) {
                    const f32x8 g1 = bf8_to_f32(nxt.g[i]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += ((nxt.c[i] >> (2 * j)) & 3u) == 0u ? g1[j] : 0.f;
                }
                bf16x8 gb = f32_to_bf8(v);
                if (FULL) *reinterpret_cast<bf16x8*>(ob + goff[i]) = gb;
                else if (px >= npx) gb = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                else *reinterpret_cast<bf16x8*>(ob + goff[i]) = gb;
                const f32x8 gq = bf8_to_f32(gb);
#pragma unroll
                for (int j = 0; j < 8; ++j) sa[j] += gq[j];
                union { struct { s16x4_ a, b; } s; bf16x8 v; } u;
                u.v = gb;
                *reinterpret_cast<s16x4_*>(zs + (r0 + px) * ZROW + zch * 16) = u.s.a;
                *reinterpret_cast<s16x4_*>(zs + (r0 + px) * ZROW + zch * 16 + 8) = u.s.b;
            }
            // the a rows of frame t (lazy transform on the way), rows past the end zero
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int e = lane + 64 * i, px = e / XC, ch = e - px * XC;
                bf16x8 v = ra[i];
                if (lazy) {
                    const f32x8 sc = load_f32x8(s_vec + ch * 8), sh = load_f32x8(s_vec + CIN + ch * 8);
                    f32x8 f = bf8_to_f32(v);
#pragma unroll
                    for (int k = 0; k < 8; ++k) f[k] = clamp_act(fmaf(f[k], sc[k], sh[k]), alo, ahi);
                    v = f32_to_bf8(f);
                }
                union { struct { s16x4_ a, b; } s; bf16x8 v; } u;
                u.v = FULL || px < npx ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<s16x4_*>(xs + (r0 + px) * XROW + ch * 16) = u.s.a;
                *reinterpret_cast<s16x4_*>(xs + (r0 + px) * XROW + ch * 16 + 8) = u.s.b;
            }
        };
        load_win(0, cur);
        load_win(1, nxt);
#pragma unroll 1
        for (int to = 0; to < To; ++to) {
            frame(2 * to, 1u, false, 0, rx[0]);
            frame(2 * to + 1, 2u, to + 1 < To, FPX, rx[1]);
            {
                // (unconditional request of the next window's a rows: this task's next window, or window 0 of the workgroup's next task)
                long tn = task;
                int wn = to + 1;
                if (wn == To) { wn = 0; tn = task + gridDim.x < ntask ? task + gridDim.x : task; }
                issue_a(tn, wn);
            }
            cur = nxt;
            load_win(to + 2, nxt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (own LDS writes landed; no other wave touches this area)
            if constexpr (NTL <= MT) {
                bf16x8 fb[NTL];
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) fb[nt] = frag(xs, XROW, nt);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8 fa = frag(zs, ZROW, mt);
#pragma unroll
                    for (int nt = 0; nt < NTL; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[nt], acc[mt][nt], 0, 0, 0);
                }
            } else {
                bf16x8 fa[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) fa[mt] = frag(zs, ZROW, mt);
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
                    const bf16x8 fb = frag(xs, XROW, nt);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb, acc[mt][nt], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (the transpose reads are done before the next window overwrites the area)
        }
    };
    // (ALLFULL: HW % 16 == 0, the launcher's choice -- one path per kernel instance: with both in one kernel the allocator spilled 44 registers)
#pragma unroll 1
    for (long task = blockIdx.x; task < ntask; task += gridDim.x) task_body(task, std::integral_constant<bool, ALLFULL>{});
    // ---- this wave's slice of the workgroup's partial product (disjoint slices: no fold)
    float* out = p.ws + ((size_t)g * gridDim.x + blockIdx.x) * (C * CIN) + (size_t)q * 64 * CIN;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(mt * 16 + lg * 4 + r) * CIN + nt * 16 + li] = acc[mt][nt][r];
    // ---- sum(g2): the eight lanes that share a chunk (lane % 8) fold their pixels, one exact publication per channel and wave
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = sa[j];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 8 && v != 0.f) stat_publish(p.sums + q * 64 + lane * 8 + j, 2 * C, blockIdx.x & (ADAMML_STAT_SLOTS - 1), v);
    }
}

template <int T, int CIN, int NQ, bool ALLFULL>
int tpp_launch(const TPP& p, int groups, int nblk, hipStream_t stream) {
    constexpr size_t lds = 2 * CIN * 4 + (size_t)NQ * 32 * ((64 * 2 + 8) + (CIN * 2 + 8));
    static AdamLdsOnce attr_once;                    // (per device: common.h)
    const int attr_dev = adamml_current_device();
    if (!attr_once.test(attr_dev)) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tpool_bwd_prod_kernel<T, CIN, NQ, ALLFULL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "temporal_pool_bwd_code_prod: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        }
        attr_once.set(attr_dev);
    }
    hipLaunchKernelGGL((tpool_bwd_prod_kernel<T, CIN, NQ, ALLFULL>), dim3((unsigned)nblk, groups), dim3(NQ * 64), lds, stream, p);
    return adamml_check_launch("temporal_pool_bwd_code_prod");
}

int tpp_blocks(int NB, int HW, int C, int groups) {
    const long ntask = (long)NB * ((HW + 15) / 16);
    // one partial [C][Cin] per workgroup: two workgroups per CU over all groups at C = 256 (four waves each), one at C = 512
    static const long cap0 = getenv("ADAMML_TPP_CAP") ? atol(getenv("ADAMML_TPP_CAP")) : 512;                    // A/B aid
    long cap = (C == 256 ? cap0 : cap0 / 2) / (groups < 1 ? 1 : groups);
    if (cap < 1) cap = 1;
    return (int)(ntask < cap ? ntask : cap);
}

}  // namespace

extern "C" int adamml_temporal_pool_bwd_code_prod_supported(int T, int C, int Cin) {
    static const bool on = !(getenv("ADAMML_TPOOL_BWD_PROD") && atoi(getenv("ADAMML_TPOOL_BWD_PROD")) == 0);            // A/B aid
    // (the stage-2 form, T = 4 / C = 512 / Cin = 128 with eight waves and 4 x 8 accumulator tiles, does not fit the register file: not built)
    return on && T == 8 && C == 256 && Cin == 64 ? 1 : 0;
}

extern "C" size_t adamml_temporal_pool_bwd_code_prod_workspace(int NB, int T, int HW, int C, int Cin, int groups) {
    if (!adamml_temporal_pool_bwd_code_prod_supported(T, C, Cin)) return 0;
    return (size_t)(groups < 1 ? 1 : groups) * tpp_blocks(NB, HW, C, groups) * C * Cin * sizeof(float);
}

extern "C" int adamml_temporal_pool_bwd_code_prod(const void* g_y, const uint16_t* code, void* g2, double* sums_a, const void* a,
                                                  const float* in_scale, const float* in_shift, int in_gstride, int in_act, float* prod,
                                                  void* workspace, size_t workspace_bytes, int NB, int T, int HW, int C, int Cin, int groups,
                                                  hipStream_t stream) {
    if (!adamml_temporal_pool_bwd_code_prod_supported(T, C, Cin))
        return adamml_set_error(ADAMML_EUNSUPPORTED, "temporal_pool_bwd_code_prod: (T, C, Cin) = (8, 256, 64)");
    if (!g_y || !code || !g2 || !sums_a || !a || !prod || !workspace) return adamml_set_error(ADAMML_EINVAL, "temporal_pool_bwd_code_prod: null argument");
    if (NB < 1 || HW < 1) return ADAMML_OK;
    if (groups < 1) groups = 1;
    const int nblk = tpp_blocks(NB, HW, C, groups);
    if (workspace_bytes < (size_t)groups * nblk * C * Cin * sizeof(float))
        return adamml_set_error(ADAMML_EINVAL, "temporal_pool_bwd_code_prod: workspace too small (adamml_temporal_pool_bwd_code_prod_workspace)");
    TPP p;
    p.gy = (const bf16_t*)g_y; p.code = code; p.g2 = (bf16_t*)g2; p.sums = sums_a; p.a = (const bf16_t*)a;
    p.in_scale = in_scale; p.in_shift = in_scale ? in_shift : nullptr; p.ws = (float*)workspace;
    p.in_gs = in_gstride; p.act = in_act; p.clips = NB; p.HW = HW; p.C = C;
    int rc = HW % 16 == 0 ? tpp_launch<8, 64, 4, true>(p, groups, nblk, stream) : tpp_launch<8, 64, 4, false>(p, groups, nblk, stream);
    if (rc) return rc;
    return adamml_launch_split_reduce_grouped((const float*)workspace, prod, (size_t)C * Cin, nblk, groups, Cin, stream);
}
