// Activation-stationary streaming kernels for the WIDE 1x1 convolutions of ResNet-50 layers 3-4 (models/resnet.py:94-113 over the
// stages built at models/resnet.py:150-154: conv3 256 -> 1024 / 512 -> 2048, the stride-2 downsample convs 512 -> 1024 / 1024 -> 2048, and
// -- as data gradients -- conv1 1024 <- 256 / 2048 <- 512), gfx950.
//
// conv_gemm_kernel walks 128 x 128 output tiles: with K = 256 a tile is 8 K steps behind 16 workgroup barriers, an LDS-staged epilogue and
// a statistics fold, and the activation tile is fetched again for every one of the 8-16 output-channel tiles -- these layers ran at
// 0.4-0.65 PFLOP/s and 2-3 TB/s, bound by neither roof (round 5, profiles/r05_per_layer_bench_conv.txt).  Here ("expanding" form, K <= 512):
//   * a workgroup (8 waves, one per CU) owns a SLAB of 128 (64) output channels -- its [SLAB][K] weights sit in LDS for the whole launch --
//     and a contiguous range of pixels of one BatchNorm group; the slabs of one pixel range run on the same XCD at the same time, so the
//     activation rows they all read come out of that XCD's L2;
//   * every wave streams its own 32-pixel tiles: the activation operand goes global -> registers -> MFMA (B fragment "lane (pixel li,
//     K chunk lg)" = one 16-byte NHWC load; lazy BatchNorm + activation of the producer applied in registers) and stays there for ALL
//     K steps and output-channel tiles of the slab; the request for the NEXT tile's K step k is issued the moment step k's fragment has been
//     consumed (same registers: one tile of look-ahead at no register cost); NO workgroup barrier in the loop;
//   * the bf16 output tile is staged in a wave-private LDS area (a wave's LDS operations execute in order), the statistics of the stored
//     values come off the matrix cores from that tile (ones . F and diag(F^T F)), and it leaves as 16-byte stores, 256 bytes per pixel.
// Same K order and rounding points as conv_gemm_kernel: bit-identical outputs; statistics differ in summation order only.
#include <type_traits>
#include <stdlib.h>
#include "common.h"
#include "../../include/adamml_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_;

template <int N, class F>
__device__ __forceinline__ void static_for_k(F&& f) {
    if constexpr (N > 0) {
        static_for_k<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

struct WXP {
    const bf16_t* x;         // [groups][Pin][K]   activations (forward) or output gradient (data gradient)
    const bf16_t* w;         // [N][K] bf16 pack
    const float* in_scale;   // lazy input transform (null: identity), group stride in_gs
    const float* in_shift;
    bf16_t* y;               // [groups][P][N]
    double* stats;           // EPI 0: [groups][SLOTS][2 N] or null
    int act, in_gs, K, N;
    long P, Pin;             // output / input pixels per group
    int stride, H, W, OH, OW;   // stride 2: output pixel (n, oh, ow) reads input pixel (n, 2 oh, 2 ow)
    int nslab, rpg;          // channel slabs, pixel ranges per group
    int dbg;                 // ADAMML_WIDE_DBG (probe builds): 1 skip the stores, 2 skip the statistics, 4 skip the staging
};

// EPI 0: forward / plain data gradient (store + optional statistics); EPI 2: data gradient accumulating into y
template <int KS, int SLAB, int NPG, int EPI, bool LAZY, bool S2>
__global__ __launch_bounds__(512, 1) void wide_expand_kernel(WXP p) {
    constexpr int KP = KS * 32;
    constexpr int NCT = SLAB / 16;
    constexpr int WROW = KP * 2 + 16;                // LDS bytes per weight row (+16: bank skew for the 16-lane row reads)
    constexpr int SROW = SLAB * 2 + 8;               // staging row bytes
    constexpr int CPR = SLAB / 8;                    // 16-byte chunks per staged pixel
    constexpr int TPX = NPG * 16;
    constexpr int NW = 8;
    static_assert(NPG == 2, "the statistics fragments read 32 staged pixels");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;                                                        // [SLAB][WROW]
    float* s_vec = reinterpret_cast<float*>(smem + SLAB * WROW);             // [2][KP]: scale, shift of the lazy input
    float* s_sum = s_vec + 2 * KP;                                           // [NW waves][2 * SLAB]
    char* s_stage = reinterpret_cast<char*>(s_sum + NW * 2 * SLAB);          // [NW waves][TPX][SROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // ---- work item of this workgroup: (group, pixel range, slab); the slabs of a range are neighbours on ONE XCD (workgroup id % 8)
    int slab, rid;
    {   // the (range, slab) list in range-major order, a contiguous eighth of it per XCD (common.h: xcd_contiguous)
        const unsigned w = xcd_contiguous(blockIdx.x, gridDim.x);
        rid = (int)(w / (unsigned)p.nslab);
        slab = (int)(w - (unsigned)rid * (unsigned)p.nslab);
    }
    const int g = rid / p.rpg, r = rid - g * p.rpg;
    const int n0 = slab * SLAB;
    p.x += (size_t)g * p.Pin * p.K;
    p.y += (size_t)g * p.P * p.N;
    if (p.in_scale) { p.in_scale += (size_t)g * p.in_gs; p.in_shift += (size_t)g * p.in_gs; }
    for (int i = tid; i < SLAB * (KP / 8); i += 512) {
        const int row = i / (KP / 8), ch = i - row * (KP / 8);
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (n0 + row < p.N && ch * 8 < p.K) v = *reinterpret_cast<const bf16x8*>(p.w + (size_t)(n0 + row) * p.K + ch * 8);
        *reinterpret_cast<bf16x8*>(s_w + row * WROW + ch * 16) = v;
    }
    for (int i = tid; i < KP; i += 512) {                // (channels >= K: raw 0 -> act(1 * 0 + 0) = 0)
        s_vec[i] = (p.in_scale && i < p.K) ? p.in_scale[i] : 1.f;
        s_vec[KP + i] = (p.in_scale && i < p.K) ? p.in_shift[i] : 0.f;
    }
    for (int i = tid; i < NW * 2 * SLAB; i += 512) s_sum[i] = 0.f;
    __syncthreads();

    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float alo = uniform(p.in_scale ? act_lo(p.act) : -INFINITY), ahi = uniform(p.in_scale ? act_hi(p.act) : INFINITY);
    float* csw = s_sum + wave * 2 * SLAB;
    char* stg = s_stage + wave * (TPX * SROW);
    // (pixel indices of one group fit 32 bits: the launcher checks)
    const int P = (int)p.P;
    const int ntile = (P + TPX - 1) / TPX;
    const int tpr = (ntile + p.rpg - 1) / p.rpg;
    const int t0 = r * tpr, t1 = t0 + tpr < ntile ? t0 + tpr : ntile;
    // element offset of the input row of output pixel px (S2: the even rows / columns of the input image)
    auto in_row = [&](int px) -> size_t {
        if (!S2) return (size_t)px * p.K;
        const int q = p.OH * p.OW;
        const int n = px / q, rem = px - n * q;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        return ((size_t)(n * p.H + 2 * oh) * p.W + 2 * ow) * p.K;
    };
    const bf16_t* rowp[NPG];
    auto set_rows = [&](int tile) {
#pragma unroll
        for (int pg = 0; pg < NPG; ++pg) {
            int px = tile * TPX + pg * 16 + li;
            px = px < P ? px : P - 1;
            rowp[pg] = p.x + in_row(px) + lg * 8;
        }
    };
    bf16x8 ring[KS][NPG];
    auto issue_k = [&](int k) {
#pragma unroll
        for (int pg = 0; pg < NPG; ++pg) ring[k][pg] = *reinterpret_cast<const bf16x8*>(rowp[pg] + k * 32);
    };
    union { s16x4_ h[2]; bf16x8 v; } ones;
    ones.h[0] = s16x4_{0x3F80, 0x3F80, 0x3F80, 0x3F80};
    ones.h[1] = ones.h[0];
    const int trow = 8 * lg + (li >> 2);

    int t = t0 + wave;
    {   // first tile of this wave (a wave without tiles requests a valid tile and discards it: no load sits under a condition)
        const int tf = t < ntile ? t : ntile - 1;
        set_rows(tf);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            issue_k(k);
            __builtin_amdgcn_sched_barrier(0);     // (request order = consumption order: the loop's waits then count the younger requests instead of draining them)
        }
    }
    // weight fragments of HALF a K step (NCT / 2 output-channel tiles of the slab), double-buffered: the reads of the next half step are
    // issued before the MFMAs of this one, so a read has NCT MFMAs (>= 128 matrix-pipe cycles per wave) to land -- fetched one at a
    // time in front of its two MFMAs, every fragment exposed its LDS latency (first form of this kernel: 0.155 ms at the layer-3 shape)
    constexpr int FSPLIT = EPI == 2 ? 4 : 2;                     // part steps per K step (the accumulating form keeps 32 registers of old rows: quarter steps)
    constexpr int HCT = NCT / FSPLIT;                            // (a whole K step per buffer -- 2 x NCT fragments -- does not fit beside ring + accumulators)
    bf16x8 fa[2][HCT];
    auto load_fa = [&](auto buf, int k, int h) {
#pragma unroll
        for (int c = 0; c < HCT; ++c)
            fa[decltype(buf)::value][c] = *reinterpret_cast<const bf16x8*>(s_w + ((h * HCT + c) * 16 + li) * WROW + k * 64 + lg * 16);
    };
    load_fa(std::integral_constant<int, 0>{}, 0, 0);
    float rs[NCT], rq[NCT];                                      // per-lane partial statistics (EPI 0): see the epilogue
#pragma unroll
    for (int cb = 0; cb < NCT; ++cb) rs[cb] = rq[cb] = 0.f;
    constexpr int NOLD = TPX * CPR / 64;
    bf16x8 old[EPI == 2 ? NOLD : 1];
    for (; t < t1; t += NW) {
        {   // rows of this wave's next tile (past the end: this tile again -- the requests stay unconditional)
            const int tn = t + NW;
            set_rows(tn < t1 ? tn : t);
        }
        const int npx = P - t * TPX < TPX ? P - t * TPX : TPX;
        bf16_t* yb = p.y + (size_t)t * TPX * p.N + n0;
        if (EPI == 2) {                                          // rows of the tensor accumulated into: requested a whole tile of MFMAs ahead of their use
#pragma unroll
            for (int i = 0; i < NOLD; ++i) {
                const int e = lane + 64 * i;
                int px = e / CPR;
                const int ch = e - px * CPR;
                px = px < npx ? px : npx - 1;
                old[i] = *reinterpret_cast<const bf16x8*>(yb + (size_t)px * p.N + ch * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 acc[NPG][NCT];
#pragma unroll
        for (int pg = 0; pg < NPG; ++pg)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[pg][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        static_for_k<KS>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            bf16x8 fb[NPG];
#pragma unroll
            for (int pg = 0; pg < NPG; ++pg) fb[pg] = ring[k][pg];
            if constexpr (LAZY) {                                  // act(scale * raw + shift), rounded as the other loaders round it
                const f32x8 sc = load_f32x8(s_vec + k * 32 + lg * 8), sh = load_f32x8(s_vec + KP + k * 32 + lg * 8);
#pragma unroll
                for (int pg = 0; pg < NPG; ++pg) {
                    f32x8 v = bf8_to_f32(fb[pg]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = clamp_act(fmaf(v[i], sc[i], sh[i]), alo, ahi);
                    fb[pg] = f32_to_bf8(v);
                }
            }
            issue_k(k);                                            // the next tile's K step k, into the registers just consumed
            static_for_k<FSPLIT>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                // fragments of the NEXT part step (the last one fetches step 0 for the next tile: the weights do not change)
                if constexpr (h + 1 < FSPLIT) load_fa(std::integral_constant<int, (h + 1) & 1>{}, k, h + 1);
                else load_fa(std::integral_constant<int, 0>{}, (k + 1) % KS, 0);
                __builtin_amdgcn_sched_barrier(0);                 // (the scheduler otherwise sinks all requests behind the last MFMA)
#pragma unroll
                for (int c = 0; c < HCT; ++c)
#pragma unroll
                    for (int pg = 0; pg < NPG; ++pg)
                        acc[pg][h * HCT + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[h & 1][c], fb[pg], acc[pg][h * HCT + c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        // ---- epilogue: D fragment lane (li, lg) = channels ct * 16 + lg * 4 .. + 3 of pixel pg * 16 + li -> wave-private staging tile
#pragma unroll
        for (int pg = 0; pg < NPG; ++pg) {
            const bool live = pg * 16 + li < npx;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                bf16x4 v = f32_to_bf4(acc[pg][ct]);
                if (!live) v = bf16x4{0, 0, 0, 0};
                *reinterpret_cast<bf16x4*>(stg + (pg * 16 + li) * SROW + (ct * 16 + lg * 4) * 2) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the wave's own LDS writes have landed; no other wave touches this area)
        if (EPI == 0 && p.stats && !(p.dbg & 2)) {
#pragma unroll
            for (int cb = 0; cb < NCT; ++cb) {
                f32x4 dsum = {0.f, 0.f, 0.f, 0.f}, dsq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int hp = 0; hp < NPG / 2 + (NPG & 1); ++hp) {             // 32 staged pixels per MFMA K (NPG = 1: rows 16 .. 31 read as zero)
                    const char* fp = stg + (hp * 32 + trow) * SROW + (cb * 16 + 4 * (li & 3)) * 2;
                    union { s16x4_ h[2]; bf16x8 v; } f;
                    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(fp));
                    if (NPG >= 2) f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(fp + 4 * SROW));
                    else f.h[1] = s16x4_{0, 0, 0, 0};
                    dsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, f.v, dsum, 0, 0, 0);
                    dsq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.v, f.v, dsq, 0, 0, 0);
                }
                // D rows = lg * 4 + r, column = li: row 0 of dsum (lanes lg == 0) holds the column sums, the diagonal of dsq sits in the lanes
                // with li >> 2 == lg.  Every lane adds ITS element into a register, tile after tile (fixed order); the owner lanes publish
                // theirs at the end -- a read-modify-write of the wave's LDS row per block and tile was a chain of 16 dependent LDS round trips
                const int rr = li & 3;
                rs[cb] += dsum[0];
                rq[cb] += rr == 0 ? dsq[0] : rr == 1 ? dsq[1] : rr == 2 ? dsq[2] : dsq[3];
            }
        }
        // ---- the tile leaves as 16-byte stores: CPR lanes cover the SLAB * 2 contiguous bytes of one pixel
        auto store_tile = [&](auto full) {
            constexpr bool FULL = decltype(full)::value;
#pragma unroll
            for (int i = 0; i < NOLD; ++i) {
                const int e = lane + 64 * i;
                const int px = e / CPR, ch = e - px * CPR;
                if (FULL || px < npx) {                           // (two 8-byte LDS reads: the staging rows are 8-byte aligned only)
                    union { struct { s16x4_ a, b; } s; bf16x8 v; } o;
                    o.s.a = *reinterpret_cast<const s16x4_*>(stg + px * SROW + ch * 16);
                    o.s.b = *reinterpret_cast<const s16x4_*>(stg + px * SROW + ch * 16 + 8);
                    if (EPI == 2) o.v = f32_to_bf8(bf8_to_f32(o.v) + bf8_to_f32(old[i]));
                    *reinterpret_cast<bf16x8*>(yb + (size_t)px * p.N + ch * 8) = o.v;
                }
            }
        };
        // (a full tile stores unconditionally: a store under a per-lane condition makes every later wait of the loop a vmcnt(0))
        if (p.dbg & 1) {} else
        if (npx == TPX) store_tile(std::true_type{}); else store_tile(std::false_type{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (staging reads done before the next tile overwrites the area)
    }
    if (EPI == 0 && p.stats) {
#pragma unroll
        for (int cb = 0; cb < NCT; ++cb) {
            if (lg == 0) csw[cb * 16 + li] = rs[cb];
            if ((li >> 2) == lg) csw[SLAB + cb * 16 + li] = rq[cb];
        }
        __syncthreads();
        for (int i = tid; i < 2 * SLAB; i += 512) {          // wave rows folded in wave order, one exact add per channel and workgroup (common.h)
            float v = s_sum[i];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += s_sum[w * 2 * SLAB + i];
            const int which = i >= SLAB, c = n0 + (which ? i - SLAB : i);
            if (c < p.N)
                stat_publish(p.stats + (size_t)g * ADAMML_STAT_SLOTS * 2 * p.N + (which ? p.N : 0) + c, 2 * p.N, blockIdx.x & (ADAMML_STAT_SLOTS - 1), v);
        }
    }
}

template <int KS, int SLAB, int NPG, int EPI, bool LAZY, bool S2>
int wide_expand_launch(WXP& p, int groups, hipStream_t stream) {
    constexpr int KP = KS * 32;
    constexpr size_t lds = (size_t)SLAB * (KP * 2 + 16) + 2 * KP * 4 + 8 * 2 * SLAB * 4 + (size_t)8 * NPG * 16 * (SLAB * 2 + 8);
    static_assert(lds <= 160 * 1024, "wide_expand: LDS budget");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wide_expand_kernel<KS, SLAB, NPG, EPI, LAZY, S2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "conv1x1 (wide): cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    p.nslab = (p.N + SLAB - 1) / SLAB;
    p.dbg = getenv("ADAMML_WIDE_DBG") ? atoi(getenv("ADAMML_WIDE_DBG")) : 0;
    const long ntile = (p.P + NPG * 16 - 1) / (NPG * 16);
    long rpg = 256 / ((long)p.nslab * groups);               // one workgroup per CU, all of them resident at once
    if (rpg < 1) rpg = 1;
    if (rpg > (ntile + 7) / 8) rpg = (ntile + 7) / 8;        // at least one tile per wave
    if (rpg < 1) rpg = 1;
    p.rpg = (int)rpg;
    hipLaunchKernelGGL((wide_expand_kernel<KS, SLAB, NPG, EPI, LAZY, S2>), dim3((unsigned)(groups * rpg * p.nslab)), dim3(512), lds, stream, p);
    return adamml_check_launch("conv1x1 (wide stream)");
}

bool wide_on() {           // A/B aid: ADAMML_WIDE_STREAM=0 = conv_gemm_kernel.  Read at every call (no cached state: a test flips it within one process)
    const char* e = getenv("ADAMML_WIDE_STREAM");
    return !(e && e[0] == '0');
}

}  // namespace

// Shapes with an instance of the expanding form: 1x1, K = 256 or 512 input channels, at least twice as many output channels (multiple of
// the slab), stride 1 -- or stride 2 in the forward direction (downsample branch) --, enough pixels to give every wave a tile.
bool adamml_conv1x1_wide_expand_supported(const adamml_conv_desc_t* d) {
    if (!wide_on() || d->KH != 1 || d->KW != 1 || d->pad != 0 || d->up > 1) return false;
    if (d->stride != 1 && !(d->stride == 2 && !d->accumulate && d->OH == (d->H - 1) / 2 + 1 && d->OW == (d->W - 1) / 2 + 1)) return false;
    // (K = 512 -- 32 B fragments per 32-pixel tile -- does not fit the register file beside the accumulators: instances exist, measured slower
    // than conv_gemm_kernel at the layer-4 shapes, 0.136 vs 0.083 ms; ADAMML_WIDE_K512=1 enables them)
    static const bool k512 = getenv("ADAMML_WIDE_K512") && getenv("ADAMML_WIDE_K512")[0] == '1';
    if (!(d->Cin == 256 || (k512 && d->Cin == 512)) || d->Cout < 2 * d->Cin || d->Cout % 128) return false;
    const long P = (long)d->N * d->OH * d->OW;
    return P >= 2048;
}

// d: forward-shaped descriptor of the launch conv_launch works with (for a data gradient: the data-gradient-shaped one)
int adamml_conv1x1_wide_expand_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                                      void* y, double* stats, hipStream_t stream) {
    WXP p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w_packed; p.in_scale = in_scale; p.in_shift = in_scale ? in_shift : nullptr;
    p.y = (bf16_t*)y; p.stats = stats; p.act = d->act; p.in_gs = d->in_gstride; p.K = d->Cin; p.N = d->Cout;
    p.P = (long)d->N * d->OH * d->OW; p.Pin = (long)d->N * d->H * d->W;
    p.stride = d->stride; p.H = d->H; p.W = d->W; p.OH = d->OH; p.OW = d->OW;
    if (p.P <= 0) return ADAMML_OK;
    const int groups = d->groups < 1 ? 1 : d->groups;
    const bool lazy = in_scale != nullptr, s2 = d->stride == 2;
    if (d->accumulate && lazy) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv1x1 (wide): an accumulating launch takes a plain operand");
    if (p.P >= (1L << 26) || p.Pin >= (1L << 26)) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv1x1 (wide): more than 2^26 pixels per group");
    if (d->Cin == 256) {
        if (s2) return lazy ? wide_expand_launch<8, 128, 2, 0, true, true>(p, groups, stream) : wide_expand_launch<8, 128, 2, 0, false, true>(p, groups, stream);
        if (d->accumulate) return wide_expand_launch<8, 128, 2, 2, false, false>(p, groups, stream);
        if (lazy) return wide_expand_launch<8, 128, 2, 0, true, false>(p, groups, stream);
        return wide_expand_launch<8, 128, 2, 0, false, false>(p, groups, stream);
    }
    if (s2) return lazy ? wide_expand_launch<16, 64, 2, 0, true, true>(p, groups, stream) : wide_expand_launch<16, 64, 2, 0, false, true>(p, groups, stream);
    if (d->accumulate) return wide_expand_launch<16, 64, 2, 2, false, false>(p, groups, stream);
    if (lazy) return wide_expand_launch<16, 64, 2, 0, true, false>(p, groups, stream);
    return wide_expand_launch<16, 64, 2, 0, false, false>(p, groups, stream);
}
