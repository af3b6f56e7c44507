// Activation-stationary kernel for the WIDE expanding 1x1 convolutions of ResNet-50 layers 2-3 (models/resnet.py:94-113 at the widths of
// models/resnet.py:148-152: layer-3 conv3 256 -> 1024, the stride-2 downsample conv 256 -> 512 of layer 2, and -- as data gradients -- the
// conv1 of layer 3, 1024 <- 256 / 512 <- 256), gfx950.
//
// conv_gemm_kernel walks 128 x 128 output tiles: with K = 256 a tile is 8 K steps behind 16 workgroup barriers, an LDS-staged epilogue and
// a statistics fold, and the activation tile is fetched again for every one of the 8 output-channel tiles -- these layers ran at
// 0.4-0.65 PFLOP/s and 2-3 TB/s, bound by neither roof (round 5, profiles/r05_per_layer_bench_conv.txt).  Here:
//   * a workgroup (8 waves, one per CU, persistent) belongs to one BatchNorm group and walks sets of 8 x 32 pixels; every wave keeps the B
//     fragments of ITS 32 pixels -- all K steps, the lazy BatchNorm + activation of the producer applied ONCE -- in registers for every
//     output channel of the layer: activations are read from memory once, transformed once;
//   * the weights stream past them: slab after slab of 64 output channels goes global -> registers -> LDS (two buffers, XOR-swizzled rows:
//     conflict-free fragment reads); the request for a slab is issued a whole slab step ahead of its ds_write;
//   * two wave groups run the same loop { barrier; 64 MFMAs of slab s; barrier; epilogue of slab s } one barrier apart: while one group
//     multiplies, the other stages its bf16 tile in wave-private LDS, takes the statistics of the stored values off the matrix cores
//     (ones . F, diag(F^T F)), stores 128 bytes per pixel and moves its share of the next weight slab;
//   * statistics: per-wave slots per slab step, folded in wave order into the workgroup's [2 N] accumulator one step later (fixed order:
//     reproducible), ONE exact publication per channel and workgroup.
// Same K order and rounding points as conv_gemm_kernel: bit-identical outputs; statistics differ in summation order only
// (tests/test_kernels_gpu.py::test_conv1x1_wide_stream_equals_conv_gemm flips ADAMML_WIDE_STREAM within one process).
//
// History of the round (all measured at the layer-3 conv3 shape, 5 x 144 frames x 14^2, 256 -> 1024, lazy input + statistics; conv_gemm_kernel
// 0.188-0.194 ms; appendix A-15): (1) one 128-channel slab per workgroup, weights LDS-resident, waves streaming pixel tiles with no barrier:
// 0.154 ms -- fabric reads = the input ONCE (the 8 slabs of a pixel range share it through their XCD's L2, PMC), but the lazy transform, the
// loads and the address arithmetic are repeated per slab: VALU 45 % / LDS 48 % / MFMA 29 % of the launch, and its waits drained the stores of
// every tile (a store under a per-lane condition turns every later vmcnt of the loop into the conservative count); (2) all slabs per
// workgroup, one barrier per slab step: 0.150-0.159 ms -- all 8 waves in the same phase at the same time, the matrix pipe idle through
// every epilogue; (3) two wave groups in opposite phases: 0.157 ms -- the phase probe (tools/wide_phase_probe.py) showed 1 900-2 200 of a
// step's 7 000 cycles in "stage + statistics": every 16-channel block was its own chain of LDS read -> MFMA -> exec-masked LDS write; (4)
// branch-free statistics (all reads, all MFMAs, unconditional slot writes with per-lane sink entries) + swizzled weight rows: 0.142 ms, a
// slab step 6 200 cycles (MFMA phase 1 800-2 100 for 1 024 of matrix pipe: fragment reads one K step ahead do not cover the LDS latency
// under the other group's epilogue traffic; epilogue 2 100-2 750).  Forms (1)-(3) are in the history of this file (commit de74dc7 ff.).
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
    int nslab, rpg;          // channel slabs; BatchNorm groups of the launch
};

// EPI 0: forward / plain data gradient (store + optional statistics); EPI 2: data gradient accumulating into y
template <int KS, int EPI, bool LAZY, bool S2>
__global__ __launch_bounds__(512, 1) void wide_all_kernel(WXP p) {
    constexpr int KP = KS * 32;
    constexpr int SLAB = 64, NCT = 4, NPG = 2, TPX = 32, NW = 8;
    constexpr int WROW = KP * 2;                     // LDS bytes per weight row; 16-byte chunk c of row r sits at chunk c ^ (r & 15): the 16 lanes
                                                     // of a ds_read_b128 service group then touch 16 distinct bank quads (pitch + 16 left 43 % of the
                                                     // LDS cycles of the slab form to bank conflicts, PMC)
    constexpr int SROW = SLAB * 2 + 8;               // staging row bytes
    constexpr int CPR = SLAB / 8;
    constexpr int WLD = SLAB * KP * 2 / (512 * 16);  // 16-byte weight loads per thread and slab
    static_assert(SLAB * KP * 2 % (512 * 16) == 0, "a slab is a whole number of workgroup-wide 16-byte loads");
    constexpr int MAXN = 2048;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;                                                        // [2][SLAB][WROW]
    float* s_vec = reinterpret_cast<float*>(smem + 2 * SLAB * WROW);         // [2][KP]
    float* s_acc = s_vec + 2 * KP;                                           // [2][MAXN]: sum, sum of squares of every output channel
    constexpr int SLOTW = 2 * SLAB + 64;                                     // a wave's slot: [sum 64 | sum of squares 64 | 64 sink entries]
    float* s_slot = s_acc + 2 * MAXN;                                        // [2 buffers][NW][SLOTW]
    char* s_stage = reinterpret_cast<char*>(s_slot + 2 * NW * SLOTW);        // [NW][TPX][SROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int nslab = p.N / SLAB;
    // ---- work: a workgroup belongs to ONE BatchNorm group (blockIdx % groups: the groups interleave over the XCDs) and walks that
    // group's sets of 8 x 32 pixels with the stride of the group's workgroups: one statistics publication per workgroup
    const int P = (int)p.P;
    const int ntile = (P + TPX - 1) / TPX;
    const int nset = (ntile + NW - 1) / NW;                                  // tile sets per group
    const int groups = p.rpg;                                                // (this form: rpg carries the group count)
    const int g = blockIdx.x % groups, wi = blockIdx.x / groups, wpg = gridDim.x / groups;
    for (int i = tid; i < 2 * MAXN; i += 512) s_acc[i] = 0.f;
    const bf16_t* xg = p.x + (size_t)g * p.Pin * p.K;
    bf16_t* y0 = p.y + (size_t)g * p.P * p.N;
    for (int i = tid; i < KP; i += 512) {                // (channels >= K: raw 0 -> act(1 * 0 + 0) = 0)
        s_vec[i] = (p.in_scale && i < p.K) ? p.in_scale[(size_t)g * p.in_gs + i] : 1.f;
        s_vec[KP + i] = (p.in_scale && i < p.K) ? p.in_shift[(size_t)g * p.in_gs + i] : 0.f;
    }
    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float alo = uniform(p.in_scale ? act_lo(p.act) : -INFINITY), ahi = uniform(p.in_scale ? act_hi(p.act) : INFINITY);
    char* stg = s_stage + wave * (TPX * SROW);
    auto in_row = [&](int px) -> size_t {
        if (!S2) return (size_t)px * p.K;
        const int q = p.OH * p.OW;
        const int n = px / q, rem = px - n * q;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        return ((size_t)(n * p.H + 2 * oh) * p.W + 2 * ow) * p.K;
    };
    union { s16x4_ h[2]; bf16x8 v; } ones;
    ones.h[0] = s16x4_{0x3F80, 0x3F80, 0x3F80, 0x3F80};
    ones.h[1] = ones.h[0];
    const int trow = 8 * lg + (li >> 2);
    // weight slab loads of this thread: chunk e = tid + 512 j of the slab's SLAB x KP x 2 contiguous bytes
    bf16x8 wreg[WLD];
    auto w_issue = [&](int slab) {
        const bf16_t* src = p.w + (size_t)slab * SLAB * p.K;
#pragma unroll
        for (int j = 0; j < WLD; ++j) wreg[j] = *reinterpret_cast<const bf16x8*>(src + (size_t)(tid + 512 * j) * 8);
    };
    auto w_write = [&](int buf) {
#pragma unroll
        for (int j = 0; j < WLD; ++j) {
            const int e = tid + 512 * j, row = e / (KP / 8), ch = e - row * (KP / 8);
            *reinterpret_cast<bf16x8*>(s_w + buf * (SLAB * WROW) + row * WROW + ((ch ^ (row & 15)) << 4)) = wreg[j];
        }
    };
    // ---- two wave groups in opposite phases.  With ONE barrier per slab step all 8 waves run the same phase at the same time -- weight
    // write, MFMAs, epilogue -- and the matrix pipe idles through every epilogue (measured 0.150 ms at the layer-3 shape, no better than the
    // slab form).  Here group A (waves 0-3) and group B (waves 4-7) run the SAME loop { barrier; MFMAs of slab s; barrier; epilogue of slab s }
    // with B one barrier behind (it passes one extra barrier first, A one extra at the end): while one group multiplies the other stages,
    // reduces, stores and moves weights.  Barrier events E0, E1, ..: A multiplies slab s in [E(2s+1), E(2s+2)) and runs its epilogue in
    // [E(2s+2), E(2s+3)); B multiplies in [E(2s+2), E(2s+3)), epilogue in [E(2s+3), E(2s+4)).  Weight slab j lives in buffer j & 1 from
    // E(2j+1) to E(2j+3); every wave writes its 4 KB share of slab j inside [E(2j-1), E(2j+1)): A in its epilogue of slab j - 1, B in its
    // epilogue of slab j - 2 (its idle first phase for j = 1) -- so every wave simply writes slabs 0, 1, 2, .. in order, one per epilogue,
    // and requests the next one right behind the write (a whole step of look-ahead).  Two buffers suffice.
    const int grp = wave >> 2;
    int wslab = 1;                                       // next slab this wave writes (slab j -> weights j % nslab, buffer j & 1)
    w_issue(0);
    w_write(0);
    w_issue(1 % nslab);
    __syncthreads();                                     // E0: s_vec / s_acc initialised, slab 0 visible
    bf16x8 bfr[KS][NPG];
    auto act_issue = [&](int ts) {
#pragma unroll
        for (int pg = 0; pg < NPG; ++pg) {
            int px = (ts * NW + wave) * TPX + pg * 16 + li;
            px = px < P ? px : P - 1;
            const bf16_t* row = xg + in_row(px) + lg * 8;
#pragma unroll
            for (int k = 0; k < KS; ++k) bfr[k][pg] = *reinterpret_cast<const bf16x8*>(row + k * 32);
        }
    };
    act_issue(wi < nset ? wi : nset - 1);
    auto w_advance = [&]() {                             // this wave's share of slab `wslab` -> its buffer; request the one after it
        w_write(wslab & 1);
        ++wslab;
        w_issue(wslab % nslab);
    };
    if (grp == 1) {
        __syncthreads();                                 // E1 (A starts slab 0)
        w_advance();                                     // B's idle first phase: slab 1
    }
    int step = 0;                                        // slab-step counter of this wave (buffer parity)
    int prev_sl = 0;
    // One slab step.  FULL (compile time): the wave holds a full 32-pixel tile -- its 4 stores are unconditional, so the wait in front of
    // the next ds_write of the weight registers COUNTS them (vmcnt(7..4)) instead of draining them (stores under a per-lane condition:
    // vmcnt(3..0) there, i.e. every step waited for the previous step's stores to reach memory).  LAST: the step after which the
    // activation registers are free -- the next tile set's rows are requested behind its MFMAs.
#ifdef WIDE_PROBE
    // phase probe (tools/wide_phase_probe.py builds this file alone with -DWIDE_PROBE): shader-clock ticks this wave spent in each phase of
    // its slab steps, summed over the launch; workgroup 0 writes them over the head of the statistics buffer at the end
    unsigned long long pr_t[7] = {0, 0, 0, 0, 0, 0, 0}, pr_last = __builtin_readcyclecounter();
#define PR_MARK(i) do { const unsigned long long pr_now = __builtin_readcyclecounter(); pr_t[i] += pr_now - pr_last; pr_last = pr_now; } while (0)
#else
#define PR_MARK(i) do {} while (0)
#endif
    auto slab_step = [&](int ts, int sl, int npx, bf16_t* yb, auto full_c, auto last_c) {
        constexpr bool FULL = decltype(full_c)::value, LAST = decltype(last_c)::value;
        const int buf = step & 1;
        PR_MARK(0);                                      // (everything between two slab steps: tile-set transform, loop control)
        __syncthreads();                                 // slab `sl` (buffer buf) is complete and visible
        PR_MARK(1);                                      // wait at the first barrier
        const char* wb = s_w + buf * (SLAB * WROW);
        constexpr int NOLD = TPX * CPR / 64;
        bf16x8 old[EPI == 2 ? NOLD : 1];
        bf16_t* ys = yb + sl * SLAB;
        if (EPI == 2) {                                  // rows of the tensor accumulated into: requested a whole MFMA phase ahead of their use
#pragma unroll
            for (int i = 0; i < NOLD; ++i) {
                const int e = lane + 64 * i;
                int px = e / CPR;
                const int ch = e - px * CPR;
                if (!FULL) px = px < npx ? px : (npx > 0 ? npx - 1 : 0);
                old[i] = *reinterpret_cast<const bf16x8*>(((FULL || npx > 0) ? ys : y0) + (size_t)px * p.N + ch * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 acc[NPG][NCT];
#pragma unroll
        for (int pg = 0; pg < NPG; ++pg)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[pg][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 fa[2][NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) fa[0][ct] = *reinterpret_cast<const bf16x8*>(wb + (ct * 16 + li) * WROW + ((lg ^ li) << 4));
        static_for_k<KS>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k + 1 < KS) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    fa[(k + 1) & 1][ct] = *reinterpret_cast<const bf16x8*>(wb + (ct * 16 + li) * WROW + ((((k + 1) * 4 + lg) ^ li) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int pg = 0; pg < NPG; ++pg)
                    acc[pg][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[k & 1][ct], bfr[k][pg], acc[pg][ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (LAST) {
            // the activation registers are free: request the next tile set's rows now, behind this step's epilogue
            const int nts = ts + wpg;
            act_issue(nts < nset ? nts : ts);
            __builtin_amdgcn_sched_barrier(0);
        }
        PR_MARK(2);                                      // MFMA phase (fragment reads, 64 MFMAs, next-set requests)
        __syncthreads();                                 // the other group starts its MFMAs of this slab / of the next one
        PR_MARK(3);                                      // wait at the second barrier
        // ---- epilogue of the slab step: stage, statistics, store, weights
#pragma unroll
        for (int pg = 0; pg < NPG; ++pg) {
            const bool live = FULL || pg * 16 + li < npx;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                bf16x4 v = f32_to_bf4(acc[pg][ct]);
                if (!live) v = bf16x4{0, 0, 0, 0};
                *reinterpret_cast<bf16x4*>(stg + (pg * 16 + li) * SROW + (ct * 16 + lg * 4) * 2) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (EPI == 0 && p.stats) {
            float* slot = s_slot + buf * (NW * SLOTW) + wave * SLOTW;
            // all fragment reads first, then all MFMAs, then UNCONDITIONAL slot writes (a lane without a valid element writes to its own sink
            // entry): under exec-masked branches every 16-channel block was its own chain of LDS read -> MFMA -> LDS write latencies
            // (phase probe: 1 900 - 2 200 of the 7 000 cycles of a slab step went to staging + statistics)
            union { s16x4_ h[2]; bf16x8 v; } fr[NCT];
#pragma unroll
            for (int cb = 0; cb < NCT; ++cb) {
                const char* fp = stg + trow * SROW + (cb * 16 + 4 * (li & 3)) * 2;
                fr[cb].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(fp));
                fr[cb].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(fp + 4 * SROW));
            }
            f32x4 dsum[NCT], dsq[NCT];
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cb = 0; cb < NCT; ++cb) {
                dsum[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, fr[cb].v, z4, 0, 0, 0);
                dsq[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[cb].v, fr[cb].v, z4, 0, 0, 0);
            }
            // D rows = lg * 4 + r, column = li: row 0 of dsum (lanes lg == 0) holds the column sums, the diagonal of dsq sits in the lanes
            // with li >> 2 == lg
            const int rr = li & 3;
            float* sink = slot + 2 * SLAB + lane;
#pragma unroll
            for (int cb = 0; cb < NCT; ++cb) {
                float* a1 = lg == 0 ? slot + cb * 16 + li : sink;
                float* a2 = (li >> 2) == lg ? slot + SLAB + cb * 16 + li : sink;
                *a1 = dsum[cb][0];
                *a2 = rr == 0 ? dsq[cb][0] : rr == 1 ? dsq[cb][1] : rr == 2 ? dsq[cb][2] : dsq[cb][3];
            }
            // fold the slots of the PREVIOUS step (all 8 waves wrote them before the barrier above; their buffer is rewritten one step
            // from now): group A, 32 entries per wave, waves in order -> the workgroup's accumulator (one owner thread per entry)
            if (grp == 0 && step > 0 && lane < 32) {
                const int ent = wave * 32 + lane;                            // 0 .. 127: [sum 64 | sum of squares 64] of slab prev_sl
                const float* sp = s_slot + (buf ^ 1) * (NW * SLOTW) + ent;
                float v = sp[0];
#pragma unroll
                for (int w = 1; w < NW; ++w) v += sp[w * SLOTW];
                float* a = s_acc + (ent < SLAB ? 0 : MAXN) + prev_sl * SLAB + (ent < SLAB ? ent : ent - SLAB);
                *a += v;
            }
        }
        PR_MARK(4);                                      // staging + statistics
        {   // (no run-time switch around these stores: one more history at the join and the waits drain them again)
#pragma unroll
            for (int i = 0; i < NOLD; ++i) {
                const int e = lane + 64 * i;
                const int px = e / CPR, ch = e - px * CPR;
                if (FULL || px < npx) {
                    union { struct { s16x4_ a, b; } s; bf16x8 v; } o;
                    o.s.a = *reinterpret_cast<const s16x4_*>(stg + px * SROW + ch * 16);
                    o.s.b = *reinterpret_cast<const s16x4_*>(stg + px * SROW + ch * 16 + 8);
                    if (EPI == 2) o.v = f32_to_bf8(bf8_to_f32(o.v) + bf8_to_f32(old[i]));
                    *reinterpret_cast<bf16x8*>(ys + (size_t)px * p.N + ch * 8) = o.v;
                }
            }
        }
        PR_MARK(5);                                      // read-back + stores
        w_advance();                                     // this wave's share of its next slab (A: slab step + 1, B: step + 2)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PR_MARK(6);                                      // weight write (waits for the slab requested one step ago) + next request
        prev_sl = sl;
        ++step;
    };
    auto tile_set = [&](int ts, auto full_c) {
        const int t = ts * NW + wave;                                        // this wave's 32-pixel tile (partial set: may lie past the group, npx = 0)
        const int npx = P - t * TPX < TPX ? (P - t * TPX > 0 ? P - t * TPX : 0) : TPX;
        bf16_t* yb = y0 + (size_t)t * TPX * p.N;
        if constexpr (LAZY) {                                                // transform the tile ONCE, in place
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const f32x8 sc = load_f32x8(s_vec + k * 32 + lg * 8), sh = load_f32x8(s_vec + KP + k * 32 + lg * 8);
#pragma unroll
                for (int pg = 0; pg < NPG; ++pg) {
                    f32x8 v = bf8_to_f32(bfr[k][pg]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = clamp_act(fmaf(v[i], sc[i], sh[i]), alo, ahi);
                    bfr[k][pg] = f32_to_bf8(v);
                }
            }
        }
        for (int sl = 0; sl + 1 < nslab; ++sl) slab_step(ts, sl, npx, yb, full_c, std::false_type{});
        slab_step(ts, nslab - 1, npx, yb, full_c, std::true_type{});
    };
    const int nfull = P / (NW * TPX);                                        // tile sets whose 8 tiles are all full
    int ts = wi;
    for (; ts < nfull; ts += wpg) tile_set(ts, std::true_type{});
    if (ts < nset) tile_set(ts, std::false_type{});                          // (at most one partial set per group, at index nfull)
    if (grp == 0) __syncthreads();                                           // A's extra barrier: pairs with B's last one
#ifdef WIDE_PROBE
    if (blockIdx.x == 0 && lane == 0 && p.stats) {
        for (int i = 0; i < 7; ++i) p.stats[wave * 8 + i] = (double)pr_t[i];
        p.stats[wave * 8 + 7] = (double)step;
        return;
    }
    if (blockIdx.x == 0) return;
#endif
    if (EPI == 0 && p.stats) {
        __syncthreads();
        if (step > 0 && tid < 2 * SLAB) {                                     // the last step's slots
            const int buf = (step - 1) & 1;
            const float* sp = s_slot + buf * (NW * SLOTW) + tid;
            float v = sp[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += sp[w * SLOTW];
            s_acc[(tid < SLAB ? 0 : MAXN) + prev_sl * SLAB + (tid < SLAB ? tid : tid - SLAB)] += v;
        }
        __syncthreads();
        for (int i = tid; i < 2 * p.N; i += 512)
            stat_publish(p.stats + (size_t)g * ADAMML_STAT_SLOTS * 2 * p.N + i, 2 * p.N, 0, s_acc[(i < p.N ? 0 : MAXN) + (i < p.N ? i : i - p.N)]);
    }
}

template <int KS, int EPI, bool LAZY, bool S2>
int wide_all_launch(WXP& p, int groups, hipStream_t stream) {
    constexpr int KP = KS * 32, SLAB = 64;
    constexpr size_t lds = (size_t)2 * SLAB * (KP * 2) + 2 * KP * 4 + 2 * 2048 * 4 + 2 * 8 * (2 * SLAB + 64) * 4 + (size_t)8 * 32 * (SLAB * 2 + 8);
    static_assert(lds <= 160 * 1024, "wide_all: LDS budget");
    static AdamLdsOnce attr_once;                    // (per device: common.h)
    const int attr_dev = adamml_current_device();
    if (!attr_once.test(attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wide_all_kernel<KS, EPI, LAZY, S2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "conv1x1 (wide): cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        attr_once.set(attr_dev);
    }
    p.nslab = p.N / SLAB;
    p.rpg = groups;
    const long ntile = (p.P + 31) / 32, nset = (ntile + 7) / 8;
    long wpg = 256 / groups;                                 // workgroups per BatchNorm group: one per CU over all groups
    if (wpg < 1) wpg = 1;
    if (wpg > nset) wpg = nset;
    hipLaunchKernelGGL((wide_all_kernel<KS, EPI, LAZY, S2>), dim3((unsigned)(groups * wpg)), dim3(512), lds, stream, p);
    return adamml_check_launch("conv1x1 (wide, all slabs)");
}


bool wide_on() {           // A/B aid: ADAMML_WIDE_STREAM=0 = conv_gemm_kernel.  Read at every call (no cached state: a test flips it within one process)
    const char* e = getenv("ADAMML_WIDE_STREAM");
    return !(e && e[0] == '0');
}

}  // namespace

// Shapes with an instance: 1x1, K = 256 input channels, at least twice as many output channels (multiple of 128, <= 2048), stride 1 -- or
// stride 2 in the forward direction (downsample branch) --, enough pixels to give the machine whole tile sets.  (K = 512 -- layer 4 -- does
// not fit: 128 registers of B fragments per wave, and 17 640 pixels are 69 sets of 256 for 256 CUs.)
bool adamml_conv1x1_wide_expand_supported(const adamml_conv_desc_t* d) {
    if (!wide_on() || d->KH != 1 || d->KW != 1 || d->pad != 0 || d->up > 1) return false;
    if (d->stride != 1 && !(d->stride == 2 && !d->accumulate && d->OH == (d->H - 1) / 2 + 1 && d->OW == (d->W - 1) / 2 + 1)) return false;
    if (d->Cin != 256 || d->Cout < 2 * d->Cin || d->Cout % 128 || d->Cout > 2048) return false;
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
    if (s2) return lazy ? wide_all_launch<8, 0, true, true>(p, groups, stream) : wide_all_launch<8, 0, false, true>(p, groups, stream);
    if (d->accumulate) return wide_all_launch<8, 2, false, false>(p, groups, stream);
    if (lazy) return wide_all_launch<8, 0, true, false>(p, groups, stream);
    return wide_all_launch<8, 0, false, false>(p, groups, stream);
}
