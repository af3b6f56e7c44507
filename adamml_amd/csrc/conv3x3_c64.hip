// 3x3 stride-1 pad-1 convolution with 64 input and 64 output channels (conv2 of the layer-1 bottlenecks,
// models/resnet.py:84-86 at 56x56, and -- with the flipped weight pack -- its data gradient), gfx950.
//
// The generic implicit-GEMM kernel re-gathers the activation tile once per tap: 9x the input through the 64 B/clk vector
// memory path of a CU, which bounds it at ~17 % of the MFMA peak.  Here a workgroup (8 waves, one per CU) keeps
//   * the whole weight matrix [64][9*64] bf16 (74 KB) resident in LDS for all its tiles, and
//   * the INPUT PATCH of a strip of R full output rows ((R+2) x (W+2) pixels x 64 ch, zero border, 144-byte pixel pitch)
//     in LDS, loaded once per tile with 16-byte coalesced loads that already apply the producer's BatchNorm+ReLU
//     (so the lazily normalised input is never materialised in HBM), the next tile's patch in flight in registers;
// the 18 K-steps (tap, 32-channel half) read both MFMA operands straight from LDS with aligned, conflict-free
// ds_read_b128 -- no per-tap staging, no gathers.  Output strips are contiguous in HBM (full rows), staged through the
// dead patch buffer and stored 16 B per lane; forward statistics come from the matrix cores, the data-gradient variant
// applies the activation mask and accumulates the BatchNorm-backward sums (conv_gemm.hip epilogue semantics).
#include <type_traits>
#include "common.h"
#include "../../include/adamml_hip.h"


// Phase timing (tools/c64_phase_probe.py builds this file alone with -DC64_PHASE_TIMING; never part of the shipped library): every wave of
// one workgroup stamps the shader clock at the phase boundaries of its first tiles.
#ifdef C64_PHASE_TIMING
// (stamps go to a small LDS area and are flushed to global memory once per tile: a global store per stamp would make every stamp wait for
// the prefetched patch loads -- vmcnt is shared)
__device__ unsigned* c64_dbg = nullptr;
#define C64_TS_ON (dbg_on && it < 16)
#define C64_TS(k) do { if (C64_TS_ON && lane == 0) s_dbg[wave * 16 + (k)] = (unsigned)__builtin_readcyclecounter(); } while (0)
#define C64_TS_DECL unsigned* s_dbg = reinterpret_cast<unsigned*>(smem + 160 * 1024 - 1024); unsigned* g_dbg = c64_dbg; const bool dbg_on = g_dbg && blockIdx.x == 8;
#define C64_TS_FLUSH do { if (C64_TS_ON) { __syncthreads(); if (tid < 128) g_dbg[it * 128 + tid] = s_dbg[tid]; __syncthreads(); } } while (0)
#define C64_LDS(x) (160 * 1024)
extern "C" int adamml_c64_set_phase_buffer(unsigned* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(c64_dbg), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#else
#define C64_TS(k) do { } while (0)
#define C64_TS_DECL
#define C64_TS_FLUSH do { } while (0)
#define C64_LDS(x) (x)
#endif

namespace {

constexpr int NT3 = 512;
constexpr int CS3_BYTES = 8 * 128 * 4;   // per-wave rows of the workgroup's channel sums (sum | second moment of 64 channels)
constexpr int C64 = 64;
constexpr int KT3 = 9 * C64;             // 576
constexpr int WROW3 = KT3 * 2 + 16;      // LDS bytes per weight row (+16 B skew): 1168
constexpr int PPIX = C64 * 2 + 16;       // patch pixel pitch: 144 B
// Start stagger.  Every workgroup runs the same phases on same-sized tiles, and the CUs stay in step: all of them request their next patch
// (36 MB chip-wide) within the same few microseconds, then all of them compute while HBM idles.  The first workgroup of each CU (the
// later ones inherit its phase) therefore starts 0..3 quarter-tile times late, by its position within the XCD: forward 0.95 -> 0.90 ms,
// weight gradient 0.89 -> 0.84 ms at the layer-1 shape (tools/c64_ab.py; two / eight / sixteen phases and longer delays measured the same).
// Only for launches whose workgroups own >= 8 tiles each: the delay (<= 18 000 cycles) must stay small against the workgroup's life.
#ifndef C64_STAGGER
#define C64_STAGGER 94                   // s_sleep units (64 cycles) per phase step; 0 = off (A/B aid)
#endif
#define C64_STAGGER_PH 4
#define C64_STAGGER_START do { if (C64_STAGGER && p.tpb >= 8 && blockIdx.x < 256) { const int ph = (blockIdx.x >> 3) & (C64_STAGGER_PH - 1); for (int i = 0; i < ph; ++i) __builtin_amdgcn_s_sleep(C64_STAGGER); } } while (0)
constexpr int SROW3 = C64 * 2 + 8;       // staging row: 136 B (144 measured the same)
constexpr int MAXPT = 4;                 // pixel tiles (16 px) per wave
constexpr int MAXPX3 = 8 * MAXPT * 16;   // 512 output pixels per tile at most
constexpr int MAXSLOT3 = 10;             // 16-byte patch slots per thread

struct C3P {
    const bf16_t* x;
    const bf16_t* w;         // [64][9][64] bf16 (pack mode 0 forward / mode 1 data gradient)
    const float* in_scale;
    const float* in_shift;
    bf16_t* y;
    double* stats;           // [G][SLOTS][128] or null
    const bf16_t* bn_z;      // data gradient fused with the BatchNorm backward of the tensor it flows into, or null
    const float* bn_vec;     // [G][4][64]
    int bn_act, act;
    int N, H, W, R, tiles_per_img, tiles_per_group, total_tiles, tpb;
    int PW, PR, npt;         // patch width / rows (pixels); pixel tiles per tile
    int in_gstride;
    int wswz;                // 1: weight rows stored with the chunk swizzle (A/B aid ADAMML_C64_WSWZ)
    size_t gxy;              // elements per group of x and y (same shape)
};

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;

// relu(scale * z + shift) of 8 bf16 values, rounded to bf16.  The generic form (unpack, fma, max, min, convert: 43 VALU instructions per
// 16-byte slot) made the patch staging the second-longest phase of a tile (tools/c64_phase_probe.py: 3500 of 24000 cycles with two waves
// per SIMD); here the FMAs are packed-fp32 and ReLU is one signed 16-bit max per PAIR on the rounded result -- round-to-nearest-even is
// monotonic and keeps the sign, so max(round(v), 0) == round(max(v, 0)) bit for bit (and -0 becomes +0 either way).
__device__ __forceinline__ bf16x8 bn_relu8(bf16x8 raw, const f32x8& sc, const f32x8& sh) {
    union { bf16x8 v; unsigned w[4]; } in;
    union { bf16x8 v; s16x2 h[4]; bf16x2 b[4]; } out;
    in.v = raw;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 z = {__uint_as_float(in.w[i] << 16), __uint_as_float(in.w[i] & 0xffff0000u)};
        const f32x2 a = {sc[2 * i], sc[2 * i + 1]}, b = {sh[2 * i], sh[2 * i + 1]};
        const f32x2 r = __builtin_elementwise_fma(z, a, b);
        out.b[i] = __builtin_convertvector(r, bf16x2);
        out.h[i] = __builtin_elementwise_max(out.h[i], s16x2{0, 0});
    }
    return out.v;
}

__device__ __forceinline__ void fold16_to_cs(const f32x8& esum, const f32x8& esq, float* cs, int lane, int ech) {
    // lanes l, l+8, .., l+56 of a wave hold partial sums of channel chunk `ech` (8 channels): DPP + lane-swap fold
    // (see conv_gemm.hip), then 4 LDS adds from the lanes with bit 3 clear into the row of THIS wave (cs = that row: one owner lane per
    // entry, so an entry's adds happen in tile order -- common.h, reproducible reductions)
    float v[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = esum[i]; v[8 + i] = esq[i]; }
#pragma unroll
    for (int i = 0; i < 16; ++i)
        v[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[i]), 0x128, 0xf, 0xf, false));
    float u[8], wv[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float a = v[2 * i], b = v[2 * i + 1];
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        u[i] = a + b;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = u[2 * j], b = u[2 * j + 1];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        wv[j] = a + b;
    }
    const int lrow = lane >> 4;
    const int vsel = ((lrow & 1) << 1) | (lrow >> 1);
    if (!(lane & 8)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int id = 4 * j + vsel;
            atomicAdd(&cs[(id >> 3) * C64 + ech * 8 + (id & 7)], wv[j]);
        }
    }
}

// BNZ: the data-gradient form (activation mask + BatchNorm-backward sums in the epilogue, p.bn_z != null) is its own instantiation, so
// that its epilogue -- which keeps a batch of z rows in flight -- does not weigh on the register allocation of the forward kernel.
// PP = patch pixel pitch in bytes.  144 (9 x 16 B) lets 8 output rows of a 56-wide image fit beside the weights, but an odd pitch in
// 16-byte units cannot be conflict-free for the gfx950 ds_read_b128 service groups (rows {0-3, 12-15} at chunk c with rows {4-11} at
// chunk c + 1: for ANY pixel order one of the two complementary groups collides -- the groups cover all 16 residues and one half is
// shifted by one).  160 (10 units) is conflict-free for every pixel offset; it costs one output row per strip (7 instead of 8 at W = 56).
template <bool BNZ, int PP>
__global__ __launch_bounds__(NT3, 1) void conv3x3_c64_kernel(C3P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;                                              // [64][WROW3]
    float* cs = reinterpret_cast<float*>(smem + C64 * WROW3);      // [8 waves][128]: every wave accumulates into its own row
    char* s_patch = smem + C64 * WROW3 + CS3_BYTES;                // patch, later the staging tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    C64_TS_DECL
    const int li = lane & 15, lg = lane >> 4;
    C64_STAGGER_START;

    for (int e = tid; e < C64 * (KT3 / 8); e += NT3) {
        const int co = e / (KT3 / 8), ch = e - co * (KT3 / 8);
        // 16-byte chunk ch of row co sits at position ch ^ wswz(co): with the 73-chunk row pitch the 16 lanes of a gfx950 ds_read_b128 service
        // group (rows {0-3, 12-15} at chunk c and rows {4-11} at chunk c + 1, or the complement) then touch 16 distinct 16-byte bank units;
        // unswizzled they touched 9 (2-way conflicts on every weight fragment read: SQ_LDS_BANK_CONFLICT 34-41 % of the LDS cycles, round 2)
        *reinterpret_cast<bf16x8*>(s_w + co * WROW3 + (ch ^ (((co >> 2) ^ (co >> 3)) & p.wswz)) * 16) = *reinterpret_cast<const bf16x8*>(p.w + (size_t)co * KT3 + ch * 8);
    }
    for (int i = tid; i < 8 * 128; i += NT3) cs[i] = 0.f;
    float* csw = cs + wave * 128;
    // publication of one BatchNorm group's sums: the eight wave rows folded in wave order, one exact add per channel and workgroup
    auto publish = [&](int grp) {
        if (tid < 128) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) { v += cs[w * 128 + tid]; cs[w * 128 + tid] = 0.f; }
            stat_publish(p.stats + (size_t)grp * ADAMML_STAT_SLOTS * 128 + tid, 128, blockIdx.x & (ADAMML_STAT_SLOTS - 1), v);
        }
    };

    // ---- patch slots: slot e -> (patch pixel e >> 3, 16-byte chunk e & 7 == tid & 7) -------------------------------------------
    // slot l of this thread is patch pixel (tid >> 3) + 64 l: its (row, column) is carried from slot to slot (the slots of a tile are always
    // loaded in order) instead of a 10-register table
    const int nslots = p.PR * p.PW * 8;
    const int ech = tid & 7;
    const int pstep_r = 64 / p.PW, pstep_c = 64 - pstep_r * p.PW;
    const int pr0 = (tid >> 3) / p.PW, pc0 = (tid >> 3) - pr0 * p.PW;
    int cpr = pr0, cpc = pc0;
    bf16x8 rp[MAXSLOT3];
    unsigned rok = 0;
    const char* nimg = nullptr;          // (uniform) image base / first input row of the tile being prefetched
    int nih0 = 0;
    auto load_begin = [&](int tile) {
        const int g = tile / p.tiles_per_group, tg = tile - g * p.tiles_per_group;
        const int n = tg / p.tiles_per_img, tr = tg - n * p.tiles_per_img;
        nih0 = tr * p.R - 1;
        nimg = reinterpret_cast<const char*>(p.x + (size_t)g * p.gxy + (size_t)n * p.H * p.W * C64);
        rok = 0;
        cpr = pr0; cpc = pc0;
        asm volatile("" : "+v"(cpr), "+v"(cpc));      // (opaque: the whole carry chain is tile-invariant and would be hoisted out of the tile loop as a 20-register table)
    };
    auto load_slot = [&](int l) {
        // UNCONDITIONAL loads from clamped (always valid) addresses, zeroed at the LDS write: a branch per slot makes
        // the compiler wait for each load before the next one is issued (one HBM round trip per slot)
        const int ih = nih0 + cpr, iw = cpc - 1;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W && tid + l * NT3 < nslots;
        rok |= (ok ? 1u : 0u) << l;
        const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
        // uniform 64-bit base + 32-bit lane offset (data gradient 1.13 -> 1.08 ms against per-lane 64-bit addresses, tools/c64_ab.py)
        const unsigned off = (unsigned)(ihc * p.W + iwc) * (C64 * 2) + ech * 16;
        rp[l] = *reinterpret_cast<const bf16x8*>(nimg + off);
        cpc += pstep_c;
        const bool wrap = cpc >= p.PW;
        cpc -= wrap ? p.PW : 0;
        cpr += pstep_r + (wrap ? 1 : 0);
    };
    auto load_patch = [&](int tile) {
        load_begin(tile);
#pragma unroll
        for (int l = 0; l < MAXSLOT3; ++l) load_slot(l);
    };
    const int tile0 = blockIdx.x * p.tpb;
    if (tile0 < p.total_tiles) load_patch(tile0);
    int cur_group = -1;

    for (int it = 0; it < p.tpb; ++it) {
        const int tile = tile0 + it;
        if (tile >= p.total_tiles) break;
        const int g = tile / p.tiles_per_group, tg = tile - g * p.tiles_per_group;
        const int n = tg / p.tiles_per_img, tr = tg - n * p.tiles_per_img;
        const int oh0 = tr * p.R;
        const int rows = min(p.R, p.H - oh0);
        const int npx = rows * p.W;
        if (g != cur_group) {
            if (p.stats && cur_group >= 0) {       // publish the finished group's sums, restart the accumulators
                __syncthreads();
                publish(cur_group);
                __syncthreads();
            }
            cur_group = g;
        }
        // ---- prefetched patch -> LDS (BatchNorm + activation of the producer applied here, once per element) ------------
        C64_TS(0);
        {
            const float lo = act_lo(p.act), hi = act_hi(p.act);
            f32x8 sc, sh;                          // (re-read per tile from L1/L2: not held across the MFMA loop)
#pragma unroll
            for (int i = 0; i < 8; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
            if (p.in_scale) {
                sc = load_f32x8(p.in_scale + (size_t)g * p.in_gstride + ech * 8);
                sh = load_f32x8(p.in_shift + (size_t)g * p.in_gstride + ech * 8);
            }
            const bool relu_bn = p.in_scale && p.act == ACT_RELU;
#pragma unroll
            for (int l = 0; l < MAXSLOT3; ++l) {
                const int e = tid + l * NT3;
                if (e < nslots) {
                    bf16x8 v = rp[l];
                    if (relu_bn) v = bn_relu8(v, sc, sh);                       // (uniform)
                    else if (p.in_scale) {
                        f32x8 f = bf8_to_f32(v);
#pragma unroll
                        for (int i = 0; i < 8; ++i) f[i] = clamp_act(fmaf(f[i], sc[i], sh[i]), lo, hi);
                        v = f32_to_bf8(f);
                    }
                    if (!((rok >> l) & 1u)) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    *reinterpret_cast<bf16x8*>(s_patch + (e >> 3) * PP + ech * 16) = v;
                }
            }
        }
        C64_TS(1);
        __syncthreads();
        C64_TS(2);
        if (it + 1 < p.tpb && tile + 1 < p.total_tiles) load_patch(tile + 1);
        // BNZ: the epilogue reads the z rows of this strip (one 128-byte line per pixel, contiguous) with nothing to overlap them with --
        // one 8-wave workgroup per CU, every register and all of the LDS taken.  One dword per line requested HERE, ahead of the ~14 us
        // of MFMA work, pulls the strip into L2: the epilogue's two batches of row loads then cost an L2 round trip instead of an HBM one.
        unsigned ztouch = 0;
        if constexpr (BNZ) {
            const size_t zb = (size_t)g * p.gxy + ((size_t)n * p.H + oh0) * p.W * C64;
            ztouch = *reinterpret_cast<const unsigned*>(p.bn_z + zb + (size_t)min(tid, npx - 1) * C64);
        }

        // ---- MFMA: wave w owns pixel tiles w, w+8, w+16, w+24 (16 consecutive output pixels, row-major over the strip) ---
        f32x4 acc[4][MAXPT];
        int pixoff[MAXPT];
#pragma unroll
        for (int j = 0; j < MAXPT; ++j) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            int q = (wave + 8 * j) * 16 + li;
            if (q >= npx) q = 0;
            const int r = q / p.W, c = q - r * p.W;
            pixoff[j] = (r * p.PW + c) * PP + lg * 16;
        }
        const char* wbase = s_w + li * WROW3 + (lg ^ (((li >> 2) ^ (li >> 3)) & p.wswz)) * 16;
        const bool last_live = (wave + 24) * 16 < npx;
        C64_TS(3);
        // Two copies of the loop, for waves with 4 and with 3 live pixel tiles (dead ones of a short strip read valid patch addresses and are
        // zeroed at staging; full strips have >= 24 live tiles): with the 4th tile behind a branch INSIDE the K step every K step was its own
        // basic block -- five ds_reads, a full LDS round trip, then MFMAs (tools/c64_phase_probe.py: 10 700 cycles for 8 064 of MFMA) --
        // straight-line K steps let the scheduler run the next step's reads under this step's MFMAs.
        auto mfma_loop = [&](auto npt_c) {
            constexpr int NPT = decltype(npt_c)::value;
#pragma unroll 1                              // (all 18 K steps unrolled: 0.86 -> 1.02 ms, measured)
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int kk = 0; kk < 6; ++kk) {             // (kw, 32-channel half)
                    const int ks = kh * 6 + kk;
                    const int aoff = (kh * p.PW + (kk >> 1)) * PP + (kk & 1) * 64;
                    bf16x8 fw[4], fa[NPT];
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) fw[ct] = *reinterpret_cast<const bf16x8*>(wbase + ct * 16 * WROW3 + ks * 64);
#pragma unroll
                    for (int j = 0; j < NPT; ++j) fa[j] = *reinterpret_cast<const bf16x8*>(s_patch + pixoff[j] + aoff);
#pragma unroll
                    for (int j = 0; j < NPT; ++j)
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
                            acc[ct][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ct], fa[j], acc[ct][j], 0, 0, 0);
                }
            }
        };
        if (__builtin_amdgcn_readfirstlane(last_live ? 1 : 0)) mfma_loop(std::integral_constant<int, 4>{});
        else mfma_loop(std::integral_constant<int, 3>{});
        if constexpr (BNZ) asm volatile("" ::"v"(ztouch));   // (keeps the touch load's destination allocated until it has landed)
        C64_TS(4);
        __syncthreads();                                     // patch consumed: its LDS becomes the staging tile
        C64_TS(5);

        // ---- stage [npt*16][64] bf16 (rows >= npx zero) ---------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < MAXPT; ++j) {
            const int pt = wave + 8 * j;
            if (pt < p.npt) {
                const int q = pt * 16 + li;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    bf16x4 v = f32_to_bf4(acc[ct][j]);
                    if (q >= npx) v = bf16x4{0, 0, 0, 0};
                    *reinterpret_cast<bf16x4*>(s_patch + q * SROW3 + (ct * 16 + lg * 4) * 2) = v;
                }
            }
        }
        C64_TS(6);
        __syncthreads();
        C64_TS(7);
        const size_t obase = (size_t)g * p.gxy + ((size_t)n * p.H + oh0) * p.W * C64;
        if constexpr (!BNZ) {
            for (int e = tid; e < npx * 8; e += NT3) {
                const int q = e >> 3;
                const s16x4 lo = *reinterpret_cast<const s16x4*>(s_patch + q * SROW3 + ech * 16);
                const s16x4 hi = *reinterpret_cast<const s16x4*>(s_patch + q * SROW3 + ech * 16 + 8);
                union { struct { s16x4 a, b; } s; bf16x8 v; } u;
                u.s.a = lo; u.s.b = hi;
                *reinterpret_cast<bf16x8*>(p.y + obase + (size_t)q * C64 + ech * 8) = u.v;
            }
            if (p.stats) {
                // waves w and w+4 share channel block w & 3 and split the 32-pixel steps; ones*F and diag(F^T F)
                union { s16x4 h[2]; bf16x8 v; } ones;
                ones.h[0] = s16x4{0x3F80, 0x3F80, 0x3F80, 0x3F80};
                ones.h[1] = ones.h[0];
                const int cb = wave & 3;
                const int trow = 8 * lg + (li >> 2);
                f32x4 dsum = {0.f, 0.f, 0.f, 0.f}, dsq = {0.f, 0.f, 0.f, 0.f};
                const int nps = (npx + 31) >> 5;
                for (int ps = wave >> 2; ps < nps; ps += 2) {
                    const char* fp = s_patch + (ps * 32 + trow) * SROW3 + (cb * 16 + 4 * (li & 3)) * 2;
                    union { s16x4 h[2]; bf16x8 v; } f;
                    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(fp));
                    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(fp + 4 * SROW3));
                    dsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, f.v, dsum, 0, 0, 0);
                    dsq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.v, f.v, dsq, 0, 0, 0);
                }
                if (lg == 0) atomicAdd(&csw[cb * 16 + li], dsum[0]);             // (own row: the only lane that ever adds to this entry)
                if ((li >> 2) == lg) {
                    const int r = li & 3;
                    const float q2 = r == 0 ? dsq[0] : r == 1 ? dsq[1] : r == 2 ? dsq[2] : dsq[3];
                    atomicAdd(&csw[64 + cb * 16 + li], q2);
                }
            }
        } else {
            // data gradient w.r.t. a lazily normalised tensor: g' = g * act'(scale*z+shift); sums of g' and g'*zhat
            const float* vec = p.bn_vec + (size_t)g * 4 * C64;
            const f32x8 bsc = load_f32x8(vec + ech * 8), bsh = load_f32x8(vec + C64 + ech * 8);
            const f32x8 mu = load_f32x8(vec + 2 * C64 + ech * 8), is = load_f32x8(vec + 3 * C64 + ech * 8);
            f32x8 esum, esq;
#pragma unroll
            for (int i = 0; i < 8; ++i) esum[i] = esq[i] = 0.f;
            // the z rows of a batch of 4 are requested before the batch's first store (unconditional loads from clamped addresses):
            // loads and stores retire in order through vmcnt and the compiler cannot move a z load above a store to y, so the
            // row-at-a-time loop exposed one HBM round trip per row (7 per tile: the data gradient ran 1.33 ms against 0.89 ms forward)
            constexpr int MAXE = MAXPX3 * 8 / NT3, EBZ = 4;
            auto rows = [&](auto relu_c) {
#pragma unroll 1
            for (int b0 = 0; b0 < MAXE; b0 += EBZ) {                // (not unrolled: eight rows of address arithmetic hoisted at once spill)
                if ((b0 * NT3) >= npx * 8) break;                  // (uniform)
                bf16x8 zr[EBZ];
#pragma unroll
                for (int k = 0; k < EBZ; ++k) {
                    const int q = min((tid + (b0 + k) * NT3) >> 3, npx - 1);
                    zr[k] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p.bn_z + obase + (size_t)q * C64 + ech * 8));
                }
#pragma unroll
                for (int k = 0; k < EBZ; ++k) {
                    const int e = tid + (b0 + k) * NT3;
                    if (e < npx * 8) {
                        const int q = e >> 3;
                        const s16x4 lo = *reinterpret_cast<const s16x4*>(s_patch + q * SROW3 + ech * 16);
                        const s16x4 hi = *reinterpret_cast<const s16x4*>(s_patch + q * SROW3 + ech * 16 + 8);
                        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
                        u.s.a = lo; u.s.b = hi;
                        f32x8 f = bf8_to_f32(u.v);
                        const f32x8 zv = bf8_to_f32(zr[k]);
                        // (the VALU work of this epilogue is 2.8 of the kernel's 6.3 VALU instructions per MFMA, profiles/r02_pmc_mfma_*:
                        // ReLU as one compare + select instead of the generic two-sided mask and its multiply; the rounded value is
                        // unpacked from the packed words the store uses -- the vector conversion re-converted every element)
                        if constexpr (decltype(relu_c)::value) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) f[i] = fmaf(zv[i], bsc[i], bsh[i]) > 0.f ? f[i] : 0.f;
                        } else {
#pragma unroll
                            for (int i = 0; i < 8; ++i) f[i] *= act_mask(fmaf(zv[i], bsc[i], bsh[i]), p.bn_act);
                        }
                        union { bf16x8 v; unsigned w[4]; } o;
                        o.v = f32_to_bf8(f);
                        *reinterpret_cast<bf16x8*>(p.y + obase + (size_t)q * C64 + ech * 8) = o.v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(o.w[i] << 16); f[2 * i + 1] = __uint_as_float(o.w[i] & 0xffff0000u); }
                        esum += f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) esq[i] += f[i] * (zv[i] - mu[i]) * is[i];
                    }
                }
            }
            };
            if (p.bn_act == ADAMML_ACT_RELU) rows(std::true_type{}); else rows(std::false_type{});      // (uniform)
            if (p.stats) fold16_to_cs(esum, esq, csw, lane, ech);
        }
        C64_TS(8);
        __syncthreads();                                     // staging consumed before the next patch lands
        C64_TS(9);
        C64_TS_FLUSH;
    }
    if (p.stats && cur_group >= 0) {
        __syncthreads();
        publish(cur_group);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same conv: dW[co][tap][ci] = sum_p dz[p][co] * a[p @ tap][ci], pixels as the MFMA reduction
// dimension.  Same strips and the same (transformed, zero-bordered) LDS input patch as the forward; the dz strip is
// staged next to it.  Both operands are pixel-major, so the fragments are hardware transpose reads with lane-supplied
// addresses: A from the dz strip, B = 16 consecutive input channels of the patch pixel shifted by the tap.  36 N tiles
// (9 taps x 4 channel blocks) are split over the 8 waves; accumulators live in registers over all strips of a workgroup,
// one partial [64][9][64] fp32 per workgroup goes to the workspace (tap-major, summed and permuted by wgrad_reduce_kernel).
struct C3WP {
    const bf16_t* x;
    const bf16_t* dz;
    const float* in_scale;
    const float* in_shift;
    float* ws;
    int act, N, H, W, R, tiles_per_img, tiles_per_group, total_tiles, tpb, PW, PR, in_gstride;
    int cp;                  // channels of x and dz (pixel pitch in elements): 64, or 128 = four 64 x 64 quadrants (blockIdx.y)
    size_t gxy;
};

constexpr int MAXDZ3 = 8;                // 16-byte dz slots per thread (MAXPX3 px x 8 chunks / 512)

__global__ __launch_bounds__(NT3, 1) void conv3x3_c64_wgrad_kernel(C3WP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_patch = smem;
    char* s_dz = smem + p.PR * p.PW * PPIX;                          // [MAXPX3][SROW3]
    int* s_poff = reinterpret_cast<int*>(s_dz + MAXPX3 * SROW3);     // patch byte offset of each strip pixel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    C64_TS_DECL
    const int li = lane & 15, lg = lane >> 4;
    const int ech = tid & 7;
    const int nslots = p.PR * p.PW * 8;
    // 128 channels (conv2 of the layer-2 bottlenecks): workgroup (x, y) computes quadrant y = (dz half qo, x half qi) of dW for the strips of
    // workgroup x -- the same kernel on 64-channel slices of 256-byte pixels; the four partials of a block land in one [128][9 x 128] block
    const int nq = p.cp >> 6, qo = blockIdx.y / nq, qi = blockIdx.y - qo * nq;
    const int cpb = p.cp * 2;                                        // pixel pitch in bytes
    C64_STAGGER_START;
    for (int q = tid; q < MAXPX3; q += NT3) {
        const int qq = q < p.R * p.W ? q : 0;
        const int r = qq / p.W, c = qq - r * p.W;
        s_poff[q] = (r * p.PW + c) * PPIX;
    }
    // patch slot l of this thread is patch pixel (tid >> 3) + 64 l: its (row, column) is carried from slot to slot (the slots of a tile are
    // always loaded in order) instead of a 10-register table
    const int pstep_r = 64 / p.PW, pstep_c = 64 - pstep_r * p.PW;
    const int pr0 = (tid >> 3) / p.PW, pc0 = (tid >> 3) - pr0 * p.PW;
    int cpr = pr0, cpc = pc0;
    bf16x8 rp[MAXSLOT3], rz[MAXDZ3];
    unsigned rok = 0;                    // bits 0..9: patch slots valid; bits 16..23: dz slots valid
    // the tile being prefetched: uniform image / strip bases, first input row, live dz elements
    const char* nimg = nullptr;
    const char* nzb = nullptr;
    int nih0 = 0, nnpx8 = 0;
    auto load_begin = [&](int tile) {
        const int g = tile / p.tiles_per_group, tg = tile - g * p.tiles_per_group;
        const int n = tg / p.tiles_per_img, tr = tg - n * p.tiles_per_img;
        nih0 = tr * p.R - 1;
        const size_t ibase = (size_t)g * p.gxy + (size_t)n * p.H * p.W * p.cp;
        nimg = reinterpret_cast<const char*>(p.x + ibase + qi * C64);
        const int oh0 = tr * p.R;
        nnpx8 = min(p.R, p.H - oh0) * p.W * 8;
        nzb = reinterpret_cast<const char*>(p.dz + ibase + (size_t)oh0 * p.W * p.cp + qo * C64);
        rok = 0;
        cpr = pr0; cpc = pc0;
        asm volatile("" : "+v"(cpr), "+v"(cpc));      // (opaque: the whole carry chain is tile-invariant and would be hoisted out of the tile loop as a 20-register table)
    };
    auto load_p = [&](int l) {           // patch slot l: unconditional load from a clamped (valid) address, zeroed at the LDS write
        const int ih = nih0 + cpr, iw = cpc - 1;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W && tid + l * NT3 < nslots;
        rok |= (ok ? 1u : 0u) << l;
        const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
        const unsigned off = (unsigned)(ihc * p.W + iwc) * (unsigned)cpb + ech * 16;   // (uniform 64-bit base + 32-bit lane offset)
        rp[l] = *reinterpret_cast<const bf16x8*>(nimg + off);
        cpc += pstep_c;
        const bool wrap = cpc >= p.PW;
        cpc -= wrap ? p.PW : 0;
        cpr += pstep_r + (wrap ? 1 : 0);
    };
    auto load_z = [&](int l) {
        const int e = tid + l * NT3;
        const unsigned ec = (unsigned)(e < nnpx8 ? e : 0);
        rz[l] = *reinterpret_cast<const bf16x8*>(nzb + (ec >> 3) * (unsigned)cpb + (ec & 7u) * 16u);
        rok |= (e < nnpx8 ? 1u : 0u) << (16 + l);
    };
    auto load_tile = [&](int tile) {
        load_begin(tile);
#pragma unroll
        for (int l = 0; l < MAXSLOT3; ++l) load_p(l);
#pragma unroll
        for (int l = 0; l < MAXDZ3; ++l) load_z(l);
    };
    f32x4 acc[4][5];                      // [cout tile][own N tile j]: N tile nt = wave + 8*j -> (tap nt >> 2, channel block nt & 3)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int noff[5];                          // patch byte offset of each own N tile: tap shift + channel block + this lane's 8-byte chunk
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int nt = min(wave + 8 * j, 35), t = nt >> 2, kh = t / 3, kw = t - kh * 3;
        noff[j] = (kh * p.PW + kw) * PPIX + ((nt & 3) * 16 + 4 * (li & 3)) * 2;
    }
    const int tile0 = blockIdx.x * p.tpb;
    if (tile0 < p.total_tiles) load_tile(tile0);
    const int trow = 8 * lg + (li >> 2);

    for (int it = 0; it < p.tpb; ++it) {
        const int tile = tile0 + it;
        if (tile >= p.total_tiles) break;
        const int g = tile / p.tiles_per_group, tg = tile - g * p.tiles_per_group;
        const int tr = tg % p.tiles_per_img;
        const int npx = min(p.R, p.H - tr * p.R) * p.W;
        C64_TS(0);
        {
            const float lo = act_lo(p.act), hi = act_hi(p.act);
            f32x8 sc, sh;
#pragma unroll
            for (int i = 0; i < 8; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
            if (p.in_scale) {
                sc = load_f32x8(p.in_scale + (size_t)g * p.in_gstride + qi * C64 + ech * 8);
                sh = load_f32x8(p.in_shift + (size_t)g * p.in_gstride + qi * C64 + ech * 8);
            }
            const bool relu_bn = p.in_scale && p.act == ACT_RELU;
#pragma unroll
            for (int l = 0; l < MAXSLOT3; ++l) {
                const int e = tid + l * NT3;
                if (e < nslots) {
                    bf16x8 v = rp[l];
                    if (relu_bn) v = bn_relu8(v, sc, sh);                       // (uniform)
                    else if (p.in_scale) {
                        f32x8 f = bf8_to_f32(v);
#pragma unroll
                        for (int i = 0; i < 8; ++i) f[i] = clamp_act(fmaf(f[i], sc[i], sh[i]), lo, hi);
                        v = f32_to_bf8(f);
                    }
                    if (!((rok >> l) & 1u)) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    *reinterpret_cast<bf16x8*>(s_patch + (e >> 3) * PPIX + ech * 16) = v;
                }
            }
#pragma unroll
            for (int l = 0; l < MAXDZ3; ++l) {
                const int e = tid + l * NT3;
                union { struct { s16x4 a, b; } s; bf16x8 v; } u;
                u.v = (rok >> (16 + l)) & 1u ? rz[l] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                char* dst = s_dz + (e >> 3) * SROW3 + ech * 16;
                *reinterpret_cast<s16x4*>(dst) = u.s.a;
                *reinterpret_cast<s16x4*>(dst + 8) = u.s.b;
            }
        }
        C64_TS(1);
        __syncthreads();
        C64_TS(2);
        // (issuing these 18 loads per thread a few per K step INSIDE the MFMA loop was measured and lost: 0.85 -> 1.00 ms, the loads
        // that start late are not back when the next tile is staged -- tools/c64_ab.py)
        if (it + 1 < p.tpb && tile + 1 < p.total_tiles) load_tile(tile + 1);
        C64_TS(3);
        const int nks = (npx + 31) >> 5;
        for (int ks = 0; ks < nks; ++ks) {
            const int q0 = ks * 32 + trow;
            const char* pb0 = s_patch + s_poff[q0];
            const char* pb1 = s_patch + s_poff[q0 + 4];
            const char* zb0 = s_dz + q0 * SROW3 + 4 * (li & 3) * 2;
            bf16x8 fa[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                union { s16x4 h[2]; bf16x8 v; } f;
                f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(zb0 + mt * 32));
                f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(zb0 + mt * 32 + 4 * SROW3));
                fa[mt] = f.v;
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                if (wave + 8 * j < 36) {                         // wave-uniform (only j == 4 can fail)
                    union { s16x4 h[2]; bf16x8 v; } f;
                    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb0 + noff[j]));
                    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb1 + noff[j]));
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], f.v, acc[mt][j], 0, 0, 0);
                }
            }
        }
        C64_TS(4);
        __syncthreads();
        C64_TS(5);
        C64_TS_FLUSH;
    }
    // partial [cp][9][cp] of this block (tap-major columns); quadrant (qo, qi) of it
    const int ktot = 9 * p.cp;
    float* out = p.ws + (size_t)blockIdx.x * p.cp * ktot + (size_t)qo * C64 * ktot + qi * C64;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int nt = wave + 8 * j;
            if (nt < 36) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(mt * 16 + lg * 4 + r) * ktot + (nt >> 2) * p.cp + (nt & 3) * 16 + li] = acc[mt][j][r];
            }
        }
}

}  // namespace

// d: the FORWARD-shaped descriptor of the conv that is executed (for a data gradient: H/W of dz == H/W of dx).
static int c3_pitch() {
    static const int pp = getenv("ADAMML_C64_PITCH") ? atoi(getenv("ADAMML_C64_PITCH")) : 144;      // A/B aid: 144 | 160
    return pp == 160 ? 160 : 144;
}

// rows per strip of the forward / data-gradient kernel: the largest R <= 512 / W (preferring, among the top four, one that divides H)
// whose patch fits beside the resident weights at pixel pitch `pitch`; 0 when none does
static int c3_rows(const adamml_conv_desc_t* d, int pitch) {
    int R = MAXPX3 / d->W;
    if (R > d->H) R = d->H;
    for (; R >= 2; --R) {
        int pick = R;
        for (int r = R; r >= R - 3 && r >= 2; --r)
            if (d->H % r == 0) { pick = r; break; }
        const int PR = pick + 2, PW = d->W + 2;
        const size_t patch = (size_t)PR * PW * pitch, stage = (size_t)((pick * d->W + 31) / 32 * 32) * SROW3;
        if (PR * PW * 8 <= MAXSLOT3 * NT3 && (pick * d->W + 31) / 32 * 2 <= 8 * MAXPT &&
            C64 * WROW3 + CS3_BYTES + (patch > stage ? patch : stage) <= 160 * 1024)
            return pick;
    }
    return 0;
}

// d: the FORWARD-shaped descriptor of the conv that is executed (for a data gradient: H/W of dz == H/W of dx).
bool adamml_conv3x3_c64_supported(const adamml_conv_desc_t* d) {
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->Cin != 64 || d->Cout != 64 || (d->up > 1)) return false;
    if (d->OH != d->H || d->OW != d->W || d->accumulate) return false;
    if (d->W < 8 || d->W > MAXPX3 / 2) return false;        // at least 2 rows per tile
    return c3_rows(d, 144) >= 2 && c3_rows(d, c3_pitch()) >= 2;
}

int adamml_conv3x3_c64_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale,
                              const float* in_shift, void* y, double* stats, const void* bn_z, const float* bn_vec, int bn_act,
                              hipStream_t stream) {
    C3P p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w_packed; p.in_scale = in_scale; p.in_shift = in_shift; p.y = (bf16_t*)y;
    p.stats = stats; p.bn_z = (const bf16_t*)bn_z; p.bn_vec = bn_vec; p.bn_act = bn_act; p.act = d->act;
    p.N = d->N; p.H = d->H; p.W = d->W;
    const int pitch = c3_pitch();
    const int R = c3_rows(d, pitch);
    static const int wswz = getenv("ADAMML_C64_WSWZ") ? atoi(getenv("ADAMML_C64_WSWZ")) & 1 : 1;
    p.wswz = wswz;
    p.R = R; p.PR = R + 2; p.PW = d->W + 2;
    p.npt = (R * d->W + 31) / 32 * 2;            // staged rows cover whole 32-pixel statistic steps
    const int groups = d->groups < 1 ? 1 : d->groups;
    p.tiles_per_img = ceil_div(d->H, R);
    p.tiles_per_group = d->N * p.tiles_per_img;
    p.total_tiles = groups * p.tiles_per_group;
    if (p.total_tiles <= 0) return ADAMML_OK;
    p.in_gstride = d->in_gstride;
    p.gxy = (size_t)d->N * d->H * d->W * C64;
    p.tpb = ceil_div(p.total_tiles, 1024);
    const size_t patch = (size_t)p.PR * p.PW * pitch, stage = (size_t)p.npt * 16 * SROW3;
    const size_t lds = C64 * WROW3 + CS3_BYTES + (patch > stage ? patch : stage);
    static AdamLdsOnce attr_once;                    // (per device: common.h)
    const int attr_dev = adamml_current_device();
    if (!attr_once.test(attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<false, 144>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<true, 144>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<false, 160>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<true, 160>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "conv3x3_c64: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        attr_once.set(attr_dev);
    }
    const dim3 grid(ceil_div(p.total_tiles, p.tpb));
    if (pitch == 160) {
        if (p.bn_z) hipLaunchKernelGGL((conv3x3_c64_kernel<true, 160>), grid, dim3(NT3), C64_LDS(lds), stream, p);
        else hipLaunchKernelGGL((conv3x3_c64_kernel<false, 160>), grid, dim3(NT3), C64_LDS(lds), stream, p);
    } else {
        if (p.bn_z) hipLaunchKernelGGL((conv3x3_c64_kernel<true, 144>), grid, dim3(NT3), C64_LDS(lds), stream, p);
        else hipLaunchKernelGGL((conv3x3_c64_kernel<false, 144>), grid, dim3(NT3), C64_LDS(lds), stream, p);
    }
    return adamml_check_launch("conv3x3_c64");
}

static int c3_geometry(const adamml_conv_desc_t* d, int* R_out) {
    int R = MAXPX3 / d->W;
    if (R > d->H) R = d->H;
    for (int r = R; r >= R - 3 && r >= 2; --r)
        if (d->H % r == 0) { R = r; break; }
    *R_out = R;
    return ceil_div(d->H, R);
}

// (128 channels: the four-quadrant form; ADAMML_C64_WGRAD_Q=0 disables it, read at every call: A/B aid)
static bool c3_wgrad_quad(const adamml_conv_desc_t* d, int cin_true) {
    const char* e = getenv("ADAMML_C64_WGRAD_Q");
    if (e && atoi(e) == 0) return false;
    // (256 channels = sixteen quadrants at the layer-3 shape, 14 x 14 images of one 196-pixel tile each, measured 0.275 ms against the generic
    // kernel's 0.256: not served)
    if (cin_true != 128 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->Cin != 128 || d->Cout != 128 || d->up > 1) return false;
    if (d->OH != d->H || d->OW != d->W || d->W < 8 || d->W > MAXPX3 / 2) return false;
    return (long)(d->groups < 1 ? 1 : d->groups) * d->N * d->H * d->W >= 65536;      // (a partial [C][9][C] per workgroup: only where the tensors dwarf them)
}

bool adamml_conv3x3_c64_wgrad_supported(const adamml_conv_desc_t* d, int cin_true) {
    if (!c3_wgrad_quad(d, cin_true) && (cin_true != C64 || !adamml_conv3x3_c64_supported(d))) return false;
    int R;
    c3_geometry(d, &R);
    const size_t lds = (size_t)(R + 2) * (d->W + 2) * PPIX + (size_t)MAXPX3 * SROW3 + MAXPX3 * sizeof(int);
    return lds <= 160 * 1024;
}

int adamml_conv3x3_c64_wgrad_blocks(const adamml_conv_desc_t* d, int* tpb_out) {
    int R;
    const long total = (long)(d->groups < 1 ? 1 : d->groups) * d->N * c3_geometry(d, &R);
    const int nwg = 256 / ((d->Cin >> 6) * (d->Cin >> 6));  // one workgroup per CU (128 / 256 channels: x 4 / 16 quadrants)
    int tpb = (int)((total + nwg - 1) / nwg);
    if (tpb < 1) tpb = 1;
    if (tpb_out) *tpb_out = tpb;
    return (int)((total + tpb - 1) / tpb);
}

int adamml_conv3x3_c64_wgrad_launch(const adamml_conv_desc_t* d, const void* dz, const void* x, const float* in_scale,
                                    const float* in_shift, float* ws, hipStream_t stream) {
    C3WP p;
    p.x = (const bf16_t*)x; p.dz = (const bf16_t*)dz; p.in_scale = in_scale; p.in_shift = in_shift; p.ws = ws; p.act = d->act;
    p.N = d->N; p.H = d->H; p.W = d->W;
    p.tiles_per_img = c3_geometry(d, &p.R);
    p.PR = p.R + 2; p.PW = d->W + 2;
    const int groups = d->groups < 1 ? 1 : d->groups;
    p.tiles_per_group = d->N * p.tiles_per_img;
    p.total_tiles = groups * p.tiles_per_group;
    p.in_gstride = d->in_gstride;
    p.cp = d->Cin;
    p.gxy = (size_t)d->N * d->H * d->W * p.cp;
    const int nblk = adamml_conv3x3_c64_wgrad_blocks(d, &p.tpb);
    const size_t lds = (size_t)p.PR * p.PW * PPIX + (size_t)MAXPX3 * SROW3 + MAXPX3 * sizeof(int);
    static AdamLdsOnce attr_once;                    // (per device: common.h)
    const int attr_dev = adamml_current_device();
    if (!attr_once.test(attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "conv3x3_c64 wgrad: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        attr_once.set(attr_dev);
    }
    hipLaunchKernelGGL(conv3x3_c64_wgrad_kernel, dim3(nblk, (p.cp >> 6) * (p.cp >> 6)), dim3(NT3), C64_LDS(lds), stream, p);
    return adamml_check_launch("conv3x3_c64 wgrad");
}

#ifdef C64_PHASE_TIMING
extern "C" int c64_probe_launch(const adamml_conv_desc_t* d, const void* x, const void* w, const float* sc, const float* sh, void* y, double* stats,
                                const void* bn_z, const float* bn_vec, int bn_act, hipStream_t s) {
    return adamml_conv3x3_c64_launch(d, x, w, sc, sh, y, stats, bn_z, bn_vec, bn_act, s);
}
extern "C" int c64_probe_wgrad_blocks(const adamml_conv_desc_t* d, int* tpb) { return adamml_conv3x3_c64_wgrad_blocks(d, tpb); }
extern "C" int c64_probe_wgrad_launch(const adamml_conv_desc_t* d, const void* dz, const void* x, const float* sc, const float* sh, float* ws, hipStream_t s) {
    return adamml_conv3x3_c64_wgrad_launch(d, dz, x, sc, sh, ws, s);
}
#endif
