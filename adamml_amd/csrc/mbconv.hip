// Forward of the first two thirds of a MobileNetV2 inverted-residual block -- 1x1 expansion -> BatchNorm -> ReLU6 -> depthwise 3x3 --
// WITHOUT the 6x-expanded tensor in HBM (models/policy_net.py:72-80, models/sound_mobilenet_v2.py:52-58).
//
// Unfused, the expansion writes E1 = [pixels][6 Cin] (bf16) and the depthwise conv reads it back: for the stride-2 blocks at the top of
// a MobileNetV2 that tensor is 4x the block's own output and the two launches are the most expensive of the whole net
// (profiles/r04_launch_table_policy_*.txt: 16 -> 96 at 80^2 x 1440 frames: 0.59 + 0.50 ms for 4.3 GB).  Here a workgroup owns an 8 x 8
// tile of depthwise OUTPUT pixels, stages the narrow input patch it needs ((7 S + 3)^2 pixels x Cin, halo included) in LDS once, and
// walks the expanded channels in chunks of 32: expansion of the patch on the matrix cores (v_mfma_f32_16x16x32_bf16, weights = A so a
// lane owns 4 consecutive channels of a pixel), rounding to bf16 (the value the unfused path stores), BatchNorm + ReLU6 in fp32 into an
// LDS tile, depthwise 3x3 on the VALU from that tile (tap order and fma chain of dwconv_fwd_kernel), raw depthwise output + its
// per-channel statistics out.  Same rounding points as the unfused pair: the outputs are the same numbers.
//
// Train-mode BatchNorm needs the batch statistics of E1 BEFORE that kernel can normalise it: PASS 0 of the same kernel computes the
// expansion only and publishes sum / sum of squares of the rounded E1 (every input pixel owned by exactly one tile), PASS 1 is the kernel
// above.  HBM traffic of the pair: two reads of the narrow input + one write of the depthwise output; E1 never exists.  The depthwise
// stage is VALU work (~20 instructions per expanded element) on a tile WITH halo (1.13x at stride 2, 1.56x at stride 1), so this pays
// where E1 is large against the output.  Forward only: a backward pass needs E1 (BatchNorm backward, ReLU6
// mask, depthwise weight gradient), so trainable nets keep the unfused pair; the frozen policy nets of the main-net stage
// (train_adamml.py:344-345) and every inference call use this one.
#include "common.h"
#include "../../include/adamml_hip.h"

ADAMML_DET_SETTER(mbconv)

namespace {

constexpr int NT = 256;
constexpr int TO = 8;          // output tile edge
constexpr int CK = 32;         // expanded channels per chunk
constexpr int MAXKS = 5;       // K steps of 32 input channels (Cin <= 160)

struct MbP {
    const bf16_t* x;           // [groups*N, H, W, Cin] narrow block input (raw; value = xs * x + xh when xs != null)
    const float* xs;
    const float* xh;
    int x_gs;
    const bf16_t* w1;          // [Cexp][Cin] bf16: forward pack of the expansion
    const float* b1s;          // BatchNorm of the expansion: scale, shift [Cexp] per group (b1_gs apart)   (PASS 1)
    const float* b1h;
    int b1_gs;
    const float* wd;           // [9][Cexp] fp32 depthwise pack                                              (PASS 1)
    bf16_t* y;                 // [groups*N, OH, OW, Cexp] raw depthwise output                              (PASS 1)
    double* stats;             // [groups][SLOTS][2 Cexp]: PASS 0: of the expansion output; PASS 1: of the depthwise output (may be null)
    int N, H, W, Cin, KP, Cexp, OH, OW, tiles_x, tiles_per_img, total_tiles, tpb, act;
    int resident;              // the block's parameters (expansion weights, BatchNorm vectors, depthwise taps, input transform) live in LDS
    size_t gx, gy;
};

template <int S, int PASS>
__global__ __launch_bounds__(NT) void mbconv_kernel(MbP p) {
    constexpr int PW = (TO - 1) * S + 3;       // patch edge: 10 / 17
    constexpr int NPX = PW * PW;
    constexpr int NPT = (NPX + 15) / 16;       // 16-pixel MFMA tiles of the patch
    constexpr int PXP = NPT * 16;
    constexpr int E1W = PASS ? 36 : 20;        // expansion tile row pitch in 4-byte words: fp32 [32] + 4 pad / bf16 [32] + 8 pad
    constexpr int NROW = PASS ? 4 : 8;         // rows of per-channel partial sums (one per wave / per pixel group)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int XP = p.KP + 8;                   // patch row pitch in bf16 elements (skews the banks; 16-byte multiple)
    const int CP = (p.Cexp + CK - 1) / CK * CK;
    bf16_t* Xs = reinterpret_cast<bf16_t*>(smem);
    char* E1 = smem + (size_t)PXP * XP * 2;
    float* st = reinterpret_cast<float*>(E1 + (size_t)PXP * E1W * 4);      // [NROW][2 CP]
    // resident parameters (p.resident): read from global inside the chunk loop they were three exposed L2 round trips per chunk -- with two
    // workgroups per CU the first version of this kernel spent 90 % of a tile waiting for them (block 2 of the policy net: 17 us per tile
    // for 1.5 us of instructions)
    bf16_t* W1s = reinterpret_cast<bf16_t*>(st + (size_t)NROW * 2 * CP);   // [CP][XP] expansion weights, zero rows / columns beyond Cexp / Cin
    float* Pv = reinterpret_cast<float*>(W1s + (size_t)CP * XP);           // [11][CP]: bn1 scale, shift, 9 depthwise taps
    float* Xv = Pv + (size_t)11 * CP;                                     // [2][KP]: lazy transform of the input

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    {
        const int g = blockIdx.y;
        p.x += (size_t)g * p.gx;
        if (p.xs) { p.xs += (size_t)g * p.x_gs; p.xh += (size_t)g * p.x_gs; }
        if (PASS) { p.y += (size_t)g * p.gy; p.b1s += (size_t)g * p.b1_gs; p.b1h += (size_t)g * p.b1_gs; }
        if (p.stats) p.stats += (size_t)g * ADAMML_STAT_SLOTS * 2 * p.Cexp;
    }
    for (int i = tid; i < NROW * 2 * CP; i += NT) st[i] = 0.f;
    const int nk = p.KP / 32, kch = p.KP / 8;
    if (p.resident) {
        for (int e = tid; e < CP * kch; e += NT) {
            const int row = e / kch, k = (e - row * kch) * 8;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (row < p.Cexp && k < p.Cin) v = *reinterpret_cast<const bf16x8*>(p.w1 + (size_t)row * p.Cin + k);
            *reinterpret_cast<bf16x8*>(W1s + (size_t)row * XP + k) = v;
        }
        if (PASS) {
            for (int i = tid; i < 11 * CP; i += NT) {
                const int r = i / CP, c = i - r * CP;
                Pv[i] = c < p.Cexp ? (r == 0 ? p.b1s[c] : r == 1 ? p.b1h[c] : p.wd[(size_t)(r - 2) * p.Cexp + c]) : 0.f;
            }
        }
        for (int i = tid; i < p.KP; i += NT) {
            Xv[i] = (p.xs && i < p.Cin) ? p.xs[i] : 1.f;
            Xv[p.KP + i] = (p.xs && i < p.Cin) ? p.xh[i] : 0.f;
        }
    }
    const float lo = act_lo(p.act), hi = act_hi(p.act);

    // The narrow input patch of the NEXT tile is requested (registers) before the chunk loop of the current one and stored to LDS after it:
    // one HBM round trip per tile otherwise, with nothing to overlap it.  XR 16-byte slots per thread cover PXP * kch <= XR * NT (the large,
    // early blocks: Cin <= 32 at stride 2, <= 128 at stride 1); wider inputs load in place.
    constexpr int XR = 8;
    const bool pre = PXP * kch <= XR * NT;
    bf16x8 xr[XR];
    unsigned xok = 0;
    auto tile_origin = [&](int tile, int& n, int& oy0, int& ox0) {
        n = tile / p.tiles_per_img;
        const int tt = tile - n * p.tiles_per_img;
        const int ty = tt / p.tiles_x;
        oy0 = ty * TO;
        ox0 = (tt - ty * p.tiles_x) * TO;
    };
    auto fetch = [&](int tile) {
        int n, oy0, ox0;
        tile_origin(tile, n, oy0, ox0);
        const bf16_t* img = p.x + (size_t)n * p.H * p.W * p.Cin;
        xok = 0;
#pragma unroll
        for (int j = 0; j < XR; ++j) {
            const int e = tid + j * NT;
            const int px = e / kch, k = (e - px * kch) * 8;
            const int pr = px / PW, pc = px - pr * PW;
            const int iy = oy0 * S - 1 + pr, ix = ox0 * S - 1 + pc;
            const bool ok = e < PXP * kch && px < NPX && k < p.Cin && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            xok |= (ok ? 1u : 0u) << j;
            // unconditional load from a clamped (always valid) address: a branch per slot would serialise the round trips
            const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1), kc = min(k, p.Cin - 8);
            xr[j] = *reinterpret_cast<const bf16x8*>(img + ((size_t)iyc * p.W + ixc) * p.Cin + kc);
        }
    };
    auto stash = [&]() {                                        // registers -> LDS, lazy transform of the producer applied once
#pragma unroll
        for (int j = 0; j < XR; ++j) {
            const int e = tid + j * NT;
            if (e < PXP * kch) {
                const int px = e / kch, k = (e - px * kch) * 8;
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if ((xok >> j) & 1u) {
                    v = xr[j];
                    if (p.xs) {
                        f32x8 sc, sh;
                        if (p.resident) { sc = load_f32x8(Xv + k); sh = load_f32x8(Xv + p.KP + k); }
                        else { sc = load_f32x8(p.xs + k); sh = load_f32x8(p.xh + k); }
                        f32x8 f = bf8_to_f32(v);
#pragma unroll
                        for (int i = 0; i < 8; ++i) f[i] = fmaf(f[i], sc[i], sh[i]);
                        v = f32_to_bf8(f);
                    }
                }
                *reinterpret_cast<bf16x8*>(Xs + (size_t)px * XP + k) = v;
            }
        }
    };
    const int tile_first = blockIdx.x * p.tpb;
    if (pre && tile_first < p.total_tiles) fetch(tile_first);

    for (int it = 0; it < p.tpb; ++it) {
        const int tile = tile_first + it;
        if (tile >= p.total_tiles) break;                       // (uniform)
        int n, oy0, ox0;
        tile_origin(tile, n, oy0, ox0);
        const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
        __syncthreads();                                        // previous tile's chunk loop is done with Xs (and st / the parameters are in place)
        if (pre) {
            stash();
            if (it + 1 < p.tpb && tile + 1 < p.total_tiles) fetch(tile + 1);
        } else {
        // ---- narrow input patch -> LDS (lazy BatchNorm of the producer applied once; zeros outside the image and in the K padding)
        const bf16_t* img = p.x + (size_t)n * p.H * p.W * p.Cin;
        for (int e = tid; e < PXP * kch; e += NT) {
            const int px = e / kch, k = (e - px * kch) * 8;
            const int pr = px / PW, pc = px - pr * PW;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = px < NPX && k < p.Cin && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) {
                v = *reinterpret_cast<const bf16x8*>(img + ((size_t)iy * p.W + ix) * p.Cin + k);
                if (p.xs) {
                    const f32x8 sc = load_f32x8(p.xs + k), sh = load_f32x8(p.xh + k);
                    f32x8 f = bf8_to_f32(v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] = fmaf(f[i], sc[i], sh[i]);
                    v = f32_to_bf8(f);
                }
            }
            *reinterpret_cast<bf16x8*>(Xs + (size_t)px * XP + k) = v;
        }
        }
        __syncthreads();

        for (int c0 = 0; c0 < CP; c0 += CK) {
            // ---- expansion of the patch for channels c0 .. c0 + 31 on the matrix cores
            bf16x8 fw[2][MAXKS];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int row = c0 + ct * 16 + li;
#pragma unroll
                for (int ks = 0; ks < MAXKS; ++ks) {
                    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                    const int k = ks * 32 + lg * 8;
                    if (ks < nk) {
                        if (p.resident) v = *reinterpret_cast<const bf16x8*>(W1s + (size_t)row * XP + k);
                        else if (row < p.Cexp && k < p.Cin) v = *reinterpret_cast<const bf16x8*>(p.w1 + (size_t)row * p.Cin + k);
                    }
                    fw[ct][ks] = v;
                }
            }
            f32x4 bsc[2], bsh[2];
            if (PASS) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int ch = c0 + ct * 16 + lg * 4;
                    const bool ok = ch < p.Cexp;
                    if (p.resident) {
                        bsc[ct] = *reinterpret_cast<const f32x4*>(Pv + ch);
                        bsh[ct] = *reinterpret_cast<const f32x4*>(Pv + CP + ch);
                    } else {
                        bsc[ct] = ok ? *reinterpret_cast<const f32x4*>(p.b1s + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
                        bsh[ct] = ok ? *reinterpret_cast<const f32x4*>(p.b1h + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            for (int pt = wave; pt < NPT; pt += 4) {
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                const bf16_t* xrow = Xs + (size_t)(pt * 16 + li) * XP + lg * 8;
#pragma unroll
                for (int ks = 0; ks < MAXKS; ++ks) {
                    if (ks < nk) {
                        const bf16x8 fx = *reinterpret_cast<const bf16x8*>(xrow + ks * 32);
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0][ks], fx, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1][ks], fx, acc[1], 0, 0, 0);
                    }
                }
                const int px = pt * 16 + li;
                const int pr = px / PW, pc = px - pr * PW;
                const bool inimg = px < NPX && (unsigned)(iy0 + pr) < (unsigned)p.H && (unsigned)(ix0 + pc) < (unsigned)p.W;
                if (PASS) {
                    // the depthwise conv zero-pads the ACTIVATED expansion: pixels outside the image are 0, not act(shift)
                    float* dst = reinterpret_cast<float*>(E1) + (size_t)px * E1W + lg * 4;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        f32x4 v = bf4_to_f32(f32_to_bf4(acc[ct]));                     // E1 as the unfused path stores it
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = inimg ? clamp_act(fmaf(v[i], bsc[ct][i], bsh[ct][i]), lo, hi) : 0.f;
                        *reinterpret_cast<f32x4*>(dst + ct * 16) = v;
                    }
                } else {
                    // statistics pass: every input pixel is counted by the ONE tile whose interior holds it
                    const bool own = inimg && pr >= 1 && pr < 1 + TO * S && pc >= 1 && pc < 1 + TO * S;
                    bf16_t* dst = reinterpret_cast<bf16_t*>(E1) + (size_t)px * (E1W * 2) + lg * 4;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        *reinterpret_cast<bf16x4*>(dst + ct * 16) = own ? f32_to_bf4(acc[ct]) : bf16x4{0, 0, 0, 0};
                }
            }
            __syncthreads();
            if (PASS) {
                // ---- depthwise 3x3 from the LDS tile: thread = 4 channels x 2 output pixels
                const int ch4 = tid & 7, ps = tid >> 3;
                const int chb = c0 + ch4 * 4;
                const bool chok = chb < p.Cexp;
                f32x4 wt[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (p.resident) wt[t] = *reinterpret_cast<const f32x4*>(Pv + (size_t)(t + 2) * CP + chb);
                    else wt[t] = chok ? *reinterpret_cast<const f32x4*>(p.wd + (size_t)t * p.Cexp + chb) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
                const float* e1 = reinterpret_cast<const float*>(E1) + ch4 * 4;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int opx = ps + 32 * j;
                    const int oy = opx >> 3, ox = opx & 7;
                    const bool valid = chok && oy0 + oy < p.OH && ox0 + ox < p.OW;
                    const float* b = e1 + (size_t)(oy * S * PW + ox * S) * E1W;
                    auto E = [&](int r, int c) { return *reinterpret_cast<const f32x4*>(b + (size_t)(r * PW + c) * E1W); };
                    f32x4 acc = E(0, 0) * wt[0];
                    acc += E(0, 1) * wt[1];
                    acc += E(0, 2) * wt[2];
                    acc += E(1, 0) * wt[3];
                    acc += E(1, 1) * wt[4];
                    acc += E(1, 2) * wt[5];
                    acc += E(2, 0) * wt[6];
                    acc += E(2, 1) * wt[7];
                    acc += E(2, 2) * wt[8];
                    const bf16x4 ob = f32_to_bf4(acc);
                    if (valid) {
                        *reinterpret_cast<bf16x4*>(p.y + (((size_t)n * p.OH + oy0 + oy) * p.OW + ox0 + ox) * p.Cexp + chb) = ob;
                        const f32x4 rv = bf4_to_f32(ob);
                        s += rv;
                        q += rv * rv;
                    }
                }
                if (p.stats) {
                    // the eight lanes of a wave that share ch4 fold their sums (fixed butterfly), lane group 0 owns the wave's LDS row
#pragma unroll
                    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
                        for (int i = 0; i < 4; ++i) { s[i] += __shfl_xor(s[i], o, 64); q[i] += __shfl_xor(q[i], o, 64); }
                    if (lane < 8 && chok) {
                        float* row = st + (size_t)wave * 2 * CP;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { row[chb + i] += s[i]; row[CP + chb + i] += q[i]; }
                    }
                }
            } else {
                // ---- statistics of the rounded expansion: thread = one channel x every 8th patch pixel, in pixel order
                const int c = tid & 31, grp = tid >> 5;
                const bf16_t* e1 = reinterpret_cast<const bf16_t*>(E1) + c;
                float s = 0.f, q = 0.f;
                for (int px = grp; px < PXP; px += 8) {
                    const float v = __uint_as_float((unsigned)e1[(size_t)px * (E1W * 2)] << 16);
                    s += v;
                    q = fmaf(v, v, q);
                }
                st[(size_t)grp * 2 * CP + c0 + c] += s;
                st[(size_t)grp * 2 * CP + CP + c0 + c] += q;
            }
            __syncthreads();                                    // the expansion tile is rewritten by the next chunk
        }
    }
    if (p.stats) {
        // rows folded in row order, one exact add per channel and workgroup (common.h: reproducible reductions)
        __syncthreads();
        const unsigned slot = blockIdx.x & (ADAMML_STAT_SLOTS - 1);
        for (int i = tid; i < 2 * p.Cexp; i += NT) {
            const int idx = i < p.Cexp ? i : CP + (i - p.Cexp);
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < NROW; ++r) v += st[(size_t)r * 2 * CP + idx];
            stat_publish(p.stats + i, 2 * (size_t)p.Cexp, slot, v);
        }
    }
}

template <int S, int PASS>
size_t mb_lds(int KP, int Cexp, bool resident) {
    constexpr int PW = (TO - 1) * S + 3, PXP = (PW * PW + 15) / 16 * 16;
    const int CP = (Cexp + CK - 1) / CK * CK;
    size_t b = (size_t)PXP * (KP + 8) * 2 + (size_t)PXP * (PASS ? 36 : 20) * 4 + (size_t)(PASS ? 4 : 8) * 2 * CP * 4;
    if (resident) b += (size_t)CP * (KP + 8) * 2 + (size_t)11 * CP * 4 + (size_t)2 * KP * 4;
    return b;
}

// parameters resident in LDS while the whole workgroup stays within half a CU's LDS (two workgroups per CU: the latency of one's
// barriers and LDS round trips is the other's work)
template <int S, int PASS>
bool mb_resident(int KP, int Cexp) { return mb_lds<S, PASS>(KP, Cexp, true) <= 80 * 1024; }

template <int S, int PASS>
int mb_launch_t(const MbP& p, int groups, unsigned nwg, hipStream_t stream, const char* what) {
    const size_t lds = mb_lds<S, PASS>(p.KP, p.Cexp, p.resident != 0);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mbconv_kernel<S, PASS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
        attr = true;
    }
    hipLaunchKernelGGL((mbconv_kernel<S, PASS>), dim3(nwg, groups), dim3(NT), lds, stream, p);
    return adamml_check_launch(what);
}

int mb_launch(const adamml_conv_desc_t* d, const void* x, int cin, const float* xs, const float* xh, int x_gs, const void* w1,
              const float* b1s, const float* b1h, int b1_gs, const float* wd, void* y, double* stats, int pass, hipStream_t stream,
              const char* what) {
    if (!adamml_mbconv_supported(d, cin)) return adamml_set_error(ADAMML_EUNSUPPORTED, "%s: unsupported block shape", what);
    if (!x || !w1 || (pass && (!b1s || !b1h || !wd || !y)) || (!pass && !stats)) return adamml_set_error(ADAMML_EINVAL, "%s: null argument", what);
    const int groups = d->groups < 1 ? 1 : d->groups;
    MbP p;
    p.x = (const bf16_t*)x; p.xs = xs; p.xh = xs ? xh : nullptr; p.x_gs = x_gs;
    p.w1 = (const bf16_t*)w1; p.b1s = b1s; p.b1h = b1h; p.b1_gs = b1_gs; p.wd = wd; p.y = (bf16_t*)y; p.stats = stats;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = cin; p.KP = (cin + 31) / 32 * 32; p.Cexp = d->Cout; p.OH = d->OH; p.OW = d->OW;
    p.act = d->act;
    p.tiles_x = ceil_div(d->OW, TO);
    p.tiles_per_img = p.tiles_x * ceil_div(d->OH, TO);
    const long total = (long)d->N * p.tiles_per_img;
    if (total <= 0) return ADAMML_OK;
    if (total >= (1L << 31)) return adamml_set_error(ADAMML_EUNSUPPORTED, "%s: too many tiles", what);
    p.total_tiles = (int)total;
    p.tpb = (int)((total * groups + 2047) / 2048);                 // ~8 workgroups per CU over all groups, statistics published once each
    if (p.tpb < 1) p.tpb = 1;
    const unsigned nwg = (unsigned)((total + p.tpb - 1) / p.tpb);
    p.gx = (size_t)d->N * d->H * d->W * cin;
    p.gy = (size_t)d->N * d->OH * d->OW * d->Cout;
    if (d->stride == 1) {
        p.resident = pass ? mb_resident<1, 1>(p.KP, p.Cexp) : mb_resident<1, 0>(p.KP, p.Cexp);
        return pass ? mb_launch_t<1, 1>(p, groups, nwg, stream, what) : mb_launch_t<1, 0>(p, groups, nwg, stream, what);
    }
    p.resident = pass ? mb_resident<2, 1>(p.KP, p.Cexp) : mb_resident<2, 0>(p.KP, p.Cexp);
    return pass ? mb_launch_t<2, 1>(p, groups, nwg, stream, what) : mb_launch_t<2, 0>(p, groups, nwg, stream, what);
}

}  // namespace

extern "C" int adamml_mbconv_supported(const adamml_conv_desc_t* d, int cin) {
    if (!d || d->KH != 3 || d->KW != 3 || d->pad != 1 || (d->stride != 1 && d->stride != 2) || d->Cin != d->Cout) return 0;
    if (cin < 8 || cin % 8 || cin > 32 * MAXKS || d->Cout % 16) return 0;
    if (d->OH != (d->H - 1) / d->stride + 1 || d->OW != (d->W - 1) / d->stride + 1) return 0;
    const int KP = (cin + 31) / 32 * 32;
    const size_t a = d->stride == 1 ? mb_lds<1, 1>(KP, d->Cout, false) : mb_lds<2, 1>(KP, d->Cout, false);
    const size_t b = d->stride == 1 ? mb_lds<1, 0>(KP, d->Cout, false) : mb_lds<2, 0>(KP, d->Cout, false);
    const size_t lds = a > b ? a : b;
    return lds <= 160 * 1024 ? 1 : 0;
}

extern "C" int adamml_mbconv_expand_stats(const adamml_conv_desc_t* d, const void* x, int cin, const float* x_scale, const float* x_shift,
                                          int x_gstride, const void* w_expand_packed, double* stats, hipStream_t stream) {
    return mb_launch(d, x, cin, x_scale, x_shift, x_gstride, w_expand_packed, nullptr, nullptr, 0, nullptr, nullptr, stats, 0, stream,
                     "mbconv_expand_stats");
}

extern "C" int adamml_mbconv_expand_dw(const adamml_conv_desc_t* d, const void* x, int cin, const float* x_scale, const float* x_shift,
                                       int x_gstride, const void* w_expand_packed, const float* bn1_scale, const float* bn1_shift,
                                       int bn1_gstride, const float* w_dw_tapmajor, void* y, double* stats, hipStream_t stream) {
    return mb_launch(d, x, cin, x_scale, x_shift, x_gstride, w_expand_packed, bn1_scale, bn1_shift, bn1_gstride, w_dw_tapmajor, y, stats, 1,
                     stream, "mbconv_expand_dw");
}
