// Backward of a 3x3 depthwise conv + train-mode BatchNorm + ReLU6 of an inverted residual (models/sound_mobilenet_v2.py:58-61,
// models/policy_net.py:66-69,80-83) in ONE pass: BatchNorm-backward apply, data gradient (with the mask and the BatchNorm-backward sums of
// the expansion it feeds) and weight gradient.
//
// The per-layer form moved the widest tensor of the block eight times: bn_bwd_apply (read g', z; write dz), dwconv_bwd_weight (read dz, x),
// dwconv_bwd_data_bn (read dz, x; write dx).  Here a thread of the column-strip walker (dwconv_gemm32.hip) owns 4 channels x 4 adjacent
// pixels, walks down the rows of a strip with a 3-row register window of dz = A g' + B z + C (formed from the two raw rows as they
// arrive, rounded to bf16 like the tensor the per-layer kernels exchange, halo columns included) and, per row,
//   * dx = sum_taps dz (.) w   (the forward walk over dz with reversed taps: same tap order as dwconv_fwd_kernel<1, BNZ>), masked by the
//     expansion's ReLU6 (from the raw expansion row x, which is read ONCE), stored, and summed into sum(dx'), sum(dx' xhat);
//   * dw[kh][kw] += a(q) dz(q - (kh-1, kw-1)) with a = the activated expansion value of the thread's own pixels (no halo on a).
// g', z, x are read once and dx written once: four passes instead of eight.  Stride 2 (S = 2): the thread owns a 2 x 4 block of input
// pixels per step and the 2 x 3 dz pixels that reach it.
#include "common.h"
#include "../../include/adamml_hip.h"

namespace {

constexpr int NT = 256;

struct DwBP {
    const bf16_t* g;        // [G][N,OH,OW,C] gradient w.r.t. the activated depthwise output, ALREADY masked by its activation
    const bf16_t* z;        // [G][N,OH,OW,C] raw depthwise output
    const float* aff;       // [G][3][C]: dz = A g + B z + C (adamml_bn_bwd_affine)
    const float* w;         // [9][C]
    const bf16_t* x;        // [G][N,H,W,C] raw expansion output (the depthwise conv's lazily normalised input)
    const float* xvec;      // [G][4][C] its BatchNorm vectors: scale, shift, mean, invstd
    bf16_t* dx;             // [G][N,H,W,C] gradient w.r.t. the activated expansion output, masked
    double* stats;          // [G][2C] deterministic accumulators: sum(dx'), sum(dx' xhat)
    float* ws;              // [G][gridDim.x][C][9] partial weight gradients
    int N, H, W, C, OH, OW, xact;
    int rows_per_thread, nseg, nrb;
    size_t gz, gx;          // group strides (elements) of g / z and of x / dx
};

// Epilogue shared by the two walkers: the threads that share a channel chunk (thread ids congruent modulo nchunk) add in turn -- a fixed
// order (common.h: reproducible reductions) --, the weight-gradient tile first (one [C][9] partial per workgroup), then, in the same LDS,
// the two statistics rows (published exactly).
__device__ __forceinline__ void fold_publish(const DwBP& p, float* dsm, const f32x4 (&accw)[9], const f32x4& s, const f32x4& q, bool any, int c) {
    const int C = p.C, nchunk = C >> 2;
    __syncthreads();                        // (the constants are dead: their LDS becomes the fold area)
    for (int i = threadIdx.x; i < 9 * C; i += NT) dsm[i] = 0.f;
    __syncthreads();
    const int nturn = (NT + nchunk - 1) / nchunk;
    for (int r = 0; r < nturn; ++r) {
        if (any && (int)threadIdx.x / nchunk == r) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) dsm[t * C + c + i] += accw[t][i];
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 9 * C; i += NT) {
        const int t = i / C, cc2 = i - t * C;
        p.ws[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 9 * C + (size_t)cc2 * 9 + t] = dsm[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += NT) dsm[i] = 0.f;
    __syncthreads();
    for (int r = 0; r < nturn; ++r) {
        if (any && (int)threadIdx.x / nchunk == r) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { dsm[c + i] += s[i]; dsm[C + c + i] += q[i]; }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 2 * C; i += NT) {
        const float v = dsm[i];
        if (v != 0.f) stat_publish(p.stats + i, 2 * (size_t)C, blockIdx.x & (ADAMML_STAT_SLOTS - 1), v);
    }
}

// Group offsets, then the per-channel constants into LDS: [16][C] = the nine taps (REVERSED order when flip), A / B / C, the expansion's
// scale / shift / mean / invstd.
__device__ __forceinline__ void group_and_constants(DwBP& p, float* dsm, bool flip) {
    const size_t g = blockIdx.y;
    p.g += g * p.gz; p.z += g * p.gz; p.x += g * p.gx; p.dx += g * p.gx;
    p.aff += g * 3 * p.C; p.xvec += g * 4 * p.C;
    p.stats += g * ADAMML_STAT_SLOTS * 2 * p.C;
    const int C = p.C;
    for (int i = threadIdx.x; i < 16 * C; i += NT) {
        const int r = i / C, cc2 = i - r * C;
        dsm[i] = r < 9 ? p.w[(size_t)(flip ? 8 - r : r) * C + cc2] : r < 12 ? p.aff[(size_t)(r - 9) * C + cc2] : p.xvec[(size_t)(r - 12) * C + cc2];
    }
    __syncthreads();
}

// Per-channel constants live in LDS ([16][C] fp32: the nine taps reversed, A / B / C, the expansion's scale / shift / mean / invstd) and
// are re-read where they are used -- 16 ds_read_b128 per row step against ~800 VALU lane-operations -- instead of pinning 64 registers;
// the dz window is kept as packed bf16 (it IS bf16: the value the per-layer kernels exchange) and unpacked one window row at a time.
template <int S, int SEGW>                  // SEGW = pixels (columns) per thread
__global__ __launch_bounds__(NT, 2) void dwconv_bwd_fused_kernel(DwBP p) {
    static_assert(S == 1, "stride-1 walker");
    constexpr int NCOL = SEGW + 2;          // dz columns feeding them
    extern __shared__ __attribute__((aligned(16))) float dsm[];          // [16][C] constants during the walk; then [9][C] weight-gradient fold, [2][C] statistics fold
    group_and_constants(p, dsm, true);
    const int C = p.C;
    const int nchunk = C >> 2;
    const int gid = blockIdx.x * NT + threadIdx.x;
    const int nthreads = gridDim.x * NT;
    const int chunk = gid % nchunk;
    const long ntasks = (long)nchunk * p.nseg * p.nrb * p.N;
    const int c = chunk * 4;
    // (read through an opaque copy of the base so that the loads stay where they are used: hoisted out of the walk as loop invariants
    // they were 64 pinned registers again -- 160 spilled)
    auto opaque = [](int b) { asm volatile("" : "+v"(b)); return b; };
    auto K = [&](int b, int r) { return *reinterpret_cast<const f32x4*>(dsm + b + r * C); };
    f32x4 accw[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) accw[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    bool any = false;
    const float lo = act_lo(p.xact), hi = act_hi(p.xact);
    for (long task = gid; task < ntasks; task += nthreads) {
        int tsk = (int)(task / nchunk);
        const int seg = tsk % p.nseg;
        tsk /= p.nseg;
        const int rb = tsk % p.nrb, n = tsk / p.nrb;
        any = true;
        const int ow_b = seg * SEGW;
        const int oh_b = rb * p.rows_per_thread;
        const int oh_e = min(p.H, oh_b + p.rows_per_thread);
        bool cok[NCOL];
        unsigned col[NCOL];                  // element offsets of the six columns (an invalid one reads column 0: never used)
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
            cok[j] = (unsigned)(ow_b - 1 + j) < (unsigned)p.W;
            col[j] = cok[j] ? (unsigned)(ow_b - 1 + j) * C * 2u : 0u;
        }
        // 32-bit element offsets from the group's (uniform) base pointers: one register per address instead of two
        const unsigned ioff = ((unsigned)n * p.H * p.W * C + c) * 2u;          // (bytes)
        const unsigned rstride = (unsigned)p.W * C * 2u;
        const char* gb = reinterpret_cast<const char*>(p.g);
        const char* zb = reinterpret_cast<const char*>(p.z);
        const char* xb = reinterpret_cast<const char*>(p.x);
        char* db = reinterpret_cast<char*>(p.dx);

        struct Raw { bf16x4 g[NCOL], z[NCOL]; };
        struct XRow { bf16x4 v[SEGW]; };
        struct WRow { bf16x4 v[NCOL]; };
        // unconditional loads from clamped addresses (a conditional request keeps the previous row's registers live across the step)
        auto load_dz = [&](int r, Raw& raw) {
            const unsigned ro = ioff + (unsigned)min(max(r, 0), p.H - 1) * rstride;
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                raw.g[j] = *reinterpret_cast<const bf16x4*>(gb + (ro + col[j]));
                raw.z[j] = *reinterpret_cast<const bf16x4*>(zb + (ro + col[j]));
            }
        };
        auto load_x = [&](int r, XRow& xr) {
            const unsigned ro = ioff + (unsigned)min(r, p.H - 1) * rstride;
#pragma unroll
            for (int o = 0; o < SEGW; ++o) xr.v[o] = *reinterpret_cast<const bf16x4*>(xb + (ro + col[o + 1]));
        };
        auto mkdz = [&](int r, const Raw& raw, WRow& dst) {
            const bool rok = (unsigned)r < (unsigned)p.H;
            const int kb = opaque(c);
            const f32x4 ca = K(kb, 9), cb = K(kb, 10), cc = K(kb, 11);
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                const f32x4 gv = bf4_to_f32(raw.g[j]), zv = bf4_to_f32(raw.z[j]);
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = fmaf(ca[i], gv[i], fmaf(cb[i], zv[i], cc[i]));
                const bf16x4 ob = f32_to_bf4(o);
                dst.v[j] = (rok && cok[j]) ? ob : bf16x4{0, 0, 0, 0};
            }
        };
        auto emit = [&](int r, const WRow& w0, const WRow& w1, const WRow& w2, const XRow& xr) {
            f32x4 acc[SEGW], a[SEGW];
            {
                const int kb = opaque(c);
                const f32x4 bsc = K(kb, 12), bsh = K(kb, 13);
#pragma unroll
                for (int o = 0; o < SEGW; ++o) {
                    const f32x4 xv = bf4_to_f32(xr.v[o]);
                    const bool ok = cok[o + 1];
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[o][i] = ok ? clamp_act(fmaf(xv[i], bsc[i], bsh[i]), lo, hi) : 0.f;
                }
            }
            // window row i holds dz row r - 1 + i: taps kh = 2 - i (reversed-tap forward walk, ascending (i, j) like dwconv_fwd_kernel)
            auto row = [&](const WRow& wr, int i) {
                f32x4 d[NCOL];
#pragma unroll
                for (int j = 0; j < NCOL; ++j) {
                    bf16x4 t = wr.v[j];
                    asm volatile("" : "+v"(t));
                    d[j] = bf4_to_f32(t);
                }
                const int kb = opaque(c);
                const f32x4 t0 = K(kb, 3 * i), t1 = K(kb, 3 * i + 1), t2 = K(kb, 3 * i + 2);
#pragma unroll
                for (int o = 0; o < SEGW; ++o) {
                    if (i == 0) acc[o] = d[o] * t0; else acc[o] += d[o] * t0;
                    acc[o] += d[o + 1] * t1;
                    acc[o] += d[o + 2] * t2;
                    // dw[kh][kw] += a(q) dz(q - (kh - 1, kw - 1)): window row 2 - kh, column o + 2 - kw
                    accw[(2 - i) * 3 + 2] += a[o] * d[o];
                    accw[(2 - i) * 3 + 1] += a[o] * d[o + 1];
                    accw[(2 - i) * 3] += a[o] * d[o + 2];
                }
            };
            row(w0, 0);
            row(w1, 1);
            row(w2, 2);
            const int kb2 = opaque(c);
            const f32x4 bmu = K(kb2, 14), bis = K(kb2, 15);
            const unsigned ro = ioff + (unsigned)r * rstride;
#pragma unroll
            for (int o = 0; o < SEGW; ++o) {
                // (a pixel past the row end has a = 0: its mask, its stored value and its sums vanish; only the store is predicated)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[o][i] *= (a[o][i] > lo && a[o][i] < hi) ? 1.f : 0.f;          // a = clamp(t): inside <=> t inside
                const bf16x4 ob = f32_to_bf4(acc[o]);
                if (cok[o + 1]) *reinterpret_cast<bf16x4*>(db + (ro + col[o + 1])) = ob;
                const f32x4 rv = bf4_to_f32(ob);
                s += rv;
                q += rv * ((bf4_to_f32(xr.v[o]) - bmu) * bis);
            }
        };

        WRow win[3];
        Raw nxt;
        XRow xcur, xnxt;
        { Raw r; load_dz(oh_b - 1, r); mkdz(oh_b - 1, r, win[0]); }
        { Raw r; load_dz(oh_b, r); mkdz(oh_b, r, win[1]); }
        load_dz(oh_b + 1, nxt);
        load_x(oh_b, xnxt);
#define DWB_STEP(R, A, B, C2)                                   \
        {                                                       \
            mkdz((R) + 1, nxt, win[C2]);                        \
            xcur = xnxt;                                        \
            load_dz((R) + 2, nxt);                              \
            load_x((R) + 1, xnxt);                              \
            emit((R), win[A], win[B], win[C2], xcur);           \
        }
        for (int r = oh_b; r < oh_e; r += 3) {
            DWB_STEP(r, 0, 1, 2);
            if (r + 1 >= oh_e) break;
            DWB_STEP(r + 1, 1, 2, 0);
            if (r + 2 >= oh_e) break;
            DWB_STEP(r + 2, 2, 0, 1);
        }
#undef DWB_STEP
    }
    fold_publish(p, dsm, accw, s, q, any, c);
}

// Stride 2 (pad 1): a thread owns 4 channels x a strip QW 2x2 input quads wide and walks down the quad
// rows.  Quad (k, j) -- input rows 2k, 2k+1, columns 2j, 2j+1 -- exchanges with exactly the four dz pixels (k + a, j + b), a, b in {0, 1}
// (pixel (0,0) through one tap, (0,1) and (1,0) through two, (1,1) through four: dwconv_bwd_data_s2_kernel, same tap order), so a strip
// needs dz columns QW js .. QW js + QW and a TWO-row window of dz: row k + 1 of one step is row k of the next.  dz is a quarter of the block's
// pixels; the full-resolution tensors x (read once: mask, xhat and the weight gradient's operand) and dx (written once) are the traffic.
template <int QW>                           // QW = quads (column pairs) per thread
__global__ __launch_bounds__(NT, 2) void dwconv_bwd_fused_s2_kernel(DwBP p) {
    constexpr int NDZ = QW + 1;             // dz columns feeding them
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    group_and_constants(p, dsm, false);
    const int C = p.C;
    const int nchunk = C >> 2;
    const int gid = blockIdx.x * NT + threadIdx.x;
    const int nthreads = gridDim.x * NT;
    const int chunk = gid % nchunk;
    const long ntasks = (long)nchunk * p.nseg * p.nrb * p.N;
    const int c = chunk * 4;
    auto opaque = [](int b) { asm volatile("" : "+v"(b)); return b; };
    auto K = [&](int b, int r) { return *reinterpret_cast<const f32x4*>(dsm + b + r * C); };
    f32x4 accw[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) accw[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    bool any = false;
    const float lo = act_lo(p.xact), hi = act_hi(p.xact);
    const char* gb = reinterpret_cast<const char*>(p.g);
    const char* zb = reinterpret_cast<const char*>(p.z);
    const char* xb = reinterpret_cast<const char*>(p.x);
    char* db = reinterpret_cast<char*>(p.dx);
    for (long task = gid; task < ntasks; task += nthreads) {
        int tsk = (int)(task / nchunk);
        const int seg = tsk % p.nseg;
        tsk /= p.nseg;
        const int rb = tsk % p.nrb, n = tsk / p.nrb;
        any = true;
        const int j0 = seg * QW;                            // first dz column / quad column
        const int kb = rb * p.rows_per_thread;              // quad rows kb .. ke - 1
        const int ke = min((p.H + 1) >> 1, kb + p.rows_per_thread);
        bool zok[NDZ], xok[2 * QW];
        unsigned zcol[NDZ], xcol[2 * QW];                    // byte offsets of the columns (an invalid one reads column 0: never used)
#pragma unroll
        for (int j = 0; j < NDZ; ++j) { zok[j] = j0 + j < p.OW; zcol[j] = zok[j] ? (unsigned)(j0 + j) * C * 2u : 0u; }
#pragma unroll
        for (int j = 0; j < 2 * QW; ++j) { xok[j] = 2 * j0 + j < p.W; xcol[j] = xok[j] ? (unsigned)(2 * j0 + j) * C * 2u : 0u; }
        const unsigned zoff = ((unsigned)n * p.OH * p.OW * C + c) * 2u, zstride = (unsigned)p.OW * C * 2u;
        const unsigned xoff = ((unsigned)n * p.H * p.W * C + c) * 2u, xstride = (unsigned)p.W * C * 2u;

        struct Raw { bf16x4 g[NDZ], z[NDZ]; };
        struct XRow { bf16x4 v[2 * QW]; };
        auto load_dz = [&](int r, Raw& raw) {               // unconditional, clamped (see the stride-1 walker)
            const unsigned ro = zoff + (unsigned)min(r, p.OH - 1) * zstride;
#pragma unroll
            for (int j = 0; j < NDZ; ++j) {
                raw.g[j] = *reinterpret_cast<const bf16x4*>(gb + (ro + zcol[j]));
                raw.z[j] = *reinterpret_cast<const bf16x4*>(zb + (ro + zcol[j]));
            }
        };
        auto load_x = [&](int r, XRow& xr) {
            const unsigned ro = xoff + (unsigned)min(r, p.H - 1) * xstride;
#pragma unroll
            for (int j = 0; j < 2 * QW; ++j) xr.v[j] = *reinterpret_cast<const bf16x4*>(xb + (ro + xcol[j]));
        };
        auto mkdz = [&](int r, const Raw& raw, f32x4 (&dst)[NDZ]) {
            const bool rok = r < p.OH;
            const int kb2 = opaque(c);
            const f32x4 ca = K(kb2, 9), cb = K(kb2, 10), cc = K(kb2, 11);
#pragma unroll
            for (int j = 0; j < NDZ; ++j) {
                const f32x4 gv = bf4_to_f32(raw.g[j]), zv = bf4_to_f32(raw.z[j]);
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = fmaf(ca[i], gv[i], fmaf(cb[i], zv[i], cc[i]));
                o = bf4_to_f32(f32_to_bf4(o));
                const bool ok = rok && zok[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = ok ? o[i] : 0.f;
                dst[j] = o;
            }
        };
        // one quad row: d0 = dz row k, d1 = dz row k + 1, x0 / x1 = input rows 2k, 2k + 1
        auto emit = [&](int k, const f32x4 (&d0)[NDZ], const f32x4 (&d1)[NDZ], const XRow& x0, const XRow& x1) {
            const int kc = opaque(c);
            const f32x4 bsc = K(kc, 12), bsh = K(kc, 13), bmu = K(kc, 14), bis = K(kc, 15);
            const bool r1ok = 2 * k + 1 < p.H;
            const unsigned ro0 = xoff + (unsigned)(2 * k) * xstride, ro1 = ro0 + xstride;
#pragma unroll
            for (int jq = 0; jq < QW; ++jq) {
                f32x4 acc[2][2], a[2][2], xv[2][2];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        xv[dy][dx] = bf4_to_f32(dy ? x1.v[2 * jq + dx] : x0.v[2 * jq + dx]);
                        const bool ok = xok[2 * jq + dx] && (dy == 0 || r1ok);
#pragma unroll
                        for (int i = 0; i < 4; ++i) a[dy][dx][i] = ok ? clamp_act(fmaf(xv[dy][dx][i], bsc[i], bsh[i]), lo, hi) : 0.f;
                    }
                const f32x4 g00 = d0[jq], g01 = d0[jq + 1], g10 = d1[jq], g11 = d1[jq + 1];
                {
                    // input (2k + dy, 2j + dx) <- dz (k + a, j + b) through tap (kh, kw) = (dy + 1 - 2a, dx + 1 - 2b), ascending (kh, kw)
                    // (explicit fused multiply-adds from zero, as dwconv_bwd_data_s2_kernel: contraction may fuse either product of a sum)
                    auto fma4 = [](const f32x4& u, const f32x4& v, const f32x4& w) {
                        f32x4 r;
#pragma unroll
                        for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(u[i], v[i], w[i]);
                        return r;
                    };
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    {
                        const int kw2 = opaque(c);
                        const f32x4 w1 = K(kw2, 1), w3 = K(kw2, 3), w4 = K(kw2, 4), w5 = K(kw2, 5), w7 = K(kw2, 7);
                        acc[0][0] = fma4(g00, w4, zero);
                        acc[0][1] = fma4(g00, w5, fma4(g01, w3, zero));
                        acc[1][0] = fma4(g00, w7, fma4(g10, w1, zero));
                    }
                    {
                        const int kw3 = opaque(c);
                        const f32x4 w0 = K(kw3, 0), w2 = K(kw3, 2), w6 = K(kw3, 6), w8 = K(kw3, 8);
                        acc[1][1] = fma4(g00, w8, fma4(g01, w6, fma4(g10, w2, fma4(g11, w0, zero))));
                    }
                }
                accw[4] += a[0][0] * g00;
                accw[3] += a[0][1] * g01;
                accw[5] += a[0][1] * g00;
                accw[1] += a[1][0] * g10;
                accw[7] += a[1][0] * g00;
                accw[0] += a[1][1] * g11;
                accw[2] += a[1][1] * g10;
                accw[6] += a[1][1] * g01;
                accw[8] += a[1][1] * g00;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        // (a pixel outside the image has a = 0: its mask, its value and its sums vanish; only the store is predicated)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[dy][dx][i] *= (a[dy][dx][i] > lo && a[dy][dx][i] < hi) ? 1.f : 0.f;
                        const bf16x4 ob = f32_to_bf4(acc[dy][dx]);
                        if (xok[2 * jq + dx] && (dy == 0 || r1ok)) *reinterpret_cast<bf16x4*>(db + ((dy ? ro1 : ro0) + xcol[2 * jq + dx])) = ob;
                        const f32x4 rv = bf4_to_f32(ob);
                        s += rv;
                        q += rv * ((xv[dy][dx] - bmu) * bis);
                    }
            }
        };

        f32x4 win[2][NDZ];
        Raw nxt;
        XRow xa, xb2, xna, xnb;
        { Raw r; load_dz(kb, r); mkdz(kb, r, win[0]); }
        load_dz(kb + 1, nxt);
        load_x(2 * kb, xna);
        load_x(2 * kb + 1, xnb);
#define DWB2_STEP(KQ, A, B)                                     \
        {                                                       \
            mkdz((KQ) + 1, nxt, win[B]);                        \
            xa = xna; xb2 = xnb;                                \
            load_dz((KQ) + 2, nxt);                             \
            load_x(2 * (KQ) + 2, xna);                          \
            load_x(2 * (KQ) + 3, xnb);                          \
            emit((KQ), win[A], win[B], xa, xb2);                \
        }
        for (int k = kb; k < ke; k += 2) {
            DWB2_STEP(k, 0, 1);
            if (k + 1 >= ke) break;
            DWB2_STEP(k + 1, 1, 0);
        }
#undef DWB2_STEP
    }
    fold_publish(p, dsm, accw, s, q, any, c);
}

// pixels per thread: 3 (256 registers, none spilled; 4 spills 78 of them, 2 re-reads every halo column)
int fused_segw() {
    static const int v = getenv("ADAMML_DWB_SEGW") ? atoi(getenv("ADAMML_DWB_SEGW")) : 3;          // A/B aid
    return v == 2 ? 2 : 3;
}

// quads per thread at stride 2: 1 (191 registers; 2 quads = 256 + 8 spilled for the same time over the net's four stride-2 layers --
// 1.185 against 1.179 ms -- so that instance is not built)
constexpr int fused_qw() { return 1; }

int fused_blocks(const adamml_conv_desc_t* d, int* rows_per_thread, int* nseg, int* nrb) {
    const long nchunk = d->Cin / 4;
    // stride 1: strips of fused_segw() pixels, walked down the image rows; stride 2: strips two quads wide, walked down the quad rows
    const int rows = d->stride == 2 ? (d->H + 1) / 2 : d->H;
    *nseg = d->stride == 2 ? ceil_div((d->W + 1) / 2, fused_qw()) : ceil_div(d->W, fused_segw());
    const int groups = d->groups < 1 ? 1 : d->groups;
    static const long min_blocks = getenv("ADAMML_DWB_MIN_BLOCKS") ? atol(getenv("ADAMML_DWB_MIN_BLOCKS")) : 256;       // A/B aids
    static const int max_rows = getenv("ADAMML_DWB_MAX_ROWS") ? atoi(getenv("ADAMML_DWB_MAX_ROWS")) : 24;
    int nb_rows = ceil_div(rows, max_rows);
    while (ceil_div(rows, nb_rows) > 3 && (long)groups * d->N * nb_rows * *nseg * nchunk < min_blocks * NT) ++nb_rows;
    const int rpt = ceil_div(rows, nb_rows);
    *rows_per_thread = rpt;
    *nrb = ceil_div(rows, rpt);
    const long threads = (long)d->N * *nrb * *nseg * nchunk;
    const long nb = (threads + NT - 1) / NT;
    // NT * nblk must be a multiple of nchunk (a thread keeps its channel chunk across tasks): 45 | nblk covers every C / 4 of the
    // MobileNetV2s; every workgroup publishes one [9][C] partial, so at most one workgroup per 144 pixels of a group
    static const long cap0 = getenv("ADAMML_DWB_CAP") ? atol(getenv("ADAMML_DWB_CAP")) : 2160;
    long cap = cap0 / groups / 45 * 45 > 0 ? cap0 / groups / 45 * 45 : 45;
    static const long px_per_wg = getenv("ADAMML_DWB_PX_PER_WG") ? atol(getenv("ADAMML_DWB_PX_PER_WG")) : 144;
    const long by_work = ((long)d->N * d->H * d->W / px_per_wg + 44) / 45 * 45;
    if (by_work < cap) cap = by_work > 45 ? by_work : 45;
    int nblk = nb >= cap ? (int)cap : (int)((nb + 44) / 45 * 45);
    if ((NT * (long)nblk) % nchunk != 0) nblk = (int)((nblk + nchunk - 1) / nchunk * nchunk);
    return nblk;
}

}  // namespace

extern "C" int adamml_dwconv_bwd_fused_supported(const adamml_conv_desc_t* d) {
    static const bool on = !(getenv("ADAMML_DW_BWD_FUSED") && atoi(getenv("ADAMML_DW_BWD_FUSED")) == 0);                // A/B aid
    return on && d && d->KH == 3 && d->KW == 3 && d->Cin == d->Cout && d->Cin % 8 == 0 && d->Cin <= 960 && d->pad == 1 &&
           ((d->stride == 1 && d->OH == d->H && d->OW == d->W) || (d->stride == 2 && d->OH == (d->H - 1) / 2 + 1 && d->OW == (d->W - 1) / 2 + 1)) &&
           (size_t)d->N * d->H * d->W * d->Cin * 2 < ((size_t)1 << 32) ? 1 : 0;       // (32-bit byte offsets within a group)
}

extern "C" size_t adamml_dwconv_bwd_fused_workspace(const adamml_conv_desc_t* d) {
    if (!adamml_dwconv_bwd_fused_supported(d)) return 0;
    int a, b, c;
    return (size_t)(d->groups < 1 ? 1 : d->groups) * fused_blocks(d, &a, &b, &c) * 9 * d->Cin * sizeof(float);
}

extern "C" int adamml_dwconv_bwd_fused(const adamml_conv_desc_t* d, const void* g, const void* z, const float* aff, const float* w,
                                       const void* x, const float* x_vec, int x_act, void* dx, double* sums, float* dw, void* workspace,
                                       size_t workspace_bytes, hipStream_t stream) {
    if (!adamml_dwconv_bwd_fused_supported(d))
        return adamml_set_error(ADAMML_EUNSUPPORTED, "dwconv_bwd_fused: 3x3 depthwise, pad 1, stride 1 / 2, C %% 8 == 0, C <= 960, a group below 4 GB");
    if (!g || !z || !aff || !w || !x || !x_vec || !dx || !sums || !dw || !workspace)
        return adamml_set_error(ADAMML_EINVAL, "dwconv_bwd_fused: null argument");
    DwBP p;
    p.g = (const bf16_t*)g; p.z = (const bf16_t*)z; p.aff = aff; p.w = w; p.x = (const bf16_t*)x; p.xvec = x_vec; p.dx = (bf16_t*)dx;
    p.stats = sums; p.ws = (float*)workspace;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->Cin; p.OH = d->OH; p.OW = d->OW; p.xact = x_act;
    const size_t P = (size_t)d->N * d->H * d->W;
    if (!P) return ADAMML_OK;
    const int groups = d->groups < 1 ? 1 : d->groups;
    const int nblk = fused_blocks(d, &p.rows_per_thread, &p.nseg, &p.nrb);
    if (workspace_bytes < (size_t)groups * nblk * 9 * p.C * sizeof(float))
        return adamml_set_error(ADAMML_EINVAL, "dwconv_bwd_fused: workspace too small (adamml_dwconv_bwd_fused_workspace)");
    p.gz = (size_t)d->N * d->OH * d->OW * d->Cin; p.gx = P * d->Cin;
    if (d->stride == 2) hipLaunchKernelGGL(dwconv_bwd_fused_s2_kernel<fused_qw()>, dim3(nblk, groups), dim3(NT), 16 * p.C * sizeof(float), stream, p);
    else if (fused_segw() == 2) hipLaunchKernelGGL((dwconv_bwd_fused_kernel<1, 2>), dim3(nblk, groups), dim3(NT), 16 * p.C * sizeof(float), stream, p);
    else hipLaunchKernelGGL((dwconv_bwd_fused_kernel<1, 3>), dim3(nblk, groups), dim3(NT), 16 * p.C * sizeof(float), stream, p);
    int rc = adamml_check_launch("dwconv_bwd_fused");
    if (rc) return rc;
    return adamml_launch_split_reduce(p.ws, dw, (size_t)9 * p.C, groups * nblk, stream);
}
