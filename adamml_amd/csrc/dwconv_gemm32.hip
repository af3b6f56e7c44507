// Depthwise 3x3 convolution (MobileNetV2 blocks: models/sound_mobilenet_v2.py:58, models/policy_net.py:66,80)
// forward / data-grad / weight-grad as VALU kernels (K = 9 per channel: bandwidth work, not MFMA work), and the
// fp32 strided GEMM used by the Linear / LSTMCell layers of the policy head and the classifier heads.
#include "common.h"
#include "../../include/adamml_hip.h"


namespace {

constexpr int NT = 256;
constexpr int MAXC = 2048;

struct ChanMap {
    int nchunk, rows_per_pass, chunk, rslot;
    bool active;
    __device__ ChanMap(int C, int tid) {
        nchunk = C >> 3;
        rows_per_pass = NT / nchunk;
        if (rows_per_pass < 1) rows_per_pass = 1;
        active = tid < rows_per_pass * nchunk;
        chunk = tid % nchunk;
        rslot = tid / nchunk;
    }
};

struct DwP {
    const bf16_t* x;
    const float* w;       // [9][C]
    const float* in_scale;
    const float* in_shift;
    bf16_t* y;
    double* stats;
    int N, H, W, C, OH, OW, stride, pad, act, accumulate, nseg, seglen;
    int nrb, rows_per_thread;   // forward: row blocks per image / output rows walked by one thread
    int flip;                   // forward kernel used as the stride-1 data gradient: taps read in reverse order
    int cw, tpb, nct;           // walker workgroup = cw channel chunks x tpb strips (dw_walk_grid); nct = (C / 4) / cw chunk tiles
    size_t P, ppb;
    size_t gx, gy;        // element strides between BatchNorm groups of x / y (blockIdx.y = group)
    int in_gstride;
    // data gradient w.r.t. a lazily normalised tensor (BNZ kernels): y = g * act'(bn(bn_z)) and stats += sum(y), sum(y zhat)
    const bf16_t* bn_z;   // raw tensor y is the gradient of (same shape / group stride as y)
    const float* bn_vec;  // [G][4][C] scale, shift, mean, invstd
    int bn_act;
    // X1 kernels (3x3 stem of a ONE-channel fp32 image, models/sound_mobilenet_v2.py:96 / models/policy_net.py:108 on a spectrogram): the
    // input is x1 [.. H, W] fp32 -- image n of group g at x1 + g * x1_g + n * x1_n floats -- broadcast over the C output channels
    const float* x1;
    size_t x1_g, x1_n;
};

// "Column-strip walker": one thread owns 4 channels x SEGW adjacent output columns and walks DOWN `rows_per_thread`
// output rows with a 3-row register window of BatchNorm+ReLU6-transformed inputs.  Every input row is loaded and
// transformed once per strip (the earlier row-wise form re-transformed each element for the 3 output rows that use it:
// 31 VALU ops per output element, VALU-bound at ~1 TB/s); the window rotates by compile-time index (rows are processed in
// groups of 3), the next input row(s) are in flight while the current output row is computed, and the per-channel
// statistics stay in registers for the whole walk (one LDS fold per thread, one global publication per workgroup).
// BNZ (stride-1 data gradient, the walk over dz with reversed taps): the outputs are the gradient w.r.t. the ACTIVATED value of a lazily
// normalised tensor z -- the mask act'(bn(z)) is applied before the store and the statistics become sum(g'), sum(g' zhat), so the
// BatchNorm-backward reduction pass over (g, z) disappears (one read of z here instead of a read of g and of z there).  The z rows
// are requested three output rows ahead (three rotating register rows, like the input window).
// WF: OW % SEGW == 0 (the launcher's choice) -- every output column of a thread's segment exists, so the stores of the row walk are
// unconditional code; the next input rows are requested unconditionally either way (load_row clamps): no load or store of the walk sits
// under a per-thread condition (behind one, every wait is a conservative one).
template <int S, bool BNZ = false, bool X1 = false, bool WF = false>
__global__ __launch_bounds__(NT, 2) void dwconv_fwd_kernel(DwP p) {
    constexpr int SEGW = S == 1 ? 4 : 2;                // output columns per thread
    constexpr int NCOL = (SEGW - 1) * S + 3;            // input columns feeding them
    __shared__ float smem[2 * MAXC];
    const unsigned lb = xcd_contiguous(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const unsigned bx = lb % gridDim.x;
    {
        const int g = lb / gridDim.x;
        if (X1) p.x1 += (size_t)g * p.x1_g; else p.x += (size_t)g * p.gx;
        p.y += (size_t)g * p.gy;
        if (p.stats) p.stats += (size_t)g * ADAMML_STAT_SLOTS * 2 * p.C;
        if (p.in_scale) { p.in_scale += (size_t)g * p.in_gstride; p.in_shift += (size_t)g * p.in_gstride; }
        if (BNZ) { p.bn_z += (size_t)g * p.gy; p.bn_vec += (size_t)g * 4 * p.C; }
    }
    // workgroup = one tile of cw channel chunks x tpb strips: the tpb strips fold their statistics in LDS, so a workgroup publishes
    // 8 cw sums instead of (with chunk-major thread ids and C / 4 close to NT) one per thread
    const int ct = bx % p.nct, cl = threadIdx.x % p.cw, tl = threadIdx.x / p.cw;
    const int chunk = ct * p.cw + cl;
    int tsk = (bx / p.nct) * p.tpb + tl;
    const int seg = tsk % p.nseg;
    tsk /= p.nseg;
    const int rb = tsk % p.nrb, n = tsk / p.nrb;
    const bool active = tl < p.tpb && n < p.N;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const int c = chunk * 4;
        const int ow_b = seg * SEGW;
        const int iw_b = ow_b * S - p.pad;
        const int oh_b = rb * p.rows_per_thread;
        const int oh_e = min(p.OH, oh_b + p.rows_per_thread);
        f32x4 wt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const f32x4*>(p.w + (size_t)(p.flip ? 8 - t : t) * p.C + c);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        const float lo = p.in_scale ? act_lo(p.act) : -INFINITY, hi = p.in_scale ? act_hi(p.act) : INFINITY;
        if (p.in_scale) {
            sc = *reinterpret_cast<const f32x4*>(p.in_scale + c);
            sh = *reinterpret_cast<const f32x4*>(p.in_shift + c);
        }
        bool cok[NCOL];
#pragma unroll
        for (int j = 0; j < NCOL; ++j) cok[j] = (unsigned)(iw_b + j) < (unsigned)p.W;
        const bf16_t* img = X1 ? nullptr : p.x + (size_t)n * p.H * p.W * p.C + c;
        const float* img1 = X1 ? p.x1 + (size_t)n * p.x1_n : nullptr;
        bf16_t* yimg = p.y + (size_t)n * p.OH * p.OW * p.C + c;
        f32x4 bsc, bsh, bmu, bis;
        float blo = 0.f, bhi = 0.f;
        const bf16_t* zimg = nullptr;
        struct ZRow { bf16x4 v[SEGW]; };
        ZRow zr[BNZ ? 3 : 1];
        if (BNZ) {
            bsc = *reinterpret_cast<const f32x4*>(p.bn_vec + c); bsh = *reinterpret_cast<const f32x4*>(p.bn_vec + p.C + c);
            bmu = *reinterpret_cast<const f32x4*>(p.bn_vec + 2 * p.C + c); bis = *reinterpret_cast<const f32x4*>(p.bn_vec + 3 * p.C + c);
            blo = act_lo(p.bn_act); bhi = act_hi(p.bn_act);
            zimg = p.bn_z + (size_t)n * p.OH * p.OW * p.C + c;
        }
        auto load_z = [&](int oh, ZRow& r) {                 // unconditional, clamped (rows / columns past the end are never used)
            const bf16_t* rp = zimg + (size_t)min(oh, p.OH - 1) * p.OW * p.C;
#pragma unroll
            for (int o = 0; o < SEGW; ++o) r.v[o] = *reinterpret_cast<const bf16x4*>(rp + (size_t)min(ow_b + o, p.OW - 1) * p.C);
        };

        struct Raw { bf16x4 v[X1 ? 1 : NCOL]; float f[X1 ? NCOL : 1]; bool rok; };
        auto load_row = [&](int ih, Raw& r) {
            r.rok = (unsigned)ih < (unsigned)p.H;
            if constexpr (X1) {
                const float* rp = img1 + (size_t)(r.rok ? ih : 0) * p.W;
#pragma unroll
                for (int j = 0; j < NCOL; ++j) r.f[j] = rp[cok[j] ? iw_b + j : 0];
            } else {
            const bf16_t* rp = img + (size_t)(r.rok ? ih : 0) * p.W * p.C;
#pragma unroll
            for (int j = 0; j < NCOL; ++j)          // unconditional loads from clamped addresses; xform() zeroes the invalid ones
                r.v[j] = *reinterpret_cast<const bf16x4*>(rp + (size_t)(cok[j] ? iw_b + j : 0) * p.C);
            }
        };
        auto xform = [&](const Raw& r, f32x4 (&dst)[NCOL]) {
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                const bool ok = r.rok && cok[j];
                if constexpr (X1) {                 // the fp32 pixel itself, for every output channel (no input BatchNorm: first layer)
                    const float v = ok ? r.f[j] : 0.f;
                    dst[j] = f32x4{v, v, v, v};
                } else {
                f32x4 v = bf4_to_f32(r.v[j]);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ok ? clamp_act(fmaf(v[i], sc[i], sh[i]), lo, hi) : 0.f;
                dst[j] = v;
                }
            }
        };
        auto emit = [&](int oh, const f32x4 (&r0)[NCOL], const f32x4 (&r1)[NCOL], const f32x4 (&r2)[NCOL], const ZRow& zrow) {
            bf16_t* yrow = yimg + (size_t)oh * p.OW * p.C;
#pragma unroll
            for (int o = 0; o < SEGW; ++o) {
                if (WF || ow_b + o < p.OW) {
                    f32x4 acc = r0[o * S] * wt[0];
                    acc += r0[o * S + 1] * wt[1];
                    acc += r0[o * S + 2] * wt[2];
                    acc += r1[o * S] * wt[3];
                    acc += r1[o * S + 1] * wt[4];
                    acc += r1[o * S + 2] * wt[5];
                    acc += r2[o * S] * wt[6];
                    acc += r2[o * S + 1] * wt[7];
                    acc += r2[o * S + 2] * wt[8];
                    if (BNZ) {
                        const f32x4 zv = bf4_to_f32(zrow.v[o]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] *= mask_act(fmaf(zv[i], bsc[i], bsh[i]), blo, bhi);
                        const bf16x4 ob = f32_to_bf4(acc);
                        *reinterpret_cast<bf16x4*>(yrow + (size_t)(ow_b + o) * p.C) = ob;
                        const f32x4 rv = bf4_to_f32(ob);
                        s += rv;
                        q += rv * ((zv - bmu) * bis);
                    } else {
                        const bf16x4 ob = f32_to_bf4(acc);
                        *reinterpret_cast<bf16x4*>(yrow + (size_t)(ow_b + o) * p.C) = ob;
                        const f32x4 rv = bf4_to_f32(ob);
                        s += rv;
                        q += rv * rv;
                    }
                }
            }
        };

        f32x4 win[3][NCOL];
        Raw nxt[S];                                          // rows in flight (S new input rows per output row)
        if (S == 1) {
            // rows ih = oh-1, oh, oh+1: window slot of input row ih is (ih - (oh_b - 1)) % 3
            { Raw r; load_row(oh_b * S - p.pad, r); xform(r, win[0]); }
            { Raw r; load_row(oh_b * S - p.pad + 1, r); xform(r, win[1]); }
            load_row(oh_b - p.pad + 2, nxt[0]);
            if (BNZ) { load_z(oh_b, zr[0]); load_z(oh_b + 1, zr[BNZ ? 1 : 0]); load_z(oh_b + 2, zr[BNZ ? 2 : 0]); }
            for (int oh = oh_b; oh < oh_e; oh += 3) {
                xform(nxt[0], win[2]);
                load_row(oh + 1 - p.pad + 2, nxt[0]);
                emit(oh, win[0], win[1], win[2], zr[0]);
                if (BNZ && oh + 3 < oh_e) load_z(oh + 3, zr[0]);
                if (oh + 1 >= oh_e) break;
                xform(nxt[0], win[0]);
                load_row(oh + 2 - p.pad + 2, nxt[0]);
                emit(oh + 1, win[1], win[2], win[0], zr[BNZ ? 1 : 0]);
                if (BNZ && oh + 4 < oh_e) load_z(oh + 4, zr[BNZ ? 1 : 0]);
                if (oh + 2 >= oh_e) break;
                xform(nxt[0], win[1]);
                load_row(oh + 3 - p.pad + 2, nxt[0]);
                emit(oh + 2, win[2], win[0], win[1], zr[BNZ ? 2 : 0]);
                if (BNZ && oh + 5 < oh_e) load_z(oh + 5, zr[BNZ ? 2 : 0]);
            }
        } else {
            // rows ih = 2oh-1, 2oh, 2oh+1; row 2oh+1 is kept as the top row of output row oh+1
            { Raw r; load_row(oh_b * 2 - p.pad, r); xform(r, win[0]); }
            load_row(oh_b * 2 - p.pad + 1, nxt[0]);
            load_row(oh_b * 2 - p.pad + 2, nxt[1]);
            for (int oh = oh_b; oh < oh_e; oh += 3) {
                xform(nxt[0], win[1]); xform(nxt[1], win[2]);
                { load_row((oh + 1) * 2 - p.pad + 1, nxt[0]); load_row((oh + 1) * 2 - p.pad + 2, nxt[1]); }
                emit(oh, win[0], win[1], win[2], zr[0]);
                if (oh + 1 >= oh_e) break;
                xform(nxt[0], win[0]); xform(nxt[1], win[1]);
                { load_row((oh + 2) * 2 - p.pad + 1, nxt[0]); load_row((oh + 2) * 2 - p.pad + 2, nxt[1]); }
                emit(oh + 1, win[2], win[0], win[1], zr[0]);
                if (oh + 2 >= oh_e) break;
                xform(nxt[0], win[2]); xform(nxt[1], win[0]);
                { load_row((oh + 3) * 2 - p.pad + 1, nxt[0]); load_row((oh + 3) * 2 - p.pad + 2, nxt[1]); }
                emit(oh + 2, win[1], win[2], win[0], zr[0]);
            }
        }
    }
    if (p.stats) {
        // every strip (tl) deposits its sums in its own LDS row [2 nch]; one thread per channel folds the rows in strip order and
        // publishes the workgroup's partial exactly (common.h: reproducible reductions).  tpb * 2 nch = 8 cw tpb <= 8 NT floats.
        const int nch = 4 * p.cw;                           // channels of this workgroup's tile
        // (the strip / chunk indices are recomputed from an opaque copy of the thread id: kept live across the walk they were the two
        // registers the BNZ instance spilled to scratch)
        int tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));
        const int cl2 = tid2 % p.cw, tl2 = tid2 / p.cw;
        if (tl2 < p.tpb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                smem[tl2 * 2 * nch + cl2 * 4 + i] = active ? s[i] : 0.f;
                smem[tl2 * 2 * nch + nch + cl2 * 4 + i] = active ? q[i] : 0.f;
            }
        }
        __syncthreads();
        const unsigned slot = (bx / p.nct) & (ADAMML_STAT_SLOTS - 1);
        for (int i = threadIdx.x; i < 2 * nch; i += NT) {
            float v = 0.f;
            for (int r = 0; r < p.tpb; ++r) v += smem[r * 2 * nch + i];
            if (v != 0.f) stat_publish(p.stats + ct * nch + (i < nch ? i : p.C + i - nch), 2 * (size_t)p.C, slot, v);
        }
    }
}

// dx[n,ih,iw,c] = sum_{kh,kw} dz[n,(ih+pad-kh)/s,(iw+pad-kw)/s,c] * w[kh,kw,c]
__global__ __launch_bounds__(NT) void dwconv_bwd_data_kernel(DwP p) {   // p.x = dz [N,OH,OW,C], p.y = dx [N,H,W,C]
    const unsigned lb = xcd_contiguous(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    p.x += (size_t)(lb / gridDim.x) * p.gx;
    p.y += (size_t)(lb / gridDim.x) * p.gy;
    ChanMap m(p.C, threadIdx.x);
    if (!m.active) return;
    const size_t pb = (size_t)(lb % gridDim.x) * p.ppb;
    const size_t pe = pb + p.ppb < p.P ? pb + p.ppb : p.P;
    const int c = m.chunk * 8;
    f32x8 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = load_f32x8(p.w + (size_t)t * p.C + c);
    for (size_t pp = pb + m.rslot; pp < pe; pp += m.rows_per_pass) {
        const int iw = (int)(pp % p.W);
        size_t r = pp / p.W;
        const int ih = (int)(r % p.H);
        const int n = (int)(r / p.H);
        f32x8 acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        if (p.accumulate) acc = bf8_to_f32(*reinterpret_cast<const bf16x8*>(p.y + pp * p.C + c));
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                int vh = ih + p.pad - kh, vw = iw + p.pad - kw;
                if (vh < 0 || vw < 0) continue;
                if (p.stride == 2) {
                    if ((vh | vw) & 1) continue;
                    vh >>= 1; vw >>= 1;
                }
                if (vh >= p.OH || vw >= p.OW) continue;
                f32x8 g = bf8_to_f32(*reinterpret_cast<const bf16x8*>(p.x + (((size_t)n * p.OH + vh) * p.OW + vw) * p.C + c));
                acc += g * wt[kh * 3 + kw];
            }
        *reinterpret_cast<bf16x8*>(p.y + pp * p.C + c) = f32_to_bf8(acc);
    }
}

// Stride 2, pad 1: a thread owns one 8-channel chunk of a 2x2 INPUT quad (rows 2k, 2k+1; columns 2j, 2j+1).  The quad receives from
// exactly the four dz pixels (k+a, j+b), a, b in {0, 1} -- pixel (0,0) one tap, (0,1) and (1,0) two, (1,1) four -- so the four loads
// are unconditional (clamped addresses, out-of-range pixels zeroed afterwards) and issued together; the per-pixel kernel above tests
// the parity of every tap and branches around each load (nine dependent L2 round trips per pixel: 1.6 TB/s).  Taps are accumulated in
// the same (kh, kw) order as there: results are bit-identical.
template <bool BNZ>
__global__ __launch_bounds__(NT) void dwconv_bwd_data_s2_kernel(DwP p) {   // p.x = dz [N,OH,OW,C], p.y = dx [N,H,W,C]; p.P = quads
    __shared__ float smem[BNZ ? 2 * MAXC : 1];
    const unsigned lb = xcd_contiguous(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    {
        const size_t g = lb / gridDim.x;
        p.x += g * p.gx;
        p.y += g * p.gy;
        if (BNZ) { p.bn_z += g * p.gy; p.bn_vec += g * 4 * p.C; p.stats += g * ADAMML_STAT_SLOTS * 2 * p.C; }
    }
    ChanMap m(p.C, threadIdx.x);
    if (!BNZ && !m.active) return;
    const size_t qb = (size_t)(lb % gridDim.x) * p.ppb;
    const size_t qe = qb + p.ppb < p.P ? qb + p.ppb : p.P;
    const int c = m.chunk * 8;
    const int QH = (p.H + 1) >> 1, QW = (p.W + 1) >> 1;
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = q[i] = 0.f;
    if (m.active) {
        f32x8 wt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = load_f32x8(p.w + (size_t)t * p.C + c);
        f32x8 bsc, bsh, bmu, bis;
        float blo = 0.f, bhi = 0.f;
        if (BNZ) {
            bsc = load_f32x8(p.bn_vec + c); bsh = load_f32x8(p.bn_vec + p.C + c);
            bmu = load_f32x8(p.bn_vec + 2 * p.C + c); bis = load_f32x8(p.bn_vec + 3 * p.C + c);
            blo = act_lo(p.bn_act); bhi = act_hi(p.bn_act);
        }
        for (size_t qi = qb + m.rslot; qi < qe; qi += m.rows_per_pass) {
            const int j = (int)(qi % QW);
            size_t r = qi / QW;
            const int k = (int)(r % QH);
            const int n = (int)(r / QH);
            bf16x8 raw[2][2], prev[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int vh = min(k + a, p.OH - 1), vw = min(j + b, p.OW - 1);
                    raw[a][b] = *reinterpret_cast<const bf16x8*>(p.x + (((size_t)n * p.OH + vh) * p.OW + vw) * p.C + c);
                }
            if (BNZ || p.accumulate) {                       // BNZ: the z values of the four pixels; accumulate: their previous gradient
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int ih = min(2 * k + dy, p.H - 1), iw = min(2 * j + dx, p.W - 1);
                        prev[dy][dx] = *reinterpret_cast<const bf16x8*>((BNZ ? p.bn_z : p.y) + (((size_t)n * p.H + ih) * p.W + iw) * p.C + c);
                    }
            }
            f32x8 g[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const bool ok = k + a < p.OH && j + b < p.OW;
                    g[a][b] = bf8_to_f32(raw[a][b]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) g[a][b][i] = ok ? g[a][b][i] : 0.f;
                }
            f32x8 acc[2][2];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[dy][dx][i] = 0.f;
                    if (!BNZ && p.accumulate) acc[dy][dx] = bf8_to_f32(prev[dy][dx]);
                }
            // input (2k + dy, 2j + dx) <- dz (k + a, j + b) through tap (kh, kw) = (dy + 1 - 2a, dx + 1 - 2b), ascending (kh, kw)
            // (explicit fused multiply-adds: left to contraction, a sum of two products may fuse either one, and the one-pass backward of
            // csrc/dwconv_bwd_fused.hip must round exactly as this kernel does)
            auto fma8 = [](const f32x8& a, const f32x8& b, const f32x8& c) {
                f32x8 r;
#pragma unroll
                for (int i = 0; i < 8; ++i) r[i] = __builtin_fmaf(a[i], b[i], c[i]);
                return r;
            };
            acc[0][0] = fma8(g[0][0], wt[4], acc[0][0]);
            acc[0][1] = fma8(g[0][1], wt[3], acc[0][1]);
            acc[0][1] = fma8(g[0][0], wt[5], acc[0][1]);
            acc[1][0] = fma8(g[1][0], wt[1], acc[1][0]);
            acc[1][0] = fma8(g[0][0], wt[7], acc[1][0]);
            acc[1][1] = fma8(g[1][1], wt[0], acc[1][1]);
            acc[1][1] = fma8(g[1][0], wt[2], acc[1][1]);
            acc[1][1] = fma8(g[0][1], wt[6], acc[1][1]);
            acc[1][1] = fma8(g[0][0], wt[8], acc[1][1]);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const bool ok = 2 * k + dy < p.H && 2 * j + dx < p.W;
                    bf16x8 ob;
                    if (BNZ) {
                        const f32x8 zv = bf8_to_f32(prev[dy][dx]);
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[dy][dx][i] *= mask_act(fmaf(zv[i], bsc[i], bsh[i]), blo, bhi);
                        ob = f32_to_bf8(acc[dy][dx]);
                        const f32x8 rv = bf8_to_f32(ob);
                        const float keep = ok ? 1.f : 0.f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float gp = keep * rv[i];
                            s[i] += gp;
                            q[i] += gp * ((zv[i] - bmu[i]) * bis[i]);
                        }
                    } else ob = f32_to_bf8(acc[dy][dx]);
                    if (ok) *reinterpret_cast<bf16x8*>(p.y + (((size_t)n * p.H + 2 * k + dy) * p.W + 2 * j + dx) * p.C + c) = ob;
                }
        }
    }
    if (BNZ) {
        // row slot r of the thread map deposits its sums in LDS row r [2C]; folded in row order, published exactly (common.h)
        if (m.active) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                smem[m.rslot * 2 * p.C + c + i] = s[i];
                smem[m.rslot * 2 * p.C + p.C + c + i] = q[i];
            }
        }
        __syncthreads();
        const int nrow = m.rows_per_pass;
        for (int i = threadIdx.x; i < 2 * p.C; i += NT) {
            float v = 0.f;
            for (int r = 0; r < nrow; ++r) v += smem[r * 2 * p.C + i];
            if (v != 0.f) stat_publish(p.stats + i, 2 * (size_t)p.C, blockIdx.x & (ADAMML_STAT_SLOTS - 1), v);
        }
    }
}

struct DwWP {
    const bf16_t* dz;
    const bf16_t* x;
    const float* in_scale;
    const float* in_shift;
    float* dw;            // [C][3][3] fp32
    float* ws;            // optional [gridDim.y][gridDim.x][9*C] partial buffer in dw layout
    int N, H, W, C, OH, OW, stride, pad, act, rows_per_thread, nseg, nrb;
    size_t gdz, gx;
    int in_gstride;
    const float* x1;      // X1 kernels: one-channel fp32 input (DwP::x1)
    size_t x1_g, x1_n;
};

// Weight gradient with the forward kernel's column-strip walk: dw[c][kh][kw] = sum_p dz[p][c] * a[p@(kh,kw)][c] over a
// 3-row register window of transformed inputs (each input row loaded and transformed once per strip), 36 register
// accumulators per thread.  The grid is capped (thread count a multiple of every channel-group count of MobileNetV2) with
// a task loop: a thread keeps its 4-channel group, publishes its accumulators once, and a workgroup writes ONE partial
// [9*C] tile (plain stores into the workspace, summed by adamml_launch_split_reduce; fp32 atomics without a workspace).
template <int S, bool X1 = false>
__global__ __launch_bounds__(NT, 2) void dwconv_bwd_weight_kernel(DwWP p) {
    constexpr int SEGW = S == 1 ? 4 : 2;
    constexpr int NCOL = (SEGW - 1) * S + 3;
    extern __shared__ float dsm[];        // [9][C]
    p.dz += (size_t)blockIdx.y * p.gdz;
    if (X1) p.x1 += (size_t)blockIdx.y * p.x1_g; else p.x += (size_t)blockIdx.y * p.gx;
    if (p.in_scale) { p.in_scale += (size_t)blockIdx.y * p.in_gstride; p.in_shift += (size_t)blockIdx.y * p.in_gstride; }
    const int nchunk = p.C >> 2;
    for (int i = threadIdx.x; i < 9 * p.C; i += NT) dsm[i] = 0.f;
    __syncthreads();
    const int gid = blockIdx.x * NT + threadIdx.x;
    const int nthreads = gridDim.x * NT;
    const int chunk = gid % nchunk;
    const long ntasks = (long)nchunk * p.nseg * p.nrb * p.N;
    const int c = chunk * 4;
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    const float lo = p.in_scale ? act_lo(p.act) : -INFINITY, hi = p.in_scale ? act_hi(p.act) : INFINITY;
    if (p.in_scale) {
        sc = *reinterpret_cast<const f32x4*>(p.in_scale + c);
        sh = *reinterpret_cast<const f32x4*>(p.in_shift + c);
    }
    bool any = false;
    for (long task = gid; task < ntasks; task += nthreads) {
        int tsk = (int)(task / nchunk);
        const int seg = tsk % p.nseg;
        tsk /= p.nseg;
        const int rb = tsk % p.nrb, n = tsk / p.nrb;
        any = true;
        const int ow_b = seg * SEGW;
        const int iw_b = ow_b * S - p.pad;
        const int oh_b = rb * p.rows_per_thread;
        const int oh_e = min(p.OH, oh_b + p.rows_per_thread);
        bool cok[NCOL], ook[SEGW];
#pragma unroll
        for (int j = 0; j < NCOL; ++j) cok[j] = (unsigned)(iw_b + j) < (unsigned)p.W;
#pragma unroll
        for (int o = 0; o < SEGW; ++o) ook[o] = ow_b + o < p.OW;
        const bf16_t* img = X1 ? nullptr : p.x + (size_t)n * p.H * p.W * p.C + c;
        const float* img1 = X1 ? p.x1 + (size_t)n * p.x1_n : nullptr;
        const bf16_t* gimg = p.dz + ((size_t)n * p.OH * p.OW + ow_b) * p.C + c;

        struct Raw { bf16x4 v[X1 ? 1 : NCOL]; float f[X1 ? NCOL : 1]; bool rok; };
        struct GRow { bf16x4 v[SEGW]; };
        auto load_row = [&](int ih, Raw& r) {
            r.rok = (unsigned)ih < (unsigned)p.H;
            if constexpr (X1) {
                const float* rp = img1 + (size_t)(r.rok ? ih : 0) * p.W;
#pragma unroll
                for (int j = 0; j < NCOL; ++j) r.f[j] = rp[cok[j] ? iw_b + j : 0];
            } else {
            const bf16_t* rp = img + (size_t)(r.rok ? ih : 0) * p.W * p.C;
#pragma unroll
            for (int j = 0; j < NCOL; ++j)          // unconditional loads from clamped addresses; xform() zeroes the invalid ones
                r.v[j] = *reinterpret_cast<const bf16x4*>(rp + (size_t)(cok[j] ? iw_b + j : 0) * p.C);
            }
        };
        auto load_g = [&](int oh, GRow& gr) {
            const bf16_t* rp = gimg + (size_t)oh * p.OW * p.C;
#pragma unroll
            for (int o = 0; o < SEGW; ++o) {
                bf16x4 v = *reinterpret_cast<const bf16x4*>(rp + (size_t)(ook[o] ? o : 0) * p.C);
                if (!ook[o]) v = bf16x4{0, 0, 0, 0};
                gr.v[o] = v;
            }
        };
        auto xform = [&](const Raw& r, f32x4 (&dst)[NCOL]) {
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                const bool ok = r.rok && cok[j];
                if constexpr (X1) {
                    const float v = ok ? r.f[j] : 0.f;
                    dst[j] = f32x4{v, v, v, v};
                } else {
                f32x4 v = bf4_to_f32(r.v[j]);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ok ? clamp_act(fmaf(v[i], sc[i], sh[i]), lo, hi) : 0.f;
                dst[j] = v;
                }
            }
        };
        auto emit = [&](const GRow& gr, const f32x4 (&r0)[NCOL], const f32x4 (&r1)[NCOL], const f32x4 (&r2)[NCOL]) {
#pragma unroll
            for (int o = 0; o < SEGW; ++o) {
                const f32x4 g = bf4_to_f32(gr.v[o]);            // zero for columns past OW
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    acc[kw] += g * r0[o * S + kw];
                    acc[3 + kw] += g * r1[o * S + kw];
                    acc[6 + kw] += g * r2[o * S + kw];
                }
            }
        };
        f32x4 win[3][NCOL];
        Raw nxt[S];
        GRow gcur, gnxt;
        { Raw r; load_row(oh_b * S - p.pad, r); xform(r, win[0]); }
        if (S == 1) {
            { Raw r; load_row(oh_b - p.pad + 1, r); xform(r, win[1]); }
            load_row(oh_b - p.pad + 2, nxt[0]);
        } else {
            load_row(oh_b * 2 - p.pad + 1, nxt[0]);
            load_row(oh_b * 2 - p.pad + 2, nxt[1]);
        }
        load_g(oh_b, gnxt);
        // window slot roles rotate with period 3 (compile-time indices): a = top row, then the S new rows
#define DW_STEP(OHV, A, B, C2)                                                                                  \
        {                                                                                                       \
            if (S == 1) xform(nxt[0], win[C2]); else { xform(nxt[0], win[B]); xform(nxt[S - 1], win[C2]); }     \
            gcur = gnxt;                                                                                        \
            if ((OHV) + 1 < oh_e) {                                                                             \
                if (S == 1) load_row((OHV) + 1 - p.pad + 2, nxt[0]);                                            \
                else { load_row(((OHV) + 1) * 2 - p.pad + 1, nxt[0]); load_row(((OHV) + 1) * 2 - p.pad + 2, nxt[S - 1]); } \
                load_g((OHV) + 1, gnxt);                                                                        \
            }                                                                                                   \
            emit(gcur, win[A], win[B], win[C2]);                                                                \
        }
        for (int oh = oh_b; oh < oh_e; oh += 3) {
            if (S == 1) {
                DW_STEP(oh, 0, 1, 2);
                if (oh + 1 >= oh_e) break;
                DW_STEP(oh + 1, 1, 2, 0);
                if (oh + 2 >= oh_e) break;
                DW_STEP(oh + 2, 2, 0, 1);
            } else {
                DW_STEP(oh, 0, 1, 2);
                if (oh + 1 >= oh_e) break;
                DW_STEP(oh + 1, 2, 0, 1);
                if (oh + 2 >= oh_e) break;
                DW_STEP(oh + 2, 1, 2, 0);
            }
        }
#undef DW_STEP
    }
    {
        // the threads that share a channel chunk (thread ids congruent modulo nchunk) add in turn: a fixed order, unlike LDS atomics
        // (common.h: reproducible reductions; <= NT / nchunk barriers once per workgroup)
        const int nturn = (NT + nchunk - 1) / nchunk;
        for (int r = 0; r < nturn; ++r) {
            if (any && (int)threadIdx.x / nchunk == r) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) dsm[t * p.C + c + i] += acc[t][i];
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * p.C; i += NT) {
        const int t = i / p.C, cc = i - t * p.C;
        if (p.ws) p.ws[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 9 * p.C + (size_t)cc * 9 + t] = dsm[i];
        else if (dsm[i] != 0.f) atomicAdd(&p.dw[(size_t)cc * 9 + t], dsm[i]);
    }
}

// ------------------------------------------------------------------------------------------------ fp32 GEMM
// C[m,n] (+)= act(sum_k A[m,k] * B[n,k] + bias[n]);  arbitrary element strides.  64x64x16 tiles, 4x4 per thread.
struct GemmP {
    const float* a; int64_t a_sm, a_sk;
    const float* b; int64_t b_sn, b_sk;
    float* c; int64_t c_sm, c_sn;
    const float* bias;
    int act, accumulate, M, N, K;
};

__global__ __launch_bounds__(NT) void gemm_f32_kernel(GemmP p) {
    __shared__ float As[16][64 + 4];
    __shared__ float Bs[16][64 + 4];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += 16) {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int e = tid + l * NT;      // 0..1023 -> (row 0..63, k 0..15)
            const int kk = e & 15, rr = e >> 4;
            const int k = k0 + kk;
            float av = 0.f, bv = 0.f;
            if (k < p.K) {
                if (m0 + rr < p.M) av = p.a[(int64_t)(m0 + rr) * p.a_sm + (int64_t)k * p.a_sk];
                if (n0 + rr < p.N) bv = p.b[(int64_t)(n0 + rr) * p.b_sn + (int64_t)k * p.b_sk];
            }
            As[kk][rr] = av;
            Bs[kk][rr] = bv;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a4[4], b4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a4[i] = As[kk][ty * 4 + i]; b4[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], b4[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mm = m0 + ty * 4 + i;
        if (mm >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nn = n0 + tx * 4 + j;
            if (nn >= p.N) continue;
            float v = acc[i][j];
            if (p.bias) v += p.bias[nn];
            v = apply_act(v, p.act);
            float* dst = p.c + (int64_t)mm * p.c_sm + (int64_t)nn * p.c_sn;
            if (p.accumulate) v += *dst;
            *dst = v;
        }
    }
}

// The same product on the matrix cores when both operands are K-contiguous (nn.Linear forward: x [M,K] row-major, weight [N,K]) and
// 16-byte aligned: exact fp32 (v_mfma_f32_16x16x4_f32 multiplies and accumulates in fp32), no LDS and no barrier.  A lane (row li,
// K group lg) loads 4 consecutive K values of its row per 16-deep K chunk (one 16-byte load per operand tile) and feeds element s of
// the vector to MFMA step s: over the four lane groups and four steps a chunk's 16 K values are each used once, in the same
// position for A and B.  A wave owns 16 x 32 outputs, a workgroup (2 x 2 waves) 32 x 64: the 64 x 64 VALU tiles above gave the joint
// FC / LSTM-gate products of the policy net (M = 360, N = 2048, K = 2048..2560) 192 workgroups of serial 16-deep K steps with two
// barriers each -- under one wave per SIMD, latency-bound at 184 us per launch (rocprofv3, round 4).
__global__ __launch_bounds__(NT) void gemm_f32_mfma_kernel(GemmP p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int m0 = blockIdx.y * 32 + (wave >> 1) * 16, n0 = blockIdx.x * 64 + (wave & 1) * 32;
    if (m0 >= p.M || n0 >= p.N) return;                                    // (wave-uniform; no barrier in this kernel)
    const float* ar = p.a + (int64_t)min(m0 + li, p.M - 1) * p.a_sm + 4 * lg;
    const float* br0 = p.b + (int64_t)min(n0 + li, p.N - 1) * p.b_sn + 4 * lg;
    const float* br1 = p.b + (int64_t)min(n0 + 16 + li, p.N - 1) * p.b_sn + 4 * lg;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int kfull = p.K & ~63;
    int k0 = 0;
    if (kfull) {
        // 64-deep steps with the 12 loads of step i + 1 requested under the MFMAs of step i.  Measured (tools/bench_gemm.py, M = 360,
        // N = 2048, K = 2560): 230 us (VALU tiles) -> 110 us, 34 TFLOP/s; the look-ahead itself changed nothing (113 us), so the bound is
        // not the L2 round trip: a fragment load touches 16 ROWS 8-10 KB apart (one 64-byte piece each), i.e. a few L2 channels per
        // instruction -- the LDS-staged form (rows read in long contiguous runs; rocBLAS needs 38 us here) is what would lift it.
        f32x4 a[4], b0[4], b1[4], na[4], nb0[4], nb1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            a[c] = *reinterpret_cast<const f32x4*>(ar + 16 * c);
            b0[c] = *reinterpret_cast<const f32x4*>(br0 + 16 * c);
            b1[c] = *reinterpret_cast<const f32x4*>(br1 + 16 * c);
        }
        for (; k0 < kfull; k0 += 64) {
            const int kn = k0 + 64 < kfull ? k0 + 64 : k0;             // (the last step re-requests its own chunk: unconditional loads)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                na[c] = *reinterpret_cast<const f32x4*>(ar + kn + 16 * c);
                nb0[c] = *reinterpret_cast<const f32x4*>(br0 + kn + 16 * c);
                nb1[c] = *reinterpret_cast<const f32x4*>(br1 + kn + 16 * c);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][s_], b0[c][s_], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][s_], b1[c][s_], acc1, 0, 0, 0);
                }
#pragma unroll
            for (int c = 0; c < 4; ++c) { a[c] = na[c]; b0[c] = nb0[c]; b1[c] = nb1[c]; }
        }
    }
    for (; k0 < p.K; k0 += 16) {                                           // K tail (K % 4 == 0: a lane's 4 values are all in or all out)
        const bool in = k0 + 4 * lg < p.K;
        const int kc = in ? k0 : 0;
        f32x4 a = *reinterpret_cast<const f32x4*>(ar + kc), b0 = *reinterpret_cast<const f32x4*>(br0 + kc), b1 = *reinterpret_cast<const f32x4*>(br1 + kc);
        if (!in) { a = f32x4{0.f, 0.f, 0.f, 0.f}; b0 = a; b1 = a; }
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s_], b0[s_], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s_], b1[s_], acc1, 0, 0, 0);
        }
    }
    // D fragment: lane (li, lg) holds rows 4 lg .. 4 lg + 3 of column li
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int nn = n0 + 16 * t + li;
        if (nn >= p.N) continue;
        const f32x4 acc = t ? acc1 : acc0;
        const float bv = p.bias ? p.bias[nn] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mm = m0 + 4 * lg + r;
            if (mm >= p.M) continue;
            float v = apply_act(acc[r] + bv, p.act);
            float* dst = p.c + (int64_t)mm * p.c_sm + (int64_t)nn * p.c_sn;
            if (p.accumulate) v += *dst;
            *dst = v;
        }
    }
}

static int dw_blocks(size_t P, int C, size_t* ppb_out) {
    const int rows = NT / (C / 8) > 0 ? NT / (C / 8) : 1;
    size_t ppb = (size_t)rows * 8;
    size_t nblk = (P + ppb - 1) / ppb;
    if (nblk > 4096) { ppb = ((P + 4095) / 4096 + rows - 1) / rows * rows; nblk = (P + ppb - 1) / ppb; }
    *ppb_out = ppb;
    return (int)nblk;
}

}  // namespace

// Grid of the column-strip walker (dwconv_fwd_kernel): rows walked per thread, the workgroup's channel-chunk tile and strip count.
// cw = a divisor of C / 4 (>= 16 lanes = one 128-byte line per strip when C allows) that leaves the fewest idle threads.
static unsigned dw_walk_grid(DwP& p, int segw, int groups) {
    p.seglen = segw;
    p.nseg = (p.OW + segw - 1) / segw;
    // rows walked per thread: long walks amortise the 2-row window prologue and the statistics epilogue (a workgroup of 3-row walks
    // spends as long publishing its sums as computing; the late 16^2 / 8^2 layers ran at half their no-statistics rate), whole image
    // columns where OH <= 24, equal row blocks otherwise; split further only while the grid is below one workgroup per CU
    static const long min_blocks = getenv("ADAMML_DW_MIN_BLOCKS") ? atol(getenv("ADAMML_DW_MIN_BLOCKS")) : 256;     // A/B aids
    static const int max_rows = getenv("ADAMML_DW_MAX_ROWS") ? atoi(getenv("ADAMML_DW_MAX_ROWS")) : 24;
    p.nrb = ceil_div(p.OH, max_rows);
    while (ceil_div(p.OH, p.nrb) > 3 && (long)groups * p.N * p.nrb * p.nseg * (p.C / 4) < min_blocks * NT) ++p.nrb;
    p.rows_per_thread = ceil_div(p.OH, p.nrb);
    p.nrb = ceil_div(p.OH, p.rows_per_thread);
    const int nchunk = p.C / 4;
    static const int cw_min = getenv("ADAMML_DW_CW_MIN") ? atoi(getenv("ADAMML_DW_CW_MIN")) : 16;         // A/B aid (1024: whole channel rows)
    int best = 0, best_active = 0;
    for (int pass = 0; pass < 2 && !best; ++pass)           // pass 0: divisors in [cw_min, 64], fewest idle threads; pass 1: the largest divisor <= 64
        for (int cw = pass ? 1 : cw_min; cw <= 64 && cw <= nchunk; ++cw) {
            if (nchunk % cw) continue;
            const int act = pass ? cw : cw * (NT / cw);
            if (act > best_active || (act == best_active && cw > best)) { best = cw; best_active = act; }
        }
    if (cw_min >= 1024 && nchunk <= NT) best = nchunk;
    p.cw = best; p.tpb = NT / best > 0 ? NT / best : 1; p.nct = nchunk / best;
    const long strips = (long)p.N * p.nrb * p.nseg;
    return (unsigned)(p.nct * ((strips + p.tpb - 1) / p.tpb));
}

static int check_dw(const adamml_conv_desc_t* d, const char* name) {
    if (!d) return adamml_set_error(ADAMML_EINVAL, "%s: null desc", name);
    if (d->KH != 3 || d->KW != 3 || d->Cin != d->Cout || d->Cin % 8 || d->Cin > MAXC || (d->stride != 1 && d->stride != 2))
        return adamml_set_error(ADAMML_EUNSUPPORTED, "%s: depthwise kernel supports 3x3, stride 1/2, C%%8==0 (C=%d k=%d s=%d)", name,
                                d->Cin, d->KH, d->stride);
    return ADAMML_OK;
}

extern "C" int adamml_dwconv_fwd(const adamml_conv_desc_t* d, const void* x, const float* w, const float* in_scale,
                                 const float* in_shift, void* y, double* stats, hipStream_t stream) {
    int rc = check_dw(d, "dwconv_fwd");
    if (rc) return rc;
    DwP p;
    p.bn_z = nullptr; p.bn_vec = nullptr; p.bn_act = 0; p.x1 = nullptr; p.x1_g = p.x1_n = 0;
    p.x = (const bf16_t*)x; p.w = w; p.in_scale = in_scale; p.in_shift = in_shift; p.y = (bf16_t*)y; p.stats = stats;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->Cin; p.OH = d->OH; p.OW = d->OW; p.stride = d->stride; p.pad = d->pad;
    p.act = d->act; p.accumulate = 0;
    p.P = (size_t)d->N * d->OH * d->OW;
    if (!p.P) return ADAMML_OK;
    const int groups = d->groups < 1 ? 1 : d->groups;
    p.gx = (size_t)d->N * d->H * d->W * d->Cin; p.gy = p.P * d->Cin; p.in_gstride = d->in_gstride;
    p.ppb = 0;
    p.flip = 0;
    const unsigned nblk = dw_walk_grid(p, d->stride == 1 ? 4 : 2, groups);          // segw == SEGW of dwconv_fwd_kernel<S>
    if (d->stride == 1) {
        if (d->OW % 4 == 0) hipLaunchKernelGGL((dwconv_fwd_kernel<1, false, false, true>), dim3(nblk, groups), dim3(NT), 0, stream, p);
        else hipLaunchKernelGGL(dwconv_fwd_kernel<1>, dim3(nblk, groups), dim3(NT), 0, stream, p);
    } else {
        if (d->OW % 2 == 0) hipLaunchKernelGGL((dwconv_fwd_kernel<2, false, false, true>), dim3(nblk, groups), dim3(NT), 0, stream, p);
        else hipLaunchKernelGGL(dwconv_fwd_kernel<2>, dim3(nblk, groups), dim3(NT), 0, stream, p);
    }
    return adamml_check_launch("dwconv_fwd");
}

static int dw_bwd_data_launch(const adamml_conv_desc_t* d, const void* dz, const float* w, void* dx, int accumulate, const void* bn_z,
                              const float* bn_vec, int bn_act, double* sums, hipStream_t stream) {
    int rc = check_dw(d, "dwconv_bwd_data");
    if (rc) return rc;
    DwP p;
    p.x1 = nullptr; p.x1_g = p.x1_n = 0;
    p.x = (const bf16_t*)dz; p.w = w; p.in_scale = nullptr; p.in_shift = nullptr; p.y = (bf16_t*)dx; p.stats = sums;
    p.bn_z = (const bf16_t*)bn_z; p.bn_vec = bn_vec; p.bn_act = bn_act;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->Cin; p.OH = d->OH; p.OW = d->OW; p.stride = d->stride; p.pad = d->pad;
    p.act = 0; p.accumulate = accumulate; p.nseg = 1; p.seglen = 0; p.flip = 0;
    p.P = (size_t)d->N * d->H * d->W;
    if (!p.P) return ADAMML_OK;
    const int groups = d->groups < 1 ? 1 : d->groups;
    p.gx = (size_t)d->N * d->OH * d->OW * d->Cin; p.gy = p.P * d->Cin; p.in_gstride = 0;
    if (d->stride == 1 && !accumulate && d->pad == 1) {
        // stride 1: the data gradient IS the forward walk over dz with the taps reversed (no transform, no statistics)
        p.H = d->OH; p.W = d->OW; p.OH = d->H; p.OW = d->W; p.flip = 1;
        const unsigned nblk = dw_walk_grid(p, 4, groups);
        const bool wf = p.OW % 4 == 0;                  // (p.OW: the width of dx here)
        if (bn_z) {
            // (no WF instance of the BatchNorm-fused form: at 256 registers it spilled 4 and measured the same)
            hipLaunchKernelGGL((dwconv_fwd_kernel<1, true>), dim3(nblk, groups), dim3(NT), 0, stream, p);
        } else {
            if (wf) hipLaunchKernelGGL((dwconv_fwd_kernel<1, false, false, true>), dim3(nblk, groups), dim3(NT), 0, stream, p);
            else hipLaunchKernelGGL(dwconv_fwd_kernel<1>, dim3(nblk, groups), dim3(NT), 0, stream, p);
        }
        return adamml_check_launch("dwconv_bwd_data");
    }
    static const bool quads = !(getenv("ADAMML_DW_S2_QUADS") && atoi(getenv("ADAMML_DW_S2_QUADS")) == 0);       // A/B aid
    if (d->stride == 2 && d->pad == 1 && (quads || bn_z)) {
        p.P = (size_t)d->N * ((d->H + 1) / 2) * ((d->W + 1) / 2);
        int nblk = dw_blocks(p.P, p.C, &p.ppb);           // (16 / 32 / 64 passes per thread instead of 8: no gain, 4.2-4.7 TB/s incl. the z read)
        if (bn_z) hipLaunchKernelGGL(dwconv_bwd_data_s2_kernel<true>, dim3(nblk, groups), dim3(NT), 0, stream, p);
        else hipLaunchKernelGGL(dwconv_bwd_data_s2_kernel<false>, dim3(nblk, groups), dim3(NT), 0, stream, p);
        return adamml_check_launch("dwconv_bwd_data");
    }
    if (bn_z) return adamml_set_error(ADAMML_EUNSUPPORTED, "dwconv_bwd_data_bn: needs pad 1 and stride 1 (no accumulate) or stride 2");
    int nblk = dw_blocks(p.P, p.C, &p.ppb);
    hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3(nblk, groups), dim3(NT), 0, stream, p);
    return adamml_check_launch("dwconv_bwd_data");
}

extern "C" int adamml_dwconv_bwd_data(const adamml_conv_desc_t* d, const void* dz, const float* w, void* dx, int accumulate,
                                      hipStream_t stream) {
    return dw_bwd_data_launch(d, dz, w, dx, accumulate, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int adamml_dwconv_bwd_data_bn_supported(const adamml_conv_desc_t* d) {
    return d && d->KH == 3 && d->KW == 3 && d->Cin == d->Cout && d->Cin % 8 == 0 && d->Cin <= MAXC && d->pad == 1 &&
           (d->stride == 1 || d->stride == 2) ? 1 : 0;
}

extern "C" int adamml_dwconv_bwd_data_bn(const adamml_conv_desc_t* d, const void* dz, const float* w, void* dx, const void* z_in,
                                         const float* bn_vec, int act, double* sums, hipStream_t stream) {
    if (!z_in || !bn_vec || !sums) return adamml_set_error(ADAMML_EINVAL, "dwconv_bwd_data_bn: null BatchNorm epilogue operand");
    if (!adamml_dwconv_bwd_data_bn_supported(d))
        return adamml_set_error(ADAMML_EUNSUPPORTED, "dwconv_bwd_data_bn: 3x3 depthwise, pad 1, stride 1 / 2, C %% 8 == 0");
    return dw_bwd_data_launch(d, dz, w, dx, 0, z_in, bn_vec, act, sums, stream);
}

static int dw_wgrad_blocks(const adamml_conv_desc_t* d, int* rows_per_thread, int* nseg, int* nrb) {
    const long nchunk = d->Cin / 4;
    const int segw = d->stride == 1 ? 4 : 2;           // == SEGW of dwconv_bwd_weight_kernel<S>
    *nseg = (d->OW + segw - 1) / segw;
    const int groups = d->groups < 1 ? 1 : d->groups;
    // equal row blocks of at most max_rows rows, split further only while the grid is short of threads (as dw_walk_grid)
    static const long min_blocks = getenv("ADAMML_DWW_MIN_BLOCKS") ? atol(getenv("ADAMML_DWW_MIN_BLOCKS")) : 256;      // A/B aids
    static const int max_rows = getenv("ADAMML_DWW_MAX_ROWS") ? atoi(getenv("ADAMML_DWW_MAX_ROWS")) : 24;
    int nb_rows = ceil_div(d->OH, max_rows);
    while (ceil_div(d->OH, nb_rows) > 3 && (long)groups * d->N * nb_rows * *nseg * nchunk < min_blocks * NT) ++nb_rows;
    const int rpt = ceil_div(d->OH, nb_rows);
    *rows_per_thread = rpt;
    *nrb = ceil_div(d->OH, rpt);
    const long threads = (long)d->N * *nrb * *nseg * nchunk;
    long nb = (threads + NT - 1) / NT;
    // NT * nblk must be a multiple of nchunk (a thread keeps its channel group across tasks): 45 | nblk covers every
    // C/4 in {8,12,24,36,48,96,144,240}; otherwise fall back to a multiple of nchunk
    long cap = 2160 / groups / 45 * 45 > 0 ? 2160 / groups / 45 * 45 : 45;
    // every workgroup publishes one [9][C] fp32 partial (36 C bytes) that a second launch sums: keep that traffic well below the
    // ~4 C bytes per pixel a workgroup reads -- at most one workgroup per 144 output pixels of a group.  (The small late layers were
    // split over the full cap: 2025 partials of 35 KB for a 90 MB problem, a 0.18 ms floor per layer.)
    const long by_work = ((long)d->N * d->OH * d->OW / 144 + 44) / 45 * 45;
    if (by_work < cap) cap = by_work > 45 ? by_work : 45;
    int nblk = nb >= cap ? (int)cap : (int)((nb + 44) / 45 * 45);
    if ((NT * (long)nblk) % nchunk != 0) nblk = (int)((nblk + nchunk - 1) / nchunk * nchunk);
    return nblk;
}

extern "C" size_t adamml_dwconv_bwd_weight_workspace(const adamml_conv_desc_t* d) {
    if (!d || d->Cin % 8) return 0;
    int a, b, c;
    return (size_t)(d->groups < 1 ? 1 : d->groups) * dw_wgrad_blocks(d, &a, &b, &c) * 9 * d->Cin * sizeof(float);
}

extern "C" int adamml_dwconv_bwd_weight(const adamml_conv_desc_t* d, const void* dz, const void* x, const float* in_scale,
                                        const float* in_shift, float* dw, void* workspace, size_t workspace_bytes,
                                        hipStream_t stream) {
    int rc = check_dw(d, "dwconv_bwd_weight");
    if (rc) return rc;
    DwWP p;
    p.x1 = nullptr; p.x1_g = p.x1_n = 0;
    p.dz = (const bf16_t*)dz; p.x = (const bf16_t*)x; p.in_scale = in_scale; p.in_shift = in_shift; p.dw = dw;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->Cin; p.OH = d->OH; p.OW = d->OW; p.stride = d->stride; p.pad = d->pad; p.act = d->act;
    const size_t P = (size_t)d->N * d->OH * d->OW;
    if (!P) return ADAMML_OK;
    const int nblk = dw_wgrad_blocks(d, &p.rows_per_thread, &p.nseg, &p.nrb);
    const int groups = d->groups < 1 ? 1 : d->groups;
    p.gdz = P * d->Cin; p.gx = (size_t)d->N * d->H * d->W * d->Cin; p.in_gstride = d->in_gstride;
    p.ws = (workspace && workspace_bytes >= (size_t)groups * nblk * 9 * p.C * sizeof(float)) ? (float*)workspace : nullptr;
    if (d->stride == 1) hipLaunchKernelGGL(dwconv_bwd_weight_kernel<1>, dim3(nblk, groups), dim3(NT), 9 * p.C * sizeof(float), stream, p);
    else hipLaunchKernelGGL(dwconv_bwd_weight_kernel<2>, dim3(nblk, groups), dim3(NT), 9 * p.C * sizeof(float), stream, p);
    if (p.ws) {
        rc = adamml_check_launch("dwconv_bwd_weight");
        if (rc) return rc;
        return adamml_launch_split_reduce(p.ws, dw, (size_t)9 * p.C, groups * nblk, stream);
    }
    return adamml_check_launch("dwconv_bwd_weight");
}

// ---- 3x3 / stride-2 stem of a ONE-channel fp32 image (spectrogram): the depthwise walkers with the pixel broadcast over the output
// channels.  d: N images per group of H x W, Cin ignored (1), Cout = C (multiple of 8, <= 64), KH = KW = 3, stride 2, pad 1.
static int check_stem1(const adamml_conv_desc_t* d, const char* name) {
    if (!d) return adamml_set_error(ADAMML_EINVAL, "%s: null desc", name);
    if (d->KH != 3 || d->KW != 3 || d->stride != 2 || d->pad != 1 || d->Cout % 8 || d->Cout > 64 || d->Cout < 8)
        return adamml_set_error(ADAMML_EUNSUPPORTED, "%s: 3x3 / stride 2 / pad 1 stem with 8..64 output channels (Cout=%d k=%d s=%d)", name, d->Cout,
                                d->KH, d->stride);
    return ADAMML_OK;
}

extern "C" int adamml_conv_stem1_supported(const adamml_conv_desc_t* d) {
    return d && d->KH == 3 && d->KW == 3 && d->stride == 2 && d->pad == 1 && d->Cout % 8 == 0 && d->Cout >= 8 && d->Cout <= 64 ? 1 : 0;
}

extern "C" int adamml_conv_stem1_fwd(const adamml_conv_desc_t* d, const float* x, size_t image_stride, size_t group_stride, const float* w,
                                     void* y, double* stats, hipStream_t stream) {
    int rc = check_stem1(d, "conv_stem1_fwd");
    if (rc) return rc;
    if (!x || !w || !y) return adamml_set_error(ADAMML_EINVAL, "conv_stem1_fwd: null argument");
    DwP p;
    p.bn_z = nullptr; p.bn_vec = nullptr; p.bn_act = 0;
    p.x = nullptr; p.x1 = x; p.x1_g = group_stride; p.x1_n = image_stride;
    p.w = w; p.in_scale = nullptr; p.in_shift = nullptr; p.y = (bf16_t*)y; p.stats = stats;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->Cout; p.OH = d->OH; p.OW = d->OW; p.stride = 2; p.pad = 1;
    p.act = 0; p.accumulate = 0;
    p.P = (size_t)d->N * d->OH * d->OW;
    if (!p.P) return ADAMML_OK;
    const int groups = d->groups < 1 ? 1 : d->groups;
    p.gx = 0; p.gy = p.P * d->Cout; p.in_gstride = 0;
    p.ppb = 0;
    p.flip = 0;
    const unsigned nblk = dw_walk_grid(p, 2, groups);
    hipLaunchKernelGGL((dwconv_fwd_kernel<2, false, true>), dim3(nblk, groups), dim3(NT), 0, stream, p);      // (no WF instance: 136 instead of 126 registers = one wave per SIMD less)
    return adamml_check_launch("conv_stem1_fwd");
}

static adamml_conv_desc_t stem1_as_dw(const adamml_conv_desc_t* d) {
    adamml_conv_desc_t e = *d;
    e.Cin = d->Cout;
    return e;
}

extern "C" size_t adamml_conv_stem1_bwd_weight_workspace(const adamml_conv_desc_t* d) {
    if (!adamml_conv_stem1_supported(d)) return 0;
    const adamml_conv_desc_t e = stem1_as_dw(d);
    return adamml_dwconv_bwd_weight_workspace(&e);
}

extern "C" int adamml_conv_stem1_bwd_weight(const adamml_conv_desc_t* d, const void* dz, const float* x, size_t image_stride, size_t group_stride,
                                            float* dw, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    int rc = check_stem1(d, "conv_stem1_bwd_weight");
    if (rc) return rc;
    if (!dz || !x || !dw) return adamml_set_error(ADAMML_EINVAL, "conv_stem1_bwd_weight: null argument");
    const adamml_conv_desc_t e = stem1_as_dw(d);
    DwWP p;
    p.dz = (const bf16_t*)dz; p.x = nullptr; p.x1 = x; p.x1_g = group_stride; p.x1_n = image_stride;
    p.in_scale = nullptr; p.in_shift = nullptr; p.dw = dw;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->Cout; p.OH = d->OH; p.OW = d->OW; p.stride = 2; p.pad = 1; p.act = 0;
    const size_t P = (size_t)d->N * d->OH * d->OW;
    if (!P) return ADAMML_OK;
    const int nblk = dw_wgrad_blocks(&e, &p.rows_per_thread, &p.nseg, &p.nrb);
    const int groups = d->groups < 1 ? 1 : d->groups;
    p.gdz = P * d->Cout; p.gx = 0; p.in_gstride = 0;
    p.ws = (workspace && workspace_bytes >= (size_t)groups * nblk * 9 * p.C * sizeof(float)) ? (float*)workspace : nullptr;
    hipLaunchKernelGGL((dwconv_bwd_weight_kernel<2, true>), dim3(nblk, groups), dim3(NT), 9 * p.C * sizeof(float), stream, p);
    if (p.ws) {
        rc = adamml_check_launch("conv_stem1_bwd_weight");
        if (rc) return rc;
        return adamml_launch_split_reduce(p.ws, dw, (size_t)9 * p.C, groups * nblk, stream);
    }
    return adamml_check_launch("conv_stem1_bwd_weight");
}

extern "C" int adamml_gemm_f32(const float* a, int64_t a_sm, int64_t a_sk, const float* b, int64_t b_sn, int64_t b_sk, float* c,
                               int64_t c_sm, int64_t c_sn, const float* bias, int act, int accumulate, int M, int N, int K,
                               hipStream_t stream) {
    if (!a || !b || !c) return adamml_set_error(ADAMML_EINVAL, "gemm_f32: null argument");
    if (M <= 0 || N <= 0) return ADAMML_OK;
    GemmP p{a, a_sm, a_sk, b, b_sn, b_sk, c, c_sm, c_sn, bias, act, accumulate, M, N, K};
    static const int use_mfma = getenv("ADAMML_GEMM_MFMA") ? atoi(getenv("ADAMML_GEMM_MFMA")) : 1;          // A/B aid
    if (use_mfma && a_sk == 1 && b_sk == 1 && K >= 16 && (K & 3) == 0 && (a_sm & 3) == 0 && (b_sn & 3) == 0 &&
        ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0) {
        hipLaunchKernelGGL(gemm_f32_mfma_kernel, dim3(ceil_div(N, 64), ceil_div(M, 32)), dim3(NT), 0, stream, p);
        return adamml_check_launch("gemm_f32 (mfma)");
    }
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(ceil_div(N, 64), ceil_div(M, 64)), dim3(NT), 0, stream, p);
    return adamml_check_launch("gemm_f32");
}
