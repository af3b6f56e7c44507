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

// Per-channel constants live in LDS ([16][C] fp32: the nine taps reversed, A / B / C, the expansion's scale / shift / mean / invstd) and
// are re-read where they are used -- 16 ds_read_b128 per row step against ~800 VALU lane-operations -- instead of pinning 64 registers;
// the dz window is kept as packed bf16 (it IS bf16: the value the per-layer kernels exchange) and unpacked one window row at a time.
template <int S, int SEGW>                  // SEGW = pixels (columns) per thread
__global__ __launch_bounds__(NT, 2) void dwconv_bwd_fused_kernel(DwBP p) {
    static_assert(S == 1, "stride-1 walker");
    constexpr int NCOL = SEGW + 2;          // dz columns feeding them
    extern __shared__ __attribute__((aligned(16))) float dsm[];          // [16][C] constants during the walk; then [9][C] weight-gradient fold, [2][C] statistics fold
    {
        const size_t g = blockIdx.y;
        p.g += g * p.gz; p.z += g * p.gz; p.x += g * p.gx; p.dx += g * p.gx;
        p.aff += g * 3 * p.C; p.xvec += g * 4 * p.C;
        p.stats += g * ADAMML_STAT_SLOTS * 2 * p.C;
    }
    const int C = p.C;
    const int nchunk = C >> 2;
    for (int i = threadIdx.x; i < 16 * C; i += NT) {
        const int r = i / C, cc2 = i - r * C;
        dsm[i] = r < 9 ? p.w[(size_t)(8 - r) * C + cc2] : r < 12 ? p.aff[(size_t)(r - 9) * C + cc2] : p.xvec[(size_t)(r - 12) * C + cc2];
    }
    __syncthreads();
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
    __syncthreads();                        // (the constants are dead: their LDS becomes the fold area)
    for (int i = threadIdx.x; i < 9 * C; i += NT) dsm[i] = 0.f;
    __syncthreads();
    // the threads that share a channel chunk (thread ids congruent modulo nchunk) add in turn: a fixed order (common.h: reproducible
    // reductions); the weight-gradient tile first, then (same LDS) the two statistics rows
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

// pixels per thread: 3 (256 registers, none spilled; 4 spills 78 of them, 2 re-reads every halo column)
int fused_segw() {
    static const int v = getenv("ADAMML_DWB_SEGW") ? atoi(getenv("ADAMML_DWB_SEGW")) : 3;          // A/B aid
    return v == 2 ? 2 : 3;
}

int fused_blocks(const adamml_conv_desc_t* d, int* rows_per_thread, int* nseg, int* nrb) {
    const long nchunk = d->Cin / 4;
    *nseg = ceil_div(d->W, fused_segw());
    const int groups = d->groups < 1 ? 1 : d->groups;
    static const long min_blocks = getenv("ADAMML_DWB_MIN_BLOCKS") ? atol(getenv("ADAMML_DWB_MIN_BLOCKS")) : 256;       // A/B aids
    static const int max_rows = getenv("ADAMML_DWB_MAX_ROWS") ? atoi(getenv("ADAMML_DWB_MAX_ROWS")) : 24;
    int nb_rows = ceil_div(d->H, max_rows);
    while (ceil_div(d->H, nb_rows) > 3 && (long)groups * d->N * nb_rows * *nseg * nchunk < min_blocks * NT) ++nb_rows;
    const int rpt = ceil_div(d->H, nb_rows);
    *rows_per_thread = rpt;
    *nrb = ceil_div(d->H, rpt);
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
    return on && d && d->KH == 3 && d->KW == 3 && d->Cin == d->Cout && d->Cin % 8 == 0 && d->Cin <= 960 && d->pad == 1 && d->stride == 1 &&
           d->OH == d->H && d->OW == d->W ? 1 : 0;
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
        return adamml_set_error(ADAMML_EUNSUPPORTED, "dwconv_bwd_fused: 3x3 depthwise, pad 1, stride 1, C %% 8 == 0");
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
    if (fused_segw() == 2) hipLaunchKernelGGL((dwconv_bwd_fused_kernel<1, 2>), dim3(nblk, groups), dim3(NT), 16 * p.C * sizeof(float), stream, p);
    else hipLaunchKernelGGL((dwconv_bwd_fused_kernel<1, 3>), dim3(nblk, groups), dim3(NT), 16 * p.C * sizeof(float), stream, p);
    int rc = adamml_check_launch("dwconv_bwd_fused");
    if (rc) return rc;
    return adamml_launch_split_reduce(p.ws, dw, (size_t)9 * p.C, groups * nblk, stream);
}
