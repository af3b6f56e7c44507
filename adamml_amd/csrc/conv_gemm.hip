// Implicit-GEMM convolution for gfx950: bf16 MFMA (v_mfma_f32_16x16x32_bf16), NHWC activations,
// weights packed [Cout][KH*KW*Cin].  One kernel serves
//   * forward conv (any KxK / stride / pad) with the producer's BatchNorm+activation applied while the
//     input patch is staged into LDS, and per-channel sum / sum-of-squares of the (bf16-rounded) output
//     accumulated in the epilogue (train-mode BN statistics, SURVEY.md section 7 hard part 1);
//   * data-gradient (same kernel on the flipped/transposed weight pack, `up` = forward stride).
// A second kernel computes the weight gradient with LDS transpose reads (ds_read_b64_tr_b16).
//
// Replaces the torch operators called at models/resnet.py:83-113,138,199-210,
// models/sound_mobilenet_v2.py:33-69 and models/policy_net.py:38-95 (nn.Conv2d + nn.BatchNorm2d + ReLU/ReLU6).
#include "common.h"
#include "../../include/adamml_hip.h"

namespace {

constexpr int BP = 128;      // pixels per block tile
constexpr int BK = 32;       // K step (one MFMA K)
constexpr int NTHREADS = 256;

struct ConvP {
    const bf16_t* x;
    const bf16_t* w;
    const float* in_scale;
    const float* in_shift;
    bf16_t* y;
    double* stats;
    int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, up, up_shift, act, accumulate;
    int P, K, cin_shift, n_ptiles, n_ctiles;
};

__device__ __forceinline__ int lds_off(int row, int chunk) {
    // 64-byte rows, 16-byte chunks, chunk ^= 2*((row>>3)&1): conflict-free ds_read_b128 for the
    // 16-lane service groups of gfx950 (MI355X_MICROARCH.md LDS table)
    return row * 64 + ((chunk ^ (((row >> 3) & 1) << 1)) << 4);
}

template <int BC, bool MULTITAP>
__global__ __launch_bounds__(NTHREADS) void conv_gemm_kernel(ConvP p) {
    constexpr int WCT = BC / 32;            // 16-wide cout tiles per wave
    constexpr int WROWS = BC / 64;          // weight rows staged per thread
    constexpr int TILE_BYTES = (BP + BC) * 64;
    constexpr int EPI_BYTES = BP * (BC * 2 + 16);
    constexpr int SMEM_BYTES = (2 * TILE_BYTES + 2 * BC * 4) > EPI_BYTES ? (2 * TILE_BYTES + 2 * BC * 4) : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wp = wave & 1;                // pixel half
    const int wc = wave >> 1;               // cout half

    // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (and its L2); give them consecutive
    // tiles so the cout tiles of one pixel tile hit the same L2 (cdna_hip_programming.md T1, bijective form)
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ctile = bid % p.n_ctiles;
    const int ptile = bid / p.n_ctiles;
    const int c0 = ctile * BC;
    const int p0 = ptile * BP;

    // ---- per-thread staging coordinates -------------------------------------------------------
    const int chunk = tid & 3;
    const int row_a = tid >> 2;             // 0..63 (+64 for second row)
    int a_n[2], a_h0[2], a_w0[2];
    bool a_ok[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        int pp = p0 + row_a + r * 64;
        a_ok[r] = pp < p.P;
        int ppc = a_ok[r] ? pp : 0;
        int n = ppc / (p.OH * p.OW);
        int rem = ppc - n * (p.OH * p.OW);
        int oh = rem / p.OW;
        int ow = rem - oh * p.OW;
        a_n[r] = n;
        a_h0[r] = oh * p.stride - p.pad;
        a_w0[r] = ow * p.stride - p.pad;
    }
    const bf16_t* wrow[WROWS];
    bool w_ok[WROWS];
#pragma unroll
    for (int r = 0; r < WROWS; ++r) {
        int co = c0 + row_a + r * 64;
        w_ok[r] = co < p.Cout;
        wrow[r] = p.w + (size_t)(w_ok[r] ? co : 0) * p.K;
    }

    bf16x8 ra[2], rw[WROWS];
    int rci = 0;
    bool rav[2];

    auto issue_loads = [&](int kt) {
        const int k = kt * BK + chunk * 8;
        const bool kok = k < p.K;
        int kh = 0, kw = 0, ci = k;
        if (MULTITAP) {
            int tap = k >> p.cin_shift;
            ci = k & (p.Cin - 1);
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        }
        rci = ci;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int ih = a_h0[r] + kh, iw = a_w0[r] + kw;
            bool ok = a_ok[r] && kok && ih >= 0 && iw >= 0;
            if (p.up > 1) {
                ok = ok && ((ih | iw) & (p.up - 1)) == 0;
                ih >>= p.up_shift;
                iw >>= p.up_shift;
            }
            ok = ok && ih < p.H && iw < p.W;
            rav[r] = ok;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) v = *reinterpret_cast<const bf16x8*>(p.x + ((size_t)(a_n[r] * p.H + ih) * p.W + iw) * p.Cin + ci);
            ra[r] = v;
        }
#pragma unroll
        for (int r = 0; r < WROWS; ++r) {
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (w_ok[r] && kok) v = *reinterpret_cast<const bf16x8*>(wrow[r] + k);
            rw[r] = v;
        }
    };
    auto store_tile = [&](int buf) {
        char* base = smem + buf * TILE_BYTES;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            bf16x8 v = ra[r];
            if (p.in_scale && rav[r]) v = f32_to_bf8(transform8(v, p.in_scale, p.in_shift, rci, p.act));
            *reinterpret_cast<bf16x8*>(base + lds_off(row_a + r * 64, chunk)) = v;
        }
#pragma unroll
        for (int r = 0; r < WROWS; ++r)
            *reinterpret_cast<bf16x8*>(base + BP * 64 + lds_off(row_a + r * 64, chunk)) = rw[r];
    };

    f32x4 acc[WCT][4];
#pragma unroll
    for (int i = 0; i < WCT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    issue_loads(0);
    store_tile(0);
    __syncthreads();

    const int li = lane & 15, lg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) issue_loads(kt + 1);
        const char* base = smem + buf * TILE_BYTES;
        bf16x8 fa[4], fw[WCT];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            fa[t] = *reinterpret_cast<const bf16x8*>(base + lds_off(wp * 64 + t * 16 + li, lg));
#pragma unroll
        for (int t = 0; t < WCT; ++t)
            fw[t] = *reinterpret_cast<const bf16x8*>(base + BP * 64 + lds_off(wc * (BC / 2) + t * 16 + li, lg));
#pragma unroll
        for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
                acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ct], fa[pt], acc[ct][pt], 0, 0, 0);
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: stage the bf16 tile through LDS, store 16 B per lane fully coalesced, and accumulate the
    //      per-channel sum / sum-of-squares of the stored (rounded) values on the way out -------------------
    constexpr int CROW = BC * 2 + 16;                  // LDS row stride (bytes): +16 B skews the banks
    static_assert(BP * CROW <= SMEM_BYTES, "epilogue tile must fit the staging buffers");
    // (the last K step ended with __syncthreads(): every wave is done reading the operand tiles)
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int ct = 0; ct < WCT; ++ct) {
            const int prow = wp * 64 + pt * 16 + li;
            const int ccol = wc * (BC / 2) + ct * 16 + lg * 4;
            *reinterpret_cast<bf16x4*>(smem + prow * CROW + ccol * 2) = f32_to_bf4(acc[ct][pt]);
        }
    __syncthreads();
    constexpr int CPR = BC / 8;                        // 16-byte chunks per tile row
    constexpr int RSTEP = NTHREADS / CPR;              // rows covered per pass
    const int ech = tid % CPR, erow0 = tid / CPR;
    const int eco = c0 + ech * 8;
    f32x8 esum, esq;
#pragma unroll
    for (int i = 0; i < 8; ++i) esum[i] = esq[i] = 0.f;
    if (eco < p.Cout) {
#pragma unroll
        for (int r = erow0; r < BP; r += RSTEP) {
            const int pp = p0 + r;
            if (pp >= p.P) break;
            bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + r * CROW + ech * 16);
            bf16_t* dst = p.y + (size_t)pp * p.Cout + eco;
            f32x8 f = bf8_to_f32(v);
            if (p.accumulate) {
                f += bf8_to_f32(*reinterpret_cast<const bf16x8*>(dst));
                v = f32_to_bf8(f);
                f = bf8_to_f32(v);
            }
            *reinterpret_cast<bf16x8*>(dst) = v;
            esum += f;
            esq += f * f;
        }
    }
    if (p.stats) {
        __syncthreads();                               // tile reads done: reuse the front of smem for the channel sums
        float* cs = reinterpret_cast<float*>(smem);
        for (int i = tid; i < 2 * BC; i += NTHREADS) cs[i] = 0.f;
        __syncthreads();
        // lanes l, l+CPR, l+2*CPR.. of a wave hold the same channel chunk: fold them first
#pragma unroll
        for (int off = CPR; off < 64; off <<= 1)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                esum[i] += __shfl_xor(esum[i], off, 64);
                esq[i] += __shfl_xor(esq[i], off, 64);
            }
        if (lane < CPR && eco < p.Cout) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                atomicAdd(&cs[ech * 8 + i], esum[i]);
                atomicAdd(&cs[BC + ech * 8 + i], esq[i]);
            }
        }
        __syncthreads();
        double* slot = p.stats + (size_t)(blockIdx.x & (ADAMML_STAT_SLOTS - 1)) * 2 * p.Cout;
        for (int i = tid; i < BC; i += NTHREADS) {
            if (c0 + i < p.Cout) {
                atomicAdd(&slot[c0 + i], (double)cs[i]);
                atomicAdd(&slot[p.Cout + c0 + i], (double)cs[BC + i]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient:  dW[co][ci][kh][kw] += sum_p dz[p][co] * a[p@(kh,kw)][ci]   (fp32 atomics, split over pixels)
// Both operands are pixel-major in HBM (NHWC), i.e. K-major for this GEMM, so MFMA fragments are fetched
// from LDS with the hardware transpose read ds_read_b64_tr_b16.
struct WgradP {
    const bf16_t* dz;      // [N,OH,OW,Cout]
    const bf16_t* x;       // [N,H,W,Cin]
    const float* in_scale;
    const float* in_shift;
    float* dw;             // OIHW fp32, Cin_true input channels
    int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, act, cin_true;
    int P, pix_per_block, n_cotiles, n_tiles, cin_shift, NK;   // NK = KH*KW*Cin: flattened (tap, ci) GEMM-N extent
};

// LDS image of one K step: [32 pixels][CH channels] bf16, row-major, 8-byte units XOR-swizzled so that the
// ds_read_b64_tr_b16 of a 32-lane service group (rows {r..r+3} U {r+8..r+11}) touches 64 distinct banks.
template <int CH>
__device__ __forceinline__ int tr_swz(int row) {
    if (CH >= 128) return ((row & 3) | (((row >> 3) & 1) << 2)) << 2;       // 256-byte rows: all rows start at bank 0
    return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 2;               // 128-byte rows: parity picks the bank half
}

template <int BM, int BN>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_kernel(WgradP p) {
    constexpr int AROW = BM * 2;            // bytes per LDS row (one pixel)
    constexpr int BROW = BN * 2;
    constexpr int TILE_BYTES = 32 * (AROW + BROW);
    constexpr int MT = BM / 32, NT = BN / 32;
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    // block -> (tile, pixel split): the tiles of one pixel range run back-to-back on ONE XCD (block b lives on XCD
    // b % 8), so the dz / activation rows they all re-read come from that XCD's L2 instead of HBM
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = j % p.n_tiles;
    const int split = (j / p.n_tiles) * 8 + xcd;
    const int co0 = (tile % p.n_cotiles) * BM;
    const int n0 = (tile / p.n_cotiles) * BN;           // offset in the flattened (tap, ci) axis
    const int ps = split * p.pix_per_block;
    const int pe = min(p.P, ps + p.pix_per_block);
    if (ps >= pe) return;

    constexpr int ACH = BM / 8, BCH = BN / 8;                 // 16-byte chunks per row
    constexpr int AL = (32 * ACH) / NTHREADS, BL = (32 * BCH) / NTHREADS;
    bf16x8 ra[AL], rb[BL];
    bool rbv[BL];
    int b_kh[BL], b_kw[BL], b_ci[BL], b_n[BL], b_oh[BL], b_ow[BL];
    bool b_ok[BL];
#pragma unroll
    for (int l = 0; l < BL; ++l) {
        int e = tid + l * NTHREADS;
        int row = e / BCH, ch = e - row * BCH;
        int n = n0 + ch * 8;
        b_ok[l] = n < p.NK;
        int tap = n >> p.cin_shift;
        b_ci[l] = n - (tap << p.cin_shift);
        b_kh[l] = tap / p.KW - p.pad;
        b_kw[l] = tap - (tap / p.KW) * p.KW - p.pad;
        int pp = ps + row;                                   // pixel of this chunk at K step 0; advanced by 32 per step
        b_n[l] = pp / (p.OH * p.OW);
        int rem = pp - b_n[l] * (p.OH * p.OW);
        b_oh[l] = rem / p.OW;
        b_ow[l] = rem - b_oh[l] * p.OW;
    }

    auto issue_loads = [&](int pbase) {
#pragma unroll
        for (int l = 0; l < AL; ++l) {
            int e = tid + l * NTHREADS;
            int row = e / ACH, ch = e - row * ACH;
            int pp = pbase + row, co = co0 + ch * 8;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (pp < pe && co < p.Cout) v = *reinterpret_cast<const bf16x8*>(p.dz + (size_t)pp * p.Cout + co);
            ra[l] = v;
        }
#pragma unroll
        for (int l = 0; l < BL; ++l) {
            int e = tid + l * NTHREADS;
            int row = e / BCH;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            int ih = b_oh[l] * p.stride + b_kh[l], iw = b_ow[l] * p.stride + b_kw[l];
            bool ok = (pbase + row < pe) && b_ok[l] && ih >= 0 && iw >= 0 && ih < p.H && iw < p.W;
            if (ok) v = *reinterpret_cast<const bf16x8*>(p.x + ((size_t)(b_n[l] * p.H + ih) * p.W + iw) * p.Cin + b_ci[l]);
            rbv[l] = ok;
            rb[l] = v;
            // advance this chunk's pixel by one K step (32 output pixels)
            b_ow[l] += 32;
            while (b_ow[l] >= p.OW) { b_ow[l] -= p.OW; ++b_oh[l]; }
            while (b_oh[l] >= p.OH) { b_oh[l] -= p.OH; ++b_n[l]; }
        }
    };
    auto store_tile = [&](int buf) {
        char* base = smem + buf * TILE_BYTES;
#pragma unroll
        for (int l = 0; l < AL; ++l) {
            int e = tid + l * NTHREADS;
            int row = e / ACH, ch = e - row * ACH;
            *reinterpret_cast<bf16x8*>(base + row * AROW + ((ch ^ (tr_swz<BM>(row) >> 1)) << 4)) = ra[l];
        }
#pragma unroll
        for (int l = 0; l < BL; ++l) {
            int e = tid + l * NTHREADS;
            int row = e / BCH, ch = e - row * BCH;
            bf16x8 v = rb[l];
            if (p.in_scale && rbv[l]) v = f32_to_bf8(transform8(v, p.in_scale, p.in_shift, b_ci[l], p.act));
            *reinterpret_cast<bf16x8*>(base + 32 * AROW + row * BROW + ((ch ^ (tr_swz<BN>(row) >> 1)) << 4)) = v;
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (pe - ps + 31) / 32;
    issue_loads(ps);
    store_tile(0);
    __syncthreads();
    const int li = lane & 15, lg = lane >> 4;
    // transpose-read addressing: lane li of a 16-lane group supplies the 8-byte unit
    // [pixel row 8*lg + (li>>2) (+4)][channels 4*(li&3) ..+3]; it receives channel li of rows 0..3.
    const int trow = 8 * lg + (li >> 2), tq = li & 3;
    const int a_lo = trow * AROW, a_hi = (trow + 4) * AROW, b_lo = trow * BROW, b_hi = (trow + 4) * BROW;
    const int ax_lo = tr_swz<BM>(trow), ax_hi = tr_swz<BM>(trow + 4), bx_lo = tr_swz<BN>(trow), bx_hi = tr_swz<BN>(trow + 4);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) issue_loads(ps + (kt + 1) * 32);
        const char* base = smem + buf * TILE_BYTES;
        bf16x8 fa[MT], fb[NT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int u = (wm * (BM / 2) + t * 16) / 4 + tq;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + a_lo + ((u ^ ax_lo) << 3)));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + a_hi + ((u ^ ax_hi) << 3)));
            union { struct { s16x4 a, b; } s; bf16x8 v; } cvt;
            cvt.s.a = lo; cvt.s.b = hi;
            fa[t] = cvt.v;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int u = (wn * (BN / 2) + t * 16) / 4 + tq;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 32 * AROW + b_lo + ((u ^ bx_lo) << 3)));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 32 * AROW + b_hi + ((u ^ bx_hi) << 3)));
            union { struct { s16x4 a, b; } s; bf16x8 v; } cvt;
            cvt.s.a = lo; cvt.s.b = hi;
            fb[t] = cvt.v;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }
    const int taps = p.KH * p.KW;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int nn = n0 + wn * (BN / 2) + nt * 16 + li;
            const int tap = nn >> p.cin_shift;
            const int ci = nn - (tap << p.cin_shift);
            if (nn >= p.NK || ci >= p.cin_true) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * (BM / 2) + mt * 16 + lg * 4 + r;
                if (co < p.Cout) atomicAdd(p.dw + ((size_t)co * p.cin_true + ci) * taps + tap, acc[mt][nt][r]);
            }
        }
}

int ilog2_exact(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return (1 << s) == v ? s : -1;
}

}  // namespace

extern "C" int adamml_conv_fwd(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale,
                               const float* in_shift, void* y, double* stats, hipStream_t stream) {
    if (!d || !x || !w_packed || !y) return adamml_set_error(ADAMML_EINVAL, "conv_fwd: null argument");
    if (d->Cin % 8 || d->Cout % 8) return adamml_set_error(ADAMML_EINVAL, "conv_fwd: channels must be multiples of 8 (Cin=%d Cout=%d)", d->Cin, d->Cout);
    ConvP p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w_packed; p.in_scale = in_scale; p.in_shift = in_shift;
    p.y = (bf16_t*)y; p.stats = stats;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.up = d->up < 1 ? 1 : d->up;
    p.up_shift = ilog2_exact(p.up);
    if (p.up_shift < 0) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_fwd: up=%d is not a power of two", p.up);
    p.act = d->act; p.accumulate = d->accumulate;
    p.P = d->N * d->OH * d->OW; p.K = d->KH * d->KW * d->Cin;
    const bool multitap = d->KH * d->KW > 1;
    p.cin_shift = 0;
    if (multitap) {
        p.cin_shift = ilog2_exact(d->Cin);
        if (p.cin_shift < 0) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_fwd: KxK conv needs power-of-two Cin (got %d)", d->Cin);
    }
    if (p.P <= 0) return ADAMML_OK;
    bool narrow = d->Cout <= 64 || (d->Cout % 128 != 0 && d->Cout < 256);
    // small problems: halve the cout tile so that at least ~2 workgroups per CU exist
    if (!narrow && (long)ceil_div(p.P, BP) * ceil_div(d->Cout, 128) < 512) narrow = true;
    const int BC = narrow ? 64 : 128;
    p.n_ptiles = ceil_div(p.P, BP);
    p.n_ctiles = ceil_div(d->Cout, BC);
    dim3 grid(p.n_ptiles * p.n_ctiles), block(NTHREADS);
    if (BC == 64) {
        if (multitap) hipLaunchKernelGGL((conv_gemm_kernel<64, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((conv_gemm_kernel<64, false>), grid, block, 0, stream, p);
    } else {
        if (multitap) hipLaunchKernelGGL((conv_gemm_kernel<128, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((conv_gemm_kernel<128, false>), grid, block, 0, stream, p);
    }
    return adamml_check_launch("conv_fwd");
}

extern "C" int adamml_conv_bwd_data(const adamml_conv_desc_t* d, const void* dz, const void* w_dgrad_packed, void* dx,
                                    int accumulate, hipStream_t stream) {
    // d describes the FORWARD conv; the data gradient is a stride-1 conv of the (zero-upsampled) dz with the
    // flipped / transposed weight pack (adamml_pack_conv_weight, mode 1).
    if (!d) return adamml_set_error(ADAMML_EINVAL, "conv_bwd_data: null desc");
    adamml_conv_desc_t g = *d;
    g.N = d->N; g.H = d->OH; g.W = d->OW; g.Cin = d->Cout;
    g.OH = d->H; g.OW = d->W; g.Cout = d->Cin;
    g.stride = 1; g.up = d->stride; g.pad = d->KH - 1 - d->pad;
    g.act = ACT_NONE; g.accumulate = accumulate;
    return adamml_conv_fwd(&g, dz, w_dgrad_packed, nullptr, nullptr, dx, nullptr, stream);
}

extern "C" int adamml_conv_bwd_weight(const adamml_conv_desc_t* d, const void* dz, const void* x, const float* in_scale,
                                      const float* in_shift, float* dw, int cin_true, hipStream_t stream) {
    if (!d || !dz || !x || !dw) return adamml_set_error(ADAMML_EINVAL, "conv_bwd_weight: null argument");
    if (d->Cin % 8 || d->Cout % 8) return adamml_set_error(ADAMML_EINVAL, "conv_bwd_weight: channels must be multiples of 8");
    WgradP p;
    p.dz = (const bf16_t*)dz; p.x = (const bf16_t*)x; p.in_scale = in_scale; p.in_shift = in_shift; p.dw = dw;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.act = d->act; p.cin_true = cin_true;
    p.P = d->N * d->OH * d->OW;
    if (p.P <= 0) return ADAMML_OK;
    const int taps = d->KH * d->KW;
    p.NK = taps * d->Cin;
    p.cin_shift = 30;                       // 1x1: tap = n >> 30 = 0
    if (taps > 1) {
        p.cin_shift = ilog2_exact(d->Cin);
        if (p.cin_shift < 0) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_weight: KxK conv needs power-of-two Cin (got %d)", d->Cin);
    }
    const int BM = d->Cout <= 64 ? 64 : 128;
    const int BN = p.NK <= 64 ? 64 : 128;
    p.n_cotiles = ceil_div(d->Cout, BM);
    const int n_ntiles = ceil_div(p.NK, BN);
    p.n_tiles = p.n_cotiles * n_ntiles;
    // pixel splits: a multiple of 8 (one per XCD per round), ~1024 workgroups in total, >= 256 pixels each
    int nsplit = ceil_div(ceil_div(1024, p.n_tiles), 8) * 8;
    int ppb = ceil_div(ceil_div(p.P, nsplit), 32) * 32;
    if (ppb < 256) ppb = 256;
    nsplit = ceil_div(ceil_div(p.P, ppb), 8) * 8;
    p.pix_per_block = ppb;
    dim3 grid(nsplit * p.n_tiles, 1, 1), block(NTHREADS);
    if (BM == 64 && BN == 64) hipLaunchKernelGGL((conv_wgrad_kernel<64, 64>), grid, block, 0, stream, p);
    else if (BM == 64) hipLaunchKernelGGL((conv_wgrad_kernel<64, 128>), grid, block, 0, stream, p);
    else if (BN == 64) hipLaunchKernelGGL((conv_wgrad_kernel<128, 64>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<128, 128>), grid, block, 0, stream, p);
    return adamml_check_launch("conv_bwd_weight");
}
