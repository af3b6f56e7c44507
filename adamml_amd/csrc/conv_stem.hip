// ResNet stem: 7x7 stride-2 pad-3 conv of a <=4-channel image to 64 channels (models/resnet.py:138-139), gfx950.
//
// The generic implicit-GEMM kernel gathers every (tap, 8-channel) chunk of the 49 taps from L1/L2: 100 KB of 16-byte
// gathers per 128-pixel tile and K = 49*8 = 392 with 5/8 of it zero padding -- L1-path bound at ~1 TB/s.  Here a workgroup
// stages the INPUT PATCH of a strip of output rows in LDS once (4 real+pad channels = 8 B per pixel) and builds the MFMA
// operands straight from it:
//   K order = (kh, kw in 0..7, c in 0..3)  -> 7 K-steps of 32 (kw = 7 and c = 3 carry zero weights): K = 224, not 392;
//   A operand of lane (pixel li, k-group lg) at K-step kh = the 16 contiguous bytes of patch pixels (2r+kh, 2c+2lg) and
//   (2r+kh, 2c+2lg+1): one aligned ds_read_b128, conflict-free across the 16 pixels of a tile (16 B apart);
//   the weights [64][224] stay resident in LDS for all tiles of the workgroup.
// HBM traffic is the input once (+ row halo from L2) and the output once; the statistics of the stored outputs come from
// the matrix cores (see conv_gemm.hip epilogue).
#include "common.h"
#include "../../include/adamml_hip.h"


namespace {

constexpr int NT = 256;
constexpr int KSTEPS = 7;            // one per kernel row
constexpr int KTOT = KSTEPS * 32;    // 224
constexpr int WROW = KTOT * 2 + 16;  // LDS bytes per weight row (+16 B skew)
constexpr int MAXPX = 256;           // output pixels per tile (16 MFMA pixel tiles, 4 per wave)
constexpr int SROW = 128 + 8;        // staging row: 64 bf16 + 8 B skew
constexpr int MAXSLOT = 6;           // 16-byte patch slots per thread

struct StemP {
    const bf16_t* x;     // [G*N, H, W, xc] bf16 (xc = 8: channels 0..3 used)
    const bf16_t* w;     // [64][7][8][4] bf16 (adamml_pack_conv_weight mode 3)
    bf16_t* y;           // [G*N, OH, OW, 64]
    double* stats;       // [G][SLOTS][128] or null
    int N, H, W, xc, OH, OW;
    int R;               // output rows per tile
    int tiles_per_img, total_tiles, tpb;
    int PW, PR;          // patch width (pixels, even) / rows
    size_t gx, gy;
};

__global__ __launch_bounds__(NT, 2) void conv_stem_kernel(StemP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;                                   // [64][WROW]
    float* cs = reinterpret_cast<float*>(smem + 64 * WROW);          // [128] channel sums / second moments
    char* s_patch = smem + 64 * WROW + 512;              // patch [PR][PW] x 8 B, later the staging tile [MAXPX][SROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int g = blockIdx.y;
    p.x += (size_t)g * p.gx;
    p.y += (size_t)g * p.gy;

    // weights -> LDS (once per workgroup)
    {
        constexpr int WSL = 64 * (KTOT / 8) / NT;          // 7 sixteen-byte slots per thread, all loads in flight together
        static_assert(64 * (KTOT / 8) % NT == 0, "weight slots must divide evenly");
        bf16x8 wv[WSL];
#pragma unroll
        for (int l = 0; l < WSL; ++l) wv[l] = *reinterpret_cast<const bf16x8*>(p.w + (size_t)(tid + l * NT) * 8);
#pragma unroll
        for (int l = 0; l < WSL; ++l) {
            const int e = tid + l * NT;
            const int co = e / (KTOT / 8), ch = e - co * (KTOT / 8);
            *reinterpret_cast<bf16x8*>(s_w + co * WROW + ch * 16) = wv[l];
        }
    }
    if (tid < 128) cs[tid] = 0.f;

    // ---- input patch of a tile: rows 2*oh0-3 .. +PR-1, columns -3 .. PW-4; out-of-image = 0; 2 pixels (16 B) per slot.
    // The slot -> (row, column) map is tile-invariant.  All loads of a patch are issued back to back into registers (one
    // HBM round trip, not one per slot) and the NEXT tile's loads are in flight while the current tile is computed.
    const int pairs_per_row = p.PW >> 1;
    const int nslots = p.PR * pairs_per_row;
    int s_off[MAXSLOT], s_ih[MAXSLOT], s_iw[MAXSLOT];
#pragma unroll
    for (int l = 0; l < MAXSLOT; ++l) {
        const int e = tid + l * NT;
        const int pr = e / pairs_per_row, pc = (e - pr * pairs_per_row) * 2;
        s_ih[l] = e < nslots ? pr : -(1 << 20);                // dead slots fail the row test
        s_iw[l] = pc - 3;
        s_off[l] = (pr * p.PW + pc) * 8;
    }
    s16x4 ra[MAXSLOT], rb[MAXSLOT];
    unsigned vmask = 0;                  // 2 validity bits per slot of the prefetched patch
    auto load_patch = [&](int tile) {
        const int n = tile / p.tiles_per_img, tr = tile - n * p.tiles_per_img;
        const int ih0 = 2 * tr * p.R - 3;
        vmask = 0;
        const bf16_t* img = p.x + (size_t)n * p.H * p.W * p.xc;
#pragma unroll
        for (int l = 0; l < MAXSLOT; ++l) {
            // unconditional loads from clamped addresses + select: per-slot branches would serialise the HBM round trips
            const int ih = ih0 + s_ih[l], iw = s_iw[l];
            const bool rok = (unsigned)ih < (unsigned)p.H;
            const bf16_t* row = img + (size_t)min(max(ih, 0), p.H - 1) * p.W * p.xc;
            // (validity is applied where the registers are CONSUMED: a select here would make the compiler wait for the
            // loads right after issuing them and defeat the prefetch)
            ra[l] = *reinterpret_cast<const s16x4*>(row + min(max(iw, 0), p.W - 1) * p.xc);
            rb[l] = *reinterpret_cast<const s16x4*>(row + min(max(iw + 1, 0), p.W - 1) * p.xc);
            vmask |= ((rok && (unsigned)iw < (unsigned)p.W) ? 1u : 0u) << (2 * l);
            vmask |= ((rok && (unsigned)(iw + 1) < (unsigned)p.W) ? 2u : 0u) << (2 * l);
        }
    };
    const int tile0 = blockIdx.x * p.tpb;
    if (tile0 < p.total_tiles) load_patch(tile0);
    // patch offset of each of this lane's pixels: tile-invariant (integer divisions kept out of the tile loop); pixels
    // past a short last strip still address valid patch rows and are zeroed at staging
    int pixoff[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        int q = (wave * 4 + pt) * 16 + li;
        if (q >= p.R * p.OW) q = 0;
        const int r = q / p.OW, c = q - r * p.OW;
        pixoff[pt] = ((2 * r) * p.PW + 2 * c + 2 * lg) * 8;
    }

    for (int it = 0; it < p.tpb; ++it) {
        const int tile = tile0 + it;
        if (tile >= p.total_tiles) break;
        const int n = tile / p.tiles_per_img, tr = tile - n * p.tiles_per_img;
        const int oh0 = tr * p.R;
        const int rows = min(p.R, p.OH - oh0);
        const int npx = rows * p.OW;
#pragma unroll
        for (int l = 0; l < MAXSLOT; ++l) {
            if (tid + l * NT < nslots) {
                union { struct { s16x4 a, b; } s; bf16x8 v; } u;
                u.s.a = (vmask >> (2 * l)) & 1u ? ra[l] : s16x4{0, 0, 0, 0};
                u.s.b = (vmask >> (2 * l)) & 2u ? rb[l] : s16x4{0, 0, 0, 0};
                *reinterpret_cast<bf16x8*>(s_patch + s_off[l]) = u.v;
            }
        }
        __syncthreads();
        if (it + 1 < p.tpb && tile + 1 < p.total_tiles) load_patch(tile + 1);

        // ---- MFMA: wave w owns pixel tiles 4w..4w+3 (16 consecutive output pixels each, row-major over the strip) -------
        f32x4 acc[4][4];                                     // [cout tile][pixel tile]
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int rowbytes = p.PW * 8;
#pragma unroll
        for (int kh = 0; kh < KSTEPS; ++kh) {
            bf16x8 fw[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) fw[ct] = *reinterpret_cast<const bf16x8*>(s_w + (ct * 16 + li) * WROW + kh * 64 + lg * 16);
            // (no per-pixel-tile guard: the few dead tiles of a short strip compute on valid patch addresses and are zeroed
            // at staging; straight-line code lets the compiler overlap the next fragments' ds_reads with these MFMAs)
            bf16x8 fa[4];
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) fa[pt] = *reinterpret_cast<const bf16x8*>(s_patch + pixoff[pt] + kh * rowbytes);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ct], fa[pt], acc[ct][pt], 0, 0, 0);
        }
        __syncthreads();                                     // patch consumed: its LDS becomes the staging tile

        // ---- stage [MAXPX][64] bf16 (rows >= npx zero), store the strip (contiguous in HBM) 16 B per lane ---------------
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int q = (wave * 4 + pt) * 16 + li;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                bf16x4 v = f32_to_bf4(acc[ct][pt]);
                if (q >= npx) v = bf16x4{0, 0, 0, 0};
                *reinterpret_cast<bf16x4*>(s_patch + q * SROW + (ct * 16 + lg * 4) * 2) = v;
            }
        }
        __syncthreads();
        bf16_t* ybase = p.y + ((size_t)n * p.OH + oh0) * p.OW * 64;
        for (int e = tid; e < npx * 8; e += NT) {
            const int q = e >> 3, ch = e & 7;
            const s16x4 lo = *reinterpret_cast<const s16x4*>(s_patch + q * SROW + ch * 16);
            const s16x4 hi = *reinterpret_cast<const s16x4*>(s_patch + q * SROW + ch * 16 + 8);
            union { struct { s16x4 a, b; } s; bf16x8 v; } u;
            u.s.a = lo; u.s.b = hi;
            *reinterpret_cast<bf16x8*>(ybase + (size_t)q * 64 + ch * 8) = u.v;
        }
        if (p.stats) {
            // wave w: channels 16w..16w+15; F = [32 pixels][16 channels] by transpose reads; ones*F and diag(F^T F)
            union { s16x4 h[2]; bf16x8 v; } ones;
            ones.h[0] = s16x4{0x3F80, 0x3F80, 0x3F80, 0x3F80};
            ones.h[1] = ones.h[0];
            const int trow = 8 * lg + (li >> 2);
            f32x4 dsum = {0.f, 0.f, 0.f, 0.f}, dsq = {0.f, 0.f, 0.f, 0.f};
            const int nps = (npx + 31) >> 5;
            for (int ps = 0; ps < nps; ++ps) {
                const char* fp = s_patch + (ps * 32 + trow) * SROW + (wave * 16 + 4 * (li & 3)) * 2;
                union { s16x4 h[2]; bf16x8 v; } f;
                f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(fp));
                f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(fp + 4 * SROW));
                dsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, f.v, dsum, 0, 0, 0);
                dsq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.v, f.v, dsq, 0, 0, 0);
            }
            if (lg == 0) cs[wave * 16 + li] += dsum[0];
            if ((li >> 2) == lg) {
                const int r = li & 3;
                cs[64 + wave * 16 + li] += r == 0 ? dsq[0] : r == 1 ? dsq[1] : r == 2 ? dsq[2] : dsq[3];
            }
        }
        __syncthreads();                                     // staging consumed before the next patch lands
    }
    if (p.stats) {
        __syncthreads();
        // (every cs entry was accumulated by one wave in tile order; one exact integer-bin add per channel and workgroup: common.h)
        if (tid < 128) det_add(p.stats + (size_t)g * ADAMML_STAT_SLOTS * 128 + tid, 128, cs[tid]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem weight gradient: dW[co][c][kh][kw] = sum_p dz[p][co] * x[2p + (kh,kw) - 3][c] as an MFMA GEMM with the PIXELS as the
// reduction dimension, M = 64 cout, N = 224 = (kh, kw 0..7, c 0..3) in the forward kernel's K order.  Both operands are
// pixel-major, so the fragments come from LDS by hardware transpose reads (ds_read_b64_tr_b16) with lane-supplied
// addresses: A from the staged dz tile [pixel][64 cout]; B from the SAME 8-byte-per-pixel input patch the forward uses --
// the 16 columns of an N tile (4 taps kw x 4 channels) are 32 contiguous patch bytes of one pixel, and consecutive pixels
// of the K step are 16 B apart, so the im2col matrix is never materialised.  Accumulators stay in registers over all the
// tiles of a workgroup; one partial [64][224] fp32 per workgroup goes to the workspace.
struct StemWP {
    const bf16_t* x;     // [G*N, H, W, xc]
    const bf16_t* dz;    // [G*N, OH, OW, 64]
    float* ws;           // [nblocks][64][224] partials
    int N, H, W, xc, OH, OW, R, tiles_per_img, total_tiles, tpb, PW, PR;
};

constexpr int DZROW = 128 + 8;       // LDS bytes per staged dz pixel
constexpr int MAXDZ = 8;             // 16-byte dz slots per thread (MAXPX px x 8 chunks / 256)

__global__ __launch_bounds__(NT, 2) void conv_stem_wgrad_kernel(StemWP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_patch = smem;                                   // [PR][PW] x 8 B
    char* s_dz = smem + ((p.PR * p.PW * 8 + 15) & ~15);     // [MAXPX][DZROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int pairs_per_row = p.PW >> 1;
    const int nslots = p.PR * pairs_per_row;
    int s_off[MAXSLOT], s_ih[MAXSLOT], s_iw[MAXSLOT];
#pragma unroll
    for (int l = 0; l < MAXSLOT; ++l) {
        const int e = tid + l * NT;
        const int pr = e / pairs_per_row, pc = (e - pr * pairs_per_row) * 2;
        s_ih[l] = e < nslots ? pr : -(1 << 20);
        s_iw[l] = pc - 3;
        s_off[l] = (pr * p.PW + pc) * 8;
    }
    s16x4 ra[MAXSLOT], rb[MAXSLOT];
    bf16x8 rz[MAXDZ];
    unsigned vmask = 0;                  // 2 validity bits per patch slot, then 1 bit per dz slot (bit 16 + l)
    auto load_tile = [&](int tile) {
        const int n = tile / p.tiles_per_img, tr = tile - n * p.tiles_per_img;
        const int ih0 = 2 * tr * p.R - 3;
        vmask = 0;
        const bf16_t* img = p.x + (size_t)n * p.H * p.W * p.xc;
#pragma unroll
        for (int l = 0; l < MAXSLOT; ++l) {
            const int ih = ih0 + s_ih[l], iw = s_iw[l];
            const bool rok = (unsigned)ih < (unsigned)p.H;
            const bf16_t* row = img + (size_t)min(max(ih, 0), p.H - 1) * p.W * p.xc;
            // (validity is applied where the registers are CONSUMED: a select here would make the compiler wait for the
            // loads right after issuing them and defeat the prefetch)
            ra[l] = *reinterpret_cast<const s16x4*>(row + min(max(iw, 0), p.W - 1) * p.xc);
            rb[l] = *reinterpret_cast<const s16x4*>(row + min(max(iw + 1, 0), p.W - 1) * p.xc);
            vmask |= ((rok && (unsigned)iw < (unsigned)p.W) ? 1u : 0u) << (2 * l);
            vmask |= ((rok && (unsigned)(iw + 1) < (unsigned)p.W) ? 2u : 0u) << (2 * l);
        }
        const int oh0 = tr * p.R;
        const int npx = min(p.R, p.OH - oh0) * p.OW;
        const bf16_t* zb = p.dz + ((size_t)n * p.OH + oh0) * p.OW * 64;
#pragma unroll
        for (int l = 0; l < MAXDZ; ++l) {
            const int e = tid + l * NT;                        // (pixel e >> 3, chunk e & 7); pixels >= npx contribute zero
            rz[l] = *reinterpret_cast<const bf16x8*>(zb + (size_t)(e < npx * 8 ? e : 0) * 8);
            vmask |= (e < npx * 8 ? 1u : 0u) << (16 + l);
        }
    };
    f32x4 acc[4][4];                                        // [cout tile][own N tile j]: N tile = wave + 4*j  (kh = nt >> 1, kw half = nt & 1)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tile0 = blockIdx.x * p.tpb;
    if (tile0 < p.total_tiles) load_tile(tile0);
    const int trow = 8 * lg + (li >> 2);                    // pixel of the K step this lane addresses (and +4)
    // patch offsets of this lane's two transposed-read rows for every K step: tile-invariant, no divisions in the tile loop
    int poff0[MAXPX / 32], poff1[MAXPX / 32];
#pragma unroll
    for (int ks = 0; ks < MAXPX / 32; ++ks) {
        int q0 = ks * 32 + trow, q1 = q0 + 4;
        if (q0 >= p.R * p.OW) q0 = 0;                       // past-the-end pixels read pixel 0 (their dz rows are zero)
        if (q1 >= p.R * p.OW) q1 = 0;
        const int r0 = q0 / p.OW, c0 = q0 - r0 * p.OW, r1 = q1 / p.OW, c1 = q1 - r1 * p.OW;
        poff0[ks] = ((2 * r0) * p.PW + 2 * c0 + (li & 3)) * 8;
        poff1[ks] = ((2 * r1) * p.PW + 2 * c1 + (li & 3)) * 8;
    }

    for (int it = 0; it < p.tpb; ++it) {
        const int tile = tile0 + it;
        if (tile >= p.total_tiles) break;
#pragma unroll
        for (int l = 0; l < MAXSLOT; ++l) {
            if (tid + l * NT < nslots) {
                union { struct { s16x4 a, b; } s; bf16x8 v; } u;
                u.s.a = (vmask >> (2 * l)) & 1u ? ra[l] : s16x4{0, 0, 0, 0};
                u.s.b = (vmask >> (2 * l)) & 2u ? rb[l] : s16x4{0, 0, 0, 0};
                *reinterpret_cast<bf16x8*>(s_patch + s_off[l]) = u.v;
            }
        }
#pragma unroll
        for (int l = 0; l < MAXDZ; ++l) {
            const int e = tid + l * NT;
            if (e < MAXPX * 8) {
                union { struct { s16x4 a, b; } s; bf16x8 v; } u;
                u.v = (vmask >> (16 + l)) & 1u ? rz[l] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                char* dst = s_dz + (e >> 3) * DZROW + (e & 7) * 16;
                *reinterpret_cast<s16x4*>(dst) = u.s.a;
                *reinterpret_cast<s16x4*>(dst + 8) = u.s.b;
            }
        }
        __syncthreads();
        if (it + 1 < p.tpb && tile + 1 < p.total_tiles) load_tile(tile + 1);
        const int tr = tile % p.tiles_per_img;
        const int npx = min(p.R, p.OH - tr * p.R) * p.OW;
        const int nks = (npx + 31) >> 5;
#pragma unroll
        for (int ks = 0; ks < MAXPX / 32; ++ks) {
            if (ks >= nks) break;
            const char* pb0 = s_patch + poff0[ks];
            const char* pb1 = s_patch + poff1[ks];
            const char* zb0 = s_dz + (ks * 32 + trow) * DZROW + 4 * (li & 3) * 2;
            bf16x8 fa[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                union { s16x4 h[2]; bf16x8 v; } f;
                f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(zb0 + mt * 32));
                f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(zb0 + mt * 32 + 4 * DZROW));
                fa[mt] = f.v;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nt = wave + 4 * j;
                if (nt < 14) {                                   // wave-uniform
                    const int toff = ((nt >> 1) * p.PW + 4 * (nt & 1)) * 8;
                    union { s16x4 h[2]; bf16x8 v; } f;
                    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb0 + toff));
                    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb1 + toff));
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], f.v, acc[mt][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                         // tile consumed before the next one lands
    }
    // partial dW of this workgroup: D rows (cout) = mt*16 + lg*4 + r, column (kh, kw, c) = nt*16 + li
    float* out = p.ws + (size_t)blockIdx.x * 64 * KTOT;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nt = wave + 4 * j;
            if (nt < 14) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(mt * 16 + lg * 4 + r) * KTOT + nt * 16 + li] = acc[mt][j][r];
            }
        }
}

// dw[co][c][kh][kw] += sum_b ws[b][co][kh][kw8][c4]: 16 elements x 16 partial-lanes per workgroup (the partial loop is the long axis)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* ws, float* dw, int nblk, int cin_true) {
    __shared__ float red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + tx;
    float a = 0.f;
    if (e < 64 * KTOT)
        for (int b = ty; b < nblk; b += 16) a += ws[(size_t)b * 64 * KTOT + e];
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && e < 64 * KTOT) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][tx];
        const int c = e & 3, kw = (e >> 2) & 7, kh = (e >> 5) % 7, co = e / KTOT;
        if (kw < 7 && c < cin_true) dw[((size_t)(co * cin_true + c) * 7 + kh) * 7 + kw] += s;
    }
}

__global__ void pack_stem_weight_kernel(const float* w, bf16_t* out, int cin_true) {
    // fp32 [64][cin_true][7][7] -> bf16 [64][7][8][4], zero for kw == 7 / c >= cin_true
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 64 * KTOT) return;
    const int c = e & 3, kw = (e >> 2) & 7, kh = (e >> 5) % 7, co = e / KTOT;
    float v = 0.f;
    if (kw < 7 && c < cin_true) v = w[((size_t)(co * cin_true + c) * 7 + kh) * 7 + kw];
    reinterpret_cast<__bf16*>(out)[e] = (__bf16)v;
}

}  // namespace

extern "C" int adamml_pack_stem_weight(const float* w, void* out, int cout, int cin_true, hipStream_t stream) {
    if (!w || !out) return adamml_set_error(ADAMML_EINVAL, "pack_stem_weight: null argument");
    if (cout != 64 || cin_true < 1 || cin_true > 4)
        return adamml_set_error(ADAMML_EUNSUPPORTED, "pack_stem_weight: needs cout == 64 and 1..4 input channels (cout=%d cin=%d)", cout, cin_true);
    hipLaunchKernelGGL(pack_stem_weight_kernel, dim3(ceil_div(64 * KTOT, 256)), dim3(256), 0, stream, w, (bf16_t*)out, cin_true);
    return adamml_check_launch("pack_stem_weight");
}

extern "C" int adamml_conv_stem_supported(const adamml_conv_desc_t* d) {
    if (!d) return 0;
    if (d->KH != 7 || d->KW != 7 || d->stride != 2 || d->pad != 3 || d->Cout != 64 || (d->Cin != 8 && d->Cin != 4)) return 0;
    if (d->W % 2 || d->OW != d->W / 2 || d->OH != (d->H + 6 - 7) / 2 + 1 || d->OW > MAXPX || d->OW < 1) return 0;
    const int R = MAXPX / d->OW;
    const int PR = 2 * R + 5, PW = d->W + 6;
    const size_t patch = (size_t)PR * PW * 8, stage = (size_t)MAXPX * SROW;
    if (PR * (PW / 2) > MAXSLOT * NT) return 0;
    return 64 * WROW + 512 + (patch > stage ? patch : stage) <= 64 * 1024;
}

extern "C" int adamml_conv_stem_fwd(const adamml_conv_desc_t* d, const void* x, const void* w_stem_packed, void* y, double* stats,
                                    hipStream_t stream) {
    if (!d || !x || !w_stem_packed || !y) return adamml_set_error(ADAMML_EINVAL, "conv_stem_fwd: null argument");
    if (!adamml_conv_stem_supported(d))
        return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_stem_fwd: needs a 7x7/2 pad 3 conv of a <=4-channel (8-padded) image to 64 "
                                "channels with even W <= %d*2 (H=%d W=%d Cin=%d Cout=%d)", MAXPX, d->H, d->W, d->Cin, d->Cout);
    StemP p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w_stem_packed; p.y = (bf16_t*)y; p.stats = stats;
    p.N = d->N; p.H = d->H; p.W = d->W; p.xc = d->Cin; p.OH = d->OH; p.OW = d->OW;
    p.R = MAXPX / d->OW;
    if (p.R > d->OH) p.R = d->OH;
    p.PR = 2 * p.R + 5; p.PW = d->W + 6;
    p.tiles_per_img = ceil_div(d->OH, p.R);
    p.total_tiles = d->N * p.tiles_per_img;
    if (p.total_tiles <= 0) return ADAMML_OK;
    const int groups = d->groups < 1 ? 1 : d->groups;
    p.gx = (size_t)d->N * d->H * d->W * d->Cin;
    p.gy = (size_t)d->N * d->OH * d->OW * 64;
    // ~1536 workgroups (3 rounds of 2 per CU): the resident weights and the statistics publication are amortised over
    // many strips of the same workgroup
    p.tpb = (int)(((long)p.total_tiles * groups + 1535) / 1536);
    if (p.tpb < 1) p.tpb = 1;
    const size_t patch = (size_t)p.PR * p.PW * 8, stage = (size_t)MAXPX * SROW;
    const size_t lds = 64 * WROW + 512 + (patch > stage ? patch : stage);
    hipLaunchKernelGGL(conv_stem_kernel, dim3(ceil_div(p.total_tiles, p.tpb), groups), dim3(NT), lds, stream, p);
    return adamml_check_launch("conv_stem_fwd");
}

static int stem_wgrad_blocks(const adamml_conv_desc_t* d, int* tpb_out) {
    const int R = MAXPX / d->OW > d->OH ? d->OH : MAXPX / d->OW;
    const long total = (long)(d->groups < 1 ? 1 : d->groups) * d->N * ceil_div(d->OH, R);
    int tpb = (int)((total + 767) / 768);
    if (tpb < 1) tpb = 1;
    *tpb_out = tpb;
    return (int)((total + tpb - 1) / tpb);
}

extern "C" size_t adamml_conv_stem_bwd_weight_workspace(const adamml_conv_desc_t* d) {
    if (!d || !adamml_conv_stem_supported(d)) return 0;
    int tpb;
    return (size_t)stem_wgrad_blocks(d, &tpb) * 64 * KTOT * sizeof(float);
}

extern "C" int adamml_conv_stem_bwd_weight(const adamml_conv_desc_t* d, const void* dz, const void* x, float* dw, int cin_true,
                                           void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!d || !dz || !x || !dw || !workspace) return adamml_set_error(ADAMML_EINVAL, "conv_stem_bwd_weight: null argument");
    if (!adamml_conv_stem_supported(d) || cin_true < 1 || cin_true > 4)
        return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_stem_bwd_weight: not a stem shape (see adamml_conv_stem_supported)");
    StemWP p;
    p.x = (const bf16_t*)x; p.dz = (const bf16_t*)dz; p.ws = (float*)workspace;
    p.N = d->N * (d->groups < 1 ? 1 : d->groups);           // groups are plain batch here: the weight gradient sums over all of them
    p.H = d->H; p.W = d->W; p.xc = d->Cin; p.OH = d->OH; p.OW = d->OW;
    p.R = MAXPX / d->OW;
    if (p.R > d->OH) p.R = d->OH;
    p.PR = 2 * p.R + 5; p.PW = d->W + 6;
    p.tiles_per_img = ceil_div(d->OH, p.R);
    p.total_tiles = p.N * p.tiles_per_img;
    if (p.total_tiles <= 0) return ADAMML_OK;
    const int nblk = stem_wgrad_blocks(d, &p.tpb);
    if (workspace_bytes < (size_t)nblk * 64 * KTOT * sizeof(float))
        return adamml_set_error(ADAMML_EINVAL, "conv_stem_bwd_weight: workspace too small (%zu < %zu bytes)", workspace_bytes,
                                (size_t)nblk * 64 * KTOT * sizeof(float));
    if (p.R * p.OW > MAXDZ * NT / 8) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_stem_bwd_weight: strip of %d pixels too long", p.R * p.OW);
    const size_t lds = ((p.PR * p.PW * 8 + 15) & ~15) + (size_t)MAXPX * DZROW;
    hipLaunchKernelGGL(conv_stem_wgrad_kernel, dim3(nblk), dim3(NT), lds, stream, p);
    int rc = adamml_check_launch("conv_stem_bwd_weight");
    if (rc) return rc;
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(ceil_div(64 * KTOT, 16)), dim3(256), 0, stream, (const float*)workspace, dw, nblk, cin_true);
    return adamml_check_launch("conv_stem_bwd_weight (reduce)");
}
