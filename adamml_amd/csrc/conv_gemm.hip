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
#include <type_traits>
#include <stdlib.h>
#include <stdio.h>


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
    int P, K, cin_shift, n_ptiles, n_ctiles, tpb;
    // data-gradient epilogue fused with the BatchNorm backward reduction of the tensor the gradient flows into
    size_t gx, gy;               // element strides between BatchNorm groups of x / y (blockIdx.y = group)
    int in_gstride;
    const bf16_t* bn_z;          // raw conv output that produced the consumer's input (same shape as y), or null
    const float* bn_vec;         // [4][Cout']: scale, shift, mean, invstd of that BatchNorm
    int bn_act;
    // residual form of that epilogue (RES kernels; the gradient flows into the OUTPUT of a residual add,
    // out = act(bn(z) [+ bn2(z2)] + ..)): the activation mask comes from the stored block output, `accumulate` adds the
    // identity-path gradient already in y, and the sums of a second BatchNorm feeding the same add (downsample branch)
    // are accumulated alongside
    const bf16_t* res_out;       // block output [same shape as y]
    const uint8_t* res_mask;     // or: 1 bit per element, act'(block output) != 0 (adamml_bn_act_add_mask) -- 1/16 of the bytes
    int res_act;
    const bf16_t* bn_z2;         // raw output of the second BatchNorm'd operand of the add, or null
    const float* bn_vec2;
    double* stats2;
    // DUAL kernels (1x1 data gradient whose input dz is the BatchNorm backward of (g, z)): the loader reads g (= x) and z
    // (= x2) and forms dz = A[c] g + B[c] z + C[c] on the way into LDS (aff = [3][K] per group); `side` (optional) receives
    // dz for the weight-gradient kernel, written by the workgroups of cout tile 0 -- adamml_bn_bwd_apply never runs
    const bf16_t* x2;
    const float* aff;
    bf16_t* side;
    // K-concatenated second input (MODE 0): K steps k >= K1 read xb [pixels][C2] (with the lazy in_scale / in_shift transform,
    // which then applies to xb ONLY) instead of x [pixels][K1]; weights are per group (gw elements apart); epi_add [Cout] per
    // group is added to every output row before the epilogue's mask / store.  Used by the algebraic BatchNorm backward
    // (adamml_conv_bwd_data_alg): dx = (W^T diag(A)) g' + (W^T diag(B) W) a + W^T C without ever forming dz.
    const bf16_t* xb;
    int K1, C2;
    size_t gw;
    const float* epi_add;
    // FADD kernels (forward 1x1 conv whose BatchNorm vectors are known BEFORE the launch -- eval mode, or train mode with the
    // statistics derived from the Gram matrix of the input, adamml_gram_stats): the epilogue applies bn_vec (scale, shift of
    // THIS conv's BatchNorm), adds the identity operand res_out (optionally lazily normalised with id_scale / id_shift),
    // applies res_act and writes the block output + its 1-bit activation mask; the raw conv output never reaches HBM
    const float* id_scale;
    const float* id_shift;
    int id_gstride;
    uint8_t* mask_out;
    // PF (RES + EID kernels): the per-group product P[g][cout][c] = sum_p g'[p][cout] * a[p][c] of the gradient tile this kernel
    // has just formed with a second, lazily normalised tensor a [pixels][pf_C] (the input of the conv whose output gradient g'
    // is -- the g'^T a of the algebraic BatchNorm backward) is accumulated on the matrix cores from the staged tile: the separate
    // pass over g' and a (adamml_conv_bwd_weight_grouped) disappears.  One partial [BC][pf_C] per workgroup -> pf_ws.
    const bf16_t* pf_a;
    const float* pf_scale;
    const float* pf_shift;
    float* pf_ws;                // [group][cout tile][split = workgroup of that pair][BC][pf_C]
    int pf_act, pf_gs, pf_nsplit;
    // MODE 3 (one parity class of the data gradient of a stride-2 conv, see conv_dgrad_stride2)
    int wK;                      // weight row stride in elements (== K except in MODE 3, where K covers the class taps only)
    int cls_nt;                  // taps of this class (0..4)
    unsigned cls_code;           // per tap, 8 bits: dh | dw << 1 | weight-pack tap index << 2
    int oH, oW, o_ph, o_pw;      // output rows are scattered: pixel (i, j) of the class grid -> (2i + o_ph, 2j + o_pw) of [oH, oW]
    // TP (FADD kernels): the block output feeds ONLY a temporal max-pool (models/common.py:4-33 after the last block of a ResNet stage,
    // models/resnet.py:205-209).  A pixel tile then holds the TP frames of a clip for BP / TP pixels (row r = frame r / (BP / TP), pixel
    // r % (BP / TP) of block tp_blk), the epilogue pools over the frames and writes the POOLED tensor tp_y [clips * TP / 2][Q][Cout] and
    // 2 bits per pooled element (tp_code, one uint16 per 8-channel chunk: window tap 0..2 of the first maximum, 3 = maximum <= 0, i.e.
    // no gradient passes the ReLU) -- the full-rate block output, its activation mask and the pool's own pass never touch HBM
    int tp_nblk, tp_Q;           // pixel blocks per clip (ceil(Q / (BP / TP))), pixels per frame
    bf16_t* tp_y;
    uint16_t* tp_code;
};

__device__ __forceinline__ int lds_off(int row, int chunk) {
    // 64-byte rows, 16-byte chunks, chunk ^= 2*((row>>3)&1): conflict-free ds_read_b128 for the
    // 16-lane service groups of gfx950 (MI355X_MICROARCH.md LDS table)
    return row * 64 + ((chunk ^ (((row >> 3) & 1) << 1)) << 4);
}

// MODE 0: 1x1 conv; MODE 1: KxK conv, fast tap addressing (per-row validity mask + LDS tap-offset table);
// MODE 2: generic path with zero-upsampled input (data gradient of strided convs other than the stride-2 fast path);
// MODE 3: one parity class of a stride-2 data gradient: MODE 1 addressing over the class's tap subset (input and weight
//         tap tables), output rows scattered with stride 2 -- no MFMA or load is spent on the zeros of the up-sampling.
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// LDS image of one K step: [32 pixels][CH channels] bf16, row-major, 8-byte units XOR-swizzled so that the
// ds_read_b64_tr_b16 of a 32-lane service group (rows {r..r+3} U {r+8..r+11}) touches 64 distinct banks.
template <int CH>
__device__ __forceinline__ int tr_swz(int row) {
    if (CH >= 128) return ((row & 3) | (((row >> 3) & 1) << 2)) << 2;       // 256-byte rows: all rows start at bank 0
    return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 2;               // 128-byte rows: parity picks the bank half
}

__device__ const uint4 g_zero_page[4] = {};          // 64 zero bytes: source of the out-of-range chunks of an LDS-DMA tile

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

// PD = register prefetch depth (K steps of global loads in flight).  PD 1 keeps 3-4 workgroups per CU (latency hidden by
// occupancy: best for the big, short-K layers); PD 3 is for small grids with long K loops (layer3/4), where a CU holds a
// single workgroup and only explicit look-ahead hides the HBM round trip.
// RES: residual form of the BatchNorm-fused data-gradient epilogue (MODE 0 only).  Its epilogue keeps four 16-byte streams
// per row in flight and two sets of per-channel vectors, so it is compiled for 2 workgroups per CU (256 VGPRs) as its own
// instantiation -- inside the shared kernel it pushed every MODE 0 instance into scratch spills.
// DUAL: the BatchNorm-backward affine of two source tensors is applied by the MODE 0 loader (own instantiation as well:
// a second register ring for z, 2 workgroups per CU, deep prefetch).
// CAT: K-concatenated second input + per-group weights + epilogue constant (algebraic BatchNorm backward); own instantiation
// for the same reason.
// FADD: forward conv + BatchNorm + residual add + activation in the epilogue (see ConvP::id_scale); own instantiation.
// GLDS: operands that are plain in memory (no lazy transform) are staged global -> LDS by LDS-DMA through a ring of three tile
// buffers with counted vmcnt across raw barriers (as conv_wgrad_glds_kernel): no register ring, no ds_write pass.
// EID (FADD / RES): the epilogue's identity-side operands (identity tensor; identity-path gradient + activation mask bits) are
// requested at the START of a tile, behind the first operand loads, instead of in the epilogue: the whole tile then costs ONE HBM
// round trip instead of three (operands, 2 x 4 identity rows), and a workgroup keeps ~3x the bytes in flight.  PMC on the layer-1
// shapes: these kernels held ~350 reads outstanding at the fabric against ~700-900 of the elementwise kernels (TCC_EA0_RDREQ_LEVEL),
// at a LOWER latency per read -- they were bound by their own request rate, not by HBM.  Costs 32 VGPRs: 2 workgroups per CU.
// LZF (GLDS, MODE 0): the lazy BatchNorm + activation transform of the INPUT is applied to the MFMA fragment after its ds_read
// instead of on the way into LDS, so that lazily normalised inputs can be staged by LDS-DMA as well (raw tile global -> LDS, deep
// counted-vmcnt pipeline, no register ring / ds_write pass / exec-masked loads).  Same fma / clamp / round sequence on the same
// values: bit-identical to the staging-side transform.  The two waves that share a pixel half repeat the transform (VALU has the
// slack on these HBM-bound layers); K <= 512 (per-channel vectors in LDS).
// EPI: which epilogue of the plain (non-RES / non-FADD) kernels is compiled.  -1: all of them, selected at run time (the CAT / DUAL /
// deep-prefetch / MODE 2 instances); 0: forward (plain store + statistics on the matrix cores); 1: data gradient with the activation
// mask and the BatchNorm-backward sums (bn_z); 2: data gradient, optionally accumulating into y.  One kernel body for all three made
// every edit of a data-gradient epilogue move the register allocation of the forward instances (spills inside their K loops); apart,
// the forward instances carry no dead epilogue state and the data-gradient ones can request a batch of rows ahead of their stores.
// PF: see ConvP::pf_a (BC = 128, 64 channels of a; RES + GLDS + EID only).
// Workgroups per CU the register allocation is sized for.  The fused instances (RES / DUAL / CAT / EID) take 256 registers; the plain
// ones rely on occupancy (PD 1: 3-4 workgroups per CU) or on their register ring (PD 3).  The instances under "fewer" did not fit
// that budget (2-13 registers in scratch, tools/kernel_resources.py) and get one workgroup less: the data-gradient epilogues that are
// staged through registers at 64-wide tiles, the strided (MODE 2) loaders and the FADD epilogue without EID.
constexpr int conv_gemm_wg_per_cu(int BC, int MODE, int PD, bool RES, bool DUAL, bool CAT, bool FADD, bool GLDS, int EID, int EPI) {
    if (RES || DUAL || CAT || EID) return 2;
    const int n = PD == 1 ? (BC == 128 ? 3 : 4) : (BC == 128 ? 2 : 3);
    const bool fewer = (MODE == 2 && !(BC == 128 && PD == 3)) || FADD ||
                       (BC == 64 && PD == 1 && ((EPI == 1 && !(MODE == 0 && GLDS)) || (EPI == 2 && MODE == 1))) ||
                       (BC == 128 && MODE == 1 && EPI == 2 && !GLDS);
    return fewer ? n - 1 : n;
}

template <int BC, int MODE, int PD, bool RES = false, bool DUAL = false, bool CAT = false, bool FADD = false, bool GLDS = false, int EID = 0,
          bool LZF = false, int EPI = -1, bool PF = false, int TP = 0>
__global__ __launch_bounds__(NTHREADS, conv_gemm_wg_per_cu(BC, MODE, PD, RES, DUAL, CAT, FADD, GLDS, EID, EPI)) void conv_gemm_kernel(ConvP p) {
    static_assert(TP == 0 || (FADD && EID && MODE == 0 && BC == 128 && (TP == 2 || TP == 4 || TP == 8)), "TP: FADD + EID, 128-wide cout tiles");
    constexpr int PPF = TP ? BP / TP : BP;  // TP: pixels of one frame in a tile
    constexpr int WCT = BC / 32;            // 16-wide cout tiles per wave
    constexpr int WROWS = BC / 64;          // weight rows staged per thread
    constexpr int TILE_BYTES = (BP + BC) * 64;
    constexpr int EPI_BYTES = BP * (BC * 2 + 16);
    constexpr int NBUF = GLDS ? 3 : 2;
    constexpr int STAGE_BYTES = (NBUF * TILE_BYTES) > EPI_BYTES ? (NBUF * TILE_BYTES) : EPI_BYTES;
    // per-channel sums of a workgroup: the forward statistics come out of the matrix cores with ONE owner wave per channel (one row of
    // 2 * BC floats); the data-gradient epilogues fold on the VALU, where all four waves hold partials of every channel -- each wave then
    // accumulates into its own row and the rows are summed in wave order at publication (common.h: reproducible reductions)
    constexpr int NSR = (FADD || EPI == 0) ? 1 : 4;
    constexpr int CS_BYTES = NSR * 2 * BC * 4;
    constexpr int CS2_OFF = STAGE_BYTES + CS_BYTES + (MODE == 0 ? 0 : 256) + (MODE == 3 ? BP * 4 + 16 : 0);     // second sum set (RES); 256: tap tables
    // MODE 0: the per-input-channel vectors of the loader transform (lazy BatchNorm scale / shift, or the three DUAL affine
    // vectors) are staged in LDS once per workgroup when K <= VEC_MAXK: read from global inside store_tile they were an
    // exposed L1/L2 round trip in every K step (the loads can only be issued when the tile registers are consumed)
    constexpr int VEC_MAXK = 1024;
    constexpr int VEC_OFF = CS2_OFF + (RES ? CS_BYTES : 0);
    constexpr int LZF_MAXK = 512;              // LZF keeps 3 workgroups per CU at BC = 128: 3 x (3 tiles + sums + 4 KB of vectors) <= 160 KB
    constexpr int PF_C = 64;                   // channels of the PF operand
    constexpr int PF_OFF = (VEC_OFF + (LZF ? 2 * LZF_MAXK * 4 : (MODE == 0 && !GLDS) ? (DUAL ? 3 : 2) * VEC_MAXK * 4 : 0) + 1023) / 1024 * 1024;
    constexpr int SMEM_BYTES = PF ? PF_OFF + BP * PF_C * 2 : VEC_OFF + (LZF ? 2 * LZF_MAXK * 4 : (MODE == 0 && !GLDS) ? (DUAL ? 3 : 2) * VEC_MAXK * 4 : 0);   // + per-channel sums + tap-offset table (+ MODE 3: output row table, weight tap table)
    __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];

    {   // BatchNorm group of this workgroup: one launch covers the S per-segment calls of the reference
        const int g = blockIdx.y;
        p.x += (size_t)g * p.gx;
        p.y += (size_t)g * p.gy;
        if (p.stats) p.stats += (size_t)g * ADAMML_STAT_SLOTS * 2 * p.Cout;
        if (p.in_scale) { p.in_scale += (size_t)g * p.in_gstride; p.in_shift += (size_t)g * p.in_gstride; }
        if (p.bn_z) { p.bn_z += (size_t)g * p.gy; p.bn_vec += (size_t)g * 4 * p.Cout; }
        if (CAT) {
            p.xb += (size_t)g * (p.gx / p.K1) * p.C2;
            p.w += (size_t)g * p.gw;
            p.epi_add += (size_t)g * p.Cout;
        }
        if (DUAL) {
            p.x2 += (size_t)g * p.gx;
            p.aff += (size_t)g * 3 * p.K;
            if (p.side) p.side += (size_t)g * p.gx;
        }
        if (FADD) {
            p.bn_vec += (size_t)g * 4 * p.Cout;
            if (p.res_out) p.res_out += (size_t)g * p.gy;
            if (p.mask_out) p.mask_out += ((size_t)g * p.gy) >> 3;
            if (p.id_scale) { p.id_scale += (size_t)g * p.id_gstride; p.id_shift += (size_t)g * p.id_gstride; }
            if (TP) { p.tp_y += (size_t)g * (p.gy >> 1); if (p.tp_code) p.tp_code += ((size_t)g * (p.gy >> 1)) >> 3; }
        }
        if (PF) {
            p.pf_a += (size_t)g * p.P * PF_C;
            if (p.pf_scale) { p.pf_scale += (size_t)g * p.pf_gs; p.pf_shift += (size_t)g * p.pf_gs; }
        }
        if (RES) {
            p.res_out += (size_t)g * p.gy;
            if (p.res_mask) p.res_mask += ((size_t)g * p.gy) >> 3;
            if (p.bn_z2) { p.bn_z2 += (size_t)g * p.gy; p.bn_vec2 += (size_t)g * 4 * p.Cout; p.stats2 += (size_t)g * ADAMML_STAT_SLOTS * 2 * p.Cout; }
        }
    }
    const int tid = threadIdx.x;
    float* s_vec = reinterpret_cast<float*>(smem + (MODE == 0 ? VEC_OFF : 0));
    const int nvec = CAT ? p.C2 : p.K;                            // entries of the loader's per-channel vectors
    const bool vec_lds = MODE == 0 && nvec <= VEC_MAXK && (DUAL || p.in_scale != nullptr);
    if (LZF) {
        // scale at [0, LZF_MAXK), shift at [LZF_MAXK, 2 LZF_MAXK); zero beyond K up to the K-step boundary (zero weights there, but
        // 0 * garbage must not make a NaN)
        const int kpad = (p.K + BK - 1) / BK * BK;
        for (int i = tid; i < kpad; i += NTHREADS) {
            s_vec[i] = i < p.K ? p.in_scale[i] : 0.f;
            s_vec[LZF_MAXK + i] = i < p.K ? p.in_shift[i] : 0.f;
        }
        __syncthreads();
    } else
    if (vec_lds) {
        if (DUAL) {
            for (int i = tid; i < 3 * p.K; i += NTHREADS) s_vec[i] = p.aff[i];
        } else {
            for (int i = tid; i < nvec; i += NTHREADS) { s_vec[i] = p.in_scale[i]; s_vec[nvec + i] = p.in_shift[i]; }
        }
        __syncthreads();
    }
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
    const int pgrp = bid / p.n_ctiles;
    const int c0 = ctile * BC;

    const int chunk = tid & 3;
    const int row_a = tid >> 2;             // 0..63 (+64 for second row)
    int* s_tapoff = reinterpret_cast<int*>(smem + STAGE_BYTES + CS_BYTES);    // MODE 1/3: element offset of each tap
    int* s_wtap = s_tapoff + 64;                                              // MODE 3: weight offset of each class tap [4]
    int* s_orow = s_wtap + 4;                                                 // MODE 3: output row of each tile pixel [BP]
    if (MODE == 1) {
        if (tid < p.KH * p.KW) {
            const int kh = tid / p.KW, kw = tid - kh * p.KW;
            s_tapoff[tid] = (kh * p.W + kw) * p.Cin;
        }
    }
    if (MODE == 3) {
        if (tid < p.cls_nt) {
            const unsigned code = (p.cls_code >> (8 * tid)) & 0xffu;
            s_tapoff[tid] = ((int)(code & 1) * p.W + (int)((code >> 1) & 1)) * p.Cin;
            s_wtap[tid] = (int)(code >> 2) * p.Cin;
        }
    }
    const bf16_t* wrow[WROWS];
    bool w_ok[WROWS];
#pragma unroll
    for (int r = 0; r < WROWS; ++r) {
        int co = c0 + row_a + r * 64;
        w_ok[r] = co < p.Cout;
        wrow[r] = p.w + (size_t)(w_ok[r] ? co : 0) * p.wK;
    }
    const int li = lane & 15, lg = lane >> 4;
    // epilogue thread mapping (fixed across the tiles of this workgroup, so the per-channel sums live in registers)
    constexpr int CROW = BC * 2 + 16;                  // LDS row stride (bytes): +16 B skews the banks
    static_assert(BP * CROW <= STAGE_BYTES, "epilogue tile must fit the staging buffers");
    float* cs = reinterpret_cast<float*>(smem + STAGE_BYTES);          // [NSR][2*BC] channel sums, live across the tile loop
    float* cs2 = reinterpret_cast<float*>(smem + CS2_OFF);             // [NSR][2*BC] sums of the second BatchNorm (RES only)
    const bool second = RES && p.bn_z2 != nullptr;
    const bool epi_bnz = EPI < 0 ? p.bn_z != nullptr : EPI == 1;
    const bool epi_acc = EPI < 0 ? p.accumulate != 0 : (EPI == 2 && p.accumulate != 0);
    if (p.stats) {
        for (int i = tid; i < NSR * 2 * BC; i += NTHREADS) cs[i] = 0.f;
        if (second)
            for (int i = tid; i < NSR * 2 * BC; i += NTHREADS) cs2[i] = 0.f;
    }
    constexpr int CPR = BC / 8;                        // 16-byte chunks per tile row
    constexpr int RSTEP = NTHREADS / CPR;              // rows covered per pass
    const int ech = tid % CPR, erow0 = tid / CPR;
    const int eco = c0 + ech * 8;

    // PF: product accumulators of this wave (64 cout x 32 a-channels of the workgroup's [128][64] block), live across the tile loop,
    // and the per-lane scale / shift of the transposed a fragment (8 pixels of ONE channel per lane: conv_wgrad_glds_kernel LZB)
    f32x4 pacc[PF ? 4 : 1][PF ? 2 : 1];
    float pfs[PF ? 2 : 1], pfh[PF ? 2 : 1];
    if constexpr (PF) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) pacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ch = wc * 32 + t * 16 + li;
            pfs[t] = p.pf_scale ? p.pf_scale[ch] : 1.f;
            pfh[t] = p.pf_scale ? p.pf_shift[ch] : 0.f;
        }
    }
    // a workgroup walks `tpb` consecutive pixel tiles of its cout tile: statistics are published once per workgroup
    for (int it = 0; it < p.tpb; ++it) {
    const int ptile = pgrp * p.tpb + it;
    if (ptile >= p.n_ptiles) break;
    const int p0 = ptile * BP;
    // TP: tile = (clip, pixel block); tile row r -> pixel (clip * TP + r / PPF) * Q + blk * PPF + r % PPF, valid while the pixel exists
    const int tp_clip = TP ? ptile / p.tp_nblk : 0;
    const int tp_q0 = TP ? (ptile - tp_clip * p.tp_nblk) * PPF : 0;
    auto tp_row = [&](int r, bool& ok) -> int {
        const int q = tp_q0 + (r % PPF);
        ok = q < p.tp_Q;
        return (tp_clip * TP + r / PPF) * p.tp_Q + (ok ? q : 0);
    };

    // ---- per-thread staging coordinates -------------------------------------------------------
    int a_n[2], a_h0[2], a_w0[2], a_base[2];
    unsigned long long a_mask[2];
    bool a_ok[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        int pp = p0 + row_a + r * 64;
        a_ok[r] = pp < p.P;
        int ppc = a_ok[r] ? pp : 0;
        if constexpr (TP != 0) {
            bool ok;
            ppc = tp_row(row_a + r * 64, ok);
            a_ok[r] = ok;
        }
        if (MODE == 0 && p.stride == 1 && p.pad == 0) {
            // 1x1 / stride 1 (pad 0): the input pixel IS the output pixel -- none of the two integer divisions below (~80 VALU
            // instructions per row, executed per tile AHEAD of the tile's first load)
            a_n[r] = a_h0[r] = a_w0[r] = 0;
            a_base[r] = ppc * p.Cin;
            a_mask[r] = 0;
            continue;
        }
        int n = ppc / (p.OH * p.OW);
        int rem = ppc - n * (p.OH * p.OW);
        int oh = rem / p.OW;
        int ow = rem - oh * p.OW;
        a_n[r] = n;
        a_h0[r] = oh * p.stride - p.pad;
        a_w0[r] = ow * p.stride - p.pad;
        a_base[r] = ((n * p.H + a_h0[r]) * p.W + a_w0[r]) * p.Cin;      // may point before the row start for padded taps
        unsigned long long m = 0;
        if (MODE == 1 && a_ok[r]) {
            for (int kh = 0; kh < p.KH; ++kh) {
                const bool hok = (unsigned)(a_h0[r] + kh) < (unsigned)p.H;
                for (int kw = 0; kw < p.KW; ++kw)
                    if (hok && (unsigned)(a_w0[r] + kw) < (unsigned)p.W) m |= 1ull << (kh * p.KW + kw);
            }
        }
        if (MODE == 3 && a_ok[r]) {
            for (int t = 0; t < p.cls_nt; ++t) {
                const unsigned code = (p.cls_code >> (8 * t)) & 0xffu;
                if ((unsigned)(a_h0[r] + (int)(code & 1)) < (unsigned)p.H && (unsigned)(a_w0[r] + (int)((code >> 1) & 1)) < (unsigned)p.W)
                    m |= 1ull << t;
            }
        }
        a_mask[r] = m;
    }
    if (MODE == 3 && tid < BP) {
        // scattered output rows of this tile (read by the epilogue, after at least one barrier)
        const int pp = p0 + tid;
        int orow = 0;
        if (pp < p.P) {
            const int n = pp / (p.OH * p.OW), rem = pp - n * (p.OH * p.OW);
            const int i = rem / p.OW, j = rem - i * p.OW;
            orow = (n * p.oH + 2 * i + p.o_ph) * p.oW + 2 * j + p.o_pw;
        }
        s_orow[tid] = orow;
    }
    if ((MODE == 1 || MODE == 3) && it == 0) __syncthreads();          // tap-offset tables visible

    // EID: identity-side operands of this tile's epilogue (every lane loads -- clamped addresses -- so that the number of
    // outstanding VMEM operations is the same for every wave: the LDS-DMA loop below waits with counted vmcnt)
    constexpr int NRE = BP / RSTEP;
    bf16x8 eid[EID ? NRE : 1];
    unsigned embits[EID ? NRE : 1];
    auto issue_eid = [&]() {
        if constexpr (PF) {
            // a tile [128 pixels][64 channels] raw, as four [32][64] transpose-read images (source-side swizzle, conv_wgrad_glds_kernel):
            // LDS chunk e = l * 256 + tid -> row e / 8, position e % 8 holds source chunk position ^ swizzle(row % 32)
            const bf16_t* zeros = reinterpret_cast<const bf16_t*>(g_zero_page);
            const unsigned pbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + PF_OFF + wave * 1024;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int e = tid + l * NTHREADS;
                const int row = e >> 3;
                const int ch = (e & 7) ^ (tr_swz<64>(row & 31) >> 1);
                const bf16_t* src = (p0 + row < p.P) ? p.pf_a + (size_t)(p0 + row) * PF_C + ch * 8 : zeros;
                glds16(src, __builtin_amdgcn_readfirstlane(pbase + l * NTHREADS * 16));
            }
        }
        if constexpr (EID != 0) {
            const int ecoc = eco < p.Cout ? eco : 0;
#pragma unroll
            for (int j = 0; j < NRE; ++j) {
                const int r = erow0 + j * RSTEP;
                size_t pre = (size_t)(p0 + r < p.P ? p0 + r : p0) * p.Cout + ecoc;
                if constexpr (TP != 0) {
                    bool ok;
                    pre = (size_t)tp_row(r, ok) * p.Cout + ecoc;
                }
                if (FADD) eid[j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p.res_out + pre));
                else {
                    eid[j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p.y + pre));
                    embits[j] = p.res_mask[pre >> 3];
                }
            }
        }
    };
    constexpr int NEID = (EID ? (FADD ? NRE : 2 * NRE) : 0) + (PF ? 4 : 0);       // VMEM loads issue_eid() puts in flight per thread

    // register prefetch ring of depth PD: global loads run PD K-steps ahead of the MFMAs.  One K step of compute is
    // ~0.15 us but an HBM round trip is 1-2 us, so a one-step look-ahead left the kernel latency-bound.
    // UNC (the deep rings): every ring load is issued UNCONDITIONALLY from a clamped (always valid) address and zeroed on its way into
    // LDS.  Behind `if (ok)` each load sat in its own exec-masked block; the compiler's wait-count bookkeeping must then assume that none
    // of the younger loads was issued and waits for a slot with vmcnt(0) -- at the top of every PD K steps the whole ring drained and the
    // most recent request was exposed as a full HBM round trip (the ring was one step deep in effect).  Unconditional loads make the
    // counts exact: the wait for a slot leaves the (PD - 1) younger slots in flight.
    // Only the CAT / DUAL instances: the plain deep-ring instances (layer 3-4 shapes, small grids) measured 5-7 % SLOWER with it.
    constexpr bool UNC = PD > 1 && (CAT || DUAL);
    bf16x8 ra[PD][2], rw[PD][WROWS];
    bf16x8 ra2[DUAL ? PD : 1][2];
    int rci[PD];
    bool rav[PD][2];
    bool rkok[UNC ? PD : 1];
#pragma unroll
    for (int i = 0; i < PD; ++i) rci[i] = 0;

    auto issue_loads = [&](auto slot_c, int kt) {
        constexpr int SL = decltype(slot_c)::value;
        const int k = kt * BK + chunk * 8;
        const bool kok = k < p.K;
        int kw_off = k;                                  // offset of this K chunk in a weight row
        if (MODE == 0) {
            const bool from_b = CAT && kt * BK >= p.K1;                 // wave-uniform: a K step lies in one source (K1 % BK == 0)
            rci[SL] = from_b ? k - p.K1 : (CAT ? -1 : k);               // < 0: no lazy transform for this step
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool ok = a_ok[r] && kok;
                rav[SL][r] = ok;
                if constexpr (UNC) {
                    const bf16_t* src = p.x + (size_t)(unsigned)(a_base[r] + k);
                    if (CAT && from_b) src = p.xb + (size_t)(unsigned)(a_base[r] / p.K1) * p.C2 + (k - p.K1);
                    ra[SL][r] = *reinterpret_cast<const bf16x8*>(ok ? src : p.x);
                    if (DUAL) ra2[DUAL ? SL : 0][r] = *reinterpret_cast<const bf16x8*>(ok ? p.x2 + (size_t)(unsigned)(a_base[r] + k) : p.x2);
                    continue;
                }
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (CAT && from_b) {
                    if (ok) v = *reinterpret_cast<const bf16x8*>(p.xb + (size_t)(unsigned)(a_base[r] / p.K1) * p.C2 + (k - p.K1));
                } else if (ok) v = *reinterpret_cast<const bf16x8*>(p.x + (size_t)(unsigned)(a_base[r] + k));
                ra[SL][r] = v;
                if (DUAL) {
                    bf16x8 v2 = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (ok) v2 = *reinterpret_cast<const bf16x8*>(p.x2 + (size_t)(unsigned)(a_base[r] + k));
                    ra2[DUAL ? SL : 0][r] = v2;
                }
            }
        } else if (MODE == 1 || MODE == 3) {
            const int tap = kok ? (k >> p.cin_shift) : 0;
            const int ci = k & (p.Cin - 1);
            const int toff = s_tapoff[tap] + ci;
            if (MODE == 3) kw_off = s_wtap[tap] + ci;
            rci[SL] = ci;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool ok = kok && ((a_mask[r] >> tap) & 1ull);
                rav[SL][r] = ok;
                if constexpr (UNC) {
                    ra[SL][r] = *reinterpret_cast<const bf16x8*>(ok ? p.x + (size_t)(unsigned)(a_base[r] + toff) : p.x);
                    continue;
                }
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) v = *reinterpret_cast<const bf16x8*>(p.x + (size_t)(unsigned)(a_base[r] + toff));
                ra[SL][r] = v;
            }
        } else {
            const int tap = k >> p.cin_shift;
            const int ci = k - (tap << p.cin_shift);
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            rci[SL] = ci;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                int ih = a_h0[r] + kh, iw = a_w0[r] + kw;
                bool ok = a_ok[r] && kok && ih >= 0 && iw >= 0;
                ok = ok && ((ih | iw) & (p.up - 1)) == 0;
                ih >>= p.up_shift;
                iw >>= p.up_shift;
                ok = ok && ih < p.H && iw < p.W;
                rav[SL][r] = ok;
                if constexpr (UNC) {
                    ra[SL][r] = *reinterpret_cast<const bf16x8*>(ok ? p.x + ((size_t)(a_n[r] * p.H + ih) * p.W + iw) * p.Cin + ci : p.x);
                    continue;
                }
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) v = *reinterpret_cast<const bf16x8*>(p.x + ((size_t)(a_n[r] * p.H + ih) * p.W + iw) * p.Cin + ci);
                ra[SL][r] = v;
            }
        }
#pragma unroll
        for (int r = 0; r < WROWS; ++r) {
            if constexpr (UNC) {
                rw[SL][r] = *reinterpret_cast<const bf16x8*>(wrow[r] + (kok ? kw_off : 0));       // (wrow is row 0 for rows past Cout)
                continue;
            }
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (w_ok[r] && kok) v = *reinterpret_cast<const bf16x8*>(wrow[r] + kw_off);
            rw[SL][r] = v;
        }
        if constexpr (UNC) rkok[SL] = kok;
    };
    auto store_tile = [&](auto slot_c, int buf) {
        constexpr int SL = decltype(slot_c)::value;
        char* base = smem + buf * TILE_BYTES;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            bf16x8 v = ra[SL][r];
            if (DUAL) {
                if (rav[SL][r]) {
                    const f32x8 gv = bf8_to_f32(v), zv = bf8_to_f32(ra2[DUAL ? SL : 0][r]);
                    f32x8 ca, cb, cc;
                    if (vec_lds) { ca = load_f32x8(s_vec + rci[SL]); cb = load_f32x8(s_vec + p.K + rci[SL]); cc = load_f32x8(s_vec + 2 * p.K + rci[SL]); }
                    else { ca = load_f32x8(p.aff + rci[SL]); cb = load_f32x8(p.aff + p.K + rci[SL]); cc = load_f32x8(p.aff + 2 * p.K + rci[SL]); }
                    f32x8 o;
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = fmaf(ca[i], gv[i], fmaf(cb[i], zv[i], cc[i]));
                    v = f32_to_bf8(o);
                    if (p.side && ctile == 0) *reinterpret_cast<bf16x8*>(p.side + (size_t)(unsigned)(a_base[r] + rci[SL])) = v;
                }
            } else if (p.in_scale && rav[SL][r] && (!CAT || rci[SL] >= 0)) {
                if (vec_lds) {
                    const f32x8 sc = load_f32x8(s_vec + rci[SL]), sh = load_f32x8(s_vec + nvec + rci[SL]);
                    const float lo = act_lo(p.act), hi = act_hi(p.act);
                    f32x8 f = bf8_to_f32(v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] = clamp_act(fmaf(f[i], sc[i], sh[i]), lo, hi);
                    v = f32_to_bf8(f);
                } else v = f32_to_bf8(transform8(v, p.in_scale, p.in_shift, rci[SL], p.act));
            }
            if (UNC && !rav[SL][r]) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<bf16x8*>(base + lds_off(row_a + r * 64, chunk)) = v;
        }
#pragma unroll
        for (int r = 0; r < WROWS; ++r) {
            bf16x8 v = rw[SL][r];
            if (UNC && !(w_ok[r] && rkok[UNC ? SL : 0])) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<bf16x8*>(base + BP * 64 + lds_off(row_a + r * 64, chunk)) = v;
        }
    };

    f32x4 acc[WCT][4];
#pragma unroll
    for (int i = 0; i < WCT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    const bool tail_rows = TP ? tp_q0 + PPF > p.tp_Q : p0 + BP > p.P;      // (uniform) this tile has rows without a pixel
    auto compute = [&](int buf, int kt = 0) {
        const char* base = smem + buf * TILE_BYTES;
        bf16x8 fa[4], fw[WCT];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            fa[t] = *reinterpret_cast<const bf16x8*>(base + lds_off(wp * 64 + t * 16 + li, lg));
        if constexpr (LZF) {
            // fragment lane (pixel li of tile t, K chunk lg) = 8 consecutive input channels kt*32 + lg*8 ..+7 of one pixel
            const f32x8 sc = load_f32x8(s_vec + kt * BK + lg * 8), sh = load_f32x8(s_vec + LZF_MAXK + kt * BK + lg * 8);
            const float lo = act_lo(p.act), hi = act_hi(p.act);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x8 f = bf8_to_f32(fa[t]);
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] = clamp_act(fmaf(f[i], sc[i], sh[i]), lo, hi);
                fa[t] = f32_to_bf8(f);
            }
            if (tail_rows) {                              // zero-filled rows must stay zero (statistics, never-stored outputs)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int r = wp * 64 + t * 16 + li;
                    if (TP ? tp_q0 + (r % PPF) >= p.tp_Q : p0 + r >= p.P) fa[t] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
        }
#pragma unroll
        for (int t = 0; t < WCT; ++t)
            fw[t] = *reinterpret_cast<const bf16x8*>(base + BP * 64 + lds_off(wc * (BC / 2) + t * 16 + li, lg));
#pragma unroll
        for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
                acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ct], fa[pt], acc[ct][pt], 0, 0, 0);
    };
    if constexpr (GLDS) {
        // LDS-DMA staging: position `chunk` of an LDS row holds source chunk chunk ^ swizzle(row) (lds_off is an involution)
        const bf16_t* zeros = reinterpret_cast<const bf16_t*>(g_zero_page);
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 1024;
        const int kc8 = (chunk ^ (((row_a >> 3) & 1) << 1)) * 8;
        auto glds_stage = [&](int kt, int buf) {
            const int k = kt * BK + kc8;
            const bool kok = k < p.K;
            const unsigned base = lds0 + buf * TILE_BYTES;
            int kw_off = k;
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const bf16_t* src = (a_ok[r] && kok) ? p.x + (size_t)(unsigned)(a_base[r] + k) : zeros;
                    glds16(src, __builtin_amdgcn_readfirstlane(base + r * 4096));
                }
            } else {
                const int tap = kok ? (k >> p.cin_shift) : 0;
                const int ci = k & (p.Cin - 1);
                const int toff = s_tapoff[tap] + ci;
                if (MODE == 3) kw_off = s_wtap[tap] + ci;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const bool ok = kok && ((a_mask[r] >> tap) & 1ull);
                    const bf16_t* src = ok ? p.x + (size_t)(unsigned)(a_base[r] + toff) : zeros;
                    glds16(src, __builtin_amdgcn_readfirstlane(base + r * 4096));
                }
            }
#pragma unroll
            for (int r = 0; r < WROWS; ++r) {
                const bf16_t* src = (w_ok[r] && kok) ? wrow[r] + kw_off : zeros;
                glds16(src, __builtin_amdgcn_readfirstlane(base + BP * 64 + r * 4096));
            }
        };
        constexpr int PER = 2 + WROWS;                       // LDS-DMAs of this thread per K step
        glds_stage(0, 0);
        if (nk > 1) glds_stage(1, 1);
        issue_eid();                                         // in flight behind steps 0 and 1, ahead of steps 2..
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            // step kt landed; step kt + 1 may fly -- and, while the steps waited for are OLDER than the EID loads (kt <= 1), so may they
            if (NEID && kt <= 1) {
                if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER + NEID) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEID) : "memory");
            } else
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");       // step kt landed, step kt + 1 may fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // ... for every wave; and everyone is past compute(kt - 1)
            if (kt + 2 < nk) glds_stage(kt + 2, buf == 0 ? 2 : buf - 1);
            compute(buf, kt);
            buf = buf == 2 ? 0 : buf + 1;
        }
    } else {
    int kt0 = 0;
    if (UNC && nk >= 2 * PD) {                           // (uniform)
        // steady state of a deep ring, with NO run-time condition around a load: prologue and every step issue, so the compiler's
        // wait counts are exact (a load issued under a condition -- even a uniform one -- forces every wait that follows the join to
        // assume it was not issued, i.e. vmcnt(0..3) for a slot instead of vmcnt((PD - 1) x loads per slot))
        static_for<PD>([&](auto sc) { issue_loads(sc, (int)decltype(sc)::value); });
        for (; kt0 + 2 * PD <= nk; kt0 += PD) {
            static_for<PD>([&](auto sc) {
                const int kt = kt0 + (int)decltype(sc)::value;
                store_tile(sc, kt & 1);
                issue_loads(sc, kt + PD);
                __syncthreads();
                compute(kt & 1);
            });
        }
    } else {
    // prologue: fill the ring
    static_for<PD>([&](auto sc) {
        if ((int)decltype(sc)::value < nk) issue_loads(sc, (int)decltype(sc)::value);
    });
    }
    issue_eid();
    for (; kt0 < nk; kt0 += PD) {
        static_for<PD>([&](auto sc) {
            const int kt = kt0 + (int)decltype(sc)::value;
            if (kt < nk) {                               // uniform
                store_tile(sc, kt & 1);                  // waits (counted vmcnt) only for this slot's loads
                if (kt + PD < nk) issue_loads(sc, kt + PD);
                __syncthreads();                         // tile kt visible; everyone is past compute(kt-1)
                compute(kt & 1);
            }
        });
    }
    }
    __syncthreads();                                     // operand tiles consumed: the epilogue reuses the LDS

    // ---- epilogue: stage the bf16 tile through LDS, store 16 B per lane fully coalesced, and accumulate the
    //      per-channel sum / sum-of-squares of the stored (rounded) values on the way out -------------------
    // (the last K step ended with __syncthreads(): every wave is done reading the operand tiles)
    f32x8 esum, esq, esq2;
#pragma unroll
    for (int i = 0; i < 8; ++i) esum[i] = esq[i] = esq2[i] = 0.f;
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int ct = 0; ct < WCT; ++ct) {
            const int prow = wp * 64 + pt * 16 + li;
            const int ccol = wc * (BC / 2) + ct * 16 + lg * 4;
            *reinterpret_cast<bf16x4*>(smem + prow * CROW + ccol * 2) = f32_to_bf4(acc[ct][pt]);
        }
    __syncthreads();
    if (FADD) {
        // out = act(scale * z + shift + identity): z is the bf16-rounded tile staged above (the value the unfused path would
        // have stored and re-read), the identity operand streams in once, the block output and its activation mask stream
        // out once.  All identity loads of a batch of rows are issued before the first store (in-order vmcnt, see RES).
        if (eco < p.Cout) {
            const f32x8 sc = load_f32x8(p.bn_vec + eco), sh = load_f32x8(p.bn_vec + p.Cout + eco);
            f32x8 isc, ish;
#pragma unroll
            for (int i = 0; i < 8; ++i) { isc[i] = 1.f; ish[i] = 0.f; }
            if (p.id_scale) { isc = load_f32x8(p.id_scale + eco); ish = load_f32x8(p.id_shift + eco); }
            const float rlo = act_lo(p.res_act), rhi = act_hi(p.res_act);
            constexpr int NR = BP / RSTEP, EB = EID ? NR : (NR < 4 ? NR : 4);
            if constexpr (TP != 0) {
                // this thread's NR = 8 rows are rows erow0 + 16 j: NPX = PPF / 16 pixels x TP frames (frame t of pixel s at j = t * NPX + s)
                static_assert(RSTEP == 16 && NR == 8, "TP epilogue mapping");
                constexpr int NPX = PPF / 16, To = TP / 2;
                bf16x8 vb[NR];
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int r = erow0 + j * RSTEP;
                    f32x8 f = bf8_to_f32(*reinterpret_cast<const bf16x8*>(smem + r * CROW + ech * 16));
                    const f32x8 w = bf8_to_f32(eid[j]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] = clamp_act(fmaf(f[i], sc[i], sh[i]) + fmaf(w[i], isc[i], ish[i]), rlo, rhi);
                    vb[j] = f32_to_bf8(f);                                  // the value the unfused path stores and the pool re-reads
                }
#pragma unroll
                for (int s = 0; s < NPX; ++s) {
                    const int q = tp_q0 + erow0 + 16 * s;
                    if (q >= p.tp_Q) continue;
#pragma unroll
                    for (int to = 0; to < To; ++to) {
                        f32x8 best;
                        unsigned code = 0;
#pragma unroll
                        for (int i = 0; i < 8; ++i) best[i] = -INFINITY;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const int tt = 2 * to - 1 + k;
                            if (tt < 0 || tt >= TP) continue;
                            const f32x8 v = bf8_to_f32(vb[tt * NPX + s]);
#pragma unroll
                            for (int i = 0; i < 8; ++i)
                                if (v[i] > best[i]) { best[i] = v[i]; code = (code & ~(3u << (2 * i))) | ((unsigned)k << (2 * i)); }   // first maximum in scan order
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (!(best[i] > rlo && best[i] < rhi)) code |= 3u << (2 * i);       // act'(maximum) == 0: no gradient through this window
                        const size_t po = ((size_t)(tp_clip * To + to) * p.tp_Q + q) * p.Cout + eco;
                        *reinterpret_cast<bf16x8*>(p.tp_y + po) = f32_to_bf8(best);
                        if (p.tp_code) p.tp_code[po >> 3] = (uint16_t)code;
                    }
                }
            } else
#pragma unroll
            for (int b0 = 0; b0 < NR; b0 += EB) {
                bf16x8 ir[EB];
                size_t pr[EB];
                bool ok[EB];
#pragma unroll
                for (int j = 0; j < EB; ++j) {
                    const int r = erow0 + (b0 + j) * RSTEP;
                    ok[j] = p0 + r < p.P;
                    pr[j] = (size_t)(ok[j] ? p0 + r : p0) * p.Cout + eco;
                    if constexpr (EID != 0) ir[j] = eid[b0 + j];                 // requested at the start of the tile
                    else if (p.res_out) ir[j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p.res_out + pr[j]));
                }
#pragma unroll
                for (int j = 0; j < EB; ++j) {
                    const int r = erow0 + (b0 + j) * RSTEP;
                    f32x8 f = bf8_to_f32(*reinterpret_cast<const bf16x8*>(smem + r * CROW + ech * 16));
                    if (p.res_out) {
                        const f32x8 w = bf8_to_f32(ir[j]);
#pragma unroll
                        for (int i = 0; i < 8; ++i) f[i] = clamp_act(fmaf(f[i], sc[i], sh[i]) + fmaf(w[i], isc[i], ish[i]), rlo, rhi);
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) f[i] = clamp_act(fmaf(f[i], sc[i], sh[i]), rlo, rhi);
                    }
                    const bf16x8 v = f32_to_bf8(f);
                    if (ok[j]) {
                        *reinterpret_cast<bf16x8*>(p.y + pr[j]) = v;
                        if (p.mask_out) {
                            const f32x8 q = bf8_to_f32(v);
                            unsigned bits = 0;
#pragma unroll
                            for (int i = 0; i < 8; ++i) bits |= (q[i] > rlo && q[i] < rhi) ? (1u << i) : 0u;
                            p.mask_out[pr[j] >> 3] = (uint8_t)bits;
                        }
                    }
                }
            }
        }
    } else if (RES) {
        // Residual form: y already holds the identity-path gradient (accumulate), the mask is act'(block output), and a
        // second BatchNorm operand of the add (downsample branch) gets its sum(g' * zhat2) in the same pass.  Rows are
        // processed in batches of EB: ALL global loads of a batch (z, identity-path gradient, block output, second z) are
        // issued before the first store.  Loads and stores retire in order through vmcnt on this ISA, so a row-at-a-time
        // loop (load, mask, store, next row) exposes one HBM round trip per row, and this epilogue is the whole kernel
        // (K = 64..512: a few MFMA steps per tile).
        if (eco < p.Cout) {
        // bn_z == nullptr: only sum(g') is accumulated (the algebraic BatchNorm backward derives sum(g' zhat) from g'^T a)
        f32x8 mu, is;
        if (p.bn_z) { mu = load_f32x8(p.bn_vec + 2 * p.Cout + eco); is = load_f32x8(p.bn_vec + 3 * p.Cout + eco); }
        f32x8 mu2, is2;
        if (second) { mu2 = load_f32x8(p.bn_vec2 + 2 * p.Cout + eco); is2 = load_f32x8(p.bn_vec2 + 3 * p.Cout + eco); }
        const float rlo = act_lo(p.res_act), rhi = act_hi(p.res_act);
        constexpr int NR = BP / RSTEP, EB = EID ? NR : (NR < 4 ? NR : 4);
#pragma unroll
        for (int b0 = 0; b0 < NR; b0 += EB) {
            bf16x8 zr[EID ? 1 : EB], dr[EB], orr[EID ? 1 : EB], z2r[EID ? 1 : EB];
            unsigned mbits[EB];
            size_t pr[EB];
            bool ok[EB];
#pragma unroll
            for (int j = 0; j < EB; ++j) {
                const int r = erow0 + (b0 + j) * RSTEP;
                ok[j] = p0 + r < p.P;
                pr[j] = (size_t)(ok[j] ? p0 + r : p0) * p.Cout + eco;       // clamped: out-of-range rows re-read row p0, never stored
                if constexpr (EID != 0) {
                    // (the launcher selects EID only for: accumulate, 1-bit mask, no z operands -- the algebraic backward's form)
                    dr[j] = eid[b0 + j];
                    mbits[j] = embits[b0 + j];
                } else {
                if (p.bn_z) zr[j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p.bn_z + pr[j]));
                if (p.accumulate) dr[j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p.y + pr[j]));
                if (p.res_mask) mbits[j] = p.res_mask[pr[j] >> 3];
                else orr[j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p.res_out + pr[j]));
                if (second) z2r[j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p.bn_z2 + pr[j]));
                }
            }
#pragma unroll
            for (int j = 0; j < EB; ++j) {
                const int r = erow0 + (b0 + j) * RSTEP;
                f32x8 f = bf8_to_f32(*reinterpret_cast<const bf16x8*>(smem + r * CROW + ech * 16));
                if (EID || p.accumulate) f += bf8_to_f32(dr[j]);
                if (EID || p.res_mask) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] = (mbits[j] >> i) & 1u ? f[i] : 0.f;
                } else {
                    const f32x8 ov = bf8_to_f32(orr[EID ? 0 : j]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] *= mask_act(ov[i], rlo, rhi);
                }
                const bf16x8 v = f32_to_bf8(f);
                if (ok[j]) *reinterpret_cast<bf16x8*>(p.y + pr[j]) = v;
                if constexpr (PF)          // g' replaces the raw GEMM tile in LDS (zero rows past the last pixel): A operand of the product
                    *reinterpret_cast<bf16x8*>(smem + r * CROW + ech * 16) = ok[j] ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                const float keep = ok[j] ? 1.f : 0.f;
                f = bf8_to_f32(v) * keep;
                esum += f;
                if constexpr (EID == 0) {
                if (p.bn_z) {
                    const f32x8 zv = bf8_to_f32(zr[j]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) esq[i] += f[i] * (zv[i] - mu[i]) * is[i];
                }
                if (second) {
                    const f32x8 z2 = bf8_to_f32(z2r[j]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) esq2[i] += f[i] * (z2[i] - mu2[i]) * is2[i];
                }
                }
            }
        }
        }
    } else if (eco < p.Cout && epi_bnz) {
        // data gradient w.r.t. a lazily normalised tensor: apply the activation mask here, store g' and accumulate
        // sum(g') and sum(g' * zhat) -- the BatchNorm-backward reduction pass never has to re-read g and z
        const f32x8 sc = load_f32x8(p.bn_vec + eco), sh = load_f32x8(p.bn_vec + p.Cout + eco);
        const f32x8 mu = load_f32x8(p.bn_vec + 2 * p.Cout + eco), is = load_f32x8(p.bn_vec + 3 * p.Cout + eco);
        f32x8 cadd;
#pragma unroll
        for (int i = 0; i < 8; ++i) cadd[i] = 0.f;
        if (CAT) cadd = load_f32x8(p.epi_add + eco);
        // EPI 1: the z rows of a batch of rows are requested before the batch's first store (clamped addresses).  The compiler cannot
        // move a z load above a store to y, and loads / stores retire in order through vmcnt: row at a time, every row exposes one
        // HBM round trip.
        constexpr int NRZ = BP / RSTEP, EBZ = (EPI == 1 || CAT || DUAL) ? (NRZ < 4 ? NRZ : 4) : 1;     // (CAT / DUAL: 2 workgroups per CU, registers to spare)
#pragma unroll 1
        for (int b0 = 0; b0 < NRZ; b0 += EBZ) {                    // (batches not unrolled: less address arithmetic hoisted into registers)
            bf16x8 zrow[EBZ];
#pragma unroll
            for (int j = 0; j < EBZ; ++j) {
                const int r = erow0 + (b0 + j) * RSTEP;
                const int rc = p0 + r < p.P ? r : 0;
                const size_t pp = MODE == 3 ? (size_t)s_orow[rc] : (size_t)(p0 + rc);
                zrow[j] = *reinterpret_cast<const bf16x8*>(p.bn_z + pp * p.Cout + eco);
            }
#pragma unroll
            for (int j = 0; j < EBZ; ++j) {
                const int r = erow0 + (b0 + j) * RSTEP;
                if (p0 + r < p.P) {
                    const size_t pp = MODE == 3 ? (size_t)s_orow[r] : (size_t)(p0 + r);
                    f32x8 f = bf8_to_f32(*reinterpret_cast<const bf16x8*>(smem + r * CROW + ech * 16)) + cadd;
                    const f32x8 zv = bf8_to_f32(zrow[j]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] *= act_mask(fmaf(zv[i], sc[i], sh[i]), p.bn_act);
                    const bf16x8 v = f32_to_bf8(f);
                    *reinterpret_cast<bf16x8*>(p.y + pp * p.Cout + eco) = v;
                    f = bf8_to_f32(v);
                    esum += f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) esq[i] += f[i] * (zv[i] - mu[i]) * is[i];
                }
            }
        }
    } else if (eco < p.Cout) {
        // (EPI 2: the rows of the tensor accumulated into are requested per batch of rows, as above)
        constexpr int NRA = BP / RSTEP, EBA = (EPI == 2 || CAT || DUAL) ? (NRA < 4 ? NRA : 4) : 1;
#pragma unroll 1
        for (int b0 = 0; b0 < NRA; b0 += EBA) {
            bf16x8 arow[EBA];
            if (epi_acc) {
#pragma unroll
                for (int j = 0; j < EBA; ++j) {
                    const int r = erow0 + (b0 + j) * RSTEP;
                    const int rc = p0 + r < p.P ? r : 0;
                    const size_t pp = MODE == 3 ? (size_t)s_orow[rc] : (size_t)(p0 + rc);
                    arow[j] = *reinterpret_cast<const bf16x8*>(p.y + pp * p.Cout + eco);
                }
            }
#pragma unroll
            for (int j = 0; j < EBA; ++j) {
                const int r = erow0 + (b0 + j) * RSTEP;
                if (p0 + r < p.P) {
                    const size_t pp = MODE == 3 ? (size_t)s_orow[r] : (size_t)(p0 + r);
                    bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + r * CROW + ech * 16);
                    f32x8 f = bf8_to_f32(v);
                    if (CAT) {
                        f += load_f32x8(p.epi_add + eco);
                        v = f32_to_bf8(f);
                    }
                    if (epi_acc) {
                        f += bf8_to_f32(arow[j]);
                        v = f32_to_bf8(f);
                    }
                    *reinterpret_cast<bf16x8*>(p.y + pp * p.Cout + eco) = v;
                }
            }
        }
    }
    if (p.stats && !epi_bnz && !RES) {
        // Forward statistics on the (otherwise ~90 % idle) matrix cores instead of the VALU: with F = the staged bf16 tile
        // [32 pixels][16 channels] as an MFMA fragment (hardware transpose read), ones * F gives the per-channel sums and
        // F^T F the Gram matrix whose diagonal is the per-channel sum of squares -- exact products of the STORED values,
        // fp32 accumulation.  The VALU form (convert + add + fma per element) cost 30 % on the HBM-bound 1x1 layers.
        // Rows past P and channels past Cout hold zeros (their operands were zero-filled).
        constexpr int NCB = BC / 64;                       // 16-channel blocks per wave
        union { s16x4 h[2]; bf16x8 v; } ones;
        ones.h[0] = s16x4{0x3F80, 0x3F80, 0x3F80, 0x3F80};
        ones.h[1] = ones.h[0];
        const int trow = 8 * lg + (li >> 2);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int ch0 = (wave * NCB + cb) * 16;
            f32x4 dsum = {0.f, 0.f, 0.f, 0.f}, dsq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ps = 0; ps < BP / 32; ++ps) {
                const char* fp = smem + (ps * 32 + trow) * CROW + (ch0 + 4 * (li & 3)) * 2;
                union { s16x4 h[2]; bf16x8 v; } f;
                f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(fp));
                f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(fp + 4 * CROW));
                dsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, f.v, dsum, 0, 0, 0);
                dsq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.v, f.v, dsq, 0, 0, 0);
            }
            // D rows = lg*4 + r, column = li: row 0 of dsum is held by lanes 0..15; the diagonal of dsq by lanes with li>>2 == lg
            if (lg == 0) cs[ch0 + li] += dsum[0];
            if ((li >> 2) == lg) {
                const int r = li & 3;
                cs[BC + ch0 + li] += r == 0 ? dsq[0] : r == 1 ? dsq[1] : r == 2 ? dsq[2] : dsq[3];
            }
        }
    }
    if (p.stats && (epi_bnz || RES)) {
        // Lanes l, l+CPR, l+2*CPR.. of a wave hold partial sums of the same channel chunk.  They are folded on the VALU
        // with the gfx950 lane-swap instructions (v_permlane32_swap / v_permlane16_swap: "swap the upper half (odd rows)
        // of a with the lower half (even rows) of b", so a' + b' folds TWO values at once and halves the register count
        // per step) -- 24 VALU ops instead of 32-48 ds_bpermute, which had made this epilogue LDS-pipe-bound (-30 % on
        // the HBM-bound 1x1 layers).  16 values -> 4 registers per lane, then 4 LDS adds per lane, into this WAVE's row of the sums (one
        // owner lane per entry: the adds of an entry happen in tile order, whatever the timing of the other waves).
        auto fold = [&](const f32x8& s1, const f32x8& s2, float* dst) {
            dst += (NSR > 1 ? wave : 0) * 2 * BC;
            float v[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i] = s1[i]; v[8 + i] = s2[i]; }
            if (CPR == 8) {
#pragma unroll
                for (int i = 0; i < 16; ++i)              // lanes l and l+8 of a 16-lane row: DPP row_ror:8
                    v[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[i]), 0x128, 0xf, 0xf, false));
            }
            float u[8], wv[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // (inline asm: with this toolchain the __builtin_amdgcn_permlane32_swap result pair is mis-compiled when both
                // halves feed a float add -- verified on hardware, tools/scratch/fold.hip; s_nop covers the VALU->permlane hazard)
                float a = v[2 * i], b = v[2 * i + 1];
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
                u[i] = a + b;                                   // rows 0,1: v[2i]; rows 2,3: v[2i+1]
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = u[2 * j], b = u[2 * j + 1];
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
                wv[j] = a + b;                                  // row r holds value 4j + {0,2,1,3}[r]
            }
            const int lrow = lane >> 4;
            const int vsel = ((lrow & 1) << 1) | (lrow >> 1);
            if (eco < p.Cout && (CPR >= 16 || !(lane & 8))) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int id = 4 * j + vsel;              // 0..7: sum of channel id; 8..15: second moment of channel id-8
                    atomicAdd(&dst[(id >> 3) * BC + ech * 8 + (id & 7)], wv[j]);
                }
            }
        };
        static_assert(NSR == 4 || EPI == 0 || FADD, "VALU-folded sums need one row per wave");
        fold(esum, esq, cs);
        if (second) fold(esum, esq2, cs2);
    }
    if constexpr (PF) {
        // P[cout][c] += sum over the tile's 128 pixels of g'[p][cout] * a[p][c]: both operands pixel-major in LDS -> transpose reads.
        // g' tile: rows of CROW bytes (the padded epilogue image, as the MFMA statistics read it); a tile: four swizzled [32][64] images
        // staged raw by LDS-DMA at the start of the tile, BatchNorm + activation applied to the fragment (one channel per lane).
        __syncthreads();                               // every row of g' is back in LDS (and the a tile landed: vmcnt(0) of the last K step)
        const int trow = 8 * lg + (li >> 2), tq = li & 3;
        const int x_lo = tr_swz<64>(trow), x_hi = tr_swz<64>(trow + 4);
        const float plo = p.pf_scale ? act_lo(p.pf_act) : -INFINITY, phi = p.pf_scale ? act_hi(p.pf_act) : INFINITY;
#pragma unroll
        for (int ks = 0; ks < BP / 32; ++ks) {
            bf16x8 fa[4], fb[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const char* fp = smem + (ks * 32 + trow) * CROW + (wp * 64 + t * 16 + 4 * tq) * 2;
                union { s16x4 h[2]; bf16x8 v; } f;
                f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(fp));
                f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(fp + 4 * CROW));
                fa[t] = f.v;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const char* base = smem + PF_OFF + ks * 32 * 128;
                const int u = (wc * 32 + t * 16) / 4 + tq;
                union { s16x4 h[2]; bf16x8 v; } f;
                f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + trow * 128 + ((u ^ x_lo) << 3)));
                f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + (trow + 4) * 128 + ((u ^ x_hi) << 3)));
                f32x8 q = bf8_to_f32(f.v);
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = clamp_act(fmaf(q[i], pfs[t], pfh[t]), plo, phi);
                fb[t] = f32_to_bf8(q);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) pacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb[nt], pacc[mt][nt], 0, 0, 0);
        }
    }
    __syncthreads();                                   // staging tile consumed before the next tile's operands land
    }   // tile loop
    if constexpr (PF) {
        // one partial [BC][64] per workgroup (plain stores; summed in a fixed order by wgrad_reduce_kernel): wave (wp, wc) holds rows
        // wp*64 + mt*16 + lg*4 + r (cout within the tile), column wc*32 + nt*16 + li
        float* out = p.pf_ws + (((size_t)blockIdx.y * p.n_ctiles + ctile) * p.pf_nsplit + pgrp) * (BC * PF_C);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(wp * 64 + mt * 16 + lg * 4 + r) * PF_C + wc * 32 + nt * 16 + li] = pacc[mt][nt][r];
    }
    if (p.stats) {
        // one publication per workgroup and channel: the wave rows are folded in wave order, the workgroup partial goes to the exact
        // integer bins (common.h)
        __syncthreads();
        const unsigned slot = blockIdx.x & (ADAMML_STAT_SLOTS - 1);
        for (int i = tid; i < 2 * BC; i += NTHREADS) {
            const int c = i < BC ? i : i - BC;                 // i < BC: sum of channel c0 + i; else second moment of channel c0 + i - BC
            if (c0 + c < p.Cout) {
                const size_t e = (i < BC ? 0 : (size_t)p.Cout) + c0 + c;
                float v = cs[i];
#pragma unroll
                for (int w = 1; w < NSR; ++w) v += cs[w * 2 * BC + i];
                stat_publish(p.stats + e, 2 * (size_t)p.Cout, slot, v);
                if (second) {
                    float v2 = cs2[i];
#pragma unroll
                    for (int w = 1; w < NSR; ++w) v2 += cs2[w * 2 * BC + i];
                    stat_publish(p.stats2 + e, 2 * (size_t)p.Cout, slot, v2);
                }
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
    float* ws;             // optional [nsplit][numel(dw)] partial buffer (plain stores) instead of atomics
    size_t dw_numel;
    int nsplit;
    int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, act, cin_true;
    int P, pix_per_block, n_cotiles, n_tiles, cin_shift, NK;   // NK = KH*KW*Cin: flattened (tap, ci) GEMM-N extent
    size_t gdz, gx;        // element strides between BatchNorm groups (blockIdx.y = group)
    int in_gstride;
    // LZ kernels: lazy transform of the dz operand as well (Gram matrix a^T a of a lazily normalised activation)
    const float* dz_scale;
    const float* dz_shift;
    int dz_act, dz_gstride;
};


template <int BM, int BN, int WPD = 1, bool LZ = false>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_kernel(WgradP p) {
    constexpr int AROW = BM * 2;            // bytes per LDS row (one pixel)
    constexpr int BROW = BN * 2;
    constexpr int TILE_BYTES = 32 * (AROW + BROW);
    constexpr int MT = BM / 32, NT = BN / 32;
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES];

    // block -> (group, pixel split, tile), tile fastest, through the XCD-contiguous bijection: each XCD works through a
    // contiguous run of this list, so the tiles of one pixel range (which all re-read the same dz / activation rows)
    // meet in one L2, and every XCD gets an equal share however few splits there are
    const int lb = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int tile = lb % p.n_tiles;
    const int unit = lb / p.n_tiles;
    const int split = unit % p.nsplit, grp = unit / p.nsplit;
    p.dz += (size_t)grp * p.gdz;
    p.x += (size_t)grp * p.gx;
    if (p.in_scale) { p.in_scale += (size_t)grp * p.in_gstride; p.in_shift += (size_t)grp * p.in_gstride; }
    if (LZ && p.dz_scale) { p.dz_scale += (size_t)grp * p.dz_gstride; p.dz_shift += (size_t)grp * p.dz_gstride; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int co0 = (tile % p.n_cotiles) * BM;
    const int n0 = (tile / p.n_cotiles) * BN;           // offset in the flattened (tap, ci) axis
    const int ps = split * p.pix_per_block;
    const int pe = min(p.P, ps + p.pix_per_block);

    constexpr int ACH = BM / 8, BCH = BN / 8;                 // 16-byte chunks per row
    constexpr int AL = (32 * ACH) / NTHREADS, BL = (32 * BCH) / NTHREADS;
    // register prefetch ring: global loads run WPD K steps (of 32 pixels) ahead of the MFMAs.  One K step is ~0.1 us of
    // matrix work but an HBM round trip is 1-2 us: with a one-step look-ahead the layer2-4 problems (3-4 workgroups per CU)
    // sat at 250-450 TFLOP/s and ~2.4 TB/s -- neither roof.  WPD = 4 costs 40 VGPRs (5 -> 3 waves per SIMD), which loses
    // on the HBM-bound layer-1 shapes and on grids with > 4 workgroups per CU, so the launcher picks per problem.
    bf16x8 ra[WPD][AL], rb[WPD][BL];
    bool rbv[WPD][BL];
    bool rav_a[LZ ? WPD : 1][AL];
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

    auto issue_loads = [&](auto slot_c, int pbase) {
        constexpr int SL = decltype(slot_c)::value;
#pragma unroll
        for (int l = 0; l < AL; ++l) {
            int e = tid + l * NTHREADS;
            int row = e / ACH, ch = e - row * ACH;
            int pp = pbase + row, co = co0 + ch * 8;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (pp < pe && co < p.Cout) v = *reinterpret_cast<const bf16x8*>(p.dz + (size_t)pp * p.Cout + co);
            ra[SL][l] = v;
            if (LZ) rav_a[LZ ? SL : 0][l] = pp < pe && co < p.Cout;
        }
#pragma unroll
        for (int l = 0; l < BL; ++l) {
            int e = tid + l * NTHREADS;
            int row = e / BCH;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            int ih = b_oh[l] * p.stride + b_kh[l], iw = b_ow[l] * p.stride + b_kw[l];
            bool ok = (pbase + row < pe) && b_ok[l] && ih >= 0 && iw >= 0 && ih < p.H && iw < p.W;
            if (ok) v = *reinterpret_cast<const bf16x8*>(p.x + ((size_t)(b_n[l] * p.H + ih) * p.W + iw) * p.Cin + b_ci[l]);
            rbv[SL][l] = ok;
            rb[SL][l] = v;
            // advance this chunk's pixel by one K step (32 output pixels)
            b_ow[l] += 32;
            while (b_ow[l] >= p.OW) { b_ow[l] -= p.OW; ++b_oh[l]; }
            while (b_oh[l] >= p.OH) { b_oh[l] -= p.OH; ++b_n[l]; }
        }
    };
    auto store_tile = [&](auto slot_c, int buf) {
        constexpr int SL = decltype(slot_c)::value;
        char* base = smem + buf * TILE_BYTES;
#pragma unroll
        for (int l = 0; l < AL; ++l) {
            int e = tid + l * NTHREADS;
            int row = e / ACH, ch = e - row * ACH;
            bf16x8 va = ra[SL][l];
            if (LZ && p.dz_scale && rav_a[LZ ? SL : 0][l]) va = f32_to_bf8(transform8(va, p.dz_scale, p.dz_shift, co0 + ch * 8, p.dz_act));
            *reinterpret_cast<bf16x8*>(base + row * AROW + ((ch ^ (tr_swz<BM>(row) >> 1)) << 4)) = va;
        }
#pragma unroll
        for (int l = 0; l < BL; ++l) {
            int e = tid + l * NTHREADS;
            int row = e / BCH, ch = e - row * BCH;
            bf16x8 v = rb[SL][l];
            if (p.in_scale && rbv[SL][l]) v = f32_to_bf8(transform8(v, p.in_scale, p.in_shift, b_ci[l], p.act));
            *reinterpret_cast<bf16x8*>(base + 32 * AROW + row * BROW + ((ch ^ (tr_swz<BN>(row) >> 1)) << 4)) = v;
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = pe > ps ? (pe - ps + 31) / 32 : 0;
    const int li = lane & 15, lg = lane >> 4;
    // transpose-read addressing: lane li of a 16-lane group supplies the 8-byte unit
    // [pixel row 8*lg + (li>>2) (+4)][channels 4*(li&3) ..+3]; it receives channel li of rows 0..3.
    const int trow = 8 * lg + (li >> 2), tq = li & 3;
    const int a_lo = trow * AROW, a_hi = (trow + 4) * AROW, b_lo = trow * BROW, b_hi = (trow + 4) * BROW;
    const int ax_lo = tr_swz<BM>(trow), ax_hi = tr_swz<BM>(trow + 4), bx_lo = tr_swz<BN>(trow), bx_hi = tr_swz<BN>(trow + 4);
    auto compute = [&](int buf) {
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
    };
    static_for<WPD>([&](auto sc) {
        if ((int)decltype(sc)::value < nk) issue_loads(sc, ps + (int)decltype(sc)::value * 32);
    });
    for (int kt0 = 0; kt0 < nk; kt0 += WPD) {
        static_for<WPD>([&](auto sc) {
            const int kt = kt0 + (int)decltype(sc)::value;
            if (kt < nk) {                               // uniform
                store_tile(sc, kt & 1);                  // waits (counted vmcnt) only for this slot's loads
                if (kt + WPD < nk) issue_loads(sc, ps + (kt + WPD) * 32);
                __syncthreads();                         // tile kt visible; everyone is past compute(kt-1)
                compute(kt & 1);
            }
        });
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
                if (co >= p.Cout) continue;
                // workspace partials are tap-major [co][tap][ci] (lanes = consecutive ci -> 64 B runs instead of 4 B
                // stores 36 B apart); the split reduction permutes to OIHW
                if (p.ws) p.ws[((size_t)grp * p.nsplit + split) * p.dw_numel + ((size_t)co * taps + tap) * p.cin_true + ci] = acc[mt][nt][r];
                else atomicAdd(p.dw + ((size_t)co * p.cin_true + ci) * taps + tap, acc[mt][nt][r]);
            }
        }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient with LDS-DMA staging (global_load_lds_dwordx4) for operands that are PLAIN in memory (no lazy transform):
// the same tiles, LDS image and MFMA schedule as conv_wgrad_kernel, but the operand tiles go global -> LDS directly --
// no VGPR ring, no ds_write pass -- through a ring of ST LDS stages with counted vmcnt across raw barriers, so that ST-2
// K steps of loads stay in flight while one is being multiplied.  (Ablation of the register-staged kernel, layer-2 3x3: 418
// TFLOP/s as is, 481 without its LDS stores, 717 without its global loads, 956 without both, 447 without its MFMAs: it is
// bound by its staging, not by the matrix cores.)  The LDS destination of an LDS-DMA is wave-uniform base + lane * 16, so the
// image is lane-linear and the bank swizzle of the transposed reads is applied to the SOURCE chunk index instead (an
// involution within a pixel row: the same cache lines are fetched).  Out-of-range chunks (padding taps, tails) read a zero page.
// LZB (1x1 convs): a lazily normalised x operand is staged RAW and its BatchNorm + activation transform is applied to the B fragment
// after the transpose read.  That fragment holds 8 pixels of ONE channel per lane, so the transform needs one scale / shift pair per
// lane and fragment (registers, loaded once) and ~28 VALU instructions beside 4-8 MFMAs -- unlike the forward kernels' fragments
// (8 channels of one pixel per lane).  Rows past the pixel range hold zeros in the dz operand, so whatever act(shift) the transform
// makes of the x operand's zero rows is multiplied by 0; K x K convs keep the staging-side transform (their padding taps must BE zero).
template <int BM, int BN, int ST, bool LZB = false>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_glds_kernel(WgradP p) {
    const bf16_t* zeros = reinterpret_cast<const bf16_t*>(g_zero_page);
    constexpr int AROW = BM * 2, BROW = BN * 2;
    constexpr int TILE_BYTES = 32 * (AROW + BROW);
    constexpr int MT = BM / 32, NT = BN / 32;
    __shared__ __attribute__((aligned(1024))) char smem[ST * TILE_BYTES];
    const int lb = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int tile = lb % p.n_tiles;
    const int unit = lb / p.n_tiles;
    const int split = unit % p.nsplit, grp = unit / p.nsplit;
    p.dz += (size_t)grp * p.gdz;
    p.x += (size_t)grp * p.gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int co0 = (tile % p.n_cotiles) * BM;
    const int n0 = (tile / p.n_cotiles) * BN;
    const int ps = split * p.pix_per_block;
    const int pe = min(p.P, ps + p.pix_per_block);
    constexpr int ACH = BM / 8, BCH = BN / 8;
    constexpr int AL = (32 * ACH) / NTHREADS, BL = (32 * BCH) / NTHREADS;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // staging slots of this thread: LDS chunk e = l * 256 + tid of a tile region, i.e. row e / CH at position e % CH, which holds
    // SOURCE chunk position ^ swizzle(row)
    int a_row[AL], a_co[AL];
    bool a_cok[AL];
#pragma unroll
    for (int l = 0; l < AL; ++l) {
        const int e = tid + l * NTHREADS;
        a_row[l] = e / ACH;
        const int ch = (e - a_row[l] * ACH) ^ (tr_swz<BM>(a_row[l]) >> 1);
        a_co[l] = co0 + ch * 8;
        a_cok[l] = a_co[l] < p.Cout;
    }
    int b_row[BL], b_kh[BL], b_kw[BL], b_ci[BL], b_n[BL], b_oh[BL], b_ow[BL];
    bool b_ok[BL];
#pragma unroll
    for (int l = 0; l < BL; ++l) {
        const int e = tid + l * NTHREADS;
        b_row[l] = e / BCH;
        const int ch = (e - b_row[l] * BCH) ^ (tr_swz<BN>(b_row[l]) >> 1);
        const int n = n0 + ch * 8;
        b_ok[l] = n < p.NK;
        const int tap = n >> p.cin_shift;
        b_ci[l] = n - (tap << p.cin_shift);
        b_kh[l] = tap / p.KW - p.pad;
        b_kw[l] = tap - (tap / p.KW) * p.KW - p.pad;
        const int pp = ps + b_row[l];
        b_n[l] = pp / (p.OH * p.OW);
        const int rem = pp - b_n[l] * (p.OH * p.OW);
        b_oh[l] = rem / p.OW;
        b_ow[l] = rem - b_oh[l] * p.OW;
    }
    auto stage = [&](int pbase, int st) {                       // 4 LDS-DMAs per thread (128 x 128 tile)
        const unsigned sbase = lds0 + st * TILE_BYTES + wave * 1024;
#pragma unroll
        for (int l = 0; l < AL; ++l) {
            const int pp = pbase + a_row[l];
            const bf16_t* src = (pp < pe && a_cok[l]) ? p.dz + (size_t)pp * p.Cout + a_co[l] : zeros;
            glds16(src, __builtin_amdgcn_readfirstlane(sbase + l * NTHREADS * 16));
        }
#pragma unroll
        for (int l = 0; l < BL; ++l) {
            const int ih = b_oh[l] * p.stride + b_kh[l], iw = b_ow[l] * p.stride + b_kw[l];
            const bool ok = (pbase + b_row[l] < pe) && b_ok[l] && ih >= 0 && iw >= 0 && ih < p.H && iw < p.W;
            const bf16_t* src = ok ? p.x + ((size_t)(b_n[l] * p.H + ih) * p.W + iw) * p.Cin + b_ci[l] : zeros;
            glds16(src, __builtin_amdgcn_readfirstlane(sbase + 32 * AROW + l * NTHREADS * 16));
            b_ow[l] += 32;
            while (b_ow[l] >= p.OW) { b_ow[l] -= p.OW; ++b_oh[l]; }
            while (b_oh[l] >= p.OH) { b_oh[l] -= p.OH; ++b_n[l]; }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = pe > ps ? (pe - ps + 31) / 32 : 0;
    const int li = lane & 15, lg = lane >> 4;
    const int trow = 8 * lg + (li >> 2), tq = li & 3;
    const int a_lo = trow * AROW, a_hi = (trow + 4) * AROW, b_lo = trow * BROW, b_hi = (trow + 4) * BROW;
    const int ax_lo = tr_swz<BM>(trow), ax_hi = tr_swz<BM>(trow + 4), bx_lo = tr_swz<BN>(trow), bx_hi = tr_swz<BN>(trow + 4);
    float bsc[LZB ? NT : 1], bsh[LZB ? NT : 1];
    if constexpr (LZB) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int nn = min(n0 + wn * (BN / 2) + t * 16 + li, p.NK - 1);        // (1x1: the flattened index IS the input channel)
            bsc[t] = p.in_scale[(size_t)grp * p.in_gstride + nn];
            bsh[t] = p.in_shift[(size_t)grp * p.in_gstride + nn];
        }
    }
    const float blo = act_lo(p.act), bhi = act_hi(p.act);
    auto compute = [&](int st) {
        const char* base = smem + st * TILE_BYTES;
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
            if constexpr (LZB) {
                f32x8 f = bf8_to_f32(fb[t]);
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] = clamp_act(fmaf(f[i], bsc[t], bsh[t]), blo, bhi);
                fb[t] = f32_to_bf8(f);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
    };
    // ring of ST stages: steps kt+1 .. kt+ST-2 stay in flight (vmcnt counts this thread's LDS-DMAs, AL + BL per step) while step kt
    // is multiplied; ONE raw barrier per step orders "step kt landed for every wave" and "everyone is done reading stage (kt-1) % ST"
    constexpr int PER = AL + BL;
#pragma unroll
    for (int s0 = 0; s0 < ST - 1; ++s0)
        if (s0 < nk) stage(ps + s0 * 32, s0);
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + ST - 2 <= nk - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + ST - 1 < nk) stage(ps + (kt + ST - 1) * 32, st == 0 ? ST - 1 : st - 1);
        compute(st);
        st = st + 1 == ST ? 0 : st + 1;
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
                if (co >= p.Cout) continue;
                p.ws[((size_t)grp * p.nsplit + split) * p.dw_numel + ((size_t)co * taps + tap) * p.cin_true + ci] = acc[mt][nt][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// 3x3 weight gradient, all nine taps per workgroup.  One K step = up to 32 output pixels of one image (a row
// segment, or floor(32/OW) whole rows); the matching input patch (with its 1-pixel halo) is staged ONCE in LDS and
// the nine shifted B operands are fetched from it with per-lane transpose reads, so dz and the activations are
// read once per (co-tile, ci-tile) instead of once per tap.  Tile: 64 co x (9 taps x 64 ci); wave w owns the 16-ci
// slice w for all taps and all 64 co (36 accumulator tiles = 144 VGPRs).
struct W3P {
    const bf16_t* dz;
    const bf16_t* x;
    const float* in_scale;
    const float* in_shift;
    float* dw;
    float* ws;
    size_t dw_numel;
    int nsplit;
    int N, H, W, Cin, OH, OW, Cout, pad, act, cin_true;
    int cw, rows, PR, PC, units_per_img, units_per_row, total_units, units_per_block, n_cotiles, n_tiles;
    size_t gdz, gx;
    int in_gstride;
};

template <int S, int MAXSLOT>
__global__ __launch_bounds__(NTHREADS, 2) void conv3x3_wgrad_kernel(W3P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lb = (int)xcd_contiguous(blockIdx.x, gridDim.x);   // (group, split, tile) list, tile fastest (see conv_wgrad_kernel)
    const int tile = lb % p.n_tiles;
    const int split = (lb / p.n_tiles) % p.nsplit, grp = (lb / p.n_tiles) / p.nsplit;
    p.dz += (size_t)grp * p.gdz;
    p.x += (size_t)grp * p.gx;
    if (p.in_scale) { p.in_scale += (size_t)grp * p.in_gstride; p.in_shift += (size_t)grp * p.in_gstride; }
    const int patch_bytes = p.PR * p.PC * 128;
    const int buf_bytes = 32 * 128 + patch_bytes;               // dz tile [32][64] + patch [PR*PC][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int co0 = (tile % p.n_cotiles) * 64;
    const int ci0 = (tile / p.n_cotiles) * 64;
    const int u0 = split * p.units_per_block;
    const int u1 = min(p.total_units, u0 + p.units_per_block);

    // ---- fixed per-thread staging slots ---------------------------------------------------------------------
    // dz tile: 32 rows x 8 chunks = 256 chunks -> one per thread
    const int a_j = tid >> 3, a_ch = tid & 7;
    const int a_r = a_j / p.cw, a_c = a_j - a_r * p.cw;
    // patch: PR*PC pixels x 8 chunks, up to MAXSLOT slots per thread
    const int n_chunks = p.PR * p.PC * 8;
    int s_pr[MAXSLOT], s_pc[MAXSLOT];
#pragma unroll
    for (int l = 0; l < MAXSLOT; ++l) {
        const int e = tid + l * NTHREADS;
        const int pix = e >> 3;
        s_pr[l] = pix / p.PC;
        s_pc[l] = pix - s_pr[l] * p.PC;
    }
    const int b_ch = tid & 7;                                   // chunk within the 64-ci row (same for all slots)
    bf16x8 ra, rb[MAXSLOT];
    bool rbv[MAXSLOT];

    auto decode = [&](int u, int& n, int& oh0, int& ow0) {
        n = u / p.units_per_img;
        int rem = u - n * p.units_per_img;
        int ug = rem / p.units_per_row;                         // row group
        int seg = rem - ug * p.units_per_row;
        oh0 = ug * p.rows;
        ow0 = seg * p.cw;
    };
    auto issue_loads = [&](int u) {
        int n, oh0, ow0;
        decode(u, n, oh0, ow0);
        {
            const int oh = oh0 + a_r, ow = ow0 + a_c;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (a_r < p.rows && oh < p.OH && ow < p.OW)
                v = *reinterpret_cast<const bf16x8*>(p.dz + ((size_t)(n * p.OH + oh) * p.OW + ow) * p.Cout + co0 + a_ch * 8);
            ra = v;
        }
        const int ih0 = oh0 * S - p.pad, iw0 = ow0 * S - p.pad;
#pragma unroll
        for (int l = 0; l < MAXSLOT; ++l) {
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            const int ih = ih0 + s_pr[l], iw = iw0 + s_pc[l];
            const bool ok = (tid + l * NTHREADS) < n_chunks && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            if (ok) v = *reinterpret_cast<const bf16x8*>(p.x + ((size_t)(n * p.H + ih) * p.W + iw) * p.Cin + ci0 + b_ch * 8);
            rbv[l] = ok;
            rb[l] = v;
        }
    };
    auto store_tile = [&](int buf) {
        char* base = smem + buf * buf_bytes;
        *reinterpret_cast<bf16x8*>(base + a_j * 128 + ((a_ch ^ (tr_swz<64>(a_j) >> 1)) << 4)) = ra;
        char* pb = base + 32 * 128;
#pragma unroll
        for (int l = 0; l < MAXSLOT; ++l) {
            if (tid + l * NTHREADS < n_chunks) {
                bf16x8 v = rb[l];
                if (p.in_scale && rbv[l]) v = f32_to_bf8(transform8(v, p.in_scale, p.in_shift, ci0 + b_ch * 8, p.act));
                const int pix = s_pr[l] * p.PC + s_pc[l];
                // 16-byte chunk swizzle by patch column: neighbouring columns that share a bank half get distinct slots
                *reinterpret_cast<bf16x8*>(pb + pix * 128 + ((b_ch ^ (((s_pc[l] >> 1) & 3) << 1)) << 4)) = v;
            }
        }
    };

    f32x4 acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fixed per-lane fragment addressing -----------------------------------------------------------------
    const int trow = 8 * lg + (li >> 2), tq = li & 3;
    const int a_lo = trow * 128, a_hi = (trow + 4) * 128;
    const int ax_lo = tr_swz<64>(trow), ax_hi = tr_swz<64>(trow + 4);
    int b_base[2], b_col[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = trow + 4 * h;
        int r = j / p.cw, c = j - r * p.cw;
        if (r >= p.rows) { r = 0; c = 0; }                       // padding k-rows (dz row is zero): read any FINITE patch pixel
        b_col[h] = c * S;                                        // patch column of tap (.,0) for k-row j
        b_base[h] = r * S * p.PC + c * S;                        // patch pixel index of tap (0,0)
    }
    const int b_unit = wave * 4 + tq;                            // 8-byte unit of this lane's 4 ci inside the 64-ci row

    if (u0 < u1) {
        issue_loads(u0);
        store_tile(0);
    }
    __syncthreads();
    for (int u = u0; u < u1; ++u) {
        const int buf = (u - u0) & 1;
        if (u + 1 < u1) issue_loads(u + 1);
        const char* base = smem + buf * buf_bytes;
        const char* pb = base + 32 * 128;
        bf16x8 fa[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int un = t * 4 + tq;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + a_lo + ((un ^ ax_lo) << 3)));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + a_hi + ((un ^ ax_hi) << 3)));
            union { struct { s16x4 a, b; } s; bf16x8 v; } cvt;
            cvt.s.a = lo; cvt.s.b = hi;
            fa[t] = cvt.v;
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                s16x4 half[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pix = b_base[h] + kh * p.PC + kw;
                    const int un = b_unit ^ ((((b_col[h] + kw) >> 1) & 3) << 2);
                    half[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + pix * 128 + (un << 3)));
                }
                union { struct { s16x4 a, b; } s; bf16x8 v; } cvt;
                cvt.s.a = half[0]; cvt.s.b = half[1];
                const bf16x8 fb = cvt.v;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt][kh * 3 + kw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb, acc[mt][kh * 3 + kw], 0, 0, 0);
            }
        if (u + 1 < u1) store_tile(buf ^ 1);
        __syncthreads();
    }
    const int ci = ci0 + wave * 16 + li;
    if (ci < p.cin_true) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + mt * 16 + lg * 4 + r;
                    if (p.ws) p.ws[((size_t)grp * p.nsplit + split) * p.dw_numel + ((size_t)co * 9 + t) * p.cin_true + ci] = acc[mt][t][r];
                    else atomicAdd(p.dw + ((size_t)co * p.cin_true + ci) * 9 + t, acc[mt][t][r]);
                }
    }
}

// dw[perm(i)] += sum_s ws[s][i]: 16 indices x 16 split lanes per workgroup (the split loop is the long axis).
// taps > 1: ws is tap-major [co][tap][cin], dw is OIHW [co][cin][tap].
// blockIdx.y = output group (per-group products of the algebraic BatchNorm backward: ws [group][split][n] -> dw [group][n], stored)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* ws, float* dw, size_t n, int nsplit, int taps, int cin, int store) {
    ws += (size_t)blockIdx.y * nsplit * n;
    dw += (size_t)blockIdx.y * n;
    __shared__ float red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const size_t i = (size_t)blockIdx.x * 16 + tx;
    float a = 0.f;
    if (i < n) {
        // four independent partial sums (fixed order): with one accumulator every load waits for the previous add -- hundreds of
        // dependent round trips when a small problem was split over ~2000 workgroups
        float a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int s = ty;
        for (; s + 48 < nsplit; s += 64) {
            a += ws[(size_t)s * n + i];
            a1 += ws[(size_t)(s + 16) * n + i];
            a2 += ws[(size_t)(s + 32) * n + i];
            a3 += ws[(size_t)(s + 48) * n + i];
        }
        for (; s < nsplit; s += 16) a += ws[(size_t)s * n + i];
        a = (a + a1) + (a2 + a3);
    }
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][tx];
        size_t o = i;
        if (taps > 1) {
            const size_t per_co = (size_t)taps * cin;
            const size_t co = i / per_co;
            const int rem = (int)(i - co * per_co), tap = rem / cin, ci = rem - tap * cin;
            o = (co * cin + ci) * taps + tap;
        }
        if (store) dw[o] = t; else dw[o] += t;
    }
}

// The same reduction, four consecutive indices per thread (n % 4 == 0): 16-byte loads, 256 contiguous bytes per split row of a workgroup
// instead of 64 -- the one-index form moved 1 TB/s and was, at 104 launches x 17 us, the largest of the step's small kernels
// (1.8 ms per step; 2.2 of 34 ms of kernel time at the per-GPU share of the reference recipe).  Same fixed summation order per index.
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* ws, float* dw, size_t n, int nsplit, int taps, int cin, int store) {
    ws += (size_t)blockIdx.y * nsplit * n;
    dw += (size_t)blockIdx.y * n;
    __shared__ f32x4 red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const size_t i = ((size_t)blockIdx.x * 16 + tx) * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        f32x4 a1 = a, a2 = a, a3 = a;
        int s = ty;
        for (; s + 48 < nsplit; s += 64) {
            a += *reinterpret_cast<const f32x4*>(ws + (size_t)s * n + i);
            a1 += *reinterpret_cast<const f32x4*>(ws + (size_t)(s + 16) * n + i);
            a2 += *reinterpret_cast<const f32x4*>(ws + (size_t)(s + 32) * n + i);
            a3 += *reinterpret_cast<const f32x4*>(ws + (size_t)(s + 48) * n + i);
        }
        for (; s < nsplit; s += 16) a += *reinterpret_cast<const f32x4*>(ws + (size_t)s * n + i);
        a = (a + a1) + (a2 + a3);
    }
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && i < n) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][tx];
        if (taps > 1) {
            const size_t per_co = (size_t)taps * cin;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t ii = i + q;
                const size_t co = ii / per_co;
                const int rem = (int)(ii - co * per_co), tap = rem / cin, ci = rem - tap * cin;
                const size_t o = (co * cin + ci) * taps + tap;
                if (store) dw[o] = t[q]; else dw[o] += t[q];
            }
        } else if ((reinterpret_cast<uintptr_t>(dw + i) & 15) == 0) {
            f32x4* o = reinterpret_cast<f32x4*>(dw + i);
            if (store) *o = t; else *o += t;
        } else {                                  // (gradient views of the flat buffer are only 4-byte aligned behind an odd-sized parameter)
#pragma unroll
            for (int q = 0; q < 4; ++q) { if (store) dw[i + q] = t[q]; else dw[i + q] += t[q]; }
        }
    }
}

// picks the 16-byte form whenever the index count allows it
void launch_wgrad_reduce(const float* ws, float* dw, size_t n, int nsplit, int taps, int cin, int store, int groups, hipStream_t stream) {
    static const bool wide = !(getenv("ADAMML_REDUCE4") && getenv("ADAMML_REDUCE4")[0] == '0');       // A/B aid
    if (wide && n % 4 == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0)
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)((n / 4 + 15) / 16), groups), dim3(256), 0, stream, ws, dw, n, nsplit, taps, cin, store);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 15) / 16), groups), dim3(256), 0, stream, ws, dw, n, nsplit, taps, cin, store);
}

int ilog2_exact(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return (1 << s) == v ? s : -1;
}

}  // namespace

bool adamml_conv3x3_c64_supported(const adamml_conv_desc_t* d);
bool adamml_conv1x1_narrow_fwd_supported(const adamml_conv_desc_t* d);
bool adamml_conv1x1_narrow_wgrad_supported(const adamml_conv_desc_t* d, int cin_true);
bool adamml_conv1x1_narrow_dual_supported(const adamml_conv_desc_t* d);
bool adamml_conv1x1_narrow_dgrad_epi_supported(const adamml_conv_desc_t* d);
int adamml_conv1x1_narrow_dgrad_epi_launch(const adamml_conv_desc_t* d, const void* dz, const void* w_packed, void* dx, const void* z_in,
                                           const float* bn_vec, int act, double* sums, hipStream_t stream);
int adamml_conv1x1_narrow_dual_launch(const adamml_conv_desc_t* d, const void* g, const void* z, const float* aff, void* dz_side,
                                      const void* w_dgrad_packed, void* dx, int accumulate, const void* z_in, const float* bn_vec, int act,
                                      double* sums, hipStream_t stream);
int adamml_conv1x1_narrow_wgrad_launch(const adamml_conv_desc_t* d, const void* dz, const void* x, const float* in_scale, const float* in_shift,
                                       float* ws, int max_blocks_per_group, int* nblk_out, hipStream_t stream);
int adamml_conv1x1_narrow_fwd_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                                     void* y, double* stats, hipStream_t stream);
int adamml_conv3x3_c64_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale,
                              const float* in_shift, void* y, double* stats, const void* bn_z, const float* bn_vec, int bn_act,
                              hipStream_t stream);
// wide 1x1 convs of ResNet layers 3-4 (conv1x1_wide.hip)
bool adamml_conv1x1_wide_expand_supported(const adamml_conv_desc_t* d);
int adamml_conv1x1_wide_expand_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                                      void* y, double* stats, hipStream_t stream);

bool adamml_conv3x3_c64_wgrad_supported(const adamml_conv_desc_t* d, int cin_true);
int adamml_conv3x3_c64_wgrad_blocks(const adamml_conv_desc_t* d, int* tpb_out);
int adamml_conv3x3_c64_wgrad_launch(const adamml_conv_desc_t* d, const void* dz, const void* x, const float* in_scale,
                                    const float* in_shift, float* ws, hipStream_t stream);

// one parity class (ph, pw) of the data gradient of a stride-2 conv (see conv_dgrad_stride2)
struct DgradClass { int nt; unsigned code; int ph, pw, OHc, OWc; };
// residual form of the BatchNorm-fused data-gradient epilogue (ConvP::res_out ..)
struct ResEpi { const void* res_out; const uint8_t* res_mask; int res_act; const void* bn_z2; const float* bn_vec2; double* stats2; };
// dual-source input of a 1x1 data gradient (ConvP::x2 ..)
struct DualIn { const void* z; const float* aff; void* side; };
// K-concatenated second input, per-group weights, epilogue constant (ConvP::xb ..)
struct CatIn { const void* xb; int C2; size_t gw; const float* epi_add; };
// forward BatchNorm + residual-add epilogue (ConvP::id_scale ..)
// product with a second tensor accumulated from the gradient tile (ConvP::pf_a ..); ws: partial workspace, nsplit: out
struct PfIn { const void* a; const float* scale; const float* shift; int act, gs, C; float* out; void* ws; size_t ws_bytes; };
struct FaddEpi { const float* vec; const void* idn; const float* id_scale; const float* id_shift; int id_gstride; int act; uint8_t* mask_out;
                 int tp_frames; void* tp_y; uint16_t* tp_code; };      // tp_frames > 0: temporal max-pool in the epilogue (ConvP::tp_y ..)

static int conv_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale,
                       const float* in_shift, void* y, double* stats, const void* bn_z, const float* bn_vec, int bn_act,
                       hipStream_t stream, const DgradClass* cls = nullptr, const ResEpi* res = nullptr, const DualIn* dual = nullptr,
                       const CatIn* cat = nullptr, const FaddEpi* fadd = nullptr, const PfIn* pf = nullptr) {
    if (!d || !x || !w_packed || !y) return adamml_set_error(ADAMML_EINVAL, "conv_fwd: null argument");
    if (d->Cin % 8 || d->Cout % 8) return adamml_set_error(ADAMML_EINVAL, "conv_fwd: channels must be multiples of 8 (Cin=%d Cout=%d)", d->Cin, d->Cout);
    if (!cls && !fadd && adamml_conv3x3_c64_supported(d))
        return adamml_conv3x3_c64_launch(d, x, w_packed, in_scale, in_shift, y, stats, bn_z, bn_vec, bn_act, stream);
    // narrow 1x1 convs of the MobileNetV2s (plain forward / plain data gradient): the barrier-free streaming kernel (conv1x1_narrow.hip)
    if (!cls && !fadd && !res && !dual && !cat && !pf && !bn_z && adamml_conv1x1_narrow_fwd_supported(d))
        return adamml_conv1x1_narrow_fwd_launch(d, x, w_packed, in_scale, in_shift, y, stats, stream);
    // expanding 1x1 convs of ResNet layers 3-4 (K = 256 / 512 -> >= 2 K channels; plain forward, stride-2 downsample forward, plain or
    // accumulating data gradient): activation-stationary streaming kernel (conv1x1_wide.hip)
    if (!cls && !fadd && !res && !dual && !cat && !pf && !bn_z && adamml_conv1x1_wide_expand_supported(d))
        return adamml_conv1x1_wide_expand_launch(d, x, w_packed, in_scale, in_shift, y, stats, stream);
    // ... their data gradients with the BatchNorm-fused epilogue (bn_z) or accumulating into the output
    if (!cls && !fadd && !res && !dual && !cat && !pf && !in_scale && (bn_z ? stats != nullptr && !d->accumulate : d->accumulate != 0) &&
        adamml_conv1x1_narrow_dgrad_epi_supported(d))
        return adamml_conv1x1_narrow_dgrad_epi_launch(d, x, w_packed, y, bn_z, bn_vec, bn_act, stats, stream);
    ConvP p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w_packed; p.in_scale = in_scale; p.in_shift = in_shift;
    p.y = (bf16_t*)y; p.stats = stats;
    p.bn_z = (const bf16_t*)bn_z; p.bn_vec = bn_vec; p.bn_act = bn_act;
    p.res_out = res ? (const bf16_t*)res->res_out : nullptr; p.res_act = res ? res->res_act : 0;
    p.res_mask = res ? res->res_mask : nullptr;
    p.bn_z2 = res ? (const bf16_t*)res->bn_z2 : nullptr; p.bn_vec2 = res ? res->bn_vec2 : nullptr; p.stats2 = res ? res->stats2 : nullptr;
    p.x2 = dual ? (const bf16_t*)dual->z : nullptr; p.aff = dual ? dual->aff : nullptr; p.side = dual ? (bf16_t*)dual->side : nullptr;
    p.xb = cat ? (const bf16_t*)cat->xb : nullptr; p.K1 = d->Cin; p.C2 = cat ? cat->C2 : 0; p.gw = cat ? cat->gw : 0;
    p.epi_add = cat ? cat->epi_add : nullptr;
    p.id_scale = fadd ? fadd->id_scale : nullptr; p.id_shift = fadd ? fadd->id_shift : nullptr; p.id_gstride = fadd ? fadd->id_gstride : 0;
    p.mask_out = fadd ? fadd->mask_out : nullptr;
    p.tp_nblk = p.tp_Q = 0; p.tp_y = nullptr; p.tp_code = nullptr;
    p.pf_a = nullptr; p.pf_scale = p.pf_shift = nullptr; p.pf_ws = nullptr; p.pf_act = p.pf_gs = p.pf_nsplit = 0;
    if (fadd) { p.bn_vec = fadd->vec; p.res_out = (const bf16_t*)fadd->idn; p.res_act = fadd->act; }
    const int groups = d->groups < 1 ? 1 : d->groups;
    p.gx = (size_t)d->N * d->H * d->W * d->Cin;
    p.gy = (size_t)d->N * d->OH * d->OW * d->Cout;
    p.in_gstride = d->in_gstride;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.up = d->up < 1 ? 1 : d->up;
    p.up_shift = ilog2_exact(p.up);
    if (p.up_shift < 0) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_fwd: up=%d is not a power of two", p.up);
    p.act = d->act; p.accumulate = d->accumulate;
    p.P = d->N * d->OH * d->OW; p.K = d->KH * d->KW * d->Cin;
    if (cat) {
        if (d->KH * d->KW != 1 || d->stride != 1 || (d->up > 1) || d->Cin % BK || cat->C2 % 8 || cls || res || dual)
            return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_data_alg: 1x1 / stride-1 convs with Cout %% 32 == 0 only");
        p.K = d->Cin + cat->C2;
    }
    p.wK = p.K;
    p.cls_nt = 0; p.cls_code = 0; p.oH = d->OH; p.oW = d->OW; p.o_ph = p.o_pw = 0;
    if (cls) {
        p.cls_nt = cls->nt; p.cls_code = cls->code; p.o_ph = cls->ph; p.o_pw = cls->pw;
        p.OH = cls->OHc; p.OW = cls->OWc;
        p.P = d->N * cls->OHc * cls->OWc;
        p.K = cls->nt * d->Cin;
    }
    const bool multitap = d->KH * d->KW > 1 || cls;
    p.cin_shift = 0;
    if (multitap) {
        p.cin_shift = ilog2_exact(d->Cin);
        if (p.cin_shift < 0) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_fwd: KxK conv needs power-of-two Cin (got %d)", d->Cin);
    }
    if (p.P <= 0) return ADAMML_OK;
    bool narrow = d->Cout <= 64 || (d->Cout % 128 != 0 && d->Cout < 256);
    // small problems: halve the cout tile so that at least ~2 workgroups per CU exist
    if (!narrow && (long)ceil_div(p.P, BP) * ceil_div(d->Cout, 128) * groups < 512) narrow = true;
    const int tp = fadd ? fadd->tp_frames : 0;
    if (tp) narrow = false;
    const int BC = narrow ? 64 : 128;
    p.n_ptiles = ceil_div(p.P, BP);
    if (tp) {
        // tiles = (clip, block of BP / tp pixels); all frames of a clip's pixel block sit in one tile
        p.tp_Q = d->OH * d->OW;
        p.tp_nblk = ceil_div(p.tp_Q, BP / tp);
        p.n_ptiles = (d->N / tp) * p.tp_nblk;
        p.tp_y = (bf16_t*)fadd->tp_y; p.tp_code = fadd->tp_code;
    }
    p.n_ctiles = ceil_div(d->Cout, BC);
    // consecutive pixel tiles per workgroup (amortises the statistics publication), keeping >= ~2048 workgroups
    p.tpb = (int)((long)p.n_ptiles * p.n_ctiles * groups / 2048);
    if (p.tpb < 1) p.tpb = 1;
    if (p.tpb > 8) p.tpb = 8;
    if (pf) {
        // one [BC][64] partial per workgroup: few, long-lived workgroups (~6 per CU slot pair over all groups and cout tiles)
        const long target = 1536;
        long nsp = target / ((long)p.n_ctiles * groups);
        if (nsp < 1) nsp = 1;
        p.tpb = (int)ceil_div(p.n_ptiles, (int)nsp);
        if (p.tpb < 1) p.tpb = 1;
    }
    dim3 grid(ceil_div(p.n_ptiles, p.tpb) * p.n_ctiles, groups), block(NTHREADS);
    const int taps = d->KH * d->KW;
    // MODE 0 needs the whole row base in 32-bit element offsets (true for every layer of the hot path)
    if ((long)d->N * d->H * d->W * d->Cin >= (1L << 31)) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_fwd: input tensor exceeds 2^31 elements");
    const int mode = cls ? 3 : (p.up > 1 ? 2 : (multitap ? (taps <= 64 ? 1 : 2) : 0));
    if (mode == 2 && !multitap) { p.cin_shift = 30; }      // 1x1 strided dgrad: tap = k >> 30 = 0, ci = k
    const int nk = ceil_div(p.K, BK);
    // < 1 wave of workgroups per CU slot and a long K loop: explicit look-ahead instead of occupancy.  Measured (tools/bench_conv.py,
    // B = 72): it pays for the 1x1 layers of layer 4 (0.115 vs 0.137 ms) and costs on its 3x3 layers (0.189 vs 0.139 ms forward,
    // 0.197 vs 0.146 ms data gradient: the tap gathers of MODE 1 hit L2 and the deeper ring only lowers the occupancy)
    const bool deep = (long)grid.x * grid.y <= 768 && nk >= 8 && mode != 1;
    if (fadd) {
        if (mode != 0 || res || dual || cat || stats) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_fwd_bn_add: only 1x1 / stride-1 convs");
        static const int fadd_glds = getenv("ADAMML_FADD_GLDS") ? atoi(getenv("ADAMML_FADD_GLDS")) : 2;     // 0: register ring, 1: LDS-DMA, 2: + EID
        if (fadd_glds && (!in_scale || p.K <= 512)) {
            // operands by LDS-DMA (a lazily normalised input is transformed at the fragment: LZF); with an identity operand, that one is
            // requested at the start of each tile (EID)
            const bool eid = fadd_glds > 1 && p.res_out;
            if (tp) {
#define LAUNCH_TP(TV)                                                                                                                                        \
                do {                                                                                                                                         \
                    if (in_scale) hipLaunchKernelGGL((conv_gemm_kernel<128, 0, 1, false, false, false, true, true, 1, true, -1, false, TV>), grid, block, 0, stream, p);  \
                    else hipLaunchKernelGGL((conv_gemm_kernel<128, 0, 1, false, false, false, true, true, 1, false, -1, false, TV>), grid, block, 0, stream, p);         \
                } while (0)
                if (tp == 8) LAUNCH_TP(8); else if (tp == 4) LAUNCH_TP(4); else LAUNCH_TP(2);
#undef LAUNCH_TP
                return adamml_check_launch("conv_fwd_bn_add_tpool");
            }
#define LAUNCH_FADD(BCV)                                                                                                                   \
            do {                                                                                                                           \
                if (in_scale) {                                                                                                            \
                    if (eid) hipLaunchKernelGGL((conv_gemm_kernel<BCV, 0, 1, false, false, false, true, true, 1, true>), grid, block, 0, stream, p);  \
                    else hipLaunchKernelGGL((conv_gemm_kernel<BCV, 0, 1, false, false, false, true, true, 0, true>), grid, block, 0, stream, p);      \
                } else {                                                                                                                   \
                    if (eid) hipLaunchKernelGGL((conv_gemm_kernel<BCV, 0, 1, false, false, false, true, true, 1, false>), grid, block, 0, stream, p); \
                    else hipLaunchKernelGGL((conv_gemm_kernel<BCV, 0, 1, false, false, false, true, true, 0, false>), grid, block, 0, stream, p);     \
                }                                                                                                                          \
            } while (0)
            if (BC == 64) LAUNCH_FADD(64); else LAUNCH_FADD(128);
#undef LAUNCH_FADD
        } else
        if (tp) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_fwd_bn_add_tpool: needs the LDS-DMA kernel (K <= 512 for a lazy input)");
        else
        if (BC == 64) hipLaunchKernelGGL((conv_gemm_kernel<64, 0, 1, false, false, false, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((conv_gemm_kernel<128, 0, 1, false, false, false, true>), grid, block, 0, stream, p);
        return adamml_check_launch("conv_fwd_bn_add");
    }
    if (res) {
        if (mode != 0) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_data_res: only 1x1 / stride-1 convs");
        static const bool res_eid = !(getenv("ADAMML_RES_EID") && getenv("ADAMML_RES_EID")[0] == '0');
        if (pf) {
            if (BC != 128 || d->Cout % 128 || pf->C != 64 || in_scale || !p.res_mask || !p.accumulate || p.bn_z || p.bn_z2 || !stats)
                return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_data_res_prod: needs Cout %% 128 == 0, a 64-channel product operand, "
                                                             "the accumulate + 1-bit-mask + sum(g')-only form");
            p.pf_nsplit = ceil_div(p.n_ptiles, p.tpb);
            const size_t need = (size_t)groups * p.n_ctiles * p.pf_nsplit * 128 * 64 * sizeof(float);
            if (!pf->ws || pf->ws_bytes < need) return adamml_set_error(ADAMML_EINVAL, "conv_bwd_data_res_prod: workspace too small (need %zu bytes)", need);
            p.pf_a = (const bf16_t*)pf->a; p.pf_scale = pf->scale; p.pf_shift = pf->scale ? pf->shift : nullptr; p.pf_act = pf->act; p.pf_gs = pf->gs;
            p.pf_ws = (float*)pf->ws;
            hipLaunchKernelGGL((conv_gemm_kernel<128, 0, 1, true, false, false, false, true, 1, false, -1, true>), grid, block, 0, stream, p);
            int rc = adamml_check_launch("conv_bwd_data_res_prod");
            if (rc) return rc;
            // P[g][ctile * 128 + r][c] = sum over the workgroups of (g, ctile), in workgroup order
            launch_wgrad_reduce((const float*)pf->ws, pf->out, (size_t)128 * 64, p.pf_nsplit, 1, 64, 1, groups * p.n_ctiles, stream);
            return adamml_check_launch("conv_bwd_data_res_prod (reduce)");
        }
        if (in_scale) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_data_res: the gradient operand is never lazy");
        if (res_eid && p.res_mask && p.accumulate && !p.bn_z && !p.bn_z2) {
            // the algebraic backward's form (identity gradient + 1-bit mask, sum(g') only): identity-side loads at the start of each tile
            if (BC == 64) hipLaunchKernelGGL((conv_gemm_kernel<64, 0, 1, true, false, false, false, true, 1>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((conv_gemm_kernel<128, 0, 1, true, false, false, false, true, 1>), grid, block, 0, stream, p);
        } else {
            if (BC == 64) hipLaunchKernelGGL((conv_gemm_kernel<64, 0, 1, true, false, false, false, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((conv_gemm_kernel<128, 0, 1, true, false, false, false, true>), grid, block, 0, stream, p);
        }
        return adamml_check_launch("conv_bwd_data_res");
    }
    if (cat) {
        if (mode != 0) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_data_alg: only 1x1 / stride-1 convs");
        if (BC == 64) hipLaunchKernelGGL((conv_gemm_kernel<64, 0, 3, false, false, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((conv_gemm_kernel<128, 0, 3, false, false, true>), grid, block, 0, stream, p);
        return adamml_check_launch("conv_bwd_data_alg");
    }
    if (dual) {
        if (mode != 0 || res) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_data_dual: only 1x1 / stride-1 convs");
        if (BC == 64) hipLaunchKernelGGL((conv_gemm_kernel<64, 0, 3, false, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((conv_gemm_kernel<128, 0, 2, false, true>), grid, block, 0, stream, p);       // (a third ring slot of g, z and weights does not fit 256 registers at 128-wide tiles)
        return adamml_check_launch("conv_bwd_data_dual");
    }
    static const bool glds_on = !(getenv("ADAMML_CONV_GLDS") && getenv("ADAMML_CONV_GLDS")[0] == '0');
    // (the fragment-side lazy transform, LZF, is NOT used for the plain forward convs: measured 5-10 % slower than the register-staged
    // loader -- layer-1 conv3 1.77 vs 1.61 ms, layer 2 0.59 vs 0.54, layer 3 0.23 vs 0.21: each pixel half is transformed by two waves,
    // between the ds_read and the MFMA; it only pays together with the early identity loads of the FADD kernel)
    const bool glds = glds_on && !in_scale && mode != 2;
    // the epilogue is a template parameter (EPI) of the LDS-DMA and one-step instances; deep-prefetch and MODE 2 keep the run-time form
    const int epi = bn_z ? 1 : (p.accumulate ? 2 : 0);
#define LAUNCH_EPI(BCV, MODEV, GL)                                                                                                          \
    do {                                                                                                                                    \
        if (epi == 0) hipLaunchKernelGGL((conv_gemm_kernel<BCV, MODEV, 1, false, false, false, false, GL, 0, false, 0>), grid, block, 0, stream, p);       \
        else if (epi == 1) hipLaunchKernelGGL((conv_gemm_kernel<BCV, MODEV, 1, false, false, false, false, GL, 0, false, 1>), grid, block, 0, stream, p);  \
        else hipLaunchKernelGGL((conv_gemm_kernel<BCV, MODEV, 1, false, false, false, false, GL, 0, false, 2>), grid, block, 0, stream, p);               \
    } while (0)
#define LAUNCH_CONV(BCV, MODEV)                                                                             \
    do {                                                                                                    \
        if (glds) LAUNCH_EPI(BCV, MODEV, true);                                                             \
        else if (deep) hipLaunchKernelGGL((conv_gemm_kernel<BCV, MODEV, 3>), grid, block, 0, stream, p);     \
        else LAUNCH_EPI(BCV, MODEV, false);                                                                 \
    } while (0)
    if (BC == 64) {
        if (mode == 0) LAUNCH_CONV(64, 0); else if (mode == 1) LAUNCH_CONV(64, 1);
        else if (mode == 2) { if (deep) hipLaunchKernelGGL((conv_gemm_kernel<64, 2, 3>), grid, block, 0, stream, p); else hipLaunchKernelGGL((conv_gemm_kernel<64, 2, 1>), grid, block, 0, stream, p); }
        else if (glds) LAUNCH_EPI(64, 3, true);
        else LAUNCH_EPI(64, 3, false);
    } else {
        if (mode == 0) LAUNCH_CONV(128, 0); else if (mode == 1) LAUNCH_CONV(128, 1);
        else if (mode == 2) { if (deep) hipLaunchKernelGGL((conv_gemm_kernel<128, 2, 3>), grid, block, 0, stream, p); else hipLaunchKernelGGL((conv_gemm_kernel<128, 2, 1>), grid, block, 0, stream, p); }
        else if (glds) LAUNCH_EPI(128, 3, true);
        else LAUNCH_EPI(128, 3, false);
    }
#undef LAUNCH_CONV
#undef LAUNCH_EPI
    return adamml_check_launch("conv_fwd");
}

extern "C" int adamml_conv1x1_narrow_supported(const adamml_conv_desc_t* d, int kind) {
    if (!d) return 0;
    if (kind == 0) return adamml_conv1x1_narrow_fwd_supported(d) ? 1 : 0;
    if (kind == 1) return adamml_conv1x1_narrow_wgrad_supported(d, d->Cin) ? 1 : 0;
    if (kind == 2) return adamml_conv1x1_narrow_dual_supported(d) ? 1 : 0;
    adamml_conv_desc_t g = *d;                                   // the data-gradient-shaped descriptor conv_launch dispatches on
    g.H = d->OH; g.W = d->OW; g.Cin = d->Cout; g.OH = d->H; g.OW = d->W; g.Cout = d->Cin; g.accumulate = 0;
    if (kind == 3) return adamml_conv1x1_narrow_fwd_supported(&g) ? 1 : 0;
    if (kind == 4) return adamml_conv1x1_narrow_dgrad_epi_supported(&g) ? 1 : 0;
    return 0;
}

extern "C" int adamml_conv1x1_wide_supported(const adamml_conv_desc_t* d, int kind) {
    if (!d) return 0;
    if (kind == 0) { adamml_conv_desc_t f = *d; f.accumulate = 0; return adamml_conv1x1_wide_expand_supported(&f) ? 1 : 0; }
    if (kind != 3 && kind != 4) return 0;
    if (d->stride != 1) return 0;                                // (strided data gradients run by parity class: conv_dgrad_stride2)
    adamml_conv_desc_t g = *d;                                   // the data-gradient-shaped descriptor conv_launch dispatches on
    g.H = d->OH; g.W = d->OW; g.Cin = d->Cout; g.OH = d->H; g.OW = d->W; g.Cout = d->Cin; g.accumulate = kind == 4; g.up = 1; g.pad = 0;
    return adamml_conv1x1_wide_expand_supported(&g) ? 1 : 0;
}

extern "C" int adamml_conv_fused_input_supported(const adamml_conv_desc_t* d) {
    return d && adamml_conv3x3_c64_supported(d) ? 1 : 0;
}

extern "C" int adamml_conv_fwd(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale,
                               const float* in_shift, void* y, double* stats, hipStream_t stream) {
    return conv_launch(d, x, w_packed, in_scale, in_shift, y, stats, nullptr, nullptr, 0, stream);
}

extern "C" int adamml_conv_fwd_bn_add_supported(const adamml_conv_desc_t* d) {
    return d && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && (d->up <= 1) && d->Cin % 8 == 0 && d->Cout % 8 == 0 ? 1 : 0;
}

// (csrc/conv1x1_fadd_stream.hip: the barrier-free streaming form of the layer-2 shape)
int adamml_conv1x1_fadd_stream_supported(const adamml_conv_desc_t* d);
int adamml_conv1x1_fadd_stream_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                                      const float* bn_vec, const void* idn, const float* id_scale, const float* id_shift, int id_gstride, int act,
                                      void* out, uint8_t* mask_out, hipStream_t stream);
extern "C" int adamml_conv_fwd_bn_add_streams(const adamml_conv_desc_t* d) {
    return d && adamml_conv_fwd_bn_add_supported(d) && adamml_conv1x1_fadd_stream_supported(d) ? 1 : 0;
}

extern "C" int adamml_conv_fwd_bn_add(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale,
                                      const float* in_shift, const float* bn_vec, const void* idn, const float* id_scale,
                                      const float* id_shift, int id_gstride, int act, void* out, uint8_t* mask_out, hipStream_t stream) {
    if (!adamml_conv_fwd_bn_add_supported(d)) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_fwd_bn_add: 1x1 / stride-1 convs only");
    if (!bn_vec) return adamml_set_error(ADAMML_EINVAL, "conv_fwd_bn_add: null BatchNorm vectors");
    if (idn && adamml_conv1x1_fadd_stream_supported(d))
        return adamml_conv1x1_fadd_stream_launch(d, x, w_packed, in_scale, in_shift, bn_vec, idn, id_scale, id_shift, id_gstride, act, out, mask_out, stream);
    FaddEpi f{bn_vec, idn, id_scale, id_shift, id_gstride, act, mask_out, 0, nullptr, nullptr};
    return conv_launch(d, x, w_packed, in_scale, in_shift, out, nullptr, nullptr, nullptr, 0, stream, nullptr, nullptr, nullptr, nullptr, &f);
}

bool adamml_conv1x1_fadd_next_supported(const adamml_conv_desc_t* d, int next_cout);
int adamml_conv1x1_fadd_next_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                                    const float* bn_vec, const void* idn, const float* id_scale, const float* id_shift, int id_gstride, int act,
                                    void* out, uint8_t* mask_out, const void* w1_packed, void* y1, double* stats1, hipStream_t stream);

extern "C" int adamml_conv_fwd_bn_add_next_supported(const adamml_conv_desc_t* d, int next_cout) {
    return d && next_cout > 0 && adamml_conv_fwd_bn_add_supported(d) && adamml_conv1x1_fadd_next_supported(d, next_cout) ? 1 : 0;
}

extern "C" int adamml_conv_fwd_bn_add_next(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale,
                                           const float* in_shift, const float* bn_vec, const void* idn, const float* id_scale,
                                           const float* id_shift, int id_gstride, int act, void* out, uint8_t* mask_out, const void* w_next,
                                           void* y_next, double* stats_next, hipStream_t stream) {
    if (!d || !x || !w_packed || !bn_vec || !out || !w_next || !y_next) return adamml_set_error(ADAMML_EINVAL, "conv_fwd_bn_add_next: null argument");
    if (!adamml_conv_fwd_bn_add_next_supported(d, 64))
        return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_fwd_bn_add_next: 1x1 / stride-1, 64 -> 256 channels, next conv 256 -> 64 only");
    return adamml_conv1x1_fadd_next_launch(d, x, w_packed, in_scale, in_shift, bn_vec, idn, id_scale, id_shift, id_gstride, act, out, mask_out,
                                           w_next, y_next, stats_next, stream);
}

// csrc/conv1x1_fadd_next.hip: the streaming form for layer 1 (64 -> 256)
bool adamml_conv1x1_fadd_tpool_supported(const adamml_conv_desc_t* d, int frames);
int adamml_conv1x1_fadd_tpool_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                                     const float* bn_vec, const void* idn, const float* id_scale, const float* id_shift, int id_gstride, int act,
                                     int frames, void* pooled, uint16_t* code, hipStream_t stream);

int adamml_conv1x1_fadd_tpool_stream_supported(const adamml_conv_desc_t* d, int frames);
int adamml_conv1x1_fadd_tpool_stream_launch(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                                            const float* bn_vec, const void* idn, const float* id_scale, const float* id_shift, int id_gstride, int act,
                                            int frames, void* pooled, uint16_t* code, hipStream_t stream);
extern "C" int adamml_conv_fwd_bn_add_tpool_supported(const adamml_conv_desc_t* d, int frames, int act, int lazy_input) {
    if (!adamml_conv_fwd_bn_add_supported(d)) return 0;
    if (!(frames == 2 || frames == 4 || frames == 8) || d->N % frames || d->Cout % 128 || act != ADAMML_ACT_RELU) return 0;
    if (lazy_input && d->Cin > 512) return 0;                         // (the fragment-side lazy transform keeps its vectors in LDS)
    static const bool on = !(getenv("ADAMML_FADD_TPOOL") && getenv("ADAMML_FADD_TPOOL")[0] == '0') &&
                           !(getenv("ADAMML_FADD_GLDS") && atoi(getenv("ADAMML_FADD_GLDS")) < 2);
    return on ? 1 : 0;
}

// which kernel serves adamml_conv_fwd_bn_add_tpool (a label for profilers): 0 = the tile kernel's TP instance, 1 = the round-5 streaming kernel
// (csrc/conv1x1_fadd_next.hip: a wave owns all 256 channels of 16 pixels), 2 = the wave-slice streaming kernel (csrc/conv1x1_fadd_stream.hip)
extern "C" int adamml_conv_fwd_bn_add_tpool_streams(const adamml_conv_desc_t* d, int frames) {
    if (!d) return 0;
    if (adamml_conv1x1_fadd_tpool_stream_supported(d, frames)) return 2;
    return adamml_conv1x1_fadd_tpool_supported(d, frames) ? 1 : 0;
}

extern "C" int adamml_conv_fwd_bn_add_tpool(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale,
                                            const float* in_shift, const float* bn_vec, const void* idn, const float* id_scale,
                                            const float* id_shift, int id_gstride, int act, int frames, void* pooled, uint16_t* code,
                                            hipStream_t stream) {
    if (!adamml_conv_fwd_bn_add_tpool_supported(d, frames, act, in_scale != nullptr))
        return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_fwd_bn_add_tpool: 1x1 / stride-1 conv, Cout %% 128 == 0, ReLU, 2 / 4 / 8 frames per clip");
    if (!bn_vec || !idn || !pooled) return adamml_set_error(ADAMML_EINVAL, "conv_fwd_bn_add_tpool: null argument");
    if (adamml_conv1x1_fadd_tpool_stream_supported(d, frames))        // (csrc/conv1x1_fadd_stream.hip: the wave-slice streaming structure)
        return adamml_conv1x1_fadd_tpool_stream_launch(d, x, w_packed, in_scale, in_shift, bn_vec, idn, id_scale, id_shift, id_gstride, act, frames, pooled,
                                                       code, stream);
    if (adamml_conv1x1_fadd_tpool_supported(d, frames))
        return adamml_conv1x1_fadd_tpool_launch(d, x, w_packed, in_scale, in_shift, bn_vec, idn, id_scale, id_shift, id_gstride, act, frames, pooled, code,
                                                stream);
    FaddEpi f{bn_vec, idn, id_scale, id_shift, id_gstride, act, nullptr, frames, pooled, code};
    // (`pooled` doubles as the kernel's y pointer: the full-rate output is never written)
    return conv_launch(d, x, w_packed, in_scale, in_shift, pooled, nullptr, nullptr, nullptr, 0, stream, nullptr, nullptr, nullptr, nullptr, &f);
}

// Per-channel sum / sum of squares of z = W a over the pixels of each group WITHOUT z: sum z[co] = W[co,:] . s and
// sum z[co]^2 = W[co,:] G W[co,:]^T with the Gram matrix G = a^T a [Cin, Cin] and the column sums s [Cin] of the conv input
// (both over the pixels, fp32 from adamml_conv_bwd_weight_grouped / adamml_lazy_colsum).  W = the bf16 forward pack the conv
// multiplies with.  sums: [groups][2*Cout] plain doubles (nslots = 1 for adamml_bn_finalize).  One wave per (group, cout).
__global__ void gram_stats_kernel(const bf16_t* w, const float* G, const float* s, double* sums, int Cout, int Cin) {
    const int co = blockIdx.x, g = blockIdx.y, lane = threadIdx.x;
    const bf16_t* wr = w + (size_t)co * Cin;
    const float* Gg = G + (size_t)g * Cin * Cin;
    const float* sg = s + (size_t)g * Cin;
    double a1 = 0.0, a2 = 0.0;
    for (int ci = lane; ci < Cin; ci += 64) {
        const double wi = (double)__builtin_bit_cast(float, (unsigned)wr[ci] << 16);
        // t = (G w)[ci] read down COLUMN ci of the symmetric G: the 64 lanes of a load touch two contiguous lines (row-wise every lane walked
        // its own 256-byte row: 64 lines per load instruction, 67 us per launch on the forward critical path of every fused conv3), four
        // independent partial sums so that the loads of four steps are in flight together
        const float* col = Gg + ci;
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        for (int cj = 0; cj < Cin; cj += 4) {                        // (Cin % 4 == 0: 64 / 128 / 256)
            t0 += (double)col[(size_t)cj * Cin] * (double)__builtin_bit_cast(float, (unsigned)wr[cj] << 16);
            t1 += (double)col[(size_t)(cj + 1) * Cin] * (double)__builtin_bit_cast(float, (unsigned)wr[cj + 1] << 16);
            t2 += (double)col[(size_t)(cj + 2) * Cin] * (double)__builtin_bit_cast(float, (unsigned)wr[cj + 2] << 16);
            t3 += (double)col[(size_t)(cj + 3) * Cin] * (double)__builtin_bit_cast(float, (unsigned)wr[cj + 3] << 16);
        }
        a1 += wi * (double)sg[ci];
        a2 += wi * ((t0 + t1) + (t2 + t3));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { a1 += __shfl_xor(a1, off, 64); a2 += __shfl_xor(a2, off, 64); }
    if (lane == 0) {
        sums[(size_t)g * 2 * Cout + co] = a1;
        sums[(size_t)g * 2 * Cout + Cout + co] = a2;
    }
}

extern "C" int adamml_gram_stats(const void* w_packed, const float* G, const float* s, double* sums, int Cout, int Cin, int groups,
                                 hipStream_t stream) {
    if (!w_packed || !G || !s || !sums || Cout < 1 || Cin < 4 || (Cin & 3) || groups < 1) return adamml_set_error(ADAMML_EINVAL, "gram_stats: bad arguments (Cin must be a multiple of 4)");
    hipLaunchKernelGGL(gram_stats_kernel, dim3(Cout, groups), dim3(64), 0, stream, (const bf16_t*)w_packed, G, s, sums, Cout, Cin);
    return adamml_check_launch("gram_stats");
}

// Data gradient of a stride-2 conv (3x3 pad 1 / 1x1 pad 0: every strided conv of the hot path) WITHOUT the 4x wasted work
// of a zero-upsampled stride-1 conv: the dx pixels split into 4 parity classes (ih % 2, iw % 2); class (ph, pw) only
// receives the taps with kh == ph + pad (mod 2), kw likewise -- 1, 2, 2 and 4 of the 9 taps of a 3x3, 1/0/0/0 of a 1x1 --
// and is a dense stride-1 conv of dz with that tap subset whose output rows are scattered with stride 2.  A class
// without taps is zero-filled (or skipped when accumulating).
static int conv_dgrad_stride2(const adamml_conv_desc_t* d, const void* dz, const void* w, void* dx, int accumulate, double* sums,
                              const void* z_in, const float* bn_vec, int act, hipStream_t stream) {
    adamml_conv_desc_t g = *d;
    g.N = d->N; g.H = d->OH; g.W = d->OW; g.Cin = d->Cout;
    g.OH = d->H; g.OW = d->W; g.Cout = d->Cin;
    g.stride = 1; g.up = 1; g.pad = 0;
    g.act = ACT_NONE; g.accumulate = accumulate; g.in_gstride = 0;
    const int taps = d->KH * d->KW;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            DgradClass c;
            c.nt = 0; c.code = 0; c.ph = ph; c.pw = pw;
            c.OHc = (d->H - ph + 1) / 2; c.OWc = (d->W - pw + 1) / 2;
            if (c.OHc <= 0 || c.OWc <= 0) continue;
            for (int kh = 0; kh < d->KH; ++kh) {
                if ((ph + d->pad - kh) & 1) continue;
                const int dh = (ph + d->pad - kh) / 2;           // dz row = i + dh, in {0, 1} for the supported shapes
                for (int kw = 0; kw < d->KW; ++kw) {
                    if ((pw + d->pad - kw) & 1) continue;
                    const int dw = (pw + d->pad - kw) / 2;
                    const unsigned wt = (unsigned)(taps - 1 - (kh * d->KW + kw));     // tap index in the flipped weight pack
                    c.code |= ((unsigned)dh | ((unsigned)dw << 1) | (wt << 2)) << (8 * c.nt);
                    ++c.nt;
                }
            }
            if (c.nt == 0 && accumulate) continue;
            int rc = conv_launch(&g, dz, w, nullptr, nullptr, dx, sums, z_in, bn_vec, act, stream, &c);
            if (rc) return rc;
        }
    return ADAMML_OK;
}

static bool dgrad_stride2_ok(const adamml_conv_desc_t* d) {
    const bool k3 = d->KH == 3 && d->KW == 3 && d->pad == 1, k1 = d->KH == 1 && d->KW == 1 && d->pad == 0;
    return d->stride == 2 && (k3 || k1) && (d->up <= 1) && ilog2_exact(d->Cout) >= 0 &&
           d->OH == (d->H + 2 * d->pad - d->KH) / 2 + 1 && d->OW == (d->W + 2 * d->pad - d->KW) / 2 + 1;
}

extern "C" int adamml_conv_bwd_data_bn(const adamml_conv_desc_t* d, const void* dz, const void* w_dgrad_packed, void* dx,
                                       const void* z_in, const float* bn_vec, int act, double* sums, hipStream_t stream) {
    if (!d || !z_in || !bn_vec || !sums) return adamml_set_error(ADAMML_EINVAL, "conv_bwd_data_bn: null argument");
    if (dgrad_stride2_ok(d)) return conv_dgrad_stride2(d, dz, w_dgrad_packed, dx, 0, sums, z_in, bn_vec, act, stream);
    adamml_conv_desc_t g = *d;
    g.N = d->N; g.H = d->OH; g.W = d->OW; g.Cin = d->Cout;
    g.OH = d->H; g.OW = d->W; g.Cout = d->Cin;
    g.stride = 1; g.up = d->stride; g.pad = d->KH - 1 - d->pad;
    g.act = ACT_NONE; g.accumulate = 0; g.in_gstride = 0;
    return conv_launch(&g, dz, w_dgrad_packed, nullptr, nullptr, dx, sums, z_in, bn_vec, act, stream);
}

// Measured on MI355X (tools/bench_fused.py, B = 72): against apply + data gradient the dual loader wins for K = Cout <= 512
// (layer 1: 3.8 vs 4.2 ms, layer 2: 1.01 vs 1.05 ms) and loses beyond (layer 3: 0.33 vs 0.30 ms; layer 4: every one of
// the 4 cout tiles repeats the affine and the vectors no longer fit the LDS table).
extern "C" int adamml_conv_bwd_data_dual_supported(const adamml_conv_desc_t* d) {
    return d && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->Cout <= 512 && d->Cin % 8 == 0 && d->Cout % 8 == 0 ? 1 : 0;
}

extern "C" int adamml_conv_bwd_data_dual(const adamml_conv_desc_t* d, const void* g, const void* z, const float* aff, void* dz_side,
                                         const void* w_dgrad_packed, void* dx, int accumulate, const void* z_in, const float* bn_vec,
                                         int act, double* sums, hipStream_t stream) {
    if (!d || !g || !z || !aff) return adamml_set_error(ADAMML_EINVAL, "conv_bwd_data_dual: null argument");
    if (!adamml_conv_bwd_data_dual_supported(d))
        return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_data_dual: only 1x1 / stride-1 convs with Cout <= 512");
    if ((z_in != nullptr) != (bn_vec != nullptr) || (z_in != nullptr) != (sums != nullptr))
        return adamml_set_error(ADAMML_EINVAL, "conv_bwd_data_dual: incomplete BatchNorm epilogue operands");
    if (z_in && accumulate) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_data_dual: the BatchNorm epilogue does not accumulate");
    if (adamml_conv1x1_narrow_dual_supported(d))             // projection convs of the MobileNetV2s: barrier-free streaming kernel (conv1x1_narrow.hip)
        return adamml_conv1x1_narrow_dual_launch(d, g, z, aff, dz_side, w_dgrad_packed, dx, accumulate, z_in, bn_vec, act, sums, stream);
    adamml_conv_desc_t gd = *d;
    gd.N = d->N; gd.H = d->OH; gd.W = d->OW; gd.Cin = d->Cout;
    gd.OH = d->H; gd.OW = d->W; gd.Cout = d->Cin;
    gd.stride = 1; gd.up = 1; gd.pad = 0;
    gd.act = ACT_NONE; gd.accumulate = accumulate ? 1 : 0; gd.in_gstride = 0;
    DualIn di{z, aff, dz_side};
    return conv_launch(&gd, g, w_dgrad_packed, nullptr, nullptr, dx, sums, z_in, bn_vec, act, stream, nullptr, nullptr, &di);
}

bool adamml_alg_stream_supported(int Cout, int Cin);
int adamml_alg_stream_launch(const adamml_conv_desc_t* d, const void* g, const void* a, const float* a_scale, const float* a_shift,
                             const void* w_alg, const float* epi_add, void* dx, int accumulate, const void* z_in, const float* bn_vec,
                             int act, double* sums, hipStream_t stream);
static bool alg_stream_enabled() {           // ADAMML_ALG_STREAM=0: A/B aid (falls back to the CAT instance of conv_gemm_kernel)
    static int on = -1;
    if (on < 0) { const char* e = getenv("ADAMML_ALG_STREAM"); on = (e && e[0] == '0') ? 0 : 1; }
    return on == 1;
}

// ---- algebraic BatchNorm backward through a 1x1 conv z = W a followed by a linear BatchNorm (dz = A g' + B z + C per channel):
//   dx = (W^T diag(A)) g' + (W^T diag(B) W) a + W^T C,   dW = A (.) (g'^T a) + B (.) (W G) + C (x) s,  G = a^T a, s = sum_p a
// -- neither z nor dz is read or written.  Per BatchNorm group g the data gradient is ONE GEMM over the concatenated input
// [g' | a] with the weight pack [Cin][Cout + Cin] built here, plus a constant per output channel.
__global__ void alg_pack_kernel(const float* w, const float* aff, const float* m_pre, bf16_t* wp, float* cadd, int Cout, int Cin, int groups) {
    // one thread per (group, ci, k): k < Cout -> W[k][ci] * A[k]; else M[ci][k - Cout] = sum_co W[co][ci] B[co] W[co][k - Cout]
    const int K = Cout + Cin;
    const size_t total = (size_t)groups * Cin * K;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(e % K);
        const int ci = (int)((e / K) % Cin);
        const int g = (int)(e / ((size_t)K * Cin));
        const float* A = aff + (size_t)g * 3 * Cout;
        const float* B = A + Cout;
        float v;
        if (k < Cout) v = w[(size_t)k * Cin + ci] * A[k];
        else {
            const int cj = k - Cout;
            if (m_pre) v = m_pre[((size_t)g * Cin + ci) * Cin + cj];      // M_g computed by a GEMM (large Cin)
            else {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;                // (Cout % 32 == 0; split accumulators: see alg_wgrad_combine)
#pragma unroll 4
                for (int co = 0; co < Cout; co += 4) {
                    a0 = fmaf(w[(size_t)co * Cin + ci] * B[co], w[(size_t)co * Cin + cj], a0);
                    a1 = fmaf(w[(size_t)(co + 1) * Cin + ci] * B[co + 1], w[(size_t)(co + 1) * Cin + cj], a1);
                    a2 = fmaf(w[(size_t)(co + 2) * Cin + ci] * B[co + 2], w[(size_t)(co + 2) * Cin + cj], a2);
                    a3 = fmaf(w[(size_t)(co + 3) * Cin + ci] * B[co + 3], w[(size_t)(co + 3) * Cin + cj], a3);
                }
                v = (a0 + a1) + (a2 + a3);
            }
        }
        wp[e] = __builtin_bit_cast(bf16_t, (__bf16)v);
        if (k == 0) {
            const float* Cc = A + 2 * Cout;
            float acc = 0.f;
            for (int co = 0; co < Cout; ++co) acc = fmaf(w[(size_t)co * Cin + ci], Cc[co], acc);
            cadd[(size_t)g * Cin + ci] = acc;
        }
    }
}

// dW[co][ci] += sum_g  A_g[co] P_g[co][ci] + B_g[co] sum_cj W[co][cj] G_g[cj][ci] + C_g[co] s_g[ci]
__global__ void alg_wgrad_combine_kernel(const float* w, const float* aff, const float* P, const float* G, const float* wg_pre, const float* s,
                                         float* dw, int Cout, int Cin, int groups) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Cout * Cin) return;
    const int co = e / Cin, ci = e - co * Cin;
    float acc = 0.f;
    for (int g = 0; g < groups; ++g) {
        const float* A = aff + (size_t)g * 3 * Cout;
        const float* Gg = G + (size_t)g * Cin * Cin;
        float wg = 0.f;
        if (wg_pre) wg = wg_pre[(size_t)co * groups * Cin + (size_t)g * Cin + ci];        // (W G_g) computed by a GEMM (large Cin)
        else {
            // (unrolled with split accumulators: one dependent L2 round trip per cj made this 0.2 ms per layer-2 block)
            float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
            const float* wr = w + (size_t)co * Cin;
#pragma unroll 4
            for (int cj = 0; cj < Cin; cj += 4) {
                w0 = fmaf(wr[cj], Gg[(size_t)cj * Cin + ci], w0);
                w1 = fmaf(wr[cj + 1], Gg[(size_t)(cj + 1) * Cin + ci], w1);
                w2 = fmaf(wr[cj + 2], Gg[(size_t)(cj + 2) * Cin + ci], w2);
                w3 = fmaf(wr[cj + 3], Gg[(size_t)(cj + 3) * Cin + ci], w3);
            }
            wg = (w0 + w1) + (w2 + w3);
        }
        acc += A[co] * P[((size_t)g * Cout + co) * Cin + ci] + A[Cout + co] * wg + A[2 * Cout + co] * s[(size_t)g * Cin + ci];
    }
    dw[e] += acc;
}

// Second BatchNorm-backward moment from the algebraic identity  sum_p g'[p,co] z[p,co] = sum_cj W[co,cj] (g'^T a)[co,cj]  (z = W a):
// sums [groups][SLOTS][2C] holds sum(g') in its first halves (epilogues run with z == NULL leave the second halves zero);
// writes sum(g' zhat) = invstd (sum_j W (.) P - mean * sum g') into slot 0 of the second half.
__global__ void alg_sumfix_kernel(const float* w, const float* P, const float* vec, double* sums, int Cout, int Cin, int groups) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Cout * groups) return;
    const int g = e / Cout, co = e - g * Cout;
    double* sg = sums + (size_t)g * ADAMML_STAT_SLOTS * 2 * Cout;
    double s1 = 0.0;
    s1 = det_decode(sg + co, 2 * (size_t)Cout);
    const float* Pg = P + ((size_t)g * Cout + co) * Cin;
    double dot = 0.0;
    for (int cj = 0; cj < Cin; ++cj) dot += (double)w[(size_t)co * Cin + cj] * (double)Pg[cj];
    const float* v = vec + (size_t)g * 4 * Cout;
    const double r = (double)v[3 * Cout + co] * (dot - (double)v[2 * Cout + co] * s1);
    det_encode(sg + Cout + co, 2 * (size_t)Cout, r);
}

extern "C" int adamml_alg_sumfix(const float* w, const float* P, const float* vec, double* sums, int Cout, int Cin, int groups,
                                 hipStream_t stream) {
    if (!w || !P || !vec || !sums) return adamml_set_error(ADAMML_EINVAL, "alg_sumfix: null argument");
    hipLaunchKernelGGL(alg_sumfix_kernel, dim3(ceil_div(Cout * groups, 128)), dim3(128), 0, stream, w, P, vec, sums, Cout, Cin, groups);
    return adamml_check_launch("alg_sumfix");
}

extern "C" int adamml_alg_pack(const float* w, const float* aff, const float* m_pre, void* w_alg, float* epi_add, int Cout, int Cin,
                               int groups, hipStream_t stream) {
    if (!w || !aff || !w_alg || !epi_add || Cout < 1 || Cin < 1 || groups < 1) return adamml_set_error(ADAMML_EINVAL, "alg_pack: bad arguments");
    if (Cout % 4) return adamml_set_error(ADAMML_EUNSUPPORTED, "alg_pack: Cout must be a multiple of 4");
    const size_t total = (size_t)groups * Cin * (Cout + Cin);
    hipLaunchKernelGGL(alg_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, aff, m_pre, (bf16_t*)w_alg, epi_add, Cout, Cin, groups);
    return adamml_check_launch("alg_pack");
}

extern "C" int adamml_alg_wgrad_combine(const float* w, const float* aff, const float* P, const float* G, const float* wg_pre, const float* s,
                                        float* dw, int Cout, int Cin, int groups, hipStream_t stream) {
    if (!w || !aff || !P || (!G && !wg_pre) || !s || !dw) return adamml_set_error(ADAMML_EINVAL, "alg_wgrad_combine: null argument");
    if (Cin % 4 || Cout % 4) return adamml_set_error(ADAMML_EUNSUPPORTED, "alg_wgrad_combine: Cin and Cout must be multiples of 4");
    hipLaunchKernelGGL(alg_wgrad_combine_kernel, dim3(ceil_div(Cout * Cin, 256)), dim3(256), 0, stream, w, aff, P, G, wg_pre, s, dw, Cout, Cin, groups);
    return adamml_check_launch("alg_wgrad_combine");
}

extern "C" int adamml_conv_bwd_data_alg(const adamml_conv_desc_t* d, const void* g, const void* a, const float* a_scale, const float* a_shift,
                                        const void* w_alg, const float* epi_add, void* dx, int accumulate, const void* z_in,
                                        const float* bn_vec, int act, double* sums, hipStream_t stream) {
    // d describes the FORWARD conv (1x1, stride 1); d->act / d->in_gstride describe the lazy transform of its input a
    if (!d || !g || !a || !w_alg || !epi_add || !dx) return adamml_set_error(ADAMML_EINVAL, "conv_bwd_data_alg: null argument");
    if (!(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0)) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_data_alg: only 1x1 / stride-1 convs");
    if ((z_in != nullptr) != (bn_vec != nullptr) || (z_in != nullptr) != (sums != nullptr))
        return adamml_set_error(ADAMML_EINVAL, "conv_bwd_data_alg: incomplete BatchNorm epilogue operands");
    if (z_in && accumulate) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_data_alg: the BatchNorm epilogue does not accumulate");
    adamml_conv_desc_t gd = *d;
    gd.N = d->N; gd.H = d->OH; gd.W = d->OW; gd.Cin = d->Cout;
    gd.OH = d->H; gd.OW = d->W; gd.Cout = d->Cin;
    gd.stride = 1; gd.up = 1; gd.pad = 0;
    gd.act = d->act; gd.accumulate = accumulate ? 1 : 0; gd.in_gstride = d->in_gstride;
    if (adamml_alg_stream_supported(d->Cout, d->Cin) && alg_stream_enabled())
        return adamml_alg_stream_launch(d, g, a, a_scale, a_shift, w_alg, epi_add, dx, accumulate, z_in, bn_vec, act, sums, stream);
    CatIn c{a, d->Cin, (size_t)d->Cin * (d->Cout + d->Cin), epi_add};
    return conv_launch(&gd, g, w_alg, a_scale, a_shift, dx, sums, z_in, bn_vec, act, stream, nullptr, nullptr, nullptr, &c);
}

extern "C" int adamml_conv_bwd_data_res_supported(const adamml_conv_desc_t* d) {
    return d && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->Cin % 8 == 0 && d->Cout % 8 == 0 ? 1 : 0;
}

// (csrc/res_prod_stream.hip: the barrier-free streaming forms)
int adamml_res_stream_supported(const adamml_conv_desc_t* d);
int adamml_res_stream_launch(const adamml_conv_desc_t* d, const void* dz, const void* w_dgrad_packed, void* dx, const uint8_t* res_mask, double* sums_a,
                             const void* z_b, const float* vec_b, double* sums_b, hipStream_t stream);
extern "C" int adamml_conv_bwd_data_res_streams(const adamml_conv_desc_t* d) {
    return d && adamml_conv_bwd_data_res_supported(d) && adamml_res_stream_supported(d) ? 1 : 0;
}

extern "C" int adamml_conv_bwd_data_res(const adamml_conv_desc_t* d, const void* dz, const void* w_dgrad_packed, void* dx,
                                        int accumulate, const void* res_out, const uint8_t* res_mask, int res_act, const void* z_a, const float* vec_a,
                                        double* sums_a, const void* z_b, const float* vec_b, double* sums_b,
                                        hipStream_t stream) {
    if (!d || !res_out || !sums_a || (z_a && !vec_a)) return adamml_set_error(ADAMML_EINVAL, "conv_bwd_data_res: null argument");
    if ((z_b != nullptr) != (vec_b != nullptr) || (z_b != nullptr) != (sums_b != nullptr))
        return adamml_set_error(ADAMML_EINVAL, "conv_bwd_data_res: incomplete second BatchNorm operand");
    if (!adamml_conv_bwd_data_res_supported(d)) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_data_res: only 1x1 / stride-1 convs");
    if (accumulate && res_mask && !z_a && adamml_res_stream_supported(d))       // (the algebraic backward's form at the layer-2 shape)
        return adamml_res_stream_launch(d, dz, w_dgrad_packed, dx, res_mask, sums_a, z_b, vec_b, sums_b, stream);
    adamml_conv_desc_t g = *d;
    g.N = d->N; g.H = d->OH; g.W = d->OW; g.Cin = d->Cout;
    g.OH = d->H; g.OW = d->W; g.Cout = d->Cin;
    g.stride = 1; g.up = 1; g.pad = 0;
    g.act = ACT_NONE; g.accumulate = accumulate ? 1 : 0; g.in_gstride = 0;
    ResEpi r{res_out, res_mask, res_act, z_b, vec_b, sums_b};
    return conv_launch(&g, dz, w_dgrad_packed, nullptr, nullptr, dx, sums_a, z_a, vec_a, ACT_NONE, stream, nullptr, &r);
}

// (csrc/res_prod_stream.hip: the barrier-free streaming form of the layer-1 shape)
int adamml_res_prod_stream_supported(const adamml_conv_desc_t* d, int a_channels);
size_t adamml_res_prod_stream_workspace(const adamml_conv_desc_t* d);
int adamml_res_prod_stream_launch(const adamml_conv_desc_t* d, const void* dz, const void* w_dgrad_packed, void* dx, const uint8_t* res_mask,
                                  double* sums_a, const void* a, const float* a_scale, const float* a_shift, int a_act, int a_gstride,
                                  float* prod, void* workspace, size_t workspace_bytes, hipStream_t stream);

extern "C" size_t adamml_conv_bwd_data_res_prod_workspace(const adamml_conv_desc_t* d) {
    if (!d) return 0;
    if (adamml_res_prod_stream_supported(d, 64)) return adamml_res_prod_stream_workspace(d);
    const int groups = d->groups < 1 ? 1 : d->groups;
    const int n_ptiles = ceil_div(d->N * d->OH * d->OW, BP), n_ctiles = ceil_div(d->Cin, 128);      // (dgrad: the output channels are d->Cin)
    long nsp = 1536 / ((long)n_ctiles * groups);
    if (nsp < 1) nsp = 1;
    const int tpb = ceil_div(n_ptiles, (int)nsp);
    return (size_t)groups * n_ctiles * ceil_div(n_ptiles, tpb < 1 ? 1 : tpb) * 128 * 64 * sizeof(float);
}

extern "C" int adamml_conv_bwd_data_res_prod_supported(const adamml_conv_desc_t* d, int a_channels) {
    if (d && adamml_conv_bwd_data_res_supported(d) && adamml_res_prod_stream_supported(d, a_channels)) return 1;
    return d && adamml_conv_bwd_data_res_supported(d) && d->Cin % 128 == 0 && a_channels == 64 &&
           (long)ceil_div(d->N * d->OH * d->OW, BP) * (d->Cin / 128) * (d->groups < 1 ? 1 : d->groups) >= 4096 ? 1 : 0;
}

extern "C" int adamml_conv_bwd_data_res_prod_streams(const adamml_conv_desc_t* d, int a_channels) {
    return d && adamml_conv_bwd_data_res_supported(d) && adamml_res_prod_stream_supported(d, a_channels) ? 1 : 0;
}

extern "C" int adamml_conv_bwd_data_res_prod(const adamml_conv_desc_t* d, const void* dz, const void* w_dgrad_packed, void* dx,
                                             const uint8_t* res_mask, int res_act, double* sums_a, const void* a, const float* a_scale,
                                             const float* a_shift, int a_act, int a_gstride, int a_channels, float* prod,
                                             void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!adamml_conv_bwd_data_res_prod_supported(d, a_channels)) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_data_res_prod: unsupported shape");
    if (!res_mask || !sums_a || !a || !prod) return adamml_set_error(ADAMML_EINVAL, "conv_bwd_data_res_prod: null argument");
    if (adamml_res_prod_stream_supported(d, a_channels))
        return adamml_res_prod_stream_launch(d, dz, w_dgrad_packed, dx, res_mask, sums_a, a, a_scale, a_shift, a_act, a_gstride, prod, workspace,
                                             workspace_bytes, stream);
    adamml_conv_desc_t dd = *d;                          // data gradient of d: swap the channel roles, as adamml_conv_bwd_data_res does
    dd.H = d->OH; dd.W = d->OW; dd.Cin = d->Cout; dd.OH = d->H; dd.OW = d->W; dd.Cout = d->Cin;
    dd.stride = 1; dd.up = 1; dd.pad = 0; dd.act = ACT_NONE; dd.accumulate = 1; dd.in_gstride = 0;
    ResEpi r{dx, res_mask, res_act, nullptr, nullptr, nullptr};       // (res_out is never read in the mask form)
    PfIn pf{a, a_scale, a_shift, a_act, a_gstride, a_channels, prod, workspace, workspace_bytes};
    return conv_launch(&dd, dz, w_dgrad_packed, nullptr, nullptr, dx, sums_a, nullptr, nullptr, 0, stream, nullptr, &r, nullptr, nullptr, nullptr, &pf);
}

extern "C" int adamml_conv_bwd_data(const adamml_conv_desc_t* d, const void* dz, const void* w_dgrad_packed, void* dx,
                                    int accumulate, hipStream_t stream) {
    // d describes the FORWARD conv; the data gradient is a stride-1 conv of the (zero-upsampled) dz with the
    // flipped / transposed weight pack (adamml_pack_conv_weight, mode 1).
    if (!d) return adamml_set_error(ADAMML_EINVAL, "conv_bwd_data: null desc");
    if (dgrad_stride2_ok(d)) return conv_dgrad_stride2(d, dz, w_dgrad_packed, dx, accumulate, nullptr, nullptr, nullptr, 0, stream);
    adamml_conv_desc_t g = *d;
    g.N = d->N; g.H = d->OH; g.W = d->OW; g.Cin = d->Cout;
    g.OH = d->H; g.OW = d->W; g.Cout = d->Cin;
    g.stride = 1; g.up = d->stride; g.pad = d->KH - 1 - d->pad;
    g.act = ACT_NONE; g.accumulate = accumulate; g.in_gstride = 0;
    return adamml_conv_fwd(&g, dz, w_dgrad_packed, nullptr, nullptr, dx, nullptr, stream);
}

constexpr int W3_SLOTS = 4;      // 16-byte patch chunks a thread of conv3x3_wgrad_kernel stages per unit (3 x 34 pixels x 8 chunks = 816 <= 4 x 256)

// split plan shared by the workspace query and the launcher
struct WgradPlan { bool use3x3; int nsplit, per_block, n_cotiles, n_tiles, BM, BN, NK, cin_shift; int cw, rows, PR, PC, upi, upr, total_units, buf_bytes; };

static int wgrad_plan(const adamml_conv_desc_t* d, int cin_true, WgradPlan* pl) {
    const int taps = d->KH * d->KW;
    pl->use3x3 = false;
    // the LDS-patch kernel only pays on wide feature maps (measured on MI355X: 56x56 1.87 ms vs 1.97 ms generic; at
    // 28x28 and below the generic implicit-GEMM gather is 5-40 % faster)
    // (stride 1 only: a 32-column strip of a stride-2 conv needs a 3 x 65 patch, more than the staging slots of a workgroup hold)
    if (d->OW > 32 && d->KH == 3 && d->KW == 3 && d->pad == 1 && d->stride == 1 && d->Cin % 64 == 0 && d->Cout % 64 == 0 &&
        cin_true == d->Cin) {
        if (d->OW > 32) { pl->cw = 32; pl->rows = 1; pl->upr = ceil_div(d->OW, 32); }
        else { pl->cw = d->OW; pl->rows = 32 / d->OW; pl->upr = 1; }
        pl->PR = (pl->rows - 1) * d->stride + 3;
        pl->PC = (pl->cw - 1) * d->stride + 3;
        pl->upi = ceil_div(d->OH, pl->rows) * pl->upr;
        pl->total_units = d->N * pl->upi;
        pl->buf_bytes = 32 * 128 + pl->PR * pl->PC * 128;
        if (pl->PR * pl->PC * 8 <= W3_SLOTS * NTHREADS && 2 * pl->buf_bytes <= 64 * 1024 && pl->total_units > 0) {
            pl->use3x3 = true;
            pl->n_cotiles = d->Cout / 64;
            pl->n_tiles = pl->n_cotiles * (d->Cin / 64);
            int nsplit = ceil_div(512, pl->n_tiles * (d->groups < 1 ? 1 : d->groups));
            int upb = ceil_div(pl->total_units, nsplit);
            if (upb < 4) upb = 4;
            pl->nsplit = ceil_div(pl->total_units, upb);
            pl->per_block = upb;
            return 0;
        }
    }
    pl->NK = taps * d->Cin;
    pl->cin_shift = 30;                     // 1x1: tap = n >> 30 = 0
    if (taps > 1) {
        pl->cin_shift = ilog2_exact(d->Cin);
        if (pl->cin_shift < 0) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_weight: KxK conv needs power-of-two Cin (got %d)", d->Cin);
    }
    pl->BM = d->Cout <= 64 ? 64 : 128;
    pl->BN = pl->NK <= 64 ? 64 : 128;
    pl->n_cotiles = ceil_div(d->Cout, pl->BM);
    pl->n_tiles = pl->n_cotiles * ceil_div(pl->NK, pl->BN);
    const int P = d->N * d->OH * d->OW;
    int nsplit = ceil_div(768, pl->n_tiles * (d->groups < 1 ? 1 : d->groups));
    int ppb = ceil_div(ceil_div(P, nsplit), 32) * 32;
    if (ppb < 256) ppb = 256;
    pl->nsplit = ceil_div(P, ppb);
    pl->per_block = ppb;
    return 0;
}

int adamml_launch_split_reduce(const float* ws, float* dw, size_t n, int nsplit, hipStream_t stream, int taps, int cin) {
    launch_wgrad_reduce(ws, dw, n, nsplit, taps, cin, 0, 1, stream);
    return adamml_check_launch("split_reduce");
}

// per-group form: ws [groups][nsplit][n] -> out [groups][n], OVERWRITTEN (the products of the algebraic BatchNorm backward)
int adamml_launch_split_reduce_grouped(const float* ws, float* out, size_t n, int nsplit, int groups, int cin, hipStream_t stream) {
    launch_wgrad_reduce(ws, out, n, nsplit, 1, cin, 1, groups, stream);
    return adamml_check_launch("split_reduce");
}

extern "C" size_t adamml_conv_bwd_weight_workspace(const adamml_conv_desc_t* d, int cin_true) {
    WgradPlan pl;
    if (!d || wgrad_plan(d, cin_true, &pl)) return 0;
    const int groups = d->groups < 1 ? 1 : d->groups;
    size_t need = (size_t)groups * pl.nsplit * d->Cout * cin_true * d->KH * d->KW * sizeof(float);
    if (adamml_conv3x3_c64_wgrad_supported(d, cin_true)) {
        const size_t n3 = (size_t)adamml_conv3x3_c64_wgrad_blocks(d, nullptr) * d->Cout * cin_true * 9 * sizeof(float);
        if (n3 > need) need = n3;
    }
    return need;
}

struct WgradExtra { const float* dz_scale; const float* dz_shift; int dz_act, dz_gstride; bool per_group; };

static int wgrad_launch(const adamml_conv_desc_t* d, const void* dz, const void* x, const float* in_scale, const float* in_shift, float* dw,
                        int cin_true, void* workspace, size_t workspace_bytes, hipStream_t stream, const WgradExtra* ex);

extern "C" int adamml_conv_bwd_weight(const adamml_conv_desc_t* d, const void* dz, const void* x, const float* in_scale,
                                      const float* in_shift, float* dw, int cin_true, void* workspace, size_t workspace_bytes,
                                      hipStream_t stream) {
    return wgrad_launch(d, dz, x, in_scale, in_shift, dw, cin_true, workspace, workspace_bytes, stream, nullptr);
}

// Per-group products for the algebraic BatchNorm backward: out[g] = dz_g^T x_g ([groups][Cout][cin_true], OVERWRITTEN), with an
// optional lazy transform of the dz operand too (Gram matrix a^T a: dz = x = the raw tensor, both transformed).  1x1 convs.
extern "C" int adamml_conv_bwd_weight_grouped(const adamml_conv_desc_t* d, const void* dz, const float* dz_scale, const float* dz_shift,
                                              int dz_act, int dz_gstride, const void* x, const float* in_scale, const float* in_shift,
                                              float* out, int cin_true, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!d || d->KH * d->KW != 1) return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_weight_grouped: 1x1 convs only");
    if (!workspace) return adamml_set_error(ADAMML_EINVAL, "conv_bwd_weight_grouped: needs the split workspace");
    WgradExtra ex{dz_scale, dz_shift, dz_act, dz_gstride, true};
    return wgrad_launch(d, dz, x, in_scale, in_shift, out, cin_true, workspace, workspace_bytes, stream, &ex);
}

static int wgrad_launch(const adamml_conv_desc_t* d, const void* dz, const void* x, const float* in_scale, const float* in_shift, float* dw,
                        int cin_true, void* workspace, size_t workspace_bytes, hipStream_t stream, const WgradExtra* ex) {
    if (!d || !dz || !x || !dw) return adamml_set_error(ADAMML_EINVAL, "conv_bwd_weight: null argument");
    if (d->Cin % 8 || d->Cout % 8) return adamml_set_error(ADAMML_EINVAL, "conv_bwd_weight: channels must be multiples of 8");
    if ((long)d->N * d->OH * d->OW <= 0) return ADAMML_OK;
    WgradPlan pl;
    int rc = wgrad_plan(d, cin_true, &pl);
    if (rc) return rc;
    const size_t dw_numel = (size_t)d->Cout * cin_true * d->KH * d->KW;
    const int groups = d->groups < 1 ? 1 : d->groups;
    if (ex && ex->per_group && !(workspace && workspace_bytes >= (size_t)groups * pl.nsplit * dw_numel * sizeof(float)))
        return adamml_set_error(ADAMML_EINVAL, "conv_bwd_weight_grouped: workspace too small");
    if (!ex && workspace && adamml_conv3x3_c64_wgrad_supported(d, cin_true)) {
        // 3x3 / 64 -> 64: LDS-patch kernel with one partial per workgroup (conv3x3_c64.hip)
        const int nblk = adamml_conv3x3_c64_wgrad_blocks(d, nullptr);
        if (workspace_bytes >= (size_t)nblk * dw_numel * sizeof(float)) {
            rc = adamml_conv3x3_c64_wgrad_launch(d, dz, x, in_scale, in_shift, (float*)workspace, stream);
            if (rc) return rc;
            return adamml_launch_split_reduce((const float*)workspace, dw, dw_numel, nblk, stream, 9, cin_true);
        }
    }
    if (!ex && workspace && adamml_conv1x1_narrow_wgrad_supported(d, cin_true) && workspace_bytes >= (size_t)groups * pl.nsplit * dw_numel * sizeof(float)) {
        // narrow 1x1 convs of the MobileNetV2s: barrier-free streaming kernel, one partial per workgroup (conv1x1_narrow.hip)
        int nblk = 0;
        rc = adamml_conv1x1_narrow_wgrad_launch(d, dz, x, in_scale, in_shift, (float*)workspace, pl.nsplit, &nblk, stream);
        if (rc) return rc;
        return adamml_launch_split_reduce((const float*)workspace, dw, dw_numel, groups * nblk, stream, 1, cin_true);
    }
    float* ws = nullptr;
    if (workspace && workspace_bytes >= (size_t)groups * pl.nsplit * dw_numel * sizeof(float)) ws = (float*)workspace;
    dim3 grid(pl.nsplit * pl.n_tiles * groups), block(NTHREADS);
    const size_t gdz = (size_t)d->N * d->OH * d->OW * d->Cout, gx = (size_t)d->N * d->H * d->W * d->Cin;
    if (pl.use3x3) {
        W3P q;
        q.dz = (const bf16_t*)dz; q.x = (const bf16_t*)x; q.in_scale = in_scale; q.in_shift = in_shift; q.dw = dw;
        q.ws = ws; q.dw_numel = dw_numel; q.nsplit = pl.nsplit;
        q.gdz = gdz; q.gx = gx; q.in_gstride = d->in_gstride;
        q.N = d->N; q.H = d->H; q.W = d->W; q.Cin = d->Cin; q.OH = d->OH; q.OW = d->OW; q.Cout = d->Cout; q.pad = d->pad;
        q.act = d->act; q.cin_true = cin_true;
        q.cw = pl.cw; q.rows = pl.rows; q.PR = pl.PR; q.PC = pl.PC; q.units_per_img = pl.upi; q.units_per_row = pl.upr;
        q.total_units = pl.total_units; q.units_per_block = pl.per_block; q.n_cotiles = pl.n_cotiles; q.n_tiles = pl.n_tiles;
        hipLaunchKernelGGL((conv3x3_wgrad_kernel<1, W3_SLOTS>), grid, block, 2 * pl.buf_bytes, stream, q);
    } else {
        WgradP p;
        p.dz = (const bf16_t*)dz; p.x = (const bf16_t*)x; p.in_scale = in_scale; p.in_shift = in_shift; p.dw = dw;
        p.ws = ws; p.dw_numel = dw_numel; p.nsplit = pl.nsplit;
        p.gdz = gdz; p.gx = gx; p.in_gstride = d->in_gstride;
        p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout;
        p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.act = d->act; p.cin_true = cin_true;
        p.P = d->N * d->OH * d->OW; p.NK = pl.NK; p.cin_shift = pl.cin_shift; p.n_cotiles = pl.n_cotiles; p.n_tiles = pl.n_tiles;
        p.pix_per_block = pl.per_block;
        p.dz_scale = ex ? ex->dz_scale : nullptr; p.dz_shift = ex ? ex->dz_shift : nullptr;
        p.dz_act = ex ? ex->dz_act : 0; p.dz_gstride = ex ? ex->dz_gstride : 0;
        static const bool glds_on = !(getenv("ADAMML_WGRAD_GLDS") && getenv("ADAMML_WGRAD_GLDS")[0] == '0');
        static const bool lzb_on = !(getenv("ADAMML_WGRAD_LZB") && getenv("ADAMML_WGRAD_LZB")[0] == '0');
        static const bool ragged256 = !(getenv("ADAMML_WGRAD_RAGGED") && getenv("ADAMML_WGRAD_RAGGED")[0] == '0');   // measured +4 % (layer-2 3x3)
        if (glds_on && lzb_on && ws && in_scale && !(ex && ex->dz_scale) && pl.BM == 128 && pl.BN == 128 && d->KH * d->KW == 1 && d->pad == 0) {
            // 1x1 conv with a lazily normalised input: LDS-DMA staging of the raw tensor, transform at the B fragment (LZB); same tile choice
            if (d->Cout % 256 == 0 && pl.n_tiles <= 64) {
                p.n_cotiles = d->Cout / 256; p.n_tiles = p.n_cotiles * ceil_div(pl.NK, 128);
                hipLaunchKernelGGL((conv_wgrad_glds_kernel<256, 128, 2, true>), dim3(pl.nsplit * p.n_tiles * groups), block, 0, stream, p);
            } else if (d->Cout == 128 && pl.NK % 256 == 0) {
                p.n_tiles = p.n_cotiles * (pl.NK / 256);
                hipLaunchKernelGGL((conv_wgrad_glds_kernel<128, 256, 2, true>), dim3(pl.nsplit * p.n_tiles * groups), block, 0, stream, p);
            } else
            hipLaunchKernelGGL((conv_wgrad_glds_kernel<128, 128, 3, true>), grid, block, 0, stream, p);
        } else
        if (glds_on && ws && !in_scale && !(ex && ex->dz_scale) && pl.BM == 128 && pl.BN == 128) {
            // both operands plain in memory: LDS-DMA staging
            // 256-wide tiles halve the operand bytes fetched per MAC (this kernel is bound by the L1 load path: 16 KB per 128 x 128 x 32
            // step = 256 cycles of 64 B/clk against 256 cycles of MFMA).  Measured (tools/bench_conv.py, B = 72, TFLOP/s 128^2 -> wide):
            // 256 x 128 for Cout % 256 == 0: layer 3 conv1 366 -> 476, conv2 458 -> 616, downsample 329 -> 451, layer-2 downsample
            // 407 -> 512; it loses where the pixel axis is short and the tile list long (layer 4 conv2 / downsample: 376 -> 367, 366 -> 339);
            // 128 x 256 for a single cout tile: layer-2 conv1 374 -> 453.
            if (d->Cout % 256 == 0 && pl.n_tiles <= 64) {
                p.n_cotiles = d->Cout / 256; p.n_tiles = p.n_cotiles * ceil_div(pl.NK, 128);
                hipLaunchKernelGGL((conv_wgrad_glds_kernel<256, 128, 2>), dim3(pl.nsplit * p.n_tiles * groups), block, 0, stream, p);
            } else if (d->Cout == 128 && (pl.NK % 256 == 0 || (ragged256 && pl.NK > 512))) {
                // (NK % 256 != 0: the last tile is half empty -- 3x3 / 128 -> 128: 5 tiles of 256 instead of 9 of 128)
                p.n_tiles = p.n_cotiles * ceil_div(pl.NK, 256);
                hipLaunchKernelGGL((conv_wgrad_glds_kernel<128, 256, 2>), dim3(pl.nsplit * p.n_tiles * groups), block, 0, stream, p);
            } else
            hipLaunchKernelGGL((conv_wgrad_glds_kernel<128, 128, 3>), grid, block, 0, stream, p);
        } else
        if (ex && ex->dz_scale) {
            if (pl.BM == 64 && pl.BN == 64) hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 1, true>), grid, block, 0, stream, p);
            else if (pl.BM == 128 && pl.BN == 128) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 1, true>), grid, block, 0, stream, p);
            else return adamml_set_error(ADAMML_EUNSUPPORTED, "conv_bwd_weight_grouped: lazy dz needs Cout == Cin in {64, >= 128}");
        } else
        if (pl.BM == 64 && pl.BN == 64) hipLaunchKernelGGL((conv_wgrad_kernel<64, 64>), grid, block, 0, stream, p);
        else if (pl.BM == 64) hipLaunchKernelGGL((conv_wgrad_kernel<64, 128>), grid, block, 0, stream, p);
        else if (pl.BN == 64) hipLaunchKernelGGL((conv_wgrad_kernel<128, 64>), grid, block, 0, stream, p);
        else if ((long)grid.x * grid.y <= 1100) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 6>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((conv_wgrad_kernel<128, 128>), grid, block, 0, stream, p);
    }
    rc = adamml_check_launch("conv_bwd_weight");
    if (rc || !ws) return rc;
    if (ex && ex->per_group) {
        launch_wgrad_reduce(ws, dw, dw_numel, pl.nsplit, 1, cin_true, 1, groups, stream);
        return adamml_check_launch("split_reduce");
    }
    return adamml_launch_split_reduce(ws, dw, dw_numel, groups * pl.nsplit, stream, d->KH * d->KW, cin_true);
}
