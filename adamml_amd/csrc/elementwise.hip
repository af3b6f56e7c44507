// HBM-bound kernels of the AdaMML hot path (gfx950): BatchNorm finalize / apply / backward, residual add,
// pooling, global average pool, input re-layout, optimizer steps.  All activation traffic is 16 B per lane
// (8 bf16 channels of one NHWC pixel), grid-stride, with per-channel reductions staged through LDS and
// finished with fp64 global atomics.
#include "common.h"
#include "../../include/adamml_hip.h"


namespace {

constexpr int NT = 256;
constexpr int MAXC = 2048;

__host__ __device__ inline int imin(int a, int b) { return a < b ? a : b; }

static inline int grid_for(size_t work, int per_block = NT, int cap = 256 * 16) {
    size_t g = (work + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > (size_t)cap) g = cap;
    return (int)g;
}

// thread -> (pixel slot, channel chunk) mapping that keeps a thread's channel chunk fixed across its pixels
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

// Per-channel sums of a workgroup (s, q: the 8-channel partials of this thread): row slot r of the thread map deposits its partials in LDS
// row r [2C]; one thread per channel folds the rows in row order and publishes the workgroup's partial exactly (common.h: reproducible
// reductions).  rows_per_pass * 2C <= 16 NT floats = the 2 * MAXC floats every caller provides.
__device__ __forceinline__ void block_channel_publish(const float (&s)[8], const float (&q)[8], const ChanMap& m, float* smem,
                                                      int C, double* out) {
    if (m.active) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            smem[m.rslot * 2 * C + m.chunk * 8 + i] = s[i];
            smem[m.rslot * 2 * C + C + m.chunk * 8 + i] = q[i];
        }
    }
    __syncthreads();
    const unsigned slot = blockIdx.x & (ADAMML_STAT_SLOTS - 1);
    for (int i = threadIdx.x; i < 2 * C; i += NT) {
        float v = 0.f;
        for (int r = 0; r < m.rows_per_pass; ++r) v += smem[r * 2 * C + i];
        stat_publish(out + i, 2 * (size_t)C, slot, v);
    }
}

// ------------------------------------------------------------------------------------------------ BN
// Every BatchNorm-related kernel below is batched over `groups` (blockIdx.y or an in-kernel loop): tensors are
// [groups][P][C], statistic accumulators [groups][nslots][2C], BatchNorm vectors vec = [groups][4][C]
// (scale, shift, mean, invstd), backward coefficients coef = [groups][3][C].
__global__ void stats_collapse_kernel(const double* stats, double* out, int C, int groups) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * C * groups) return;
    const int g = i / (2 * C), j = i - g * 2 * C;
    // the 32 slot (or deterministic-bin) values are folded in the order of the xor-butterfly of slot_sums_groups (16, 8, 4, 2, 1):
    // a collapsed vector fed to the finalize kernels with nslots = 1 then yields bit for bit what they compute from the 32 slots
    // themselves -- a SyncBatchNorm step over ONE rank equals the plain step exactly (tests/test_rccl_gpu.py)
    double v[ADAMML_STAT_SLOTS];
    const double* st = stats + (size_t)g * ADAMML_STAT_SLOTS * 2 * C + j;
#pragma unroll
    for (int k = 0; k < ADAMML_STAT_SLOTS; ++k) v[k] = det_bin_value(st, 2 * (size_t)C, k);
#pragma unroll
    for (int off = ADAMML_STAT_SLOTS / 2; off >= 1; off >>= 1)
#pragma unroll
        for (int k = 0; k < off; ++k) v[k] += v[k + off];
    out[i] = v[0];
}

// one 32-lane group per channel: lane k reads slot k, the group folds with xor-shuffles (slot-parallel loads).
// Slot sums of up to 32 groups at once: the loads of all groups are in flight together (eight groups per batch: the S = 5 segments of the
// benchmark are ONE batch -- with four per batch the second, one-group batch was a second dependent round trip in each of the step's 314
// finalize launches), every lane of the 32-lane channel group ends up holding the sums of group g0 + k in (m1, m2): the per-group
// arithmetic that follows (fp64 divisions, square root) then runs one group per lane instead of one after the other on the lead lane.
__device__ __forceinline__ void slot_sums_groups(const double* stats, int nslots, int C, int c, int k, int g0, int groups, double& m1, double& m2) {
    m1 = 0.0; m2 = 0.0;
    const int ge = groups - g0 < 32 ? groups : g0 + 32;
    constexpr int GB = 8;
    for (int g = g0; g < ge; g += GB) {
        double a1[GB], a2[GB];
        const int nb = ge - g < GB ? ge - g : GB;           // (uniform)
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            a1[j] = 0.0; a2[j] = 0.0;
            if (j < nb && c < C && k < nslots) {
                const double* st = stats + (size_t)(g + j) * nslots * 2 * C;
                if (nslots == ADAMML_STAT_SLOTS) {              // integer bins; nslots == 1: plain doubles (collapsed / all-reduced sums)
                    a1[j] = det_bin_value(st + c, 2 * (size_t)C, k);
                    a2[j] = det_bin_value(st + C + c, 2 * (size_t)C, k);
                } else {
                    a1[j] = st[(size_t)k * 2 * C + c];
                    a2[j] = st[(size_t)k * 2 * C + C + c];
                }
            }
        }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
#pragma unroll
            for (int j = 0; j < GB; ++j) {
                if (j < nb) {
                    a1[j] += __shfl_xor(a1[j], off, 64);
                    a2[j] += __shfl_xor(a2[j], off, 64);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < GB; ++j)
            if (j < nb && k == g + j - g0) { m1 = a1[j]; m2 = a2[j]; }
    }
}

// The running statistics see the groups IN ORDER (the reference updates them once per segment call): lane g of a channel's
// 32-lane group finalises group g, the lead lane then folds the groups' (mean, unbiased variance) in order.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* stats, int nslots, int groups, double count, const float* gamma, const float* beta,
                                   float* rm, float* rv, float momentum, float eps, float* vec, int C) {
    const int c = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5), k = threadIdx.x & 31;
    float rmean = 0.f, rvar = 0.f;
    const bool lead = c < C && k == 0;
    if (lead && rm) { rmean = rm[c]; rvar = rv[c]; }
    float ga = 0.f, be = 0.f;
    if (c < C) { ga = gamma[c]; be = beta[c]; }
    for (int g0 = 0; g0 < groups; g0 += 32) {
        double s1, s2;
        slot_sums_groups(stats, nslots, C, c, k, g0, groups, s1, s2);
        const int g = g0 + k;
        double mu = s1 / count;
        double var = s2 / count - mu * mu;
        if (var < 0.0) var = 0.0;
        float is = (float)(1.0 / sqrt(var + (double)eps));
        float sc = ga * is;
        if (c < C && g < groups) {
            float* v = vec + (size_t)g * 4 * C;
            v[c] = sc;
            v[C + c] = be - (float)mu * sc;
            v[2 * C + c] = (float)mu;
            v[3 * C + c] = is;
        }
        double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        const float muf = (float)mu, unf = (float)unbiased;
        const int ng = groups - g0 < 32 ? groups - g0 : 32, base = threadIdx.x & 32;
        for (int j = 0; j < ng; ++j) {
            rmean = (1.f - momentum) * rmean + momentum * __shfl(muf, base + j, 64);
            rvar = (1.f - momentum) * rvar + momentum * __shfl(unf, base + j, 64);
        }
    }
    if (lead && rm) { rm[c] = rmean; rv[c] = rvar; }
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* scale, float* shift, int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// Row-walking elementwise kernels: a thread keeps ONE 8-channel chunk for all its pixels (ChanMap), so the per-channel
// BatchNorm vectors are loaded into registers once per thread instead of once per 16 B of activation traffic (the
// per-element form issued 14 cached vector loads per 2 streaming loads and ran at ~3.5 TB/s; this form streams at ~6).
template <bool STREAM>
__global__ __launch_bounds__(NT) void bn_act_add_kernel(const bf16_t* z, const float* scale, const float* shift, int z_gs, int act,
                                                        const bf16_t* idn, const float* id_scale, const float* id_shift, int id_gs,
                                                        bf16_t* out, uint8_t* mask_out, size_t P, int C, size_t ppb) {
    const size_t goff = (size_t)blockIdx.y * P * C;
    z += goff; out += goff;
    if (idn) idn += goff;
    if (mask_out) mask_out += goff >> 3;
    ChanMap m(C, threadIdx.x);
    if (!m.active) return;
    const int c = m.chunk * 8;
    f32x8 sc, sh, isc, ish;
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = 1.f; sh[i] = 0.f; isc[i] = 1.f; ish[i] = 0.f; }
    if (scale) { sc = load_f32x8(scale + (size_t)blockIdx.y * z_gs + c); sh = load_f32x8(shift + (size_t)blockIdx.y * z_gs + c); }
    if (idn && id_scale) { isc = load_f32x8(id_scale + (size_t)blockIdx.y * id_gs + c); ish = load_f32x8(id_shift + (size_t)blockIdx.y * id_gs + c); }
    const float lo = act_lo(act), hi = act_hi(act);
    const size_t pb = (size_t)blockIdx.x * ppb;
    const size_t pe = pb + ppb < P ? pb + ppb : P;
    auto row = [&](size_t p, bf16x8 zraw, bf16x8 iraw) {
        f32x8 v = bf8_to_f32(zraw);
        if (idn) {
            const f32x8 w = bf8_to_f32(iraw);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = clamp_act(fmaf(v[i], sc[i], sh[i]) + fmaf(w[i], isc[i], ish[i]), lo, hi);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = clamp_act(fmaf(v[i], sc[i], sh[i]), lo, hi);
        }
        if (STREAM) __builtin_nontemporal_store(f32_to_bf8(v), reinterpret_cast<bf16x8*>(out + p * C + c));
        else *reinterpret_cast<bf16x8*>(out + p * C + c) = f32_to_bf8(v);
        if (mask_out) {
            // act'(out) of the STORED value, one bit per element: the residual backward (adamml_conv_bwd_data_res) reads
            // 1/16 of the bytes it would read from `out`
            const f32x8 r = bf8_to_f32(f32_to_bf8(v));
            unsigned bits = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) bits |= (r[i] > lo && r[i] < hi) ? (1u << i) : 0u;
            mask_out[(p * C + c) >> 3] = (uint8_t)bits;
        }
    };
    auto ld = [&](const bf16_t* base, size_t p) {
        return STREAM ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(base + p * C + c)) : *reinterpret_cast<const bf16x8*>(base + p * C + c);
    };
    constexpr int U = 4;                                  // rows in flight per thread (one-shot workgroups: see adamml_bn_bwd_apply)
    const size_t step = m.rows_per_pass;
    size_t p = pb + m.rslot;
    for (; p + (U - 1) * step < pe; p += U * step) {
        bf16x8 zr[U], ir[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            zr[u] = ld(z, p + u * step);
            ir[u] = idn ? ld(idn, p + u * step) : zr[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) row(p + u * step, zr[u], ir[u]);
    }
    for (; p < pe; p += step) row(p, ld(z, p), idn ? ld(idn, p) : bf16x8{});
}

__global__ void act_bwd_from_output_kernel(const bf16_t* g_out, const bf16_t* out, int act, bf16_t* g, size_t nchunks) {
    const float lo = act_lo(act), hi = act_hi(act);
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < nchunks; e += (size_t)gridDim.x * NT) {
        f32x8 gv = bf8_to_f32(*reinterpret_cast<const bf16x8*>(g_out + e * 8));
        f32x8 ov = bf8_to_f32(*reinterpret_cast<const bf16x8*>(out + e * 8));
#pragma unroll
        for (int i = 0; i < 8; ++i) gv[i] *= mask_act(ov[i], lo, hi);
        *reinterpret_cast<bf16x8*>(g + e * 8) = f32_to_bf8(gv);
    }
}

__global__ __launch_bounds__(NT) void bn_bwd_reduce_kernel(const bf16_t* g, const bf16_t* z, const float* vec, int act, double* sums,
                                                           size_t P, int C, size_t ppb) {
    __shared__ float smem[2 * MAXC];
    g += (size_t)blockIdx.y * P * C;
    z += (size_t)blockIdx.y * P * C;
    vec += (size_t)blockIdx.y * 4 * C;
    sums += (size_t)blockIdx.y * ADAMML_STAT_SLOTS * 2 * C;
    ChanMap m(C, threadIdx.x);
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = q[i] = 0.f;
    const size_t pb = (size_t)blockIdx.x * ppb;
    const size_t pe = pb + ppb < P ? pb + ppb : P;
    if (m.active) {
        const int c = m.chunk * 8;
        f32x8 sc = load_f32x8(vec + c), sh = load_f32x8(vec + C + c), mu = load_f32x8(vec + 2 * C + c), is = load_f32x8(vec + 3 * C + c);
        const float lo = act_lo(act), hi = act_hi(act);
        for (size_t p = pb + m.rslot; p < pe; p += m.rows_per_pass) {
            f32x8 gv = bf8_to_f32(*reinterpret_cast<const bf16x8*>(g + p * C + c));
            f32x8 zv = bf8_to_f32(*reinterpret_cast<const bf16x8*>(z + p * C + c));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float gp = gv[i] * mask_act(fmaf(zv[i], sc[i], sh[i]), lo, hi);
                s[i] += gp;
                q[i] += gp * (zv[i] - mu[i]) * is[i];
            }
        }
    }
    block_channel_publish(s, q, m, smem, C, sums);
}

// Residual-add backward (models/resnet.py:110-111 / sound_mobilenet_v2.py:67): g2 = g_out * act'(out), plus the
// BatchNorm-backward sums of the one or two lazily normalised operands of the add (bn3 and, in the first block of a
// stage, the downsample BN) in the same pass over g_out / out.
template <bool STREAM>
__global__ __launch_bounds__(NT) void residual_bwd_kernel(const bf16_t* g_out, const bf16_t* out, int act, bf16_t* g2,
                                                          const bf16_t* za, const float* veca, double* sumsa,
                                                          const bf16_t* zb, const float* vecb, double* sumsb,
                                                          size_t P, int C, size_t ppb) {
    __shared__ float smem[2 * MAXC];
    {
        const size_t goff = (size_t)blockIdx.y * P * C;
        g_out += goff; out += goff; g2 += goff;
        if (za) { za += goff; veca += (size_t)blockIdx.y * 4 * C; sumsa += (size_t)blockIdx.y * ADAMML_STAT_SLOTS * 2 * C; }
        if (zb) { zb += goff; vecb += (size_t)blockIdx.y * 4 * C; sumsb += (size_t)blockIdx.y * ADAMML_STAT_SLOTS * 2 * C; }
    }
    ChanMap m(C, threadIdx.x);
    float sa[8], qa[8], sb[8], qb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sa[i] = qa[i] = sb[i] = qb[i] = 0.f;
    const size_t pb = (size_t)blockIdx.x * ppb;
    const size_t pe = pb + ppb < P ? pb + ppb : P;
    if (m.active) {
        const int c = m.chunk * 8;
        f32x8 mua, isa, mub, isb;
        if (za) { mua = load_f32x8(veca + 2 * C + c); isa = load_f32x8(veca + 3 * C + c); }
        if (zb) { mub = load_f32x8(vecb + 2 * C + c); isb = load_f32x8(vecb + 3 * C + c); }
        const float lo = act_lo(act), hi = act_hi(act);
        for (size_t p = pb + m.rslot; p < pe; p += m.rows_per_pass) {
            auto ld = [](const bf16_t* q) {
                return STREAM ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(q)) : *reinterpret_cast<const bf16x8*>(q);
            };
            f32x8 gv = bf8_to_f32(ld(g_out + p * C + c));
            const f32x8 ov = bf8_to_f32(ld(out + p * C + c));
#pragma unroll
            for (int i = 0; i < 8; ++i) gv[i] *= mask_act(ov[i], lo, hi);
            const bf16x8 gb = f32_to_bf8(gv);
            if (g2 != g_out || act != ACT_NONE) {
                if (STREAM) __builtin_nontemporal_store(gb, reinterpret_cast<bf16x8*>(g2 + p * C + c));
                else *reinterpret_cast<bf16x8*>(g2 + p * C + c) = gb;
            }
            gv = bf8_to_f32(gb);
            if (za) {
                const f32x8 zv = bf8_to_f32(ld(za + p * C + c));
#pragma unroll
                for (int i = 0; i < 8; ++i) { sa[i] += gv[i]; qa[i] += gv[i] * (zv[i] - mua[i]) * isa[i]; }
            }
            if (zb) {
                const f32x8 zv = bf8_to_f32(ld(zb + p * C + c));
#pragma unroll
                for (int i = 0; i < 8; ++i) { sb[i] += gv[i]; qb[i] += gv[i] * (zv[i] - mub[i]) * isb[i]; }
            }
        }
    }
    if (za) block_channel_publish(sa, qa, m, smem, C, sumsa);
    if (zb) {
        __syncthreads();
        block_channel_publish(sb, qb, m, smem, C, sumsb);
    }
}

__global__ void bn_bwd_finalize_kernel(const double* sums, int nslots, int groups, double count, const float* gamma, const float* vec,
                                       float* dgamma, float* dbeta, float* coef, float* aff, int C, float grad_scale) {
    const int c = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5), k = threadIdx.x & 31;
    const bool lead = c < C && k == 0;
    float dg = 0.f, db = 0.f;
    for (int g0 = 0; g0 < groups; g0 += 32) {           // lane g: coefficients of group g; lead lane: dgamma / dbeta over the groups in order
        double sg, sgz;
        slot_sums_groups(sums, nslots, C, c, k, g0, groups, sg, sgz);
        const int g = g0 + k;
        if (c < C && g < groups) {
            float* cf = coef + (size_t)g * 3 * C;
            const float* v = vec + (size_t)g * 4 * C;
            const float k0 = gamma[c] * v[3 * C + c], k1 = (float)(sg / count), k2 = (float)(sgz / count);
            cf[c] = k0;
            cf[C + c] = k1;
            cf[2 * C + c] = k2;
            if (aff) {               // dz = A g' + B z + C of bn_bwd_affine_kernel in the same launch (same expressions on the same fp32 values)
                const float mu = v[2 * C + c], is = v[3 * C + c];
                float* a = aff + (size_t)g * 3 * C;
                a[c] = k0;
                a[C + c] = -k0 * k2 * is;
                a[2 * C + c] = k0 * (k2 * mu * is - k1);
            }
        }
        const float sgf = (float)sg, sgzf = (float)sgz;
        const int ng = groups - g0 < 32 ? groups - g0 : 32, base = threadIdx.x & 32;
        for (int j = 0; j < ng; ++j) {
            dg += __shfl(sgzf, base + j, 64);
            db += __shfl(sgf, base + j, 64);
        }
    }
    if (lead) {
        if (dgamma) dgamma[c] += dg * grad_scale;
        if (dbeta) dbeta[c] += db * grad_scale;
    }
}

// dz = k0 (g' - k1 - zhat k2), zhat = (z - mu) invstd  ==  A g' + B z + C  per channel: the form the dual-source loader of
// the 1x1 data gradient applies (adamml_conv_bwd_data_dual).  coef [G][3][C], vec [G][4][C] -> aff [G][3][C].
__global__ void bn_bwd_affine_kernel(const float* coef, const float* vec, float* aff, int C, int groups) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * groups) return;
    const int g = i / C, c = i - g * C;
    const float* cf = coef + (size_t)g * 3 * C;
    const float* v = vec + (size_t)g * 4 * C;
    const float k0 = cf[c], k1 = cf[C + c], k2 = cf[2 * C + c], mu = v[2 * C + c], is = v[3 * C + c];
    float* a = aff + (size_t)g * 3 * C;
    a[c] = k0;
    a[C + c] = -k0 * k2 * is;
    a[2 * C + c] = k0 * (k2 * mu * is - k1);
}

// s[g][c] = sum over the pixels of group g of act(scale x + shift)  (column sums of a lazily normalised activation: the
// C (x) s term of the algebraic BatchNorm backward).  Row walker; fp32 atomics of per-workgroup partials (s zeroed here).
__global__ __launch_bounds__(NT) void lazy_colsum_kernel(const bf16_t* x, const float* scale, const float* shift, int gs, int act, float* s,
                                                         size_t P, int C, size_t ppb) {
    __shared__ float smem[MAXC];
    x += (size_t)blockIdx.y * P * C;
    s += (size_t)blockIdx.y * C;
    if (scale) { scale += (size_t)blockIdx.y * gs; shift += (size_t)blockIdx.y * gs; }
    ChanMap m(C, threadIdx.x);
    for (int i = threadIdx.x; i < C; i += NT) smem[i] = 0.f;
    __syncthreads();
    float accd[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (m.active) {
        const int c = m.chunk * 8;
        f32x8 acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        const size_t pb = (size_t)blockIdx.x * ppb;
        const size_t pe = pb + ppb < P ? pb + ppb : P;
        // (rounded to bf16 like the operand the conv kernels stage: s is then the exact column sum of what the MFMAs multiply)
        for (size_t p = pb + m.rslot; p < pe; p += m.rows_per_pass)
            acc += bf8_to_f32(f32_to_bf8(transform8(*reinterpret_cast<const bf16x8*>(x + p * C + c), scale, shift, c, act)));
#pragma unroll
        for (int i = 0; i < 8; ++i) accd[i] = acc[i];
    }
    {
        // one workgroup per group (see the launcher); the row slots add in turn, in a fixed order
        for (int r = 0; r < m.rows_per_pass; ++r) {
            if (m.active && m.rslot == r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) smem[m.chunk * 8 + i] += accd[i];
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += NT) atomicAdd(&s[i], smem[i]);
}

// STREAM: the tensors are larger than the 256 MB Infinity Cache -> non-temporal accesses (nothing is re-used from cache);
// smaller tensors keep default caching so that the consumers of dz (data / weight gradient) still find it in L2 / MALL.
template <bool STREAM, int U>
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(const bf16_t* g, const bf16_t* z, const float* vec, int act, const float* coef,
                                                          bf16_t* dz, size_t P, int C, size_t ppb) {
    {
        const size_t goff = (size_t)blockIdx.y * P * C;
        g += goff; z += goff; dz += goff;
        vec += (size_t)blockIdx.y * 4 * C;
        coef += (size_t)blockIdx.y * 3 * C;
    }
    ChanMap m(C, threadIdx.x);
    if (!m.active) return;
    const int c = m.chunk * 8;
    const f32x8 sc = load_f32x8(vec + c), sh = load_f32x8(vec + C + c), mu = load_f32x8(vec + 2 * C + c), is = load_f32x8(vec + 3 * C + c);
    const f32x8 k0 = load_f32x8(coef + c), k1 = load_f32x8(coef + C + c), k2 = load_f32x8(coef + 2 * C + c);
    const float lo = act_lo(act), hi = act_hi(act);
    const size_t pb = (size_t)blockIdx.x * ppb;
    const size_t pe = pb + ppb < P ? pb + ppb : P;
    auto one = [&](bf16x8 graw, bf16x8 zraw) {
        const f32x8 gv = bf8_to_f32(graw), zv = bf8_to_f32(zraw);
        f32x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float gp = gv[i] * mask_act(fmaf(zv[i], sc[i], sh[i]), lo, hi);
            float zh = (zv[i] - mu[i]) * is[i];
            o[i] = k0[i] * (gp - k1[i] - zh * k2[i]);
        }
        return f32_to_bf8(o);
    };
    const size_t step = m.rows_per_pass;
    size_t p = pb + m.rslot;
    for (; p + (U - 1) * step < pe; p += U * step) {            // U rows (2 U loads) in flight per thread
        bf16x8 gr[U], zr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bf16x8* gp = reinterpret_cast<const bf16x8*>(g + (p + u * step) * C + c);
            const bf16x8* zp = reinterpret_cast<const bf16x8*>(z + (p + u * step) * C + c);
            if (STREAM) { gr[u] = __builtin_nontemporal_load(gp); zr[u] = __builtin_nontemporal_load(zp); }
            else { gr[u] = *gp; zr[u] = *zp; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bf16x8* op = reinterpret_cast<bf16x8*>(dz + (p + u * step) * C + c);
            if (STREAM) __builtin_nontemporal_store(one(gr[u], zr[u]), op);
            else *op = one(gr[u], zr[u]);
        }
    }
    for (; p < pe; p += step) *reinterpret_cast<bf16x8*>(dz + p * C + c) = one(*reinterpret_cast<const bf16x8*>(g + p * C + c),
                                                                                 *reinterpret_cast<const bf16x8*>(z + p * C + c));
}

// ------------------------------------------------------------------------------------------------ pooling
// ZSEL: also store the RAW input value at the arg-max tap (z_sel, same shape as y).  The BatchNorm backward of the pool's input
// then takes its two sums from (g_y, z_sel) -- a quarter of the pixels -- with the plain bn_bwd_reduce kernel: every pooled
// gradient lands on exactly one input pixel, so sum(g') and sum(g' zhat) over the input equal the same sums over the windows.
template <bool ZSEL>
__global__ void maxpool_fwd_kernel(const bf16_t* x, const float* scale, const float* shift, int gs, int act, bf16_t* y,
                                   uint8_t* idx, bf16_t* zsel, int N, int H, int W, int C, int OH, int OW) {
    x += (size_t)blockIdx.y * N * H * W * C;
    y += (size_t)blockIdx.y * N * OH * OW * C;
    idx += (size_t)blockIdx.y * N * OH * OW * C;
    if (ZSEL) zsel += (size_t)blockIdx.y * N * OH * OW * C;
    if (scale) { scale += (size_t)blockIdx.y * gs; shift += (size_t)blockIdx.y * gs; }
    const int cpr = C >> 3;
    const size_t total = (size_t)N * OH * OW * cpr;
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
        const int ch = (int)(e % cpr);
        size_t pix = e / cpr;
        const int ow = (int)(pix % OW);
        pix /= OW;
        const int oh = (int)(pix % OH);
        const int n = (int)(pix / OH);
        f32x8 best;
        int bi[8];
        bf16x8 zs;
#pragma unroll
        for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; bi[i] = 0; zs[i] = 0; }
        // all nine taps are loaded UNCONDITIONALLY from clamped coordinates and the padding taps are turned into -inf afterwards: a
        // branch around each load makes the compiler wait for one tap before the next is requested (nine dependent L2 / HBM round
        // trips per output chunk: 2.6 ms at the stem shape against 1.2 ms of traffic)
        bf16x8 raw[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = min(max(oh * 2 - 1 + kh, 0), H - 1), iw = min(max(ow * 2 - 1 + kw, 0), W - 1);
                raw[kh * 3 + kw] = *reinterpret_cast<const bf16x8*>(x + (((size_t)n * H + ih) * W + iw) * C + ch * 8);
            }
        f32x8 tsc, tsh;
#pragma unroll
        for (int i = 0; i < 8; ++i) { tsc[i] = 1.f; tsh[i] = 0.f; }
        if (scale) { tsc = load_f32x8(scale + ch * 8); tsh = load_f32x8(shift + ch * 8); }
        const float tlo = scale ? act_lo(act) : -INFINITY, thi = scale ? act_hi(act) : INFINITY;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
                const bool ok = ih >= 0 && iw >= 0 && ih < H && iw < W;
                f32x8 v = bf8_to_f32(raw[kh * 3 + kw]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float t = ok ? clamp_act(fmaf(v[i], tsc[i], tsh[i]), tlo, thi) : -INFINITY;
                    if (t > best[i]) {
                        best[i] = t; bi[i] = kh * 3 + kw;
                        if (ZSEL) zs[i] = raw[kh * 3 + kw][i];
                    }
                }
            }
        *reinterpret_cast<bf16x8*>(y + e * 8) = f32_to_bf8(best);
        if (ZSEL) *reinterpret_cast<bf16x8*>(zsel + e * 8) = zs;
        uint64_t packed = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) packed |= (uint64_t)bi[i] << (8 * i);
        *reinterpret_cast<uint64_t*>(idx + e * 8) = packed;
    }
}

// Column walker of the same pool: a thread owns one 8-channel chunk of one output COLUMN and walks down `rows` output rows.  Input row
// 2oh+1 is both the bottom row of window oh and the top row of window oh+1, so its three transformed taps are carried over: six new
// loads per output instead of nine (the per-output form is bound by its nine L1 requests per 16 bytes of output, not by HBM), and the
// two rows of the next window are in flight while the current one is reduced.  Same scan order (kh, kw ascending, strict >): same
// arg-max taps.
template <bool ZSEL>
__global__ __launch_bounds__(NT) void maxpool_fwd_walk_kernel(const bf16_t* x, const float* scale, const float* shift, int gs, int act, bf16_t* y,
                                                              uint8_t* idx, bf16_t* zsel, int N, int H, int W, int C, int OH, int OW, int rows,
                                                              int nrb) {
    x += (size_t)blockIdx.y * N * H * W * C;
    y += (size_t)blockIdx.y * N * OH * OW * C;
    idx += (size_t)blockIdx.y * N * OH * OW * C;
    if (ZSEL) zsel += (size_t)blockIdx.y * N * OH * OW * C;
    if (scale) { scale += (size_t)blockIdx.y * gs; shift += (size_t)blockIdx.y * gs; }
    const int cpr = C >> 3;
    const size_t total = (size_t)N * nrb * OW * cpr;
    const size_t e = (size_t)blockIdx.x * NT + threadIdx.x;
    if (e >= total) return;
    const int ch = (int)(e % cpr);
    size_t r = e / cpr;
    const int ow = (int)(r % OW);
    r /= OW;
    const int rb = (int)(r % nrb);
    const int n = (int)(r / nrb);
    f32x8 tsc, tsh;
#pragma unroll
    for (int i = 0; i < 8; ++i) { tsc[i] = 1.f; tsh[i] = 0.f; }
    if (scale) { tsc = load_f32x8(scale + ch * 8); tsh = load_f32x8(shift + ch * 8); }
    const float tlo = scale ? act_lo(act) : -INFINITY, thi = scale ? act_hi(act) : INFINITY;
    const bf16_t* img = x + (size_t)n * H * W * C + ch * 8;
    int iwc[3];
    bool cok[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 - 1 + kw;
        cok[kw] = (unsigned)iw < (unsigned)W;
        iwc[kw] = min(max(iw, 0), W - 1);
    }
    struct Row { bf16x8 v[3]; };
    auto load_row = [&](int ih, Row& rw) {                   // unconditional, clamped
        const bf16_t* rp = img + (size_t)min(max(ih, 0), H - 1) * W * C;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) rw.v[kw] = *reinterpret_cast<const bf16x8*>(rp + (size_t)iwc[kw] * C);
    };
    auto xform = [&](const Row& rw, int ih, f32x8 (&t)[3]) {
        const bool rok = (unsigned)ih < (unsigned)H;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const f32x8 v = bf8_to_f32(rw.v[kw]);
            const bool ok = rok && cok[kw];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[kw][i] = ok ? clamp_act(fmaf(v[i], tsc[i], tsh[i]), tlo, thi) : -INFINITY;
        }
    };
    const int oh_b = rb * rows, oh_e = min(OH, oh_b + rows);
    Row top, mid, bot, nmid, nbot;
    f32x8 ttop[3], tmid[3], tbot[3];
    load_row(oh_b * 2 - 1, top);
    load_row(oh_b * 2, nmid);
    load_row(oh_b * 2 + 1, nbot);
    xform(top, oh_b * 2 - 1, ttop);
    for (int oh = oh_b; oh < oh_e; ++oh) {
        mid = nmid; bot = nbot;
        load_row(oh * 2 + 2, nmid);                          // (unconditional: load_row clamps; under `oh + 1 < oh_e` every wait of the walk
        load_row(oh * 2 + 3, nbot);                          //  behind the request was a conservative one -- appendix A-18)
        xform(mid, oh * 2, tmid);
        xform(bot, oh * 2 + 1, tbot);
        f32x8 best;
        int bi[8];
        bf16x8 zs;
#pragma unroll
        for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; bi[i] = 0; zs[i] = 0; }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (ttop[kw][i] > best[i]) { best[i] = ttop[kw][i]; bi[i] = kw; if (ZSEL) zs[i] = top.v[kw][i]; }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (tmid[kw][i] > best[i]) { best[i] = tmid[kw][i]; bi[i] = 3 + kw; if (ZSEL) zs[i] = mid.v[kw][i]; }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (tbot[kw][i] > best[i]) { best[i] = tbot[kw][i]; bi[i] = 6 + kw; if (ZSEL) zs[i] = bot.v[kw][i]; }
        const size_t o = ((((size_t)n * OH + oh) * OW + ow) * cpr + ch) * 8;
        *reinterpret_cast<bf16x8*>(y + o) = f32_to_bf8(best);
        if (ZSEL) *reinterpret_cast<bf16x8*>(zsel + o) = zs;
        uint64_t packed = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) packed |= (uint64_t)bi[i] << (8 * i);
        *reinterpret_cast<uint64_t*>(idx + o) = packed;
        top = bot;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) ttop[kw] = tbot[kw];
    }
}

__global__ void maxpool_bwd_kernel(const bf16_t* gy, const uint8_t* idx, bf16_t* gx, int N, int H, int W, int C, int OH,
                                   int OW, int accumulate) {
    const int cpr = C >> 3;
    const size_t total = (size_t)N * H * W * cpr;
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
        const int ch = (int)(e % cpr);
        size_t pix = e / cpr;
        const int iw = (int)(pix % W);
        pix /= W;
        const int ih = (int)(pix % H);
        const int n = (int)(pix / H);
        f32x8 acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        if (accumulate) acc = bf8_to_f32(*reinterpret_cast<const bf16x8*>(gx + e * 8));
        const int oh_lo = ih >> 1, oh_hi = (ih + 1) >> 1;   // windows with oh*2-1 <= ih <= oh*2+1
        const int ow_lo = iw >> 1, ow_hi = (iw + 1) >> 1;
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            if (oh >= OH) continue;
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                if (ow >= OW) continue;
                const int tap = (ih - (oh * 2 - 1)) * 3 + (iw - (ow * 2 - 1));
                const size_t o = ((((size_t)n * OH + oh) * OW + ow) * cpr + ch) * 8;
                const uint64_t packed = *reinterpret_cast<const uint64_t*>(idx + o);
                f32x8 g = bf8_to_f32(*reinterpret_cast<const bf16x8*>(gy + o));
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if ((int)((packed >> (8 * i)) & 0xff) == tap) acc[i] += g[i];
            }
        }
        *reinterpret_cast<bf16x8*>(gx + e * 8) = f32_to_bf8(acc);
    }
}

// MaxPool2d(3, 2, 1) backward fused with the BatchNorm backward of the tensor the pool reads (the ResNet stem:
// models/resnet.py:199-202 conv1 -> bn1 -> relu -> maxpool, the pool being the stem output's only consumer).  The routed
// gradient g[n,ih,iw,c] = sum over the <= 4 windows whose recorded arg-max is this pixel is cheap to recompute from the
// pooled gradient and the 1-byte indices (1/4 of the pixels), so it is never written: pass 1 (APPLY = false) accumulates
// sum(g') and sum(g' zhat), pass 2 (APPLY = true) writes dz = k0 (g' - k1 - zhat k2) -- instead of maxpool_bwd (write g),
// bn_bwd_reduce (read g, z) and bn_bwd_apply (read g, z, write dz): 29 -> 17 GB at the benchmark shape.
// A thread owns one 8-channel chunk of a 2x2 input quad (rows 2k, 2k+1; columns 2j, 2j+1): the quad is covered by exactly
// the four windows (k+a, j+b), a, b in {0, 1}, and pixel (dy, dx) belongs to window (a, b) iff a <= dy and b <= dx, at tap
// (dy - 2a + 1) * 3 + (dx - 2b + 1) -- four (index, gradient) loads per four pixels instead of 2.25 per pixel.
// EVEN: H and W even (the launcher's choice) -- every pixel of a quad exists, the four stores of a quad are unconditional code (appendix A-18)
template <bool APPLY, bool EVEN = false>
__global__ __launch_bounds__(NT) void maxpool_bwd_bn_kernel(const bf16_t* gy, const uint8_t* idx, const bf16_t* z, const float* vec, int act,
                                                            double* sums, const float* coef, bf16_t* dz, int N, int H, int W, int C,
                                                            int OH, int OW, size_t qpb) {
    __shared__ float smem[APPLY ? 1 : 2 * MAXC];
    const size_t P = (size_t)N * H * W;
    const int QH = (H + 1) >> 1, QW = (W + 1) >> 1;
    const size_t Q = (size_t)N * QH * QW;
    {
        const size_t po = (size_t)blockIdx.y * N * OH * OW * C;
        gy += po; idx += po;
        z += (size_t)blockIdx.y * P * C;
        vec += (size_t)blockIdx.y * 4 * C;
        if (APPLY) { dz += (size_t)blockIdx.y * P * C; coef += (size_t)blockIdx.y * 3 * C; }
        else sums += (size_t)blockIdx.y * ADAMML_STAT_SLOTS * 2 * C;
    }
    ChanMap m(C, threadIdx.x);
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = q[i] = 0.f;
    const size_t qb = (size_t)blockIdx.x * qpb;
    const size_t qe = qb + qpb < Q ? qb + qpb : Q;
    if (m.active) {
        const int c = m.chunk * 8;
        const f32x8 sc = load_f32x8(vec + c), sh = load_f32x8(vec + C + c), mu = load_f32x8(vec + 2 * C + c), is = load_f32x8(vec + 3 * C + c);
        f32x8 k0, k1, k2;
        if (APPLY) { k0 = load_f32x8(coef + c); k1 = load_f32x8(coef + C + c); k2 = load_f32x8(coef + 2 * C + c); }
        const float lo = act_lo(act), hi = act_hi(act);
        for (size_t qi = qb + m.rslot; qi < qe; qi += m.rows_per_pass) {
            const int n = (int)(qi / ((size_t)QH * QW));
            const int rem = (int)(qi - (size_t)n * QH * QW);
            const int k = rem / QW, j = rem - k * QW;
            // the four windows of the quad
            uint64_t wi[2][2];
            bf16x8 wg[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const bool ok = k + a < OH && j + b < OW;
                    const size_t o = (((size_t)n * OH + (ok ? k + a : 0)) * OW + (ok ? j + b : 0)) * C + c;
                    const uint64_t iv = *reinterpret_cast<const uint64_t*>(idx + o);              // (unconditional: clamped address)
                    wi[a][b] = ok ? iv : ~0ull;                                                   // 0xff never equals a tap
                    wg[a][b] = *reinterpret_cast<const bf16x8*>(gy + o);
                }
            bf16x8 zr[2][2];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const bool ok = 2 * k + dy < H && 2 * j + dx < W;
                    const size_t pp = ((size_t)n * H + (ok ? 2 * k + dy : 0)) * W + (ok ? 2 * j + dx : 0);
                    zr[dy][dx] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(z + pp * C + c));
                }
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const bool ok = 2 * k + dy < H && 2 * j + dx < W;
                    f32x8 acc;
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
                    for (int a = 0; a <= dy; ++a)
#pragma unroll
                        for (int b = 0; b <= dx; ++b) {
                            const unsigned tap = (unsigned)((dy - 2 * a + 1) * 3 + (dx - 2 * b + 1));
                            const f32x8 g = bf8_to_f32(wg[a][b]);
#pragma unroll
                            for (int i = 0; i < 8; ++i)
                                if ((unsigned)((wi[a][b] >> (8 * i)) & 0xff) == tap) acc[i] += g[i];
                        }
                    const f32x8 gv = bf8_to_f32(f32_to_bf8(acc));      // the unfused path stores the routed gradient in bf16
                    const f32x8 zv = bf8_to_f32(zr[dy][dx]);
                    if (APPLY) {
                        f32x8 o;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float gp = gv[i] * mask_act(fmaf(zv[i], sc[i], sh[i]), lo, hi);
                            const float zh = (zv[i] - mu[i]) * is[i];
                            o[i] = k0[i] * (gp - k1[i] - zh * k2[i]);
                        }
                        if (EVEN || ok) {
                            const size_t pp = ((size_t)n * H + 2 * k + dy) * W + 2 * j + dx;
                            __builtin_nontemporal_store(f32_to_bf8(o), reinterpret_cast<bf16x8*>(dz + pp * C + c));
                        }
                    } else {
                        const float keep = ok ? 1.f : 0.f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float gp = keep * gv[i] * mask_act(fmaf(zv[i], sc[i], sh[i]), lo, hi);
                            s[i] += gp;
                            q[i] += gp * (zv[i] - mu[i]) * is[i];
                        }
                    }
                }
        }
    }
    if (!APPLY) block_channel_publish(s, q, m, smem, C, sums);
}

// x: [NB, T, HWC] -> y: [NB, To, HWC], To = (T-1)/2+1
__global__ void temporal_pool_fwd_kernel(const bf16_t* x, const float* scale, const float* shift, int gs, int act, bf16_t* y,
                                         int NB, int T, int To, size_t hwc8, int C, int mode) {
    x += (size_t)blockIdx.y * NB * T * hwc8 * 8;
    y += (size_t)blockIdx.y * NB * To * hwc8 * 8;
    if (scale) { scale += (size_t)blockIdx.y * gs; shift += (size_t)blockIdx.y * gs; }
    const size_t total = (size_t)NB * To * hwc8;
    const int cpr = C >> 3;
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
        const size_t in = e % hwc8;
        size_t r = e / hwc8;
        const int to = (int)(r % To);
        const int nb = (int)(r / To);
        const int c = (int)(in % cpr) * 8;
        f32x8 accv;
#pragma unroll
        for (int i = 0; i < 8; ++i) accv[i] = mode == 0 ? -INFINITY : 0.f;
        // the three frames are loaded unconditionally (clamped frame index) and the padding frames neutralised afterwards: a branch
        // around each load serialises the three round trips
        bf16x8 raw[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int t = min(max(2 * to + k - 1, 0), T - 1);
            raw[k] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(x + (((size_t)nb * T + t) * hwc8 + in) * 8));
        }
        f32x8 tsc, tsh;
#pragma unroll
        for (int i = 0; i < 8; ++i) { tsc[i] = 1.f; tsh[i] = 0.f; }
        if (scale) { tsc = load_f32x8(scale + c); tsh = load_f32x8(shift + c); }
        const float tlo = scale ? act_lo(act) : -INFINITY, thi = scale ? act_hi(act) : INFINITY;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int t = 2 * to + k - 1;
            const bool ok = t >= 0 && t < T;
            const f32x8 v = bf8_to_f32(raw[k]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float tv = clamp_act(fmaf(v[i], tsc[i], tsh[i]), tlo, thi);
                accv[i] = mode == 0 ? fmaxf(accv[i], ok ? tv : -INFINITY) : accv[i] + (ok ? tv : 0.f);
            }
        }
        if (mode == 1) accv *= (1.f / 3.f);
        *reinterpret_cast<bf16x8*>(y + e * 8) = f32_to_bf8(accv);
    }
}

// The same pool as a column walk for the frame counts of the hot path (T = 8, 4, 2: models/resnet.py:207-210 halves the frames after
// layers 1-3): a thread owns one 8-channel chunk of one pixel for ALL T frames, requests the T rows up front, transforms each once
// and emits the To = T/2 outputs -- every input row is read once (the per-output form reads the shared odd frames twice, 1.5x the
// loads) and the frame loop is compile-time (no branches around the loads).
template <int T>
__global__ __launch_bounds__(NT) void temporal_pool_fwd_walk_kernel(const bf16_t* x, const float* scale, const float* shift, int gs, int act,
                                                                   bf16_t* y, int NB, size_t hwc8, int C, int mode) {
    constexpr int To = (T - 1) / 2 + 1;
    x += (size_t)blockIdx.y * NB * T * hwc8 * 8;
    y += (size_t)blockIdx.y * NB * To * hwc8 * 8;
    if (scale) { scale += (size_t)blockIdx.y * gs; shift += (size_t)blockIdx.y * gs; }
    const size_t total = (size_t)NB * hwc8;
    const int cpr = C >> 3;
    const float lo = scale ? act_lo(act) : -INFINITY, hi = scale ? act_hi(act) : INFINITY;
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
        const size_t in = e % hwc8, nb = e / hwc8;
        const int c = (int)(in % cpr) * 8;
        bf16x8 raw[T];
#pragma unroll
        for (int t = 0; t < T; ++t) raw[t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(x + ((nb * T + t) * hwc8 + in) * 8));
        f32x8 sc, sh;
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
        if (scale) { sc = load_f32x8(scale + c); sh = load_f32x8(shift + c); }
        f32x8 v[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            v[t] = bf8_to_f32(raw[t]);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[t][i] = clamp_act(fmaf(v[t][i], sc[i], sh[i]), lo, hi);
        }
#pragma unroll
        for (int to = 0; to < To; ++to) {
            f32x8 acc;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = mode == 0 ? -INFINITY : 0.f;
#pragma unroll
            for (int k = -1; k <= 1; ++k) {
                const int t = 2 * to + k;
                if (t < 0 || t >= T) continue;                   // (compile-time)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = mode == 0 ? fmaxf(acc[i], v[t][i]) : acc[i] + v[t][i];
            }
            if (mode == 1) acc *= (1.f / 3.f);
            __builtin_nontemporal_store(f32_to_bf8(acc), reinterpret_cast<bf16x8*>(y + ((nb * To + to) * hwc8 + in) * 8));
        }
    }
}

// gradient w.r.t. the ACTIVATED input value (the lazy transform's own backward is the producer's business)
__global__ void temporal_pool_bwd_kernel(const bf16_t* gy, const bf16_t* x, const float* scale, const float* shift, int gs, int act,
                                         bf16_t* gx, int NB, int T, int To, size_t hwc8, int C, int mode) {
    gy += (size_t)blockIdx.y * NB * To * hwc8 * 8;
    x += (size_t)blockIdx.y * NB * T * hwc8 * 8;
    gx += (size_t)blockIdx.y * NB * T * hwc8 * 8;
    if (scale) { scale += (size_t)blockIdx.y * gs; shift += (size_t)blockIdx.y * gs; }
    const size_t total = (size_t)NB * T * hwc8;
    const int cpr = C >> 3;
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
        const size_t in = e % hwc8;
        size_t r = e / hwc8;
        const int t = (int)(r % T);
        const int nb = (int)(r / T);
        const int c = (int)(in % cpr) * 8;
        f32x8 acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        const int to_lo = t >> 1, to_hi = (t + 1) >> 1;
        for (int to = to_lo; to <= to_hi; ++to) {
            if (to >= To) continue;
            f32x8 g = bf8_to_f32(*reinterpret_cast<const bf16x8*>(gy + (((size_t)nb * To + to) * hwc8 + in) * 8));
            if (mode == 1) {
                acc += g * (1.f / 3.f);
                continue;
            }
            // recompute the window's first arg-max (scan order 2to-1, 2to, 2to+1, strict >)
            f32x8 best;
            int bt[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; bt[i] = -1; }
#pragma unroll
            for (int k = -1; k <= 1; ++k) {
                const int tt = 2 * to + k;
                if (tt < 0 || tt >= T) continue;
                f32x8 v = transform8(*reinterpret_cast<const bf16x8*>(x + (((size_t)nb * T + tt) * hwc8 + in) * 8), scale, shift, c, act);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (v[i] > best[i]) { best[i] = v[i]; bt[i] = tt; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (bt[i] == t) acc[i] += g[i];
        }
        *reinterpret_cast<bf16x8*>(gx + e * 8) = f32_to_bf8(acc);
    }
}

// Temporal max-pool backward fused with the residual-add backward of the block that produced the pool's input
// (models/common.py:28-33 after models/resnet.py:110-111): the pool is the only consumer of the block output `out`, so
// its input gradient IS the block-output gradient.  One thread walks the T frames of one (clip, pixel, 8-channel chunk)
// column: window arg-maxes from `out` (first maximum in scan order 2to-1, 2to, 2to+1, as the forward kernel), routed
// gradient, activation mask act'(out), store g', and the BatchNorm-backward sums of the add's BatchNorm'd operand z.
// Replaces temporal_pool_bwd (write gx) + residual_bwd (re-read gx, out): 6.5 -> 3.5 tensor passes.
template <int T>
__global__ __launch_bounds__(NT) void temporal_pool_residual_bwd_kernel(const bf16_t* gy, const bf16_t* out, int act, bf16_t* g2,
                                                                        const bf16_t* za, const float* veca, double* sumsa,
                                                                        size_t NBHW, int HW, int C, size_t cpb) {
    constexpr int To = (T - 1) / 2 + 1;
    __shared__ float smem[2 * MAXC];
    const int NB = (int)(NBHW / HW);
    {
        const size_t goff = (size_t)blockIdx.y * NB * T * HW * C;
        gy += (size_t)blockIdx.y * NB * To * HW * C;
        out += goff; g2 += goff;
        if (za) { za += goff; veca += (size_t)blockIdx.y * 4 * C; }        // za == nullptr: sum(g') only (algebraic BatchNorm backward)
        sumsa += (size_t)blockIdx.y * ADAMML_STAT_SLOTS * 2 * C;
    }
    ChanMap m(C, threadIdx.x);
    float sa[8], qa[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sa[i] = qa[i] = 0.f;
    const size_t cb = (size_t)blockIdx.x * cpb;
    const size_t ce = cb + cpb < NBHW ? cb + cpb : NBHW;
    if (m.active) {
        const int c = m.chunk * 8;
        f32x8 mua, isa;
#pragma unroll
        for (int i = 0; i < 8; ++i) { mua[i] = 0.f; isa[i] = 0.f; }
        if (za) { mua = load_f32x8(veca + 2 * C + c); isa = load_f32x8(veca + 3 * C + c); }
        const float lo = act_lo(act), hi = act_hi(act);
        const size_t fstride = (size_t)HW * C;                    // elements between consecutive frames of a clip
        for (size_t col = cb + m.rslot; col < ce; col += m.rows_per_pass) {
            const size_t nb = col / HW, hw = col - nb * HW;
            const size_t base = (nb * T * HW + hw) * C + c, gbase = (nb * To * HW + hw) * C + c;
            bf16x8 ov[T], zv[T], gv[To];
#pragma unroll
            for (int t = 0; t < T; ++t) ov[t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(out + base + t * fstride));
#pragma unroll
            for (int t = 0; t < To; ++t) gv[t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(gy + gbase + t * fstride));
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (za) zv[t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(za + base + t * fstride));
                else zv[t] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};          // isa == 0: the second moment stays 0
            }
            int bt[To][8];
#pragma unroll
            for (int to = 0; to < To; ++to) {
                f32x8 best;
#pragma unroll
                for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; bt[to][i] = -1; }
#pragma unroll
                for (int k = -1; k <= 1; ++k) {
                    const int tt = 2 * to + k;
                    if (tt < 0 || tt >= T) continue;
                    const f32x8 v = bf8_to_f32(ov[tt]);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (v[i] > best[i]) { best[i] = v[i]; bt[to][i] = tt; }
                }
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f32x8 o = bf8_to_f32(ov[t]);
                f32x8 acc;
                const f32x8 g0 = bf8_to_f32(gv[t >> 1]);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = bt[t >> 1][i] == t ? g0[i] : 0.f;
                if ((t & 1) && ((t + 1) >> 1) < To) {
                    const f32x8 g1 = bf8_to_f32(gv[((t + 1) >> 1) < To ? ((t + 1) >> 1) : 0]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += bt[((t + 1) >> 1) < To ? ((t + 1) >> 1) : 0][i] == t ? g1[i] : 0.f;
                }
                // (the unfused path rounds the routed gradient to bf16 before masking: same here)
                acc = bf8_to_f32(f32_to_bf8(acc));
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] *= mask_act(o[i], lo, hi);
                const bf16x8 gb = f32_to_bf8(acc);
                __builtin_nontemporal_store(gb, reinterpret_cast<bf16x8*>(g2 + base + t * fstride));
                const f32x8 gq = bf8_to_f32(gb), z = bf8_to_f32(zv[t]);
#pragma unroll
                for (int i = 0; i < 8; ++i) { sa[i] += gq[i]; qa[i] += gq[i] * (z[i] - mua[i]) * isa[i]; }
            }
        }
    }
    block_channel_publish(sa, qa, m, smem, C, sumsa);
}

// The same backward when the forward fused the pool into the producing conv (adamml_conv_fwd_bn_add_tpool): the block output was never
// stored; 2 bits per pooled element say which window tap held the first maximum (3: the maximum did not pass the ReLU).  One thread
// owns the T frames of one (clip, pixel, 8-channel chunk) column: To gradient rows + To code words in, T gradient rows out.
template <int T>
__global__ __launch_bounds__(NT) void temporal_pool_code_bwd_kernel(const bf16_t* gy, const uint16_t* code, bf16_t* g2, double* sumsa,
                                                                    size_t NBHW, int HW, int C, size_t cpb) {
    constexpr int To = T / 2;
    __shared__ float smem[2 * MAXC];
    const int NB = (int)(NBHW / HW);
    {
        const size_t poff = (size_t)blockIdx.y * NB * To * HW * C;
        gy += poff;
        code += poff >> 3;
        g2 += (size_t)blockIdx.y * NB * T * HW * C;
        sumsa += (size_t)blockIdx.y * ADAMML_STAT_SLOTS * 2 * C;
    }
    ChanMap m(C, threadIdx.x);
    float sa[8], qa[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sa[i] = qa[i] = 0.f;
    const size_t cb = (size_t)blockIdx.x * cpb;
    const size_t ce = cb + cpb < NBHW ? cb + cpb : NBHW;
    if (m.active) {
        const int c = m.chunk * 8;
        const size_t fstride = (size_t)HW * C;
        for (size_t col = cb + m.rslot; col < ce; col += m.rows_per_pass) {
            const size_t nb = col / HW, hw = col - nb * HW;
            const size_t base = (nb * T * HW + hw) * C + c, gbase = (nb * To * HW + hw) * C + c;
            bf16x8 gv[To];
            unsigned cw[To];
#pragma unroll
            for (int t = 0; t < To; ++t) {
                gv[t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(gy + gbase + t * fstride));
                cw[t] = code[(gbase + t * fstride) >> 3];
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                // frame t is tap (t - (2 to - 1)) of window to: tap 1 of window t / 2 for even t; for odd t, tap 2 of window (t - 1) / 2 and tap 0 of
                // window (t + 1) / 2 (when that window exists)
                f32x8 acc;
                const int w0 = t >> 1, k0 = t - (2 * w0 - 1);
                const f32x8 g0 = bf8_to_f32(gv[w0]);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = ((cw[w0] >> (2 * i)) & 3u) == (unsigned)k0 ? g0[i] : 0.f;
                if ((t & 1) && ((t + 1) >> 1) < To) {
                    const int w1 = (t + 1) >> 1;
                    const f32x8 g1 = bf8_to_f32(gv[w1 < To ? w1 : 0]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += ((cw[w1 < To ? w1 : 0] >> (2 * i)) & 3u) == 0u ? g1[i] : 0.f;
                }
                const bf16x8 gb = f32_to_bf8(acc);
                __builtin_nontemporal_store(gb, reinterpret_cast<bf16x8*>(g2 + base + t * fstride));
                const f32x8 gq = bf8_to_f32(gb);
#pragma unroll
                for (int i = 0; i < 8; ++i) sa[i] += gq[i];
            }
        }
    }
    block_channel_publish(sa, qa, m, smem, C, sumsa);
}

// ------------------------------------------------------------------------------------------------ fused classifier head
// models/resnet.py:212-221 / models/sound_mobilenet_v2.py:155-158: AdaptiveAvgPool2d(1) -> Dropout -> Linear -> mean over the
// remaining frames of a clip, one workgroup per clip.  feat (the pooled, dropout-masked features) is kept for the backward.
__global__ __launch_bounds__(NT) void head_fwd_kernel(const bf16_t* x, const float* scale, const float* shift, int gs, int act,
                                                      const uint8_t* keep, float inv_keep, const float* W, const float* bias, float* feat,
                                                      float* logits, int clips_per_group, int T, int HW, int C, int K) {
    const int n = blockIdx.x, g = n / clips_per_group;
    if (scale) { scale += (size_t)g * gs; shift += (size_t)g * gs; }
    const int cpr = C >> 3;
    const float inv = 1.f / (float)HW;
    for (int e = threadIdx.x; e < T * cpr; e += NT) {
        const int t = e / cpr, ch = e - t * cpr;
        const size_t row = (size_t)n * T + t;
        f32x8 acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int p = 0; p < HW; ++p)
            acc += transform8(*reinterpret_cast<const bf16x8*>(x + (row * HW + p) * C + ch * 8), scale, shift, ch * 8, act);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = acc[i] * inv;
            if (keep) v = keep[row * C + ch * 8 + i] ? v * inv_keep : 0.f;
            feat[row * C + ch * 8 + i] = v;
        }
    }
    __syncthreads();                                     // this workgroup's feat rows are visible to all its waves
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* f = feat + (size_t)n * T * C;
    for (int k = wave; k < K; k += NT / 64) {
        float a = 0.f;
        for (int e = lane; e < T * C; e += 64) a = fmaf(f[e], W[(size_t)k * C + (e % C)], a);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
        if (lane == 0) logits[(size_t)n * K + k] = a / (float)T + (bias ? bias[k] : 0.f);
    }
}

// backward of the head w.r.t. the activated input of the pool: gx[n,t,hw,c] = keep * (1/(T*HW)) * sum_k g[n,k] W[k,c];
// side output gy[n*T+t, k] = g[n,k] / T (the rows of the weight-gradient GEMM gy^T feat)
__global__ __launch_bounds__(NT) void head_bwd_kernel(const float* g, const uint8_t* keep, float inv_keep, const float* W, bf16_t* gx,
                                                      float* gy, int T, int HW, int C, int K) {
    const size_t row = blockIdx.x;                        // (clip, frame)
    const size_t n = row / T;
    const int cpr = C >> 3;
    const float sc = 1.f / ((float)T * (float)HW);
    if (gy)
        for (int k = threadIdx.x; k < K; k += NT) gy[row * K + k] = g[n * K + k] / (float)T;      // (any class count: K = 400 / 1000 heads)
    for (int ch = threadIdx.x; ch < cpr; ch += NT) {
        f32x8 acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int k = 0; k < K; ++k) {
            const float gk = g[n * K + k];
            const f32x8 w = load_f32x8(W + (size_t)k * C + ch * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(gk, w[i], acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = acc[i] * sc;
            if (keep) v = keep[row * C + ch * 8 + i] ? v * inv_keep : 0.f;
            acc[i] = v;
        }
        const bf16x8 o = f32_to_bf8(acc);
        for (int p = 0; p < HW; ++p) *reinterpret_cast<bf16x8*>(gx + (row * HW + p) * C + ch * 8) = o;
    }
}

// out[c] (+)= sum_r a[r, c]: one workgroup, rows added in order (deterministic); bias gradients of the heads
__global__ void colsum_f32_kernel(const float* a, float* out, int rows, int cols, int accumulate) {
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += a[(size_t)r * cols + c];
        out[c] = accumulate ? out[c] + s : s;
    }
}

__global__ void gap_fwd_kernel(const bf16_t* x, const float* scale, const float* shift, int gs, int act, float* out, int N, int HW,
                               int C) {
    x += (size_t)blockIdx.y * N * HW * C;
    out += (size_t)blockIdx.y * N * C;
    if (scale) { scale += (size_t)blockIdx.y * gs; shift += (size_t)blockIdx.y * gs; }
    const int cpr = C >> 3;
    const int total = N * cpr;
    for (int e = blockIdx.x * NT + threadIdx.x; e < total; e += gridDim.x * NT) {
        const int ch = e % cpr, n = e / cpr;
        f32x8 acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int p = 0; p < HW; ++p)
            acc += transform8(*reinterpret_cast<const bf16x8*>(x + ((size_t)n * HW + p) * C + ch * 8), scale, shift, ch * 8, act);
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int i = 0; i < 8; ++i) out[(size_t)n * C + ch * 8 + i] = acc[i] * inv;
    }
}

__global__ void gap_bwd_kernel(const float* g, bf16_t* gx, int N, int HW, int C) {
    const int cpr = C >> 3;
    const size_t total = (size_t)N * HW * cpr;
    const float inv = 1.f / (float)HW;
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
        const int ch = (int)(e % cpr);
        const int n = (int)(e / ((size_t)HW * cpr));
        f32x8 v = load_f32x8(g + (size_t)n * C + ch * 8);
        v *= inv;
        *reinterpret_cast<bf16x8*>(gx + e * 8) = f32_to_bf8(v);
    }
}

// ------------------------------------------------------------------------------------------------ input re-layout
// x [B, S*F*C, H, W] fp32 -> y [S][B*Fk][OH][OW][c_pad] bf16; frames f = fk*frame_step; bilinear when OH != H.
__global__ void clip_to_nhwc_kernel(const float* x, bf16_t* y, int B, int S, int F, int C, int H, int W, int OH, int OW,
                                    int frame_step, int Fk, int c_pad) {
    const size_t total = (size_t)S * B * Fk * OH * OW;
    const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
    const bool resize = (OH != H) || (OW != W);
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
        size_t r = e;
        const int ow = (int)(r % OW); r /= OW;
        const int oh = (int)(r % OH); r /= OH;
        const int fk = (int)(r % Fk); r /= Fk;
        const int b = (int)(r % B);
        const int s = (int)(r / B);
        const int f = fk * frame_step;
        const float* src = x + (((size_t)b * S + s) * F + f) * C * (size_t)H * W;
        int h0 = oh, h1 = oh, w0 = ow, w1 = ow;
        float lh1 = 0.f, lw1 = 0.f;
        if (resize) {
            float fh = fmaxf(sh * (oh + 0.5f) - 0.5f, 0.f), fw = fmaxf(sw * (ow + 0.5f) - 0.5f, 0.f);
            h0 = (int)fh; w0 = (int)fw;
            h1 = h0 + (h0 < H - 1 ? 1 : 0); w1 = w0 + (w0 < W - 1 ? 1 : 0);
            lh1 = fh - h0; lw1 = fw - w0;
        }
        const float lh0 = 1.f - lh1, lw0 = 1.f - lw1;
        bf16_t* dst = y + e * c_pad;
        if (c_pad == 4) {
            // 4-channel pixels (8 bytes): the layout the 7x7 stem kernels read for <= 4 input channels -- half the bytes of the 8-channel
            // padding in this kernel's store and in the stem's forward / weight-gradient loads
            f32x4 v4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float val = 0.f;
                if (i < C) {
                    const float* pl = src + (size_t)i * H * W;
                    if (resize)
                        val = lh0 * (lw0 * pl[(size_t)h0 * W + w0] + lw1 * pl[(size_t)h0 * W + w1]) +
                              lh1 * (lw0 * pl[(size_t)h1 * W + w0] + lw1 * pl[(size_t)h1 * W + w1]);
                    else
                        val = pl[(size_t)oh * W + ow];
                }
                v4[i] = val;
            }
            *reinterpret_cast<bf16x4*>(dst) = f32_to_bf4(v4);
            continue;
        }
        for (int c8 = 0; c8 < c_pad; c8 += 8) {
            f32x8 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = c8 + i;
                float val = 0.f;
                if (c < C) {
                    const float* pl = src + (size_t)c * H * W;
                    if (resize)
                        val = lh0 * (lw0 * pl[(size_t)h0 * W + w0] + lw1 * pl[(size_t)h0 * W + w1]) +
                              lh1 * (lw0 * pl[(size_t)h1 * W + w0] + lw1 * pl[(size_t)h1 * W + w1]);
                    else
                        val = pl[(size_t)oh * W + ow];
                }
                v[i] = val;
            }
            *reinterpret_cast<bf16x8*>(dst + c8) = f32_to_bf8(v);
        }
    }
}

// The main-net RGB path of the benchmark (no resize, <= 4 channels -> 8-byte pixels, W % 4 == 0): a thread owns FOUR consecutive pixels
// of a row -- one 16-byte load per channel plane, 32 contiguous output bytes -- where the generic kernel above moves 4 bytes per lane and
// load (round 5: it ran at 3.2 TB/s for 2.9 GB).  Same values: a re-layout with one fp32 -> bf16 rounding per element.
__global__ void clip_to_nhwc4_kernel(const float* x, bf16_t* y, int B, int S, int F, int C, int H, int W, int frame_step, int Fk) {
    const int W4 = W >> 2;
    const size_t total = (size_t)S * B * Fk * H * W4;
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
        size_t r = e;
        const int w4 = (int)(r % W4); r /= W4;
        const int oh = (int)(r % H); r /= H;
        const int fk = (int)(r % Fk); r /= Fk;
        const int b = (int)(r % B);
        const int s = (int)(r / B);
        const float* src = x + (((size_t)b * S + s) * F + fk * frame_step) * C * (size_t)H * W + (size_t)oh * W + 4 * w4;
        f32x4 pl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pl[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < C) pl[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)i * H * W));
        }
        union { bf16x4 q[4]; bf16x8 h[2]; } o;
#pragma unroll
        for (int px = 0; px < 4; ++px) o.q[px] = f32_to_bf4(f32x4{pl[0][px], pl[1][px], pl[2][px], pl[3][px]});
        bf16x8* dst = reinterpret_cast<bf16x8*>(y + ((((size_t)s * B + b) * Fk + fk) * H + oh) * (size_t)W * 4 + 16 * (size_t)w4);
        dst[0] = o.h[0];
        dst[1] = o.h[1];
    }
}

// Decoded-frame input path (utils/video_transforms.py:302-343 Stack -> ToTorchFormatTensor -> GroupNormalize, then
// models/adamml.py:42-67): x [B][H][W][S*F*C] uint8 -- the HW(FC) array `Stack` produces, one byte per value instead of the
// four of the normalised fp32 tensor -- to y [S][B*Fk][OH][OW][c_pad] bf16 with value ((u8 / 255) - mean[c % nm]) / std[c % nm]
// in fp32 exactly as the reference evaluates it, bilinear (align_corners = False) when OH != H.  A thread owns one output
// pixel for every (segment, frame): it reads the 1..4 source pixels' contiguous S*F*C bytes and writes S*Fk chunks.
struct NormVec { float mean[4], std[4]; int n; };
// DIFF: the source holds C/3 + 1 consecutive RGB frames per frame group and the C output channels are the C/3 RGB differences
// of neighbours, quantised exactly as utils/video_dataset.py:32-38 does (uint8((next - cur + 255) * 0.5), truncation).
template <bool DIFF>
__global__ void clip_u8_to_nhwc_kernel(const uint8_t* x, bf16_t* y, int B, int S, int F, int C, int H, int W, int OH, int OW,
                                       int frame_step, int Fk, int c_pad, NormVec nv, int div255) {
    const size_t total = (size_t)B * OH * OW;
    const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
    const bool resize = (OH != H) || (OW != W);
    const int CS = DIFF ? C + 3 : C;                     // source channels per frame group
    const int SFC = S * F * CS;
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
        size_t r = e;
        const int ow = (int)(r % OW); r /= OW;
        const int oh = (int)(r % OH);
        const int b = (int)(r / OH);
        int h0 = oh, h1 = oh, w0 = ow, w1 = ow;
        float lh1 = 0.f, lw1 = 0.f;
        if (resize) {
            float fh = fmaxf(sh * (oh + 0.5f) - 0.5f, 0.f), fw = fmaxf(sw * (ow + 0.5f) - 0.5f, 0.f);
            h0 = (int)fh; w0 = (int)fw;
            h1 = h0 + (h0 < H - 1 ? 1 : 0); w1 = w0 + (w0 < W - 1 ? 1 : 0);
            lh1 = fh - h0; lw1 = fw - w0;
        }
        const float lh0 = 1.f - lh1, lw0 = 1.f - lw1;
        const uint8_t* p00 = x + (((size_t)b * H + h0) * W + w0) * SFC;
        const uint8_t* p01 = x + (((size_t)b * H + h0) * W + w1) * SFC;
        const uint8_t* p10 = x + (((size_t)b * H + h1) * W + w0) * SFC;
        const uint8_t* p11 = x + (((size_t)b * H + h1) * W + w1) * SFC;
        for (int s = 0; s < S; ++s)
            for (int fk = 0; fk < Fk; ++fk) {
                const int off = (s * F + fk * frame_step) * CS;
                bf16_t* dst = y + (((((size_t)s * B + b) * Fk + fk) * OH + oh) * OW + ow) * c_pad;
                const int cw = c_pad == 4 ? 4 : 8;               // 4-channel pixels: see clip_to_nhwc_kernel
                for (int c8 = 0; c8 < c_pad; c8 += 8) {
                    f32x8 v;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int c = c8 + i;
                        float val = 0.f;
                        if (c < C && i < cw) {
                            const float m = nv.mean[c % nv.n], sd = nv.std[c % nv.n];
                            auto nrm = [&](const uint8_t* q) {
                                float t = (float)q[off + c];
                                if (DIFF) t = floorf(((float)q[off + c + 3] - t + 255.f) * 0.5f);
                                if (div255) t = t / 255.f;
                                return (t - m) / sd;
                            };
                            val = resize ? lh0 * (lw0 * nrm(p00) + lw1 * nrm(p01)) + lh1 * (lw0 * nrm(p10) + lw1 * nrm(p11)) : nrm(p00);
                        }
                        v[i] = val;
                    }
                    if (cw == 4) *reinterpret_cast<bf16x4*>(dst) = f32_to_bf4(f32x4{v[0], v[1], v[2], v[3]});
                    else *reinterpret_cast<bf16x8*>(dst + c8) = f32_to_bf8(v);
                }
            }
    }
}

// The RGB case of the kernel above (C = 3 source channels, 8 frames per segment: every visual input of the reference recipes that is not
// RGB-diff) at streaming speed.  Round 5 priced the generic kernel with `bench.py --u8-input`: 11 ms per step more than the fp32 path -- it
// issues one BYTE load per value (120 per pixel, each touching 64 lines of a wave's 7.5 KB) and two fp32 divisions per value.  Here a
// thread reads its pixel's S * 24 bytes with 8-byte loads into registers (the frames' bytes are then at compile-time positions: S and
// the frame step are template parameters), and the normalisation ((u8 / 255) - mean) / std is a 3 x 256-entry fp32 table in LDS filled
// with exactly that expression -- bit-identical to the generic kernel by construction, no division per value.  Stores as there: 8 or 16
// bytes per lane, consecutive lanes = consecutive pixels of one frame plane.
template <int S, int STEP, bool RESIZE>
__global__ __launch_bounds__(NT) void clip_u8_rgb_kernel(const uint8_t* x, bf16_t* y, int B, int H, int W, int OH, int OW, int c_pad, NormVec nv,
                                                         int div255) {
    constexpr int F = 8, FK = F / STEP, NW = 3 * S;           // S * F * 3 bytes = NW 8-byte words per source pixel
    __shared__ float lut[3][256];
    for (int i = threadIdx.x; i < 3 * 256; i += NT) {
        const int c = i >> 8;
        float t = (float)(i & 255);
        if (div255) t = t / 255.f;
        lut[c][i & 255] = (t - nv.mean[c % nv.n]) / nv.std[c % nv.n];
    }
    __syncthreads();
    const size_t total = (size_t)B * OH * OW;
    const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
        size_t r = e;
        const int ow = (int)(r % OW); r /= OW;
        const int oh = (int)(r % OH);
        const int b = (int)(r / OH);
        int h0 = oh, h1 = oh, w0 = ow, w1 = ow;
        float lh1 = 0.f, lw1 = 0.f;
        if (RESIZE) {
            float fh = fmaxf(sh * (oh + 0.5f) - 0.5f, 0.f), fw = fmaxf(sw * (ow + 0.5f) - 0.5f, 0.f);
            h0 = (int)fh; w0 = (int)fw;
            h1 = h0 + (h0 < H - 1 ? 1 : 0); w1 = w0 + (w0 < W - 1 ? 1 : 0);
            lh1 = fh - h0; lw1 = fw - w0;
        }
        const float lh0 = 1.f - lh1, lw0 = 1.f - lw1;
        constexpr int NSRC = RESIZE ? 4 : 1;
        uint64_t wd[NSRC][NW];
        const uint64_t* src[4] = {reinterpret_cast<const uint64_t*>(x + (((size_t)b * H + h0) * W + w0) * (NW * 8)),
                                  reinterpret_cast<const uint64_t*>(x + (((size_t)b * H + h0) * W + w1) * (NW * 8)),
                                  reinterpret_cast<const uint64_t*>(x + (((size_t)b * H + h1) * W + w0) * (NW * 8)),
                                  reinterpret_cast<const uint64_t*>(x + (((size_t)b * H + h1) * W + w1) * (NW * 8))};
#pragma unroll
        for (int q = 0; q < NSRC; ++q)
#pragma unroll
            for (int i = 0; i < NW; ++i) wd[q][i] = src[q][i];
        auto byte_of = [&](int q, int pos) { return (unsigned)(wd[q][pos >> 3] >> (8 * (pos & 7))) & 255u; };      // pos: compile-time after unrolling
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int fk = 0; fk < FK; ++fk) {
                const int off = (s * F + fk * STEP) * 3;
                float v[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (RESIZE)
                        v[c] = lh0 * (lw0 * lut[c][byte_of(0, off + c)] + lw1 * lut[c][byte_of(1, off + c)]) +
                               lh1 * (lw0 * lut[c][byte_of(2, off + c)] + lw1 * lut[c][byte_of(3, off + c)]);
                    else
                        v[c] = lut[c][byte_of(0, off + c)];
                }
                bf16_t* dst = y + (((((size_t)s * B + b) * FK + fk) * OH + oh) * OW + ow) * c_pad;
                if (c_pad == 4) *reinterpret_cast<bf16x4*>(dst) = f32_to_bf4(f32x4{v[0], v[1], v[2], 0.f});
                else {
                    f32x8 o8 = {v[0], v[1], v[2], 0.f, 0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<bf16x8*>(dst) = f32_to_bf8(o8);
                    for (int c8 = 8; c8 < c_pad; c8 += 8) *reinterpret_cast<bf16x8*>(dst + c8) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
    }
}

template <int S>
static bool launch_u8_rgb(const uint8_t* x, bf16_t* y, int B, int H, int W, int OH, int OW, int frame_step, int c_pad, const NormVec& nv, int div255,
                          size_t n, hipStream_t stream) {
    const bool resize = OH != H || OW != W;
    const dim3 grid(grid_for(n)), block(NT);
    if (frame_step == 1 && !resize) hipLaunchKernelGGL((clip_u8_rgb_kernel<S, 1, false>), grid, block, 0, stream, x, y, B, H, W, OH, OW, c_pad, nv, div255);
    else if (frame_step == 1) hipLaunchKernelGGL((clip_u8_rgb_kernel<S, 1, true>), grid, block, 0, stream, x, y, B, H, W, OH, OW, c_pad, nv, div255);
    else if (frame_step == 2 && !resize) hipLaunchKernelGGL((clip_u8_rgb_kernel<S, 2, false>), grid, block, 0, stream, x, y, B, H, W, OH, OW, c_pad, nv, div255);
    else if (frame_step == 2) hipLaunchKernelGGL((clip_u8_rgb_kernel<S, 2, true>), grid, block, 0, stream, x, y, B, H, W, OH, OW, c_pad, nv, div255);
    else return false;
    return true;
}

// ------------------------------------------------------------------------------------------------ weights
__device__ __forceinline__ void pack_one(const float* w, void* out, int cout, int cin_true, int cin_pad, int taps, int mode, size_t e) {
    if (mode == 0) {            // [co][tap][ci]
        const int ci = (int)(e % cin_pad);
        const int tap = (int)((e / cin_pad) % taps);
        const int co = (int)(e / ((size_t)cin_pad * taps));
        float v = ci < cin_true ? w[((size_t)co * cin_true + ci) * taps + tap] : 0.f;
        reinterpret_cast<__bf16*>(out)[e] = (__bf16)v;
    } else if (mode == 1) {     // [ci][tap flipped][co]
        const int co = (int)(e % cout);
        const int tap = (int)((e / cout) % taps);
        const int ci = (int)(e / ((size_t)cout * taps));
        float v = ci < cin_true ? w[((size_t)co * cin_true + ci) * taps + (taps - 1 - tap)] : 0.f;
        reinterpret_cast<__bf16*>(out)[e] = (__bf16)v;
    } else {                    // depthwise [tap][c] fp32
        const int c = (int)(e % cout);
        const int tap = (int)(e / cout);
        reinterpret_cast<float*>(out)[e] = w[(size_t)c * taps + tap];
    }
}

// All weight packs of a backbone in ONE launch (they were ~190 launches of ~5 us at the head of every training step, on the
// critical path of each stream).  table: n rows of 6 x int64 {w, out, cout | cin_true << 32, cin_pad | kh << 32,
// kw | mode << 32, first block}; a block of PACK_EPB elements finds its row by binary search over the first-block column.
constexpr int PACK_EPB = 2048;
__global__ __launch_bounds__(NT) void pack_conv_weights_batched_kernel(const long long* table, int n) {
    int lo = 0, hi = n - 1;
    const long long b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(size_t)mid * 6 + 5] <= b) lo = mid; else hi = mid - 1;
    }
    const long long* row = table + (size_t)lo * 6;
    const float* w = reinterpret_cast<const float*>(row[0]);
    void* out = reinterpret_cast<void*>(row[1]);
    const int cout = (int)(row[2] & 0xffffffff), cin_true = (int)(row[2] >> 32), cin_pad = (int)(row[3] & 0xffffffff), kh = (int)(row[3] >> 32),
              kw = (int)(row[4] & 0xffffffff), mode = (int)(row[4] >> 32);
    const int taps = kh * kw;
    const size_t total = mode == 2 ? (size_t)taps * cout : (size_t)cout * taps * cin_pad;
    const size_t e0 = (size_t)(b - row[5]) * PACK_EPB;
    for (size_t e = e0 + threadIdx.x; e < e0 + PACK_EPB && e < total; e += NT) pack_one(w, out, cout, cin_true, cin_pad, taps, mode, e);
}

__global__ void pack_conv_weight_kernel(const float* w, void* out, int cout, int cin_true, int cin_pad, int kh, int kw, int mode) {
    const int taps = kh * kw;
    const size_t total = mode == 2 ? (size_t)taps * cout : (size_t)cout * taps * cin_pad;
    for (size_t e = (size_t)blockIdx.x * NT + threadIdx.x; e < total; e += (size_t)gridDim.x * NT) {
        pack_one(w, out, cout, cin_true, cin_pad, taps, mode, e);
    }
}

// ------------------------------------------------------------------------------------------------ optimizers
__global__ void sgd_step_kernel(float* p, const float* g, float* mom, size_t n, float lr, float momentum, float wd,
                                int nesterov, int first) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        float d = g[i] + wd * p[i];
        if (momentum != 0.f) {
            float b = first ? d : momentum * mom[i] + d;
            mom[i] = b;
            d = nesterov ? d + momentum * b : b;
        }
        p[i] -= lr * d;
    }
}

__global__ void adam_step_kernel(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2,
                                 float eps, float wd, float bc1, float bc2) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        float d = g[i] + wd * p[i];
        float mi = b1 * m[i] + (1.f - b1) * d;
        float vi = b2 * v[i] + (1.f - b2) * d * d;
        m[i] = mi;
        v[i] = vi;
        float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] -= (lr / bc1) * mi / denom;
    }
}

}  // namespace

#define CHECK_C(C, name)                                                                                     \
    if ((C) % 8 != 0 || (C) > MAXC || (C) <= 0)                                                              \
        return adamml_set_error(ADAMML_EINVAL, name ": C=%d must be a multiple of 8 in (0, %d]", (C), MAXC);

extern "C" int adamml_stats_collapse(const double* stats, double* out, int C, int groups, hipStream_t stream) {
    if (!stats || !out) return adamml_set_error(ADAMML_EINVAL, "stats_collapse: null argument");
    hipLaunchKernelGGL(stats_collapse_kernel, dim3(ceil_div(2 * C * groups, 128)), dim3(128), 0, stream, stats, out, C, groups);
    return adamml_check_launch("stats_collapse");
}

extern "C" int adamml_bn_finalize(const double* stats, int nslots, int groups, double count, const float* gamma, const float* beta,
                                  float* rm, float* rv, float momentum, float eps, float* vec, int C, hipStream_t stream) {
    if (!stats || !gamma || !beta || !vec) return adamml_set_error(ADAMML_EINVAL, "bn_finalize: null argument");
    if (nslots < 1 || nslots > ADAMML_STAT_SLOTS || groups < 1) return adamml_set_error(ADAMML_EINVAL, "bn_finalize: nslots=%d groups=%d", nslots, groups);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 8)), dim3(256), 0, stream, stats, nslots, groups, count, gamma, beta, rm, rv,
                       momentum, eps, vec, C);
    return adamml_check_launch("bn_finalize");
}

extern "C" int adamml_bn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                     float* scale, float* shift, int C, hipStream_t stream) {
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, stream, gamma, beta, rm, rv, eps, scale, shift, C);
    return adamml_check_launch("bn_eval_affine");
}

// rows per workgroup of the row-walking kernels: >= 16 passes per thread, <= ~4096 workgroups over all groups
static void rowwalk_grid(size_t P, int C, int groups, int cap_total, size_t* ppb_out, size_t* nblk_out) {
    const int rows = NT / (C / 8) > 0 ? NT / (C / 8) : 1;
    size_t ppb = (size_t)rows * 16;
    size_t nblk = (P + ppb - 1) / ppb;
    const size_t cap = cap_total / (groups < 1 ? 1 : groups) + 1;
    if (nblk > cap) { ppb = ((P + cap - 1) / cap + rows - 1) / rows * rows; nblk = (P + ppb - 1) / ppb; }
    *ppb_out = ppb;
    *nblk_out = nblk;
}

static int bn_act_add_launch(const void* z, const float* scale, const float* shift, int z_gstride, int act, const void* idn,
                                 const float* id_scale, const float* id_shift, int id_gstride, void* out, uint8_t* mask_out, size_t P, int C, int groups,
                                 hipStream_t stream) {
    CHECK_C(C, "bn_act_add");
    if (!P) return ADAMML_OK;
    if (groups < 1) groups = 1;
    size_t ppb, nblk;
    static const int aa_passes = getenv("ADAMML_ACTADD_PASSES") ? atoi(getenv("ADAMML_ACTADD_PASSES")) : 0;          // A/B aid (16: the row walk)
    const int rows = NT / (C / 8) > 0 ? NT / (C / 8) : 1;
    ppb = (size_t)rows * (aa_passes > 0 ? aa_passes : (idn && (size_t)groups * P * C * 2 > ((size_t)512 << 20) ? 4 : 8));      // one-shot workgroups (two-pass form: 8 rows)
    nblk = (P + ppb - 1) / ppb;
    if (aa_passes >= 16) rowwalk_grid(P, C, groups, 8192, &ppb, &nblk);
    if ((size_t)groups * P * C * 2 > ((size_t)256 << 20))
        hipLaunchKernelGGL(bn_act_add_kernel<true>, dim3((unsigned)nblk, groups), dim3(NT), 0, stream, (const bf16_t*)z, scale, shift, z_gstride,
                           act, (const bf16_t*)idn, id_scale, id_shift, id_gstride, (bf16_t*)out, mask_out, P, C, ppb);
    else
        hipLaunchKernelGGL(bn_act_add_kernel<false>, dim3((unsigned)nblk, groups), dim3(NT), 0, stream, (const bf16_t*)z, scale, shift, z_gstride,
                           act, (const bf16_t*)idn, id_scale, id_shift, id_gstride, (bf16_t*)out, mask_out, P, C, ppb);
    return adamml_check_launch("bn_act_add");
}

extern "C" int adamml_bn_act_add(const void* z, const float* scale, const float* shift, int z_gstride, int act, const void* idn,
                                 const float* id_scale, const float* id_shift, int id_gstride, void* out, size_t P, int C, int groups,
                                 hipStream_t stream) {
    return bn_act_add_launch(z, scale, shift, z_gstride, act, idn, id_scale, id_shift, id_gstride, out, nullptr, P, C, groups, stream);
}

extern "C" int adamml_bn_act_add_mask(const void* z, const float* scale, const float* shift, int z_gstride, int act, const void* idn,
                                      const float* id_scale, const float* id_shift, int id_gstride, void* out, uint8_t* mask_out,
                                      size_t P, int C, int groups, hipStream_t stream) {
    return bn_act_add_launch(z, scale, shift, z_gstride, act, idn, id_scale, id_shift, id_gstride, out, mask_out, P, C, groups, stream);
}

extern "C" int adamml_act_bwd_from_output(const void* g_out, const void* out, int act, void* g, size_t n, hipStream_t stream) {
    if (n % 8) return adamml_set_error(ADAMML_EINVAL, "act_bwd_from_output: n must be a multiple of 8");
    if (!n) return ADAMML_OK;
    hipLaunchKernelGGL(act_bwd_from_output_kernel, dim3(grid_for(n / 8)), dim3(NT), 0, stream, (const bf16_t*)g_out,
                       (const bf16_t*)out, act, (bf16_t*)g, n / 8);
    return adamml_check_launch("act_bwd_from_output");
}

static void reduce_grid(size_t P, int C, int groups, size_t* ppb_out, size_t* nblk_out) {
    rowwalk_grid(P, C, groups, 2048, ppb_out, nblk_out);
}

extern "C" int adamml_bn_bwd_reduce(const void* g, const void* z, const float* vec, int act, double* sums, size_t P, int C, int groups,
                                    hipStream_t stream) {
    CHECK_C(C, "bn_bwd_reduce");
    if (!P) return ADAMML_OK;
    if (groups < 1) groups = 1;
    size_t ppb, nblk;
    reduce_grid(P, C, groups, &ppb, &nblk);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)nblk, groups), dim3(NT), 0, stream, (const bf16_t*)g, (const bf16_t*)z, vec, act,
                       sums, P, C, ppb);
    return adamml_check_launch("bn_bwd_reduce");
}

extern "C" int adamml_residual_bwd(const void* g_out, const void* out, int act, void* g2, const void* za, const float* veca,
                                   double* sumsa, const void* zb, const float* vecb, double* sumsb, size_t P, int C, int groups,
                                   hipStream_t stream) {
    CHECK_C(C, "residual_bwd");
    if (!P) return ADAMML_OK;
    if (groups < 1) groups = 1;
    if ((za && (!veca || !sumsa)) || (zb && (!vecb || !sumsb))) return adamml_set_error(ADAMML_EINVAL, "residual_bwd: null BN operands");
    size_t ppb, nblk;
    reduce_grid(P, C, groups, &ppb, &nblk);
    if ((size_t)groups * P * C * 2 > ((size_t)256 << 20))
        hipLaunchKernelGGL(residual_bwd_kernel<true>, dim3((unsigned)nblk, groups), dim3(NT), 0, stream, (const bf16_t*)g_out, (const bf16_t*)out,
                           act, (bf16_t*)g2, (const bf16_t*)za, veca, sumsa, (const bf16_t*)zb, vecb, sumsb, P, C, ppb);
    else
        hipLaunchKernelGGL(residual_bwd_kernel<false>, dim3((unsigned)nblk, groups), dim3(NT), 0, stream, (const bf16_t*)g_out, (const bf16_t*)out,
                           act, (bf16_t*)g2, (const bf16_t*)za, veca, sumsa, (const bf16_t*)zb, vecb, sumsb, P, C, ppb);
    return adamml_check_launch("residual_bwd");
}

extern "C" int adamml_bn_bwd_finalize(const double* sums, int nslots, int groups, double count, const float* gamma, const float* vec,
                                      float* dgamma, float* dbeta, float* coef, int C, float grad_scale, hipStream_t stream) {
    if (nslots < 1 || nslots > ADAMML_STAT_SLOTS || groups < 1) return adamml_set_error(ADAMML_EINVAL, "bn_bwd_finalize: nslots=%d groups=%d", nslots, groups);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 8)), dim3(256), 0, stream, sums, nslots, groups, count, gamma, vec, dgamma,
                       dbeta, coef, (float*)nullptr, C, grad_scale);
    return adamml_check_launch("bn_bwd_finalize");
}

extern "C" int adamml_bn_bwd_finalize_affine(const double* sums, int nslots, int groups, double count, const float* gamma, const float* vec,
                                             float* dgamma, float* dbeta, float* coef, float* aff, int C, float grad_scale, hipStream_t stream) {
    if (nslots < 1 || nslots > ADAMML_STAT_SLOTS || groups < 1) return adamml_set_error(ADAMML_EINVAL, "bn_bwd_finalize_affine: nslots=%d groups=%d", nslots, groups);
    if (!coef || !aff) return adamml_set_error(ADAMML_EINVAL, "bn_bwd_finalize_affine: null argument");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 8)), dim3(256), 0, stream, sums, nslots, groups, count, gamma, vec, dgamma,
                       dbeta, coef, aff, C, grad_scale);
    return adamml_check_launch("bn_bwd_finalize_affine");
}

extern "C" int adamml_lazy_colsum(const void* x, const float* scale, const float* shift, int gstride, int act, float* s, size_t P, int C,
                                  int groups, hipStream_t stream) {
    CHECK_C(C, "lazy_colsum");
    if (!x || !s) return adamml_set_error(ADAMML_EINVAL, "lazy_colsum: null argument");
    if (groups < 1) groups = 1;
    (void)hipMemsetAsync(s, 0, (size_t)groups * C * sizeof(float), stream);
    if (!P) return ADAMML_OK;
    size_t ppb, nblk;
    reduce_grid(P, C, groups, &ppb, &nblk);
    ppb = P; nblk = 1;             // one workgroup per group: its fp32 adds run in a fixed order (fallback path only: adamml_gram_colsum
                                   // serves the channel counts of the model)
    hipLaunchKernelGGL(lazy_colsum_kernel, dim3((unsigned)nblk, groups), dim3(NT), 0, stream, (const bf16_t*)x, scale, shift, gstride, act, s, P, C, ppb);
    return adamml_check_launch("lazy_colsum");
}

extern "C" int adamml_bn_bwd_affine(const float* coef, const float* vec, float* aff, int C, int groups, hipStream_t stream) {
    if (!coef || !vec || !aff || C < 1 || groups < 1) return adamml_set_error(ADAMML_EINVAL, "bn_bwd_affine: bad arguments");
    hipLaunchKernelGGL(bn_bwd_affine_kernel, dim3(ceil_div(C * groups, 256)), dim3(256), 0, stream, coef, vec, aff, C, groups);
    return adamml_check_launch("bn_bwd_affine");
}

extern "C" int adamml_bn_bwd_apply(const void* g, const void* z, const float* vec, int act, const float* coef, void* dz, size_t P, int C,
                                   int groups, hipStream_t stream) {
    CHECK_C(C, "bn_bwd_apply");
    if (!P) return ADAMML_OK;
    if (groups < 1) groups = 1;
    size_t ppb, nblk;
    // One-shot workgroups: every thread loads U = 4 rows of (g, z), stores them and (for tensors beyond 512 MB) retires -- 6.2-6.4 TB/s
    // against 5.3-5.5 for the long row walk (>= 16 rows per thread, <= 8192 workgroups) and 6.05 for torch's add on the same tensors:
    // with nothing to amortise (the 7 per-channel vectors are 28 cached loads per thread) the short-lived form keeps more requests
    // in flight across workgroup turnover.  Measured (tools/bench_elementwise.py): U / rows per thread 2/2 5.0, 4/4 6.2, 4/8 5.9-6.4,
    // 8/8 5.8-6.3, 8/16 5.1-5.7 TB/s; default caching instead of non-temporal accesses: -4 %.
    const int rows = NT / (C / 8) > 0 ? NT / (C / 8) : 1;
    ppb = (size_t)rows * ((size_t)groups * P * C * 2 > ((size_t)512 << 20) ? 4 : 8);
    nblk = (P + ppb - 1) / ppb;
    static const bool walk = getenv("ADAMML_BNAPPLY_WALK") && atoi(getenv("ADAMML_BNAPPLY_WALK"));          // A/B aid: the long row walk
    if (walk) rowwalk_grid(P, C, groups, 8192, &ppb, &nblk);
    if ((size_t)groups * P * C * 2 > ((size_t)256 << 20))
        hipLaunchKernelGGL((bn_bwd_apply_kernel<true, 4>), dim3((unsigned)nblk, groups), dim3(NT), 0, stream, (const bf16_t*)g, (const bf16_t*)z,
                           vec, act, coef, (bf16_t*)dz, P, C, ppb);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<false, 4>), dim3((unsigned)nblk, groups), dim3(NT), 0, stream, (const bf16_t*)g, (const bf16_t*)z,
                           vec, act, coef, (bf16_t*)dz, P, C, ppb);
    return adamml_check_launch("bn_bwd_apply");
}

extern "C" int adamml_maxpool2d_fwd(const void* x, const float* scale, const float* shift, int gstride, int act, void* y, uint8_t* idx,
                                    void* z_sel, int N, int H, int W, int C, int OH, int OW, int groups, hipStream_t stream) {
    CHECK_C(C, "maxpool2d_fwd");
    const size_t n = (size_t)N * OH * OW * (C / 8);
    if (!n) return ADAMML_OK;
    if (groups < 1) groups = 1;
    static const int walk = getenv("ADAMML_MAXPOOL_WALK") ? atoi(getenv("ADAMML_MAXPOOL_WALK")) : 8;          // output rows per thread; 0: per-output kernel (A/B aid)
    if (walk > 0 && OH >= 2 * walk) {
        const int nrb = (OH + walk - 1) / walk;
        const size_t nth = (size_t)N * nrb * OW * (C / 8);
        const dim3 grid((unsigned)((nth + NT - 1) / NT), groups);
        if (z_sel) hipLaunchKernelGGL(maxpool_fwd_walk_kernel<true>, grid, dim3(NT), 0, stream, (const bf16_t*)x, scale, shift, gstride, act, (bf16_t*)y,
                                      idx, (bf16_t*)z_sel, N, H, W, C, OH, OW, walk, nrb);
        else hipLaunchKernelGGL(maxpool_fwd_walk_kernel<false>, grid, dim3(NT), 0, stream, (const bf16_t*)x, scale, shift, gstride, act, (bf16_t*)y,
                                idx, nullptr, N, H, W, C, OH, OW, walk, nrb);
        return adamml_check_launch("maxpool2d_fwd");
    }
    if (z_sel)
        hipLaunchKernelGGL(maxpool_fwd_kernel<true>, dim3(grid_for(n, NT, 4096 / groups + 1), groups), dim3(NT), 0, stream, (const bf16_t*)x,
                           scale, shift, gstride, act, (bf16_t*)y, idx, (bf16_t*)z_sel, N, H, W, C, OH, OW);
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel<false>, dim3(grid_for(n, NT, 4096 / groups + 1), groups), dim3(NT), 0, stream, (const bf16_t*)x,
                           scale, shift, gstride, act, (bf16_t*)y, idx, nullptr, N, H, W, C, OH, OW);
    return adamml_check_launch("maxpool2d_fwd");
}

extern "C" int adamml_maxpool2d_bwd(const void* g_y, const uint8_t* idx, void* g_x, int N, int H, int W, int C, int OH, int OW,
                                    int accumulate, hipStream_t stream) {
    CHECK_C(C, "maxpool2d_bwd");
    const size_t n = (size_t)N * H * W * (C / 8);
    if (!n) return ADAMML_OK;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(n)), dim3(NT), 0, stream, (const bf16_t*)g_y, idx, (bf16_t*)g_x, N, H, W, C,
                       OH, OW, accumulate);
    return adamml_check_launch("maxpool2d_bwd");
}

extern "C" int adamml_maxpool2d_bwd_bn_reduce(const void* g_y, const uint8_t* idx, const void* z, const float* vec, int act, double* sums,
                                              int N, int H, int W, int C, int OH, int OW, int groups, hipStream_t stream) {
    CHECK_C(C, "maxpool2d_bwd_bn_reduce");
    if (!g_y || !idx || !z || !vec || !sums) return adamml_set_error(ADAMML_EINVAL, "maxpool2d_bwd_bn_reduce: null argument");
    const size_t P = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2);          // 2x2 input quads
    if (!P) return ADAMML_OK;
    if (groups < 1) groups = 1;
    size_t ppb, nblk;
    reduce_grid(P, C, groups, &ppb, &nblk);
    hipLaunchKernelGGL(maxpool_bwd_bn_kernel<false>, dim3((unsigned)nblk, groups), dim3(NT), 0, stream, (const bf16_t*)g_y, idx,
                       (const bf16_t*)z, vec, act, sums, (const float*)nullptr, (bf16_t*)nullptr, N, H, W, C, OH, OW, ppb);
    return adamml_check_launch("maxpool2d_bwd_bn_reduce");
}

extern "C" int adamml_maxpool2d_bwd_bn_apply(const void* g_y, const uint8_t* idx, const void* z, const float* vec, int act, const float* coef,
                                             void* dz, int N, int H, int W, int C, int OH, int OW, int groups, hipStream_t stream) {
    CHECK_C(C, "maxpool2d_bwd_bn_apply");
    if (!g_y || !idx || !z || !vec || !coef || !dz) return adamml_set_error(ADAMML_EINVAL, "maxpool2d_bwd_bn_apply: null argument");
    const size_t P = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2);          // 2x2 input quads
    if (!P) return ADAMML_OK;
    if (groups < 1) groups = 1;
    size_t ppb, nblk;
    rowwalk_grid(P, C, groups, 8192, &ppb, &nblk);
    if (H % 2 == 0 && W % 2 == 0)
        hipLaunchKernelGGL((maxpool_bwd_bn_kernel<true, true>), dim3((unsigned)nblk, groups), dim3(NT), 0, stream, (const bf16_t*)g_y, idx,
                           (const bf16_t*)z, vec, act, (double*)nullptr, coef, (bf16_t*)dz, N, H, W, C, OH, OW, ppb);
    else
        hipLaunchKernelGGL(maxpool_bwd_bn_kernel<true>, dim3((unsigned)nblk, groups), dim3(NT), 0, stream, (const bf16_t*)g_y, idx,
                           (const bf16_t*)z, vec, act, (double*)nullptr, coef, (bf16_t*)dz, N, H, W, C, OH, OW, ppb);
    return adamml_check_launch("maxpool2d_bwd_bn_apply");
}

extern "C" int adamml_temporal_pool_fwd(const void* x, const float* scale, const float* shift, int gstride, int act, void* y, int NB,
                                        int T, size_t HWC, int C, int mode, int groups, hipStream_t stream) {
    CHECK_C(C, "temporal_pool_fwd");
    if (mode == 1 && T < 3)   // models/common.py:20 nn.AvgPool3d raises for T < kernel (torch: "input image smaller than kernel size")
        return adamml_set_error(ADAMML_EINVAL, "temporal_pool_fwd: avg pooling needs T >= 3 (got %d), as in the reference", T);
    const int To = (T - 1) / 2 + 1;
    const size_t n = (size_t)NB * To * (HWC / 8);
    if (!n) return ADAMML_OK;
    if (groups < 1) groups = 1;
    if (T == 8 || T == 4 || T == 2) {
        const dim3 grid(grid_for((size_t)NB * (HWC / 8), NT, 8192 / groups + 1), groups);
#define LAUNCH_TP(TV) hipLaunchKernelGGL(temporal_pool_fwd_walk_kernel<TV>, grid, dim3(NT), 0, stream, (const bf16_t*)x, scale, shift, gstride, act, \
                                         (bf16_t*)y, NB, HWC / 8, C, mode)
        if (T == 8) LAUNCH_TP(8); else if (T == 4) LAUNCH_TP(4); else LAUNCH_TP(2);
#undef LAUNCH_TP
        return adamml_check_launch("temporal_pool_fwd");
    }
    hipLaunchKernelGGL(temporal_pool_fwd_kernel, dim3(grid_for(n, NT, 4096 / groups + 1), groups), dim3(NT), 0, stream, (const bf16_t*)x,
                       scale, shift, gstride, act, (bf16_t*)y, NB, T, To, HWC / 8, C, mode);
    return adamml_check_launch("temporal_pool_fwd");
}

extern "C" int adamml_temporal_pool_bwd(const void* g_y, const void* x, const float* scale, const float* shift, int gstride, int act,
                                        void* g_x, int NB, int T, size_t HWC, int C, int mode, int groups, hipStream_t stream) {
    CHECK_C(C, "temporal_pool_bwd");
    const int To = (T - 1) / 2 + 1;
    const size_t n = (size_t)NB * T * (HWC / 8);
    if (!n) return ADAMML_OK;
    if (groups < 1) groups = 1;
    hipLaunchKernelGGL(temporal_pool_bwd_kernel, dim3(grid_for(n, NT, 4096 / groups + 1), groups), dim3(NT), 0, stream, (const bf16_t*)g_y,
                       (const bf16_t*)x, scale, shift, gstride, act, (bf16_t*)g_x, NB, T, To, HWC / 8, C, mode);
    return adamml_check_launch("temporal_pool_bwd");
}

extern "C" int adamml_temporal_pool_bwd_res_supported(int T, int C, int mode) {
    return (T == 2 || T == 4 || T == 8) && mode == 0 && C % 8 == 0 && C <= MAXC ? 1 : 0;
}

extern "C" int adamml_temporal_pool_bwd_res(const void* g_y, const void* out, int act, void* g2, const void* z_a, const float* vec_a,
                                            double* sums_a, int NB, int T, int HW, int C, int groups, hipStream_t stream) {
    CHECK_C(C, "temporal_pool_bwd_res");
    if (!adamml_temporal_pool_bwd_res_supported(T, C, 0)) return adamml_set_error(ADAMML_EUNSUPPORTED, "temporal_pool_bwd_res: T=%d C=%d", T, C);
    if (!g_y || !out || !g2 || !sums_a || (z_a && !vec_a)) return adamml_set_error(ADAMML_EINVAL, "temporal_pool_bwd_res: null argument");
    const size_t cols = (size_t)NB * HW;
    if (!cols) return ADAMML_OK;
    if (groups < 1) groups = 1;
    size_t cpb, nblk;
    reduce_grid(cols, C, groups, &cpb, &nblk);
#define LAUNCH_TPR(TV)                                                                                                              \
    hipLaunchKernelGGL(temporal_pool_residual_bwd_kernel<TV>, dim3((unsigned)nblk, groups), dim3(NT), 0, stream, (const bf16_t*)g_y, \
                       (const bf16_t*)out, act, (bf16_t*)g2, (const bf16_t*)z_a, vec_a, sums_a, cols, HW, C, cpb)
    if (T == 8) LAUNCH_TPR(8); else if (T == 4) LAUNCH_TPR(4); else LAUNCH_TPR(2);
#undef LAUNCH_TPR
    return adamml_check_launch("temporal_pool_bwd_res");
}

extern "C" int adamml_temporal_pool_bwd_code(const void* g_y, const uint16_t* code, void* g2, double* sums_a, int NB, int T, int HW, int C,
                                             int groups, hipStream_t stream) {
    CHECK_C(C, "temporal_pool_bwd_code");
    if (!adamml_temporal_pool_bwd_res_supported(T, C, 0)) return adamml_set_error(ADAMML_EUNSUPPORTED, "temporal_pool_bwd_code: T=%d C=%d", T, C);
    if (!g_y || !code || !g2 || !sums_a) return adamml_set_error(ADAMML_EINVAL, "temporal_pool_bwd_code: null argument");
    const size_t cols = (size_t)NB * HW;
    if (!cols) return ADAMML_OK;
    if (groups < 1) groups = 1;
    size_t cpb, nblk;
    reduce_grid(cols, C, groups, &cpb, &nblk);
#define LAUNCH_TPC(TV)                                                                                                          \
    hipLaunchKernelGGL(temporal_pool_code_bwd_kernel<TV>, dim3((unsigned)nblk, groups), dim3(NT), 0, stream, (const bf16_t*)g_y, \
                       code, (bf16_t*)g2, sums_a, cols, HW, C, cpb)
    if (T == 8) LAUNCH_TPC(8); else if (T == 4) LAUNCH_TPC(4); else LAUNCH_TPC(2);
#undef LAUNCH_TPC
    return adamml_check_launch("temporal_pool_bwd_code");
}

extern "C" int adamml_gap_fwd(const void* x, const float* scale, const float* shift, int gstride, int act, float* out, int N, int HW,
                              int C, int groups, hipStream_t stream) {
    CHECK_C(C, "gap_fwd");
    const size_t n = (size_t)N * (C / 8);
    if (!n) return ADAMML_OK;
    if (groups < 1) groups = 1;
    hipLaunchKernelGGL(gap_fwd_kernel, dim3(grid_for(n, 64), groups), dim3(NT), 0, stream, (const bf16_t*)x, scale, shift, gstride, act, out,
                       N, HW, C);
    return adamml_check_launch("gap_fwd");
}

extern "C" int adamml_gap_bwd(const float* g, void* g_x, int N, int HW, int C, hipStream_t stream) {
    CHECK_C(C, "gap_bwd");
    const size_t n = (size_t)N * HW * (C / 8);
    if (!n) return ADAMML_OK;
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for(n)), dim3(NT), 0, stream, g, (bf16_t*)g_x, N, HW, C);
    return adamml_check_launch("gap_bwd");
}

extern "C" int adamml_head_fwd(const void* x, const float* scale, const float* shift, int gstride, int act, const uint8_t* keep_mask,
                               float inv_keep, const float* weight, const float* bias, float* feat, float* logits, int clips, int T, int HW,
                               int C, int K, int groups, hipStream_t stream) {
    CHECK_C(C, "head_fwd");
    if (!x || !weight || !feat || !logits) return adamml_set_error(ADAMML_EINVAL, "head_fwd: null argument");
    if (groups < 1) groups = 1;
    if (clips % groups || T < 1 || HW < 1 || K < 1) return adamml_set_error(ADAMML_EINVAL, "head_fwd: clips=%d groups=%d T=%d HW=%d K=%d", clips, groups, T, HW, K);
    if (!clips) return ADAMML_OK;
    hipLaunchKernelGGL(head_fwd_kernel, dim3(clips), dim3(NT), 0, stream, (const bf16_t*)x, scale, shift, gstride, act, keep_mask, inv_keep, weight,
                       bias, feat, logits, clips / groups, T, HW, C, K);
    return adamml_check_launch("head_fwd");
}

extern "C" int adamml_head_bwd(const float* g, const uint8_t* keep_mask, float inv_keep, const float* weight, void* g_x, float* g_rows, int clips,
                               int T, int HW, int C, int K, hipStream_t stream) {
    CHECK_C(C, "head_bwd");
    if (!g || !weight || !g_x) return adamml_set_error(ADAMML_EINVAL, "head_bwd: null argument");
    if (K < 1 || T < 1 || HW < 1) return adamml_set_error(ADAMML_EINVAL, "head_bwd: T=%d HW=%d K=%d", T, HW, K);
    if (!clips) return ADAMML_OK;
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)clips * T), dim3(NT), 0, stream, g, keep_mask, inv_keep, weight, (bf16_t*)g_x, g_rows, T, HW, C, K);
    return adamml_check_launch("head_bwd");
}

extern "C" int adamml_colsum_f32(const float* a, float* out, int rows, int cols, int accumulate, hipStream_t stream) {
    if (!a || !out || rows < 0 || cols < 1) return adamml_set_error(ADAMML_EINVAL, "colsum_f32: bad arguments");
    hipLaunchKernelGGL(colsum_f32_kernel, dim3(1), dim3(256), 0, stream, a, out, rows, cols, accumulate);
    return adamml_check_launch("colsum_f32");
}

extern "C" int adamml_clip_to_nhwc(const float* x, void* y, int B, int S, int F, int C, int H, int W, int OH, int OW,
                                   int frame_step, int c_pad, hipStream_t stream) {
    if ((c_pad % 8 && c_pad != 4) || c_pad < C || frame_step < 1) return adamml_set_error(ADAMML_EINVAL, "clip_to_nhwc: bad c_pad/frame_step");
    const int Fk = (F + frame_step - 1) / frame_step;
    const size_t n = (size_t)S * B * Fk * OH * OW;
    if (!n) return ADAMML_OK;
    if (c_pad == 4 && OH == H && OW == W && (W & 3) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0) {
        hipLaunchKernelGGL(clip_to_nhwc4_kernel, dim3(grid_for(n / 4)), dim3(NT), 0, stream, x, (bf16_t*)y, B, S, F, C, H, W, frame_step, Fk);
        return adamml_check_launch("clip_to_nhwc (4-pixel)");
    }
    hipLaunchKernelGGL(clip_to_nhwc_kernel, dim3(grid_for(n)), dim3(NT), 0, stream, x, (bf16_t*)y, B, S, F, C, H, W, OH, OW,
                       frame_step, Fk, c_pad);
    return adamml_check_launch("clip_to_nhwc");
}

extern "C" int adamml_clip_u8_to_nhwc(const uint8_t* x, void* y, int B, int S, int F, int C, int H, int W, int OH, int OW, int frame_step,
                                      int c_pad, const float* mean, const float* std, int n_mean, int div255, hipStream_t stream) {
    if (!x || !y || !mean || !std) return adamml_set_error(ADAMML_EINVAL, "clip_u8_to_nhwc: null argument");
    if ((c_pad % 8 && c_pad != 4) || c_pad < C || frame_step < 1) return adamml_set_error(ADAMML_EINVAL, "clip_u8_to_nhwc: bad c_pad/frame_step");
    if (n_mean < 1 || n_mean > 4 || C % n_mean) return adamml_set_error(ADAMML_EINVAL, "clip_u8_to_nhwc: %d mean/std values for %d channels", n_mean, C);
    const int Fk = (F + frame_step - 1) / frame_step;
    const size_t n = (size_t)B * OH * OW;
    if (!n || !S || !Fk) return ADAMML_OK;
    NormVec nv;
    nv.n = n_mean;
    for (int i = 0; i < 4; ++i) { nv.mean[i] = i < n_mean ? mean[i] : 0.f; nv.std[i] = i < n_mean ? std[i] : 1.f; }
    static const int fast = getenv("ADAMML_U8_FAST") ? atoi(getenv("ADAMML_U8_FAST")) : 1;            // A/B aid: 0 = the generic kernel
    if (fast && C == 3 && F == 8 && n_mean == 3 && S >= 1 && S <= 5 && ((uintptr_t)x & 7) == 0) {
        bool ok = false;
        switch (S) {
            case 1: ok = launch_u8_rgb<1>(x, (bf16_t*)y, B, H, W, OH, OW, frame_step, c_pad, nv, div255, n, stream); break;
            case 2: ok = launch_u8_rgb<2>(x, (bf16_t*)y, B, H, W, OH, OW, frame_step, c_pad, nv, div255, n, stream); break;
            case 3: ok = launch_u8_rgb<3>(x, (bf16_t*)y, B, H, W, OH, OW, frame_step, c_pad, nv, div255, n, stream); break;
            case 4: ok = launch_u8_rgb<4>(x, (bf16_t*)y, B, H, W, OH, OW, frame_step, c_pad, nv, div255, n, stream); break;
            case 5: ok = launch_u8_rgb<5>(x, (bf16_t*)y, B, H, W, OH, OW, frame_step, c_pad, nv, div255, n, stream); break;
        }
        if (ok) return adamml_check_launch("clip_u8_to_nhwc (rgb)");
    }
    hipLaunchKernelGGL(clip_u8_to_nhwc_kernel<false>, dim3(grid_for(n)), dim3(NT), 0, stream, x, (bf16_t*)y, B, S, F, C, H, W, OH, OW, frame_step,
                       Fk, c_pad, nv, div255);
    return adamml_check_launch("clip_u8_to_nhwc");
}

extern "C" int adamml_clip_u8_rgbdiff_to_nhwc(const uint8_t* x, void* y, int B, int S, int F, int D, int H, int W, int OH, int OW,
                                              int frame_step, int c_pad, const float* mean, const float* std, int n_mean,
                                              hipStream_t stream) {
    const int C = 3 * D;
    if (!x || !y || !mean || !std) return adamml_set_error(ADAMML_EINVAL, "clip_u8_rgbdiff_to_nhwc: null argument");
    if (D < 1 || c_pad % 8 || c_pad < C || frame_step < 1) return adamml_set_error(ADAMML_EINVAL, "clip_u8_rgbdiff_to_nhwc: bad D/c_pad/frame_step");
    if (n_mean < 1 || n_mean > 4 || C % n_mean) return adamml_set_error(ADAMML_EINVAL, "clip_u8_rgbdiff_to_nhwc: %d mean/std values for %d channels", n_mean, C);
    const int Fk = (F + frame_step - 1) / frame_step;
    const size_t n = (size_t)B * OH * OW;
    if (!n || !S || !Fk) return ADAMML_OK;
    NormVec nv;
    nv.n = n_mean;
    for (int i = 0; i < 4; ++i) { nv.mean[i] = i < n_mean ? mean[i] : 0.f; nv.std[i] = i < n_mean ? std[i] : 1.f; }
    hipLaunchKernelGGL(clip_u8_to_nhwc_kernel<true>, dim3(grid_for(n)), dim3(NT), 0, stream, x, (bf16_t*)y, B, S, F, C, H, W, OH, OW, frame_step,
                       Fk, c_pad, nv, 1);
    return adamml_check_launch("clip_u8_rgbdiff_to_nhwc");
}

extern "C" int adamml_pack_conv_weight(const float* w, void* out, int cout, int cin_true, int cin_pad, int kh, int kw, int mode,
                                       hipStream_t stream) {
    if (mode < 0 || mode > 2) return adamml_set_error(ADAMML_EINVAL, "pack_conv_weight: mode %d", mode);
    const size_t n = mode == 2 ? (size_t)kh * kw * cout : (size_t)cout * kh * kw * cin_pad;
    if (!n) return ADAMML_OK;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid_for(n)), dim3(NT), 0, stream, w, out, cout, cin_true, cin_pad, kh, kw, mode);
    return adamml_check_launch("pack_conv_weight");
}

extern "C" int adamml_pack_conv_weights_batched(const int64_t* table, int n, int64_t total_blocks, hipStream_t stream) {
    if (!table || n < 1 || total_blocks < 1) return adamml_set_error(ADAMML_EINVAL, "pack_conv_weights_batched: bad arguments");
    hipLaunchKernelGGL(pack_conv_weights_batched_kernel, dim3((unsigned)total_blocks), dim3(NT), 0, stream, (const long long*)table, n);
    return adamml_check_launch("pack_conv_weights_batched");
}

extern "C" int adamml_pack_block_elems(void) { return PACK_EPB; }

extern "C" int adamml_sgd_step(float* p, const float* g, float* mom, size_t n, float lr, float momentum, float weight_decay,
                               int nesterov, int first_step, hipStream_t stream) {
    if (!n) return ADAMML_OK;
    hipLaunchKernelGGL(sgd_step_kernel, dim3(grid_for(n)), dim3(NT), 0, stream, p, g, mom, n, lr, momentum, weight_decay, nesterov,
                       first_step);
    return adamml_check_launch("sgd_step");
}

extern "C" int adamml_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int step, hipStream_t stream) {
    if (!n) return ADAMML_OK;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid_for(n)), dim3(NT), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                       bc1, bc2);
    return adamml_check_launch("adam_step");
}
