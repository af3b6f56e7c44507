// Policy causality head of AdaMML on gfx950: LSTMCell recurrence over the segments with the previous logits fed back,
// per-modality FC heads, hard Gumbel-softmax gate (models/policy_net.py:283-290,341-370), and the decision-gated late
// fusion of the main nets' logits (models/joint_resnet_mobilenetv2.py:94,112-127 + models/adamml.py:86-88).
//
// The recurrence is independent per video, so one workgroup owns one video for all S segments: no cross-workgroup
// synchronisation, no per-segment launches.  The feature half of the gate GEMM (W_ih[:, :F] . feat) does not depend on
// the recurrence and is computed for all segments at once by adamml_gemm_f32; this kernel adds the recurrent half
// (W_hh . h and W_ih[:, F:] . previous logits), the cell, the heads and the gate.  Matrix rows are read with the 64
// lanes across K (one coalesced 1 KB row per wave load) and folded with a butterfly reduction.
#include "common.h"
#include "../../include/adamml_hip.h"

#define HID 256            // nn.LSTMCell hidden size (policy_net.py:276)
#define MAXM 4             // modalities with a policy decision (rgb, sound, rgbdiff/flow: <= 3 in the reference)

struct HeadPtrs {
    const float* w[MAXM];  // fcs[m].weight [2, HID]
    const float* b[MAXM];  // fcs[m].bias   [2]
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// F.gumbel_softmax(logits, tau, hard=True)[:, -1] for one 2-way row (torch/nn/functional.py gumbel_softmax):
//   gumbels = -log(Exponential(1));  y_soft = softmax((logits + gumbels) / tau);  y_hard = onehot(argmax y_soft);
//   value = y_hard - y_soft.detach() + y_soft   (1 or 0 up to one rounding; the gradient flows through y_soft)
__device__ __forceinline__ float gumbel_gate(float l0, float l1, float e0, float e1, float tau, float* ys0, float* ys1) {
    const float a0 = (l0 - logf(e0)) / tau, a1 = (l1 - logf(e1)) / tau;
    const float mx = fmaxf(a0, a1);
    const float x0 = expf(a0 - mx), x1 = expf(a1 - mx);
    const float s = x0 + x1;
    const float y0 = x0 / s, y1 = x1 / s;
    *ys0 = y0;
    *ys1 = y1;
    const float hard1 = (y1 > y0) ? 1.f : 0.f;       // ties -> index 0 (first maximum)
    return (hard1 - y1) + y1;
}

// ---------------------------------------------------------------------------------------------------------------------
// grid = B workgroups x 256 threads.  gates_x [S,B,4*HID] = W_ih[:, :F] feat + b_ih (+ b_hh added here).
// saves: h_all [S+1,B,HID] (slot 0 = zeros = initial state), c_all [S+1,B,HID], gact [S,B,4*HID] (post-nonlinearity
// i,f,g,o), prev_all [S,B,2M] (logits fed into step s), ysoft [S,M,B,2].
__global__ void __launch_bounds__(256) policy_head_fwd_kernel(
    const float* __restrict__ gates_x, const float* __restrict__ w_prev, int ld_ih, const float* __restrict__ w_hh,
    const float* __restrict__ b_hh, HeadPtrs fc, const float* __restrict__ expo, float tau, float* __restrict__ decisions,
    float* __restrict__ logits, float* __restrict__ h_all, float* __restrict__ c_all, float* __restrict__ gact,
    float* __restrict__ prev_all, float* __restrict__ ysoft, int S, int B, int M) {
    __shared__ float sh_h[HID];
    __shared__ float sh_prev[2 * MAXM];
    __shared__ float sh_logit[2 * MAXM];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    sh_h[t] = 0.f;
    if (t < 2 * MAXM) sh_prev[t] = 0.f;
    float c = 0.f;
    h_all[(size_t)b * HID + t] = 0.f;
    c_all[(size_t)b * HID + t] = 0.f;
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        // ---- recurrent half of the gates: wave w owns hidden units [64w, 64w+64); lane j keeps unit 64w+j
        const f32x4 hreg = *reinterpret_cast<const f32x4*>(&sh_h[lane * 4]);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            for (int j = 0; j < 64; ++j) {
                const int u = wave * 64 + j;
                float p[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(w_hh + (size_t)(g * HID + u) * HID + lane * 4);
                    p[g] = w[0] * hreg[0] + w[1] * hreg[1] + w[2] * hreg[2] + w[3] * hreg[3];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float r = wave_sum(p[g]);
                    if (lane == j) acc[g] = r;
                }
            }
        }
        float gt[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = g * HID + t;
            float v = gates_x[((size_t)s * B + b) * (4 * HID) + row] + b_hh[row] + acc[g];
            if (s > 0)
                for (int q = 0; q < 2 * M; ++q) v += w_prev[(size_t)row * ld_ih + q] * sh_prev[q];
            gt[g] = v;
        }
        const float ig = sigmoidf_(gt[0]), fg = sigmoidf_(gt[1]), gg = tanhf(gt[2]), og = sigmoidf_(gt[3]);
        c = fg * c + ig * gg;
        const float h = og * tanhf(c);
        float* ga = gact + ((size_t)s * B + b) * (4 * HID);
        ga[t] = ig; ga[HID + t] = fg; ga[2 * HID + t] = gg; ga[3 * HID + t] = og;
        h_all[((size_t)(s + 1) * B + b) * HID + t] = h;
        c_all[((size_t)(s + 1) * B + b) * HID + t] = c;
        if (t < 2 * M) prev_all[((size_t)s * B + b) * (2 * M) + t] = sh_prev[t];
        __syncthreads();                       // everyone has read sh_h / sh_prev of the previous step
        sh_h[t] = h;
        __syncthreads();
        // ---- heads: 2M rows of HID, one row per wave pass
        const f32x4 hn = *reinterpret_cast<const f32x4*>(&sh_h[lane * 4]);
        for (int r = wave; r < 2 * M; r += 4) {
            const int m = r >> 1, j = r & 1;
            const f32x4 w = *reinterpret_cast<const f32x4*>(fc.w[m] + j * HID + lane * 4);
            const float v = wave_sum(w[0] * hn[0] + w[1] * hn[1] + w[2] * hn[2] + w[3] * hn[3]);
            if (lane == 0) sh_logit[r] = v + fc.b[m][j];
        }
        __syncthreads();
        if (t < M) {
            const float l0 = sh_logit[2 * t], l1 = sh_logit[2 * t + 1];
            const size_t row = ((size_t)s * M + t) * B + b;           // [S][M][B]
            float y0, y1;
            const float dsn = gumbel_gate(l0, l1, expo[row * 2], expo[row * 2 + 1], tau, &y0, &y1);
            decisions[row] = dsn;
            logits[row * 2] = l0; logits[row * 2 + 1] = l1;
            ysoft[row * 2] = y0; ysoft[row * 2 + 1] = y1;
        }
        if (t < 2 * M) sh_prev[t] = sh_logit[t];      // [m0 j0, m0 j1, m1 j0, ...] == the reference's permute(1,0,2) row
        __syncthreads();
    }
}

// Reverse recurrence.  d_dec [S,M,B] (gradient of the decisions), d_logits_in [S,M,B,2] or NULL (direct gradient of the
// returned logits).  Outputs: d_gates [S,B,4*HID] (pre-activation gate gradients: the weight / feature gradients are
// GEMMs over it, issued by the caller), d_logits [S,M,B,2] (total gradient of each step's logits, for the FC heads).
__global__ void __launch_bounds__(256) policy_head_bwd_kernel(
    const float* __restrict__ d_dec, const float* __restrict__ d_logits_in, const float* __restrict__ w_prev, int ld_ih,
    const float* __restrict__ w_hh, HeadPtrs fc, float tau, const float* __restrict__ c_all, const float* __restrict__ gact,
    const float* __restrict__ ysoft, float* __restrict__ d_gates, float* __restrict__ d_logits, int S, int B, int M) {
    __shared__ float sh_dg[4 * HID];
    __shared__ float sh_dl[2 * MAXM];
    __shared__ float sh_dprev[2 * MAXM];
    __shared__ float sh_red[4][2 * MAXM];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float dh_next = 0.f, dc_next = 0.f;
    if (t < 2 * MAXM) sh_dprev[t] = 0.f;
    __syncthreads();
    for (int s = S - 1; s >= 0; --s) {
        if (t < M) {
            const size_t row = ((size_t)s * M + t) * B + b;
            const float y0 = ysoft[row * 2], y1 = ysoft[row * 2 + 1];
            const float dy1 = d_dec[row];                        // decision = y_soft[1] + const
            const float dot = dy1 * y1;                          // sum_j dy_j y_j with dy_0 = 0
            float g0 = y0 * (0.f - dot) / tau, g1 = y1 * (dy1 - dot) / tau;
            if (d_logits_in) { g0 += d_logits_in[row * 2]; g1 += d_logits_in[row * 2 + 1]; }
            g0 += sh_dprev[2 * t]; g1 += sh_dprev[2 * t + 1];    // fed back into step s+1 as `prev`
            sh_dl[2 * t] = g0; sh_dl[2 * t + 1] = g1;
            d_logits[row * 2] = g0; d_logits[row * 2 + 1] = g1;
        }
        __syncthreads();
        float dh = dh_next;
        for (int m = 0; m < M; ++m) dh += sh_dl[2 * m] * fc.w[m][t] + sh_dl[2 * m + 1] * fc.w[m][HID + t];
        const float* ga = gact + ((size_t)s * B + b) * (4 * HID);
        const float ig = ga[t], fg = ga[HID + t], gg = ga[2 * HID + t], og = ga[3 * HID + t];
        const float cs = c_all[((size_t)(s + 1) * B + b) * HID + t], cp = c_all[((size_t)s * B + b) * HID + t];
        const float tc = tanhf(cs);
        const float dc = dh * og * (1.f - tc * tc) + dc_next;
        const float dgi = dc * gg * ig * (1.f - ig), dgf = dc * cp * fg * (1.f - fg), dgg = dc * ig * (1.f - gg * gg),
                    dgo = dh * tc * og * (1.f - og);
        dc_next = dc * fg;
        float* dgp = d_gates + ((size_t)s * B + b) * (4 * HID);
        dgp[t] = dgi; dgp[HID + t] = dgf; dgp[2 * HID + t] = dgg; dgp[3 * HID + t] = dgo;
        sh_dg[t] = dgi; sh_dg[HID + t] = dgf; sh_dg[2 * HID + t] = dgg; sh_dg[3 * HID + t] = dgo;
        __syncthreads();
        if (s > 0) {
            // dh_prev[k] = sum_r dgates[r] W_hh[r,k]  (thread k; rows coalesced across the workgroup)
            float a0 = 0.f, a1 = 0.f;
            for (int r = 0; r < 4 * HID; r += 2) {
                a0 += sh_dg[r] * w_hh[(size_t)r * HID + t];
                a1 += sh_dg[r + 1] * w_hh[(size_t)(r + 1) * HID + t];
            }
            dh_next = a0 + a1;
            // dprev[q] = sum_r dgates[r] W_ih[r, F+q]
            float pq[2 * MAXM];
#pragma unroll
            for (int q = 0; q < 2 * MAXM; ++q) pq[q] = 0.f;
            for (int g = 0; g < 4; ++g) {
                const int row = g * HID + t;
                const float d = sh_dg[row];
#pragma unroll
                for (int q = 0; q < 2 * MAXM; ++q)
                    if (q < 2 * M) pq[q] += d * w_prev[(size_t)row * ld_ih + q];
            }
#pragma unroll
            for (int q = 0; q < 2 * MAXM; ++q) {
                const float r = wave_sum(pq[q]);
                if (lane == 0) sh_red[wave][q] = r;
            }
            __syncthreads();
            if (t < 2 * M) sh_dprev[t] = sh_red[0][t] + sh_red[1][t] + sh_red[2][t] + sh_red[3][t];
        }
        __syncthreads();
    }
}

// Stand-alone gate for the head without causality modelling (policy_net.py:330-340): rows of 2 logits.
__global__ void gumbel_gate_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ expo, float tau,
                                       float* __restrict__ decisions, float* __restrict__ ysoft, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float y0, y1;
    decisions[r] = gumbel_gate(logits[2 * r], logits[2 * r + 1], expo[2 * r], expo[2 * r + 1], tau, &y0, &y1);
    ysoft[2 * r] = y0; ysoft[2 * r + 1] = y1;
}

__global__ void gumbel_gate_bwd_kernel(const float* __restrict__ d_dec, const float* __restrict__ ysoft, float tau,
                                       float* __restrict__ d_logits, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float y0 = ysoft[2 * r], y1 = ysoft[2 * r + 1], dy1 = d_dec[r], dot = dy1 * y1;
    d_logits[2 * r] = y0 * (0.f - dot) / tau;
    d_logits[2 * r + 1] = y1 * (dy1 - dot) / tau;
}

// ---------------------------------------------------------------------------------------------------------------------
// Late fusion over modalities and mean over segments: out[b,c] = (1/S) sum_s sum_m w_m dec[s,m,b] logit_m[s,b,c],
// w = cat(lf_weights, 1 - sum lf_weights) (learnable) or 1/M.
struct FusePtrs {
    const float* x[MAXM];   // per-modality logits [S*B, C]
    float* dx[MAXM];
};

__device__ __forceinline__ float fuse_weight(const float* lf, int m, int M) {
    if (!lf) return 1.f / (float)M;
    if (m < M - 1) return lf[m];
    float s = 0.f;
    for (int i = 0; i < M - 1; ++i) s += lf[i];
    return 1.f - s;
}

__global__ void fusion_fwd_kernel(FusePtrs p, const float* __restrict__ dec, const float* __restrict__ lf,
                                  float* __restrict__ out, int S, int B, int C, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) {
        float v = 0.f;
        for (int m = 0; m < M; ++m) {
            const float d = dec ? dec[((size_t)s * M + m) * B + b] : 1.f;
            v += fuse_weight(lf, m, M) * (p.x[m][((size_t)s * B + b) * C + c] * d);
        }
        acc += v;
    }
    out[i] = acc / (float)S;
}

// one workgroup (64 lanes) per (s, b): d_x[m][s,b,:], d_dec[s,m,b], per-row partial of d_lf -> d_lf_part[s*B+b][M]
__global__ void __launch_bounds__(64) fusion_bwd_kernel(FusePtrs p, const float* __restrict__ dec, const float* __restrict__ lf,
                                                        const float* __restrict__ g, float* __restrict__ d_dec,
                                                        float* __restrict__ d_lf_part, int S, int B, int C, int M) {
    const int sb = blockIdx.x, s = sb / B, b = sb - s * B, lane = threadIdx.x;
    const float inv_s = 1.f / (float)S;
    for (int m = 0; m < M; ++m) {
        const float d = dec ? dec[((size_t)s * M + m) * B + b] : 1.f;
        const float w = fuse_weight(lf, m, M);
        float dot = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float gv = g[(size_t)b * C + c] * inv_s;
            const float xv = p.x[m][(size_t)sb * C + c];
            if (p.dx[m]) p.dx[m][(size_t)sb * C + c] = gv * w * d;
            dot += gv * xv;
        }
        dot = wave_sum(dot);
        if (lane == 0) {
            if (d_dec) d_dec[((size_t)s * M + m) * B + b] = w * dot;
            if (d_lf_part) d_lf_part[(size_t)sb * M + m] = d * dot;      // d out / d w_m  (row contribution)
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
static int fill_heads(HeadPtrs* hp, const float* const* fc_w, const float* const* fc_b, int M, const char* who) {
    if (M < 1 || M > MAXM) return adamml_set_error(ADAMML_EUNSUPPORTED, "%s: %d modalities (1..%d supported)", who, M, MAXM);
    for (int m = 0; m < MAXM; ++m) {
        hp->w[m] = m < M ? fc_w[m] : nullptr;
        hp->b[m] = m < M ? fc_b[m] : nullptr;
        if (m < M && (!hp->w[m] || !hp->b[m])) return adamml_set_error(ADAMML_EINVAL, "%s: null FC head %d", who, m);
    }
    return ADAMML_OK;
}

extern "C" int adamml_policy_head_fwd(const float* gates_x, const float* w_prev, int ld_ih, const float* w_hh,
                                      const float* b_hh, const float* const* fc_w, const float* const* fc_b,
                                      const float* expo, float tau, float* decisions, float* logits, float* h_all,
                                      float* c_all, float* gate_act, float* prev_all, float* ysoft, int S, int B, int M,
                                      int hidden, hipStream_t stream) {
    if (hidden != HID) return adamml_set_error(ADAMML_EUNSUPPORTED, "adamml_policy_head_fwd: hidden size %d (256 supported)", hidden);
    if (!gates_x || !w_prev || !w_hh || !b_hh || !expo || !decisions || !logits || !h_all || !c_all || !gate_act || !prev_all || !ysoft)
        return adamml_set_error(ADAMML_EINVAL, "adamml_policy_head_fwd: null pointer");
    if (S < 1 || B < 1 || !(tau > 0.f)) return adamml_set_error(ADAMML_EINVAL, "adamml_policy_head_fwd: S=%d B=%d tau=%g", S, B, (double)tau);
    HeadPtrs hp;
    int rc = fill_heads(&hp, fc_w, fc_b, M, "adamml_policy_head_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(policy_head_fwd_kernel, dim3(B), dim3(256), 0, stream, gates_x, w_prev, ld_ih, w_hh, b_hh, hp, expo,
                       tau, decisions, logits, h_all, c_all, gate_act, prev_all, ysoft, S, B, M);
    return adamml_check_launch("adamml_policy_head_fwd");
}

extern "C" int adamml_policy_head_bwd(const float* d_decisions, const float* d_logits_in, const float* w_prev, int ld_ih,
                                      const float* w_hh, const float* const* fc_w, float tau, const float* c_all,
                                      const float* gate_act, const float* ysoft, float* d_gates, float* d_logits, int S,
                                      int B, int M, int hidden, hipStream_t stream) {
    if (hidden != HID) return adamml_set_error(ADAMML_EUNSUPPORTED, "adamml_policy_head_bwd: hidden size %d (256 supported)", hidden);
    if (!d_decisions || !w_prev || !w_hh || !c_all || !gate_act || !ysoft || !d_gates || !d_logits)
        return adamml_set_error(ADAMML_EINVAL, "adamml_policy_head_bwd: null pointer");
    if (S < 1 || B < 1 || !(tau > 0.f)) return adamml_set_error(ADAMML_EINVAL, "adamml_policy_head_bwd: S=%d B=%d tau=%g", S, B, (double)tau);
    HeadPtrs hp;
    const float* dummy[MAXM] = {fc_w ? fc_w[0] : nullptr, nullptr, nullptr, nullptr};
    for (int m = 0; m < MAXM && m < M; ++m) dummy[m] = fc_w[m];
    int rc = fill_heads(&hp, fc_w, dummy, M, "adamml_policy_head_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(policy_head_bwd_kernel, dim3(B), dim3(256), 0, stream, d_decisions, d_logits_in, w_prev, ld_ih, w_hh,
                       hp, tau, c_all, gate_act, ysoft, d_gates, d_logits, S, B, M);
    return adamml_check_launch("adamml_policy_head_bwd");
}

extern "C" int adamml_gumbel_gate_fwd(const float* logits, const float* expo, float tau, float* decisions, float* ysoft,
                                      int rows, hipStream_t stream) {
    if (!logits || !expo || !decisions || !ysoft) return adamml_set_error(ADAMML_EINVAL, "adamml_gumbel_gate_fwd: null pointer");
    if (rows < 1 || !(tau > 0.f)) return adamml_set_error(ADAMML_EINVAL, "adamml_gumbel_gate_fwd: rows=%d tau=%g", rows, (double)tau);
    hipLaunchKernelGGL(gumbel_gate_fwd_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, stream, logits, expo, tau, decisions, ysoft, rows);
    return adamml_check_launch("adamml_gumbel_gate_fwd");
}

extern "C" int adamml_gumbel_gate_bwd(const float* d_decisions, const float* ysoft, float tau, float* d_logits, int rows,
                                      hipStream_t stream) {
    if (!d_decisions || !ysoft || !d_logits) return adamml_set_error(ADAMML_EINVAL, "adamml_gumbel_gate_bwd: null pointer");
    if (rows < 1 || !(tau > 0.f)) return adamml_set_error(ADAMML_EINVAL, "adamml_gumbel_gate_bwd: rows=%d tau=%g", rows, (double)tau);
    hipLaunchKernelGGL(gumbel_gate_bwd_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, stream, d_decisions, ysoft, tau, d_logits, rows);
    return adamml_check_launch("adamml_gumbel_gate_bwd");
}

extern "C" int adamml_fusion_fwd(const float* const* x, const float* decisions, const float* lf_weights, float* out, int S,
                                 int B, int C, int M, hipStream_t stream) {
    if (M < 1 || M > MAXM) return adamml_set_error(ADAMML_EUNSUPPORTED, "adamml_fusion_fwd: %d modalities (1..%d supported)", M, MAXM);
    if (!x || !out || S < 1 || B < 1 || C < 1) return adamml_set_error(ADAMML_EINVAL, "adamml_fusion_fwd: bad arguments");
    FusePtrs p;
    for (int m = 0; m < MAXM; ++m) {
        p.x[m] = m < M ? x[m] : nullptr;
        p.dx[m] = nullptr;
        if (m < M && !p.x[m]) return adamml_set_error(ADAMML_EINVAL, "adamml_fusion_fwd: null logits %d", m);
    }
    hipLaunchKernelGGL(fusion_fwd_kernel, dim3(ceil_div(B * C, 256)), dim3(256), 0, stream, p, decisions, lf_weights, out, S, B, C, M);
    return adamml_check_launch("adamml_fusion_fwd");
}

extern "C" int adamml_fusion_bwd(const float* const* x, const float* decisions, const float* lf_weights, const float* g_out,
                                 float* const* d_x, float* d_decisions, float* d_lf_part, int S, int B, int C, int M,
                                 hipStream_t stream) {
    if (M < 1 || M > MAXM) return adamml_set_error(ADAMML_EUNSUPPORTED, "adamml_fusion_bwd: %d modalities (1..%d supported)", M, MAXM);
    if (!x || !g_out || S < 1 || B < 1 || C < 1) return adamml_set_error(ADAMML_EINVAL, "adamml_fusion_bwd: bad arguments");
    FusePtrs p;
    for (int m = 0; m < MAXM; ++m) {
        p.x[m] = m < M ? x[m] : nullptr;
        p.dx[m] = (m < M && d_x) ? d_x[m] : nullptr;
        if (m < M && !p.x[m]) return adamml_set_error(ADAMML_EINVAL, "adamml_fusion_bwd: null logits %d", m);
    }
    hipLaunchKernelGGL(fusion_bwd_kernel, dim3(S * B), dim3(64), 0, stream, p, decisions, lf_weights, g_out, d_decisions,
                       d_lf_part, S, B, C, M);
    return adamml_check_launch("adamml_fusion_bwd");
}
