"""fp32 autograd ops of the policy head backed by libadamml_hip (adamml_gemm_f32)."""
import torch

from . import hip
from .hip import ptr
from .runtime import gemm_f32, ACT_NONE, ACT_RELU


class _HipLinear(torch.autograd.Function):
    """y = act(x @ W^T + b) -- nn.Linear (+ReLU) at models/policy_net.py:228-231,279 and the LSTMCell gate GEMMs (:278)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = x.contiguous()
        y = gemm_f32(x, weight, bias=bias, act=act)
        ctx.save_for_backward(x, weight, y)
        ctx.act = act
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gy = gy.contiguous()
        if ctx.act == ACT_RELU:
            gy = gy * (y > 0).to(gy.dtype)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gemm_f32(gy, weight, trans_b=False)                    # [M,N] @ [N,K]
        if ctx.needs_input_grad[1]:
            gw = gemm_f32(gy, x, trans_a=True, trans_b=False)           # [N,M] @ [M,K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(0)
        return gx, gw, gb, None


def hip_linear(x, weight, bias=None, act=ACT_NONE):
    return _HipLinear.apply(x, weight, bias, act)


class _PolicyHead(torch.autograd.Function):
    """LSTMCell recurrence + FC heads + hard Gumbel-softmax gate of PolicyNet.forward (models/policy_net.py:341-370) as
    one launch per direction (adamml_policy_head_fwd / _bwd); the batched GEMMs around it run on adamml_gemm_f32.
    inputs: feats [S,B,F], W_ih [4H, F+2M], W_hh [4H,H], b_ih, b_hh, expo [S,M,B,2], tau, M, then fc weights / biases."""

    @staticmethod
    def forward(ctx, feats, w_ih, w_hh, b_ih, b_hh, expo, tau, M, *fc):
        S, B, F = feats.shape
        H = w_hh.shape[1]
        fc_w, fc_b = fc[:M], fc[M:]
        feats = feats.contiguous()
        dev = feats.device
        gates_x = torch.empty(S * B, 4 * H, dtype=torch.float32, device=dev)
        gemm_f32(feats.view(S * B, F), w_ih[:, :F], out=gates_x, bias=b_ih)
        dec = torch.empty(S, M, B, dtype=torch.float32, device=dev)
        logits = torch.empty(S, M, B, 2, dtype=torch.float32, device=dev)
        h_all = torch.empty(S + 1, B, H, dtype=torch.float32, device=dev)
        c_all = torch.empty(S + 1, B, H, dtype=torch.float32, device=dev)
        gact = torch.empty(S, B, 4 * H, dtype=torch.float32, device=dev)
        prev = torch.empty(S, B, 2 * M, dtype=torch.float32, device=dev)
        ysoft = torch.empty(S, M, B, 2, dtype=torch.float32, device=dev)
        expo = expo.contiguous()
        hip.call("adamml_policy_head_fwd", ptr(gates_x), w_ih.data_ptr() + 4 * F, w_ih.stride(0), ptr(w_hh), ptr(b_hh),
                 hip.ptr_array(fc_w), hip.ptr_array(fc_b), ptr(expo), float(tau), ptr(dec), ptr(logits), ptr(h_all), ptr(c_all),
                 ptr(gact), ptr(prev), ptr(ysoft), S, B, M, H)
        ctx.save_for_backward(feats, w_ih, w_hh, h_all, c_all, gact, prev, ysoft, *fc_w)
        ctx.tau, ctx.M = float(tau), M
        return dec, logits

    @staticmethod
    def backward(ctx, d_dec, d_logits_in):
        feats, w_ih, w_hh, h_all, c_all, gact, prev, ysoft = ctx.saved_tensors[:8]
        fc_w = ctx.saved_tensors[8:]
        M = ctx.M
        S, B, F = feats.shape
        H = w_hh.shape[1]
        dev = feats.device
        d_dec = d_dec.contiguous()
        d_gates = torch.empty(S * B, 4 * H, dtype=torch.float32, device=dev)
        d_logits = torch.empty(S, M, B, 2, dtype=torch.float32, device=dev)
        hip.call("adamml_policy_head_bwd", ptr(d_dec), ptr(d_logits_in.contiguous()) if d_logits_in is not None else None,
                 w_ih.data_ptr() + 4 * F, w_ih.stride(0), ptr(w_hh), hip.ptr_array(fc_w), ctx.tau, ptr(c_all), ptr(gact),
                 ptr(ysoft), ptr(d_gates), ptr(d_logits), S, B, M, H)
        need = ctx.needs_input_grad
        g_feats = g_wih = g_whh = g_bih = g_bhh = None
        if need[0]:
            g_feats = gemm_f32(d_gates, w_ih[:, :F], trans_b=False).view(S, B, F)           # [SB,4H] @ [4H,F]
        if need[1]:
            g_wih = torch.empty_like(w_ih)
            gemm_f32(d_gates, feats.view(S * B, F), out=g_wih[:, :F], trans_a=True, trans_b=False)
            gemm_f32(d_gates, prev.view(S * B, 2 * M), out=g_wih[:, F:], trans_a=True, trans_b=False)
        if need[2]:
            g_whh = gemm_f32(d_gates, h_all[:S].reshape(S * B, H), trans_a=True, trans_b=False)
        if need[3] or need[4]:
            g_bih = d_gates.sum(0)
            g_bhh = g_bih
        g_fc_w, g_fc_b = [], []
        hs = h_all[1:].reshape(S * B, H)
        for m in range(M):
            dl = d_logits[:, m].reshape(S * B, 2)
            g_fc_w.append(gemm_f32(dl, hs, trans_a=True, trans_b=False) if need[8 + m] else None)
            g_fc_b.append(dl.sum(0) if need[8 + M + m] else None)
        return (g_feats, g_wih, g_whh, g_bih, g_bhh, None, None, None, *g_fc_w, *g_fc_b)


def policy_head(feats, lstm, fcs, tau, expo):
    """feats [S,B,F] -> (decisions [S,M,B], logits [S,M,B,2]); expo [S,M,B,2] Exponential(1) noise."""
    M = len(fcs)
    return _PolicyHead.apply(feats, lstm.weight_ih, lstm.weight_hh, lstm.bias_ih, lstm.bias_hh, expo, tau, M,
                             *[fc.weight for fc in fcs], *[fc.bias for fc in fcs])


class _GumbelGate(torch.autograd.Function):
    """F.gumbel_softmax(logits, tau, hard=True)[:, -1] on rows of 2 logits (models/policy_net.py:283-290)."""

    @staticmethod
    def forward(ctx, logits, expo, tau):
        logits = logits.contiguous()
        R = logits.numel() // 2
        dec = torch.empty(R, dtype=torch.float32, device=logits.device)
        ysoft = torch.empty(R, 2, dtype=torch.float32, device=logits.device)
        hip.call("adamml_gumbel_gate_fwd", ptr(logits), ptr(expo.contiguous()), float(tau), ptr(dec), ptr(ysoft), R)
        ctx.save_for_backward(ysoft)
        ctx.tau = float(tau)
        return dec

    @staticmethod
    def backward(ctx, d_dec):
        ysoft, = ctx.saved_tensors
        R = ysoft.shape[0]
        d_logits = torch.empty(R, 2, dtype=torch.float32, device=ysoft.device)
        hip.call("adamml_gumbel_gate_bwd", ptr(d_dec.contiguous()), ptr(ysoft), ctx.tau, ptr(d_logits), R)
        return d_logits, None, None


def gumbel_gate(logits, expo, tau):
    return _GumbelGate.apply(logits, expo, tau)


class _Fusion(torch.autograd.Function):
    """Decision-gated late fusion over modalities + mean over segments (models/joint_resnet_mobilenetv2.py:94,112-127,
    models/adamml.py:86-88) as one launch per direction.  xs: M tensors [S*B, C]; decisions [S,M,B] or None;
    lf_weights [M-1] or None."""

    @staticmethod
    def forward(ctx, decisions, lf_weights, S, *xs):
        M = len(xs)
        xs = [x.contiguous() for x in xs]
        SB, C = xs[0].shape
        B = SB // S
        out = torch.empty(B, C, dtype=torch.float32, device=xs[0].device)
        dec = decisions.contiguous() if decisions is not None else None
        hip.call("adamml_fusion_fwd", hip.ptr_array(xs), ptr(dec), ptr(lf_weights), ptr(out), S, B, C, M)
        ctx.save_for_backward(dec, lf_weights, *xs)
        ctx.S = S
        return out

    @staticmethod
    def backward(ctx, g):
        dec, lf = ctx.saved_tensors[:2]
        xs = ctx.saved_tensors[2:]
        M, S = len(xs), ctx.S
        SB, C = xs[0].shape
        B = SB // S
        need = ctx.needs_input_grad
        dev = g.device
        d_xs = [torch.empty_like(x) if need[3 + m] else None for m, x in enumerate(xs)]
        d_dec = torch.empty(S, M, B, dtype=torch.float32, device=dev) if (dec is not None and need[0]) else None
        d_lfp = torch.empty(SB, M, dtype=torch.float32, device=dev) if (lf is not None and need[1]) else None
        hip.call("adamml_fusion_bwd", hip.ptr_array(xs), ptr(dec), ptr(lf), ptr(g.contiguous()), hip.ptr_array(d_xs), ptr(d_dec),
                 ptr(d_lfp), S, B, C, M)
        d_lf = None
        if d_lfp is not None:
            dw = d_lfp.sum(0)                       # d loss / d w_m, w = cat(lf, 1 - sum(lf))
            d_lf = dw[:M - 1] - dw[M - 1]
        return (d_dec, d_lf, None, *d_xs)


def fuse_segments(xs, decisions, lf_weights, S):
    """xs: list of M [S*B, C] logits -> [B, C] (mean over segments of the gated, weighted sum over modalities)."""
    return _Fusion.apply(decisions, lf_weights, S, *xs)
