"""fp32 autograd ops of the policy head backed by libadamml_hip (adamml_gemm_f32)."""
import torch

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
