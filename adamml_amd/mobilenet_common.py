"""Shared executor of the two MobileNetV2 variants (sound main net, policy net): inverted-residual blocks as
pointwise MFMA convs + depthwise VALU convs with lazily applied BatchNorm/ReLU6."""
from .runtime import conv_bn, conv_bn_add, conv_bn_add_supported, add_act, temporal_pool, ACT_NONE, ACT_RELU6


class BlockPlan:
    __slots__ = ("pw", "dw", "pwl", "residual", "tpool")

    def __init__(self, pw, dw, pwl, residual, tpool=None):
        self.pw, self.dw, self.pwl, self.residual, self.tpool = pw, dw, pwl, residual, tpool


def run_blocks(rt, h, plans):
    """plans: list of BlockPlan; pw/dw/pwl are (ConvState, BatchNorm2d) pairs (pw may be None)."""
    for bp in plans:
        x = h
        if bp.tpool:
            x = temporal_pool(rt, x, bp.tpool, "max", sole_consumer=True)
        y = x
        if bp.pw is not None:
            # the expansion conv is recorded before the block's own residual add, so it is reversed after it: last consumer
            # (without a residual add the block input feeds nothing else: its BatchNorm-backward sums come from this conv's data gradient)
            y = conv_bn(rt, y, bp.pw[0], bp.pw[1], ACT_RELU6, last_consumer=True, sole_consumer=not bp.residual)
        # the expansion output feeds only the depthwise conv (without an expansion the block input may also feed the residual add)
        y = conv_bn(rt, y, bp.dw[0], bp.dw[1], ACT_RELU6, sole_consumer=bp.pw is not None or not bp.residual)
        if bp.residual and conv_bn_add_supported(rt, y, bp.pwl[0], rt.tape.need_grad, x):
            h = conv_bn_add(rt, y, bp.pwl[0], bp.pwl[1], x, ACT_NONE)               # inference: projection + BatchNorm + add in one kernel
            continue
        y = conv_bn(rt, y, bp.pwl[0], bp.pwl[1], ACT_NONE, sole_consumer=True)      # the dw output feeds only this conv
        h = add_act(rt, y, x, ACT_NONE) if bp.residual else y
    return h
