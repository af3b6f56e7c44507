"""Fused flat optimizers: one launch updates a whole sub-network's flat fp32 parameter
buffer.  Semantics of torch.optim.SGD(momentum, weight_decay) / torch.optim.Adam(weight_decay) as used at
train_adamml.py:251-257."""
import torch

from .hip import call, ptr


class FlatSGD:
    def __init__(self, flat_buffers, lr, momentum=0.9, weight_decay=0.0, nesterov=False):
        self.fb, self.lr, self.momentum, self.weight_decay, self.nesterov = flat_buffers, lr, momentum, weight_decay, nesterov
        self.mom = None
        self.steps = 0

    def step(self):
        fb = self.fb
        if fb.flat_grad is None:
            return
        if self.mom is None or self.mom.numel() != fb.flat.numel():
            self.mom = torch.zeros_like(fb.flat)
            self.steps = 0
        call("adamml_sgd_step", ptr(fb.flat), ptr(fb.flat_grad), ptr(self.mom), fb.flat.numel(), self.lr, self.momentum,
             self.weight_decay, 1 if self.nesterov else 0, 1 if self.steps == 0 else 0)
        self.steps += 1
        _mark_dirty(fb.module)

    def zero_grad(self):
        if self.fb.flat_grad is not None:
            self.fb.flat_grad.zero_()

    def state_dict(self):
        return {"flat": "sgd", "lr": self.lr, "steps": self.steps, "momentum_buffer": None if self.mom is None else self.mom.detach().cpu()}

    def load_state_dict(self, sd):
        self.lr, self.steps = sd.get("lr", self.lr), sd.get("steps", 0)
        mb = sd.get("momentum_buffer")
        self.mom = None if mb is None or self.fb.flat is None or mb.numel() != self.fb.flat.numel() else mb.to(self.fb.flat.device)


class FlatAdam:
    def __init__(self, flat_buffers, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.fb, self.lr, self.betas, self.eps, self.weight_decay = flat_buffers, lr, betas, eps, weight_decay
        self.m = self.v = None
        self.steps = 0

    def step(self):
        fb = self.fb
        if fb.flat_grad is None:
            return
        if self.m is None or self.m.numel() != fb.flat.numel():
            self.m, self.v = torch.zeros_like(fb.flat), torch.zeros_like(fb.flat)
            self.steps = 0
        self.steps += 1
        call("adamml_adam_step", ptr(fb.flat), ptr(fb.flat_grad), ptr(self.m), ptr(self.v), fb.flat.numel(), self.lr,
             self.betas[0], self.betas[1], self.eps, self.weight_decay, self.steps)
        _mark_dirty(fb.module)

    def zero_grad(self):
        if self.fb.flat_grad is not None:
            self.fb.flat_grad.zero_()

    def state_dict(self):
        return {"flat": "adam", "lr": self.lr, "steps": self.steps, "exp_avg": None if self.m is None else self.m.detach().cpu(),
                "exp_avg_sq": None if self.v is None else self.v.detach().cpu()}

    def load_state_dict(self, sd):
        self.lr, self.steps = sd.get("lr", self.lr), sd.get("steps", 0)
        m, v = sd.get("exp_avg"), sd.get("exp_avg_sq")
        ok = m is not None and v is not None and self.fb.flat is not None and m.numel() == self.fb.flat.numel()
        self.m, self.v = (m.to(self.fb.flat.device), v.to(self.fb.flat.device)) if ok else (None, None)
        if not ok:
            self.steps = 0


def _mark_dirty(module):
    for m in module.modules():
        if hasattr(m, "mark_weights_dirty"):
            m.mark_weights_dirty()
