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
        _check_attached(fb, "FlatSGD")
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
        """torch.optim.SGD layout (`param_groups` + per-parameter `state[idx]['momentum_buffer']`, parameters in
        `module.parameters()` order): the reference's `optimizer.load_state_dict(checkpoint['optimizer'])`
        (train_adamml.py:296-297) accepts it as is."""
        ref = torch.optim.SGD(self.fb.params, lr=self.lr, momentum=self.momentum, weight_decay=self.weight_decay, nesterov=self.nesterov)
        sd = ref.state_dict()
        sd["state"] = {} if self.mom is None else {i: {"momentum_buffer": v} for i, v in enumerate(_split_like(self.mom, self.fb.params))}
        return sd

    def load_state_dict(self, sd):
        """Accepts torch.optim.SGD's state_dict (a reference checkpoint) or this class's own."""
        g = sd["param_groups"][0]
        self.lr, self.momentum, self.weight_decay = g["lr"], g.get("momentum", self.momentum), g.get("weight_decay", self.weight_decay)
        self.nesterov = g.get("nesterov", self.nesterov)
        self.mom = _join_state(sd.get("state", {}), "momentum_buffer", self.fb)
        self.steps = 0 if self.mom is None else 1          # torch creates the buffer on the first step: first_step only without one


class FlatAdam:
    def __init__(self, flat_buffers, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.fb, self.lr, self.betas, self.eps, self.weight_decay = flat_buffers, lr, betas, eps, weight_decay
        self.m = self.v = None
        self.steps = 0

    def step(self):
        fb = self.fb
        _check_attached(fb, "FlatAdam")
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
        """torch.optim.Adam layout (`state[idx] = {step, exp_avg, exp_avg_sq}`), loadable by the reference
        (train_adamml.py:298-299)."""
        ref = torch.optim.Adam(self.fb.params, lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay)
        sd = ref.state_dict()
        if self.m is not None:
            ms, vs = _split_like(self.m, self.fb.params), _split_like(self.v, self.fb.params)
            sd["state"] = {i: {"step": torch.tensor(float(self.steps)), "exp_avg": a, "exp_avg_sq": b} for i, (a, b) in enumerate(zip(ms, vs))}
        return sd

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = g["lr"], tuple(g.get("betas", self.betas)), g.get("eps", self.eps)
        self.weight_decay = g.get("weight_decay", self.weight_decay)
        st = sd.get("state", {})
        self.m, self.v = _join_state(st, "exp_avg", self.fb), _join_state(st, "exp_avg_sq", self.fb)
        if self.m is None or self.v is None:
            self.m = self.v = None
            self.steps = 0
        else:
            self.steps = int(max(float(e["step"]) for e in st.values()))


def _split_like(flat, params):
    """Per-parameter CPU copies of a flat state buffer (flat-buffer order == module.parameters() order)."""
    out, off = [], 0
    for p in params:
        n = p.numel()
        out.append(flat[off:off + n].detach().view(p.shape).cpu().clone())
        off += n
    return out


def _check_attached(fb, who):
    """The flat optimizers update from the flat gradient buffer; once the parameter gradients are delivered through autograd
    (backbone.enable_autograd_param_grads: stock DistributedDataParallel / torch.optim own .grad) that buffer does not exist and a
    silent return would leave the weights untouched for the whole run."""
    if fb.detached:
        raise RuntimeError("%s.step(): this sub-network's parameter gradients are delivered through autograd "
                           "(enable_autograd_param_grads / a DistributedDataParallel wrap): use torch.optim on its parameters, or call "
                           "adamml_amd.backbone.enable_autograd_param_grads(model, False) to return to the flat buffers" % who)


def _join_state(state, key, fb, log=None):
    """Flat device buffer from torch's per-parameter optimizer state.  torch.optim creates an entry only for parameters that have
    received a gradient, so entries may be MISSING (a never-updated parameter of a reference checkpoint): those stay zero, as a
    fresh torch buffer would be.  None only when there is no state at all, or when an entry's size does not match its parameter
    (a checkpoint of another architecture) -- the latter is reported."""
    params = fb.params
    if fb.flat is None or not state:
        return None
    flat = torch.zeros_like(fb.flat)
    off, missing = 0, 0
    for i, p in enumerate(params):
        e = state.get(i, state.get(str(i)))
        if e is None or key not in e:
            missing += 1
        elif e[key].numel() != p.numel():
            (log or print)("optimizer state: entry %d ('%s') has %d elements, parameter has %d -- state discarded"
                           % (i, key, e[key].numel(), p.numel()))
            return None
        else:
            flat[off:off + p.numel()].copy_(e[key].reshape(-1))
        off += p.numel()
    if missing == len(params):
        return None
    return flat


def _mark_dirty(module):
    nets = module.__dict__.get("_hip_backbones")
    if nets is None:                    # (the backbones of a sub-network are fixed after construction: walk the module tree once)
        nets = [m for m in module.modules() if hasattr(m, "mark_weights_dirty")]
        module.__dict__["_hip_backbones"] = nets
    for m in nets:
        m.mark_weights_dirty()
