"""FlatAdam: torch.optim.Adam semantics (no weight decay / amsgrad) over ONE flat fp32 bucket,
stepped by a single fused HIP kernel.

The reference builds `Adam(params, lr=1e-4, betas=(0.5, 0.9))` for each network
(experiment/experiment.py:111-117).  FlatAdam is the MI355X-native equivalent behind the same
`.step()` / `.zero_grad()` interface:
  * parameters are re-homed as views of one contiguous bucket, gradients likewise, so a
    data-parallel step is one RCCL all-reduce of the gradient bucket followed by one kernel;
  * the step counter lives on the device, so the whole step is hipGraph-capturable;
  * zero_grad() is one fill of the gradient bucket; the .grad views stay alive and the
    weight-grad kernels accumulate into them directly (no AccumulateGrad adds, no per-step
    re-allocation of 60+ tensors).
"""
import torch

from ._ops import prims as P


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        defaults = dict(lr=lr, betas=betas, eps=eps)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam keeps one bucket: pass a single parameter group")
        self._flat = None
        self.grad_scale = 1.0          # set to 1/world_size when gradients are all-reduce-summed

    # ---- bucket management
    def _params(self):
        return [p for p in self.param_groups[0]["params"] if p.requires_grad or p.grad is not None]

    def _bucket_ok(self):
        if self._flat is None:
            return False
        fp, fg, _, _, _, views = self._flat
        ps = self.param_groups[0]["params"]
        if len(views) != len(ps):
            return False
        for p, (off, n) in zip(ps, views):
            if p.data_ptr() != fp.data_ptr() + 4 * off or p.device != fp.device:
                return False
        return True

    def flatten(self):
        """(Re)builds the bucket; call after the module has been moved to its device."""
        ps = self.param_groups[0]["params"]
        dev = ps[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam: parameters must live on the HIP device (got %s); there is "
                               "no CPU path" % dev)
        offs, total = [], 0
        for p in ps:
            if p.dtype != torch.float32:
                raise RuntimeError("FlatAdam: fp32 parameters only")
            offs.append((total, p.numel()))
            total += (p.numel() + 3) // 4 * 4          # keep every view 16-byte aligned
        old = self._flat
        fp = torch.zeros(total, dtype=torch.float32, device=dev)
        # the gradient bucket carries 4 spare floats behind the last parameter: element [total] is the
        # step's loss value, so that under data parallelism the rank-sum of the loss rides in the same
        # all-reduce as the gradient slice it sits behind (train.py) -- no extra collective
        self._grad_store = torch.zeros(total + 4, dtype=torch.float32, device=dev)
        fg = self._grad_store[:total]
        if old is not None and old[0].numel() == total and old[0].device == dev:
            m, v, step = old[2], old[3], old[4]
        else:
            m = torch.zeros(total, dtype=torch.float32, device=dev)
            v = torch.zeros(total, dtype=torch.float32, device=dev)
            step = torch.zeros(1, dtype=torch.int32, device=dev)
        with torch.no_grad():
            for p, (off, n) in zip(ps, offs):
                fp[off:off + n].copy_(p.data.reshape(-1))
                p.data = fp[off:off + n].view(p.shape)
                g = fg[off:off + n].view(p.shape)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g
                p._ms_slot = g          # the backward kernels accumulate here directly (functional._sink_for)
        self._flat = (fp, fg, m, v, step, offs)

    @property
    def flat_params(self):
        if not self._bucket_ok():
            self.flatten()
        return self._flat[0]

    @property
    def flat_grads(self):
        if not self._bucket_ok():
            self.flatten()
        return self._flat[1]

    @property
    def grad_store(self):
        """The gradient bucket plus its 4 trailing spare floats ([numel-4] = the loss slot)."""
        self.flat_grads
        return self._grad_store

    @property
    def loss_slot(self):
        """0-d view of the spare float behind the gradient bucket (see flatten)."""
        return self.grad_store[-4:-3].view(())

    def bucket_version(self):
        """Changes whenever the buckets are re-homed: captured hipGraphs bake the bucket addresses in, so
        the trainers re-plan when this differs from the value they captured with."""
        fp = self.flat_params
        return (fp.data_ptr(), self._flat[1].data_ptr(), self._flat[2].data_ptr())

    def grad_views(self):
        """Per-parameter views of the gradient bucket (state_dict order)."""
        fg = self.flat_grads
        ps = self.param_groups[0]["params"]
        return [fg[off:off + n].view(p.shape) for p, (off, n) in zip(ps, self._flat[5])]

    # ---- Optimizer interface
    def zero_grad(self, set_to_none=True):
        """One fill of the flat gradient bucket; the per-parameter .grad views stay bound to it
        (set_to_none is accepted for interface compatibility and ignored)."""
        if not self._bucket_ok():
            self.flatten()
        ps = self.param_groups[0]["params"]
        fg = self._flat[1]
        fg.zero_()
        for p, (off, n) in zip(ps, self._flat[5]):
            if p.grad is None or p.grad.data_ptr() != fg.data_ptr() + 4 * off:
                p.grad = fg[off:off + n].view(p.shape)
            p._ms_slot = p.grad

    def _gather_stray_grads(self):
        """A gradient produced outside the bucket (plain autograd assigns a fresh tensor when
        .grad was None, e.g. after `net.zero_grad()`) is copied into its slot and .grad is re-bound
        to the slot.  The backward kernels only ever write a slot while it IS p.grad
        (functional._bound_slot), so a slot whose parameter has no .grad holds nothing of this step
        and is cleared: such a parameter steps with a zero gradient (torch.optim.Adam would skip it)."""
        ps = self.param_groups[0]["params"]
        fg = self._flat[1]
        with torch.no_grad():
            for p, (off, n) in zip(ps, self._flat[5]):
                if p.grad is None:
                    fg[off:off + n].zero_()
                elif p.grad.data_ptr() != fg.data_ptr() + 4 * off:
                    fg[off:off + n].copy_(p.grad.reshape(-1))
                    p.grad = fg[off:off + n].view(p.shape)
                    p._ms_slot = p.grad

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._bucket_ok():
            self.flatten()
        self._gather_stray_grads()
        fp, fg, m, v, step, _ = self._flat
        g = self.param_groups[0]
        P.adam_step(fp, fg, m, v, step, g["lr"], g["betas"][0], g["betas"][1], g["eps"], self.grad_scale)
        return loss

    def step_count(self):
        return int(self._flat[4].item()) if self._flat is not None else 0

    # ---- checkpoints: the moments and the step counter live in the flat bucket, not in self.state
    def state_dict(self):
        sd = super().state_dict()
        if self._flat is not None:
            _, _, m, v, step, offs = self._flat
            sd["flat"] = {"exp_avg": m.detach().cpu(), "exp_avg_sq": v.detach().cpu(),
                          "step": step.detach().cpu(), "layout": [tuple(o) for o in offs]}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        flat = state_dict.pop("flat", None)
        super().load_state_dict(state_dict)
        if flat is not None:
            # keep the bucket where it is when it is intact: captured hipGraphs (train.py) hold its addresses;
            # re-homing parameters and gradients here would leave them replaying on freed storage
            if not self._bucket_ok():
                self.flatten()
            _, _, m, v, step, offs = self._flat
            if [tuple(o) for o in offs] != [tuple(o) for o in flat["layout"]]:
                raise ValueError("FlatAdam.load_state_dict: the checkpoint's bucket layout does not match "
                                 "these parameters")
            m.copy_(flat["exp_avg"]); v.copy_(flat["exp_avg_sq"]); step.copy_(flat["step"])
