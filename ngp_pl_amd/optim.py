"""Fused Adam for the NGP field: apex FusedAdam's update (the reference's optimizer, /root/reference/train.py:131-137:
`FusedAdam(net_params, lr, eps=1e-15)` under `CosineAnnealingLR`; adam_w_mode, bias correction) as a `torch.optim.Optimizer`
that is constructed the way the reference constructs apex's -- from a list of parameters -- and that recognises the two parameter
tensors of an `ngp_pl_amd.networks.NGP` among them.

Two sources of gradients, decided per step by what the backward left behind:
  * NATIVE (`model.native_grads = True`: `Trainer`, bench.py's `api_path`): the fused backward leaves the packed-f16 grid gradient
    and the f32 per-workgroup MLP partial rows in native buffers (`model._native`); `ngp_adam_step_field` applies unscale + Adam +
    f32->f16 parameter cast to the grid table and both MLP blocks in ONE launch and ONE pass over the 11.4 M parameters;
  * `.grad` (`model.native_grads = False`, the default of a freshly built NGP: what an unchanged train.py with Lightning's
    GradScaler / DistributedDataParallel sees): f32 gradients on the Parameters, as tiny-cuda-nn's modules produce them;
    `ngp_adam_step` per parameter tensor (Adam + the f16 working-copy refresh tiny-cuda-nn performs each forward).
Parameters that belong to no NGP (none in the reference's recipe; `dR` / `dT` have their own torch Adam, train.py:117-122) are
updated by the same per-tensor kernel.  CPU parameters are refused: the product path has no CPU fallback.
"""
import contextlib
import math
import weakref

import torch

from ._lib import call, device_guard, ptr, stream


_bump_version = getattr(torch._C, "_increment_version", None)


def tag_parameters(model):
    """Called by NGP.__init__: lets an optimizer that is handed bare parameter tensors (train.py:123-131) find the model whose
    native gradient buffers they belong to.  The tags live on the Parameter objects (`.to(device)` keeps the objects)."""
    ref = weakref.ref(model)
    for role, mod in (("enc", model.xyz_encoder), ("rgb", model.rgb_net)):
        mod.params._ngp_model = ref
        mod.params._ngp_role = role


class FusedAdam(torch.optim.Optimizer):
    """apex.optimizers.FusedAdam's constructor (params, lr, bias_correction, betas, eps, adam_w_mode, weight_decay, amsgrad,
    set_grad_none).  `params` may also be an NGP module (its parameters are taken, and the model is switched to native gradients:
    what `Trainer` does)."""

    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-8, adam_w_mode=True, weight_decay=0.0,
                 amsgrad=False, set_grad_none=True, native_grads=None):
        if amsgrad:
            raise RuntimeError("FusedAdam does not support the AMSGrad variant.")          # apex's own message
        if not (bias_correction and adam_w_mode):
            raise NotImplementedError("the native kernels implement apex's defaults: bias_correction=True, adam_w_mode=True")
        model = None
        if isinstance(params, torch.nn.Module):
            model = params
            params = list(model.parameters())
            if native_grads is None:
                native_grads = True
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.set_grad_none = set_grad_none
        self._roles = {}
        self._t = 0                      # optimizer steps so far (1-based inside a step): see the `t` property
        self._step_state = None          # device-side counts of APPLIED steps, once a skip flag has been seen (ngp_adam_step_field)
        # the NGP whose (xyz_encoder.params, rgb_net.params) are both in this optimizer
        for group in self.param_groups:
            for p in group["params"]:
                ref = getattr(p, "_ngp_model", None)
                m = ref() if ref is not None else None
                if m is None:
                    continue
                if model is None:
                    model = m
                if m is model:
                    self._roles[p._ngp_role] = p
        self.model = model if (model is not None and set(self._roles) == {"enc", "rgb"}) else None
        if self.model is not None:
            enc, net = self.model.xyz_encoder, self.model.rgb_net
            for p in (enc.params, net.params):
                self._moments(p)
            # make sure the f16 working copies exist; from now on this optimizer keeps them fresh
            if enc.params.is_cuda:
                enc._half.get(enc.params); net._half.get(net.params)
            if native_grads is not None:
                self.model.native_grads = bool(native_grads)

    # -- state -----------------------------------------------------------------------------------
    @property
    def t(self):
        """Bias-correction step count of the model's two parameter tensors.  ONE count, whichever route the gradients take: the
        native route (and the trainer's native stepper) advances it here, the `.grad` route through state[p]['step'] -- the two
        are kept equal, so that switching `model.native_grads` mid-run or reloading a checkpoint continues the same count."""
        return self._t

    @t.setter
    def t(self, v):
        self._t = int(v)
        for p in self._roles.values():
            if "exp_avg" in self.state[p]:
                self.state[p]["step"] = self._t

    def state_dict(self):
        """torch's optimizer state (per-parameter step / exp_avg / exp_avg_sq, param groups) plus what the native route keeps
        outside it: the step count and, once a skip flag has been in use, the device-side counts of APPLIED steps."""
        sd = super().state_dict()
        sd["ngp_native"] = {"t": self._t, "step_state": None if self._step_state is None else self._step_state.tolist()}
        return sd

    def load_state_dict(self, state_dict):
        extra = state_dict.get("ngp_native")
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "ngp_native"})
        if extra is None:                # written by another optimizer class / an older version: the per-parameter counts decide
            steps = [int(self.state[p].get("step", 0)) for p in self._roles.values() if p in self.state]
            self._t = max(steps, default=0)
            self._step_state = None
        else:
            self.t = extra["t"]
            ss = extra.get("step_state")
            if ss is None:
                self._step_state = None
            else:
                self._step_state = torch.tensor(ss, dtype=torch.int32, device=self._roles["enc"].device)

    def _moments(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p.data)
            st["exp_avg_sq"] = torch.zeros_like(p.data)
        return st["exp_avg"], st["exp_avg_sq"]

    def moments(self, role):
        """(m, v) of the model's 'enc' (density MLP + grid table) or 'rgb' parameter tensor."""
        return self._moments(self._roles[role])

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @property
    def betas(self):
        return self.param_groups[0]["betas"]

    @property
    def eps(self):
        return self.param_groups[0]["eps"]

    @property
    def weight_decay(self):
        return self.param_groups[0]["weight_decay"]

    def step_state(self, found_inf):
        """Device pointer of the applied-step counts (ngp_adam_step_field's `step_state`), or None while no skip flag has ever been
        handed to this optimizer: without a flag every call applies, and the host count `t` IS the bias-correction step."""
        if found_inf is None and self._step_state is None:
            return None
        return self.ensure_step_state(self.t - 1).data_ptr()                   # (called inside a step: t counts this call already)

    def ensure_step_state(self, applied=None):
        """The device-side applied-step counts (4 x i32), created on first use holding `applied` (default: the calls so far)."""
        if self._step_state is None:
            dev = self._roles["enc"].device
            self._step_state = torch.full((4,), self.t if applied is None else applied, dtype=torch.int32, device=dev)
        return self._step_state

    def applied_steps(self):
        """Steps actually applied to (MLP blocks, grid block) -- equal to `t` unless a skip flag was raised (syncs)."""
        if self._step_state is None:
            return self.t, self.t
        s = self._step_state.tolist()
        return s[self.t & 1], s[2 + (self.t & 1)]

    def zero_grad(self, set_to_none=None):
        # (the native gradient buffers need no clearing: every table backward overwrites them, the MLP partial rows are rewritten)
        super().zero_grad(set_to_none=self.set_grad_none if set_to_none is None else set_to_none)

    # -- the update ------------------------------------------------------------------------------
    def step(self, closure=None, grad_scale=1.0, found_inf=None, stream_handle=None):
        """torch.optim.Optimizer.step protocol (closure, registered step pre / post hooks) without torch's per-call profiler
        wrapper (`Optimizer.profile_hook_step`: a record_function scope, hook-dict chains and a no_grad decorator measured ~60 us of
        host time per step on the reference-shaped path, which is host-bound): `step.hooked` below tells Optimizer.__init__ this
        method runs the hooks itself."""
        if self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks or _global_hooks():
            return self._step_with_hooks(closure, grad_scale, found_inf, stream_handle)
        prev = torch.is_grad_enabled()
        torch.set_grad_enabled(False)
        try:
            return self._step(closure, grad_scale, found_inf, stream_handle)
        finally:
            torch.set_grad_enabled(prev)

    def _step_with_hooks(self, closure, grad_scale, found_inf, stream_handle):
        from torch.optim import optimizer as _o
        args, kwargs = (self,), dict(closure=closure, grad_scale=grad_scale, found_inf=found_inf, stream_handle=stream_handle)
        for hook in list(_o._global_optimizer_pre_hooks.values()) + list(self._optimizer_step_pre_hooks.values()):
            result = hook(self, args, kwargs)
            if result is not None:
                args, kwargs = result
        with torch.no_grad():
            out = self._step(**kwargs)
        for hook in list(self._optimizer_step_post_hooks.values()) + list(_o._global_optimizer_post_hooks.values()):
            hook(self, args, kwargs)
        return out

    def _step(self, closure=None, grad_scale=1.0, found_inf=None, stream_handle=None):
        """grad_scale: extra factor the caller put on the loss (a GradScaler-style scale applied OUTSIDE torch's GradScaler, which
        unscales `.grad` itself); the kernels' own loss scale is taken from the native record.  found_inf: device int32 flag
        (non-zero: skip) -- with it the bias-correction count lives on the device too and does not advance on a skipped step."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        model = self.model
        nat = model._native if model is not None else None
        native_done = False
        if nat is not None:
            self._step_native(nat, grad_scale, found_inf, stream_handle)
            native_done = True
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if native_done and (p is model.xyz_encoder.params or p is model.rgb_net.params):
                    raise RuntimeError("both a native gradient record and a .grad tensor for the same NGP parameter: a backward ran with "
                                       "model.native_grads switched in between")
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam: parameter on %s -- the native optimizer has no CPU fallback" % p.device)
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients, please consider SparseAdam instead")
                if p.numel() == 0:
                    continue
                m, v = self._moments(p)
                self.state[p]["step"] += 1
                if model is not None and (p is model.xyz_encoder.params or p is model.rgb_net.params):
                    self._t = max(self._t, self.state[p]["step"])          # (one count for both gradient routes, see `t`)
                if p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                    p.grad = p.grad.float().contiguous()
                half = self._half_of(p)
                with device_guard(p.device):
                    # f32 gradient, CONSUMED: the kernel leaves .grad zero-filled (apex leaves it alone; every training loop clears
                    # it next, Lightning included -- and a copy to preserve it would add 2 x 45.7 MB of traffic per step)
                    call("ngp_adam_step", ptr(p.data), ptr(half), ptr(p.grad), 1, ptr(m), ptr(v), p.numel(), group["lr"], b1, b2,
                         group["eps"], group["weight_decay"], self.state[p]["step"], float(grad_scale), ptr(found_inf),
                         stream_handle if stream_handle is not None else stream())
                # the kernel wrote p behind autograd's back: bump the version counter, so that an f16 working copy this optimizer did
                # NOT refresh (a module that lost its tag, e.g. a deep copy of the model) is re-cast at its next forward ...
                if _bump_version is not None:
                    _bump_version([p])
                mod = self._module_of(p)
                if mod is not None and half is not None:
                    mod._half.mark_fresh(p)             # ... and one it did refresh is recorded as current for the new version
        return loss

    def _module_of(self, p):
        ref = getattr(p, "_tcnn_module", None)
        return ref() if ref is not None else None

    def _half_of(self, p):
        """The f16 working copy a tiny-cuda-nn style module keeps of `p` (refreshed by the kernel), or None for a plain tensor."""
        mod = self._module_of(p)
        if mod is None or not hasattr(mod, "_half"):
            return None
        if mod._half.t is None or mod._half.t.shape != p.shape or mod._half.t.device != p.device:
            mod._half.get(p)
        return mod._half.t

    def _step_native(self, nat, grad_scale, found_inf, stream_handle):
        model = self.model
        enc, net = model.xyz_encoder, model.rgb_net
        self.t += 1
        group = self.param_groups[0]
        lr = group["lr"]
        b1, b2 = group["betas"]
        total_scale = nat["scale"] * grad_scale
        sq = stream_handle if stream_handle is not None else stream()
        h = nat.get("stepper")
        if h is not None:
            # gradients left by render()'s native node: its stepper hands this launch the dynamic loss scale its field backward ran
            # under and names the overflow flag that backward may have raised (the skip decision, GradScaler's, on the device)
            import ctypes as C
            fi = C.c_void_p()
            call("ngp_stepper_before_update", h, C.byref(fi))
            if found_inf is None and fi.value:
                found_inf = _FlagRef(fi.value)
        m, v = self.moments("enc")
        rm, rv = self.moments("rgb")
        ne = enc.n_mlp
        # raw pointers by arithmetic (a tensor slice costs ~4 us of host time, this call passes 6 of them)
        p_enc, p_half, p_m, p_v = enc.params.data_ptr(), enc._half.t.data_ptr(), m.data_ptr(), v.data_ptr()
        guard = device_guard(enc.params.device) if stream_handle is None else contextlib.nullcontext()     # a raw stream handle names its device
        try:
            with guard:
                call("ngp_adam_step_field", p_enc + 4 * ne, p_half + 2 * ne, ptr(nat["grid16"]), p_m + 4 * ne, p_v + 4 * ne, enc.n_grid,
                     p_enc, p_half, ptr(nat["density_partials"]), p_m, p_v, ne,
                     net.params.data_ptr(), net._half.t.data_ptr(), ptr(nat["rgb_partials"]), rm.data_ptr(), rv.data_ptr(), net.params.numel(),
                     nat["n_partials"], lr, b1, b2, group["eps"], group["weight_decay"], self.t, total_scale, 0, ptr(found_inf),
                     self.step_state(found_inf), sq)      # 0: every table backward of this package overwrites the gradient
        finally:
            if h is not None:                         # (a launch that went out has consumed the hand-over; one that raised must not leave it behind)
                call("ngp_adam_use_loss_scaler", None, 0, 2.0, 0.5, 1, 1.0, 1.0)
        enc._half.mark_fresh(enc.params); net._half.mark_fresh(net.params)
        model._native = None


class _FlagRef:
    """A device int32 flag owned by the library (the stepper's overflow guard), passed where a found_inf tensor is expected."""
    def __init__(self, address):
        self.address = address

    def data_ptr(self):
        return self.address


FusedAdam.step.hooked = True          # Optimizer._patch_step_function: do not wrap (step() runs registered hooks itself)


def _global_hooks():
    from torch.optim import optimizer as _o
    return bool(_o._global_optimizer_pre_hooks) or bool(_o._global_optimizer_post_hooks)


def cosine_lr(base_lr, epoch, num_epochs, eta_min_ratio=1 / 30):
    """torch CosineAnnealingLR(opt, num_epochs, lr/30) stepped per epoch (train.py:135-137)."""
    eta_min = base_lr * eta_min_ratio
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / num_epochs)) / 2
