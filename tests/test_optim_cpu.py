"""CPU: the optimizer's host-side contract (no kernels run here: the product path has no CPU fallback, and says so)."""
import pytest
import torch


def test_fused_adam_has_apex_constructor_and_is_a_torch_optimizer():
    """train.py:131 `FusedAdam(net_params, self.hparams.lr, eps=1e-15)` -- positional lr, apex's keyword set; the things the native
    kernels do not implement are refused with apex's wording, not ignored."""
    from torch.optim.lr_scheduler import CosineAnnealingLR
    from ngp_pl_amd.optim import FusedAdam
    params = [torch.nn.Parameter(torch.zeros(8)), torch.nn.Parameter(torch.zeros(0), requires_grad=False)]
    opt = FusedAdam(params, 1e-2, eps=1e-15)
    assert isinstance(opt, torch.optim.Optimizer) and opt.model is None
    g = opt.param_groups[0]
    assert g["lr"] == 1e-2 and g["eps"] == 1e-15 and g["betas"] == (0.9, 0.999) and g["weight_decay"] == 0.0
    sch = CosineAnnealingLR(opt, 30, 1e-2 / 30)
    opt.step(); sch.step()                              # no gradients: nothing to launch, the schedule still moves
    assert 1e-2 / 30 < opt.param_groups[0]["lr"] < 1e-2
    with pytest.raises(RuntimeError, match="AMSGrad"):
        FusedAdam(params, 1e-2, amsgrad=True)
    with pytest.raises(NotImplementedError):
        FusedAdam(params, 1e-2, adam_w_mode=False)
    sd = opt.state_dict()
    opt.load_state_dict(sd)
    assert type(opt).step.hooked is True                # torch's per-call profiler wrapper is not applied ...
    seen = []
    opt.register_step_pre_hook(lambda o, a, k: seen.append("pre"))
    opt.register_step_post_hook(lambda o, a, k: seen.append("post"))
    opt.step()
    assert seen == ["pre", "post"]                      # ... and registered hooks still run


def test_fused_adam_refuses_cpu_gradients():
    from ngp_pl_amd.optim import FusedAdam
    p = torch.nn.Parameter(torch.ones(4))
    opt = FusedAdam([p], 1e-2)
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()
    opt.zero_grad()
    assert p.grad is None                               # set_grad_none=True, apex's default


def test_fused_adam_finds_the_model_behind_bare_parameters():
    """The optimizer is handed tensors, not the model (train.py:123-131): NGP tags its two parameter tensors so that the native
    gradient buffers can be found from them; a list that holds only one of the two is treated as plain tensors."""
    from ngp_pl_amd.networks import NGP
    from ngp_pl_amd.optim import FusedAdam
    m = NGP(scale=0.5)
    params = [p for _, p in m.named_parameters()]
    assert len(params) == 3 and not params[1].requires_grad and params[1].numel() == 0      # dir_encoder.params: empty, no gradient (DDP)
    opt = FusedAdam(params, 1e-2, eps=1e-15)
    assert opt.model is m and m.native_grads is False
    assert FusedAdam(params, 1e-2, native_grads=True).model is m and m.native_grads is True
    m.native_grads = False
    assert FusedAdam([m.rgb_net.params], 1e-2).model is None
    m2 = FusedAdam(m, lr=1e-2, eps=1e-15)               # Trainer's form: the module itself -> native gradients
    assert m2.model is m and m.native_grads is True
    em, ev = m2.moments("enc")
    assert em.shape == m.xyz_encoder.params.shape and float(ev.abs().sum()) == 0.0


def test_state_dict_carries_the_step_count_of_the_native_route():
    """Lightning checkpoints the optimizer (`optimizer.state_dict()`): warm moments reloaded with a step count of 0 would restart
    the bias correction at step 1.  The native route's count travels in the state dict and in state[p]['step'] alike."""
    from ngp_pl_amd.networks import NGP
    from ngp_pl_amd.optim import FusedAdam
    m = NGP(scale=0.5)
    opt = FusedAdam(m, lr=1e-2, eps=1e-15)
    opt.t += 1; opt.t += 1; opt.t += 1                     # what three native steps do
    assert all(opt.state[p]["step"] == 3 for p in (m.xyz_encoder.params, m.rgb_net.params))
    opt.moments("rgb")[0].fill_(0.25)
    sd = opt.state_dict()
    assert sd["ngp_native"] == {"t": 3, "step_state": None}
    m2 = NGP(scale=0.5)
    opt2 = FusedAdam(m2, lr=1e-2, eps=1e-15)
    opt2.load_state_dict(sd)
    assert opt2.t == 3 and opt2.state[m2.rgb_net.params]["step"] == 3 and float(opt2.moments("rgb")[0][0]) == 0.25
    del sd["ngp_native"]                                   # a state dict from a version without the extra record
    opt3 = FusedAdam(NGP(scale=0.5), lr=1e-2, eps=1e-15)
    opt3.load_state_dict(sd)
    assert opt3.t == 3


def test_frame_bytes_is_the_survey_accounting():
    import bench
    # per ray AABB 32 + march 24 in + composite 52; per sample march 32 + encode 588 + MLPs 210 + composite 28
    assert bench.frame_bytes(1, 0) == 108.0 and bench.frame_bytes(0, 0) == 0.0
    assert bench.frame_bytes(640000, 3.5) == 640000 * 108.0 + 640000 * 3.5 * 858.0
