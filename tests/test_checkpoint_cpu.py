"""CPU: the checkpoint surface (utils.py:4-39 of the reference) and the known-answer lengths of `xyz_encoder.params`
(3 072 MLP weights + the hash grid; 11 423 136 if the level table is evaluated exactly,
11 448 112 if it is evaluated in float32 the way tiny-cuda-nn's grid.h does)."""
import pytest
import torch

from ngp_pl_amd import utils
from ngp_pl_amd.networks import NGP


def lightning_ckpt(model, n_enc=None):
    sd = {"model." + k: v.clone() for k, v in model.state_dict().items()}
    if n_enc is not None:
        sd["model.xyz_encoder.params"] = torch.arange(n_enc, dtype=torch.float32) * 1e-7
    sd["directions"] = torch.zeros(4, 3); sd["poses"] = torch.zeros(2, 3, 4); sd["val_lpips.net.weight"] = torch.zeros(3)
    return {"state_dict": sd, "epoch": 29}


def test_param_count_known_answers():
    m32, mex = NGP(scale=0.5), NGP(scale=0.5, level_table="exact")
    assert m32.xyz_encoder.params.numel() == 11_448_112 and mex.xyz_encoder.params.numel() == 11_423_136
    assert [mex.xyz_encoder.meta.resolution[l] for l in range(16)] == [16, 22, 28, 37, 49, 64, 85, 112, 148, 195, 256, 338, 446, 589, 777, 1024]
    r32 = [m32.xyz_encoder.meta.resolution[l] for l in range(16)]
    # float32: 2^(0.4 l) * 16 - 1 lands a few 1e-6 above the integers 63, 255, 1023 at l = 5, 10, 15 -> one more vertex there
    assert r32 == [16, 22, 28, 37, 49, 65, 85, 112, 148, 195, 257, 338, 446, 589, 777, 1025]
    assert m32.rgb_net.params.numel() == mex.rgb_net.params.numel() == 7168 and m32.dir_encoder.params.numel() == 0
    # mip-NeRF360 recipe: scale 16 -> b = exp(ln(2048)/15), 6 cascades
    big = NGP(scale=16.0, level_table="exact")
    assert big.cascades == 6 and big.xyz_encoder.meta.resolution[15] == 32768


def test_load_ckpt_roundtrip_and_slim():
    src = NGP(scale=0.5, level_table="exact")
    src.register_training_buffers()
    with torch.no_grad():
        src.xyz_encoder.params.uniform_(-1, 1); src.rgb_net.params.uniform_(-1, 1); src.density_bitfield.fill_(7)
    ck = lightning_ckpt(src)
    slim = utils.slim_ckpt({"state_dict": dict(ck["state_dict"])})
    assert "model.density_grid" not in slim and "directions" not in slim and "poses" not in slim and "val_lpips.net.weight" not in slim
    assert "model.xyz_encoder.params" in slim and "model.density_bitfield" in slim
    dst = NGP(scale=0.5, level_table="exact")                  # no training buffers: the slim checkpoint has none either
    utils.load_ckpt(dst, slim)
    assert torch.equal(dst.xyz_encoder.params, src.xyz_encoder.params) and torch.equal(dst.rgb_net.params, src.rgb_net.params)
    assert torch.equal(dst.density_bitfield, src.density_bitfield)
    # prefixes_to_ignore (utils.py:12-15)
    d2 = NGP(scale=0.5, level_table="exact")
    before = d2.rgb_net.params.detach().clone()
    utils.load_ckpt(d2, slim, prefixes_to_ignore=["rgb_net"])
    assert torch.equal(d2.rgb_net.params, before) and torch.equal(d2.xyz_encoder.params, src.xyz_encoder.params)


def test_checkpoint_of_the_other_level_table_is_refused_with_the_remedy():
    m32 = NGP(scale=0.5)
    with pytest.raises(RuntimeError, match=r"11423136 entries.*exact arithmetic.*level_table='exact'"):
        utils.load_ckpt(m32, lightning_ckpt(m32, n_enc=11_423_136))
    mex = NGP(scale=0.5, level_table="exact")
    utils.load_ckpt(mex, lightning_ckpt(m32, n_enc=11_423_136))          # ... and loads into the table it names
    assert float(mex.xyz_encoder.params.detach()[-1]) == pytest.approx((11_423_136 - 1) * 1e-7)
    with pytest.raises(RuntimeError, match=r"11448112 entries.*float32 arithmetic.*level_table='float32'"):
        utils.load_ckpt(mex, lightning_ckpt(m32))
    with pytest.raises(RuntimeError, match="different scale"):
        utils.load_ckpt(m32, lightning_ckpt(m32, n_enc=12345))
