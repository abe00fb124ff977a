"""GPU parity of the diffusion-prior engine (k22_prior_* through the drop-in PriorDiffusionModelHIP) against golden
outputs of the REFERENCE's PriorTransformer.forward and PriorDiffusionModel.forward (kandinsky2/model/prior.py) with
injected sampler noise.

Tolerances: fp32 engine — transformer 2e-4 of the output scale, final sample 1e-3 max-abs relative to the sample scale
(exact-fp32 MFMA, different summation order over K = 2048 / 8192); bf16 engine — transformer 3e-2 of the scale.
"""
import os

import pytest
import torch

import kandinsky2_amd as k22

pytestmark = pytest.mark.gpu


def _inputs(bs, seed=7):
    g = torch.Generator().manual_seed(seed)
    N = 2 * bs
    cm, cs = torch.randn(768, generator=g) * 0.1, torch.rand(768, generator=g) + 0.5
    txt_feat, txt_seq = torch.randn(N, 768, generator=g), torch.randn(N, 77, 768, generator=g)
    mask = torch.zeros(N, 77, dtype=torch.bool)
    for r in range(bs):
        mask[r, : 9 + 11 * r] = True
    mask[bs:, :2] = True
    x = torch.randn(N, 768, generator=g)
    return cm, cs, txt_feat, txt_seq, mask, x, g


# prior_tiny: 512 wide x 4 layers; prior_full: the production configuration (2048 wide x 20 layers, MLP K = 8192:
# CONFIG_2_1["prior"], kandinsky2/configs.py:101-111), which runs other tile configurations and split-K factors
FIXTURES = ["prior_tiny", "prior_full"]


def _fixture(golden_dir, name="prior_tiny"):
    p = os.path.join(golden_dir, name + ".pt")
    if not os.path.exists(p):
        pytest.skip(f"{name}.pt not generated")
    return torch.load(p, weights_only=False)


_SD = {}


def _prior_sd(fx):
    key = (fx["hp"]["xf_width"], fx["hp"]["xf_layers"], fx["seed_w"])
    if key not in _SD:
        _SD.clear()
        _SD[key] = k22.init_prior_state_dict(fx["hp"], seed=fx["seed_w"])
    return _SD[key]


def _model(fx, backend, cm, cs):
    m = k22.PriorDiffusionModelHIP(fx["hp"], k22.PRIOR_DIFFUSION_2_1, cm, cs, backend_dtype=backend)
    m.load_state_dict(_prior_sd(fx))
    return m.to("cuda")


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("backend,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2), (torch.float16, 4e-3)])
def test_prior_transformer_vs_reference_golden(golden_dir, name, backend, tol):
    fx = _fixture(golden_dir, name)
    cm, cs, txt_feat, txt_seq, mask, x, g = _inputs(fx["bs"])
    m = _model(fx, backend, cm, cs)
    out = m.transformer(x.cuda(), fx["t"].cuda(), txt_feat.cuda(), txt_seq.cuda(), mask.cuda()).cpu()
    ref = fx["forward_out"]
    scale = ref.abs().max().item()
    err = (out - ref).abs().max().item()
    print(f"{name} transformer {backend}: max|d|={err:.3e} = {err / scale:.3e} of scale {scale:.3f}")
    assert err <= tol * scale


@pytest.mark.parametrize("name", FIXTURES)
def test_prior_sample_vs_reference_golden_fp32(golden_dir, name):
    fx = _fixture(golden_dir, name)
    cm, cs, txt_feat, txt_seq, mask, x, g = _inputs(fx["bs"])
    N, steps = 2 * fx["bs"], fx["steps"]
    x_T, noise_seq = torch.randn(N, 768, generator=g), torch.randn(steps, N, 768, generator=g)
    m = _model(fx, torch.float32, cm, cs)
    out = m(txt_feat.cuda(), txt_seq.cuda(), mask.cuda(), fx["scales"].cuda(), timestep_respacing=str(steps),
            noise=x_T.cuda(), noise_seq=noise_seq.cuda()).cpu()
    ref = fx["sample"]
    scale = ref.abs().max().item()
    err = (out - ref).abs().max().item()
    print(f"{name} {steps}-step sample fp32: max|d|={err:.3e} scale={scale:.3f}")
    assert out.shape == ref.shape and err <= 1e-3 * scale


@pytest.mark.parametrize("name", FIXTURES)
def test_prior_bf16_sample_drift_is_bounded(golden_dir, name):
    fx = _fixture(golden_dir, name)
    cm, cs, txt_feat, txt_seq, mask, x, g = _inputs(fx["bs"])
    N, steps = 2 * fx["bs"], fx["steps"]
    x_T, noise_seq = torch.randn(N, 768, generator=g), torch.randn(steps, N, 768, generator=g)
    m = _model(fx, torch.bfloat16, cm, cs)
    out = m(txt_feat.cuda(), txt_seq.cuda(), mask.cuda(), fx["scales"].cuda(), timestep_respacing=str(steps),
            noise=x_T.cuda(), noise_seq=noise_seq.cuda()).cpu()
    ref = fx["sample"]
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    print(f"{name} bf16 {steps}-step sample drift: {err:.3e} of the sample scale")
    assert torch.isfinite(out).all() and err <= 0.1


def test_prior_rejects_cpu_tensor(golden_dir):
    fx = _fixture(golden_dir)
    cm, cs, txt_feat, txt_seq, mask, x, g = _inputs(fx["bs"])
    m = _model(fx, torch.float32, cm, cs)
    with pytest.raises(RuntimeError):
        m.transformer(x, fx["t"], txt_feat, txt_seq, mask)
