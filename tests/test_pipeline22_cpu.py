"""CPU suite for the Kandinsky 2.2 drivers' host logic (pipeline22.py): size rules, schedule slicing of img2img, the conditioner
protocol of the wrapper, loud failure without a GPU.  The 2.2 arithmetic itself is PARITY UNPINNED (oracle/unet22_ref.py)."""
import pytest
import torch

import kandinsky2_amd as k22
from kandinsky2_amd import pipeline22


def test_latent_size_rule_and_new_h_w():
    # downscale_height_and_width of the diffusers Kandinsky pipelines: ceil(x / 64) * 8
    for hw, want in [((768, 768), (96, 96)), ((512, 520), (64, 72)), ((100, 130), (16, 24)), ((64, 63), (8, 8))]:
        assert pipeline22._downscale(*hw) == want
    m = pipeline22.Kandinsky2_2HIP.__new__(pipeline22.Kandinsky2_2HIP)
    assert m.get_new_h_w(512, 512) == (512, 512) and m.get_new_h_w(500, 65) == (512, 128)     # kandinsky2_2_model.py:46-53


def test_img2img_timestep_slice():
    d = pipeline22.KandinskyV22Img2ImgDecoderHIP(None, None, None)
    full = list(range(990, -1, -10))
    assert d.get_timesteps(100, 1.0, "cpu") == full
    assert d.get_timesteps(100, 0.4, "cpu") == full[60:]            # reference default strength (kandinsky2_2_model.py:86)
    assert d.get_timesteps(10, 0.55, "cpu") == [400, 300, 200, 100, 0]
    with pytest.raises(ValueError):
        d.get_timesteps(10, 0.05, "cpu")


def test_seeded_prior_protocol():
    c = pipeline22.SeededPrior22()
    a, an = c.prior22("a cat", "", 2, 25, 4, "cpu")
    b, bn = c.prior22("a dog", None, 2, 25, 4, "cpu")
    _, cn = c.prior22("a bird", None, 2, 25, 4, "cpu")
    assert a.shape == (2, 1280) and torch.equal(a[0], a[1]) and not torch.equal(a, b)
    assert torch.equal(bn, cn) and not torch.equal(an, bn)          # negative_prompt=None: the zero-image embedding, prompt-independent
    assert torch.equal(c.prior22("a cat", "", 2, 25, 4, "cpu")[0], a)
    img = torch.zeros(1, 3, 8, 8)
    assert c.encode_image22(img, "cpu").shape == (1, 1280) and torch.equal(c.encode_image22(img, "cpu"), c.encode_image22(img.clone(), "cpu"))


def test_wrapper_argument_checks_and_no_cpu_fallback():
    with pytest.raises(ValueError):
        pipeline22.Kandinsky2_2HIP("cuda", "superres", unet_state_dict={}, movq_state_dict={})
    with pytest.raises(FileNotFoundError):
        pipeline22.Kandinsky2_2HIP("cuda", "text2img")
    with pytest.raises(ValueError):
        pipeline22.Kandinsky2_2HIP("cuda", "img2img", unet_state_dict={}, movq_state_dict={}, controlnet=True)
    d = pipeline22.KandinskyV22DecoderHIP(None, None)
    with pytest.raises(RuntimeError):
        d(torch.zeros(1, 1280), torch.zeros(1, 1280))
    with pytest.raises(ValueError):
        d._check(type("T", (), {"device": torch.device("cuda"), "shape": (5, 1280)})())        # CFG batch 10 > 8
    sch = k22.DDPMSchedulerHIP().set_timesteps(50, device="cpu")
    assert sch.timesteps.tolist() == list(range(980, -1, -20))
