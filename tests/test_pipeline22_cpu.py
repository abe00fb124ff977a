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
    with pytest.raises(ValueError, match="conditioner"):      # no silent seeded-noise conditioning (the prompt would be ignored)
        pipeline22.Kandinsky2_2HIP("cuda", "text2img", unet_state_dict={}, movq_state_dict={})
    with pytest.raises(ValueError, match="conditioner"):
        k22.Kandinsky2_1HIP(k22.CONFIG_2_1, {}, {}, "cuda")
    with pytest.raises(FileNotFoundError):                    # get_kandinsky2 reads cache_dir (both versions) and never substitutes weights
        k22.get_kandinsky2("cuda", cache_dir="/nonexistent", model_version="2.1")
    with pytest.raises(FileNotFoundError):
        k22.get_kandinsky2("cuda", cache_dir="/nonexistent", model_version="2.2", conditioner="seeded")


def test_decoder22_loads_from_a_cache_dir_and_its_json_drives_the_build(tmp_path):
    """kandinsky2_2_model.py:26-41 through local files: unet/config.json and scheduler_config.json decide architecture and scheduler."""
    import json
    from safetensors.torch import save_file
    root = tmp_path / "kandinsky-2-2-decoder"
    for sub in ("unet", "movq", "scheduler"):
        (root / sub).mkdir(parents=True)
    save_file({"conv_in.weight": torch.zeros(2, 2)}, str(root / "unet" / "diffusion_pytorch_model.safetensors"))
    torch.save({"decoder.conv_in.weight": torch.zeros(1)}, str(root / "movq" / "diffusion_pytorch_model.bin"))
    json.dump(dict(k22.tiny_unet22_config(), _class_name="UNet2DConditionModel"), open(root / "unet" / "config.json", "w"))
    json.dump(dict(k22.SCHEDULER_CONFIG_2_2, variance_type="learned_range"), open(root / "scheduler" / "scheduler_config.json", "w"))
    got = pipeline22.load_decoder22_from_cache_dir(str(tmp_path))
    assert set(got["unet"]) == {"conv_in.weight"} and set(got["movq"]) == {"decoder.conv_in.weight"}
    assert tuple(got["unet_config"]["block_out_channels"]) == (128, 256, 384, 512) and got["scheduler_config"]["variance_type"] == "learned_range"
    assert k22.make_arch22(got["unet_config"]).model_channels == 128
    assert k22.DDPMSchedulerHIP.from_config(got["scheduler_config"]).config.variance_type == "learned_range"
    with pytest.raises(FileNotFoundError):
        pipeline22.load_decoder22_from_cache_dir(str(tmp_path), task_type="inpainting")
    d = pipeline22.KandinskyV22DecoderHIP(None, None)
    with pytest.raises(RuntimeError):
        d(torch.zeros(1, 1280), torch.zeros(1, 1280))
    with pytest.raises(ValueError):
        d._check(type("T", (), {"device": torch.device("cuda"), "shape": (5, 1280)})())        # CFG batch 10 > 8
    sch = k22.DDPMSchedulerHIP.from_config(k22.SCHEDULER_CONFIG_2_2).set_timesteps(50, device="cpu")
    assert sch.timesteps.tolist() == list(range(980, -1, -20))


def test_scheduler_and_unet_configs_are_config_driven():
    """Nothing about the 2.2 checkpoint is hard-coded: absent keys take diffusers' defaults and are reported, as in the notebook log
    (notebooks/lora_decoder.ipynb:3661-3662, 3668-3670); values the engine was not built for raise."""
    sch = k22.DDPMSchedulerHIP.from_config(dict(k22.SCHEDULER_CONFIG_2_2, _diffusers_version="0.18.0.dev0"))
    logged = {"trained_betas", "sample_max_value", "dynamic_thresholding_ratio", "clip_sample_range", "variance_type", "timestep_spacing"}
    assert logged <= set(sch.missing_keys) and sch.config.variance_type == "fixed_small" and sch.config.timestep_spacing == "leading"
    assert k22.DDPMSchedulerHIP().config.beta_end == 0.02 and k22.DDPMSchedulerHIP().clip == 1.0           # diffusers' own defaults
    assert k22.DDPMSchedulerHIP.from_config(k22.SCHEDULER_CONFIG_2_2_LEARNED_RANGE).clip == 2.0
    for bad in (dict(prediction_type="v_prediction"), dict(thresholding=True), dict(variance_type="learned"), dict(_class_name="DDIMScheduler"),
                dict(rescale_betas_zero_snr=True)):
        with pytest.raises((NotImplementedError, ValueError)):
            k22.DDPMSchedulerHIP.from_config(dict(k22.SCHEDULER_CONFIG_2_2, **bad))
    for n, spacing, want in ((4, "leading", [750, 500, 250, 0]), (4, "trailing", [999, 749, 499, 249]), (4, "linspace", [999, 666, 333, 0])):
        s2 = k22.DDPMSchedulerHIP.from_config(dict(k22.SCHEDULER_CONFIG_2_2, timestep_spacing=spacing)).set_timesteps(n, device="cpu")
        assert s2.timesteps.tolist() == want
    full = k22.resolve_unet22_config(k22.UNET_CONFIG_2_2)
    assert {"addition_time_embed_dim", "transformer_layers_per_block", "num_attention_heads"} <= set(full["_missing_keys"])
    assert full["transformer_layers_per_block"] == 1 and full["num_attention_heads"] is None
    assert k22.make_arch22(dict(k22.UNET_CONFIG_2_2, addition_embed_type="image_hint", in_channels=8)).hint_channels == 3     # from the json
    assert k22.make_arch22(dict(k22.UNET_CONFIG_2_2, in_channels=9)).inpainting
    for bad in (dict(attention_head_dim=8), dict(resnet_time_scale_shift="default"), dict(act_fn="gelu"), dict(encoder_hid_dim_type="text_proj"),
                dict(down_block_types=("CrossAttnDownBlock2D",) * 4), dict(in_channels=5)):
        with pytest.raises(NotImplementedError):
            k22.make_arch22(dict(k22.UNET_CONFIG_2_2, **bad))


def test_aux_engine_dtypes_one_rule_for_both_drivers():
    """the engine built to reproduce the reference's fp32 images ("f16x3") gets an fp32 prior and an fp32 MoVQ in BOTH drivers; round 6:
    beside "f16x2" (activation operands at fp16 anyway) the prior and the MoVQ run in fp16 like the reference under use_fp16 (image-level bound
    measured on the GPU: tests/test_pipeline_gpu.py); 16-bit engines get an fp16 MoVQ (also bf16: a bf16 decode is 24 grey levels off); an
    explicit movq_dtype wins."""
    import inspect
    import torch
    from kandinsky2_amd import pipeline, pipeline22
    f = pipeline.aux_engine_dtypes
    assert f("f16x3") == (torch.float32, torch.float32) and f("f16x2") == (torch.float16, torch.float16)
    assert f("f16x2", torch.float32) == (torch.float16, torch.float32)
    assert f(torch.float32) == (torch.float32, torch.float32)
    assert f(torch.bfloat16) == (torch.bfloat16, torch.float16) and f(torch.float16) == (torch.float16, torch.float16)
    assert f("f16x3", torch.float16) == (torch.float32, torch.float16) and f(torch.bfloat16, torch.float32)[1] == torch.float32
    # both drivers go through it (no second copy of the rule)
    assert "aux_engine_dtypes" in inspect.getsource(pipeline22.Kandinsky2_2HIP.__init__)
    assert "aux_engine_dtypes" in inspect.getsource(pipeline.Kandinsky2_1HIP.__init__)
