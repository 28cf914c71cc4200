"""The construction path of scripts/blobctrl_inference.py:220-279 (`construct_pipeline`) on this package's classes, from a tiny
on-disk model tree: UNet from `<sd15>/unet` + the conv_in 4 -> 5 surgery through `unet.conv_in`, BlobNet, DINOv2,
`StableDiffusionBlobNetPipeline.from_pretrained(sd15, unet=, blobnet=, torch_dtype=, dinov2_processor=, dinov2=)`,
`load_lora_weights`, `set_adapters`, scheduler swap through `from_config`.  No GPU: everything up to packing is host work; running an
edit without a GPU raises (there is no CPU fallback).  The GPU twin is tests/test_pipeline_construct_gpu.py."""
import os

import numpy as np
import pytest
import torch

from tests.common import CLIP_TEST_WORDS, g, tiny_clip_vocab, write_model_tree


def construct_pipeline(paths, device, weight_dtype=torch.float16):
    """The statements of inf:220-279, with this package's classes in place of diffusers' / transformers' / blobctrl's."""
    from blobctrl_amd.dinov2 import Dinov2Model
    from blobctrl_amd.image_processor import Dinov2ImageProcessor as AutoImageProcessor
    from blobctrl_amd.modules import BlobNetModel, UNet2DConditionModel
    from blobctrl_amd.pipeline import StableDiffusionBlobNetPipeline
    from blobctrl_amd.schedulers import UniPCMultistepScheduler
    unet = UNet2DConditionModel.from_pretrained(paths["sd15"], subfolder="unet", device=device)
    with torch.no_grad():
        initial_input_channels = unet.config.in_channels
        new_conv_in = torch.nn.Conv2d(initial_input_channels + 1, unet.conv_in.out_channels, kernel_size=3, stride=1, padding=1,
                                      bias=unet.conv_in.bias is not None, dtype=unet.dtype, device=unet.device)
        assert new_conv_in.weight.dtype == torch.float32        # (the surgery happens in the checkpoint's fp32, as in the reference)
        new_conv_in.weight.zero_()
        new_conv_in.weight[:, :initial_input_channels].copy_(unet.conv_in.weight)
        if unet.conv_in.bias is not None:
            new_conv_in.bias.copy_(unet.conv_in.bias)
        unet.conv_in = new_conv_in
    blobnet = BlobNetModel.from_pretrained(paths["blobnet"], ignore_mismatched_sizes=True, device=device)
    dinov2_processor = AutoImageProcessor.from_pretrained(paths["dinov2"])
    dinov2 = Dinov2Model.from_pretrained(paths["dinov2"], device=device).to(device)
    pipeline = StableDiffusionBlobNetPipeline.from_pretrained(paths["sd15"], unet=unet, blobnet=blobnet, torch_dtype=weight_dtype,
                                                              dinov2_processor=dinov2_processor, dinov2=dinov2)
    pipeline.load_lora_weights(paths["unet_lora"], adapter_name="default")
    pipeline.set_adapters(["default"])
    pipeline.scheduler = UniPCMultistepScheduler.from_config(pipeline.scheduler.config)
    pipeline.to(device)
    pipeline.set_progress_bar_config(leave=False)
    return pipeline


def expected_unet_state_dict(pieces, adapter_weight=1.0):
    from blobctrl_amd import checkpoint as ck
    from blobctrl_amd.weights import merge_lora
    sd5 = ck.expand_conv_in(pieces["unet4"], 1)
    return merge_lora(sd5, {k[len("unet."):]: v for k, v in pieces["lora"].items()}, adapter_scale=adapter_weight)


def test_construct_pipeline_like_the_script_on_the_host(tmp_path):
    from blobctrl_amd import _lib
    from blobctrl_amd.schedulers import DDIMScheduler, PNDMScheduler, UniPCMultistepScheduler, scheduler_from_config_dir
    paths, pieces = write_model_tree(tmp_path)
    assert isinstance(scheduler_from_config_dir(os.path.join(paths["sd15"], "scheduler")), PNDMScheduler)
    pipe = construct_pipeline(paths, "cpu")
    assert isinstance(pipe.scheduler, UniPCMultistepScheduler) and pipe.scheduler.config.beta_end == 0.012
    assert pipe.unet.config.in_channels == 4 and pipe.unet.trunk_config.in_channels == 5          # inf:233-249 leaves config at 4
    assert pipe.get_active_adapters() == ["default"]
    want = expected_unet_state_dict(pieces)
    got = pipe.unet.effective_state_dict()
    assert list(got) == list(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    # adapter weights / unloading re-merge from the base weights
    pipe.set_adapters(["default"], [0.5])
    half = pipe.unet.effective_state_dict()
    want_half = expected_unet_state_dict(pieces, 0.5)
    assert all(torch.allclose(half[k], want_half[k], atol=1e-7) for k in want_half)
    pipe.unload_lora_weights()
    base5 = pipe.unet.effective_state_dict()
    assert torch.equal(base5["conv_in.weight"][:, :4], pieces["unet4"]["conv_in.weight"]) and float(base5["conv_in.weight"][:, 4].abs().max()) == 0
    with pytest.raises(ValueError, match="not in the list of present adapters"):
        pipe.set_adapters(["nope"])
    with pytest.raises(KeyError, match="LoRA targets not present"):
        pipe.load_lora_weights({"unet.no.such.layer.lora_A.weight": g(1, 2, 4), "unet.no.such.layer.lora_B.weight": g(2, 4, 2)})
    # the PNDM holder only carries the configuration
    with pytest.raises(NotImplementedError, match="PNDM is not tabulated"):
        scheduler_from_config_dir(os.path.join(paths["sd15"], "scheduler")).set_timesteps(10)
    assert isinstance(DDIMScheduler.from_config(pipe.scheduler.config), DDIMScheduler)
    # no GPU here: running an edit must fail loudly, never fall back to a CPU path
    if not torch.cuda.is_available():
        with pytest.raises((_lib.BlobCtrlHipError, RuntimeError, AssertionError)):
            pipe(prompt="a frog", fg_image=torch.zeros(1, 3, 64, 64), bg_image=torch.zeros(1, 3, 64, 64),
                 gs_score=torch.zeros(1, 2, 8, 8), height=64, width=64, num_inference_steps=2, output_type="latent")


def test_scheduler_configuration_reaches_the_engine_tables():
    """`X.from_config(cfg_with_other_betas)` must denoise with ITS alphas (the engine used to tabulate SD-1.5 defaults)."""
    from blobctrl_amd.pipeline import BlobCtrlEngine
    from blobctrl_amd.schedulers import DDIMScheduler
    s = DDIMScheduler(beta_start=0.0001, beta_end=0.02, num_train_timesteps=500)
    assert s.table_params() == (500, 0.0001, 0.02)
    eng = BlobCtrlEngine.__new__(BlobCtrlEngine)
    eng._sched_cache, eng.scheduler_kind, eng.scheduler_params = {}, "unipc", (1000, 0.00085, 0.012)
    eng.set_scheduler(s.kind, s.table_params())
    tab = eng._scheduler_table(10)
    s.set_timesteps(10)
    assert torch.equal(tab.timesteps, s.table_impl.timesteps) and torch.equal(tab.table(), s.table_impl.table())
    eng.set_scheduler("ddim", (1000, 0.00085, 0.012))
    assert not torch.equal(eng._scheduler_table(10).table(), tab.table())
    with pytest.raises(NotImplementedError):
        eng.set_scheduler("pndm")


def test_clip_tokenizer_matches_transformers_on_a_synthetic_vocabulary(tmp_path):
    """blobctrl_amd.clip_tokenizer (openai/CLIP's published BPE) against the installed transformers CLIPTokenizer - the class the
    reference loads from `<sd15>/tokenizer` (pipe:206-243) - on the same vocab.json / merges.txt."""
    tr = pytest.importorskip("transformers")
    from blobctrl_amd.clip_tokenizer import CLIPTokenizer
    paths, _ = write_model_tree(tmp_path, lora=False)
    d = os.path.join(paths["sd15"], "tokenizer")
    mine = CLIPTokenizer.from_pretrained(paths["sd15"], subfolder="tokenizer")
    try:
        ref = tr.CLIPTokenizer.from_pretrained(d)
    except Exception as e:                                       # pragma: no cover
        pytest.skip(f"transformers could not load the synthetic tokenizer: {e}")
    prompts = ["a frog sits on a rock in a pond, with a top hat beside it.", "The butterflies and flowers", "", "blurry, low quality",
               "  a   hat  ", " ".join(CLIP_TEST_WORDS * 6)]
    for p in prompts:
        a = mine(p, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        b = ref(p, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        assert a.shape == b.shape == (1, 77) and torch.equal(a, b.to(torch.int64)), p
    a = mine(prompts[:2], padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    b = ref(prompts[:2], padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    assert torch.equal(a, b.to(torch.int64))
    vocab, _ = tiny_clip_vocab()
    assert int(a.max()) < len(vocab)
