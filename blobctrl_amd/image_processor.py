"""Host-side image pre/post-processing either side of the pipeline (no arithmetic worth a kernel: a few MB per edit on the host).

  * VaeImageProcessor      - D/image_processor.py:469-600 (preprocess) and :602-680 (postprocess) for the inputs the pipeline accepts:
                             PIL images (lanczos resize to (height, width) rounded down to a multiple of the VAE scale factor, THEN
                             RGB conversion, [0,1] -> [-1,1]), numpy arrays, tensors; outputs "latent" | "pt" | "np" | "pil".
  * Dinov2ImageProcessor   - what `AutoImageProcessor.from_pretrained(dinov2_path)` resolves to for facebook/dinov2-* (transformers'
                             BitImageProcessor, called at pipeline_blobnet.py:696): RGB, bicubic resize of the SHORTEST edge to 256,
                             centre crop 224 x 224, 1/255, ImageNet mean / std.  Pinned against the installed BitImageProcessor
                             (tests/golden/image_processors.npz).
"""
from typing import List, Optional, Union

import numpy as np
import torch


def _pil():
    from PIL import Image
    return Image


class VaeImageProcessor:
    def __init__(self, vae_scale_factor: int = 8, do_resize: bool = True, resample: str = "lanczos", do_normalize: bool = True,
                 do_convert_rgb: bool = False, vae_latent_channels: int = 4):
        self.config = type("Config", (), dict(vae_scale_factor=vae_scale_factor, do_resize=do_resize, resample=resample,
                                              do_normalize=do_normalize, do_convert_rgb=do_convert_rgb))()
        self.vae_latent_channels = vae_latent_channels

    # ---- D/image_processor.py:426-467
    def get_default_height_width(self, image, height=None, width=None):
        Image = _pil()
        if height is None:
            height = image.height if isinstance(image, Image.Image) else (image.shape[2] if torch.is_tensor(image) else image.shape[1])
        if width is None:
            width = image.width if isinstance(image, Image.Image) else (image.shape[3] if torch.is_tensor(image) else image.shape[2])
        f = self.config.vae_scale_factor
        return height - height % f, width - width % f

    def preprocess(self, image, height: Optional[int] = None, width: Optional[int] = None) -> torch.Tensor:
        Image = _pil()
        if not isinstance(image, list):
            image = [image]
        if not all(isinstance(i, (Image.Image, np.ndarray, torch.Tensor)) for i in image) or not image:
            raise ValueError("Input is in incorrect format. Currently, we only support PIL.Image.Image, np.ndarray, torch.Tensor")
        if isinstance(image[0], Image.Image):
            if self.config.do_resize:
                height, width = self.get_default_height_width(image[0], height, width)
                rs = {"lanczos": Image.LANCZOS, "bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC, "nearest": Image.NEAREST}
                image = [i.resize((width, height), resample=rs[self.config.resample]) for i in image]
            if self.config.do_convert_rgb:
                image = [i.convert("RGB") for i in image]
            arr = np.stack([np.array(i).astype(np.float32) / 255.0 for i in image], axis=0)
            if arr.ndim == 3:
                arr = arr[..., None]
            x = torch.from_numpy(arr.transpose(0, 3, 1, 2))
        elif isinstance(image[0], np.ndarray):
            arr = np.concatenate(image, axis=0) if image[0].ndim == 4 else np.stack(image, axis=0)
            if arr.ndim == 3:
                arr = arr[..., None]
            x = torch.from_numpy(arr.transpose(0, 3, 1, 2))
            height, width = self.get_default_height_width(x, height, width)
            if self.config.do_resize:
                x = torch.nn.functional.interpolate(x, size=(height, width))
        else:
            x = torch.cat(image, dim=0) if image[0].ndim == 4 else torch.stack(image, dim=0)
            if x.shape[1] == self.vae_latent_channels:            # already latents
                return x
            height, width = self.get_default_height_width(x, height, width)
            if self.config.do_resize:
                x = torch.nn.functional.interpolate(x, size=(height, width))
        if self.config.do_normalize and not (x.min() < 0):         # tensors already in [-1, 1] are passed through (:580-587)
            x = 2.0 * x - 1.0
        return x

    def postprocess(self, image: torch.Tensor, output_type: str = "pil", do_denormalize: Optional[List[bool]] = None):
        if output_type not in ("latent", "pt", "np", "pil"):
            output_type = "np"                                       # (:640-646: deprecation fallback)
        if output_type == "latent":
            return image
        if do_denormalize is None:
            do_denormalize = [self.config.do_normalize] * image.shape[0]
        image = torch.stack([(image[i] / 2 + 0.5).clamp(0, 1) if do_denormalize[i] else image[i] for i in range(image.shape[0])])
        if output_type == "pt":
            return image
        arr = image.cpu().permute(0, 2, 3, 1).float().numpy()
        if output_type == "np":
            return arr
        Image = _pil()
        u8 = (arr * 255).round().astype("uint8")
        if u8.shape[-1] == 1:
            return [Image.fromarray(a.squeeze(), mode="L") for a in u8]
        return [Image.fromarray(a) for a in u8]


class Dinov2ImageProcessor:
    """`dinov2_processor.preprocess(images=..., do_resize=True, return_tensors="pt", do_convert_rgb=True)` (pipe:696) for the
    facebook/dinov2 preprocessor_config.json values; returns an object with `.pixel_values` that is also a mapping (`**inputs`)."""

    def __init__(self, shortest_edge: int = 256, crop_size: int = 224, image_mean=(0.485, 0.456, 0.406),
                 image_std=(0.229, 0.224, 0.225), rescale_factor: float = 1 / 255):
        self.shortest_edge, self.crop_size = shortest_edge, crop_size
        self.image_mean, self.image_std, self.rescale_factor = tuple(image_mean), tuple(image_std), rescale_factor

    @classmethod
    def from_pretrained(cls, path, **_ignored):
        import json
        import os
        with open(os.path.join(path, "preprocessor_config.json")) as f:
            c = json.load(f)
        size = c.get("size", {"shortest_edge": 256})
        crop = c.get("crop_size", {"height": 224, "width": 224})
        return cls(size["shortest_edge"] if isinstance(size, dict) else int(size), crop["height"] if isinstance(crop, dict) else int(crop),
                   c.get("image_mean", (0.485, 0.456, 0.406)), c.get("image_std", (0.229, 0.224, 0.225)), c.get("rescale_factor", 1 / 255))

    def _one(self, img) -> np.ndarray:
        Image = _pil()
        if isinstance(img, np.ndarray):
            img = Image.fromarray(img)
        img = img.convert("RGB")
        w, h = img.size
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = self.shortest_edge, int(self.shortest_edge * long / short)
        nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        img = img.resize((nw, nh), resample=Image.BICUBIC)
        top, left = (nh - self.crop_size) // 2, (nw - self.crop_size) // 2
        arr = np.array(img)[top:top + self.crop_size, left:left + self.crop_size].astype(np.float32) * np.float32(self.rescale_factor)
        arr = (arr - np.array(self.image_mean, np.float32)) / np.array(self.image_std, np.float32)
        return arr.transpose(2, 0, 1)

    def preprocess(self, images, do_resize=True, return_tensors="pt", do_convert_rgb=True, **_ignored):
        if not isinstance(images, (list, tuple)):
            images = [images]
        px = torch.from_numpy(np.stack([self._one(i) for i in images], 0))
        return _BatchFeature(pixel_values=px)

    __call__ = preprocess


class _BatchFeature(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, device=None, dtype=None):
        return _BatchFeature({k: (v.to(device=device, dtype=dtype) if torch.is_tensor(v) and v.is_floating_point() else
                                  (v.to(device) if torch.is_tensor(v) else v)) for k, v in self.items()})
