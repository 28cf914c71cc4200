"""On-disk weight formats of the reference -> the state dicts the engine packs.

What the reference loads (README.md:115-128, scripts/blobctrl_inference.py:222-280):
  * `<sd15>/unet/{config.json, diffusion_pytorch_model.safetensors}`           UNet2DConditionModel.from_pretrained
    followed by the 4 -> 5 channel `conv_in` surgery of inf:233-249 (new input channel zero-initialised);
  * `<blobnet>/{config.json, diffusion_pytorch_model.safetensors}`             BlobNetModel.from_pretrained
  * `<unet_lora>/pytorch_lora_weights.safetensors`                             pipeline.load_lora_weights (inf:270-273):
    keys filtered by the "unet." prefix and renamed to `<module>.lora_A.weight` / `<module>.lora_B.weight`
    (D/loaders/unet.py:271-340, D/utils/state_dict_utils.py:38-50,141-171); rank = lora_B.shape[1]; scale = alpha / rank
    with the alpha defaults of D/utils/peft_utils.py:150-192 (no alpha keys: alpha = the FIRST module's rank for every
    module).  Linear and Conv2d LoRA pairs (conv_in included) are merged into the base weights once - the engine has no
    adapter path at run time.

Pure host code (no GPU): safetensors files are read with a small reader of the published format (8-byte little-endian
header length, JSON header, raw little-endian tensor bytes), so the converter works without the `safetensors` package.
"""
import json
import os
import struct
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .engine import TrunkConfig
from .weights import merge_lora

_DTYPES = {"F64": (np.float64, torch.float64), "F32": (np.float32, torch.float32), "F16": (np.float16, torch.float16),
           "I64": (np.int64, torch.int64), "I32": (np.int32, torch.int32), "U8": (np.uint8, torch.uint8),
           "BOOL": (np.bool_, torch.bool)}


def read_safetensors(path: str) -> "OrderedDict[str, torch.Tensor]":
    """All tensors of one .safetensors file (BF16 is widened to fp32)."""
    with open(path, "rb") as f:
        raw = f.read(8)
        if len(raw) != 8:
            raise ValueError(f"{path}: not a safetensors file (shorter than its 8-byte header length)")
        (n,) = struct.unpack("<Q", raw)
        if n <= 0 or n > os.path.getsize(path) - 8:
            raise ValueError(f"{path}: bad safetensors header length {n}")
        header = json.loads(f.read(n).decode("utf-8"))
        data = f.read()
    out = OrderedDict()
    for name, meta in header.items():
        if name == "__metadata__":
            continue
        b, e = meta["data_offsets"]
        shape = tuple(meta["shape"])
        dt = meta["dtype"]
        buf = data[b:e]
        if dt == "BF16":
            u16 = np.frombuffer(buf, dtype="<u2").astype(np.uint32) << 16
            t = torch.from_numpy(u16.view(np.float32).copy()).reshape(shape)
        elif dt in _DTYPES:
            npdt, _ = _DTYPES[dt]
            arr = np.frombuffer(buf, dtype=np.dtype(npdt).newbyteorder("<"))
            if arr.size != int(np.prod(shape, dtype=np.int64)):
                raise ValueError(f"{path}: tensor {name} has {arr.size} elements, header says {shape}")
            t = torch.from_numpy(arr.astype(npdt, copy=True)).reshape(shape)
        else:
            raise ValueError(f"{path}: unsupported safetensors dtype {dt} for {name}")
        out[name] = t
    return out


def write_safetensors(path: str, tensors: Dict[str, torch.Tensor], metadata: Optional[Dict[str, str]] = None) -> None:
    """Minimal writer (fp32 / fp16 / integer tensors) - used by the tests and by `export_packed_inputs`."""
    names = {v[1]: k for k, v in _DTYPES.items()}
    header, blobs, off = {}, [], 0
    if metadata:
        header["__metadata__"] = metadata
    for k, t in tensors.items():
        t = t.detach().cpu().contiguous()
        if t.dtype not in names:
            raise ValueError(f"unsupported dtype {t.dtype} for {k}")
        raw = t.numpy().tobytes()
        header[k] = {"dtype": names[t.dtype], "shape": list(t.shape), "data_offsets": [off, off + len(raw)]}
        blobs.append(raw)
        off += len(raw)
    hj = json.dumps(header, separators=(",", ":")).encode("utf-8")
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for b in blobs:
            f.write(b)


def _model_file(path: str) -> str:
    if os.path.isdir(path):
        for name in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors",
                     "pytorch_lora_weights.safetensors", "model.safetensors"):
            p = os.path.join(path, name)
            if os.path.exists(p):
                return p
        raise FileNotFoundError(f"no .safetensors model file under {path}")
    return path


def _config(path: str) -> dict:
    d = path if os.path.isdir(path) else os.path.dirname(path)
    p = os.path.join(d, "config.json")
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f)
    return {}


def expand_conv_in(sd: Dict[str, torch.Tensor], extra_channels: int = 1) -> Dict[str, torch.Tensor]:
    """inf:233-249: `conv_in` grows from C to C + extra input channels; the new channels start at zero, the bias is kept."""
    w = sd["conv_in.weight"]
    new = torch.zeros(w.shape[0], w.shape[1] + extra_channels, *w.shape[2:], dtype=w.dtype)
    new[:, : w.shape[1]] = w
    out = OrderedDict(sd)
    out["conv_in.weight"] = new
    return out


# D/utils/state_dict_utils.py:38-50 (UNET_TO_DIFFUSERS): first matching pattern wins, in this order
_UNET_LORA_RENAMES = (
    (".to_out_lora.up", ".to_out.0.lora_B"), (".to_out_lora.down", ".to_out.0.lora_A"),
    (".to_q_lora.down", ".to_q.lora_A"), (".to_q_lora.up", ".to_q.lora_B"),
    (".to_k_lora.down", ".to_k.lora_A"), (".to_k_lora.up", ".to_k.lora_B"),
    (".to_v_lora.down", ".to_v.lora_A"), (".to_v_lora.up", ".to_v.lora_B"),
    (".lora.up", ".lora_B"), (".lora.down", ".lora_A"),
    (".to_out.lora_magnitude_vector", ".to_out.0.lora_magnitude_vector"),
)


def convert_unet_lora_key(k: str) -> str:
    """One key through `convert_unet_state_dict_to_peft` (state_dict_utils.py:141-171, 248-253)."""
    k = k.replace(".processor.", ".")                              # KEYS_TO_ALWAYS_REPLACE
    for pat, new in _UNET_LORA_RENAMES:
        if pat in k:
            return k.replace(pat, new)
    return k


def load_lora(path: str, unet_identifier_key: str = "unet") -> Tuple[Dict[str, torch.Tensor], Dict[str, float]]:
    """-> (lora, alphas): `lora` has `<module>.lora_A.weight` / `<module>.lora_B.weight`; `alphas[<module>]` is the effective
    lora_alpha of each module, so `merge_lora(sd, lora, alphas)` applies W + (alpha / r) * B @ A exactly like the peft
    adapter the reference injects."""
    return lora_from_state_dict(read_safetensors(_model_file(path)), unet_identifier_key, what=path)


def lora_from_state_dict(raw: Dict[str, torch.Tensor], unet_identifier_key: str = "unet", what: str = "state dict"):
    """The same conversion for an in-memory state dict (`pipeline.load_lora_weights(dict)`, lora_pipeline.py:94-100)."""
    path = what
    pre = unet_identifier_key + "."
    keys = [k for k in raw if k.startswith(pre)]
    sd = OrderedDict((k[len(pre):], raw[k]) for k in keys) if keys else raw           # unet.py:289-301
    net_alpha = {k[: -len(".alpha")]: float(v) for k, v in sd.items() if k.endswith(".alpha")}
    sd = OrderedDict((k, v) for k, v in sd.items() if not k.endswith(".alpha"))
    lora = OrderedDict((convert_unet_lora_key(k), v.float()) for k, v in sd.items())
    if any("lora_magnitude_vector" in k for k in lora):
        raise NotImplementedError("DoRA adapters (lora_magnitude_vector) are not supported")
    if not any(".lora_A." in k or ".lora_B." in k for k in lora):
        raise ValueError(f"{path}: no LoRA tensors found (expected lora_A / lora_B or lora.down / lora.up keys)")
    ranks = OrderedDict((k[: -len(".lora_B.weight")], v.shape[1]) for k, v in lora.items() if k.endswith(".lora_B.weight"))
    for m in ranks:
        if m + ".lora_A.weight" not in lora:
            raise ValueError(f"{path}: {m} has lora_B but no lora_A")
    # peft_utils.py:150-178: default alpha = first rank; with alpha keys: the most common alpha, others per module
    default_alpha = float(next(iter(ranks.values())))
    # alpha entries are keyed like the down weight they belong to ("<module>.lora.down.weight.alpha") or by the bare module
    conv_alpha = {}
    for k, a in net_alpha.items():
        ck = convert_unet_lora_key(k)
        conv_alpha[ck.split(".lora_A.")[0] if ".lora_A." in ck else ck] = a
    if conv_alpha:
        vals = list(conv_alpha.values())
        default_alpha = max(set(vals), key=vals.count)
    alphas = {m: conv_alpha.get(m, default_alpha) for m in ranks}
    return lora, alphas


def load_unet(path: str, extra_in_channels: int = 1, lora_path: Optional[str] = None, lora_scale: float = 1.0):
    """-> (state_dict, TrunkConfig) of the patched SD-1.5 UNet: file weights, conv_in surgery, LoRA merged (in that order:
    the released LoRA was trained on the 5-channel conv_in, inf:233-273)."""
    cfg = _config(path)
    sd = OrderedDict((k, v.float()) for k, v in read_safetensors(_model_file(path)).items())
    if extra_in_channels:
        sd = expand_conv_in(sd, extra_in_channels)
    if lora_path is not None:
        lora, alphas = load_lora(lora_path)
        missing = [m for m in alphas if m + ".weight" not in sd]
        if missing:
            raise KeyError(f"LoRA targets not present in the UNet: {missing[:4]}{' ...' if len(missing) > 4 else ''}")
        sd = merge_lora(sd, lora, alphas, adapter_scale=lora_scale)
    boc = tuple(cfg.get("block_out_channels", (320, 640, 1280, 1280)))
    heads = cfg.get("attention_head_dim", 8)                     # SD-1.5: this field holds the number of heads
    tc = TrunkConfig(in_channels=sd["conv_in.weight"].shape[1], block_out_channels=boc,
                     num_heads=heads if isinstance(heads, int) else heads[0], norm_num_groups=cfg.get("norm_num_groups", 32),
                     cross_attention_dim=cfg.get("cross_attention_dim", 768), out_channels=sd["conv_out.weight"].shape[0],
                     is_blobnet=False)
    return sd, tc


def load_blobnet(path: str):
    """-> (state_dict, TrunkConfig) of BlobNetModel (blobctrl/models/blobnet.py:150-260; conv_in takes in_channels +
    conditioning_channels = 4 + 1 + 1024)."""
    cfg = _config(path)
    sd = OrderedDict((k, v.float()) for k, v in read_safetensors(_model_file(path)).items())
    boc = tuple(cfg.get("block_out_channels", (320, 640, 1280, 1280)))
    heads = cfg.get("attention_head_dim", 8)
    cin = sd["conv_in.weight"].shape[1]
    want = cfg.get("in_channels", 4) + cfg.get("conditioning_channels", cin - 4)
    if cin != want:
        raise ValueError(f"BlobNet conv_in has {cin} input channels, config says {want}")
    tc = TrunkConfig(in_channels=cin, block_out_channels=boc, num_heads=heads if isinstance(heads, int) else heads[0],
                     norm_num_groups=cfg.get("norm_num_groups", 32), cross_attention_dim=None, out_channels=0, is_blobnet=True)
    return sd, tc
