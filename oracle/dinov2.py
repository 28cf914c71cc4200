"""Oracle (TEST INFRASTRUCTURE): functional fp32 restatement of `transformers.Dinov2Model(...).pooler_output`.

The arithmetic lives in a THIRD-PARTY dependency that is not under /root/reference: `transformers`
(pinned 4.49.0 in /root/reference/requirements.txt:2; 5.x installed in the build container), model file
`transformers/models/dinov2/modeling_dinov2.py`.  Reference call site: blobctrl/pipelines/pipeline_blobnet.py:690-703
(`self.dinov2(**dinov2_input).pooler_output`, ViT-L/14 on a 224x224 crop).  Published algorithm restated:
patch-embed conv (kernel = stride = patch) -> [CLS] + tokens -> + position embeddings (bicubic-resized from the
training grid, align_corners=False) -> L x { x + ls1 * Attn(LN(x)) ; x + ls2 * MLP(LN(x)) } with exact-erf GELU
-> final LayerNorm -> token 0.  Pinned by tests against the installed transformers implementation on a small
random Dinov2Config (tests/test_oracle_golden.py::test_dinov2_oracle_vs_transformers and tests/golden/dinov2_tiny.npz).
"""
import torch
import torch.nn.functional as F


def interpolated_position_embeddings(pos, grid_h, grid_w):
    """Dinov2Embeddings.interpolate_pos_encoding: pos [1, 1+P, D] -> [1, 1+grid_h*grid_w, D]."""
    n = pos.shape[1] - 1
    if n == grid_h * grid_w and grid_h == grid_w:
        return pos
    side = int(round(n ** 0.5))
    dim = pos.shape[-1]
    patch = pos[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2)
    patch = F.interpolate(patch.float(), size=(grid_h, grid_w), mode="bicubic", align_corners=False)
    patch = patch.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat([pos[:, :1], patch], dim=1)


def dinov2_pooled(sd, pixel_values, num_heads, patch, eps=1e-6):
    """pixel_values [B,3,H,W] fp32 -> pooled CLS [B, D]."""
    B, _, H, W = pixel_values.shape
    x = F.conv2d(pixel_values, sd["embeddings.patch_embeddings.projection.weight"],
                 sd["embeddings.patch_embeddings.projection.bias"], stride=patch)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd["embeddings.cls_token"].expand(B, -1, -1), x], dim=1)
    x = x + interpolated_position_embeddings(sd["embeddings.position_embeddings"], H // patch, W // patch)
    D = x.shape[-1]
    d = D // num_heads
    i = 0
    while f"encoder.layer.{i}.norm1.weight" in sd:
        p = f"encoder.layer.{i}."
        n = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
        q = F.linear(n, sd[p + "attention.attention.query.weight"], sd[p + "attention.attention.query.bias"])
        k = F.linear(n, sd[p + "attention.attention.key.weight"], sd[p + "attention.attention.key.bias"])
        v = F.linear(n, sd[p + "attention.attention.value.weight"], sd[p + "attention.attention.value.bias"])
        q = q.view(B, -1, num_heads, d).transpose(1, 2)
        k = k.view(B, -1, num_heads, d).transpose(1, 2)
        v = v.view(B, -1, num_heads, d).transpose(1, 2)
        a = torch.softmax(torch.matmul(q, k.transpose(2, 3)) * (d ** -0.5), dim=-1)
        a = torch.matmul(a, v).transpose(1, 2).reshape(B, -1, D)
        a = F.linear(a, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
        x = a * sd[p + "layer_scale1.lambda1"] + x
        n = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
        m = F.gelu(F.linear(n, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        m = F.linear(m, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        x = m * sd[p + "layer_scale2.lambda1"] + x
        i += 1
    x = F.layer_norm(x, (D,), sd["layernorm.weight"], sd["layernorm.bias"], eps)
    return x[:, 0]
