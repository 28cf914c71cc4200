"""Oracle (TEST INFRASTRUCTURE): functional fp32 restatement of `transformers.CLIPTextModel(input_ids)[0]` (the prompt embeddings).

The arithmetic lives in a THIRD-PARTY dependency that is not under /root/reference: `transformers` (pinned 4.49.0 in
/root/reference/requirements.txt:2; 5.x installed in the build container), model file `transformers/models/clip/modeling_clip.py`.
Reference call site: blobctrl/pipelines/pipeline_blobnet.py:599-611 (`self.text_encoder(text_input_ids)[0]`, and the clip_skip
branch `final_layer_norm(hidden_states[-(clip_skip+1)])`).  Published algorithm restated: token embedding + learned position
embedding -> L x pre-LN blocks { x + out_proj(causal MHA(LN1 x)) ; x + fc2(quick_gelu(fc1(LN2 x))) } -> final LayerNorm;
quick_gelu(x) = x * sigmoid(1.702 x); attention scale d^-0.5, causal mask (key j visible to query i iff j <= i).
Pinned by tests against the installed transformers implementation on a small random CLIPTextConfig
(tests/golden/clip_text_tiny.npz, made by tools/make_golden.py::golden_clip_text).
"""
import torch
import torch.nn.functional as F


def _strip(sd):
    """Accept both the on-disk layout ("text_model." prefix, transformers 4.x) and the prefix-less one (5.x)."""
    return {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items()}


def clip_text_hidden(sd, input_ids, num_heads, eps=1e-5, clip_skip=None):
    """input_ids [B, T] int64 -> prompt embeddings [B, T, D] fp32."""
    sd = _strip(sd)
    B, T = input_ids.shape
    x = sd["embeddings.token_embedding.weight"][input_ids] + sd["embeddings.position_embedding.weight"][:T]
    D = x.shape[-1]
    d = D // num_heads
    L = 0
    while f"encoder.layers.{L}.layer_norm1.weight" in sd:
        L += 1
    mask = torch.full((T, T), float("-inf")).triu(1)
    hidden = [x]
    for i in range(L):
        p = f"encoder.layers.{i}."
        h = F.layer_norm(x, (D,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], eps)
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * d ** -0.5
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q, k, v = (t.view(B, T, num_heads, d).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, T, D)
        x = x + F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (D,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], eps)
        h = F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        h = h * torch.sigmoid(1.702 * h)
        x = x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        hidden.append(x)
    src = hidden[-1] if clip_skip is None else hidden[-(clip_skip + 1)]
    return F.layer_norm(src, (D,), sd["final_layer_norm.weight"], sd["final_layer_norm.bias"], eps)
