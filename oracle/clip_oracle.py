"""CPU oracle for the frozen CLIP text encoder (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it).  Plain-PyTorch restatement of transformers' ``CLIPTextModel`` forward as the reference
calls it (``text_encoder(batch["input_ids"])[0]``, train/ppft_train.py:1014-1019; requirements.txt pins
transformers==4.26.1): token + position embedding, pre-LN layers with causal self-attention (q scaled by d^-1/2) and a
quick_gelu MLP, final LayerNorm.  PINNED: tests/test_clip.py checks it against tests/golden/clip_text_tiny.npz, which
tests/golden/make_clip_golden.py produced by running transformers' own CLIPTextModel (the package is installed in this image;
it is not part of /root/reference), and, when transformers is importable, against a live instance as well."""
import torch
import torch.nn.functional as F


def _rb(t, on):
    return t.to(torch.bfloat16).float() if on else t


def clip_text_forward(sd, cfg, input_ids, bf16=False):
    sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): _rb(v.detach().float().cpu(), bf16)
          for k, v in sd.items() if not k.endswith("position_ids")}
    r = lambda t: _rb(t, bf16)  # noqa: E731
    B, N = input_ids.shape
    Hd, nh, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
    d = Hd // nh
    # the HIP path keeps the embedding tables in fp32 and rounds their sum once
    h = r(sd["embeddings.token_embedding.weight"][input_ids.long()] + sd["embeddings.position_embedding.weight"][:N][None])
    mask = torch.full((N, N), float("-inf")).triu(1)
    for i in range(cfg["num_hidden_layers"]):
        p = f"encoder.layers.{i}."
        x = r(F.layer_norm(h, (Hd,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], eps))
        q, k, v = (r(x @ sd[p + f"self_attn.{n}.weight"].t() + sd[p + f"self_attn.{n}.bias"]).view(B, N, nh, d).transpose(1, 2)
                   for n in ("q_proj", "k_proj", "v_proj"))
        a = torch.softmax((q * d ** -0.5) @ k.transpose(-1, -2) + mask, dim=-1) @ v
        a = r(a.transpose(1, 2).reshape(B, N, Hd))
        h = r(a @ sd[p + "self_attn.out_proj.weight"].t() + sd[p + "self_attn.out_proj.bias"] + h)
        x = r(F.layer_norm(h, (Hd,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], eps))
        m = r(x @ sd[p + "mlp.fc1.weight"].t() + sd[p + "mlp.fc1.bias"])
        m = r(m * torch.sigmoid(1.702 * m))
        h = r(m @ sd[p + "mlp.fc2.weight"].t() + sd[p + "mlp.fc2.bias"] + h)
    return r(F.layer_norm(h, (Hd,), sd["final_layer_norm.weight"], sd["final_layer_norm.bias"], eps))
