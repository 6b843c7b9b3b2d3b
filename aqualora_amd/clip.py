"""Frozen CLIP text encoder (transformers ``CLIPTextModel``; ``encoder_hidden_states = text_encoder(input_ids)[0]``,
train/ppft_train.py:1014-1019) on the HIP kernels.

SD-1.5's text tower: 12 pre-LN transformer layers, width 768, 12 heads of 64, 77 positions, causal mask, ``quick_gelu``
MLP (3072), final LayerNorm; the step consumes the last hidden state [B,77,768].  Inference only (frozen in all three
training scripts).  Per layer: LayerNorm (``aql_layernorm_fwd``) -> ONE bf16 GEMM for q|k|v (weights concatenated) ->
``aql_causal_attn_small`` -> out-proj GEMM with the residual in its epilogue -> LayerNorm -> fc1 GEMM ->
``aql_quick_gelu`` -> fc2 GEMM with residual.  The embedding gather is a torch index op (plumbing).

State-dict keys are those of transformers (with or without the ``text_model.`` prefix).  Pinned: tests/golden/
clip_text_tiny.npz holds weights, ids and outputs of transformers' own CLIPTextModel (tests/golden/make_clip_golden.py).
"""
import torch

from . import _lib as L
from . import ops

SD15_CLIP = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                 max_position_embeddings=77, layer_norm_eps=1e-5)


def clip_keys(cfg=SD15_CLIP):
    """{key: shape} of the CLIPTextModel state dict (no ``text_model.`` prefix)."""
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    out = {"embeddings.token_embedding.weight": (cfg["vocab_size"], H),
           "embeddings.position_embedding.weight": (cfg["max_position_embeddings"], H)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out[p + f"self_attn.{n}.weight"], out[p + f"self_attn.{n}.bias"] = (H, H), (H,)
        for n in ("layer_norm1", "layer_norm2"):
            out[p + n + ".weight"], out[p + n + ".bias"] = (H,), (H,)
        out[p + "mlp.fc1.weight"], out[p + "mlp.fc1.bias"] = (I, H), (I,)
        out[p + "mlp.fc2.weight"], out[p + "mlp.fc2.bias"] = (H, I), (H,)
    out["final_layer_norm.weight"], out["final_layer_norm.bias"] = (H,), (H,)
    return out


class CLIPTextModel:
    """``CLIPTextModel(state_dict)(input_ids) -> last_hidden_state`` [B, N, hidden] bf16."""

    def __init__(self, state_dict, cfg=SD15_CLIP, device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise L.AqlError("CLIPTextModel needs an MI355X (cuda device); there is no CPU path")
        self.cfg, self.device = dict(cfg), dev
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in state_dict.items()}
        missing = [k for k in clip_keys(cfg) if k not in sd]
        if missing:
            raise L.AqlError(f"CLIPTextModel: state dict lacks {len(missing)} keys, e.g. {missing[:3]}")
        for k, shp in clip_keys(cfg).items():
            if tuple(sd[k].shape) != tuple(shp):
                raise L.AqlError(f"CLIPTextModel: {k} has shape {tuple(sd[k].shape)}, expected {shp}")
        f = lambda k: sd[k].detach().to(dev).float()  # noqa: E731
        bf = lambda k: f(k).to(torch.bfloat16).contiguous()  # noqa: E731
        self.tok, self.pos = f("embeddings.token_embedding.weight"), f("embeddings.position_embedding.weight")
        self.layers = []
        for i in range(cfg["num_hidden_layers"]):
            p = f"encoder.layers.{i}."
            wqkv = torch.cat([f(p + f"self_attn.{n}.weight") for n in ("q_proj", "k_proj", "v_proj")], 0)
            bqkv = torch.cat([f(p + f"self_attn.{n}.bias") for n in ("q_proj", "k_proj", "v_proj")], 0)
            self.layers.append(dict(
                ln1=(bf(p + "layer_norm1.weight"), bf(p + "layer_norm1.bias")),
                ln2=(bf(p + "layer_norm2.weight"), bf(p + "layer_norm2.bias")),
                qkv=ops.PackedLinear(wqkv, bqkv),
                out=ops.PackedLinear(f(p + "self_attn.out_proj.weight"), f(p + "self_attn.out_proj.bias")),
                fc1=ops.PackedLinear(f(p + "mlp.fc1.weight"), f(p + "mlp.fc1.bias")),
                fc2=ops.PackedLinear(f(p + "mlp.fc2.weight"), f(p + "mlp.fc2.bias"))))
        self.final = (bf("final_layer_norm.weight"), bf("final_layer_norm.bias"))

    @torch.no_grad()
    def __call__(self, input_ids):
        cfg = self.cfg
        B, N = input_ids.shape
        Hd, nh, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
        d = Hd // nh
        ids = input_ids.to(self.device).long()
        h = (self.tok[ids] + self.pos[:N][None]).to(torch.bfloat16).reshape(B * N, Hd).contiguous()
        st = L.stream_ptr()
        for ly in self.layers:
            x = ops.layernorm(h, ly["ln1"][0], ly["ln1"][1], eps)
            qkv = ops.gemm_bf16(x, ly["qkv"].w, ly["qkv"].bias)            # [B*N, 3*Hd]
            a = torch.empty(B * N, Hd, dtype=torch.bfloat16, device=self.device)
            L.call("aql_causal_attn_small", L.ptr(qkv), L.ptr(qkv[:, Hd:]), L.ptr(qkv[:, 2 * Hd:]), 3 * Hd, B, nh, N, d,
                   float(d ** -0.5), L.ptr(a), Hd, st)
            h = ops.gemm_bf16(a, ly["out"].w, ly["out"].bias, residual=h)
            x = ops.layernorm(h, ly["ln2"][0], ly["ln2"][1], eps)
            m = ops.gemm_bf16(x, ly["fc1"].w, ly["fc1"].bias)
            L.call("aql_quick_gelu", L.ptr(m), m.numel(), L.ptr(m), st)
            h = ops.gemm_bf16(m, ly["fc2"].w, ly["fc2"].bias, residual=h)
        out = ops.layernorm(h, self.final[0], self.final[1], eps)
        return out.view(B, N, Hd)
