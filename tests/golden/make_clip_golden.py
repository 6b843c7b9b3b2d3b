"""Generates tests/golden/clip_text_tiny.npz by running transformers' own CLIPTextModel (installed in this image; the
reference pins transformers==4.26.1 and calls ``text_encoder(input_ids)[0]`` at train/ppft_train.py:1014-1019) on a small
seeded configuration.  The fixture holds the state dict, the token ids and the last hidden state.
Usage: python tests/golden/make_clip_golden.py"""
import os

import numpy as np
import torch
from transformers import CLIPTextConfig, CLIPTextModel

TINY_CLIP = dict(vocab_size=320, hidden_size=64, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                 max_position_embeddings=77, layer_norm_eps=1e-5)


def main():
    torch.manual_seed(2048)
    cfg = CLIPTextConfig(hidden_act="quick_gelu", bos_token_id=0, eos_token_id=319, pad_token_id=1, **TINY_CLIP)
    model = CLIPTextModel(cfg).eval()
    with torch.no_grad():   # the default init is tiny (std 0.02): widen it so that every op matters
        for k, v in model.state_dict().items():
            if v.dtype.is_floating_point and v.dim() == 2 and "embedding" not in k:
                v.mul_(3.0)
            if k.endswith("layer_norm1.bias") or k.endswith("layer_norm2.bias") or k.endswith("proj.bias"):
                v.normal_(0, 0.1)
    ids = torch.randint(2, 319, (3, 77))
    ids[:, 0] = 0
    ids[0, 20:] = 319   # padded prompts, as the tokenizer produces them
    ids[1, 50:] = 319
    with torch.no_grad():
        out = model(ids)[0]
    blob = {"ids": ids.numpy().astype(np.int64), "out": out.numpy().astype(np.float32)}
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            blob["sd/" + k] = v.numpy().astype(np.float32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clip_text_tiny.npz")
    np.savez_compressed(path, **blob)
    print(path, os.path.getsize(path), "bytes; out abs-mean", float(out.abs().mean()))


if __name__ == "__main__":
    main()
