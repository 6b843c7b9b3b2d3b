"""Frozen CLIP text encoder.  CPU: the oracle is PINNED to the golden fixture produced by
transformers' own CLIPTextModel (and to a live transformers instance when importable); key inventory.  GPU: HIP vs the
bf16-mirroring oracle and vs the transformers golden output."""
import os

import numpy as np
import pytest
import torch

from aqualora_amd.clip import SD15_CLIP, clip_keys
from oracle.clip_oracle import clip_text_forward

HERE = os.path.dirname(os.path.abspath(__file__))
TINY_CLIP = dict(vocab_size=320, hidden_size=64, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                 max_position_embeddings=77, layer_norm_eps=1e-5)


def _golden():
    z = np.load(os.path.join(HERE, "golden", "clip_text_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    return sd, torch.from_numpy(z["ids"]), torch.from_numpy(z["out"])


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max())


def test_clip_oracle_is_pinned_to_transformers_golden():
    sd, ids, want = _golden()
    got = clip_text_forward(sd, TINY_CLIP, ids)
    assert got.shape == want.shape == (3, 77, 64)
    assert _rel(got, want) < 2e-5, _rel(got, want)
    # causal: the hidden state at position p must not depend on tokens after p
    ids2 = ids.clone()
    ids2[:, 40:] = 5
    got2 = clip_text_forward(sd, TINY_CLIP, ids2)
    assert torch.equal(got2[:, :40], got[:, :40]) and not torch.equal(got2[:, 40:], got[:, 40:])


def test_clip_oracle_vs_live_transformers():
    tr = pytest.importorskip("transformers")
    torch.manual_seed(7)
    cfg = tr.CLIPTextConfig(hidden_act="quick_gelu", bos_token_id=0, eos_token_id=319, pad_token_id=1, **TINY_CLIP)
    model = tr.CLIPTextModel(cfg).eval()
    ids = torch.randint(0, 320, (2, 33))     # a shorter sequence than max_position_embeddings
    with torch.no_grad():
        want = model(ids)[0]
    got = clip_text_forward(model.state_dict(), TINY_CLIP, ids)
    assert _rel(got, want) < 2e-5


def test_clip_key_inventory_matches_sd15_text_tower():
    keys = clip_keys()
    assert len(keys) == 2 + 12 * 16 + 2
    n = sum(torch.Size(s).numel() for s in keys.values())
    assert n == 123_060_480, n        # CLIP ViT-L/14 text tower: 123.06 M parameters
    sd, _, _ = _golden()
    assert set(clip_keys(TINY_CLIP)) == {k[len("text_model."):] if k.startswith("text_model.") else k for k in sd}


@pytest.mark.gpu
def test_clip_hip_vs_oracle_and_transformers_golden():
    from aqualora_amd.clip import CLIPTextModel
    sd, ids, want = _golden()
    model = CLIPTextModel(sd, TINY_CLIP, "cuda")
    got = model(ids.cuda())
    assert got.shape == (3, 77, 64) and got.dtype == torch.bfloat16
    mirror = clip_text_forward(sd, TINY_CLIP, ids, bf16=True)
    assert _rel(got, mirror) < 2e-2, _rel(got, mirror)           # same bf16 storage points
    assert _rel(got, want) < 4e-2, _rel(got, want)               # transformers fp32
    assert torch.equal(model(ids.cuda()), got)                   # deterministic


@pytest.mark.gpu
def test_clip_full_size_runs_and_matches_oracle():
    """SD-1.5 text tower (12 x 768, 12 heads of 64) with synthetic weights, batch 2."""
    from aqualora_amd import synth
    from aqualora_amd.clip import CLIPTextModel
    sd = {}
    for k, shp in clip_keys().items():
        if "layer_norm" in k:
            sd[k] = torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
        elif k.endswith("bias"):
            sd[k] = synth.normal("clip." + k, shp, 0.02, 3)
        elif "embedding" in k:
            sd[k] = synth.normal("clip." + k, shp, 0.5, 3)
        else:
            sd[k] = synth.normal("clip." + k, shp, shp[1] ** -0.5, 3)
    ids = synth.randint("clip.ids", (2, 77), 49408, 3)
    got = CLIPTextModel(sd, SD15_CLIP, "cuda")(ids.cuda())
    want = clip_text_forward(sd, SD15_CLIP, ids, bf16=True)
    assert got.shape == (2, 77, 768)
    assert _rel(got, want) < 3e-2, _rel(got, want)
