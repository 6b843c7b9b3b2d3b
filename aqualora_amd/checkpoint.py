"""Artefact layout of the PPFT stage (reference train/ppft_train.py:443-471, 1203-1229; resume :626-633).

``pytorch_lora_weights.safetensors``: 384 fp32 tensors ``unet.<site>.{down,up}.weight`` with the legacy diffusers
"processor" spelling, metadata {"format": "pt"}, no ``.alpha`` keys -- consumed unchanged by the reference's
scripts/create_wm_lora.py:24-41, scripts/diffusers_lora_to_webui.py and ``pipe.load_lora_weights``.
``mapper.pt`` / ``msgdecoder.pt``: plain ``torch.save(state_dict())``.
"""
import os

import torch
from safetensors.torch import load_file, save_file

from .lora import load_unet_keys, _walk

WEIGHT_NAME = "pytorch_lora_weights.safetensors"


def site_to_ckpt_key(key):
    k = key.replace(".proj_in", ".proj_in.lora").replace(".proj_out", ".proj_out.lora")
    k = k.replace(".to_q", ".processor.to_q_lora").replace(".to_k", ".processor.to_k_lora")
    k = k.replace(".to_v", ".processor.to_v_lora").replace(".to_out.0", ".processor.to_out_lora")
    if "ff" in k:
        k = k + ".lora"
    return "unet." + k


def lora_state_dict(unet, keys=None):
    keys = keys if keys is not None else load_unet_keys(unet)
    sd = {}
    for key in keys:
        lora = _walk(unet, key).lora_layer
        ck = site_to_ckpt_key(key)
        sd[ck + ".down.weight"] = lora.down.weight.detach().float().cpu().contiguous().clone()
        sd[ck + ".up.weight"] = lora.up.weight.detach().float().cpu().contiguous().clone()
    return sd


def save_lora_weights(save_directory, unet, mapper=None, msgdecoder_state=None, keys=None):
    os.makedirs(save_directory, exist_ok=True)
    save_file(lora_state_dict(unet, keys), os.path.join(save_directory, WEIGHT_NAME), metadata={"format": "pt"})
    if mapper is not None:
        torch.save({k: v.detach().cpu() for k, v in mapper.state_dict().items()},
                   os.path.join(save_directory, "mapper.pt"))
    if msgdecoder_state is not None:
        torch.save(msgdecoder_state, os.path.join(save_directory, "msgdecoder.pt"))


def load_lora_state(directory):
    """--resume_from_lora: rewrite checkpoint keys back to ``<site>.{down,up}.weight`` (ppft_train.py:626-633)."""
    value_dict = load_file(os.path.join(directory, WEIGHT_NAME))
    out = {}
    for k, v in value_dict.items():
        k = k.replace("lora.", "").replace(".processor.", ".").replace("unet.", "")
        k = k.replace("_down.", ".down.").replace("_up.", ".up.").replace(".to_out.", ".to_out.0.")
        k = k.replace("_lora.", ".")
        out[k] = v
    return out
