"""Runs the REFERENCE'S OWN consumers, unchanged, on an artefact written by aqualora_amd.checkpoint.save_lora_weights, in the authoring
container (needs /root/reference; never runs on the GPU box), and writes what they produced to tests/golden/consumers.npz:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/run_reference_consumers.py

  1. aqualora_amd.checkpoint.save_lora_weights(dir, unet, mapper)         pytorch_lora_weights.safetensors + mapper.pt
                                                                          (the layout of train/ppft_train.py:1203-1229)
  2. scripts/create_wm_lora.py:create_watermark_lora(dir, 1.03, 48, msg)  bakes the message, writes dir/<msg>/pytorch_lora_weights.safetensors
  3. scripts/diffusers_lora_to_webui.py:diffuers2webui                    key conversion to the kohya / A1111 names
  4. scripts/merge_lora.py:merge_to_sd_model                              W <- W + up @ down into the reference's own U-Net twin
                                                                          (scripts/lib/original_unet.py), one module per LoRA key

The checkpoint is the SD-1.5 LoRA topology (16 attention blocks x 12 sites = 192 sites, 384 tensors; utils/unet_keys.json) at the
reduced channel widths of tests/common.TINY and rank 320 (create_wm_lora.py:19 hard-codes MapperNet(output_size=320)); its values are
counter-based (aqualora_amd.synth), so the GPU test rebuilds the same checkpoint by name and compares OUR bake / fuse with what the
reference's scripts made of it.  Third-party imports of the reference's modules that are not installed here are the inert stubs of
make_golden.py; the arithmetic that runs is the reference's.  Only outputs are stored: key names, per-tensor checksums, three full
tensors.  No reference source.
"""
import contextlib
import io
import os
import sys
import tempfile

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.golden import make_golden as MG  # noqa: E402  (stubs + the reference-U-Net builder)

RANK = 320
FULL = ("unet.down_blocks.0.attentions.0.proj_in.lora.down.weight",
        "unet.mid_block.attentions.0.transformer_blocks.0.attn2.processor.to_k_lora.down.weight",
        "unet.up_blocks.3.attentions.2.transformer_blocks.0.ff.net.0.proj.lora.up.weight")


MERGED_FULL = ("down_blocks.0.attentions.0.proj_in", "mid_block.attentions.0.transformer_blocks.0.attn2.to_k",
               "up_blocks.3.attentions.2.transformer_blocks.0.ff.net.2")


def consumers_checkpoint(device="cpu"):
    """(unet with the rank-320 LoRA injected, mapper, message string): every value by name -- also what the GPU test builds."""
    from aqualora_amd import synth
    from aqualora_amd.lora import inject_lora
    from aqualora_amd.unet import lora_keys
    from aqualora_amd.watermark import MapperNet
    from tests.common import SEED, tiny_lora, tiny_unet
    unet = tiny_unet(device, torch.float32 if device == "cpu" else torch.bfloat16)
    keys = lora_keys(unet)
    lw = tiny_lora(keys, unet, rank=RANK, up_std=0.05)
    state = {}
    for k, (d, u) in lw.items():
        state[k + ".down.weight"], state[k + ".up.weight"] = d, u
    inject_lora(unet, RANK, keys, state)
    mapper = MapperNet(48, RANK)
    with torch.no_grad():
        mapper.bit_embeddings.weight.copy_(synth.normal("cons.E", (48, RANK), 1.0, SEED))
    msg = "".join(str(int(b)) for b in synth.bits("cons.msg", (48,), SEED).tolist())
    return unet, keys, mapper, msg


def checksums(sd, names):
    return np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in names])


def main():
    from safetensors.torch import load_file, save_file
    from aqualora_amd import checkpoint as CK
    torch.set_num_threads(8)
    # scripts/lib/model_util.py imports these from the real transformers package: resolve them BEFORE the torchvision stub exists
    # (transformers probes torchvision when it is importable and the inert stub does not survive that)
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer, logging  # noqa: F401
    lm, models, misc, ou = MG.import_reference()
    sys.path.insert(0, os.path.join(REF, "scripts"))
    import create_wm_lora as cw
    import diffusers_lora_to_webui as d2w
    import merge_lora as ml

    unet, keys, mapper, msg = consumers_checkpoint()
    with tempfile.TemporaryDirectory() as d:
        CK.save_lora_weights(d, unet, mapper, keys=keys)
        written = load_file(os.path.join(d, "pytorch_lora_weights.safetensors"))
        # (2) the reference's bake, saving like the script's __main__ does
        hid, baked = cw.create_watermark_lora(d, 1.03, 48, msg, save=True)
        assert hid == msg
        baked_file = os.path.join(d, msg, "pytorch_lora_weights.safetensors")
        on_disk = load_file(baked_file)
        assert set(on_disk) == set(written) and len(on_disk) == 2 * len(keys) == 384
        # (3) the converter iterates the module-global `checkpoint` its __main__ sets (scripts/diffusers_lora_to_webui.py:5-7,36)
        d2w.checkpoint = on_disk
        with contextlib.redirect_stdout(io.StringIO()):
            webui = d2w.diffuers2webui(None)
        assert len(webui) == 384
        webui_file = os.path.join(d, "webui.safetensors")
        save_file({k: v.contiguous() for k, v in webui.items()}, webui_file)
        # (4) merge into the reference's own U-Net twin at the same widths; a text encoder without target modules
        from tests.common import TINY
        ref_unet, _, _, _ = MG.build_reference_unet(ou, lm, TINY, 4)
        for key in keys:                       # the LoRA hosts of build_reference_unet: back to plain modules for the merge
            m = ref_unet
            for sub in key.split("."):
                m = getattr(m, sub)
            m.__class__ = torch.nn.Conv2d if isinstance(m, torch.nn.Conv2d) else torch.nn.Linear
            m.lora_layer = None
            if "forward" in m.__dict__:
                del m.__dict__["forward"]
        w0 = {k: _site(ref_unet, k).weight.detach().clone() for k in keys}
        log = io.StringIO()
        with contextlib.redirect_stdout(log):
            ml.merge_to_sd_model(torch.nn.Module(), ref_unet, [webui_file], [1.0], torch.float32)
        not_found = log.getvalue().count("no module found")
        merged = {k: _site(ref_unet, k).weight.detach() for k in keys}
        changed = sum(int(not torch.equal(merged[k], w0[k])) for k in keys)
    names = sorted(on_disk)
    arrs = dict(msg=np.array(msg), names=np.array(names), baked_checksums=checksums(on_disk, names),
                webui_names=np.array(sorted(webui)), n_module_not_found=np.array(not_found), n_sites_merged=np.array(changed),
                sites=np.array(keys), merged_delta_checksums=checksums({k: merged[k] - w0[k] for k in keys}, keys))
    for i, k in enumerate(FULL):
        arrs[f"full{i}"] = on_disk[k].numpy()
    for i, k in enumerate(MERGED_FULL):
        arrs[f"merged_full{i}"] = merged[k].numpy()
        arrs[f"merged_w0_{i}"] = w0[k].numpy()
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "consumers.npz"), **arrs)
    print(f"consumers.npz: {len(names)} tensors baked, {len(webui)} webui keys, {changed} / {len(keys)} sites merged, "
          f"{not_found} keys without a module")


def _site(root, key):
    m = root
    for sub in key.split("."):
        m = getattr(m, sub)
    return m


if __name__ == "__main__":
    main()
