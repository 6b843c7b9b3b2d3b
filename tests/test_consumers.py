"""The reference's own consumers ran, unchanged, on an artefact written by aqualora_amd.checkpoint.save_lora_weights
(tests/golden/run_reference_consumers.py, authoring container): scripts/create_wm_lora.py, scripts/diffusers_lora_to_webui.py and
scripts/merge_lora.py:merge_to_sd_model.  tests/golden/consumers.npz holds what they produced.  CPU: the key contracts and the
oracle's bake; GPU: OUR bake / fuse of the same checkpoint (rebuilt by name) against the reference scripts' outputs."""
import numpy as np
import pytest
import torch

from tests.golden.run_reference_consumers import FULL, MERGED_FULL, RANK, consumers_checkpoint


def test_reference_consumers_accepted_our_checkpoint(golden):
    g = golden("consumers.npz")
    from aqualora_amd.checkpoint import site_to_ckpt_key
    from aqualora_amd.unet import lora_keys
    from tests.common import tiny_unet
    keys = lora_keys(tiny_unet())
    assert len(keys) == 192 and list(g["sites"]) == keys
    # create_wm_lora.py kept every key of our file (its substring branches, :24-37, matched all of them) ...
    ours = sorted(site_to_ckpt_key(k) + s for k in keys for s in (".down.weight", ".up.weight"))
    assert list(g["names"]) == ours and len(ours) == 384
    # ... diffusers_lora_to_webui.py turned them into kohya names that merge_lora.py:61-77 derives from the U-Net's module tree ...
    want = sorted("lora_unet_" + k.replace(".", "_") + s for k in keys for s in (".lora_down.weight", ".lora_up.weight"))
    assert list(g["webui_names"]) == want
    # ... and merge_to_sd_model found a module for every one of them and changed all 192 weights (no ".alpha" keys: scale 1, :101-103)
    assert int(g["n_module_not_found"]) == 0 and int(g["n_sites_merged"]) == 192
    # the oracle's restatement of the bake (scripts/create_wm_lora.py:21-37) == the script's output on OUR file
    from aqualora_amd import synth
    from oracle import ppft_oracle as O
    from tests.common import SEED, tiny_lora
    msg = torch.tensor([[float(c) for c in str(g["msg"])]])
    S = O.mapper(msg, synth.normal("cons.E", (48, RANK), 1.0, SEED))[0]
    lw = tiny_lora(keys, tiny_unet(), rank=RANK, up_std=0.05)
    by_ckpt = {site_to_ckpt_key(k): v for k, v in lw.items()}
    for i, name in enumerate(FULL):
        site, which = name.rsplit(".", 2)[0], name.rsplit(".", 2)[1]
        down, up = by_ckpt[site]
        ref = torch.from_numpy(g[f"full{i}"])
        if which == "up":
            assert torch.equal(ref, up)
        else:
            got = down * S.view(-1, *([1] * (down.dim() - 1))) * 1.03
            assert float((got - ref).abs().max()) < 1e-6 * float(ref.abs().max()) + 1e-9
    # the merged weights are W + up @ down' (merge_lora.py:105-118)
    for i, site in enumerate(MERGED_FULL):
        down, up = lw[site]
        dprime = (down * S.view(-1, *([1] * (down.dim() - 1))) * 1.03).reshape(RANK, -1)
        want_w = torch.from_numpy(g[f"merged_w0_{i}"]).reshape(up.shape[0], -1) + up.reshape(-1, RANK) @ dprime
        got_w = torch.from_numpy(g[f"merged_full{i}"]).reshape(up.shape[0], -1)
        assert float((got_w - want_w).abs().max()) < 1e-5 * float(want_w.abs().max())


@pytest.mark.gpu
def test_our_bake_and_fuse_equal_the_reference_scripts_outputs(golden):
    """inference.create_watermark_lora == scripts/create_wm_lora.py on the same file (1e-6 of every tensor's checksums, three
    tensors element-wise); inference.fuse_lora of that bake on the HIP U-Net == what scripts/merge_lora.py made of the reference's
    twin U-Net (same synthetic base weights; ours are stored in bf16: 2^-8 relative)."""
    from aqualora_amd.checkpoint import lora_state_dict
    from aqualora_amd.inference import create_watermark_lora, fuse_lora
    g = golden("consumers.npz")
    unet, keys, mapper, msg = consumers_checkpoint("cuda")
    assert msg == str(g["msg"])
    mapper.cuda()
    sd = lora_state_dict(unet, keys)
    hid, baked = create_watermark_lora(sd, mapper, msg, scale=1.03)
    assert hid == msg and sorted(baked) == list(g["names"])
    for name, (s, a) in zip(g["names"], g["baked_checksums"]):
        t = baked[str(name)].double().cpu()
        assert abs(float(t.abs().sum()) - a) < 1e-6 * a and abs(float(t.sum()) - s) < 1e-6 * a, name
    for i, name in enumerate(FULL):
        ref = torch.from_numpy(g[f"full{i}"])
        assert float((baked[name].float().cpu() - ref).abs().max()) < 1e-6 * float(ref.abs().max()) + 1e-9
    fuse_lora(unet, baked, 1.0, keys)
    for i, site in enumerate(MERGED_FULL):
        w = unet.get_submodule(site).weight.detach().float().cpu().reshape(-1)
        ref = torch.from_numpy(g[f"merged_full{i}"]).reshape(-1)
        w0 = torch.from_numpy(g[f"merged_w0_{i}"]).reshape(-1)
        assert float((w - ref).abs().max()) < 2 ** -7 * float(ref.abs().max())
        # the fused delta is really there: closer to the merged weights than to the base weights
        assert float((w - ref).norm()) < 0.5 * float((w - w0).norm())
        assert unet.get_submodule(site).lora_layer is None
