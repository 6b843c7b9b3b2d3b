"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE in the authoring container.

Run from the repo root (needs /root/reference; never runs on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported from /root/reference, unmodified: utils/lora_modules.py (the four forwards), utils/models.py
(MapperNet, SecretEncoder), utils/misc.py (LR schedule), scripts/lib/original_unet.py (the in-tree SD U-Net twin)
and train/ppft_train.py (only ``unet_attn_processors_state_dict``, for the checkpoint key layout).  Third-party
packages that are not installed here (diffusers, torchvision, lpips, timm, ...) are replaced by inert stub modules
whose only job is to make the imports succeed; the arithmetic that runs is the reference's.
Only inputs-by-name (aqualora_amd.synth) and the reference's OUTPUTS are written; no reference source is stored.
"""
import importlib.machinery
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from aqualora_amd import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
SEED = 2048


# ------------------------------------------------------------------------------------------- stubs
class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Dummy()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        cls = type(name, (_Dummy,), {})
        setattr(self, name, cls)
        return cls


def _stub(name):
    m = _StubModule(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    m.__path__ = []
    sys.modules[name] = m
    return m


class LoRACompatibleLinear(nn.Linear):  # stub host classes with the attribute set diffusers 0.24 has
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.lora_layer = None

    def set_lora_layer(self, l):
        self.lora_layer = l


class LoRACompatibleConv(nn.Conv2d):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.lora_layer = None

    def set_lora_layer(self, l):
        self.lora_layer = l


class LoRALinearLayer(nn.Module):
    def __init__(self, in_features, out_features, rank=4, network_alpha=None):
        super().__init__()
        self.down = nn.Linear(in_features, rank, bias=False)
        self.up = nn.Linear(rank, out_features, bias=False)
        self.network_alpha, self.rank = network_alpha, rank


class LoRAConv2dLayer(nn.Module):
    def __init__(self, in_features, out_features, rank=4, kernel_size=(1, 1), stride=(1, 1), padding=0,
                 network_alpha=None):
        super().__init__()
        self.down = nn.Conv2d(in_features, rank, kernel_size=kernel_size, stride=stride, padding=padding, bias=False)
        self.up = nn.Conv2d(rank, out_features, kernel_size=(1, 1), bias=False)
        self.network_alpha, self.rank = network_alpha, rank


def install_stubs():
    for n in ["diffusers", "diffusers.loaders", "diffusers.models", "diffusers.models.lora", "diffusers.optimization",
              "diffusers.utils", "diffusers.utils.import_utils", "diffusers.utils.torch_utils",
              "diffusers.schedulers", "diffusers.schedulers.scheduling_ddpm", "diffusers.schedulers.scheduling_ddim",
              "diffusers.schedulers.scheduling_euler_discrete", "diffusers.configuration_utils",
              "diffusers.training_utils", "torchvision", "torchvision.models", "torchvision.models.efficientnet",
              "torchvision.transforms", "torchvision.utils", "torchvision.io", "torchvision.transforms.functional",
              "diffusers.pipelines", "diffusers.pipelines.stable_diffusion", "lpips", "timm", "kornia", "kornia.augmentation",
              "xformers", "wandb", "bitsandbytes"]:
        if n not in sys.modules:
            _stub(n)
    lora = sys.modules["diffusers.models.lora"]
    lora.LoRACompatibleLinear = LoRACompatibleLinear
    lora.LoRACompatibleConv = LoRACompatibleConv
    lora.LoRALinearLayer = LoRALinearLayer
    lora.LoRAConv2dLayer = LoRAConv2dLayer
    sys.modules["diffusers"].__version__ = "0.24.0"


def import_reference():
    install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "scripts", "lib"))
    import utils.lora_modules as lm
    import utils.models as models
    import utils.misc as misc
    import original_unet as ou
    return lm, models, misc, ou


def T(name, shape, std=1.0):
    return synth.normal(name, shape, std, SEED)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", name, {k: np.asarray(v).shape for k, v in arrs.items()})


# --------------------------------------------------------------------------------- G1/G2 LoRA forwards
LORA_CASES = [("lin_a", 320, 320, 16, 8), ("lin_b", 768, 320, 77, 8), ("lin_c", 320, 2560, 8, 32),
              ("lin_d", 1280, 320, 8, 32)]


def gen_lora(lm):
    out = {}
    for tag, cin, cout, n, r in LORA_CASES:
        B = 2
        host = LoRACompatibleLinear(cin, cout)
        lora = LoRALinearLayer(cin, cout, r)
        with torch.no_grad():
            host.weight.copy_(T(f"{tag}.w", (cout, cin), cin ** -0.5))
            host.bias.copy_(T(f"{tag}.b", (cout,), 0.02))
            lora.down.weight.copy_(T(f"{tag}.down", (r, cin), 1.0 / r))
            lora.up.weight.copy_(T(f"{tag}.up", (cout, r), 0.05))
        host.set_lora_layer(lora)
        host.forward = types.MethodType(lm.CustomLoRACompatibleLinearforward, host)
        lora.forward = types.MethodType(lm.CustomLoRALinearLayerforward, lora)
        x = T(f"{tag}.x", (B, n, cin)).requires_grad_(True)
        S = (T(f"{tag}.S", (B, r), 0.3) + 1.0).requires_grad_(True)
        dy = T(f"{tag}.dy", (B, n, cout))
        y = host(x, S)
        y.backward(dy)
        out[f"{tag}.y"] = y.detach().numpy()
        out[f"{tag}.dx"] = x.grad.numpy()
        out[f"{tag}.dS"] = S.grad.numpy()
        out[f"{tag}.ddown"] = lora.down.weight.grad.numpy()
        out[f"{tag}.dup"] = lora.up.weight.grad.numpy()
        out[f"{tag}.y_float_scale"] = host(x.detach(), 0.5).detach().numpy()
        out[f"{tag}.lora_only"] = lora(x.detach(), S.detach()).detach().numpy()
        # conv 1x1 form on the same numbers (G2): x as [B, cin, h, w] with h*w = n
        if n % 4 == 0:
            h, w = 4, n // 4
            hostc = LoRACompatibleConv(cin, cout, 1)
            lorac = LoRAConv2dLayer(cin, cout, r)
            with torch.no_grad():
                hostc.weight.copy_(host.weight.view(cout, cin, 1, 1))
                hostc.bias.copy_(host.bias)
                lorac.down.weight.copy_(lora.down.weight.view(r, cin, 1, 1))
                lorac.up.weight.copy_(lora.up.weight.view(cout, r, 1, 1))
            hostc.set_lora_layer(lorac)
            hostc.forward = types.MethodType(lm.CustomLoRACompatibleConvforward, hostc)
            lorac.forward = types.MethodType(lm.CustomLoRAConv2dLayerforward, lorac)
            xc = x.detach().permute(0, 2, 1).reshape(B, cin, h, w)
            out[f"{tag}.conv_y"] = hostc(xc, S.detach()).detach().numpy()
    save("lora_forwards.npz", **out)


# ------------------------------------------------------------------------------ G3/G4 mapper + encoder
def gen_watermark(models):
    bits, r = 48, 32
    mp = models.MapperNet(input_size=bits, output_size=r)
    E = T("mapper.E", (bits, r))
    with torch.no_grad():
        mp.bit_embeddings.weight.copy_(E)
    msg = synth.bits("msg", (4, bits), SEED)
    S = mp(msg)
    dS = T("mapper.dS", (4, r))
    S.backward(dS)
    torch.manual_seed(0)
    fresh = models.MapperNet(input_size=bits, output_size=r, std=1.0).bit_embeddings.weight.detach()
    enc = models.SecretEncoder(bits)
    zero_out = enc(torch.zeros(4, 4, 64, 64), msg)[1]
    with torch.no_grad():
        enc.secret_scaler[0].weight.copy_(T("enc.lin.w", (1024, bits), bits ** -0.5))
        enc.secret_scaler[0].bias.copy_(T("enc.lin.b", (1024,), 0.1))
        enc.secret_scaler[5].weight.copy_(T("enc.conv.w", (4, 4, 3, 3), 0.05))
        enc.secret_scaler[5].bias.copy_(T("enc.conv.b", (4,), 0.01))
    x = T("enc.x", (4, 4, 64, 64))
    xc, c = enc(x, msg)
    save("watermark.npz", S=S.detach().numpy(), dE=mp.bit_embeddings.weight.grad.numpy(),
         fresh_row_std=fresh.std(dim=1).numpy(), fresh_gram_offdiag_max=np.float32(
             (fresh @ fresh.T - torch.diag(torch.diag(fresh @ fresh.T))).abs().max().item()),
         zero_init_absmax=np.float32(zero_out.abs().max().item()), c=c.detach().numpy(),
         x_plus_c_checksum=np.float64(xc.double().sum().item()),
         enc_keys=np.array(sorted(enc.state_dict().keys())), mapper_keys=np.array(sorted(mp.state_dict().keys())))


def gen_misc(misc):
    class _Opt(torch.optim.SGD):
        pass
    p = nn.Parameter(torch.zeros(1))
    rows = []
    for (warm, total, lr_end) in [(0, 100, 0.01), (10, 100, 0.01), (5, 50, 0.1), (0, 8, 0.0)]:
        opt = _Opt([p], lr=1.0)
        sch = misc.get_cosine_schedule_with_warmup_lr_end(opt, warm, total, lr_end=lr_end)
        vals = []
        for _ in range(total + 5):
            vals.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        rows.append(np.array([warm, total, lr_end] + vals, dtype=np.float64))
    save("lr_schedule.npz", **{f"case{i}": r for i, r in enumerate(rows)})


# -------------------------------------------------------------------------------- G8 tiny U-Net PPFT step
TINY = dict(block_out_channels=(32, 64, 64, 64), cross_attention_dim=32, attention_heads=2, layers_per_block=2)
TINY_RANK = 8


def build_reference_unet(ou, lm, cfg, rank, seed_tag="unet"):
    ou.BLOCK_OUT_CHANNELS = tuple(cfg["block_out_channels"])
    ou.TIMESTEP_INPUT_DIM = cfg["block_out_channels"][0]
    ou.TIME_EMBED_DIM = cfg["block_out_channels"][0] * 4
    unet = ou.UNet2DConditionModel(sample_size=16, attention_head_dim=cfg["attention_heads"],
                                   cross_attention_dim=cfg["cross_attention_dim"], use_linear_projection=False,
                                   upcast_attention=False)
    # same synthetic weights as aqualora_amd.unet.init_synthetic
    with torch.no_grad():
        for name, p in unet.named_parameters():
            if name.endswith("weight") and p.dim() >= 2:
                p.copy_(synth.normal(name, p.shape, p[0].numel() ** -0.5, SEED))
            elif name.endswith("weight"):
                p.fill_(1.0)
            elif "norm" in name:
                p.zero_()
            else:
                p.copy_(synth.normal(name, p.shape, 0.02, SEED))
    keys = json.load(open(os.path.join(REF, "utils", "unet_keys.json")))
    holder = types.SimpleNamespace(scale=None)
    loras = {}
    for key in keys:
        m = unet
        for sub in key.split("."):
            m = getattr(m, sub)
        if isinstance(m, nn.Conv2d):
            m.__class__ = LoRACompatibleConv
            lora = LoRAConv2dLayer(m.in_channels, m.out_channels, rank, m.kernel_size, m.stride, m.padding)
            lora.forward = types.MethodType(lm.CustomLoRAConv2dLayerforward, lora)
            fwd = lm.CustomLoRACompatibleConvforward
        else:
            m.__class__ = LoRACompatibleLinear
            lora = LoRALinearLayer(m.in_features, m.out_features, rank)
            lora.forward = types.MethodType(lm.CustomLoRALinearLayerforward, lora)
            fwd = lm.CustomLoRACompatibleLinearforward
        with torch.no_grad():
            lora.down.weight.copy_(synth.normal(key + ".lora.down", lora.down.weight.shape, 1.0 / rank, SEED))
            lora.up.weight.copy_(synth.normal(key + ".lora.up", lora.up.weight.shape, 0.1, SEED))
        m.lora_layer = lora
        m.forward = types.MethodType(lambda self, x, _f=fwd: _f(self, x, holder.scale), m)
        loras[key] = lora
    return unet, keys, holder, loras


def gen_tiny_ppft(lm, models, ou):
    cfg = TINY
    unet, keys, holder, loras = build_reference_unet(ou, lm, cfg, TINY_RANK)
    B, bits = 2, 48
    mapper = models.MapperNet(input_size=bits, output_size=TINY_RANK)
    with torch.no_grad():
        mapper.bit_embeddings.weight.copy_(T("ppft.mapper.E", (bits, TINY_RANK)))
    msg = synth.bits("ppft.msg", (B, bits), SEED)
    z = T("ppft.z", (B, 4, 16, 16))
    wm = T("ppft.wm", (B, 4, 16, 16), 0.5)
    eps = T("ppft.eps", (B, 4, 16, 16))
    t = synth.randint("ppft.t", (B,), 1000, SEED)
    ctx = T("ppft.ctx", (B, 77, cfg["cross_attention_dim"]))
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    acp = torch.cumprod(1 - betas, 0)
    sa = (acp[t] ** 0.5)[:, None, None, None]
    sb = ((1 - acp[t]) ** 0.5)[:, None, None, None]
    x_t = sa * z + sb * eps
    x_t_wm = sa * (z + wm) + sb * eps
    S = mapper(msg)
    holder.scale = torch.zeros_like(S)
    clean = unet(x_t, t, ctx).sample.detach()
    holder.scale = S
    pred = unet(x_t_wm, t, ctx).sample
    loss = torch.nn.functional.mse_loss(pred.float(), clean.float(), reduction="mean")
    loss.backward()
    params = []
    for k in keys:
        params += [loras[k].down.weight, loras[k].up.weight]
    gnorms = np.array([p.grad.norm().item() for p in params], dtype=np.float64)
    total_norm = float(np.sqrt((gnorms ** 2).sum()))
    pick = [keys[0], keys[1], keys[4], keys[6], keys[10], keys[-1]]
    full = {}
    for k in pick:
        full["g." + k + ".down"] = loras[k].down.weight.grad.numpy().copy()
        full["g." + k + ".up"] = loras[k].up.weight.grad.numpy().copy()
    mapper_grad = mapper.bit_embeddings.weight.grad.numpy().copy()
    # one optimiser step exactly as ppft_train.py:1059-1068, 779-787
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt = torch.optim.AdamW([{"params": params}, {"params": mapper.parameters()}], lr=1e-4, betas=(0.9, 0.999),
                            weight_decay=1e-2, eps=1e-8)
    opt.step()
    post = {"p." + pick[0] + ".down": loras[pick[0]].down.weight.detach().numpy().copy(),
            "p." + pick[0] + ".up": loras[pick[0]].up.weight.detach().numpy().copy(),
            "p.mapper": mapper.bit_embeddings.weight.detach().numpy().copy()}
    save("tiny_ppft.npz", clean=clean.numpy(), pred=pred.detach().numpy(), loss=np.float64(loss.item()),
         grad_norms=gnorms, total_norm=np.float64(total_norm), mapper_grad=mapper_grad, S=S.detach().numpy(),
         keys=np.array(keys), **full, **post)
    return unet, keys, loras


def gen_checkpoint_layout(unet, keys, loras):
    """Key names + shapes produced by the reference's own unet_attn_processors_state_dict (ppft_train.py:443-471)."""
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "train"))
    try:
        sys.argv = ["ppft_train.py"]
        import importlib.util
        spec = importlib.util.spec_from_file_location("ref_ppft_train", os.path.join(REF, "train", "ppft_train.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        unet.attn_processors = {}
        sd = mod.unet_attn_processors_state_dict(unet)
    finally:
        os.chdir(cwd)
    names = sorted("unet." + k for k in sd.keys())  # save_lora_weights prefixes "unet." (diffusers, recalled)
    shapes = np.array([list(sd[n[len("unet."):]].shape) + [0] * (4 - sd[n[len("unet."):]].dim()) for n in names])
    save("checkpoint_layout.npz", names=np.array(names), shapes=shapes, rank=np.int64(TINY_RANK))


def gen_full_block(lm, ou):
    """G9: one full-width transformer block (C=320, 8 heads, 16x16 tokens) through the reference twin + forwards."""
    ou.BLOCK_OUT_CHANNELS = (320, 640, 1280, 1280)
    tr = ou.Transformer2DModel(8, 40, in_channels=320, cross_attention_dim=768, use_linear_projection=False)
    with torch.no_grad():
        for name, p in tr.named_parameters():
            full = "blk." + name
            if name.endswith("weight") and p.dim() >= 2:
                p.copy_(synth.normal(full, p.shape, p[0].numel() ** -0.5, SEED))
            elif name.endswith("weight"):
                p.fill_(1.0)
            elif "norm" in name:
                p.zero_()
            else:
                p.copy_(synth.normal(full, p.shape, 0.02, SEED))
    x = T("blk.x", (1, 320, 16, 16))
    ctx = T("blk.ctx", (1, 77, 768))
    y = tr(x, ctx).sample
    save("full_block.npz", y=y.detach().numpy())


def gen_jpeg_and_metrics():
    import utils.noise_layers.jpeg_compression as jc
    layer = jc.JpegCompression("cpu")
    out = {}
    for tag, shape in (("a", (2, 3, 64, 64)), ("b", (1, 3, 50, 44))):
        x = T(f"jpeg.{tag}.x", shape, 0.5).requires_grad_(True)
        y = layer([x, None])[0]
        dy = T(f"jpeg.{tag}.dy", shape)
        y.backward(dy)
        out[f"{tag}.y"] = y.detach().numpy()
        out[f"{tag}.dx"] = x.grad.numpy()
    mask = layer.get_mask((3, 8, 8)).numpy()
    out["mask8"] = mask
    out["mask_counts"] = mask.reshape(3, -1).sum(1)
    save("jpeg.npz", **out)
    # metrics: evaluation/utils_eval.py get_threshold / calculate_fpr (file imports heavy third-party modules; stubs)
    for n in ["PIL", "diffusers.pipelines", "tqdm"]:
        pass
    sys.path.insert(0, os.path.join(REF, "evaluation"))
    import importlib
    try:
        ue = importlib.import_module("utils_eval")
        ks = [16, 32, 48, 64]
        fprs = [1e-2, 1e-3, 1e-6, 1e-9]
        thr = np.array([[ue.get_threshold(k, f) for f in fprs] for k in ks])
        fpr_tab = np.array([ue.calculate_fpr(t, 48) for t in range(0, 48, 4)])
        save("metrics.npz", ks=np.array(ks), fprs=np.array(fprs), thresholds=thr, fpr48=fpr_tab)
    except Exception as e:  # noqa: BLE001
        print("utils_eval import failed:", repr(e))
        raise


def gen_create_wm_lora(models):
    """scripts/create_wm_lora.py:create_watermark_lora on a small synthetic checkpoint (rank 320 is hard-coded there)."""
    import tempfile
    from safetensors.torch import save_file
    sys.path.insert(0, os.path.join(REF, "scripts"))
    import create_wm_lora as cw
    r = 320
    sd = {}
    for k, dshape, ushape in (("unet.down_blocks.0.attentions.0.proj_in.lora", (r, 32, 1, 1), (32, r, 1, 1)),
                              ("unet.down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor.to_q_lora", (r, 32), (32, r)),
                              ("unet.down_blocks.0.attentions.0.transformer_blocks.0.ff.net.2.lora", (r, 128), (32, r))):
        sd[k + ".down.weight"] = T(k + ".down", dshape, 1.0 / r)
        sd[k + ".up.weight"] = T(k + ".up", ushape, 0.05)
    mp = models.MapperNet(input_size=48, output_size=r)
    with torch.no_grad():
        mp.bit_embeddings.weight.copy_(T("cwl.E", (48, r)))
    msg = "".join(str(int(b)) for b in synth.bits("cwl.msg", (48,), SEED).tolist())
    with tempfile.TemporaryDirectory() as d:
        save_file(sd, os.path.join(d, "pytorch_lora_weights.safetensors"))
        torch.save(mp.state_dict(), os.path.join(d, "mapper.pt"))
        hid, out = cw.create_watermark_lora(d, 1.03, 48, msg, save=False)
    assert hid == msg
    save("create_wm_lora.npz", msg=np.array(msg), **{k: v.detach().numpy() for k, v in out.items()})


def prvl_case(i, B, H, W, amp):
    a = T(f"prvl.a{i}", (B, 3, H, W), 0.5)
    d = T(f"prvl.d{i}", (B, 3, H, W), amp)
    d[:, :, H // 3: H // 3 + 20, W // 4: W // 4 + 25] *= 4.0   # a localised artefact, the thing PRVL looks for
    return a, (a + d)


def gen_stage1():
    """PRVL_loss of the reference's stage-1 script (train/latent_wm_pretrain.py:42-50), imported with the third-party
    modules stubbed (tensorboard, torchsummary and the kornia/torchvision based noiser are absent here)."""
    for n in ["torch.utils.tensorboard", "torchsummary", "kornia.augmentation", "torchvision.transforms"]:
        if n not in sys.modules:
            _stub(n)
    sys.path.insert(0, os.path.join(REF, "train"))
    import importlib
    st = importlib.import_module("latent_wm_pretrain")
    out = {}
    for i, (B, H, W, amp) in enumerate([(1, 512, 512, 0.05), (2, 96, 80, 0.3), (3, 40, 33, 1.0)]):
        a, b = prvl_case(i, B, H, W, amp)
        b.requires_grad_(True)
        loss = st.PRVL_loss(a, b)
        loss.backward()
        out[f"loss{i}"] = loss.detach().numpy()
        if i == 0:   # full-size case: inputs are regenerated by the test (counter-based generator), gradient summarised
            g = b.grad
            nz = g.nonzero()
            out["grad_nnz0"], out["grad_abs_sum0"] = np.array(len(nz)), g.abs().sum().numpy()
            out["grad_bbox0"] = np.array([nz[:, 2].min(), nz[:, 2].max(), nz[:, 3].min(), nz[:, 3].max()])
        else:
            out[f"a{i}"], out[f"b{i}"], out[f"grad_b{i}"] = a.numpy(), b.detach().numpy(), b.grad.numpy()
    save("stage1_prvl.npz", **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    lm, models, misc, ou = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "stage1":
        gen_stage1()
        sys.exit(0)
    gen_lora(lm)
    gen_watermark(models)
    gen_misc(misc)
    gen_full_block(lm, ou)
    gen_jpeg_and_metrics()
    gen_create_wm_lora(models)
    unet, keys, loras = gen_tiny_ppft(lm, models, ou)
    try:
        gen_checkpoint_layout(unet, keys, loras)
    except Exception as e:  # noqa: BLE001
        print("checkpoint layout via reference function failed:", repr(e))
        raise
