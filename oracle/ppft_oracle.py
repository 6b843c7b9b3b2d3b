"""ORACLE -- test infrastructure only.  CPU restatement (plain PyTorch fp32) of the reference's PPFT hot path.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import this file; the product
(aqualora_amd/) never does.  Parity status: PINNED -- tests/test_oracle_golden.py checks every function below
against tests/golden/*.npz, which were produced by running the reference's own code (tests/golden/make_golden.py).

Each function cites the reference lines it restates (paths relative to /root/reference):
  lora_linear / lora_conv1x1      utils/lora_modules.py:9-26, 28-44, 46-54, 56-62
  mapper                          utils/models.py:110-115
  secret_encoder                  utils/models.py:57-64, 74-81
  add_noise, alphas_cumprod       diffusers DDPMScheduler.add_noise as called at train/ppft_train.py:1010-1011
                                  (SD-1.5 scaled_linear schedule), utils/cschedulers.py:56-72
  lr_lambda                       utils/misc.py:23-33
  unet_forward                    scripts/lib/original_unet.py:323-361, 413-462, 543-640, 708-890, 893-1249, 1464-1585
  ppft_loss / ppft_step           train/ppft_train.py:987-1068 (+ optimizer 779-787)
  lora_state_dict_keys            train/ppft_train.py:443-471, 1217-1221

``rb`` (round-to-bf16 and back) is applied after every operator when ``bf16=True`` so the oracle mirrors the
product's bf16 activation storage while keeping fp32 arithmetic inside each op -- this is what the GPU parity
tests compare against.  With ``bf16=False`` it is the plain fp32 maths that is pinned to the golden vectors.
"""
import math

import torch
import torch.nn.functional as F


def _rb(t, on):
    return t.to(torch.bfloat16).float() if on else t


# ------------------------------------------------------------------------------------------- LoRA forwards
def lora_branch(x, down_w, up_w, scale):
    """up(down(x) @ diag_embed(scale)); float scale multiplies the output (lora_modules.py:9-26)."""
    t = x @ down_w.reshape(down_w.shape[0], -1).T
    if isinstance(scale, torch.Tensor):
        t = t * scale[:, None, :] if t.dim() == 3 else t * scale
        return t @ up_w.reshape(up_w.shape[0], -1).T
    return scale * (t @ up_w.reshape(up_w.shape[0], -1).T)


def lora_linear(x, w, b, down_w=None, up_w=None, scale=1.0):
    """nn.Linear(x) [+ lora(x, scale)] (lora_modules.py:56-62)."""
    y = F.linear(x, w, b)
    if down_w is not None and scale is not None:
        y = y + lora_branch(x, down_w, up_w, scale)
    return y


def lora_conv1x1(x, w, b, down_w=None, up_w=None, scale=1.0):
    """Conv2d 1x1 [+ up(down(x) * scale[:, :, None, None])] on NCHW (lora_modules.py:46-54, 28-44)."""
    y = F.conv2d(x, w.reshape(w.shape[0], -1, 1, 1), b)
    if down_w is not None and scale is not None:
        t = F.conv2d(x, down_w.reshape(down_w.shape[0], -1, 1, 1))
        if isinstance(scale, torch.Tensor):
            t = t * scale[:, :, None, None]
            y = y + F.conv2d(t, up_w.reshape(up_w.shape[0], -1, 1, 1))
        else:
            y = y + scale * F.conv2d(t, up_w.reshape(up_w.shape[0], -1, 1, 1))
    return y


# ---------------------------------------------------------------------------------- watermark modules
def mapper(msg, E):
    """(E * x[:, :, None]).sum(1) / sqrt(bits) + 1 (models.py:110-115)."""
    return (E[None] * msg[:, :, None]).sum(dim=1) / math.sqrt(E.shape[0]) + 1.0


def secret_encoder(msg, lin_w, lin_b, conv_w, conv_b, base_res=32, res=64):
    """Linear -> SiLU -> view [B,1,R,R] -> repeat 4 ch -> nearest upsample -> conv3x3 (models.py:57-64)."""
    h = F.silu(F.linear(msg, lin_w, lin_b)).view(-1, 1, base_res, base_res).repeat(1, 4, 1, 1)
    h = F.interpolate(h, scale_factor=float(res // base_res), mode="nearest")
    return F.conv2d(h, conv_w, conv_b, padding=1)


def alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def add_noise(x0, noise, t, acp=None):
    acp = alphas_cumprod() if acp is None else acp
    sa = (acp[t] ** 0.5).view(-1, 1, 1, 1)
    sb = ((1 - acp[t]) ** 0.5).view(-1, 1, 1, 1)
    return sa * x0 + sb * noise


def lr_lambda(step, warm, total, lr_end, cycles=0.5):
    if step < warm:
        return float(step) / float(max(1, warm))
    progress = float(step - warm) / float(max(1, total - warm))
    return max(lr_end, 0.5 * (1.0 + math.cos(math.pi * float(cycles) * 2.0 * progress)))


# -------------------------------------------------------------------------------------------- U-Net
def timestep_embedding(t, dim):
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / half
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)  # flip_sin_to_cos=True, shift 0


class UNetOracle:
    """Functional SD-1.5-style U-Net over a diffusers-keyed state dict ``sd`` and an optional LoRA dict
    ``lora[key] = (down_w, up_w)``; ``scale`` as in the reference (tensor [B,r] | float | None)."""

    def __init__(self, sd, cfg, lora=None, bf16=False):
        self.sd = {k: v.float() for k, v in sd.items()}
        self.cfg = cfg
        self.lora = lora or {}
        self.bf16 = bf16
        if bf16:
            self.sd = {k: _rb(v, True) for k, v in self.sd.items()}

    def r(self, t):
        return _rb(t, self.bf16)

    def _lw(self, key):
        if key in self.lora:
            d, u = self.lora[key]
            return (self.r(d.float()), self.r(u.float())) if self.bf16 else (d, u)
        return None, None

    def lin(self, key, x, scale, bias=True):
        d, u = self._lw(key)
        s = self.r(scale) if isinstance(scale, torch.Tensor) else scale
        return lora_linear(x, self.sd[key + ".weight"], self.sd.get(key + ".bias") if bias else None, d, u, s)

    def conv1(self, key, x, scale):
        d, u = self._lw(key)
        s = self.r(scale) if isinstance(scale, torch.Tensor) else scale
        return lora_conv1x1(x, self.sd[key + ".weight"], self.sd[key + ".bias"], d, u, s)

    def conv3(self, key, x, stride=1):
        return F.conv2d(x, self.sd[key + ".weight"], self.sd[key + ".bias"], stride=stride, padding=1)

    def gn(self, key, x, eps, silu):
        y = F.group_norm(x, 32, self.sd[key + ".weight"], self.sd[key + ".bias"], eps)
        return self.r(F.silu(y) if silu else y)

    def ln(self, key, x):
        return self.r(F.layer_norm(x, (x.shape[-1],), self.sd[key + ".weight"], self.sd[key + ".bias"], 1e-5))

    def resnet(self, p, x, temb_act, scale):
        h = self.gn(p + ".norm1", x, 1e-5, True)
        tp = self.r(self.lin(p + ".time_emb_proj", temb_act, scale))
        h = self.r(self.r(self.conv3(p + ".conv1", h)) + tp[:, :, None, None])
        h = self.gn(p + ".norm2", h, 1e-5, True)
        sc = x
        if (p + ".conv_shortcut.weight") in self.sd:
            sc = self.r(self.conv1(p + ".conv_shortcut", x, scale))
        return self.r(self.r(self.conv3(p + ".conv2", h)) + sc)

    def attn(self, p, x, ctx, scale, heads, residual):
        c = x if ctx is None else ctx
        q = self.r(self.lin(p + ".to_q", x, scale, bias=False))
        k = self.r(self.lin(p + ".to_k", c, scale, bias=False))
        v = self.r(self.lin(p + ".to_v", c, scale, bias=False))
        B, N, C = q.shape
        d = C // heads
        qh, kh, vh = [t.view(B, -1, heads, d).transpose(1, 2) for t in (q, k, v)]
        a = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, dim=-1)
        o = self.r((a @ vh).transpose(1, 2).reshape(B, N, C))
        return self.r(self.r(self.lin(p + ".to_out.0", o, scale)) + residual)

    def transformer(self, p, x, ctx, scale):
        heads = self.cfg["attention_heads"]
        B, C, H, W = x.shape
        h = self.gn(p + ".norm", x, 1e-6, False)
        h = self.r(self.conv1(p + ".proj_in", h, scale))
        t = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        tb = p + ".transformer_blocks.0"
        t = self.attn(tb + ".attn1", self.ln(tb + ".norm1", t), None, scale, heads, t)
        t = self.attn(tb + ".attn2", self.ln(tb + ".norm2", t), ctx, scale, heads, t)
        f = self.r(self.lin(tb + ".ff.net.0.proj", self.ln(tb + ".norm3", t), scale))
        hh, g = f.chunk(2, dim=-1)
        f = self.r(hh * F.gelu(g))
        t = self.r(self.r(self.lin(tb + ".ff.net.2", f, scale)) + t)
        h = t.reshape(B, H, W, C).permute(0, 3, 1, 2)
        return self.r(self.r(self.conv1(p + ".proj_out", h, scale)) + x)

    def forward(self, sample, t, ctx, scale=1.0):
        cfg = self.cfg
        boc = cfg["block_out_channels"]
        L = cfg["layers_per_block"]
        down_attn = cfg.get("down_attn", (True, True, True, False))
        up_attn = cfg.get("up_attn", (False, True, True, True))
        sample, ctx = self.r(sample.float()), self.r(ctx.float())
        temb = self.r(timestep_embedding(t, boc[0]))
        e = self.r(self.lin("time_embedding.linear_1", temb, scale))
        emb = self.r(self.lin("time_embedding.linear_2", self.r(F.silu(e)), scale))
        ta = self.r(F.silu(emb))
        h = self.r(self.conv3("conv_in", sample))
        skips = [h]
        for i in range(len(boc)):
            for j in range(L):
                h = self.resnet(f"down_blocks.{i}.resnets.{j}", h, ta, scale)
                if down_attn[i]:
                    h = self.transformer(f"down_blocks.{i}.attentions.{j}", h, ctx, scale)
                skips.append(h)
            if i != len(boc) - 1:
                h = self.r(self.conv3(f"down_blocks.{i}.downsamplers.0.conv", h, stride=2))
                skips.append(h)
        h = self.resnet("mid_block.resnets.0", h, ta, scale)
        h = self.transformer("mid_block.attentions.0", h, ctx, scale)
        h = self.resnet("mid_block.resnets.1", h, ta, scale)
        for i in range(len(boc)):
            for j in range(L + 1):
                h = torch.cat([h, skips.pop()], dim=1)
                h = self.resnet(f"up_blocks.{i}.resnets.{j}", h, ta, scale)
                if up_attn[i]:
                    h = self.transformer(f"up_blocks.{i}.attentions.{j}", h, ctx, scale)
            if i != len(boc) - 1:
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
                h = self.r(self.conv3(f"up_blocks.{i}.upsamplers.0.conv", h))
        h = self.gn("conv_norm_out", h, 1e-5, True)
        return self.r(self.conv3("conv_out", h))


def ppft_loss(sd, cfg, lora, E, msg, z, wm, eps, t, ctx, bf16=False):
    """ppft_train.py:989-1051 with injected inputs: returns (loss, pred, clean, S)."""
    acp = alphas_cumprod()
    S = mapper(msg, E)
    x_t = add_noise(z, eps, t, acp)
    x_t_wm = add_noise(z + wm, eps, t, acp)
    net = UNetOracle(sd, cfg, lora, bf16)
    with torch.no_grad():
        clean = net.forward(x_t, t, ctx, scale=None)  # == the reference's all-zero scale (LoRA term is exactly 0)
    pred = net.forward(x_t_wm, t, ctx, scale=S)
    loss = F.mse_loss(pred.float(), clean.float(), reduction="mean")
    return loss, pred, clean, S


def lora_state_dict_keys(unet_keys):
    """Checkpoint key mapping of ppft_train.py:443-471 (+ the "unet." prefix added by save_lora_weights)."""
    out = []
    for key in unet_keys:
        k = key.replace(".proj_in", ".proj_in.lora").replace(".proj_out", ".proj_out.lora")
        k = k.replace(".to_q", ".processor.to_q_lora").replace(".to_k", ".processor.to_k_lora")
        k = k.replace(".to_v", ".processor.to_v_lora").replace(".to_out.0", ".processor.to_out_lora")
        if "ff" in k:
            k = k + ".lora"
        out += [f"unet.{k}.down.weight", f"unet.{k}.up.weight"]
    return out


# ------------------------------------------------------------------------------------ distortions + metrics
def jpeg_zigzag_mask(keep):
    """8x8 keep-mask: first `keep` positions of sorted((i,j), key=(i+j, -j if (i+j) odd else j)) (jpeg_compression.py:31-41)."""
    order = sorted(((i, j) for i in range(8) for j in range(8)), key=lambda p: (p[0] + p[1], -p[1] if (p[0] + p[1]) % 2 else p[1]))
    m = torch.zeros(8, 8)
    for i, j in order[:keep]:
        m[i, j] = 1.0
    return m


def jpeg_compression(x, keep=(25, 9, 9)):
    """utils/noise_layers/jpeg_compression.py:127-162 on NCHW fp32: zero-pad to multiples of 8, rgb2yuv (:53-57),
    blockwise DCT with dct_coeff (:44-45), zig-zag keep-mask, IDCT with idct_coeff (:48-50), yuv2rgb (:60-64), crop."""
    B, C, H, W = x.shape
    ph, pw = (8 - H % 8) % 8, (8 - W % 8) % 8
    xp = F.pad(x, (0, pw, 0, ph))
    rgb2yuv = torch.tensor([[0.299, 0.587, 0.114], [-0.14713, -0.28886, 0.436], [0.615, -0.51499, -0.10001]])
    yuv2rgb = torch.tensor([[1.0, 0.0, 1.13983], [1.0, -0.39465, -0.58060], [1.0, 2.03211, 0.0]])
    n = torch.arange(8, dtype=torch.float64)
    Tm = torch.cos(math.pi / 8 * (n[None, :] + 0.5) * n[:, None]).float()                       # T[k][n]
    Um = (((n[None, :] == 0).double() * -0.5 + torch.cos(math.pi / 8 * (n[:, None] + 0.5) * n[None, :])) *
          math.sqrt(1 / 16)).float()                                                            # U[k][n] = idct(n,k)
    yuv = torch.einsum("ij,bjhw->bihw", rgb2yuv, xp)
    Hp, Wp = yuv.shape[2:]
    blocks = yuv.reshape(B, 3, Hp // 8, 8, Wp // 8, 8).permute(0, 1, 2, 4, 3, 5)                # [B,3,bh,bw,8,8]
    coef = Tm @ blocks @ Tm.T
    mask = torch.stack([jpeg_zigzag_mask(k) for k in keep])[None, :, None, None]
    rec = Um @ (coef * mask) @ Um.T
    rec = rec.permute(0, 1, 2, 4, 3, 5).reshape(B, 3, Hp, Wp)
    out = torch.einsum("ij,bjhw->bihw", yuv2rgb, rec)
    return out[:, :, :H, :W]


def calculate_fpr(tau, k):
    """evaluation/utils_eval.py:131-134."""
    return sum(math.comb(k, i) for i in range(tau + 1, k + 1)) / (2 ** k)


def get_threshold(k, fpr):
    """evaluation/utils_eval.py:136-140."""
    tau = 0
    while calculate_fpr(tau, k) > fpr:
        tau += 1
    return tau


def ddim_step(x, eps_u, eps_c, t, t_prev, guidance, acp=None):
    """DDIM (eta 0, epsilon prediction, no clipping) + classifier-free guidance; recalled diffusers semantics
    (SURVEY.md App. C): alpha_prev = alphas_cumprod[t_prev] if t_prev >= 0 else alphas_cumprod[0]."""
    acp = alphas_cumprod().double() if acp is None else acp
    a_t = acp[t]
    a_p = acp[t_prev] if t_prev >= 0 else acp[0]
    eps = eps_u.double() + guidance * (eps_c.double() - eps_u.double())
    x0 = (x.double() - (1 - a_t).sqrt() * eps) / a_t.sqrt()
    return (a_p.sqrt() * x0 + (1 - a_p).sqrt() * eps).float()


def leading_timesteps(n, T=1000, steps_offset=1):
    """diffusers' timestep_spacing="leading": k * (T // (n + 1)) for k = n .. 1, plus steps_offset -- the grid DPMSolverMultistepScheduler
    and UniPCMultistepScheduler build from the SD-1.5 scheduler config (recalled; UNPINNED)."""
    ratio = T // (n + 1)
    return [k * ratio + steps_offset for k in range(n, 0, -1)]


def dpmpp2m_steps(num_inference_steps, num_train_timesteps=1000, acp=None):
    """Own restatement of the DPM-Solver++(2M) trajectory (independent of aqualora_amd.inference.dpmpp2m_schedule): timesteps =
    `leading_timesteps` (diffusers' "leading" spacing with steps_offset 1, as the SD-1.5 scheduler config sets it), data-prediction multistep update of Lu et al. 2022
    (arXiv:2211.01095, Alg. 2) in half-log-SNR lambda = log(alpha / sigma):
        h = lambda_next - lambda_t;  x_next = (sigma_next / sigma_t) x - alpha_next (e^{-h} - 1) D,
        D = x0_t (first step; last step too when n < 15: diffusers' lower_order_final) else x0_t + (x0_t - x0_prev) / (2 r0),
        r0 = (lambda_t - lambda_prev) / h;   the final step lands on alphas_cumprod[0] (final_sigmas_type "zero" is not used by
    the reference's rob-finetune sampler, rob_enhance_finetune.py:993,1012).  Returns [(t, t_next)] and the lambda function.
    UNPINNED (diffusers is not on disk)."""
    import math
    import numpy as np
    acp = (alphas_cumprod() if acp is None else acp).double().numpy()
    ts = leading_timesteps(num_inference_steps, num_train_timesteps)
    lam = lambda t: 0.5 * math.log(acp[t] / (1.0 - acp[t]))  # noqa: E731
    return ts, lam, acp


def dpmpp2m_sample(eps_model, x, schedule, guidance=1.0):
    """DPM-Solver++(2M) sampler.  ``schedule`` is either the number of inference steps (the oracle then builds the whole
    trajectory itself from `dpmpp2m_steps` -- nothing of the product is involved) or, for backward compatibility, a list of
    (t, alpha_t, sigma_t, a, b, c) rows.  ``eps_model(x, t) -> (eps_uncond, eps_cond)``.  UNPINNED (see dpmpp2m_steps)."""
    import math
    x = x.double()
    x0_prev = torch.zeros_like(x)
    if isinstance(schedule, int):
        n = schedule
        ts, lam, acp = dpmpp2m_steps(n)
        for i, t in enumerate(ts):
            nxt = ts[i + 1] if i + 1 < n else 0
            al, sg = math.sqrt(acp[t]), math.sqrt(1.0 - acp[t])
            al_n, sg_n = math.sqrt(acp[nxt]), math.sqrt(1.0 - acp[nxt])
            eu, ec = eps_model(x.float(), t)
            eps = eu.double() + guidance * (ec.double() - eu.double())
            x0 = (x - sg * eps) / al
            h = lam(nxt) - lam(t)
            first_order = i == 0 or (i == n - 1 and n < 15)
            if first_order:
                D = x0
            else:
                r0 = (lam(t) - lam(ts[i - 1])) / h
                D = x0 + (x0 - x0_prev) / (2.0 * r0)
            x = (sg_n / sg) * x - al_n * math.expm1(-h) * D
            x0_prev = x0
        return x.float()
    for t, al, sg, a, b, c in schedule:
        eu, ec = eps_model(x.float(), t)
        eps = eu.double() + guidance * (ec.double() - eu.double())
        x0 = (x - sg * eps) / al
        x = a * x + b * x0 + c * x0_prev
        x0_prev = x0
    return x.float()

# ---------------------------------------------------------------------------------------------------------------------
# Sigma-space samplers (evaluation/utils_eval.py:83-101: euler / heun / lms / kdpm2), independent restatement of the step rules
# for tests/test_samplers.py.  UNPINNED (diffusers absent): Karras et al. 2022 / k-diffusion, SD-1.5 scheduler conventions.
def k_sigmas_oracle(num_inference_steps, acp):
    """(timesteps, sigmas + [0]) with leading spacing and steps_offset 1; acp: float64 alphas_cumprod [1000]."""
    ratio = 1000 // num_inference_steps
    ts = [i * ratio + 1 for i in range(num_inference_steps)][::-1]
    sig = [float(((1 - acp[t]) / acp[t]) ** 0.5) for t in ts] + [0.0]
    return ts, sig


def lms_coeff_oracle(sig, order, i, j):
    """Exact integral of the Lagrange basis polynomial (polynomial arithmetic with numpy.poly1d instead of quadrature)."""
    import numpy as np
    poly = np.poly1d([1.0])
    for k in range(order):
        if k == j:
            continue
        poly = poly * np.poly1d([1.0, -sig[i - k]]) / (sig[i - j] - sig[i - k])
    P = poly.integ()
    return float(P(sig[i + 1]) - P(sig[i]))


def k_sample_oracle(eps_fn, x, ts, sig, sampler):
    """eps_fn(x, sigma, t) -> eps in k-space.  Python loops, float64-friendly."""
    import math
    hist = []
    for i in range(len(ts)):
        s, sn = sig[i], sig[i + 1]
        d = eps_fn(x, s, ts[i])
        if sampler == "euler":
            x = x + (sn - s) * d
        elif sampler == "heun":
            xp = x + (sn - s) * d
            x = xp if sn == 0 else x + 0.5 * (sn - s) * (d + eps_fn(xp, sn, ts[i + 1]))
        elif sampler == "kdpm2":
            if sn == 0:
                x = x + (sn - s) * d
            else:
                sm = math.sqrt(s * sn)            # geometric mean == exp of the mean log
                x = x + (sn - s) * eps_fn(x + (sm - s) * d, sm, None)
        elif sampler == "lms":
            hist = ([d] + hist)[:4]
            x = x + sum(lms_coeff_oracle(sig, len(hist), i, j) * hist[j] for j in range(len(hist)))
        elif sampler == "kdpm2a":   # k-diffusion sample_dpm_2_ancestral, eta = 1; noise_fn rides on eps_fn.noise (test plumbing)
            if sn == 0:
                x = x - s * d
            else:
                up = min(sn, math.sqrt(sn * sn * (s * s - sn * sn) / (s * s)))
                down = math.sqrt(sn * sn - up * up)
                sm = math.sqrt(s * down)
                x = x + (down - s) * eps_fn(x + (sm - s) * d, sm, None)
                x = x + up * eps_fn.noise(i, x)
        else:
            raise ValueError(sampler)
    return x


def dpms_sde_oracle(eps_fn, x, ts, sig, brownian):
    """DPM-Solver++ SDE (``dpms_sde`` of evaluation/utils_eval.py:95-96: diffusers' DPMSolverSDEScheduler, itself k-diffusion's
    ``sample_dpmpp_sde`` with r = 1/2), written the way the scheduler words it -- in t = -log sigma with expm1, data predictions, and a
    noise sampler called with (sigma_from, sigma_to) -- independently of the sigma-space coefficient program of
    aqualora_amd/ksamplers.py.  UNPINNED (diffusers and torchsde absent): Lu et al. 2022 (arXiv:2211.01095) section 4 / k-diffusion.
    eps_fn(x, sigma, t) -> eps;  brownian(sigma_from, sigma_to) -> (W(sigma_to) - W(sigma_from)) / sqrt(|sigma_to - sigma_from|) of ONE
    Brownian path W (what BrownianTreeNoiseSampler returns; the overall sign convention is immaterial, the nesting is not)."""
    import math

    def t_fn(sigma):
        return -math.log(sigma)

    def sigma_fn(t):
        return math.exp(-t)

    def ancestral(sigma_from, sigma_to):
        up = min(sigma_to, math.sqrt(sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2))
        return math.sqrt(sigma_to ** 2 - up ** 2), up

    for i in range(len(ts)):
        s, sn = sig[i], sig[i + 1]
        denoised = x - s * eps_fn(x, s, ts[i])
        if sn == 0:
            x = x + ((x - denoised) / s) * (sn - s)
            continue
        t, t_next = t_fn(s), t_fn(sn)
        t_mid = t + 0.5 * (t_next - t)
        # first stage: to the midpoint
        down, up = ancestral(sigma_fn(t), sigma_fn(t_mid))
        t_anc = t_fn(down)
        x_mid = (sigma_fn(t_anc) / sigma_fn(t)) * x - math.expm1(t - t_anc) * denoised
        x_mid = x_mid + brownian(sigma_fn(t), sigma_fn(t_mid)) * up
        denoised_mid = x_mid - sigma_fn(t_mid) * eps_fn(x_mid, sigma_fn(t_mid), None)
        # second stage: the whole step from the ORIGINAL sample with the midpoint's data prediction
        down, up = ancestral(sigma_fn(t), sigma_fn(t_next))
        t_anc = t_fn(down)
        x = (sigma_fn(t_anc) / sigma_fn(t)) * x - math.expm1(t - t_anc) * denoised_mid
        x = x + brownian(sigma_fn(t), sigma_fn(t_next)) * up
    return x


def plms_oracle(eps_fn, x, n_steps, acp):
    """Independent restatement of the PLMS sampler as diffusers' PNDMScheduler runs it for the SD-1.5 config (skip_prk_steps, leading
    spacing with offset 1, final alpha = alphas_cumprod[0]): written as explicit phases instead of the scheduler's counter logic.
    Phase 1 (Heun-like warm-up at the first timestep), then Adams-Bashforth of growing order 2, 3, 4 on the eps history."""
    import math
    ratio = 1000 // n_steps
    grid = [1 + i * ratio for i in range(n_steps)][::-1]          # t_0 > t_1 > ... > t_{N-1} = 1

    def transfer(x_, t, tp, e):
        a_t = float(acp[t])
        a_p = float(acp[tp]) if tp >= 0 else float(acp[0])
        return math.sqrt(a_p / a_t) * x_ - (a_p - a_t) * e / (a_t * math.sqrt(1 - a_p) + math.sqrt(a_t * (1 - a_t) * a_p))
    e0 = eps_fn(x, grid[0])
    x1 = transfer(x, grid[0], grid[1], e0)
    e1 = eps_fn(x1, grid[1])
    x = transfer(x, grid[0], grid[1], 0.5 * (e0 + e1))              # corrected first step
    hist = [e0]
    coeffs = {2: (1.5, -0.5), 3: (23 / 12, -16 / 12, 5 / 12), 4: (55 / 24, -59 / 24, 37 / 24, -9 / 24)}
    for k in range(1, n_steps):
        hist = (hist + [eps_fn(x, grid[k])])[-4:]
        c = coeffs[len(hist)]
        e = sum(ci * hi for ci, hi in zip(c, hist[::-1]))
        x = transfer(x, grid[k], grid[k] - ratio, e)
    return x


# ---------------------------------------------------------------------------------------------------------------------
# DPM-Solver++ single-step (2S) and UniPC (evaluation/utils_eval.py:93-94,101-102), written the way the schedulers keep their state
# (lists of data predictions and timesteps), independent of the coefficient programs of aqualora_amd/ksamplers.py.  UNPINNED
# (diffusers absent): Lu et al. 2022 (arXiv:2211.01095) Alg. 1 for 2S with the midpoint rule; Zhao et al. 2023 (arXiv:2302.04867)
# UniP / UniC with B(h) = e^h - 1 ("bh2"), data prediction, order 2, lower_order_final.  eps_fn(x, t) -> (guided) eps at integer t.
def _dpm_grid(n_steps, acp, leading=False):
    import numpy as np
    a = np.asarray(acp, dtype=np.float64)
    ts = leading_timesteps(n_steps) if leading else [int(v) for v in np.linspace(0, 999, n_steps + 1).round()[::-1][:-1]]
    al, sg = np.sqrt(a), np.sqrt(1.0 - a)
    return ts, al, sg, np.log(al / sg)


def dpms_singlestep_oracle(eps_fn, x, n_steps, acp):
    import math
    ts, al, sg, lam = _dpm_grid(n_steps, acp)
    orders = [1, 2] * (n_steps // 2) + ([1] if n_steps % 2 else [])
    outs, saved = [], None
    for k, s0 in enumerate(ts):
        t = ts[k + 1] if k + 1 < n_steps else 0
        m0 = (x - sg[s0] * eps_fn(x, s0)) / al[s0]
        outs = (outs + [m0])[-2:]
        if orders[k] == 1:
            saved = x
            h = lam[t] - lam[s0]
            x = (sg[t] / sg[s0]) * saved - al[t] * math.expm1(-h) * m0
        else:
            s1 = ts[k - 1]
            m1 = outs[-2]
            h, h_0 = lam[t] - lam[s1], lam[s0] - lam[s1]
            r0 = h_0 / h
            D0, D1 = m1, (m0 - m1) / r0
            x = (sg[t] / sg[s1]) * saved - al[t] * math.expm1(-h) * D0 - 0.5 * al[t] * math.expm1(-h) * D1
    return x


def unipc_oracle(eps_fn, x, n_steps, acp, solver_order=2, lower_order_final=True):
    import math
    import numpy as np
    ts, al, sg, lam = _dpm_grid(n_steps, acp, leading=True)   # UniPC honours the config's timestep_spacing; the single-step class cannot

    def bh(order, rks, hh):
        h_phi_1 = math.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1.0
        B_h = math.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append([rk ** (i - 1) for rk in rks])
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1.0 / fact
        return np.array(R), np.array(b), h_phi_1, B_h

    outs, tlist = [], []
    last_sample, this_order, lower = None, 1, 0
    for k, s in enumerate(ts):
        t_next = ts[k + 1] if k + 1 < n_steps else 0
        m_t = (x - sg[s] * eps_fn(x, s)) / al[s]
        if k > 0:    # UniC: correct the sample at s with the new prediction
            s0, m0, order = tlist[-1], outs[-1], this_order
            h = lam[s] - lam[s0]
            rks, D1s = [], []
            for i in range(1, order):
                si, mi = tlist[-(i + 1)], outs[-(i + 1)]
                rk = (lam[si] - lam[s0]) / h
                rks.append(rk)
                D1s.append((mi - m0) / rk)
            rks.append(1.0)
            R, b, h_phi_1, B_h = bh(order, rks, -h)
            rhos = np.array([0.5]) if order == 1 else np.linalg.solve(R, b)
            x_ = (sg[s] / sg[s0]) * last_sample - al[s] * h_phi_1 * m0
            corr = sum(float(rhos[j]) * D1s[j] for j in range(len(D1s))) if D1s else 0.0
            x = x_ - al[s] * B_h * (corr + float(rhos[-1]) * (m_t - m0))
        outs, tlist = (outs + [m_t])[-solver_order:], (tlist + [s])[-solver_order:]
        this_order = min(min(solver_order, n_steps - k) if lower_order_final else solver_order, lower + 1)
        last_sample = x
        # UniP: predict the sample at t_next
        s0, m0 = tlist[-1], outs[-1]
        h = lam[t_next] - lam[s0]
        rks, D1s = [], []
        for i in range(1, this_order):
            si, mi = tlist[-(i + 1)], outs[-(i + 1)]
            rk = (lam[si] - lam[s0]) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(1.0)
        R, b, h_phi_1, B_h = bh(this_order, rks, -h)
        x_ = (sg[t_next] / sg[s0]) * x - al[t_next] * h_phi_1 * m0
        if D1s:
            rhos_p = np.array([0.5]) if this_order == 2 else np.linalg.solve(R[:-1, :-1], b[:-1])
            x_ = x_ - al[t_next] * B_h * sum(float(rhos_p[j]) * D1s[j] for j in range(len(D1s)))
        x = x_
        lower += 1
    return x
