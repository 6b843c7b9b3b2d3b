"""ORACLE -- test infrastructure only.  Host restatement (float64, plain torch) of the arithmetic of one `aql_sampler_step` launch
(csrc/aql_elem.hip) and of the loop `ksamplers.SamplerMachine` runs around it, so that the coefficient PROGRAMS of
aqualora_amd/ksamplers.py (the samplers of evaluation/utils_eval.py:83-102 as lists of linear phases) can be checked on the CPU
against the direct restatements of the same samplers in oracle/ppft_oracle.py, without a GPU and without the U-Net."""
import torch


def step(state, coef, flag, e):
    """state: dict(x, aux, hist [4, ...], noise); one kernel launch with the guided prediction `e` (ignored when flag says so)."""
    g, cx, ca, ce, ch0, ch1, ch2, ch3, cn, nscale, psrc, pe = coef
    dst, push, nsrc, mode, src = flag
    no_eval, save = mode & 1, mode & 2
    x, aux, h = state["x"], state["aux"], state["hist"]
    e = torch.zeros_like(x) if no_eval else e
    if push:
        h = torch.stack([psrc * (aux if src else x) + pe * e, h[0], h[1], h[2]])
    if save:
        aux = x
    v = cx * x + ca * aux + ce * e + ch0 * h[0] + ch1 * h[1] + ch2 * h[2] + ch3 * h[3]
    if cn != 0.0:
        v = v + cn * state["noise"]
    if dst:
        aux = v
    else:
        x = v
    state.update(x=x, aux=aux, hist=h)
    return nscale * (aux if nsrc else x)


def run(program, eps_fn, latents, noise_fn=None):
    """``eps_fn(model_input, t) -> guided eps`` (the model sees the scaled input, like the U-Net).  Returns the final state."""
    x = latents.double() * program.init_scale
    st = dict(x=x, aux=torch.zeros_like(x), hist=torch.zeros((4,) + tuple(x.shape), dtype=torch.float64), noise=torch.zeros_like(x))
    uin = program.first_in_scale * x
    for i, ph in enumerate(program.phases):
        if ph.noise:        # noise buffer <- mix[0] * its content + mix[1] * fresh (dpms_sde: increments of one Brownian path)
            st["noise"] = ph.noise_mix[0] * st["noise"] + ph.noise_mix[1] * noise_fn(i, x).double()
        e = eps_fn(uin, ph.t).double()
        for coef, flag in ph.calls:
            uin = step(st, coef, flag, e)
    return st["x"]
