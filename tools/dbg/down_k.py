import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aqualora_amd import _lib as L
torch.manual_seed(0)
for M in (154, 512):
  for K in (32, 64, 96, 128, 160, 192, 224, 256, 288, 320, 480):
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    A = (torch.randn(32, K, device="cuda") / 8).to(torch.bfloat16)
    S = (torch.randn(2, 32, device="cuda")).to(torch.bfloat16)
    T = torch.full((M, 32), float("nan"), device="cuda", dtype=torch.bfloat16); Ts = T.clone()
    L.call("aql_lora_down", L.ptr(X), K, M, K, L.ptr(A), 32, L.ptr(S), M // 2, L.ptr(T), L.ptr(Ts), None, None, L.stream_ptr())
    ref = X.float() @ A.float().t()
    e = float((T.float() - ref).abs().max() / ref.abs().max())
    # with Tref / dS
    Tref = torch.randn(M, 32, device="cuda").to(torch.bfloat16); dS = torch.zeros(2, 32, device="cuda")
    T2 = torch.empty_like(T); Ts2 = torch.empty_like(T)
    L.call("aql_lora_down", L.ptr(X), K, M, K, L.ptr(A), 32, L.ptr(S), M // 2, L.ptr(T2), L.ptr(Ts2), L.ptr(Tref), L.ptr(dS), L.stream_ptr())
    dref = (T2.float() * Tref.float()).view(2, M // 2, 32).sum(1)
    print(f"M={M} K={K}: T err {e:.2e} finite {bool(torch.isfinite(T.float()).all())}; dS err {float((dS - dref).abs().max() / dref.abs().max()):.2e}", flush=True)
