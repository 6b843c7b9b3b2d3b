"""Counter-based synthetic tensor generator.

Every value is a pure function of (seed, tensor name, flat index): splitmix64 -> four 16-bit fields summed
(Irwin-Hall n=4, ~normal) -> one fp32 multiply.  Integer arithmetic only up to the final multiply, so numpy on
the CPU, torch on the CPU and torch on an MI355X produce bit-identical tensors: the oracle, the golden
fixtures and the GPU runs all see the same weights and inputs without relying on any library RNG stream.
"""
import zlib

import numpy as np
import torch

_M64 = (1 << 64) - 1
_G = 0x9E3779B97F4A7C15
_C1 = 0xBF58476D1CE4E5B9
_C2 = 0x94D049BB133111EB
_IH_SCALE = 1.0 / (65536.0 * (4.0 / 12.0) ** 0.5)  # Irwin-Hall(4) of U16 -> unit variance


def _s64(u):
    """python int (mod 2^64) -> signed int64 value"""
    u &= _M64
    return u - (1 << 64) if u >= (1 << 63) else u


def _base(name, seed):
    return ((zlib.crc32(name.encode()) << 32) ^ ((seed * _G) & _M64)) & _M64


def _lsr(x, k):
    return (x >> k) & ((1 << (64 - k)) - 1)


def _mix_torch(x):
    x = x + _s64(_G)
    x = (x ^ _lsr(x, 30)) * _s64(_C1)
    x = (x ^ _lsr(x, 27)) * _s64(_C2)
    return x ^ _lsr(x, 31)


def normal(name, shape, std=1.0, seed=0, device="cpu", dtype=torch.float32):
    """~N(0, std^2) tensor, bit-identical on every backend."""
    n = 1
    for s in shape:
        n *= int(s)
    idx = torch.arange(n, dtype=torch.int64, device=device) + _s64(_base(name, seed))
    z = _mix_torch(idx)
    s = (z & 0xFFFF) + (_lsr(z, 16) & 0xFFFF) + (_lsr(z, 32) & 0xFFFF) + (_lsr(z, 48) & 0xFFFF)
    v = (s - 2 * 65535).to(torch.float32) * np.float32(std * _IH_SCALE).item()
    return v.view(*shape).to(dtype)


def normal_np(name, shape, std=1.0, seed=0):
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) + np.uint64(_base(name, seed))
        x = x + np.uint64(_G)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(_C1)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(_C2)
        z = x ^ (x >> np.uint64(31))
    s = ((z & np.uint64(0xFFFF)) + ((z >> np.uint64(16)) & np.uint64(0xFFFF)) + ((z >> np.uint64(32)) & np.uint64(0xFFFF))
         + (z >> np.uint64(48))).astype(np.int64)
    v = (s - 2 * 65535).astype(np.float32) * np.float32(std * _IH_SCALE)
    return v.reshape(shape)


def bits(name, shape, seed=0, device="cpu"):
    """Bernoulli(0.5) message bits as float {0,1}."""
    n = 1
    for s in shape:
        n *= int(s)
    idx = torch.arange(n, dtype=torch.int64, device=device) + _s64(_base(name, seed))
    return (_lsr(_mix_torch(idx), 63) & 1).to(torch.float32).view(*shape)


def randint(name, shape, high, seed=0, device="cpu"):
    n = 1
    for s in shape:
        n *= int(s)
    idx = torch.arange(n, dtype=torch.int64, device=device) + _s64(_base(name, seed))
    return (_lsr(_mix_torch(idx), 20) % high).view(*shape)
