"""Watermark extraction metrics (reference evaluation/utils_eval.py:131-140, 193-211): exact integer / host arithmetic.

``bits = argmax(logits, -1)`` is the decoder read-out used at ppft_train.py:1179 and utils_eval.py:195; bit accuracy is
the mean match against the ground-truth message and TPR the fraction of images whose accuracy reaches
``tau = get_threshold(k, fpr) / k``.
"""
from math import comb

import torch


def calculate_fpr(tau, k):
    return sum(comb(k, i) for i in range(tau + 1, k + 1)) / (2 ** k)


def get_threshold(k, fpr):
    tau = 0
    while calculate_fpr(tau, k) > fpr:
        tau += 1
    return tau


def extract_bits(logits):
    """[B, bits, 2] decoder logits -> [B, bits] int64 message bits."""
    return torch.argmax(logits, dim=-1)


def bit_accuracy(bits, msg_gt):
    """per-image fraction of matching bits; bits/msg_gt: [B, k] (0/1)."""
    return (bits.long() == msg_gt.long()).double().mean(dim=-1)


def tpr_at_fpr(bits, msg_gt, fpr=1e-6):
    k = bits.shape[-1]
    tau = get_threshold(k, fpr) / k
    acc = bit_accuracy(bits, msg_gt)
    return float(acc.mean()), float((acc >= tau).double().mean())
