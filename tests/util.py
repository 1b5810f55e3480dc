"""Shared helpers for the tests: synthetic stand-ins for *-unproj.h5 and golden loaders."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def synth(N, F, k=16, seed=2215, sp=0.6, sn=1.0, noise=0.15, label_mode="alternate"):
    """d = U^T z + eps clipped to [-1,1]; label 1 = match."""
    rng = np.random.default_rng(seed)
    U = np.linalg.qr(rng.standard_normal((F, k)))[0].T.astype(np.float32)
    if label_mode == "alternate":
        labels = (np.arange(N) % 2 == 0).astype(np.uint8)
    else:  # ragged: unbalanced, with a few labels that are neither 0 nor 1
        labels = (rng.random(N) < 0.37).astype(np.uint8)
        labels[rng.integers(0, N, max(1, N // 50))] = 2
    z = rng.standard_normal((N, k)).astype(np.float32)
    z *= np.where(labels[:, None] == 1, sp, sn).astype(np.float32)
    d = z @ U + noise * rng.standard_normal((N, F)).astype(np.float32)
    return np.clip(d, -1, 1).astype(np.float32), labels


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def relmax(a, b):
    """max |a-b| / max |b|"""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
