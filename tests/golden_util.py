"""Loader for the committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    weights = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    rest = {k: z[k] for k in z.files if not k.startswith("w:")}
    return weights, rest
