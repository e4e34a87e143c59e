"""Helpers to read the committed golden fixtures (tests/golden/*.npz, made by make_golden.py)."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    arrays, sd, grads = {}, {}, {}
    for k in z.files:
        if k == "meta":
            continue
        t = torch.from_numpy(z[k])
        if k.startswith("sd/"):
            sd[k[3:]] = t
        elif k.startswith("g/"):
            grads[k[2:]] = t
        else:
            arrays[k] = t
    return meta, arrays, sd, grads


def names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))
