"""Helpers shared by the tests: numpy <-> device tensors (torch is only the device-memory carrier)."""
import numpy as np
import torch

_VIEW = {np.dtype(np.uint16): np.int16, np.dtype(np.uint32): np.int32}


def to_dev(a):
    a = np.ascontiguousarray(a)
    v = a.view(_VIEW[a.dtype]) if a.dtype in _VIEW else a
    return torch.from_numpy(v.copy()).cuda()


def to_np(t, dtype):
    return t.cpu().numpy().view(dtype)


def rand_mask(rng, shape):
    m = rng.integers(0, 256, shape, dtype=np.uint8)
    sel = rng.random(shape)
    m[sel < 0.3] = 0
    m[sel > 0.7] = 255
    m[(sel > 0.3) & (sel < 0.36)] = rng.integers(250, 256)
    m[(sel > 0.36) & (sel < 0.4)] = rng.integers(1, 5)
    return m


OPAS = [0, 2, 3, 4, 100, 127, 128, 250, 251, 252, 253, 254, 255]
