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


def load_scene(path):
    """A golden display list (oracle/gen_golden_scenes.py): LVGL's frame buffer and the recorded descriptors."""
    z = np.load(path)
    fb, meta, src, mask = z["fb"], z["meta"], z["src"], z["mask"]
    items, so, mo = [], 0, 0
    for x1, y1, x2, y2, color, opa, mode, has_src, has_mask in meta.tolist():
        w, h = x2 - x1 + 1, y2 - y1 + 1
        it = dict(area=(x1, y1, x2, y2), color=color, opa=opa, blend_mode=mode, src=None, mask=None)
        if has_src:
            it["src"] = src[so:so + w * h].reshape(h, w)
            so += w * h
        if has_mask:
            it["mask"] = mask[mo:mo + w * h].reshape(h, w)
            mo += w * h
        items.append(it)
    return fb, items
