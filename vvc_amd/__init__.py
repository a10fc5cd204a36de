"""vvc_amd -- LVGL 8.x software pixel path (fill / alpha blend / RGB565 conversion / flush rotate) as batched HIP
kernels for AMD Instinct MI355X, behind a C ABI (include/lvgpu.h) that keeps LVGL's plug-in surface.

`vvc_amd.lvgpu` is the host-side mirror of the reference's drawing interface; it needs the in-tree
vvc_amd/lib/liblvgpu.so (make -C vvc_amd/csrc) and fails loudly without it.
"""
from . import _lib  # noqa: F401

__all__ = ["lvgpu", "_lib"]


def __getattr__(name):
    if name == "lvgpu":
        import importlib
        return importlib.import_module(".lvgpu", __name__)
    raise AttributeError(name)
