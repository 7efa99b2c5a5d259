"""hashmod-mi355: MI355X-native (gfx950) hot path of HashModNFFBanks-IDR.

Host side = Python on PyTorch-ROCm mirroring the reference's module tree
(``model.implicit_differentiable_renderer.IDRNetwork`` ...); compute = hand-written HIP kernels
in libhashmod.so behind the C ABI of include/hashmod.h.  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"
