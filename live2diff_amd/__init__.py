"""live2diff_amd -- MI355X-native (gfx950) backend for Live2Diff's per-frame streaming UNet step.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every device op on
the hot path is a hand-written HIP kernel in `csrc/`, reached through the C-ABI declared in
`include/l2d.h` (`libl2d_hip.so`).  There is no CPU / eager fallback: importing the ops without the
built library raises.
"""
from .config import UNetConfig, sd15_config, tiny_config, motion_module_layout  # noqa: F401

__version__ = "0.1.0"
