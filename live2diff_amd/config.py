"""Static description of the Live2Diff streaming UNet (SD-1.5 + AnimateDiff-style motion modules).

Mirrors the constructor arguments of the reference's `UNet3DConditionStreamingModel`
(reference: live2diff/animatediff/models/unet_depth_streaming.py:39-88 and
configs/base_config.yaml:6-28) restricted to the values the Live2Diff path actually uses.
"""
from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    cross_attention_dim: int = 768          # SD-1.5 CLIP-L width
    num_heads: int = 8                      # SD-1.5 `attention_head_dim: 8` is used as the head COUNT
    #                                         (unet_blocks_streaming.py:340-343)
    norm_num_groups: int = 32
    norm_eps: float = 1e-5                  # resnet / conv_norm_out GroupNorm eps
    transformer_norm_eps: float = 1e-6      # attention.py:57, motion_module.py:181
    mapping_channels: Tuple[int, ...] = (16, 32, 96, 256)   # resnet.py:26
    # temporal (motion module) parameters, configs/base_config.yaml:14-28
    temporal_heads: int = 8
    temporal_max_len: int = 24              # temporal_position_encoding_max_len
    window_size: int = 16                   # L = sink + rolling
    sink_size: int = 8                      # == WARMUP_FRAMES in the reference pipeline

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @property
    def num_levels(self) -> int:
        return len(self.block_out_channels)


def sd15_config(window_size: int = 16, sink_size: int = 8, temporal_max_len: int = 0) -> UNetConfig:
    """The real model: SD-1.5 widths, 1 277.7 M parameters."""
    return UNetConfig(window_size=window_size, sink_size=sink_size,
                      temporal_max_len=max(temporal_max_len, 24, window_size))


def tiny_config(window_size: int = 16, sink_size: int = 8, channels=(64, 128, 256, 256),
                cross_attention_dim: int = 96) -> UNetConfig:
    """Same topology at test scale (seconds on CPU)."""
    return UNetConfig(block_out_channels=tuple(channels), cross_attention_dim=cross_attention_dim,
                      window_size=window_size, sink_size=sink_size,
                      temporal_max_len=max(24, window_size))


def motion_module_layout(cfg: UNetConfig, h: int, w: int) -> List[Tuple[int, int, int, int]]:
    """(channels, h, w, level) of every temporal attention, in `motion_module_idx` order.

    Reference: `set_info_for_attn` walks down blocks -> mid -> up blocks in module order and numbers
    every temporal attention depth-first (unet_depth_streaming.py:252-281); each motion module holds
    two attentions (base_config.yaml:18). Down block i has `layers_per_block` motion modules at
    resolution 2^-i; up block i has `layers_per_block + 1` at resolution 2^-(3-i).
    """
    out = []
    nl = cfg.num_levels
    hh, ww = h, w
    for i, c in enumerate(cfg.block_out_channels):
        for _ in range(cfg.layers_per_block):
            out += [(c, hh, ww, i)] * 2
        if i != nl - 1:
            hh, ww = hh // 2, ww // 2
    rev = list(reversed(cfg.block_out_channels))
    for i, c in enumerate(rev):
        for _ in range(cfg.layers_per_block + 1):
            out += [(c, hh, ww, nl - 1 - i)] * 2
        if i != nl - 1:
            hh, ww = hh * 2, ww * 2
    return out
