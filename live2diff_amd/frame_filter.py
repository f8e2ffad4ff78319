"""Near-duplicate frame gate for `StreamAnimateDiffusionDepth.enable_similar_image_filter`.

Behavioural contract = the reference's `SimilarImageFilter` (live2diff/image_filter.py:7-50, enabled through
pipeline_stream_animation_depth.py:112-118, consulted at :631-635): a frame whose cosine similarity to the last frame that
was let through exceeds the threshold is dropped with probability `1 - (1 - cos) / (1 - threshold)` (0 when
threshold >= 1), at most `max_skip_frame + 1` times in a row; the gate answers with the frame itself or `None`.

Written for this backend: the two dot products and the norm come from ONE small device reduction (`torch.stack` of three
fp32 sums -> a single 12-byte read-back; the pass / drop decision is host control flow, so one sync per gated frame is the
floor), the remembered frame is a preallocated fp32 buffer that is overwritten in place, and the random source is
injectable (default: the `random` module, like the reference, so `random.seed` reproduces the reference's decisions --
tests/test_host_logic.py pins that against a trace captured from the reference class)."""
import random as _random
from typing import Optional

import torch


class SimilarImageFilter:
    def __init__(self, threshold: float = 0.98, max_skip_frame: float = 10, rng=None) -> None:
        self.threshold = threshold
        self.max_skip_frame = max_skip_frame
        self.skip_count = 0
        self._rng = rng if rng is not None else _random
        self._kept: Optional[torch.Tensor] = None      # fp32 copy of the last frame that passed
        self._kept_sq: Optional[torch.Tensor] = None   # its squared norm (device scalar)

    def set_threshold(self, threshold: float) -> None:
        self.threshold = threshold

    def set_max_skip_frame(self, max_skip_frame: float) -> None:
        self.max_skip_frame = max_skip_frame

    @property
    def prev_tensor(self) -> Optional[torch.Tensor]:
        return self._kept

    def _remember(self, flat: torch.Tensor, sq: Optional[torch.Tensor] = None) -> None:
        if self._kept is None or self._kept.shape != flat.shape or self._kept.device != flat.device:
            self._kept = flat.clone()
        else:
            self._kept.copy_(flat)
        self._kept_sq = (flat * flat).sum() if sq is None else sq

    def similarity(self, x: torch.Tensor, flat: Optional[torch.Tensor] = None) -> float:
        """cos(prev, x) with the reference's eps = 1e-6 clamp on each norm (torch.nn.CosineSimilarity semantics).  Deviation from the
        reference, on purpose: the three reductions run in fp32 whatever the frame's dtype (the reference's CosineSimilarity
        module computes in the input dtype, fp16 on the GPU); for frames within ~1e-3 of the threshold, where `skip_prob` is steep,
        the two can decide differently.  The golden trace (tests/golden/frame_filter.json) pins the fp32 decisions."""
        if flat is None:
            flat = x.detach().reshape(-1).float()
        stats = torch.stack([(self._kept * flat).sum(), (flat * flat).sum(), self._kept_sq]).tolist()
        dot, xsq, psq = stats
        return dot / (max(psq ** 0.5, 1e-6) * max(xsq ** 0.5, 1e-6))

    def __call__(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        flat = x.detach().reshape(-1).float()
        if self._kept is None:
            self._remember(flat)
            return x
        cos_sim = self.similarity(x, flat)
        draw = self._rng.uniform(0, 1)
        skip_prob = 0.0 if self.threshold >= 1 else max(0.0, 1.0 - (1.0 - cos_sim) / (1.0 - self.threshold))
        if skip_prob < draw:                       # let it through and make it the new comparison frame
            self._remember(flat)
            return x
        if self.skip_count > self.max_skip_frame:  # too many drops in a row: force one through
            self.skip_count = 0
            self._remember(flat)
            return x
        self.skip_count += 1
        return None
