"""MI355X-native EQUSS hot path: DINO-ViT patch features -> expansion head -> product-quantisation
assign, behind the reference's own nn.Module surface.  The compute runs in hand-written HIP kernels
(csrc/) reached through a C ABI (include/equss_hip.h); PyTorch-ROCm supplies device memory, streams
and torch.distributed only."""
from . import configs, synth  # noqa: F401

__all__ = ["configs", "synth"]
