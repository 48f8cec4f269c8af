"""Backbone registry — same seam as zonos/backbone/__init__.py:24-36.  Both reference keys resolve to the HIP
transformer backbone; the hybrid (Mamba2) architecture is not built yet."""
from ._hip import HipEngine, HipZonosBackbone

BACKBONES = {"hip": HipZonosBackbone, "torch": HipZonosBackbone}

__all__ = ["BACKBONES", "HipZonosBackbone", "HipEngine"]
