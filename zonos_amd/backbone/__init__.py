"""Backbone registry — same seam as zonos/backbone/__init__.py:24-36.  Every reference key ("torch", "mamba_ssm") resolves
to the HIP backbone, which serves both architectures (`supported_architectures = ["transformer", "hybrid"]`, like the
reference's MambaSSMZonosBackbone, _mamba_ssm.py:25)."""
from ._hip import HipEngine, HipZonosBackbone

BACKBONES = {"hip": HipZonosBackbone, "torch": HipZonosBackbone, "mamba_ssm": HipZonosBackbone}

__all__ = ["BACKBONES", "HipZonosBackbone", "HipEngine"]
