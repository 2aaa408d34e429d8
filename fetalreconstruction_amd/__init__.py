"""MI355X-native slice-to-volume super-resolution hot path (drop-in for the reference's
reconstructionGPU2 `class Reconstruction`, reconstruction_cuda2.cuh:92-341)."""
__version__ = "0.1.0"
