"""vqa_amd -- MI355X-native fusion hot path of the hanleju/VQA model family.

Public surface mirrors the reference: fusion modules (vqa_amd.fusion), the model wrapper
(vqa_amd.model), the DP-SGD engine (vqa_amd.dp) and MIA scoring (vqa_amd.mia), all running on
the HIP kernels of libvqafuse.so through the C ABI in include/vqaf.h.
"""
from ._lib import LIB_PATH, VqafError, load  # noqa: F401

__version__ = "0.1.0"
