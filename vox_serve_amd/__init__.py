"""vox_serve_amd — MI355X-native hot path of the vox-serve speech-LM serving loop.

Drop-in modules (same names / signatures as the reference package `vox_serve`):
    flashinfer_utils, sampling, requests, model, tokenizer, worker
Native code: libvoxhip.so (csrc/, C ABI in include/voxhip.h).
"""
__version__ = "0.1.0"
