"""mammo-clip-dissect_amd: the MI355X (gfx950) dissection core of Mammo-CLIP-Dissect.

Layout:
  csrc/         hand-written HIP kernels + the C ABI (include/mcd_hip.h) -> csrc/libmcd_hip.so
  _lib.py       ctypes binding of the C ABI (fails loudly when the library is missing)
  core.py       tensor-level wrappers (device pointers of torch tensors -> C ABI)
  pipeline.py   fused all-layer dissection + the image-sharded multi-GPU path
  concept_vit/  drop-in mirror of the reference's concept_vit/{similarity,utils,...}.py interface

There is no CPU implementation in this package and no fallback: every compute entry point
raises if the HIP library is missing or the tensors are not on a GPU.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
