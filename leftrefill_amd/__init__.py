"""leftrefill_amd -- MI355X (gfx950) native implementation of LeftRefill's diffusion-sampling hot path.

csrc/    hand-written HIP kernels + the C ABI (include/leftrefill_hip.h)
_lib.py  ctypes binding (fails loudly when the .so is missing)
ops.py   torch-tensor front end (memory + stream plumbing only)
dropin/  `ldm.*` / `inpainting_ldm.*` modules with the reference's operator API, running the kernels
"""
__version__ = "0.1.0"
