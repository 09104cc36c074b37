"""rvcx -- MI355X-native RVC v2 inference hot path (HuBERT -> RMVPE -> FAISS blend ->
TextEncoder / flow / NSF-HiFi-GAN) behind the reference's ``rvc.infer`` entry points.

Sub-modules
  _lib       ctypes binding of the C-ABI in include/rvcx.h (librvcx.so, HIP/gfx950)
  weights    checkpoint dict -> tensor table handed across the C-ABI
  synthetic  deterministic synthetic checkpoints / clips (no real weights exist offline)
  infer      drop-in mirror of rvc/infer/{infer,pipeline}.py (Config, load_hubert, get_vc,
             rvc_infer, VC)
Nothing here falls back to a CPU implementation: if librvcx.so is missing or the GPU is
absent the product path raises.
"""
__version__ = "0.1.0"
