"""MI355X-native quantized-training hot path (adaptive quantisation, `adp_qtz` scripts).

Sub-modules
  _lib          ctypes binding of lib/libaq_hip.so (the C ABI declared in include/aq_hip.h)
  ops           tensor-level wrappers + autograd Functions over the HIP kernels
  layers        MyConv2d / MyLinear / Residual / PreResidual / ResNet / model_loss drop-ins
  adp_qtz_fan, adp_qtz_fix, adp_qtz_hint
                per-variant modules with the reference scripts' names and signatures
  train         training step, HIP-graph capture and data-parallel wrapper
"""
__all__ = ["_lib", "ops", "layers", "adp_qtz_fan", "adp_qtz_fix", "adp_qtz_hint", "train"]
