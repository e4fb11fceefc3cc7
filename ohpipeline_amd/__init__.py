"""ohpipeline_amd -- MI355X-native PCM hot path of ohPipeline (ramp, attenuation, sample-format
conversion, polyphase sample-rate conversion) behind the C ABI of include/ohgpu.h.

csrc/   HIP kernels (gfx950) and the C ABI
host/   C++ host adapter mirroring the reference's Msg / IPcmProcessor surface
capi.py ctypes binding used by tests and bench (no CPU fallback)
"""
__all__ = ["capi", "build"]
