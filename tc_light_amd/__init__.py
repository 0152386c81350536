"""tc_light_amd -- MI355X-native engine for TC-Light's two hot paths.

Python here is host plumbing (device memory, streams, torch.distributed); the arithmetic lives in
csrc/*.hip behind the C ABI of include/tclight_hip.h.
"""
__all__ = ["lib"]
