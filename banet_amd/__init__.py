"""banet_amd -- MI355X-native bundle-adjustment layer (the hot path of frobelbest/BANet).

    banet_amd.bundlenet   BundleNet + module functions, mirroring /root/reference/bundlenet.py
    banet_amd.legacy      Tracker, mirroring /root/reference/legacy/ba.py
    banet_amd.dense       dense multi-level window solver used by bench.py
    banet_amd.ops         the custom ops (equation_construction[_grad]) and fused entry points
    banet_amd.parallel    one-process-per-GPU sharding of independent windows

Everything computes in libbanet_hip.so (hand-written HIP for gfx950) through the C ABI in
include/banet_hip.h; there is no CPU or eager-PyTorch fallback.
"""
__version__ = "0.1.0"
