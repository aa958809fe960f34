"""dynamicpdb_amd -- MI355X-native engine for the DFOLDv2 trajectory-prediction hot
path of fudan-generative-vision/dynamicPDB (IPA + 5x5 conv tower + backbone-frame
update + SE(3) diffusion score/noise/denoise + triangle pair operators).

Importing this package creates no HIP context (fork-safe for DataLoader workers,
as the reference's dataset-side diffuser requires); the C-ABI library
(csrc/libdfold_hip.so) is loaded lazily on first device op and its absence is a
hard error -- there is no CPU fallback for the device path.
"""
__version__ = "0.1.0"
