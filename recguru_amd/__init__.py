"""recguru_amd -- MI355X-native AE+GAN training step of RecGURU (hand-written HIP behind a C ABI).

The product path is the HIP library (recguru_amd/librecguru_hip.so, built in-tree by
recguru_amd.build / __graft_entry__.build()).  There is no CPU fallback: importing
recguru_amd.hip without the library, or running an op without a GPU, raises.
"""
__version__ = "0.1.0"
