"""
learn_nerf — MI355X-native drop-in for the hot path of unixpickle/learn-nerf.

Same module and symbol names as the reference package (learn_nerf.model / render / train /
dataset / instant_ngp / ref_nerf); arrays are torch tensors on a ROCm device and the
arithmetic runs in hand-written HIP kernels behind the C ABI in include/lnrf.h.
"""
