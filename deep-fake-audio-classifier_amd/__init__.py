"""dfa_amd -- MI355X-native hot path of kingdomseed/Deep-Fake-Audio-Classifier.

Host side mirrors the reference's module layout (model.py, model_cnn1d.py, model_cae.py, dataset.py,
dataloaders.py, evaluation.py, predict.py, ...) so that `from dfa_amd.model import CNN2D` is a drop-in for the
reference's `from model import CNN2D`.  The arithmetic runs in hand-written HIP kernels (csrc/) reached through
the C ABI of include/dfa_hip.h via ctypes (_lib.py); PyTorch-ROCm tensors are storage only.
"""
__version__ = "0.1.0"
