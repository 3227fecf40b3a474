"""Training-side pieces: checkpoint format (counterpart of src/training/checkpoint.py) and, in train_step.py, the
train-mode forward/backward path."""
from .checkpoint import build_config_dict, load_checkpoint, save_checkpoint  # noqa: F401


def cnn2d_train_forward(model, x, return_embedding=False):
    from .train_step import cnn2d_train_forward as _impl
    return _impl(model, x, return_embedding)


def cnn1d_train_forward(model, x):
    from .train_step import cnn1d_train_forward as _impl
    return _impl(model, x)


def cae_train_forward(model, x):
    from .train_step import cae_train_forward as _impl
    return _impl(model, x)
