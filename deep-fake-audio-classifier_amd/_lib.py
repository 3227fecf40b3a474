"""ctypes binding of libdfa_hip.so (include/dfa_hip.h).  There is no CPU fallback: if the shared library is
missing or a call fails, an exception is raised (ValueError for shape/dtype errors, RuntimeError otherwise,
mirroring how the reference surfaces torch shape errors and its own ValueErrors)."""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdfa_hip.so")

DFA_OK = 0
E_BAD_SHAPE, E_BAD_DTYPE, E_NULL_PTR, E_NOT_PREPARED, E_HIP, E_WORKSPACE, E_UNSUPPORTED = -1, -2, -3, -4, -5, -6, -7
DTYPE_F32, DTYPE_BF16 = 0, 1
PREC_F32, PREC_BF16, PREC_BF16X3 = 0, 1, 2
MODEL_CNN2D, MODEL_CNN1D, MODEL_CAE = 0, 1, 2
PRECISIONS = {"fp32": PREC_F32, "f32": PREC_F32, "float32": PREC_F32, "bf16": PREC_BF16, "bfloat16": PREC_BF16,
              "bf16x3": PREC_BF16X3}

_lib = None
_lock = threading.Lock()

# (name, restype, argtypes): every symbol include/dfa_hip.h declares
SYMBOLS = [
    ("dfa_version", C.c_int, []),
    ("dfa_ctx_create", C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("dfa_ctx_destroy", C.c_int, [C.c_void_p]),
    ("dfa_ctx_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("dfa_ctx_set_option", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("dfa_last_error", C.c_char_p, [C.c_void_p]),
    ("dfa_error_name", C.c_char_p, [C.c_int]),
    ("dfa_cnn2d_set_params", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    ("dfa_cnn2d_prepare", C.c_int, [C.c_void_p, C.c_int]),
    ("dfa_cnn2d_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                    C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("dfa_cnn2d_train_workspace_bytes", C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("dfa_cnn2d_forward_train", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                                          C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_uint64, C.c_uint64, C.c_float,
                                          C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("dfa_cnn2d_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                     C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t]),
    ("dfa_bce_smooth_fwd_bwd", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p,
                                         C.c_void_p]),
    ("dfa_augment_batch", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                    C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint64]),
    ("dfa_cnn2d_set_train_augment", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_float, C.c_uint64, C.c_uint64]),
    ("dfa_cnn1d_set_train_augment", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_float, C.c_uint64, C.c_uint64]),
    ("dfa_ctx_set_bn_sync", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    ("dfa_adamw_step", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float,
                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float]),
    ("dfa_cnn1d_set_params", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    ("dfa_cnn1d_prepare", C.c_int, [C.c_void_p]),
    ("dfa_cnn1d_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                    C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("dfa_cnn1d_train_workspace_bytes", C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    ("dfa_cnn1d_forward_train", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                                          C.c_int64, C.c_int64, C.c_float, C.c_uint64, C.c_uint64, C.c_float, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_size_t]),
    ("dfa_cnn1d_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                     C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t]),
    ("dfa_cae_set_params", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    ("dfa_cae_prepare", C.c_int, [C.c_void_p, C.c_int]),
    ("dfa_cae_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                  C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_size_t]),
    ("dfa_cae_train_workspace_bytes", C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("dfa_cae_forward_train", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                        C.c_int64, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_size_t]),
    ("dfa_cae_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                   C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t]),
    ("dfa_mse_fwd_bwd", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                  C.c_int64, C.c_void_p, C.c_void_p]),
    ("dfa_workspace_bytes", C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("dfa_dominant_kernel", C.c_char_p, [C.c_int, C.c_int]),
    ("dfa_ctx_timing_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("dfa_ctx_timing_reset", C.c_int, [C.c_void_p]),
    ("dfa_ctx_timing_read", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    ("dfa_ctx_debug_read", C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]),
    ("dfa_ctx_clock_read", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_int)]),
]


def load():
    """dlopen libdfa_hip.so once and declare prototypes.  Raises RuntimeError when it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `make -C deep-fake-audio-classifier_amd/csrc`).  dfa_amd has no CPU fallback.")
            lib = C.CDLL(LIB_PATH)
            for name, res, args in SYMBOLS:
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            _lib = lib
    return _lib


def check(ctx_handle, code):
    if code == DFA_OK:
        return
    lib = load()
    msg = lib.dfa_last_error(ctx_handle).decode() if ctx_handle else ""
    text = f"{lib.dfa_error_name(code).decode()}: {msg}"
    if code in (E_BAD_SHAPE, E_BAD_DTYPE, E_UNSUPPORTED):
        raise ValueError(text)
    raise RuntimeError(text)


class Context:
    """One dfa_ctx per (process, device).  Holds the activation workspace (a torch uint8 tensor: storage only)."""
    _by_device: dict = {}

    def __init__(self, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError(f"dfa_amd runs on an AMD GPU (torch device 'cuda'), got device '{device}'. "
                               "There is no CPU path.")
        self.device = device
        self.index = device.index if device.index is not None else torch.cuda.current_device()
        self.lib = load()
        h = C.c_void_p()
        code = self.lib.dfa_ctx_create(self.index, None, C.byref(h))
        if code != DFA_OK:
            raise RuntimeError(f"dfa_ctx_create(device={self.index}) failed: {self.lib.dfa_error_name(code).decode()}")
        self.handle = h
        self._ws = None
        # A dfa_ctx has ONE weight slot per model class (cnn2d / cnn1d / cae).  _owner remembers which nn.Module bound a
        # slot last, so a model whose own cache looks valid still re-binds after ANOTHER model of its class used the slot
        # (two live CNN2D instances used alternately).  _train_gen counts train-mode forwards per slot: a backward whose
        # forward is no longer the slot's latest would read another forward's saved activations and raises instead.
        self._owner = {}
        self._train_gen = {}

    @classmethod
    def get(cls, device) -> "Context":
        device = torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if idx not in cls._by_device:
            cls._by_device[idx] = Context(torch.device("cuda", idx))
        return cls._by_device[idx]

    def owner_changed(self, kind: str, model) -> bool:
        """Claim the ctx's `kind` weight slot for `model`; True when another model held it (caches must be rebuilt)."""
        token = model.__dict__.get("_ctx_token")
        if token is None:
            token = model.__dict__["_ctx_token"] = object()
        changed = self._owner.get(kind) is not token
        self._owner[kind] = token
        return changed

    def next_train_gen(self, kind: str) -> int:
        g = self._train_gen.get(kind, 0) + 1
        self._train_gen[kind] = g
        return g

    def check_train_gen(self, kind: str, gen: int, model=None):
        if model is not None and self._owner.get(kind) is not model.__dict__.get("_ctx_token"):
            raise RuntimeError(
                f"backward of a {kind} train-mode forward after ANOTHER {kind} model used this device's weight slot: "
                "run backward before calling a second model of the same class")
        if self._train_gen.get(kind) != gen:
            raise RuntimeError(
                f"backward of a {kind} train-mode forward that is no longer the latest one on this device: the saved "
                "activations live in one workspace per model class and were overwritten by a later train-mode forward. "
                "Call backward before the next forward (losses of two forwards cannot be summed on this path).")

    def use_current_stream(self):
        s = torch.cuda.current_stream(self.index).cuda_stream
        check(self.handle, self.lib.dfa_ctx_set_stream(self.handle, C.c_void_p(s)))

    def workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=torch.device("cuda", self.index))
        return self._ws

    def set_option(self, name: str, value: int):
        check(self.handle, self.lib.dfa_ctx_set_option(self.handle, name.encode(), int(value)))

    # ---- timing -----------------------------------------------------------------------------------------------
    def timing(self, enable):
        """False/0 = off, True/1 = every slot, other int = bit mask of slots (1 << slot)."""
        check(self.handle, self.lib.dfa_ctx_timing_enable(self.handle, int(enable)))

    def timing_reset(self):
        check(self.handle, self.lib.dfa_ctx_timing_reset(self.handle))

    def timing_read(self, slot: int):
        ms, n = C.c_float(), C.c_int()
        check(self.handle, self.lib.dfa_ctx_timing_read(self.handle, slot, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- synchronised BatchNorm --------------------------------------------------------------------------------
    BN_SYNC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)

    def set_bn_sync(self, process_group=None, enable=True):
        """Synchronised BatchNorm for data-parallel CNN2D training (dfa_ctx_set_bn_sync): every BatchNorm layer's per-channel
        sums are added over the ranks of `process_group` between their reduction and their use, so N ranks x B utterances
        behave like one rank x N*B.  enable=False (or a world of 1) restores local statistics."""
        import torch.distributed as dist
        world = dist.get_world_size(process_group) if (enable and dist.is_available() and dist.is_initialized()) else 1
        if not enable or world == 1:
            check(self.handle, self.lib.dfa_ctx_set_bn_sync(self.handle, None, None, 1, None, 0))
            self._bn_sync = None
            return
        buf = torch.zeros(1024, dtype=torch.float32, device=self.device)

        def hook(_user, _buf, count):
            try:       # in place, ordered after the work already on the current (= the context's) stream
                dist.all_reduce(buf[:count], op=dist.ReduceOp.SUM, group=process_group)
                return 0
            except Exception:   # noqa: BLE001 -- reported through the C ABI's error path
                return -1
        cb = self.BN_SYNC_FN(hook)
        self._bn_sync = (buf, cb, hook)                 # keep the tensor and the callback object alive
        check(self.handle, self.lib.dfa_ctx_set_bn_sync(self.handle, C.cast(cb, C.c_void_p), None, world,
                                                        C.c_void_p(buf.data_ptr()), buf.numel()))

    def clock_read(self):
        """(median, min, max GHz, workgroups) of the last bf16 CNN2D block-3 launch run with set_option("clock_probe", 1)."""
        med, lo, hi, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        check(self.handle, self.lib.dfa_ctx_clock_read(self.handle, C.byref(med), C.byref(lo), C.byref(hi), C.byref(n)))
        return med.value, lo.value, hi.value, n.value


def x_dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return DTYPE_F32
    if t.dtype == torch.bfloat16:
        return DTYPE_BF16
    raise ValueError(f"dfa_amd kernels take float32 or bfloat16 input, got {t.dtype}")


def ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr
