"""ctypes binding of libcmoop_hip.so (include/cmoop.h).

There is NO CPU fallback: if the HIP library is missing or a call fails the
product path raises.  ``build()`` compiles the library in-tree with hipcc for
gfx950 (works without a GPU); loading it needs libamdhip64 only.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("CMOOP_LIB_PATH") or os.path.join(CSRC, "libcmoop_hip.so")   # override: A/B kernel builds
HEADER = os.path.join(os.path.dirname(_HERE), "include", "cmoop.h")

_lock = threading.Lock()
_lib = None


class CmoopError(RuntimeError):
    pass


class Config(C.Structure):
    """cmoop_config (include/cmoop.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "variant", "classes", "epochs", "batch", "patience", "early_stop", "restore_best", "acc_readout",
        "fpr_variant", "shuffle", "eval_batch", "n_slots", "profile_every", "gemm_mode")] + \
        [(n, C.c_double) for n in ("lr", "beta1", "beta2", "adam_eps", "bn_eps", "bn_momentum", "dropout")]


class DatasetStruct(C.Structure):
    """cmoop_dataset (include/cmoop.h)."""
    _fields_ = [("x_train", C.c_void_p), ("y_train", C.c_void_p), ("n_train", C.c_int64),
                ("x_val", C.c_void_p), ("y_val", C.c_void_p), ("n_val", C.c_int64),
                ("T", C.c_int32), ("F", C.c_int32)]


#: cmoop_next_fn (include/cmoop.h): int32_t (*)(void* ctx)
NEXT_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p)


def build(verbose: bool = False) -> str:
    """Compile libcmoop_hip.so for gfx950 in-tree (make; hipcc cross-compiles on CPU-only hosts)."""
    jobs = str(min(8, os.cpu_count() or 1))
    r = subprocess.run(["make", "-C", CSRC, "-j", jobs], capture_output=True, text=True)
    if r.returncode != 0:
        raise CmoopError("building libcmoop_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])
    return LIB_PATH


def declared_symbols():
    """Function names declared in include/cmoop.h."""
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"\b(cmoop_[a-z0-9_]+)\s*\(", txt)))


def lib():
    """The loaded library; raises CmoopError (never falls back) when it is absent."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise CmoopError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C cmoop_audio_processing_amd/csrc`). There is no CPU fallback for this path.")
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:
            raise CmoopError(f"cannot load {LIB_PATH}: {e}") from e
        L.cmoop_last_error.restype = C.c_char_p
        L.cmoop_abi_version.restype = C.c_int
        for name in declared_symbols():
            fn = getattr(L, name)          # AttributeError if the header and the library disagree
            if name not in ("cmoop_last_error", "cmoop_config_default"):
                fn.restype = C.c_int
        L.cmoop_config_default.restype = None
        _lib = L
        return L


def check(rc: int) -> None:
    if rc != 0:
        raise CmoopError(lib().cmoop_last_error().decode("utf-8", "replace"))


def default_config() -> Config:
    c = Config()
    lib().cmoop_config_default(C.byref(c))
    return c


def last_kernels():
    """Launch-path variant names of the GEMM kernels the calling thread's last kernel-level call launched."""
    buf = C.create_string_buffer(1024)
    check(lib().cmoop_last_kernels(buf, C.c_int32(1024)))
    return [k for k in buf.value.decode().split(";") if k]


def profile_entries():
    """[(kernel name, launches, total ms, total flops)] of the HIP-event-sampled MFMA GEMM launches."""
    L, cnt, out = lib(), C.c_int32(), []
    check(L.cmoop_profile_count(C.byref(cnt)))
    for i in range(cnt.value):
        name, n, ms, fl = C.create_string_buffer(160), C.c_int64(), C.c_double(), C.c_double()
        check(L.cmoop_profile_entry(i, name, 160, C.byref(n), C.byref(ms), C.byref(fl)))
        out.append((name.value.decode(), int(n.value), float(ms.value), float(fl.value)))
    return out


def profile_variants():
    """Launch-path variants (instantiation + "+sk" / "+stats" / "+tab" / "+slabs") of the sampled launches."""
    L, cnt, out = lib(), C.c_int32(), []
    check(L.cmoop_profile_variant_count(C.byref(cnt)))
    for i in range(cnt.value):
        name = C.create_string_buffer(200)
        check(L.cmoop_profile_variant(i, name, 200))
        out.append(name.value.decode())
    return out


def ptr(t):
    """Raw device/host pointer of a torch tensor or numpy array (must be contiguous)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        assert t.is_contiguous(), "tensor must be contiguous"
        return C.c_void_p(t.data_ptr())
    assert t.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return C.c_void_p(t.ctypes.data)
