"""ctypes binding of libq2048_hip.so (the C ABI of include/q2048.h) -- and, for the explicit device
"cpu" only, of libq2048_host.so: the same ABI on host memory, compiled from the same per-lane
arithmetic (csrc/q2048_host.cpp).

There is no fallback in either direction: a library that is missing or does not export the ABI
raises, every op of the package fails loudly, and which library a call goes to is decided by the
device the caller named (`lib_for`), never by what happens to be available.  `build()` compiles the
HIP library in-tree with hipcc for gfx950 (cross-compiles without a GPU), `build_host()` the host
library with g++."""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
INCLUDE = os.path.abspath(os.path.join(_PKG, "..", "include"))
LIB_PATH = os.environ.get("Q2048_LIB_PATH") or os.path.join(CSRC, "libq2048_hip.so")  # override: experiments
# the measurement build (-DQ2048_EXPERIMENTS: extra flag bits 8..23 for ablations and sort widths);
# tools/ and two parity tests load it explicitly, the package never does
EXPERIMENTS_LIB_PATH = os.path.abspath(os.path.join(_PKG, "..", "tools", "variants", "libq2048_hip_exp.so"))
HOST_LIB_PATH = os.path.join(CSRC, "libq2048_host.so")
HOST_DEPS = ["q2048_host.cpp", "q2048_core.hpp", "q2048_core5.hpp", "q2048_luts.inc"]
SOURCES = ["q2048_kernels.hip"]
DEPS = ["q2048_kernels.hip", "q2048_core.hpp", "q2048_core5.hpp", "q2048_luts.inc"]

OK, PENDING, ERR_ALLOC, ERR_BUSY = 0, 1, -8, -10
GROW_VERIFY_COUNT = 1
STATUS_BAD_ACTION, STATUS_TILE_OVERFLOW, STATUS_TABLE_FULL, STATUS_DEEP_ROW = 1, 2, 4, 8
FLAG_INDEPENDENT, FLAG_SINGLE_ENV, FLAG_TD_CAS = 1, 2, 4
FLAG_ENV_DQN, FLAG_RESET_SHAPING, FLAG_PLAY_ONLY, FLAG_NO_LEARN, FLAG_NO_NEW_ROWS = 8, 16, 32, 64, 128
FLAG_LINE_SUMMARY = 1 << 24
ABI_VERSION = 7
ST_STEPS, ST_EPISODES, ST_VALID, ST_SCORE, ST_INSERTS, ST_DROPS, ST_EXPLORE, ST_CAS_RETRY = range(8)
ST_HIST0, NSTAT_I = 8, 32
ST_HIST_BINS, ST_CAS_FALLBACK = 23, 31
SF_RETURN, SF_RETURN_SQ, SF_REWARD, NSTAT_F = 0, 1, 2, 4
SIZEOF_AUX, SIZEOF_SLOT, SIZEOF_EPISODE = 16, 32, 48
MIRROR_SEQ, MIRROR_WORDS = NSTAT_I + NSTAT_F, NSTAT_I + NSTAT_F + 1


class RolloutOpts(C.Structure):
    """q2048_rollout_opts (include/q2048.h): the optional extras of q2048_fused_rollout_opts."""
    _fields_ = [("size", C.c_uint32), ("reserved", C.c_uint32), ("log", C.c_void_p),
                ("log_capacity", C.c_int64), ("log_count", C.c_void_p), ("row_cache", C.c_void_p),
                ("stats_mirror", C.c_void_p), ("mirror_ticket", C.c_void_p)]

    def __init__(self, **kw):
        super().__init__(size=C.sizeof(RolloutOpts), **kw)


class NativeError(RuntimeError):
    """A q2048_* entry point returned an error code (`code`: the Q2048_ERR_* value, when known)."""

    def __init__(self, message="", code=None):
        super().__init__(message)
        self.code = code


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise FileNotFoundError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, d) for d in DEPS] + [os.path.join(INCLUDE, "q2048.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(out: str, defines=(), verbose: bool = False) -> str:
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", *defines,
           "-I", INCLUDE, "-I", CSRC, "-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -shared: builds csrc/libq2048_hip.so in-tree."""
    if not force and not is_stale():
        return LIB_PATH
    _compile(LIB_PATH, verbose=verbose)
    global _lib
    _lib = None
    return LIB_PATH


def build_host(force: bool = False, verbose: bool = False) -> str:
    """g++ -O3 -shared: builds csrc/libq2048_host.so, the CPU twin (device "cpu")."""
    deps = [os.path.join(CSRC, d) for d in HOST_DEPS] + [os.path.join(INCLUDE, "q2048.h")]
    if not force and os.path.exists(HOST_LIB_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB_PATH) for d in deps):
        return HOST_LIB_PATH
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise FileNotFoundError("no C++ compiler found for the host library (set CXX)")
    cmd = [cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", INCLUDE, "-I", CSRC, "-o", HOST_LIB_PATH,
           os.path.join(CSRC, "q2048_host.cpp")]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    global _host_lib
    _host_lib = None
    return HOST_LIB_PATH


def build_experiments(force: bool = False, verbose: bool = False) -> str:
    """The measurement build of the same sources (tools/variants/libq2048_hip_exp.so)."""
    out = EXPERIMENTS_LIB_PATH
    if not force and os.path.exists(out):
        t = os.path.getmtime(out)
        deps = [os.path.join(CSRC, d) for d in DEPS] + [os.path.join(INCLUDE, "q2048.h")]
        if not any(os.path.getmtime(d) > t for d in deps):
            return out
    return _compile(out, defines=("-DQ2048_EXPERIMENTS",), verbose=verbose)


# every pointer argument is a device pointer and travels as an integer (c_void_p)
_SIGNATURES = {
    "q2048_abi_version": (C.c_int, []),
    "q2048_strerror": (C.c_char_p, [C.c_int]),
    "q2048_claim_timeouts": (C.c_int, [C.POINTER(C.c_uint64)]),
    "q2048_sizeof_aux": (C.c_size_t, []),
    "q2048_sizeof_slot": (C.c_size_t, []),
    "q2048_env_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_uint64,
                                 C.c_uint64, C.c_void_p]),
    "q2048_env_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                  C.c_uint64, C.c_uint64, C.c_void_p]),
    "q2048_env_reset_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                     C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]),
    "q2048_env_step_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                    C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "q2048_env_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                 C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "q2048_env_step_to": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                    C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "q2048_sizeof_rowcache": (C.c_size_t, [C.c_int]),
    "q2048_q_choose_cached": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_double,
                                        C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "q2048_q_update_cached": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_double,
                                        C.c_double, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "q2048_env_step_draws": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "q2048_q_choose_draws": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.c_int, C.c_double, C.c_uint64, C.c_uint32,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "q2048_q_choose": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_double,
                                 C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "q2048_q_update": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_double,
                                 C.c_double, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "q2048_q_lookup": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_uint64,
                                 C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "q2048_fused_rollout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                      C.c_int, C.c_int64, C.c_double, C.c_double, C.c_double,
                                      C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "q2048_fused_rollout_log": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                          C.c_int, C.c_int64, C.c_double, C.c_double, C.c_double,
                                          C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.c_void_p]),
    "q2048_fused_rollout_opts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                           C.c_int, C.c_int64, C.c_double, C.c_double, C.c_double,
                                           C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.POINTER(RolloutOpts), C.c_void_p]),
    "q2048_table_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "q2048_table_reserve": (C.c_int, [C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "q2048_table_grow": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_int64), C.c_void_p]),
    "q2048_table_grow_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "q2048_table_grow_poll": (C.c_int, [C.c_void_p]),
    "q2048_table_grow_wait": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "q2048_table_grow_commit": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_void_p), C.c_void_p]),
    "q2048_table_grow_finish": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "q2048_table_grow_abort": (C.c_int, [C.c_void_p]),
    "q2048_table_trim": (C.c_int, [C.c_void_p]),
    "q2048_table_free": (C.c_int, [C.c_void_p]),
    "q2048_table_probe": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_uint64, C.c_void_p]),
    "q2048_table_count": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "q2048_table_summarise": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "q2048_det_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "q2048_det_rollout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                    C.c_int64, C.c_double, C.c_double, C.c_double, C.c_uint64,
                                    C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "q2048_det_rollout_cached": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                           C.c_int64, C.c_double, C.c_double, C.c_double, C.c_uint64,
                                           C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "q2048_rowcache_rebind": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                        C.c_void_p]),
    "q2048_legal_moves": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "q2048_encode_onehot": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "q2048_rt_choose": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_uint64, C.c_uint64,
                                  C.c_uint32, C.c_void_p, C.c_void_p]),
    "q2048_rt_lookup": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "q2048_rt_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_void_p,
                                  C.c_void_p]),
    "q2048_rt_fused_rollout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                         C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_uint64,
                                         C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "q2048_table_import": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_int, C.c_void_p, C.c_void_p]),
    "q2048_table_export": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_int, C.c_void_p, C.c_void_p]),
}

_lib = None


def load(path: str) -> C.CDLL:
    """Loads one build of the library and checks its ABI (every symbol, version, struct sizes)."""
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the native extension is required (nothing substitutes for it). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
    L = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the ABI is incomplete
        fn.restype, fn.argtypes = res, args
    if L.q2048_abi_version() != ABI_VERSION:
        raise ImportError(f"ABI version {L.q2048_abi_version()} != {ABI_VERSION}")
    if L.q2048_sizeof_aux() != SIZEOF_AUX or L.q2048_sizeof_slot() != SIZEOF_SLOT:
        raise ImportError("ABI struct sizes changed")
    return L


def lib() -> C.CDLL:
    """Loads the library or raises; never substitutes another implementation."""
    global _lib
    if _lib is None:
        _lib = load(LIB_PATH)
    return _lib


_host_lib = None


def host_lib() -> C.CDLL:
    """The CPU twin (libq2048_host.so): loaded only for the explicit device "cpu"."""
    global _host_lib
    if _host_lib is None:
        _host_lib = load(HOST_LIB_PATH)
    return _host_lib


def lib_for(device) -> C.CDLL:
    """The library of a torch device: "cuda" -> the HIP library, "cpu" -> the host library.  Nothing else,
    and never one for the other."""
    kind = getattr(device, "type", device)
    if kind == "cuda":
        return lib()
    if kind == "cpu":
        return host_lib()
    raise ValueError(f"no implementation for device {device!r} (cuda[:i] or cpu)")


def use_experiments_build() -> None:
    """tools/ and tests only: route this process's calls through the measurement build."""
    global _lib
    _lib = load(build_experiments())


def claim_timeouts(L: C.CDLL | None = None) -> int:
    """Lanes that ever gave up a bounded wait in the 5x5 row-creation protocol (expected: 0)."""
    out = C.c_uint64(0)
    check((L or lib()).q2048_claim_timeouts(C.byref(out)), "q2048_claim_timeouts")
    return int(out.value)


def check(code: int, what: str) -> None:
    if code != OK:
        L = _lib or _host_lib or lib()
        raise NativeError(f"{what}: {L.q2048_strerror(code).decode()} (code {code})", code)
