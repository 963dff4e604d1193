"""ctypes binding of libromtime_hip.so (declared in include/romtime_hip.h).

There is no CPU fallback: if the library is missing or no MI355X is visible the hot-path
entry points raise.  The only thing that works without a GPU is loading the library and
inspecting its symbols (used by the CPU-side tests).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libromtime_hip.so")

ROW_MAJOR, COL_MAJOR = 0, 1
RT_OK = 0
WARN_ZERO_NORM, WARN_SINGULAR = 1, 2

_p = C.c_void_p
_i64 = C.c_int64
_int = C.c_int

# name -> (restype, argtypes); mirrors include/romtime_hip.h one to one
SIGNATURES = {
    "rt_version": (_int, []),
    "rt_ctx_create": (_int, [C.POINTER(_p), _int]),
    "rt_ctx_destroy": (None, [_p]),
    "rt_ctx_set_stream": (_int, [_p, _p]),
    "rt_ctx_synchronize": (_int, [_p]),
    "rt_last_error": (C.c_char_p, [_p]),
    "rt_last_launch_info": (_int, [_p, C.POINTER(_i64)]),
    "rt_ctx_set_profile": (_int, [_p, _int]),
    "rt_ctx_set_option": (_int, [_p, C.c_char_p, _int]),
    "rt_last_gemm_ms": (_int, [_p, C.POINTER(C.c_double)]),
    "rt_last_gram_ms": (_int, [_p, C.POINTER(C.c_double)]),
    "rt_gram_plan_info": (_int, [_int, _i64, _i64, C.POINTER(_int)]),
    "rt_ctx_get_counter": (_int, [_p, C.c_char_p, C.POINTER(_i64)]),
    "rt_stream_create_cu_range": (_int, [_int, _int, _int, C.POINTER(_p)]),
    "rt_stream_destroy": (_int, [_p]),
    "rt_last_sweep_stats": (_int, [_p, C.POINTER(_i64)]),
    "rt_gram": (_int, [_p, _p, _i64, _i64, _i64, _int, _p]),
    "rt_gram_scale": (_int, [_p, _p, _i64, _p, _int, _p]),
    "rt_pod_backproject_weights": (_int, [_p, _p, _i64, _i64, _p, _p, _p]),
    "rt_pod_enqueue": (_int, [_p, _p, _i64, _i64, _i64, _int, _i64, _int, _p, _p, _p, _p, _p, _p, _p]),
    "rt_pod_orth": (_int, [_p, _p, _i64, _i64, _i64, _int, _i64, C.c_double, _int, _p, _i64, C.POINTER(_i64), _p, _p,
                           C.POINTER(_int)]),
    "rt_gemm_tn": (_int, [_p, _p, _i64, _int, _p, _i64, _int, _i64, _i64, _i64, _p, _i64]),
    "rt_gemm_nn": (_int, [_p, _p, _i64, _int, _p, _i64, _i64, _i64, _i64, _p, _i64, _int]),
    "rt_rank_update": (_int, [_p, _p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, C.c_double, _p, _i64]),
    "rt_gemm_nn_axpby": (_int, [_p, _p, _i64, _int, _p, _i64, _i64, _i64, _i64, C.c_double, C.c_double, _p, _i64, _int]),
    "rt_transpose": (_int, [_p, _p, _i64, _i64, _i64, _p, _i64]),
    "rt_deim_greedy": (_int, [_p, _p, _i64, _i64, _i64, _int, _p, _p, _p]),
    "rt_csr_spmm": (_int, [_p, _p, _p, _p, _i64, _p, _i64, _i64, _p, _i64]),
    "rt_project_csr": (_int, [_p, _p, _p, _p, _i64, _p, _i64, _i64, _p]),
    "rt_project_csr_batched": (_int, [_p, _p, _p, _p, _i64, _int, _i64, _i64, _p, _i64, _i64, _p]),
    "rt_dense_solve_batched": (_int, [_p, _p, _p, _i64, _i64, _p]),
    "rt_dense_solve_multi": (_int, [_p, _p, _i64, _p, _p, _i64, _p]),
    "rt_tracked_solve_batched": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _p]),
    "rt_rom_bdf_sweep": (_int, [_p, _p, _p]),
    "rt_hrom_bdf_sweep": (_int, [_p, _p, _p]),
    "rt_p1_local_assembly": (_int, [_p, _int, _i64, _p, _p, _i64, _i64, _p, _p, _int, _p, _p]),
    "rt_sym_eig_values": (_int, [_p, _p, _i64, _p, _p]),
    "rt_sym_eig_values_part": (_int, [_p, _p, _i64, _i64, _i64, _p, _p]),
    "rt_sym_eig_vectors": (_int, [_p, _i64, _i64, _p, _p]),
    "rt_host_jacobi_eigh": (_int, [_p, _i64, _p, _p, _int, C.POINTER(_int)]),
    "rt_bench_mfma_f64": (_int, [_p, _int, C.POINTER(C.c_double)]),
    "rt_bench_copy": (_int, [_p, _p, _p, _i64, _int, C.POINTER(C.c_double)]),
}



class SweepDesc(C.Structure):
    """rt_sweep_desc of include/romtime_hip.h."""

    _fields_ = [("N", _i64), ("nnz", _i64), ("r", _i64), ("n_mu", _i64), ("nt", _i64), ("dt", C.c_double),
                ("bdf2", _int), ("indptr", _p), ("indices", _p), ("V", _p), ("mass_values", _p), ("n_terms", _i64),
                ("term_values", _p), ("term_coef", _p), ("tril_values", _p), ("n_rhs", _i64), ("rhs_terms", _p),
                ("rhs_coef", _p)]


class HSweepDesc(C.Structure):
    """rt_hsweep_desc of include/romtime_hip.h."""

    _fields_ = [("r", _i64), ("n_mu", _i64), ("nt", _i64), ("dt", C.c_double), ("bdf2", _int), ("m_mass", _i64),
                ("m_lin", _i64), ("m_nl", _i64), ("m_rhs", _i64), ("Z", _p), ("Zf", _p), ("F_mass", _p), ("F_lin", _p),
                ("F_rhs", _p), ("W", _p), ("C_nl", _p), ("S_nl", _p)]


_lib = None
_lock = threading.Lock()


class RomtimeHipError(RuntimeError):
    pass


def load():
    """Load the shared library (no GPU needed) and bind every declared symbol."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RomtimeHipError(
                f"{LIB_PATH} is missing: build it with `python -m romtime_amd.build` "
                "(hipcc --offload-arch=gfx950). romtime_amd has no CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


class Context:
    """One rt_ctx per (thread, device); the stream follows torch's current stream."""

    _tls = threading.local()
    _live = weakref.WeakSet()      # every ctx not yet destroyed (pipeline.shutdown unbinds them from streams it destroys)
    _live_lock = threading.Lock()

    def __init__(self, device: int):
        import torch

        if not torch.cuda.is_available():
            raise RomtimeHipError("no MI355X visible: romtime_amd's hot path runs on the GPU only")
        self.lib = load()
        self.device = device
        h = _p()
        rc = self.lib.rt_ctx_create(C.byref(h), device)
        if rc != RT_OK:
            raise RomtimeHipError(f"rt_ctx_create(device={device}) failed with {rc}")
        self.handle = h
        self.options = {}
        self._pid = os.getpid()
        with Context._live_lock:
            Context._live.add(self)

    @classmethod
    def unbind_streams(cls):
        """Point every live ctx at the null stream (before streams they may still be bound to are destroyed)."""
        with cls._live_lock:
            live = list(cls._live)
        for ctx in live:
            if ctx.handle and ctx._pid == os.getpid():
                ctx.lib.rt_ctx_set_stream(ctx.handle, None)

    def destroy(self):
        """Release the device resources of this ctx now (idempotent; a destroyed ctx must not be used again)."""
        try:
            if getattr(self, "handle", None) and getattr(self, "_pid", None) == os.getpid():
                self.lib.rt_ctx_destroy(self.handle)
        finally:
            self.handle = None

    @classmethod
    def destroy_all(cls):
        """Destroy every live ctx of the process (``romtime_amd.shutdown``): threads that still hold one - the calling
        thread's cache included - get a fresh one on their next operator call."""
        with cls._live_lock:
            live = list(cls._live)
            cls._live.clear()
        for ctx in live:
            ctx.destroy()
        cache = getattr(cls._tls, "cache", None)
        if cache:
            cache.clear()

    def __del__(self):
        # Not in a forked child: after fork() the interpreter drops the thread states of every thread but the forking
        # one, and with them the thread-local contexts of worker threads (pipeline.PodWorkers) - their finalisers would
        # call into a HIP runtime the child must not touch (seen as a segmentation fault in a child of multiprocessing
        # started after a tree walk had run).  The parent still owns the handle.
        try:
            if getattr(self, "handle", None) and getattr(self, "_pid", None) == os.getpid():
                self.lib.rt_ctx_destroy(self.handle)
            self.handle = None
        except Exception:
            pass

    @classmethod
    def current(cls) -> "Context":
        import torch

        dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
        pinned = getattr(cls._tls, "pinned", None)
        if pinned is not None:   # inside Context.use(): a pipeline stage with its own ctx (own arenas) on its own stream
            pinned.lib.rt_ctx_set_stream(pinned.handle, _p(torch.cuda.current_stream().cuda_stream))
            return pinned
        cache = getattr(cls._tls, "cache", None)
        if cache is None:
            cache = cls._tls.cache = {}
        ctx = cache.get(dev)
        if ctx is None or not ctx.handle:      # none yet, or destroyed by destroy_all() from another thread
            ctx = cache[dev] = cls(dev)
        ctx.lib.rt_ctx_set_stream(ctx.handle, _p(torch.cuda.current_stream().cuda_stream))
        return ctx

    def use(self, stream):
        """``with ctx.use(stream):`` - the operators called inside run on ``stream`` (a torch stream) through THIS ctx
        instead of the thread's default one: concurrent pipeline stages must not share scratch arenas."""
        import contextlib

        import torch

        @contextlib.contextmanager
        def scope():
            prev = getattr(Context._tls, "pinned", None)
            Context._tls.pinned = self
            try:
                with torch.cuda.stream(stream):
                    yield self
            finally:
                Context._tls.pinned = prev

        return scope()

    def check(self, rc: int, what: str) -> int:
        if rc < 0:
            msg = self.lib.rt_last_error(self.handle)
            raise RomtimeHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
        return rc

    def set_option(self, name: str, value: int):
        self.check(self.lib.rt_ctx_set_option(self.handle, name.encode(), int(value)), "rt_ctx_set_option")
        self.options[name] = int(value)

    def set_profile(self, on: bool):
        self.profiling = bool(on)
        self.lib.rt_ctx_set_profile(self.handle, int(bool(on)))

    def last_gemm_ms(self) -> float:
        out = C.c_double()
        self.check(self.lib.rt_last_gemm_ms(self.handle, C.byref(out)), "rt_last_gemm_ms")
        return out.value

    def last_gram_ms(self) -> float:
        """Duration of the most recent snapshot Gram kernel (gram128 route) in profile mode; NaN if the last Gram
        took the generic GEMM route."""
        out = C.c_double()
        if self.lib.rt_last_gram_ms(self.handle, C.byref(out)) != 0:
            return float("nan")
        return out.value

    def counter(self, name: str) -> int:
        """Device-side event counter (rt_ctx_get_counter); synchronises the stream."""
        out = _i64()
        self.check(self.lib.rt_ctx_get_counter(self.handle, name.encode(), C.byref(out)), "rt_ctx_get_counter")
        return int(out.value)

    def sweep_stats(self) -> dict:
        """How the most recent online sweep solved its reduced systems (rt_last_sweep_stats)."""
        buf = (_i64 * 4)()
        self.check(self.lib.rt_last_sweep_stats(self.handle, buf), "rt_last_sweep_stats")
        return dict(newton_iterations=int(buf[0]), restarts=int(buf[1]), lu_fallbacks=int(buf[2]), solves=int(buf[3]))

    def launch_info(self):
        buf = (_i64 * 3)()
        self.lib.rt_last_launch_info(self.handle, buf)
        return dict(grid=int(buf[0]), splits=int(buf[1]), tile=(int(buf[2]) // 1000, int(buf[2]) % 1000))
