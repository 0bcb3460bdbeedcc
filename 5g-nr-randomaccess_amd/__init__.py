"""5g-nr-randomaccess_amd — MI355X-native PRACH random-access Monte-Carlo engine (host-side binding).

Thin ctypes binding of ``libprach_hip.so`` (C ABI: ``include/prach.h``).  The simulation itself runs
in hand-written HIP kernels (``csrc/prach_kernels.hip``); this module only marshals parameter
structs.  There is NO CPU fallback: if the library is not built or no gfx950 device is present the
calls raise.

The directory name is not a Python identifier; load it with ``__graft_entry__.load_package()`` or
``importlib`` (tests/conftest.py does), under the module name ``nr_randomaccess_amd``.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRACH_LIB") or os.path.join(HERE, "libprach_hip.so")  # PRACH_LIB: diagnostic builds only
CLI_PATH = os.path.join(HERE, "prach_sim")

VARIANT_BETA_C, VARIANT_WITHNOMA_C, VARIANT_NOMA_C = 0, 1, 2
RNG_GLIBC, RNG_PHILOX = 0, 1
FLAG_SECTOR_GRANTS, FLAG_NOMA_NONSECTOR = 1, 2
OK = 0


class PrachCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("variant", "uniform", "nUE", "nPreamble", "backoff", "nGrantUL",
                                         "maxRarWindow", "maxMsg2TxCount", "accessTime", "rng_mode")] + [
        ("seed", C.c_uint64), ("stream_offset", C.c_uint64), ("max_steps", C.c_int32), ("flags", C.c_int32),
        ("cellRadius", C.c_float), ("hBS", C.c_float), ("hUT", C.c_float)]


class PrachResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("status", "time_exit", "maxTime", "nSuccessUE", "failedUEs",
                                         "preambleTxCount", "failCounts", "collisionPreambles", "totalPreambleTxop",
                                         "activeCheck", "nAccessUE", "continueFaliedUEs", "finalSuccessUEs")] + [
        ("totalDelay", C.c_float), ("sumTimer", C.c_int64), ("draws", C.c_uint64), ("steps", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


UE_FIELDS = ("idx", "timer", "active", "txTime", "firstTxTime", "secondTxTime", "nowBackoff", "preamble",
             "preambleChange", "rarWindow", "maxRarCounter", "preambleTxCounter", "msg2Flag",
             "connectionRequest", "msg4Flag", "failCount")


class PrachUeLog(C.Structure):
    _fields_ = [(n, C.c_int32) for n in UE_FIELDS]


class PrachTiming(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("upload_ms", C.c_double), ("total_ms", C.c_double),
                ("launches", C.c_int32), ("workgroups", C.c_int32), ("updates", C.c_uint64),
                ("cluster_size", C.c_int32), ("resident_limit", C.c_int32), ("fallback_trials", C.c_int32), ("spin_timeouts", C.c_int32),
                ("rec_mode", C.c_int32), ("xcd_packed", C.c_int32), ("group_visits", C.c_uint64), ("event_ues", C.c_uint64),
                ("trial_kernel_reruns", C.c_int32), ("noma_host_ues", C.c_int32)]


class PrachError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        try:
            msg = lib().prach_strerror(status).decode()
        except Exception:  # pragma: no cover
            msg = "?"
        super().__init__(f"libprach_hip status {status} ({msg}) {what}")


_lib = None


def lib():
    """Load libprach_hip.so (raises if it has not been built: there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: run __graft_entry__.build() "
                                    f"(make -C 5g-nr-randomaccess_amd/csrc); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.prach_engine_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.prach_engine_destroy.argtypes = [vp]
        L.prach_engine_destroy.restype = None
        L.prach_engine_set.argtypes = [vp, C.c_char_p, C.c_int64]
        L.prach_run_trials.argtypes = [vp, C.POINTER(PrachCfg), C.c_int, C.POINTER(PrachResult), C.POINTER(C.POINTER(PrachUeLog))]
        L.prach_last_timing.argtypes = [vp, C.POINTER(PrachTiming)]
        L.prach_cfg_defaults.argtypes = [C.POINTER(PrachCfg), C.c_int]
        L.prach_cfg_defaults.restype = None
        L.prach_cfg_validate.argtypes = [C.POINTER(PrachCfg)]
        L.prach_max_time.argtypes = [C.POINTER(PrachCfg)]
        L.prach_trial_cost.argtypes = [C.POINTER(PrachCfg)]
        L.prach_trial_cost.restype = C.c_double
        L.prach_arrival_schedule.argtypes = [C.POINTER(PrachCfg), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32)]
        L.prach_glibc_stream.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]
        L.prach_glibc_stream.restype = None
        L.prach_device_glibc_stream.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]
        L.prach_strerror.argtypes = [C.c_int]
        L.prach_strerror.restype = C.c_char_p
        L.prach_format_logs.argtypes = [C.POINTER(PrachUeLog), C.c_int, C.c_char_p, C.c_size_t]
        L.prach_format_logs.restype = C.c_size_t
        L.prach_format_results.argtypes = [C.POINTER(PrachCfg), C.POINTER(PrachResult), C.c_double, C.c_char_p, C.c_size_t]
        L.prach_format_results.restype = C.c_size_t
        L.prach_format_stdout.argtypes = [C.POINTER(PrachCfg), C.POINTER(PrachResult), C.c_double, C.c_char_p, C.c_size_t]
        L.prach_format_stdout.restype = C.c_size_t
        L.prach_result_file_name.argtypes = [C.POINTER(PrachCfg), C.c_int, C.c_char_p, C.c_size_t]
        L.prach_write_trial_files.argtypes = [C.POINTER(PrachCfg), C.POINTER(PrachResult), C.POINTER(PrachUeLog), C.c_double, C.c_char_p]
        L.prach_noma_activation_table.argtypes = [C.POINTER(PrachCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.prach_noma_activation_range.argtypes = [C.POINTER(PrachCfg), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.prach_noma_activation_table_device.argtypes = [C.c_void_p, C.POINTER(PrachCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.prach_format_noma_line.argtypes = [C.POINTER(PrachCfg), C.POINTER(PrachResult), C.c_char_p, C.c_size_t]
        L.prach_format_noma_line.restype = C.c_size_t
        L.prach_results_csv_accumulate.argtypes = [C.POINTER(C.c_double), C.c_char_p]
        L.prach_results_csv_row.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_char_p, C.c_size_t]
        L.prach_results_csv_row.restype = C.c_size_t
        _lib = L
    return _lib


EXPORTS = ("prach_engine_create", "prach_engine_destroy", "prach_engine_set", "prach_run_trials", "prach_last_timing",
           "prach_cfg_defaults", "prach_cfg_validate", "prach_max_time", "prach_trial_cost", "prach_arrival_schedule", "prach_glibc_stream",
           "prach_strerror", "prach_format_logs", "prach_format_results", "prach_format_stdout",
           "prach_result_file_name", "prach_write_trial_files", "prach_noma_activation_table", "prach_format_noma_line",
           "prach_results_csv_accumulate", "prach_results_csv_row", "prach_device_glibc_stream", "prach_noma_activation_range", "prach_noma_activation_stream",
           "prach_noma_activation_table_device")


def make_cfg(nUE, variant=VARIANT_BETA_C, uniform=0, rng_mode=RNG_GLIBC, seed=0, stream_offset=0, **kw) -> PrachCfg:
    """Defaults of the named program (Beta.c:47-57 / WithNOMA:70-88), overridden by keywords."""
    c = PrachCfg()
    lib().prach_cfg_defaults(C.byref(c), variant)
    c.nUE, c.uniform, c.rng_mode, c.seed, c.stream_offset = nUE, uniform, rng_mode, seed, stream_offset
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


class Engine:
    """One engine per process/GPU: owns the HIP stream and the device arena."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        rc = lib().prach_engine_create(device, C.byref(self._h))
        if rc != OK:
            raise PrachError(rc, "(prach_engine_create)")

    def close(self):
        if self._h:
            lib().prach_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set(self, key: str, value: int):
        rc = lib().prach_engine_set(self._h, key.encode(), value)
        if rc != OK:
            raise PrachError(rc, f"(set {key})")

    def run_trials(self, cfgs, want_logs=False):
        """Run the trials concurrently on the device. Returns (results, logs) — logs[k] is a ctypes
        array of PrachUeLog or None.  want_logs: True (every trial), False, or the indices of the trials
        whose per-UE log is wanted (the C ABI takes NULL for the others)."""
        n = len(cfgs)
        arr = (PrachCfg * n)(*cfgs)
        res = (PrachResult * n)()
        logs = [None] * n
        lp = None
        if want_logs:
            which = range(n) if want_logs is True else sorted(set(int(k) for k in want_logs))
            for k in which:
                logs[k] = (PrachUeLog * cfgs[k].nUE)()
            lp = (C.POINTER(PrachUeLog) * n)(*[C.cast(l, C.POINTER(PrachUeLog)) if l is not None else C.POINTER(PrachUeLog)()
                                               for l in logs])
        rc = lib().prach_run_trials(self._h, arr, n, res, lp)
        if rc != OK:
            raise PrachError(rc, "(prach_run_trials)")
        return list(res), logs

    def device_glibc_stream(self, seed, first, n):
        import numpy as np
        out = np.empty(n, dtype=np.int32)
        rc = lib().prach_device_glibc_stream(self._h, seed, first, n, out.ctypes.data)
        if rc != OK:
            raise PrachError(rc, "(prach_device_glibc_stream)")
        return out

    def timing(self) -> PrachTiming:
        t = PrachTiming()
        lib().prach_last_timing(self._h, C.byref(t))
        return t


def arrival_schedule(cfg: PrachCfg):
    cap = 60000 // max(1, cfg.accessTime) + 2
    out = (C.c_int32 * cap)()
    na = C.c_int32(0)
    n = lib().prach_arrival_schedule(C.byref(cfg), out, cap, C.byref(na))
    return list(out[:n]), na.value


def glibc_stream(seed, first, n):
    import numpy as np
    out = np.empty(n, dtype=np.int32)
    lib().prach_glibc_stream(seed, first, n, out.ctypes.data)
    return out


def format_logs(logs, nUE) -> bytes:
    cap = nUE * 384 + 1  # a line is at most 378 bytes (prach_host.c): one formatting pass
    buf = C.create_string_buffer(cap)
    n = lib().prach_format_logs(logs, nUE, buf, cap)
    return buf.raw[:n]


def format_results(cfg, res, latency=0.0) -> bytes:
    buf = C.create_string_buffer(2048)
    n = lib().prach_format_results(C.byref(cfg), C.byref(res), latency, buf, 2048)
    return buf.raw[:n]


def format_stdout(cfg, res, latency=0.0) -> bytes:
    buf = C.create_string_buffer(4096)
    n = lib().prach_format_stdout(C.byref(cfg), C.byref(res), latency, buf, 4096)
    return buf.raw[:n]


def noma_activation_table(cfg: PrachCfg):
    """(preamble0, sector, gain, ln gain, draws) per UE of the NOMA.c variant (host side, no GPU)."""
    import numpy as np
    n = cfg.nUE
    pre0, sec = np.empty(n, np.int32), np.empty(n, np.int32)
    gain, lgain = np.empty(n, np.float64), np.empty(n, np.float64)
    nd = np.empty(n, np.uint32)
    rc = lib().prach_noma_activation_table(C.byref(cfg), pre0.ctypes.data, sec.ctypes.data, gain.ctypes.data, lgain.ctypes.data, nd.ctypes.data)
    if rc != OK:
        raise PrachError(rc, "(prach_noma_activation_table)")
    return pre0, sec, gain, lgain, nd


def noma_activation_table_device(engine, cfg: PrachCfg):
    """The same table as the engine builds it on the GPU (Philox mode) + a per-UE flag: recomputed on the host (see include/prach.h)."""
    import numpy as np
    n = cfg.nUE
    pre0, sec = np.empty(n, np.int32), np.empty(n, np.int32)
    gain, lgain = np.empty(n, np.float64), np.empty(n, np.float64)
    nd, flagged = np.empty(n, np.uint32), np.empty(n, np.uint8)
    rc = lib().prach_noma_activation_table_device(engine._h, C.byref(cfg), pre0.ctypes.data, sec.ctypes.data, gain.ctypes.data, lgain.ctypes.data,
                                                  nd.ctypes.data, flagged.ctypes.data)
    if rc != OK:
        raise PrachError(rc, "(prach_noma_activation_table_device)")
    return pre0, sec, gain, lgain, nd, flagged


def format_noma_line(cfg, res) -> bytes:
    buf = C.create_string_buffer(256)
    n = lib().prach_format_noma_line(C.byref(cfg), C.byref(res), buf, 256)
    return buf.raw[:n]


def results_csv(rows_of_results_texts):
    """results.csv bytes for [[Results.txt text of seed 0, seed 1, ...] per nUE point] (AveragePerformance.py)."""
    out = b""
    for texts in rows_of_results_texts:
        acc = (C.c_double * 6)()
        for t in texts:
            rc = lib().prach_results_csv_accumulate(acc, t if isinstance(t, bytes) else t.encode())
            if rc != OK:
                raise PrachError(rc, "(prach_results_csv_accumulate)")
        buf = C.create_string_buffer(512)
        n = lib().prach_results_csv_row(acc, len(texts), buf, 512)
        out += buf.raw[:n]
    return out
