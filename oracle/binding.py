"""oracle/binding.py — TEST INFRASTRUCTURE: ctypes binding of liboracle_prach.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(the product never does: it must fail loudly when its HIP library is missing).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle_prach.so")

VARIANT_BETA_C, VARIANT_WITHNOMA_C = 0, 1
RNG_GLIBC, RNG_PHILOX = 0, 1
SCAN_SETS, SCAN_LITERAL = 0, 1


class OracleCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "variant", "uniform", "nUE", "nPreamble", "backoff", "nGrantUL", "maxRarWindow",
        "maxMsg2TxCount", "accessTime", "scan_mode", "max_steps", "sector_grants")]


UE_FIELDS = ("idx", "timer", "active", "txTime", "firstTxTime", "secondTxTime", "nowBackoff", "preamble",
             "preambleChange", "rarWindow", "maxRarCounter", "preambleTxCounter", "msg2Flag",
             "connectionRequest", "msg4Flag", "failCount")


class OracleUE(C.Structure):
    _fields_ = [(n, C.c_int32) for n in UE_FIELDS]


class OracleResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "time_exit", "maxTime", "nSuccessUE", "failedUEs", "preambleTxCount", "failCounts",
        "collisionPreambles", "totalPreambleTxop", "activeCheck", "nAccessUE", "continueFaliedUEs",
        "finalSuccessUEs")] + [("totalDelay", C.c_float), ("sumTimer", C.c_int64), ("draws", C.c_uint64),
                               ("steps", C.c_uint64), ("collisionCalls", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def build(force: bool = False) -> str:
    """Compile the oracle (gcc) if needed; returns the .so path."""
    srcs = [os.path.join(HERE, f) for f in ("prach_oracle.c", "noma_oracle.c", "prach_oracle.h", "noma_oracle.h",
                                            "glibc_rand.h", "philox.h")]
    srcs = [s for s in srcs if os.path.exists(s)]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.oracle_rng_new.restype = C.c_void_p
        L.oracle_rng_new.argtypes = [C.c_int, C.c_uint64]
        L.oracle_rng_free.argtypes = [C.c_void_p]
        L.oracle_rng_consumed.restype = C.c_uint64
        L.oracle_rng_consumed.argtypes = [C.c_void_p]
        L.oracle_rng_next_glibc.restype = C.c_int
        L.oracle_rng_next_glibc.argtypes = [C.c_void_p]
        L.oracle_philox_draw31.restype = C.c_int
        L.oracle_philox_draw31.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.oracle_arrival_schedule.restype = C.c_int
        L.oracle_arrival_schedule.argtypes = [C.POINTER(OracleCfg), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32)]
        L.oracle_run_trial.restype = C.c_int
        L.oracle_run_trial.argtypes = [C.POINTER(OracleCfg), C.c_void_p, C.POINTER(OracleResult), C.POINTER(OracleUE)]
        L.oracle_format_logs.restype = C.c_size_t
        L.oracle_format_logs.argtypes = [C.POINTER(OracleUE), C.c_int, C.c_char_p, C.c_size_t]
        L.oracle_format_results.restype = C.c_size_t
        L.oracle_format_results.argtypes = [C.POINTER(OracleCfg), C.POINTER(OracleResult), C.c_char_p, C.c_size_t]
        L.oracle_format_stdout.restype = C.c_size_t
        L.oracle_format_stdout.argtypes = [C.POINTER(OracleCfg), C.POINTER(OracleResult), C.c_char_p, C.c_size_t]
        _lib = L
    return _lib


def make_cfg(nUE, variant=VARIANT_BETA_C, uniform=0, nPreamble=54, backoff=20, nGrantUL=None, maxRarWindow=6,
             maxMsg2TxCount=9, accessTime=5, scan_mode=SCAN_SETS, max_steps=0, sector_grants=0) -> OracleCfg:
    if nGrantUL is None:
        nGrantUL = 54 if variant == VARIANT_BETA_C else 12  # Beta.c:49 / WithNOMA:73
    return OracleCfg(variant, uniform, nUE, nPreamble, backoff, nGrantUL, maxRarWindow, maxMsg2TxCount, accessTime,
                     scan_mode, max_steps, sector_grants)


class Rng:
    """One RNG stream (glibc: continues across trials like the reference's single srand per seed)."""

    def __init__(self, mode, seed):
        self.mode, self.seed = mode, seed
        self.h = lib().oracle_rng_new(mode, seed)

    def consumed(self):
        return lib().oracle_rng_consumed(self.h)

    def next_glibc(self):
        return lib().oracle_rng_next_glibc(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_rng_free(self.h)
            self.h = None


def run_trial(cfg: OracleCfg, rng: Rng, want_ues=True):
    res = OracleResult()
    ues = (OracleUE * cfg.nUE)() if want_ues else None
    rc = lib().oracle_run_trial(C.byref(cfg), rng.h, C.byref(res), ues)
    if rc != 0:
        raise RuntimeError(f"oracle_run_trial rc={rc}")
    return res, ues


def format_logs(ues, nUE) -> bytes:
    n = lib().oracle_format_logs(ues, nUE, None, 0)
    buf = C.create_string_buffer(n + 1)
    lib().oracle_format_logs(ues, nUE, buf, n + 1)
    return buf.raw[:n]


def format_results(cfg, res) -> bytes:
    buf = C.create_string_buffer(2048)
    n = lib().oracle_format_results(C.byref(cfg), C.byref(res), buf, 2048)
    return buf.raw[:n]


def format_stdout(cfg, res) -> bytes:
    buf = C.create_string_buffer(4096)
    n = lib().oracle_format_stdout(C.byref(cfg), C.byref(res), buf, 4096)
    return buf.raw[:n]


def arrival_schedule(cfg):
    cap = 60000 // max(1, cfg.accessTime) + 2
    out = (C.c_int32 * cap)()
    na = C.c_int32(0)
    n = lib().oracle_arrival_schedule(C.byref(cfg), out, cap, C.byref(na))
    return list(out[:n]), na.value


# ---- phase model (CPU model of the kernels' parallel decomposition; test infrastructure) ----------

def glibc_stream(seed: int, n: int):
    import numpy as np
    out = np.empty(n, dtype=np.int32)
    L = lib()
    L.model_glibc_stream.argtypes = [C.c_uint, C.c_uint64, C.c_void_p]
    L.model_glibc_stream.restype = None
    L.model_glibc_stream(seed, n, out.ctypes.data)
    return out


def model_run_trial(cfg: OracleCfg, rng_mode, seed, stream=None, stream_off=0, nranges=16, want_ues=True):
    L = lib()
    L.model_run_trial.restype = C.c_int
    L.model_run_trial.argtypes = [C.POINTER(OracleCfg), C.c_int, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64,
                                  C.c_int, C.POINTER(OracleResult), C.POINTER(OracleUE)]
    res = OracleResult()
    ues = (OracleUE * cfg.nUE)() if want_ues else None
    sp = stream.ctypes.data if stream is not None else None
    sl = len(stream) if stream is not None else 0
    rc = L.model_run_trial(C.byref(cfg), rng_mode, seed, sp, sl, stream_off, nranges, C.byref(res), ues)
    if rc != 0:
        raise RuntimeError(f"model_run_trial rc={rc}")
    return res, ues


# ---- NOMA.c variant ---------------------------------------------------------------------------------

class NomaCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nUE", "nPreamble", "backoff", "nGrantUL", "maxRarWindow", "maxMsg1ReTx",
                                         "accessTime", "max_steps", "nonsector")] + [("cellRadius", C.c_float)]


class NomaResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nSuccessUE", "delay", "nTxP", "activeCheck", "time_exit", "raFailedUEs")] + [
        ("draws", C.c_uint64), ("steps", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


NOMA_UE_FIELDS = ("idx", "timer", "active", "txTime", "firstTxTime", "secondTxTime", "nowBackoff", "preamble", "sector",
                  "rarWindow", "msg1ReTx", "nTxPreamble", "msg2", "msg3Wait", "RA", "RaFailed")


class NomaUE(C.Structure):
    _fields_ = [(n, C.c_int32) for n in NOMA_UE_FIELDS] + [("channelGain", C.c_double)]


def make_noma_cfg(nUE, nPreamble=54, backoff=20, nGrantUL=2, maxRarWindow=5, maxMsg1ReTx=10, accessTime=5, max_steps=0,
                  cellRadius=500.0, nonsector=0) -> NomaCfg:
    return NomaCfg(nUE, nPreamble, backoff, nGrantUL, maxRarWindow, maxMsg1ReTx, accessTime, max_steps, nonsector, cellRadius)


def noma_run_trial(cfg: NomaCfg, rng: Rng, want_ues=True):
    L = lib()
    L.noma_oracle_run_trial.restype = C.c_int
    L.noma_oracle_run_trial.argtypes = [C.POINTER(NomaCfg), C.c_void_p, C.POINTER(NomaResult), C.POINTER(NomaUE)]
    res = NomaResult()
    ues = (NomaUE * cfg.nUE)() if want_ues else None
    rc = L.noma_oracle_run_trial(C.byref(cfg), rng.h, C.byref(res), ues)
    if rc != 0:
        raise RuntimeError(f"noma_oracle_run_trial rc={rc}")
    return res, ues


def noma_format_line(cfg, res) -> bytes:
    L = lib()
    L.noma_format_result_line.restype = C.c_size_t
    L.noma_format_result_line.argtypes = [C.POINTER(NomaCfg), C.POINTER(NomaResult), C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(256)
    n = L.noma_format_result_line(C.byref(cfg), C.byref(res), buf, 256)
    return buf.raw[:n]
