"""ctypes binding of libechohip.so (the C ABI declared in include/echo_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libechohip.so")
MAX_BANDS = 3

EXPORTS = [
    "echo_create", "echo_destroy", "echo_reserve", "echo_strerror", "echo_version",
    "echo_fir_bank", "echo_bandpass_bank", "echo_spectrum", "echo_mag2db", "echo_correlate",
    "echo_first_strong_peak", "echo_envelope_detection", "echo_argmax", "echo_confidence_fuse", "echo_kalman_update",
    "echo_synth_frames", "echo_chain", "echo_chain_full", "echo_chain_prefetch", "echo_prefetch_ms", "echo_graph_create", "echo_graph_launch", "echo_graph_destroy",
    "echo_set_profiling", "echo_last_kernel_ms", "echo_chain_noalloc", "echo_noalloc_max_frames",
    "echo_spectrum_rows", "echo_xcorr", "echo_envelope_rows", "echo_band_confidence", "echo_normalize_confidences",
    "echo_kalman_update_qr", "echo_chain_wave", "echo_kernel_times", "echo_chain_ordered", "echo_set_profiling_every",
    "echo_first_strong_peak_rows", "echo_spectrum_any", "echo_confined_iir_passes",
]


class EchoError(RuntimeError):
    pass


class EchoConfigC(C.Structure):
    _fields_ = [
        ("fs", C.c_int32), ("n_rx", C.c_int32), ("n_chirp", C.c_int32), ("n_bands", C.c_int32),
        ("iir_ncoef", C.c_int32), ("n_taps", C.c_int32), ("min_delay", C.c_int32), ("noise_n", C.c_int32),
        ("detector", C.c_int32), ("confidence", C.c_int32), ("fusion", C.c_int32), ("range_offset", C.c_int32),
        ("snr_gate_db", C.c_double), ("sound_speed", C.c_double), ("kalman_q", C.c_double), ("kalman_r", C.c_double),
        ("freq_weights", C.c_double * MAX_BANDS),
        ("b", C.c_void_p), ("a", C.c_void_p), ("zi", C.c_void_p), ("taps", C.c_void_p), ("chirps", C.c_void_p),
    ]


_lib = None


def load():
    """Load libechohip.so once; raises EchoError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EchoError(f"{LIB_PATH} not found: build it with `make -C dsp_amd/csrc` "
                        "(or python -c 'import __graft_entry__ as g; g.build()')")
    # torch first: its libamdhip64 must be the HIP runtime of the process (device memory and streams come from it);
    # loaded the other way round, the library would bring in the system copy and see no device beside torch's
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    P, I = C.c_void_p, C.c_int
    lib.echo_create.argtypes = [C.POINTER(EchoConfigC), C.POINTER(P)]
    lib.echo_destroy.argtypes = [P]; lib.echo_destroy.restype = None
    lib.echo_reserve.argtypes = [P, I]
    lib.echo_strerror.argtypes = [I]; lib.echo_strerror.restype = C.c_char_p
    lib.echo_version.argtypes = []; lib.echo_version.restype = C.c_char_p
    lib.echo_fir_bank.argtypes = [P, P, P, I, P]
    lib.echo_bandpass_bank.argtypes = [P, P, P, I, P]
    lib.echo_spectrum.argtypes = [P, P, P, I, I, P]
    lib.echo_mag2db.argtypes = [P, P, C.c_size_t, P]
    lib.echo_correlate.argtypes = [P, P, P, I, P]
    lib.echo_first_strong_peak.argtypes = [P, P, P, P, P, I, P]
    lib.echo_envelope_detection.argtypes = [P, P, P, P, P, P, P, P, I, P]
    lib.echo_argmax.argtypes = [P, P, I, I, P]
    lib.echo_confidence_fuse.argtypes = [P, P, P, P, P, I, P]
    lib.echo_kalman_update.argtypes = [P, P, P, P, P, P, P, I, I, P]
    lib.echo_chain.argtypes = [P, P, P, P, P, P, P, I, I, P]
    lib.echo_spectrum_rows.argtypes = [P, P, P, I, I, P]
    lib.echo_xcorr.argtypes = [P, P, P, I, I, I, P]
    lib.echo_first_strong_peak_rows.argtypes = [P, I, I, I, I, C.c_double, P, P, P]
    lib.echo_spectrum_any.argtypes = [P, P, I, I, I, P]
    lib.echo_envelope_rows.argtypes = [P, P, I, I, I, P, P, P, P, P, P, I, P]
    lib.echo_band_confidence.argtypes = [P, P, P, P, P, I, P]
    lib.echo_normalize_confidences.argtypes = [P, P, I, I, P]
    lib.echo_kalman_update_qr.argtypes = [P, P, P, P, P, P, I, I, C.c_double, C.c_double, P]
    lib.echo_chain_ordered.argtypes = [P, P, P, P, P, P, P, I, I, I, P, P, P]
    lib.echo_set_profiling_every.argtypes = [P, I, I]
    lib.echo_kernel_times.argtypes = [P, C.POINTER(C.c_float), I]
    lib.echo_confined_iir_passes.argtypes = [P]
    lib.echo_confined_iir_passes.restype = C.c_longlong
    lib.echo_chain_wave.argtypes = [P, P, P, P, P, P, P, P, P, I, I, P]
    lib.echo_chain_noalloc.argtypes = [P, P, P, P, P, P, P, I, I, P]
    lib.echo_noalloc_max_frames.argtypes = [P]
    lib.echo_chain_full.argtypes = [P, P, P, P, P, P, P, P, I, I, P]
    lib.echo_synth_frames.argtypes = [P, P, P, I, C.c_longlong, C.c_ulonglong, P]
    lib.echo_chain_prefetch.argtypes = [P, P, I]
    lib.echo_prefetch_ms.argtypes = [P, C.POINTER(C.c_float)]
    lib.echo_graph_create.argtypes = [P, P, P, P, P, P, P, I, I, P, C.POINTER(P)]
    lib.echo_graph_launch.argtypes = [P, P]
    lib.echo_graph_destroy.argtypes = [P]; lib.echo_graph_destroy.restype = None
    lib.echo_set_profiling.argtypes = [P, I]
    lib.echo_last_kernel_ms.argtypes = [P] + [C.POINTER(C.c_float)] * 4
    for name in EXPORTS:
        getattr(lib, name)                  # AttributeError here: the built library is older than this binding
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().echo_strerror(rc).decode()
        raise EchoError(f"{what} failed: {msg} ({rc})")
