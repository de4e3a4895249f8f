#!/usr/bin/env python3
"""ctypes binding of libd2d_probe.so (include/d2d_hip_diag.h, csrc/d2d_probe.hip): the streaming-store probe behind
bench.py's `box_write_ceiling`.  Measurement equipment - the product library (libd2d_hip.so) does not contain it.

    python tools/write_probe.py            # the per-variant rates of the fill family as JSON lines (on a GPU box)
"""
import ctypes as C
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LIB_PATH = ROOT / 'gym_d2d_amd' / 'lib' / 'libd2d_probe.so'
VARIANTS = 32
_lib = None


def load():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(f'{LIB_PATH} is missing - build it with `python -m gym_d2d_amd.build`')
        lib = C.CDLL(str(LIB_PATH))
        lib.d2d_probe_last_error.restype = C.c_char_p
        lib.d2d_probe_write_variants.argtypes = [C.c_int32, C.c_size_t, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32]
        lib.d2d_probe_write_staged.argtypes = [C.c_int32, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
        _lib = lib
    return _lib


def _check(rc):
    if rc:
        raise ValueError('write probe: ' + load().d2d_probe_last_error().decode(errors='replace'))


def write_variants(nbytes: int, iters: int = 5, device: int = 0):
    """(best GB/s, [per-variant GB/s ..., hipMemsetAsync]) - d2d_probe_write_variants."""
    best = C.c_double()
    arr = (C.c_double * (VARIANTS + 1))()
    _check(load().d2d_probe_write_variants(device, nbytes, iters, C.byref(best), arr, VARIANTS + 1))
    return best.value, list(arr)


def write_staged(nbytes: int, variant: int, stagger: int = 0, iters: int = 5, dst_ptr: int = 0, device: int = 0) -> float:
    """GB/s of ONE member of the fill family with the obs kernel's timing structure (variant + 32: LDS stage + barrier, + 64: wave
    stagger, + 128 k: store policy); dst_ptr = 0 writes a scratch buffer, else that device memory - d2d_probe_write_staged."""
    g = C.c_double()
    _check(load().d2d_probe_write_staged(device, C.c_void_p(dst_ptr or None), nbytes, variant, stagger, iters, C.byref(g)))
    return g.value


if __name__ == '__main__':
    best, rates = write_variants(8 << 30, 5)
    blocks, rows = (768, 1024, 512, 256), (2, 4, 8, 32)
    for v, r in enumerate(rates):
        what = 'hipMemsetAsync' if v == VARIANTS else {'block': blocks[v & 3], 'rows_per_wg': rows[(v >> 2) & 3], 'nontemporal': not (v & 16)}
        print(json.dumps({'variant': v, 'shape': what, 'GB_per_s': round(r, 1)}))
    print(json.dumps({'best_GB_per_s': round(best, 1)}))
    sys.exit(0)
