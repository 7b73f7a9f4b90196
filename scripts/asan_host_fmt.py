"""Random rows through the native CSV cell formatters (csrc/mcd_host.c) against numpy's str(row), meant to run on a SANITISED build:
    gcc -O1 -g -fPIC -shared -fsanitize=address,undefined -ffp-contract=off -std=c11 -o /tmp/libmcd_host_asan.so mammo-clip-dissect_amd/csrc/mcd_host.c -lm
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python scripts/asan_host_fmt.py
(CPU only; round 2: 3 000 batches incl. random bit patterns, NaN / inf, zeros: no sanitizer report, no mismatch.)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mammo_clip_dissect_amd.pipeline as pl
# swap in the sanitised build of libmcd_host.so
L = ctypes.CDLL("/tmp/libmcd_host_asan.so")
for fn in (L.mcd_fmt_f32_rows, L.mcd_fmt_i64_rows):
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
L.mcd_csv_og_rows.restype = ctypes.c_int64
L.mcd_csv_og_rows.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                              ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
pl._host_lib = L
rng = np.random.default_rng(0)
bad = 0
for it in range(3000):
    n = int(rng.integers(1, 65))
    rows = int(rng.integers(1, 20))
    kind = it % 6
    if kind == 0:
        a = rng.standard_normal((rows, n)).astype(np.float32) * np.float32(10.0 ** int(rng.integers(-6, 9)))
    elif kind == 1:
        a = (rng.standard_normal((rows, n)) * 3 - 400).astype(np.float32)
    elif kind == 2:
        a = rng.standard_normal((rows, n)).astype(np.float32); a[rng.random(a.shape) < 0.1] = 0.0
    elif kind == 3:
        a = rng.standard_normal((rows, n)).astype(np.float32); a[0, 0] = np.nan; a[-1, -1] = np.inf
    elif kind == 4:
        a = np.frombuffer(rng.bytes(rows * n * 4), np.float32).reshape(rows, n).copy()
    else:
        a = np.round(rng.standard_normal((rows, n)) * 1000).astype(np.float32)
    with np.errstate(all="ignore"):
        got = pl.format_f32_rows(a)
    want = [str(r) for r in a]
    if got != want:
        bad += 1
        if bad < 5:
            print("f32 mismatch kind", kind, repr(got[0])[:80], repr(want[0])[:80])
    b = rng.integers(-2**40 if it % 3 == 0 else 0, 2**40 if it % 3 == 0 else 60000, size=(rows, n)).astype(np.int64)
    if pl.format_i64_rows(b) != [str(r) for r in b]:
        bad += 1
        print("i64 mismatch")
print("done, mismatches:", bad)
