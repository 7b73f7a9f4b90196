"""CPU: the accurate-log table of K4 (csrc/log_table.inc, scripts/gen_log_table.py).

The kernel's algorithm (k_wpmi.hip: log_tab2) is emulated here operation for operation in numpy -- every fp32
operation rounded on its own, fma through float64 (exact for the r step, at most a double rounding elsewhere) --
on the table the generator emits, and compared with the correctly rounded log and with torch.log (the reference's
log: MKL vsLn).  The GPU side of the same check is tests/test_gpu_kernels.py::test_accurate_log_matches_torch."""
import importlib.util
import os
import re

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


def _gen():
    spec = importlib.util.spec_from_file_location("gen_log_table", os.path.join(ROOT, "scripts", "gen_log_table.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def emulate(x, gen):
    rows = gen.table()
    V = np.array([r[2] for r in rows])
    inv = np.array([r[3] for r in rows], dtype=f32)
    vh_t = V.astype(f32)
    vl_t = (V - vh_t.astype(np.float64)).astype(f32)
    base = (127 + gen.E_MIN) * gen.NI
    i = (x.view(np.uint32) >> 16).astype(np.int64) - base
    assert i.min() >= 0 and i.max() < len(rows)
    n = x.size
    vh, vl = vh_t[i], vl_t[i]
    r = _fma(x, inv[i], np.full(n, -1, f32))
    assert np.array_equal(r.astype(np.float64), x.astype(np.float64) * inv[i].astype(np.float64) - 1)  # exact
    assert np.abs(r).max() < 2.0 ** -7
    q = _fma(r, np.full(n, -0.25, f32), np.full(n, f32(float.fromhex("0x1.555556p-2")), f32))
    q = _fma(r, q, np.full(n, -0.5, f32))
    low = _fma((r * r).astype(f32), q, vl)
    H = (vh + r).astype(f32)
    err = (r - (H - vh).astype(f32)).astype(f32)
    low = (low + err).astype(f32)
    return (H + low).astype(f32)


def _ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def test_inc_file_matches_generator():
    gen = _gen()
    txt = open(os.path.join(ROOT, "mammo-clip-dissect_amd", "csrc", "log_table.inc")).read()
    recs = re.findall(r"\{0x([0-9a-f]{8})u, 0x([0-9a-f]{8})u, 0x([0-9a-f]{8})u\}", txt)
    rows = gen.table()
    assert len(recs) == len(rows) == (1 - gen.E_MIN) * gen.NI
    for k in (0, 1, 777, len(rows) - 129, len(rows) - 128, len(rows) - 1):
        V, inv = rows[k][2], rows[k][3]
        hi = f32(V)
        want = [int(np.array([v], f32).view(np.uint32)[0]) for v in (hi, f32(V - float(hi)), f32(inv))]
        assert [int(h, 16) for h in recs[k]] == want


def test_log_emulation_is_nearly_correctly_rounded():
    gen = _gen()
    rng = np.random.default_rng(5)
    n = 2_000_000
    x = np.concatenate([np.exp(rng.uniform(np.log(2.0 ** gen.E_MIN), np.log(2.0), n // 2)),   # the whole table range
                        rng.uniform(0.002, 0.06, n // 2)]).astype(f32)                  # where soft-WPMI arguments live
    x = x[(x >= f32(2.0 ** gen.E_MIN)) & (x < 2)]
    got = emulate(x, gen)
    cr = np.log(x.astype(np.float64)).astype(f32)
    t = torch.log(torch.from_numpy(x)).numpy()
    assert _ulps(got, cr).max() <= 1
    assert (got == cr).mean() >= 0.99995
    assert (got == t).mean() >= 0.9995


def test_log_emulation_around_one_and_at_edges():
    gen = _gen()
    lo, hi = f32(0.96).view(np.uint32), f32(1.02).view(np.uint32)
    x = np.arange(lo, hi, 7, dtype=np.uint32).view(f32)
    got = emulate(x, gen)
    cr = np.log(x.astype(np.float64)).astype(f32)
    assert _ulps(got, cr).max() <= 1 and (got == cr).mean() >= 0.999
    one = np.array([1.0, np.nextafter(f32(1), f32(0)), np.nextafter(f32(1), f32(2)), 2.0 ** gen.E_MIN,
                    np.nextafter(f32(2), f32(0)), 1e-7, 0.5], dtype=f32)
    got = emulate(one, gen)
    assert got[0] == 0.0
    assert np.array_equal(got, np.log(one.astype(np.float64)).astype(f32))
