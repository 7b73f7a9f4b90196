"""GEMM solution selection for the encoder forwards (PyTorch-ROCm TunableOp).

The forwards are PyTorch-ROCm host code (fp32 hipBLASLt / rocBLAS GEMMs are 80 % of a dissection step).  PyTorch's
TunableOp can pick, per GEMM shape, the fastest of the libraries' solutions; `tunableop_gfx950.csv` holds the picks
for the shapes of the ViT-B/16 + text-tower forwards at batch 250 on an MI355X (tuned once with
`python bench.py --tune`; +4.6 % images/s).  The file is only honoured when its validator lines (PyTorch, HIP,
hipBLASLt, rocBLAS versions, gfx arch) match the running stack; otherwise -- and for shapes it does not list -- the
libraries' default solutions run.  All solutions are fp32; only the summation order inside a GEMM differs."""
import os

RESULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")


def enable_gemm_tuning(tune=False, results=RESULTS):
    """Turn TunableOp on with the packaged picks.  tune=True also tunes shapes that are not in the file (slow the
    first time each shape is seen); PyTorch then writes the merged table to ./tunableop_results<device>.csv on exit.
    Returns True when the packaged picks were accepted."""
    import torch
    try:
        t = torch.cuda.tunable
        t.enable(True)
        t.tuning_enable(bool(tune))
        ok = bool(t.read_file(results)) if os.path.exists(results) else False
        return ok
    except (AttributeError, RuntimeError):
        return False
