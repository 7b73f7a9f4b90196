"""Static audit of the round-4 K1s kernel (csrc/k_gexp_v4.inc) on its compiled assembly (hipcc cross-compiles without a GPU).

The kernel keeps its 256 accumulators in a0..a255 BY NAME inside asm statements (cdna_hip_programming.md section 5.7 item 4):
the compiler must never place a value of its own there -- a spill to the accumulator file or a compiler-made v_accvgpr_* would be
silent corruption -- and the K loop must stay free of scratch traffic."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
@pytest.mark.parametrize("ablate,sync", [(0, 0), (0, 1), (1, 2)])
def test_k1s_v4_keeps_the_compiler_out_of_the_accumulator_file(tmp_path, ablate, sync):
    out = tmp_path / "k1s.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
           "-DAB=%d" % ablate, "-DSY=%d" % sync, "-S", "--cuda-device-only", "-o", str(out), os.path.join(HERE, "k1s_v4_tu.hip")]
    subprocess.run(cmd, check=True, cwd=HERE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    text = out.read_text()
    body = text[text.index("gemm_nt_bf16_exp_v4_kernel"):]
    in_asm, mfma, acc_reads, offenders = False, 0, 0, []
    for ln in body.splitlines():
        s = ln.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if s.startswith(";") or s.startswith(".") or not s:
            continue
        if in_asm:
            mfma += s.startswith("v_mfma_f32_16x16x32_bf16")
            acc_reads += s.startswith("v_accvgpr_read_b32")
            continue
        if re.search(r"(^|[\s,\[])a\[?\d", s) or "accvgpr" in s or s.startswith("scratch_"):
            offenders.append(s)
    assert not offenders, offenders[:5]
    assert mfma == 8 * 64                       # the tile's first round + the loop's round, 4 k-steps x 64 blocks each
    assert acc_reads == 256
    meta = text[text.index(".amdhsa_kernel"):]
    assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta)
    assert re.search(r"\.vgpr_spill_count:\s+0\b", text) and re.search(r"\.sgpr_spill_count:\s+[0-3]\b", text)
