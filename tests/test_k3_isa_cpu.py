"""Static audit of the selection kernels (csrc/k_topk.hip) on their compiled assembly (hipcc cross-compiles without a GPU).

Until round 4 the K3 kernel's NaN compares sat inside the per-quad bounds branch, hipcc put `s_waitcnt vmcnt(0)` behind every
16-byte load there, and a wave had ONE load in flight: five memory round trips in a row per neuron (profiles/r04_k3_notes.txt (e)).
Nothing in the results shows that -- only the time does -- so the shape of the load phase is pinned here: every load of a row is
issued before the first wait on any of them, and the register-resident classes do not spill."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "mammo-clip-dissect_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    out = tmp_path_factory.mktemp("k3isa") / "k_topk.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-S",
           "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "k_topk.hip")]
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return out.read_text().split("\n")


def kernel_body(lines, mangled_part):
    st = next(i for i, ln in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(mangled_part) + r"\S*:", ln))
    end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return [ln.strip() for ln in lines[st + 1:end]]


def longest_unwaited_run(body, load_re):
    """the longest run of matching loads with no s_waitcnt vmcnt and no branch between them"""
    best = run = 0
    for s in body:
        if re.match(load_re, s):
            run += 1
            best = max(best, run)
        elif re.match(r"s_waitcnt.*vmcnt|s_cbranch|s_branch|s_barrier", s) or s.endswith(":"):
            run = 0
    return best


# (threads, quads): the default classes of configs[1] (10 000 images), of the stress shape (25 000) and two around them
@pytest.mark.parametrize("threads,quads", [(256, 10), (512, 13), (512, 8), (512, 25)])
def test_k3_issues_all_loads_of_a_row_before_it_waits(asm, threads, quads):
    name = "neuron_topk_fast_kernelILi%dELi%dELi256E" % (threads, quads)
    body = kernel_body(asm, name)
    assert longest_unwaited_run(body, r"buffer_load_dwordx4 ") == quads
    text = "\n".join(asm)
    meta = text[text.index(".amdhsa_kernel _ZN12_GLOBAL__N_123" + name):]
    meta = meta[:meta.index(".end_amdhsa_kernel")]
    assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta), "the class spills"
    assert not any(s.startswith("scratch_") for s in body)


def test_k6_short_rows_and_the_transpose_keep_their_loads_in_flight(asm):
    assert longest_unwaited_run(kernel_body(asm, "row_topk_short_kernelILi10E"), r"buffer_load_dwordx4 ") == 4
    assert longest_unwaited_run(kernel_body(asm, "transpose_kernel"), r"global_load_dword ") >= 8
