"""Static audit of the K1s kernel (csrc/k_gexp_v6.inc) on the assembly of the PRODUCT translation unit (csrc/k_gemm.hip compiled
for the device only; hipcc cross-compiles without a GPU).

The kernel owns most of the 512-entry register file BY NAME inside asm statements (cdna_hip_programming.md section 5.7 item 4):
48 accumulator blocks in v64..v255, the constant block in v60..v63, 16 accumulator blocks, both fragment sets, the row-sum block
and the selector fragments in a0..a255.  The compiler is held to v0..v59 (amdgpu_num_vgpr) and must never place a value of its own
in the rest -- a spill to the accumulator file, a compiler-made v_accvgpr_* or a temporary in v60+ would be silent corruption --
and nothing may go to scratch.  The instruction counts pin the structure: which k-steps exist as plain k-steps, which ride the
epilogue (the overlapped boundary phase), how many copies of the epilogue there are."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "mammo-clip-dissect_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
NVGPR = 60          # G6_NVGPR: the compiler's registers end at v59


@pytest.fixture(scope="module")
def k1s_asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("k1s") / "k_gemm.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
           "-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "k_gemm.hip")]
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return out.read_text()


def _kernel_body(text, needle):
    names = sorted(set(re.findall(r"^(_Z\w*%s\w*):" % needle, text, flags=re.M)))
    assert len(names) == 1, names                      # the product library carries ONE instantiation of ONE K1s kernel
    a = text.index(names[0] + ":")
    return names[0], text[a:text.index(".Lfunc_end", a)]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_k1s_keeps_the_compiler_out_of_the_named_registers(k1s_asm):
    name, body = _kernel_body(k1s_asm, "gemm_nt_bf16_exp_v6_kernel")
    assert "ILi0ELi1ELi2ELi2E" in name                 # ABLATE 0, PLACE 1, two k-steps ride the epilogue, nt stores
    in_asm, offenders, count = False, [], {}
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
        op = s.split()[0]
        if in_asm:
            count[op] = count.get(op, 0) + 1
            continue
        hi = [int(m) for m in re.findall(r"\bv(\d+)\b", s)] + [int(m) for m in re.findall(r"\bv\[\d+:(\d+)\]", s)]
        if re.search(r"(^|[\s,\[])a\[?\d", s) or "accvgpr" in s or s.startswith("scratch_") or any(r >= NVGPR for r in hi):
            offenders.append(s)
    assert not offenders, offenders[:5]
    # M0 (the LDS destination of a DMA piece) is written by asm statements one MFMA ahead of the loads that use it: behind the prologue
    # (the builtin form of the first four k-steps' pieces, in front of the first s_barrier) the compiler must not touch it
    first_bar = body.index("s_barrier")
    in_asm, stray = False, []
    for ln in body[first_bar:].splitlines():
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif not in_asm and re.search(r"\bm0\b", t) and not t.startswith(";"):
            stray.append(t)
    assert not stray, stray[:5]
    # no waterfall loop: a store descriptor the compiler cannot prove wave-uniform gets every buffer_store wrapped in
    # v_readfirstlane / s_and_saveexec loops (cdna_hip_programming.md T20) -- measured once in round 5: +8 % on the whole kernel
    assert "s_and_saveexec_b64" not in body and "v_readfirstlane_b32" not in "".join(
        ln for ln in body.splitlines() if "buffer_store" in ln)
    # plain k-steps: 0 and 1 of a workgroup's first tile, 2 and 3 of every tile, the ring's steady round: 8 x 64; the boundary phase
    # in two copies (with / without the edge mask): 128 MFMAs of k-steps 0 and 1 + 32 row sums each; the last tile's epilogue: 32
    assert count["v_mfma_f32_16x16x32_bf16"] == 8 * 64 + 2 * (128 + 32) + 32
    assert count["v_exp_f32"] == 3 * 192               # 48 VGPR-resident blocks x 4, three copies of the epilogue
    assert count["v_cvt_pk_bf16_f32"] == 3 * 96
    assert count["v_accvgpr_read_b32"] == 3 * (64 + 2)  # the 16 AGPR-resident blocks + the two row-sum registers a lane stores
    assert count["ds_read_b128"] == 8 * 16 + 16 + 2 * 32   # fragment reads: per plain k-step, the prologue, k-steps 1 and 2 in each phase copy
    # a plain k-step's instruction stream (the ring's steady round, between its last two s_barrier): 64 MFMAs, 16 fragment reads, 8
    # DMA pieces and 7 more -- two M0 writes (one per operand's four pieces, one MFMA ahead of the first), one scalar add, one move,
    # two waits, the barrier.  Every extra instruction here is an issue slot of the one wave that also feeds the matrix pipe: the
    # builtin form of the DMA (an M0 write + a wait state + a scalar offset add per piece) was 117 (profiles/r05_gexp_v6.txt (o))
    ins = [ln.strip() for ln in body.splitlines() if ln.strip() and ln.strip()[0] not in ";."]
    bars = [i for i, x in enumerate(ins) if x == "s_barrier"]
    step = ins[bars[-2]:bars[-1]]
    assert sum(x.startswith("v_mfma") for x in step) == 64 and sum(x.startswith("ds_read_b128") for x in step) == 16
    assert sum(x.startswith("buffer_load_dwordx4") and x.endswith("lds") for x in step) == 8
    assert sum("m0" in x for x in step) == 2 and len(step) <= 96, len(step)
    meta = k1s_asm[k1s_asm.index(".amdhsa_kernel " + name):]
    meta = meta[:meta.index(".end_amdhsa_kernel")]
    assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta)
    assert re.search(r"\.amdhsa_next_free_vgpr 512\b", meta) and re.search(r"\.amdhsa_accum_offset 256\b", meta)
    tail = k1s_asm[k1s_asm.index(".name:", k1s_asm.index("amdhsa.kernels")):]
    kmeta = tail[tail.index(name) - 2000:tail.index(name) + 2000]
    assert re.search(r"\.vgpr_spill_count:\s+0\b", kmeta) and re.search(r"\.sgpr_spill_count:\s+0\b", kmeta)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_k1s_is_the_only_exp_gemm_in_the_product(k1s_asm):
    """Round 5 retired the 12-wave, the piece-major 4-wave and the v4 kernels that rounds 2-4 kept as shape fallbacks."""
    names = set(re.findall(r"gemm_nt_bf16_exp\w*?_kernel", k1s_asm))
    assert names == {"gemm_nt_bf16_exp_v6_kernel"}, names


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_k1_fp32_issues_the_next_tiles_dma_between_the_mfmas(k1s_asm):
    """K1 (gemm_nt_f32_dma_kernel, the product's SPREAD = 1 form): inside the K loop no two `buffer_load ... lds` follow each other
    without MFMAs between them -- eight of them back to back behind the barrier were ~1 200 idle cycles of the matrix pipe per K-tile for
    a workgroup alone on its CU (profiles/r05_k1_notes.txt) -- and the compiler has put no vmcnt wait of its own into the loop."""
    for kb in ("Lb1E", "Lb0E"):
        names = sorted(set(re.findall(r"^(_Z\w*gemm_nt_f32_dma_kernelI%sLi32ELi1ELb1E\w*):" % kb, k1s_asm, flags=re.M)))      # SPREAD 1, buffer-store epilogue
        assert len(names) == 1, names
        a = k1s_asm.index(names[0] + ":")
        body = k1s_asm[a:k1s_asm.index(".Lfunc_end", a)]
        lines = [ln.strip() for ln in body.splitlines() if ln.strip() and not ln.strip().startswith((";", "."))]
        assert sum(1 for ln in lines if ln.startswith("v_mfma_f32_32x32x2_f32")) == 64                   # ONE copy of the K-tile's MFMAs
        bar = [i for i, ln in enumerate(lines) if ln == "s_barrier"]
        assert len(bar) == 1, bar
        dma = [i for i, ln in enumerate(lines) if ln.startswith("buffer_load_dwordx4") and ln.endswith("lds")]
        assert len(dma) == 16 and sum(1 for i in dma if i < bar[0]) == 8, dma                            # the first tile's eight in front of the loop
        loop_dma = [i for i in dma if i > bar[0]]
        for i, j in zip(loop_dma, loop_dma[1:]):       # (the blocks that hold them follow each other in the text in both instantiations)
            between = sum(1 for ln in lines[i + 1:j] if ln.startswith("v_mfma_f32_32x32x2_f32"))
            assert between == 4, (i, j, between)
        assert sum(1 for ln in lines[bar[0]:loop_dma[0]] if ln.startswith("v_mfma_f32_32x32x2_f32")) == 2   # the first one behind two MFMAs
        # the only vmcnt waits of the kernel: the written one and __syncthreads' in front of the barrier
        assert not [ln for ln in lines[bar[0] + 1:] if ln.startswith("s_waitcnt") and "vmcnt" in ln], "a vmcnt wait inside the K-tile"
        # the epilogue: 64 stores through ONE provably uniform descriptor (no waterfall loop), one address add per store at most
        st = [i for i, ln in enumerate(lines) if ln.startswith("buffer_store_dword")]
        assert len(st) == 64 and "s_and_saveexec_b64" not in body and not [ln for ln in lines if ln.startswith("global_store")]
        assert len(lines) - st[0] <= 64 + 70, len(lines) - st[0]
