"""The shape of the headline kernel's machine code, checked on the CPU build (hipcc cross-compiles gfx950 without a GPU).

Round 5 found 6 % of the headline kernel in the ORDER of its instructions, not in their number: sixteen dependent
LDS round trips in the gather, `s_waitcnt vmcnt` waits inherited from the prologue that ran in every iteration, a wait for just-issued
stores in front of the staged row.  None of that is visible to a parity test, and a compiler update can bring any of it back.  This
test compiles `oct_fused_kernel<10, IN_U16, RS_CUBIC, MODE_LOG>` the way csrc/Makefile does and asserts the shape."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "octproz_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = "/opt/rocm/bin/hipcc"
KERNEL = "_ZN3oct16oct_fused_kernelILi10ELi1ELi2ELi4EEEvNS_9FusedArgsE"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("isa") / "fused_10_rs2.s")
    flags = re.search(r"^SCHED_ILP\s*:=\s*(.*)$", open(os.path.join(CSRC, "Makefile")).read(), re.M).group(1).split()
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-inline-asm", "-Wno-pass-failed", "-Wno-unused-value"] + flags +
                          ["-DOCT_LOG2N=10", "-DOCT_FUSED_RS=2", "-S", "--cuda-device-only", "-o", out, "fused_inst.hip"], cwd=CSRC, stderr=subprocess.DEVNULL)
    return out


def _loop(path):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = (0, 0, 0)
    for i, l in enumerate(body):
        m = re.search(r"s_(?:c)?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    return [l.strip() for l in body[best[1]:best[2] + 1] if l.strip() and not l.strip().startswith((";", "."))], lines[end:end + 400]


def test_headline_kernel_fits_two_waves_per_simd_without_scratch(asm):
    _, meta = _loop(asm)
    text = "\n".join(meta)
    assert int(re.search(r"; NumVgprs: (\d+)", text).group(1)) <= 256
    assert int(re.search(r"; ScratchSize: (\d+)", text).group(1)) == 0
    assert int(re.search(r"; Occupancy: (\d+)", text).group(1)) >= 2


def test_gather_issues_its_tap_reads_in_groups_and_waits_for_no_vector_memory(asm):
    loop, _ = _loop(asm)
    prio = [i for i, l in enumerate(loop) if l.startswith("s_setprio")]
    g0 = next(i for i in prio if loop[i].split()[1] == "3")   # gather
    g1 = next(i for i in prio if i > g0)                        # transform
    gather = loop[g0:g1]
    reads = [i for i, l in enumerate(gather) if l.startswith("ds_read2_b32")]
    assert len(reads) == 32, "four taps of sixteen samples as two ds_read2_b32 each"
    # the first sixteen reads (two groups of four samples) go out before the first wait
    first_wait = next(i for i, l in enumerate(gather) if l.startswith("s_waitcnt"))
    assert sum(1 for i in reads if i < first_wait) >= 16, "tap reads are issued sample by sample again (one LDS round trip per sample)"
    # prologue_wait(): the tables are not guarded by vmcnt waits inside the loop
    assert not any("vmcnt" in l for l in gather), "vmcnt waits in the gather: the prologue's loads are pending at loop entry again"
    # a handful of full drains at most (sixteen before round 5)
    assert sum(1 for l in gather if re.search(r"lgkmcnt\(0\)", l)) <= 3


def test_staged_row_waits_for_its_loads_not_for_the_stores_behind_them(asm):
    loop, _ = _loop(asm)
    stores = [i for i, l in enumerate(loop) if l.startswith("buffer_store")]
    loads = [i for i, l in enumerate(loop) if l.startswith("buffer_load")]
    assert len(stores) == 8 and len(loads) == 4
    waits = [int(re.search(r"vmcnt\((\d+)\)", l).group(1)) for l in loop if l.startswith("s_waitcnt") and "vmcnt" in l]
    assert waits, "the prefetched row is consumed somewhere"
    # four loads, then eight stores, are outstanding when the row is staged: waiting for the loads alone is vmcnt(11) ... vmcnt(8)
    assert min(waits) >= 8, "the staging waits for stores issued a moment ago (vmcnt(%d))" % min(waits)


def test_isa_sequence_tool_reads_the_same_file(asm):
    import isa_sequence
    seq = isa_sequence.sequence(asm, "oct_fused_kernelILi10ELi1ELi2ELi4EE")
    assert "P3" in seq and "P2" in seq and seq.count("r") >= 40


def test_in_store_sinusoidal_variants_have_no_scratch_and_read_their_work_list_with_scalar_loads(asm):
    """MODE_SINUS (round 6): the four image variants of the N = 1024 cubic kernel and the four with the rolling average fit their register
    budget without scratch, and the work-list entries arrive by s_load_dwordx4 (constant address space) -- not by a vector load per lane"""
    text = open(asm).read()
    for mode in (32, 36, 40, 44, 33, 37, 41, 45):
        name = "_ZN3oct16oct_fused_kernelILi10ELi1ELi2ELi%dEEEvNS_9FusedArgsE" % mode
        assert name + ":" in text, "MODE %d is not instantiated" % mode
        start = text.index(name + ":")
        end = text.index("s_endpgm", start)
        meta = text[end:end + 20000]
        assert int(re.search(r"; ScratchSize: (\d+)", meta).group(1)) == 0, mode
        assert int(re.search(r"; NumVgprs: (\d+)", meta).group(1)) <= 256, mode
        body = text[start:end]
        assert len(re.findall(r"\bs_load_dwordx4\b", body)) >= 3, "work-list entries are not read with scalar loads (mode %d)" % mode
        # the correction's blend is never contracted into an FMA with the difference: v_sub, v_mul, v_add per value (cu:506-510 bit for bit)
        assert "v_sub_f32" in body and "v_mul_f32" in body


def test_display_fold_variants_of_the_headline_object_have_no_scratch(asm):
    """MODE_DISP (the opt-in display frames from the image store, OCTPIPE_ROUTE_FUSED_DISPLAY): ADVICE r5 found variants that spill at N = 2048 /
    4096 -- those are no longer instantiated (fused_inst.hip, route.h); the ones of this object stay inside their register budget"""
    text = open(asm).read()
    for mode in (16, 20, 24, 28, 17, 21, 25, 29):
        name = "_ZN3oct16oct_fused_kernelILi10ELi1ELi2ELi%dEEEvNS_9FusedArgsE" % mode
        assert name + ":" in text, "MODE %d is not instantiated" % mode
        end = text.index("s_endpgm", text.index(name + ":"))
        assert int(re.search(r"; ScratchSize: (\d+)", text[end:end + 20000]).group(1)) == 0, mode


def test_in_store_sinusoidal_variants_of_the_1664_team_kernel_have_no_scratch(tmp_path):
    """team1664_kernel.h MODE_SINUS keeps the previous row's grey values in eight registers per lane (the LDS of four teams per CU has no room for
    them): all 24 variants -- cubic resampling with its 52 tap weights in registers included -- stay inside 256 VGPRs without scratch or AGPR copies"""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path / "team1664.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-inline-asm", "-Wno-pass-failed", "-Wno-unused-value", "-S", "--cuda-device-only",
                           "-o", out, "team1664_inst.hip"], cwd=CSRC, stderr=subprocess.DEVNULL)
    text = open(out).read()
    seen = 0
    for m in re.finditer(r"^(_ZN3oct19oct_team1664_kernelILi1ELi(\d)ELi(\d+)EEEvNS_9FusedArgsE):", text, re.M):
        if not int(m.group(3)) & 32:
            continue
        seen += 1
        end = text.index("s_endpgm", m.end())
        meta = text[end:end + 20000]
        assert int(re.search(r"; ScratchSize: (\d+)", meta).group(1)) == 0, m.group(1)
        assert int(re.search(r"; NumVgprs: (\d+)", meta).group(1)) <= 256 and int(re.search(r"; NumAgprs: (\d+)", meta).group(1)) == 0, m.group(1)
        assert "s_load_dwordx4" in text[m.end():end], "work-list entries are not read with scalar loads"
    assert seen == 24
