"""Structural guards on the gfx950 code objects (hipcc cross-compiles here: no GPU needed).

fill_local16.hip issues its state prefetch by hand (`global_load_dwordx4` inside an asm statement) and waits for it with a
counted `s_waitcnt vmcnt(3)`: the compiler tracks neither the loads nor the registers they land in (cdna_hip_programming.md
5.7), so a compiler change could place a read or a copy of those registers between the load and its wait — wrong results on
some waves, or a memory fault when the register held an address (that happened once: DESIGN.md).  This test reads the
assembly and asserts that from every hand-issued load to the end of its basic block no instruction names a destination
register, that the counted waits are there, that the kernels use no scratch and spill no VGPR, and that the -DSTITCH_CHECK
diagnostic build compiles.
fill_regs.hip keeps a read's row state in registers: a VGPR spill inside its column loop would put state back into memory."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stitch_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def compile_asm(src, tmp_path, *defs):
    out = os.path.join(str(tmp_path), os.path.basename(src) + ".s")
    from stitch_amd import build as sbuild
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, src), "-o", out] + \
          sbuild.FILE_FLAGS.get(src, []) + list(defs)          # (the flags the product build gives this file)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    # (-S does not run the assembler: an operand an inline asm statement cannot take only shows when the listing is assembled)
    a = subprocess.run([os.path.join(os.path.dirname(HIPCC), "..", "lib", "llvm", "bin", "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", out, "-o", out + ".o"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert a.returncode == 0, a.stdout[-3000:]
    return open(out).read(), r.stdout


def resources(remarks):
    """{kernel: {field: int}} from -Rpass-analysis=kernel-resource-usage"""
    res, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return res


def regs_of(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return out


@pytest.fixture(scope="module")
def local16(tmp_path_factory):
    return compile_asm("fill_local16.hip", tmp_path_factory.mktemp("isa"))


def test_hand_issued_prefetch_registers_are_untouched_until_their_wait(local16):
    asm, _ = local16
    lines = asm.splitlines()
    beg = next(i for i, l in enumerate(lines) if "fill_local16_kernel" in l and l.rstrip().endswith(":") or re.match(r"_ZN6stitch19fill_local16_kernel\S*:", l))
    end = next(i for i in range(beg, len(lines)) if "s_endpgm" in lines[i])
    body = lines[beg:end + 1]
    # every asm statement (;;#ASMSTART .. ;;#ASMEND) that loads into registers
    blocks, i = [], 0
    while i < len(body):
        if body[i].strip().startswith(";;#ASMSTART"):
            j, dst = i + 1, set()
            while not body[j].strip().startswith(";;#ASMEND"):
                m = re.match(r"\s+global_load_dword(?:x\d)?\s+(\S+),", body[j])
                if m:
                    dst |= regs_of(m.group(1))
                j += 1
            if dst:
                blocks.append((i, j + 1, dst))
            i = j
        i += 1
    assert len(blocks) >= 8 and sum(len(d) for _, _, d in blocks) >= 8 * 4, "the hand-issued state loads are gone from the kernel"
    # From a hand-issued load to the end of its basic block nothing may name a destination register: the compiler does not know the
    # load is in flight, and the failure seen so far (a copy of the destination "right behind the asm", fill_local16.hip) sits
    # exactly there.  Beyond the block the kernel's own structure takes over: the top of the slot loop waits (the asm waits are
    # counted below), and scratch / spill counts of zero (next test) rule out the other way a destination can move.
    for i, j, dst in blocks:
        k = j
        while k < len(body) and not re.match(r"\.LBB\d+_\d+:", body[k]) and not re.match(r"\s+s_(c?branch|endpgm|setpc)", body[k]):
            ins = body[k].split(";")[0]
            if body[k].strip().startswith(";;#ASMSTART"):
                break                                   # (the next asm statement: another load of the same prefetch, or a wait)
            if re.match(r"\s+[a-z]", ins):
                assert not (regs_of(ins) & dst), f"a prefetch destination is touched in the block of its load: {body[k].strip()}"
            k += 1
    waits = [i for i, l in enumerate(body) if re.search(r"s_waitcnt vmcnt\((3|0)\)", l) and any(";;#ASMSTART" in x for x in body[max(0, i - 2):i])]
    assert len(waits) >= 6, "the counted waits of the hand-issued loads are gone"        # per column instance: top of slot (x2), after the loop


def test_fill_kernels_use_no_scratch_and_spill_no_vgpr(local16, tmp_path):
    _, remarks = local16
    r = [v for k, v in resources(remarks).items() if "fill_local16_kernel" in k]
    assert r and all(x["ScratchSize"] == 0 and x["VGPRs Spill"] == 0 for x in r), r
    asm, remarks = compile_asm("fill_regs.hip", tmp_path)
    res = resources(remarks)
    one = [v for k, v in res.items() if "fill_regs_kernelILi1" in k]           # plain and circular instance, each with 8-byte y-suffix records only and with the one-bit kind
    assert len(one) == 4 and all(x["VGPRs"] <= 256 and x["VGPRs Spill"] <= 4 for x in one), one
    assert all(x["Occupancy"] == 2 for x in one)        # two waves per SIMD: eight waves of 256 registers fill a CU's register file
    four = [v for k, v in res.items() if "fill_regs_kernelILi4" in k]
    assert len(four) == 4 and all(x["VGPRs Spill"] <= 16 for x in four), four   # (more than 64 contigs: three more granule registers per lane; the one-bit-record instances a few more)
    # Round 4: the instances of the headline workload and the kernels launched beside resident teams use NO scratch memory at all.  The persistent teams stay resident for a whole call while the host launches
    # the fix-up / walk kernels and copies beside them; measured on the MI355X, a resident kernel WITH a scratch allocation held every
    # other launch and copy of the process back until its waves left (gpurun_out/r4f, r4g: walks of 7-40 s, copies of 39 s), one without
    # did not (r4d).  The granule records of NQ = 4 therefore go through LDS, not through a register array indexed at run time.
    assert all(x["ScratchSize"] == 0 for x in one), one
    # (NQ = 4, more than 64 contigs: a handful of prologue values the epilogue needs again still go to scratch — outside the column loop,
    # checked below for the headline instances and here by size; cfg5's persistent runs are stable with it, gpurun_out/r4k, the walk kernels
    # beside them being scratch-free; getting these to zero as well is open)
    assert all(x["ScratchSize"] <= 48 for x in four), four
    _, wr = compile_asm("fill_kernel.hip", tmp_path)
    walk = {k: v for k, v in resources(wr).items() if "fixup_walk_kernel" in k or "fixup_only_kernel" in k or "walk_all_kernel" in k}
    assert len(walk) == 3 and all(v["ScratchSize"] == 0 for v in walk.values()), walk      # (the kernels launched beside the teams)
    assert all(v["VGPRs"] <= 120 for v in walk.values()), walk      # ... each wave fits beside a resident fill wave (256 of a SIMD's 512 registers free)
    # what little is spilled (values of the prologue that the final unpack needs again) stays OUTSIDE the column loop: no scratch
    # access between the loop's header and its last block, in the two instances that run the headline workload
    lines = asm.splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"_ZN6stitch16fill_regs_kernelILi1ELb[01]E\S*:", l)]
    assert len(starts) == 4       # (each with and without the one-bit y-suffix records)
    for beg in starts:
        end = next(i for i in range(beg, len(lines)) if "s_endpgm" in lines[i])
        hdr = next(i for i in range(beg, end) if "Loop Header: Depth=1" in lines[i] and "Child Loop" in lines[i + 1])
        label = re.match(r"\.L(BB\d+_\d+):", lines[hdr]).group(1)
        last = max(i for i in range(hdr, end) if f"Header={label} " in lines[i])
        assert last - hdr > 5000, "this is not the column loop"
        bad = [lines[i].strip() for i in range(hdr, last + 1) if re.match(r"\s+scratch_", lines[i])]
        assert not bad, bad[:5]


def test_diagnostic_build_compiles(tmp_path):
    asm, _ = compile_asm("fill_local16.hip", tmp_path, "-DSTITCH_CHECK")
    assert "fill_local16_kernel" in asm
