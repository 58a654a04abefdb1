"""Builds libstitch_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.  No CPU fallback is built."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libstitch_amd.so")
SOURCES = ["fill_regs32.hip", "fill_kernel.hip", "fill_local16.hip", "fill_regs.hip", "prealign_kernel.hip", "prealign_skew16.hip", "prealign_window.hip", "prealign_band.hip", "stitch_api.cpp", "host_align.cpp", "prealign.cpp"]
HEADERS = ["dp_core.h", "walk_core.h", "fill_common.h", "host_align.h", "prealign.h", os.path.join("..", "..", "include", "stitch_gpu.h")]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


# per-file compiler flags.  fill_regs.hip: its sweeps are long straight-line blocks of compare -> select pairs, and gfx950 wants two
# wait states between a vector compare's scalar result and the vector instruction that reads it; the default scheduler leaves ~300
# s_nop in the column loop, the ILP scheduler ~90 (each an issue slot of a wave that is short of them: -2 % launch time, measured).
FILE_FLAGS = {"fill_regs.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
              "fill_regs32.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]}      # (+3 % on the cfg2 shape in global mode, measured)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    for d in os.environ.get("STITCH_DEFINES", "").split():
        common.append("-D" + d)                 # experiments only
    common += os.environ.get("STITCH_HIPCC_FLAGS", "").split()       # experiments only
    if os.environ.get("STITCH_PROFILE_BUILD"):
        common.append("-DSTITCH_PROFILE")       # diagnostic build with in-kernel stamps (never shipped)
    if verbose:
        common.append("-Rpass-analysis=kernel-resource-usage")
    # one object per source (compiled side by side), then the link
    objdir = os.path.join(LIB_DIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    # an object is rebuilt when its source, a header or the flags changed (fill_regs32.hip alone takes two minutes)
    stamp = " ".join(common) + repr(sorted(FILE_FLAGS.items()))
    stamp_path = os.path.join(objdir, "flags.txt")
    same_flags = os.path.exists(stamp_path) and open(stamp_path).read() == stamp
    newest_header = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    for f in SOURCES:
        obj = os.path.join(objdir, f + ".o")
        objs.append(obj)
        if (not force and same_flags and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(os.path.join(CSRC, f))
                and os.path.getmtime(obj) > newest_header):
            continue
        cmd = common + FILE_FLAGS.get(f, []) + ["-c", os.path.join(CSRC, f), "-o", obj]
        procs.append((f, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    out, failed = "", []
    for f, obj, pr in procs:
        o, _ = pr.communicate()
        out += o
        if pr.returncode != 0:
            failed.append(f)
    if failed:
        raise RuntimeError("hipcc failed on " + ", ".join(failed) + ":\n" + out)
    with open(stamp_path, "w") as fh:
        fh.write(stamp)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + objs,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("linking libstitch_amd.so failed:\n" + r.stdout)
    if verbose:
        print(out)
    return LIB_PATH


CLI_SRC = os.path.join(HERE, "cli", "stitch_align.cpp")
CLI_PATH = os.path.join(HERE, "bin", "stitch-align")


def build_cli(force=False):
    """The `stitch align` front end (stitch_amd/cli/stitch_align.cpp): a plain C++ program over the C ABI."""
    lib = build()
    if not force and os.path.exists(CLI_PATH) and os.path.getmtime(CLI_PATH) > max(os.path.getmtime(CLI_SRC), os.path.getmtime(lib)):
        return CLI_PATH
    os.makedirs(os.path.dirname(CLI_PATH), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O2", "-std=c++17", "-o", CLI_PATH, CLI_SRC, "-L" + LIB_DIR, "-lstitch_amd", "-lz", "-Wl,-rpath,$ORIGIN/../lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("building stitch-align failed:\n" + r.stdout)
    return CLI_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB_PATH)
    print(build_cli(force="--force" in sys.argv))
