"""Builds libstitch_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.  No CPU fallback is built."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libstitch_amd.so")
SOURCES = ["fill_kernel.hip", "fill_local16.hip", "fill_regs.hip", "prealign_kernel.hip", "stitch_api.cpp", "host_align.cpp", "prealign.cpp"]
HEADERS = ["dp_core.h", "walk_core.h", "fill_common.h", "host_align.h", "prealign.h", os.path.join("..", "..", "include", "stitch_gpu.h")]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB_PATH] + \
          [os.path.join(CSRC, f) for f in SOURCES]
    for d in os.environ.get("STITCH_DEFINES", "").split():
        cmd.append("-D" + d)                    # experiments only
    cmd += os.environ.get("STITCH_HIPCC_FLAGS", "").split()          # experiments only
    if os.environ.get("STITCH_PROFILE_BUILD"):
        cmd.append("-DSTITCH_PROFILE")          # diagnostic build with in-kernel stamps (never shipped)
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout)
    if verbose:
        print(r.stdout)
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
