"""bench.py prints ONE JSON line with the keys the driver reads (plus `roofline` and `cpu_baseline`); a small workload on
the GPU, checked field by field.  The CPU leg only checks that the script refuses to run without a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available() or os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    r = run_bench("--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and b"needs a GPU" in r.stderr + r.stdout


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    r = run_bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--reads-per-step", "6", "--read-len", "500", "--contigs", "4",
                  "--contig-len", "700", "--cpu-reads-per-worker", "2", "--cpu-prefix", "200", "--cpu-threads", "2")
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["metric"] == "reads_per_sec" and out["unit"] == "reads/s" and out["higher_is_better"] is True
    assert (out["n_gpus"], out["steps"], out["warmup"], out["scaling"], out["data"], out["vs_baseline"]) == (1, 2, 1, "weak", "synthetic", None)
    assert out["value"] > 0 and out["ms_per_step"] > 0 and abs(out["value"] - 6 * 1000.0 / out["ms_per_step"]) < 1e-6 * out["value"]
    assert "workload" in out["config"] and "model" not in out["config"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["achieved"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["kernel"] in ("stitch::fill_local16_kernel", "stitch::fill_regs_kernel")
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 2 and cb["value"] > 0 and cb["unit"] == "reads/s" and cb["sample"]
    assert cb["sam_identical_on_sample"] is True and cb["sam_reads_compared"] == 4      # two timed reads per worker
    assert cb["warm"] is True and [x["threads"] for x in cb["scaling"]] == [1, 2] and cb["mcells_per_sec_per_thread"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_rehearsal():
    """the N > 1 path of bench.py (index broadcast, rank shards, barrier, max-over-ranks time, summed counters) with two
    ranks sharing the one GPU of the test box over gloo (STITCH_BENCH_DEVICE / STITCH_BENCH_BACKEND: rehearsal knobs; the
    driver's runs use one GPU per rank over RCCL)"""
    env = dict(os.environ, STITCH_BENCH_DEVICE="0", STITCH_BENCH_BACKEND="gloo", STITCH_ARENA_BYTES=str(8 << 30))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--reads-per-step", "5", "--read-len", "600", "--contigs", "4", "--contig-len", "800"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                   # rank 0 only
    out = json.loads(lines[0])
    assert (out["n_gpus"], out["steps"], out["warmup"], out["scaling"]) == (2, 2, 1, "weak")
    assert abs(out["value"] - 2 * 5 * 1000.0 / out["ms_per_step"]) < 1e-6 * out["value"]     # whole-job rate over both ranks
    assert "cpu_baseline" not in out                                # N = 1 only
