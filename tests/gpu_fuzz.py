"""GPU fuzz of the Local kernel's multi-tile paths (and the pre-alignment filter) against the oracle.  Not collected by pytest:
run `FUZZ_SECONDS=600 FUZZ_SEED=1000 python tests/gpu_fuzz.py` on a GPU box (round 1: 9971 cases, all equal; FUZZ_MODES=1 also draws the non-local modes, FUZZ_PREALIGN=1 sends every case through the pre-alignment filter)."""
import os, random, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stitch_amd
from oracle import oracle as orc
from test_gpu_parity import chimera, rand_seq


def draw(seed):
    """the case of one seed: targets, reads, the product's options, how many reads are checked, the environment of the run"""
    rng = random.Random(seed)
    T = rng.randint(1, 9)
    lens = [rng.choice([rng.randint(1, 40), rng.randint(200, 700), rng.randint(700, 2600)]) for _ in range(T)]
    targets = [(f"t{k}", rand_seq(rng, n)) for k, n in enumerate(lens)]
    double = rng.random() < 0.4
    opts = dict(double_strand=double, circular=rng.random() < 0.3, suboptimal=rng.random() < 0.25)
    if rng.random() < 0.4:
        opts.update(match_score=rng.choice([1, 2]), mismatch_score=rng.choice([-1, -4, -6]), gap_open=rng.choice([-6, -3, 0]), gap_extend=rng.choice([-2, -1]),
                    default_jump_score=rng.choice([-10, -5, -1]))
    if os.environ.get("FUZZ_MODES") and rng.random() < 0.5:       # the generic int32 kernel: the other clipping modes
        opts.update(mode=rng.choice(["query-local", "target-local", "global"]))
    elif os.environ.get("FUZZ_PREALIGN"):                             # every case through the filter: drawn bands, register-window / LDS-ring / full-matrix kernels
        opts.update(pre_align=True, pre_align_min_score=rng.choice([1, 20, 60, 80]), kmer_size=rng.choice([6, 8, 11, 13]), band_width=rng.choice([0, 5, 30, 50, 62, 70]),
                    pre_align_subset_contigs=rng.random() < 0.7)
    elif rng.random() < 0.2:
        opts.update(pre_align=True, pre_align_min_score=rng.choice([20, 60]), kmer_size=rng.choice([8, 11]), band_width=rng.choice([5, 30]))
    if os.environ.get("FUZZ_SUBOPT"): opts["suboptimal"] = True      # every case with one chain per contig: the joined walks of traceback_all
    nreads = rng.choice([1, 2, 5, 30])
    n_check = 6
    env = {}
    if os.environ.get("FUZZ_STREAM"):      # persistent teams (stitch_api.cpp run_jobs_streaming): few teams, few arena blocks, many reads, every one checked
        env = {"STITCH_REGS_MIN_ROWS": "0", "STITCH_STREAM_TEAMS": str(rng.choice([1, 2, 3, 5])), "STITCH_STREAM_BLOCKS": str(rng.choice([2, 3, 4, 9]))}
        nreads = rng.choice([9, 17, 40]); n_check = nreads
    if os.environ.get("FUZZ_REGS"): env = dict(env, STITCH_REGS_MIN_ROWS="0")      # every eligible read on fill_regs.hip, launch by launch unless FUZZ_STREAM
    big = [t for t in targets if len(t[1]) > 30] or targets
    reads = [chimera(rng, big, rng.randint(20, rng.choice([200, 900, 1600])), err=rng.choice([0.02, 0.08]), both=double) for _ in range(nreads)]
    return targets, reads, opts, n_check, env, lens


def oracle_opts(opts):
    return {{"match_score": "match", "mismatch_score": "mismatch", "default_jump_score": "jump_score"}.get(k, k): v for k, v in opts.items()}


def main():
    t_end = time.time() + float(os.environ.get("FUZZ_SECONDS", "600"))
    seed0 = int(os.environ.get("FUZZ_SEED", "1000"))
    n_ok = 0; n_undefined = 0; n_stream = 0; seed = seed0
    while time.time() < t_end:
        targets, reads, opts, n_check, env, lens = draw(seed); seed += 1
        os.environ.update(env)
        al = stitch_amd.Builder(**opts).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in targets])
        o = orc.Aligners(targets, **oracle_opts(opts))
        try:
            res = al.align(reads)
        except stitch_amd.StitchError as e:                             # the reference indexes out of range here: both must say so
            assert "shorter contig" in str(e), (seed - 1, str(e))
            n_undefined += 1
            continue
        n_stream += 1 if al.timing().get("stream_runs", 0) else 0
        for k, read in enumerate(reads[:n_check]):
            try:
                want = o.align(read)
            except RuntimeError as e:
                assert "out of range" in str(e), (seed - 1, str(e))
                break
            assert [c.key() for c in res[k][0]] == [c.key() for c in want], (seed - 1, k, opts, lens, len(read))
            if opts.get("pre_align"):
                assert res[k][1] == o.prealign_score(), (seed - 1, k)
        n_ok += 1
        if n_ok % 10 == 0:
            print("cases ok:", n_ok, "last seed", seed - 1, flush=True)
    print("DONE cases ok:", n_ok, "seeds", seed0, "..", seed - 1, "cases that ran persistent teams:", n_stream, "undefined in the reference:", n_undefined, flush=True)


if __name__ == "__main__":
    main()
