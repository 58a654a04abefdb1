#!/usr/bin/env python3
"""Extracts the known-answer vectors (inputs + expected values, i.e. DATA) of the reference's inline unit tests
into JSON fixtures.  Run once in the build container (where /root/reference exists); the JSON it writes is
committed, so nothing under tests/ reads /root/reference at test time.

Sources (all under /root/reference/fg-stitch-lib/src/align/):
  aligners/single_contig_aligner.rs:915-1773   63 tests  -> single_contig.json
  aligners/multi_contig_aligner.rs:465-737      9 tests  -> multi_contig.json
Only literals are extracted: sequences, scoring numbers, mode, circular flag and the arguments of
assert_alignment (arithmetic such as `14 - (5 + 1)` is evaluated).  The other fixtures (alignment.json,
packed_cell.json, aligners.json) are small enough that they were transcribed by hand from
alignment.rs:395-707, traceback/packed_length_cell.rs:193-259 and aligners/mod.rs:984-1003.
"""
import json
import os
import re
import sys

REF = "/root/reference/fg-stitch-lib/src/align/aligners"
HERE = os.path.dirname(os.path.abspath(__file__))


def strip_display(bases):  # fn s(), single_contig_aligner.rs:887-893
    return "".join(c for c in bases if c not in "- _").upper()


def split_tests(src):
    """Yields (name, line_number, body) for every `fn test_*() { ... }`."""
    for m in re.finditer(r"fn (test_\w+)\(\) \{", src):
        start = m.end()
        depth, k = 1, start
        while depth:
            c = src[k]
            depth += (c == "{") - (c == "}")
            k += 1
        yield m.group(1), src.count("\n", 0, m.start()) + 1, src[start:k - 1]


def eval_int(expr):
    expr = expr.replace("_", "")
    assert re.fullmatch(r"[0-9+\-*() \n]+", expr), expr
    return int(eval(expr))


def split_args(s):
    out, depth, cur = [], 0, ""
    for c in s:
        if c == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            depth += (c in "([") - (c in ")]")
            cur += c
    if cur.strip():
        out.append(cur.strip())
    return out


def single():
    src = open(f"{REF}/single_contig_aligner.rs").read()
    tests_at = src.index("pub mod tests")
    out = []
    for name, line, body in split_tests(src[tests_at:]):
        line += src.count("\n", 0, tests_at)
        x = strip_display(re.search(r'let x = s\("([^"]*)"\)', body).group(1))
        y = strip_display(re.search(r'let y = s\("([^"]*)"\)', body).group(1))
        sc = dict(match=1, mismatch=-1, gap_open=-5, gap_extend=-1, jump=-10)  # Default, :85-90
        mp = re.search(r"MatchParams::new\((-?[\d_]+), (-?[\d_]+)\)", body)
        if mp:
            sc["match"], sc["mismatch"] = eval_int(mp.group(1)), eval_int(mp.group(2))
        nw = re.search(r"SingleContigAligner::new\((-?[\d_]+), (-?[\d_]+), (-?[\d_]+), match_fn\)", body)
        if nw:
            sc["gap_open"], sc["gap_extend"], sc["jump"] = (eval_int(nw.group(k)) for k in (1, 2, 3))
        else:
            assert "SingleContigAligner::default()" in body, name
        js = re.search(r"set_jump_score\((-?[\d_]+)\)", body)
        if js:
            sc["jump"] = eval_int(js.group(1))
        go = re.search(r"aligner\.scoring\.gap_open = (-?[\d_]+);", body)
        if go:
            sc["gap_open"] = eval_int(go.group(1))
        mode = re.search(r"aligner\.(global|querylocal|targetlocal|local)\(&x, &y\)", body).group(1)
        circular = "set_circular(true)" in body
        args = split_args(re.search(r"assert_alignment\(\s*&alignment,(.*?)\);", body, re.S).group(1))
        assert len(args) == 7, (name, args)
        exp = dict(xstart=eval_int(args[0]), xend=eval_int(args[1]), ystart=eval_int(args[2]), yend=eval_int(args[3]),
                   score=eval_int(args[4]), cigar=json.loads(args[5]), length=eval_int(args[6]), start_contig_idx=0)
        out.append(dict(name=name, ref=f"single_contig_aligner.rs:{line}", mode=mode, x=x, y=y, scoring=sc,
                        circular=circular, expect=exp))
    return out


def multi():
    src = open(f"{REF}/multi_contig_aligner.rs").read()
    tests_at = src.index("pub mod tests")
    out = []
    for name, line, body in split_tests(src[tests_at:]):
        line += src.count("\n", 0, tests_at)
        seqs = {}
        for m in re.finditer(r'let (\w+) = (reverse_complement\()?s\("([^"]*)"\)\)?;', body):
            seqs[m.group(1)] = ("rc:" if m.group(2) else "") + strip_display(m.group(3))
        for m in re.finditer(r"let (\w+) = reverse_complement\(&(\w+)\);", body):
            seqs[m.group(1)] = "rc:" + seqs[m.group(2)]
        contigs = []
        if name == "test_many_contigs":
            sc = re.search(r"scoring_local_custom\((-?[\d_]+), (-?[\d_]+), (-?[\d_]+), (-?[\d_]+)\)", body)
            for k, v in enumerate(["x1", "x2", "x3", "x4"]):
                contigs.append(dict(name=f"contig-{k}", is_forward=True, seq=seqs[v], kind="local",
                                    scoring=[eval_int(sc.group(q)) for q in (1, 2, 3, 4)]))
        else:
            for m in re.finditer(r'add_contig\(\s*"(\w+)",\s*(true|false),\s*&(\w+),\s*false,\s*'
                                 r'(scoring_global\(\)|scoring_(global|local)_custom\((-?[\d_]+), (-?[\d_]+), (-?[\d_]+), (-?[\d_]+)\)),?\s*\)', body):
                if m.group(4) == "scoring_global()":
                    kind, sc = "global", [-1, -5, -1, -10]
                else:
                    kind, sc = m.group(5), [eval_int(m.group(q)) for q in (6, 7, 8, 9)]
                contigs.append(dict(name=m.group(1), is_forward=m.group(2) == "true", seq=seqs[m.group(3)], kind=kind,
                                    scoring=sc))
        assert contigs, name
        yname = "y1" if "y1" in seqs else "y"
        cases = []
        jumps = re.findall(r"set_jump_scores\((-?\d+), (-?\d+), (-?\d+)\)", body)
        asserts = re.findall(r"assert_alignment\(\s*&alignment,(.*?)\);", body, re.S)
        assert len(asserts) == max(1, len(jumps)), name
        for k, a in enumerate(asserts):
            args = split_args(a)
            assert len(args) == 8, (name, args)
            exp = dict(xstart=eval_int(args[0]), xend=eval_int(args[1]), ystart=eval_int(args[2]), yend=eval_int(args[3]),
                       score=eval_int(args[4]), start_contig_idx=eval_int(args[5]), cigar=json.loads(args[6]),
                       length=eval_int(args[7]))
            cases.append(dict(jump_scores=[int(v) for v in jumps[k]] if jumps else None, expect=exp))
        out.append(dict(name=name, ref=f"multi_contig_aligner.rs:{line}", contigs=contigs, y=seqs[yname], cases=cases,
                        note="scoring = [mismatch, gap_open, gap_extend, jump]; match is 1; kind global => all clip "
                             "penalties MIN_SCORE, local => 0 (multi_contig_aligner.rs:437-463); seq 'rc:' => "
                             "reverse complement (util/dna.rs:31-41) of what follows"))
    return out


if __name__ == "__main__":
    s = single()
    m = multi()
    assert len(s) == 63, len(s)
    assert len(m) == 9, len(m)
    json.dump(s, open(f"{HERE}/single_contig.json", "w"), indent=1)
    json.dump(m, open(f"{HERE}/multi_contig.json", "w"), indent=1)
    print(f"wrote {len(s)} single-contig and {len(m)} multi-contig vectors", file=sys.stderr)
