"""Hand-traced known answers for the origin re-alignment of circular contigs (`realign_origin`'s accept
rule is pinned by no reference test).  The vectors in tests/golden/realign_known_answers.json were derived BY HAND from the
reference's code, not produced by the oracle; both the oracle (CPU) and the product (GPU) must reproduce them.

One circular plasmid P of 40 bases (no 5-mer twice), CLI scoring (match 1, mismatch -4, gap -6 -2, jump -10), slop 20.

1. read = P[0:8] + P[25:40].  Local alignment of the read as it is: the suffix alone scores 15; prefix + same-contig jump + suffix
   scores 8 - 10 + 15 = 13; the circular zero-cost jump only leads from the contig's last row to its first (single_contig_aligner.rs:
   258-289), not from x = 8 to x = 25.  So the first alignment is 15 matches, x 25..40, y 8..23.  get_start_and_end_contig_indexes_
   for_realignment (mod.rs:365-409): xstart = 25 > slop: no contig at the start; xlen = 40 <= xend + slop: the contig at the END is P,
   and ystart = 8 > 0.  realign_origin (:516-549) rotates the read at ystart: P[25:40] + P[0:8], which aligns end to end across the
   origin: 15 matches, Xjump(P, 0) at no cost, 8 matches = 23.  realign_and_split_at_y (:411-430) accepts: 23 > 15, the new chain
   starts on P, the old one ends on P.  split_at_y(23 - 8 = 15) (alignment.rs:207-360): the pre-pivot half is the 15 matches (x 25..40,
   y 0..15), the jump at the pivot is skipped, the post-pivot half the 8 matches (x 0..8, y 15..23); joined post first: 8 matches, then
   Xjump(P, 25) because pre.xstart = 25 != post.xend = 8, no Yjump (23 + 0 - 23 = 0), then the 15 matches: x 0..40, y 0..23, score 23,
   debug cigar 8=17J15=.  The second rotation (:531-546) is the same read (no jump in the first alignment): 23 > 23 fails, no change.
2. read = P[0:15] + P[33:40] (the cut points are chosen so that the bases on either side of a junction differ: P[15] != P[33],
   P[14] != P[32]; otherwise the jump may sit a base or two away at the same score): the mirror image through the contig-at-START
   branch (:468-513): first alignment 15 matches x 0..15, y 0..15 (xstart = 0 <= slop, yend = 15 < 22; 15 - 10 + 7 = 12 is worse);
   rotation at yend: P[33:40] + P[0:15] = 7 matches, free jump, 15 matches = 22, accepted; split_at_y(22 - 15 = 7): post half first
   (15 matches, x 0..15), Xjump(P, 33), the 7 matches: 15=18J7=.
3. read = P[0:15] + 8 bases that occur nowhere in P: the rotated read scores 15 on P again, 15 > 15 fails: the chain stays 15=.
4. read = P[20:40] + P[0:20], the whole plasmid rotated: the FIRST alignment already crosses the origin by the free jump (40 matches,
   xstart = xend = 20); it starts within slop of the start and ends within slop of the end of the SAME contig: (None, None), no
   re-alignment (:385-389): 20=40j20=.
5. The accept rule's asymmetry, two circular plasmids A (= P) and B (no 4-mer in common): read = A[0:15] + B[10:25] + A[33:40].  As it
   is: A[0:15], jump, B[10:25] = 15 - 10 + 15 = 20 (adding A[33:40] behind a second jump would make 17); it starts at A's start
   (contig at the start = A, yend = 30 < 37) and ends on B (ystart = 0: nothing for the end branch).  The rotation at yend,
   A[33:40] + A[0:15] + B[10:25], has a chain that starts on A and ENDS on A — 7 matches, the free jump across A's origin, 15 matches =
   22 > 20, what traceback_from(n, A) returns (:419) — but the rule also wants the OLD chain to end on that contig
   (`best_alignment.end_contig_idx == contig_idx`, :423), and it ends on B: rejected.  The second rotation (at y = 15, the first base not
   on A: B[10:25] + A[33:40] + A[0:15]) ends best on A with 15 - 10 + 7 + 15 = 27 but starts on B: rejected too.  The chain stays
   15=1C5j15= with score 20.
6. traceback_all (traceback/mod.rs:152-217) and the suboptimal filter (aligners/mod.rs:318-329), linear contigs A, B and C = forty N
   (equal to no base of the read), read = A[5:25] + B[5:35], --suboptimal.  Best chain: 20 matches, jump, 30 matches = 40, ends on B; it
   marks A (start) and B (end, jumped to) as seen.  C is left: the best cell of C is in the LAST column — a cell of column j takes the
   jump from the best cell of column j - 1 (a B cell with 10 + (j - 21) for j > 20) and mismatches: cmax(j-1) - 10 - 4, largest at j = 50:
   39 - 14 = 25 (a diagonal step inside C would cost another -4, clipping the read earlier loses a base of B) — in every row alike, equal
   lengths, so the x-suffix rule (:406-429, a later row only on a longer length) keeps row 1: 20 matches, Xjump(B, 5), 29 matches,
   Xjump(C, 0), one substitution = 25 >= 20 % of 40: the second chain.  With A and B alone both are seen after the first chain: one chain."""
import json
import os

import pytest

from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "realign_known_answers.json")))
KIND = {"M": 0, "J": 6}


def expand(ops):
    out = []
    for o in ops:
        if o[0] == "M":
            out += [(0, 0, 0)] * o[1]
        elif o[0] == "X":
            out += [(1, 0, 0)] * o[1]
        else:
            out.append((6, o[1], o[2]))
    return out


def check(chain, want, ops_of):
    for f in ("score", "xstart", "xend", "ystart", "yend", "start_contig_idx", "end_contig_idx"):
        assert getattr(chain, f) == want[f], (f, getattr(chain, f), want[f])
    assert chain.xlen == 40
    assert [tuple(o) for o in ops_of(chain)] == expand(want["ops"])
    assert chain.cigar() == want["cigar"]


@pytest.mark.parametrize("case", VEC["cases"], ids=[c["name"] for c in VEC["cases"]])
def test_oracle_reproduces_the_hand_traced_realignment(case):
    o = orc.Aligners([("P", VEC["plasmid"])], circular=True)
    chains = o.align(case["read"])
    assert len(chains) == 1
    check(chains[0], case["want"], lambda c: c.ops)


@pytest.mark.gpu
@pytest.mark.parametrize("case", VEC["cases"], ids=[c["name"] for c in VEC["cases"]])
def test_product_reproduces_the_hand_traced_realignment(case):
    import stitch_amd
    al = stitch_amd.Builder(circular=True).build_aligners([stitch_amd.TargetSeq("P", VEC["plasmid"])], device=0)
    chains, _ = al.align([case["read"]])[0]
    assert len(chains) == 1
    check(chains[0], case["want"], lambda c: c.operations)


TWO = VEC["two_contig_cases"]


@pytest.mark.parametrize("case", TWO, ids=[c["name"] for c in TWO])
def test_oracle_keeps_the_accept_rules_asymmetry(case):
    o = orc.Aligners([("A", VEC["plasmid"]), ("B", VEC["second_plasmid"])], circular=True)
    chains = o.align(case["read"])
    assert len(chains) == 1
    check(chains[0], case["want"], lambda c: c.ops)


@pytest.mark.gpu
@pytest.mark.parametrize("case", TWO, ids=[c["name"] for c in TWO])
def test_product_keeps_the_accept_rules_asymmetry(case):
    import stitch_amd
    al = stitch_amd.Builder(circular=True).build_aligners([stitch_amd.TargetSeq("A", VEC["plasmid"]), stitch_amd.TargetSeq("B", VEC["second_plasmid"])], device=0)
    chains, _ = al.align([case["read"]])[0]
    assert len(chains) == 1
    check(chains[0], case["want"], lambda c: c.operations)


ALL = VEC["traceback_all_cases"]


@pytest.mark.parametrize("case", ALL, ids=[c["name"] for c in ALL])
def test_oracle_reproduces_the_hand_traced_suboptimal_chains(case):
    o = orc.Aligners([tuple(t) for t in case["targets"]], suboptimal=True)
    chains = o.align(case["read"])
    assert len(chains) == len(case["want"])
    for c, w in zip(chains, case["want"]):
        check(c, w, lambda c: c.ops)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ALL, ids=[c["name"] for c in ALL])
def test_product_reproduces_the_hand_traced_suboptimal_chains(case):
    import stitch_amd
    al = stitch_amd.Builder(suboptimal=True).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in case["targets"]], device=0)
    chains, _ = al.align([case["read"]])[0]
    assert len(chains) == len(case["want"])
    for c, w in zip(chains, case["want"]):
        check(c, w, lambda c: c.operations)


def test_the_plasmid_has_no_repeats():
    p = VEC["plasmid"]
    k5 = [(p + p)[i:i + 5] for i in range(len(p))]
    assert len(set(k5)) == len(k5) and len(p) == 40
