"""Deterministic synthetic inputs of SURVEY.md §8(d): construct databases and chimeric long reads.
PRNG = SplitMix64 (k = next() % range), so the same sets can be regenerated anywhere without files."""
import numpy as np

MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def below(self, n):
        return self.next() % n

    def uniform(self):
        return (self.next() >> 11) / float(1 << 53)


def _np_rng(seed):
    # bulk draws (sequence content, error positions) use numpy's PCG64 seeded from SplitMix64 for speed
    return np.random.Generator(np.random.PCG64(SplitMix64(seed).next()))


ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
COMP[:] = np.arange(256)
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b


def make_db(n_contigs, length, seed, prefix="construct"):
    """list of (name, bytes): iid uniform ACGT contigs (db50x5k: n=50, len=5000, seed=1001)."""
    g = _np_rng(seed)
    return [(f"{prefix}_{k:02d}", ACGT[g.integers(0, 4, size=length)].tobytes()) for k in range(n_contigs)]


def make_reads(db, n_reads, length, seed, sub=0.03, ins=0.02, dele=0.02, random_frac=0.10, both_strands=False,
               circular=False, max_segments=4, dup_every=50):
    """Chimeric reads: 1..max_segments segments, each a random window of a random contig (optionally reverse
    complemented), per-base substitution/insertion/deletion errors, trimmed/extended to exactly `length`;
    `random_frac` of the reads are pure random sequence; every `dup_every`-th read is repeated once."""
    rng = SplitMix64(seed)
    g = _np_rng(seed + 7919)
    arrs = [np.frombuffer(s, dtype=np.uint8) for _, s in db]
    reads = []
    while len(reads) < n_reads:
        if rng.uniform() < random_frac:
            r = ACGT[g.integers(0, 4, size=length)]
        else:
            nseg = 1 + rng.below(max_segments)
            cuts = sorted(rng.below(length) for _ in range(nseg - 1))
            bounds = [0] + cuts + [length]
            parts = []
            for k in range(nseg):
                seg_len = max(1, bounds[k + 1] - bounds[k])
                c = arrs[rng.below(len(arrs))]
                want = int(seg_len * 1.1) + 8            # slack for deletions
                start = rng.below(len(c))
                if circular:
                    idx = (start + np.arange(want)) % len(c)
                    piece = c[idx]
                else:
                    start = min(start, max(0, len(c) - min(want, len(c))))
                    piece = c[start:start + want]
                if both_strands and rng.below(2):
                    piece = COMP[piece[::-1]]
                # errors
                u = g.random(len(piece))
                keep = u >= dele
                subst = (u >= dele) & (u < dele + sub)
                piece = piece.copy()
                piece[subst] = ACGT[g.integers(0, 4, size=int(subst.sum()))]
                out = piece[keep]
                ins_mask = g.random(len(out)) < ins
                if ins_mask.any():
                    pos = np.nonzero(ins_mask)[0]
                    out = np.insert(out, pos, ACGT[g.integers(0, 4, size=len(pos))])
                parts.append(out[:seg_len])
            r = np.concatenate(parts)
            if len(r) < length:
                r = np.concatenate([r, ACGT[g.integers(0, 4, size=length - len(r))]])
            r = r[:length]
        reads.append(r.tobytes())
        if dup_every and len(reads) % dup_every == 0 and len(reads) < n_reads:
            reads.append(reads[-1])
    return reads[:n_reads]
