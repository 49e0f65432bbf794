"""Generates tests/golden/synthetic_golden.json with the CPU oracle.

Run from the repo root:  python tests/golden/make_golden.py [--full]
Inputs are the library's seeded SplitMix64 sequences (uniform A..Z, the
distribution of the reference's own generator, main.c:73,81); expected outputs
are the oracle's (offset, char_offset, ch, score %.17g).  --full adds the
BASELINE.json full-size cases (1M x 10k; minutes of CPU time).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))

from oracle import oracle  # noqa: E402
import psa_amd  # noqa: E402  (generator only; no GPU needed)

W_README = [1.5, 2.6, 0.3, 0.2]
CASES = [
    # name, seed1, seed2, n, m, weights, is_max
    ("tiny_max", 11, 12, 200, 7, [1, 1, 1, 1], True),
    ("tiny_min", 11, 12, 200, 7, [1, 3, 4, 2], False),
    ("small_max", 21, 22, 5000, 100, W_README, True),
    ("small_min", 21, 22, 5000, 100, W_README, False),
    ("odd_max", 31, 32, 20011, 257, [0.8, 0.54, 2.6, 13.7], True),
    ("odd_min", 31, 32, 20011, 257, [0.8, 0.54, 2.6, 13.7], False),
    ("c2_max", 1001, 2001, 100000, 1000, W_README, True),     # BASELINE.json configs[1]
    ("c2_min", 1001, 2001, 100000, 1000, W_README, False),
]
FULL = [
    ("c3_max", 1002, 2002, 1000000, 10000, W_README, True),   # BASELINE.json configs[2]
    ("c3_min", 1002, 2002, 1000000, 10000, W_README, False),
]
# --acgt: a plausible low-entropy workload at full size: four-letter alphabet (numpy default_rng(4).choice over b"ACGT",
# Seq1 first, then Seq2 — the same draw bench.py's `low_entropy_acgt` leg and the GPU test make)
ACGT = [
    ("acgt_max", None, None, 1000000, 10000, W_README, True),
    ("acgt_min", None, None, 1000000, 10000, W_README, False),
]
# --protein: a protein-like input at full size: the 20 amino-acid letters with their natural frequencies (Robinson &
# Robinson percentages), numpy default_rng(5): Seq1 first, then Seq2 — bench.py's `low_entropy_protein` leg and the GPU test
PROTEIN = [
    ("protein_max", "protein", None, 1000000, 10000, W_README, True),
    ("protein_min", "protein", None, 1000000, 10000, W_README, False),
]
PROTEIN_LETTERS = b"ARNDCQEGHILKMFPSTWYV"
PROTEIN_PERCENT = [7.8, 5.1, 4.5, 5.4, 1.9, 4.3, 6.3, 7.4, 2.2, 5.1, 9.1, 5.7, 2.2, 3.9, 5.2, 7.1, 5.8, 1.3, 3.2, 6.4]


def protein_pair(n, m):
    import numpy as np
    rng = np.random.default_rng(5)
    ab = np.frombuffer(PROTEIN_LETTERS, np.uint8)
    p = np.asarray(PROTEIN_PERCENT) / np.sum(PROTEIN_PERCENT)
    return rng.choice(ab, n, p=p).tobytes(), rng.choice(ab, m, p=p).tobytes()


def acgt_pair(n, m):
    import numpy as np
    rng = np.random.default_rng(4)
    ab = np.frombuffer(b"ACGT", np.uint8)
    return rng.choice(ab, n).tobytes(), rng.choice(ab, m).tobytes()


def main():
    out_path = os.path.join(ROOT, "tests", "golden", "synthetic_golden.json")
    special = "--acgt" in sys.argv or "--protein" in sys.argv
    cases = ([] if special else CASES) + (FULL if "--full" in sys.argv else []) + \
            (ACGT if "--acgt" in sys.argv else []) + (PROTEIN if "--protein" in sys.argv else [])
    old = {}
    if os.path.exists(out_path):
        old = {c["name"]: c for c in json.load(open(out_path))["cases"]}
    done = []
    threads = os.cpu_count() or 1
    for name, seed1, seed2, n, m, w, is_max in cases:
        if seed1 == "protein":
            s1, s2 = protein_pair(n, m)
        elif seed1 is None:
            s1, s2 = acgt_pair(n, m)
        else:
            s1 = psa_amd.generate_sequence(seed1, n)
            s2 = psa_amd.generate_sequence(seed2, m)
        t0 = time.time()
        score, off, ci, ch = oracle.search(s1, s2, w, is_max, threads=threads)
        dt = time.time() - t0
        done.append(dict(name=name, seed1=seed1, seed2=seed2, n=n, m=m, weights=w, is_max=is_max,
                         offset=off, char_offset=ci, ch=ch, score="%.17g" % score,
                         oracle_seconds=round(dt, 3), oracle_threads=threads))
        print(name, off, ci, ch, "%.17g" % score, f"{dt:.2f}s", flush=True)
    names = {c["name"] for c in done}
    done += [c for k, c in old.items() if k not in names]   # keep full-size cases from an earlier --full run
    json.dump({"_provenance": "tests/golden/make_golden.py (CPU oracle, oracle/psa_oracle.c)", "cases": done},
              open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
