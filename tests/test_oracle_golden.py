"""Pins the CPU oracle to every fixture the reference holds for this path
(SURVEY.md §8c): input.txt -> output.txt, the README sign table, the README
worked examples, plus the survey-stage known answers (Appendix B)."""
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, DISTINCT_RECORDS


def test_output_txt_byte_for_byte(oracle, tmp_path):
    """input.txt record 0 -> output.txt (cpu_funcs.c:96-98, 373-378), all thread/rank splits."""
    want = open(os.path.join(GOLDEN, "reference_output.txt"), "rb").read()
    for threads, ranks in ((1, 1), (4, 1), (3, 2), (1, 8)):
        out = tmp_path / f"out_{threads}_{ranks}.txt"
        subprocess.run([oracle.CLI_PATH, os.path.join(GOLDEN, "reference_input.txt"), str(out),
                        "0", str(threads), str(ranks)], check=True, stdout=subprocess.DEVNULL)
        assert out.read_bytes() == want


def test_readme_sign_table(oracle):
    """README.md:105-133: 27x27 table printed by print_hash (cpu_funcs.c:322-349), '_' shown as ' '."""
    rows = open(os.path.join(GOLDEN, "readme_sign_table.txt")).read().split("\n")
    body = [r for r in rows if len(r) > 2 and r[1:3] == " |"]
    assert len(body) == 27
    symbols = [chr(ord("A") + k) for k in range(26)] + ["-"]
    for r, line in zip(symbols, body):
        assert line[0] == r
        cells = line[3:]
        for k, c in enumerate(symbols):
            shown = cells[2 * k] if 2 * k < len(cells) else " "
            got = oracle.table_sign(r, c)
            assert (got if got != "_" else " ") == shown, (r, c)


def test_readme_examples(oracle):
    ex = json.load(open(os.path.join(GOLDEN, "readme_examples.json")))
    p = ex["pairwise"]
    _, signs = oracle.score_and_signs(p["seq1"].encode(), p["seq2"].encode(), [1, 1, 1, 1], 0)
    assert signs == p["signs"]
    for a in ex["alignments"]:
        score, signs = oracle.score_and_signs(a["seq1"].encode(), a["seq2"].encode(), a["weights"], a["offset"])
        assert signs == a["signs"]
        assert score == pytest.approx(a["score"], abs=1e-9)
    rule = ex["substitution_rule"]
    for old, new in rule["forbidden"]:
        assert oracle.table_sign(old, new) == ":"
    for old, new in rule["allowed"]:
        assert oracle.table_sign(old, new) != ":"


def test_survey_appendix_b(oracle, reference_records):
    """All 10 distinct stacked records; record 5 is the sequential-rounding trap
    (offset 9, an order-independent sum would pick offset 2)."""
    gold = json.load(open(os.path.join(GOLDEN, "survey_appendix_b.json")))["records"]
    assert [g["rec"] for g in gold] == DISTINCT_RECORDS
    for g in gold:
        w, s1, s2, is_max = reference_records[g["rec"]]
        assert (len(s1), len(s2), "max" if is_max else "min") == (g["n"], g["m"], g["mode"])
        score, off, ci, ch = oracle.search(s1, s2, w, is_max, threads=4)
        assert (off, ci, ch) == (g["offset"], g["char_offset"], g["new"])
        assert chr(s2[ci]) == g["old"]
        assert "%.17g" % score == g["score"]


def test_survey_delta_sets_pin_max_mode(oracle, psa):
    """The one independent MAX-mode fact this repo holds (VERDICT r4 missing #2).  The reference ships no MAX fixture
    (`output.txt` is a MIN run, README.md prints no MAX result), so get_max_substitute (cuda_funcs.cu:320-345) and
    get_optimal_substitute (cuda_funcs.cu:396-421) are pinned by the line-cited restatement only.  SURVEY.md Appendix B
    ("Substitution-table facts") records what the REFERENCE'S OWN functions gave when the survey compiled them: the set of
    distinct substitution deltas over all pairs — w = 1 3 4 2: MAX {-3, 1, 3, 5}, MIN {-5, -3, -2, -1, 1, 2, 3};
    w = 1.5 2.6 0.3 0.2 (README.md:57): MIN 9 distinct values.  These are SURVEY-STAGE OUTPUTS of the reference, not
    reference-held fixtures; the oracle and psa_build_tables must both reproduce them, and stay inside the by-construction
    bounds (MAX <= 6 outcomes, MIN <= 10)."""
    import numpy as np
    symbols = [chr(ord("A") + k) for k in range(26)] + ["-"]

    def oracle_deltas(w, is_max):
        out = set()
        for a in symbols:
            for b in symbols:
                p = oracle.weight(oracle.table_sign(a, b), w)
                out.add(oracle.weight(oracle.table_sign(a, oracle.substitute(a, b, w, is_max)), w) - p)
        return out

    def table_deltas(w, is_max):
        return set(np.asarray(psa.build_tables(w, is_max).DELTA)[:27, :27].ravel().tolist())

    assert oracle_deltas([1, 3, 4, 2], True) == {-3.0, 1.0, 3.0, 5.0} == table_deltas([1, 3, 4, 2], True)
    assert oracle_deltas([1, 3, 4, 2], False) == {-5.0, -3.0, -2.0, -1.0, 1.0, 2.0, 3.0} == table_deltas([1, 3, 4, 2], False)
    readme_min = oracle_deltas([1.5, 2.6, 0.3, 0.2], False)
    assert len(readme_min) == 9 and readme_min == table_deltas([1.5, 2.6, 0.3, 0.2], False)
    for w in ([1, 3, 4, 2], [1.5, 2.6, 0.3, 0.2], [0.8, 0.54, 2.6, 13.7], [5, 4, 3, 2]):
        assert len(oracle_deltas(w, True)) <= 6 and len(oracle_deltas(w, False)) <= 10
        assert oracle_deltas(w, True) == table_deltas(w, True)
    # MAX rule facts the restatement cites (cuda_funcs.cu:326-333): a '.'/'_' pair is raised to '*' by the Seq1 char itself,
    # a '*' pair must leave it (mutation is mandatory, quirk 1) and prefers '.' over '_' on a tie ('>=', quirk 3)
    assert oracle.substitute("A", "W", [1, 3, 4, 2], True) == "A"
    assert oracle.table_sign("A", oracle.substitute("A", "A", [1, 2, 2, 2], True)) == "."


def test_substitute_never_missing(oracle):
    """get_substitute returns a character for all 27x27 pairs in both modes."""
    symbols = [chr(ord("A") + k) for k in range(26)] + ["-"]
    for w in ([1, 3, 4, 2], [1.5, 2.6, 0.3, 0.2], [0.8, 0.54, 2.6, 13.7]):
        for is_max in (True, False):
            for a in symbols:
                for b in symbols:
                    assert oracle.substitute(a, b, w, is_max) != ""
    # SUB can be '-' (cuda_funcs.cu:326-328)
    assert oracle.substitute("-", "A", [1, 1, 1, 1], True) == "-"


def test_golden_synthetic_matches_oracle(oracle, psa):
    """The committed synthetic vectors (tests/golden/make_golden.py) still reproduce."""
    gold = json.load(open(os.path.join(GOLDEN, "synthetic_golden.json")))
    for case in gold["cases"]:
        if case["n"] * case["m"] > 3_000_000:
            continue   # the larger ones are checked against the GPU in test_gpu_parity.py
        s1 = psa.generate_sequence(case["seed1"], case["n"])
        s2 = psa.generate_sequence(case["seed2"], case["m"])
        score, off, ci, ch = oracle.search(s1, s2, case["weights"], case["is_max"], threads=4)
        assert (off, ci, ch, "%.17g" % score) == (case["offset"], case["char_offset"], case["ch"], case["score"])
