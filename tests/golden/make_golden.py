#!/usr/bin/env python3
"""Transcribe the reference's committed known-answer values into a JSON fixture.

Run in the build container only (it reads /root/reference, which does not exist on
the GPU box).  Output: tests/golden/reference_norms.json -- pure data: for every
committed result log the `norm:` value of every size, with file:line provenance.

Columns: the reference prints 11 variants per line (bm04/bm05) or 5 (bm01).  All
valid variants agree to the printed 10 digits; bm05 column 7 ("Cuda (Coales)") is the
variant with the output-index bug (benchmark05/benchmark05.cc:193-194) and is skipped.
We record the value of column 9 ("Cuda (QP/Shared)", the variant to beat) for
bm04/bm05 and column 2 (Thrust) for bm01, and assert the other valid columns match.
"""
import glob
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_norms.json")


def parse(path, key, col, skip_cols=()):
    rows = []
    with open(path) as fh:
        for lineno, line in enumerate(fh, 1):
            tok = line.split()
            if len(tok) < 4 or tok[2] != "norm:" or tok[0] != key:
                continue
            vals = tok[3:]
            ref = vals[col]
            for c, v in enumerate(vals):
                if c in skip_cols:
                    continue
                assert v == ref, (path, lineno, c, v, ref)
            rows.append({"n": int(tok[1]), "norm": ref, "line": lineno})
    return rows


def main():
    out = {"_provenance": "transcribed by tests/golden/make_golden.py from the committed "
                          "logs of CFD-Xing/gpu-benchmarking; values kept as printed "
                          "(setprecision(10))",
           "hex": {}, "quad": {}, "l2norm": {}}
    for path in sorted(glob.glob(f"{REF}/benchmark05/nq*.log")):
        nq = int(re.match(r"nq(\d+)x", os.path.basename(path)).group(1))
        out["hex"][str(nq)] = {"file": os.path.relpath(path, REF),
                               "rows": parse(path, "nelmt", 8, skip_cols=(6,))}
    for path in sorted(glob.glob(f"{REF}/benchmark04/nq*.log")):
        nq = int(re.match(r"nq(\d+)x", os.path.basename(path)).group(1))
        out["quad"][str(nq)] = {"file": os.path.relpath(path, REF),
                                "rows": parse(path, "nelmt", 8)}
    path = f"{REF}/benchmark01/outfile.log"
    out["l2norm"] = {"file": os.path.relpath(path, REF), "rows": parse(path, "Size", 1)}
    path = f"{REF}/benchmark02/outfile.log"
    out["vecadd"] = {"file": os.path.relpath(path, REF), "rows": parse(path, "Size", 1)}
    path = f"{REF}/benchmark03/outfile.log"
    out["matvec"] = {"file": os.path.relpath(path, REF), "rows": parse(path, "Size", 1)}
    # grammar fixtures: the first lines (banner + two sizes) of one log per benchmark, verbatim
    # result data of the reference, used by tests/test_logfmt.py to pin the log grammar
    out["log_excerpts"] = {}
    for rel, nlines in (("benchmark05/nq8x8x8.log", 10), ("benchmark04/nq8x8.log", 10),
                        ("benchmark01/outfile.log", 9), ("benchmark02/outfile.log", 9),
                        ("benchmark03/outfile.log", 9)):
        with open(f"{REF}/{rel}") as fh:
            out["log_excerpts"][rel] = "".join(fh.readlines()[:nlines])
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1)
    n = sum(len(v["rows"]) for v in out["hex"].values()) + \
        sum(len(v["rows"]) for v in out["quad"].values()) + len(out["l2norm"]["rows"])
    assert n == 216
    n += len(out["vecadd"]["rows"]) + len(out["matvec"]["rows"])
    print(f"wrote {OUT}: {n} known-answer values")


if __name__ == "__main__":
    main()
