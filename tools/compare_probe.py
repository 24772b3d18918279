#!/usr/bin/env python3
"""Compares a run of interop/probe (the real crates) with the oracle's expectations (tests/golden/recall_expectations.json).

    cd interop/probe && cargo run --release > /tmp/probe.json && cd ../.. && python3 tools/compare_probe.py /tmp/probe.json

Prints one line per key (ok / DIFFERS / missing) and, for a difference, which restated convention it points at; exit code 0 only
when every compared key agrees — at which point the [RECALL] conventions of DESIGN.md 2 are confirmed against the crates and the
oracle is pinned by reference-run values.  Exit 1: differences; 2: the probe output could not be read."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_probe(path: str) -> dict:
    text = open(path).read()
    try:
        return json.loads(text)
    except json.JSONDecodeError:
        # the informational block carries Rust-escaped Debug text, which is not always JSON: compare the rest
        cut = re.sub(r',?\s*"informational":.*\n', "\n", text)
        cut = re.sub(r",\s*}\s*$", "\n}", cut.strip())
        return json.loads(cut)


def norm(v):
    if isinstance(v, str):
        return v.lower().removeprefix("0x")
    if isinstance(v, list):
        return [norm(x) for x in v]
    return v


def compare(probe: dict, expect: dict):
    """-> (rows, n_bad): rows of (key, status, hint)"""
    skip = set(expect.get("_not_compared", []))
    hints = expect.get("_if_a_key_differs", {})

    def hint(key):
        for pat, h in hints.items():
            for p in pat.split(" / "):
                if re.fullmatch(p.replace("*", ".*"), key):
                    return h
        return ""

    rows, bad = [], 0
    for key, want in expect.items():
        if key.startswith("_") or key in skip:
            continue
        if key not in probe:
            rows.append((key, "missing from the probe output", hint(key)))
            bad += 1
        elif norm(probe[key]) == norm(want):
            rows.append((key, "ok", ""))
        else:
            rows.append((key, "DIFFERS", hint(key)))
            bad += 1
    return rows, bad


def main(argv):
    if len(argv) != 2:
        print(__doc__)
        return 2
    try:
        probe = load_probe(argv[1])
    except Exception as e:  # noqa: BLE001
        print(f"cannot read {argv[1]}: {e}")
        return 2
    expect = json.load(open(os.path.join(ROOT, "tests", "golden", "recall_expectations.json")))
    rows, bad = compare(probe, expect)
    for key, status, h in rows:
        print(f"{key:34s} {status}" + (f"   <- {h}" if h and status != "ok" else ""))
    print(f"\n{len(rows) - bad} of {len(rows)} conventions confirmed" + ("" if bad else ": the oracle's restatement matches the crates on every probed value"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
