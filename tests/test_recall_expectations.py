"""tests/golden/recall_expectations.json — what interop/probe (Rust, the real crates; compiled by nobody here: no toolchain) is
expected to print — is regenerated from the oracle on every run, so the file a maintainer compares against cannot rot; and
tools/compare_probe.py accepts the oracle's own values and flags a changed one."""
import importlib.util
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_expectations_file_is_what_the_oracle_computes():
    mk = _load(os.path.join(GOLD, "make_recall_expectations.py"), "make_recall_expectations")
    committed = json.load(open(os.path.join(GOLD, "recall_expectations.json")))
    fresh = dict(mk.NOTES)
    fresh.update(mk.expectations())
    assert json.loads(json.dumps(fresh)) == committed
    # anchors that do not come from the oracle: the multiplicative identity's Montgomery limbs as halo2curves prints them in its own
    # source (R mod r) and the published bn256 ZETA constant
    assert committed["fr_one_limbs"] == ["ac96341c4ffffffb", "36fc76959f60cd29", "666ea36f7879462e", "0e0a77c19a07df2f"]
    assert committed["fr_zeta_repr"] == bytes.fromhex("b3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd").rjust(32, b"\0")[::-1].hex()
    assert committed["g1_generator_bytes"] == (1).to_bytes(32, "little").hex()  # (1, 2): x = 1, y even -> no flag


def test_every_probe_key_has_an_expectation_and_the_compare_tool_works(tmp_path):
    src = open(os.path.join(ROOT, "interop", "probe", "src", "main.rs")).read()
    printed = set(re.findall(r'put\("([a-z0-9_]+)"', src))
    committed = json.load(open(os.path.join(GOLD, "recall_expectations.json")))
    expected = {k for k in committed if not k.startswith("_")}
    assert printed - {"informational"} == expected - {"srs5_secret"}
    # the reference's dependency pins, as the reference has them (Cargo.toml:13,16,18)
    cargo = open(os.path.join(ROOT, "interop", "probe", "Cargo.toml")).read()
    assert 'tag = "v2023_02_02"' in cargo and cargo.count('branch = "axiom-dev-0406"') == 2
    # compare tool: the oracle's own values pass, one changed value fails and names the convention
    probe = {k: v for k, v in committed.items() if not k.startswith("_") and k != "srs5_secret"}
    probe["informational"] = {"vk5_pinned_debug": "PinnedVerificationKey { .. }"}
    p = tmp_path / "probe.json"
    p.write_text(json.dumps(probe))
    tool = os.path.join(ROOT, "tools", "compare_probe.py")
    r = subprocess.run([sys.executable, tool, str(p)], capture_output=True, text=True)
    assert r.returncode == 0 and "every probed value" in r.stdout, r.stdout
    probe["fr_zeta_repr"] = probe["fr_root_of_unity_repr"]
    probe.pop("transcript_bytes")
    p.write_text(json.dumps(probe))
    r = subprocess.run([sys.executable, tool, str(p)], capture_output=True, text=True)
    assert r.returncode == 1
    assert re.search(r"fr_zeta_repr\s+DIFFERS.*coset generator", r.stdout) and re.search(r"transcript_bytes\s+missing", r.stdout)
