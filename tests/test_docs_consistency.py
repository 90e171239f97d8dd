"""The documents that describe the library are checked against the sources they describe (CPU only, no compute):
every tuning switch the kernels read is in DESIGN.md §10 and in the GPU switch test, every C-ABI entry point of the header is
mentioned in INTEGRATION.md or DESIGN.md, and the profile files the documents cite exist."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bijectors.jl_amd", "csrc")


def _read(*parts):
    with open(os.path.join(ROOT, *parts)) as f:
        return f.read()


def _switches_in_sources():
    names = set()
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".h")):
            names |= set(re.findall(r'(?:getenv|env_int)\("(BJX_[A-Z0-9_]+)"', _read("bijectors.jl_amd", "csrc", f)))
    return names


def test_every_tuning_switch_is_documented():
    design = _read("DESIGN.md")
    section = design[design.index("## 10. Tuning switches"):]
    documented = set(re.findall(r"`(BJX_[A-Z0-9_]+)`", section))
    missing = _switches_in_sources() - documented
    assert not missing, f"switches read by csrc/ but absent from DESIGN.md §10: {sorted(missing)}"
    # what §10 lists and the kernels do not read must be a Python-side variable (BJX_LIB_PATH) — not a leftover of a removed switch
    stale = documented - _switches_in_sources()
    py = "".join(_read("bijectors.jl_amd", f) for f in os.listdir(os.path.join(ROOT, "bijectors.jl_amd")) if f.endswith(".py"))
    assert all(s in py for s in stale), f"DESIGN.md §10 lists switches nothing reads: {sorted(s for s in stale if s not in py)}"


def test_every_tuning_switch_has_a_gpu_test_setting():
    src = _read("tests", "test_gpu_env_switches.py")
    block = re.search(r"SETTINGS = \[(.*?)\]\n", src, re.S).group(1)
    covered = set(re.findall(r'"(BJX_[A-Z0-9_]+)=', block))
    assert _switches_in_sources() == covered, (sorted(_switches_in_sources() - covered), sorted(covered - _switches_in_sources()))


def test_every_entry_point_of_the_header_is_described():
    header = _read("include", "bjx.h")
    entries = set(re.findall(r"\b(bjx_[a-z0-9_]+)\s*\(", header))
    entries = {e for e in entries if re.search(r"\b(?:int|void|const char\s*\*|bjx_ctx\s*\*|double|int64_t|uint32_t)\s+\*?\s*" + e + r"\s*\(", header)}
    assert len(entries) >= 60, len(entries)
    docs = _read("INTEGRATION.md") + _read("DESIGN.md") + _read("README.md")
    binding = _read("julia", "BijectorsBJX.jl")
    undocumented = sorted(e for e in entries if e not in docs and e not in binding)
    assert not undocumented, f"declared in include/bjx.h, mentioned nowhere: {undocumented}"


def test_cited_profile_files_exist():
    cited = set()
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        cited |= set(re.findall(r"`(profiles/[A-Za-z0-9_./\-]+?\.(?:md|json|csv|txt|jsonl))`", _read(doc)))
    missing = sorted(c for c in cited if not os.path.exists(os.path.join(ROOT, c)))
    assert not missing, f"documents cite profile files that are not in the tree: {missing}"


def test_readme_batch_exponents_match_the_bench_defaults():
    """the README's results table said 64 × 2²² for the headline that bench.py runs at 2²⁴.  Every BASELINE row of
    the table names its batch as a superscript power of two; it must be bench.DEFAULT_LOG2's."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_for_docs", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sup = str.maketrans("⁰¹²³⁴⁵⁶⁷⁸⁹", "0123456789")
    rows = [l for l in _read("README.md").splitlines() if re.match(r"\| C[1-5][ab]? ", l)]
    seen = {}
    for l in rows:
        name = re.match(r"\| (C[1-5][ab]?) ", l).group(1).lower()
        m = re.search(r"2([⁰¹²³⁴⁵⁶⁷⁸⁹]+)", l.split("|")[1])
        if m:
            seen[name] = int(m.group(1).translate(sup))
    assert {"c2", "c3", "c4", "c5a", "c5b"} <= set(seen), seen
    for name, lb in seen.items():
        want = 20 if name == "c1" else bench.DEFAULT_LOG2[name]      # c1: ONE vector of 2^20 elements
        assert lb == want, f"README row {name.upper()} says 2^{lb}, bench.py runs 2^{want}"
