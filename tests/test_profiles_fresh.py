"""The rocprof evidence under profiles/ must come from the binary in the tree (the round-3 files named
`planar_reg2_kernel<8, false, 2>` when the shipped library only had four-parameter instantiations).  Every kernel named in
profiles/traffic.json and in the <tag>_<workload>_kernel_stats.csv files of its tag has to exist — same template arguments — among the
symbols of bijectors.jl_amd/libbjx_hip.so.  Runs without a GPU: the code objects sit uncompressed in the library's fat-binary section,
the mangled names are read out of the file and demangled with llvm-cxxfilt / c++filt."""
import csv
import glob
import json
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bijectors.jl_amd", "libbjx_hip.so")
TRAFFIC = os.path.join(ROOT, "profiles", "traffic.json")


def _norm(name):
    """`void (anonymous namespace)::k<float, 4>(args...) [clone .kd]` -> `k<float, 4>`."""
    s = name.strip().replace("(anonymous namespace)::", "")
    s = re.sub(r"\s*\[clone [^\]]*\]", "", s)
    s = re.sub(r"\.kd$", "", s)
    if s.startswith("void "):
        s = s[5:]
    depth = 0
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return s[:i].strip()
    return s.strip()


@pytest.fixture(scope="module")
def lib_kernels():
    if not os.path.exists(LIB):
        pytest.skip("libbjx_hip.so not built")
    filt = shutil.which("llvm-cxxfilt") or shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    if not os.path.exists(filt) and not shutil.which(filt):
        pytest.skip("no demangler")
    data = open(LIB, "rb").read()
    mangled = sorted({m.decode() for m in re.findall(rb"_Z[A-Za-z0-9_]{4,}", data)})
    out = subprocess.run([filt], input="\n".join(mangled), capture_output=True, text=True, timeout=300).stdout.splitlines()
    return {_norm(n) for n in out}


def _evidence_names():
    if not os.path.exists(TRAFFIC):
        return None, []
    traffic = json.load(open(TRAFFIC))
    tag = traffic.get("_tag")
    names = []
    for wl, entry in traffic.items():
        if isinstance(entry, dict) and entry.get("tag") == tag:
            names += [(f"traffic.json[{wl}]", k) for k in entry.get("kernels", [])]
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{tag}_*_kernel_stats.csv"))):
        for row in csv.DictReader(open(f)):
            n = row.get("Name") or row.get("Kernel_Name") or ""
            if n:
                names.append((os.path.basename(f), n))
    return tag, names


def test_norm():
    assert _norm("void (anonymous namespace)::chain_kernel<float, 4, (bool)1>(Args<float>, long) [clone .kd]") == "chain_kernel<float, 4, (bool)1>"
    assert _norm("bjx_finalize_kernel(double*, int)") == "bjx_finalize_kernel"


def test_profiled_kernels_exist_in_the_shipped_library(lib_kernels):
    tag, names = _evidence_names()
    if not tag or not names:
        pytest.skip("profiles/traffic.json carries no tag yet (scripts/collect_profiles.py writes it)")
    # kernel_stats.csv also lists what torch launched around the workload (fills, copies, softmax): a name is OURS when the library
    # holds its function template under any arguments; the dominant kernels of traffic.json are ours by construction
    bases = {k.split("<")[0] for k in lib_kernels}
    ours = [(src, n) for src, n in names if src.startswith("traffic.json") or _norm(n).split("<")[0] in bases]
    # names can be cut by the profiler's CSV (160 characters in traffic.json): compare on the common prefix
    missing = []
    for src, n in ours:
        k = _norm(n)
        if k in lib_kernels:
            continue
        if len(n) >= 150 and any(x.startswith(k[:120]) for x in lib_kernels):
            continue
        missing.append(f"{src}: {k}")
    assert not missing, f"profiles/ ({tag}) names kernels the library in the tree does not contain — stale evidence:\n  " + "\n  ".join(sorted(set(missing))[:20])
    assert len(ours) >= 5
