"""The C-ABI library: loads, exports every symbol include/orz_hip.h declares, and fails loudly
without a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "orz_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(orz_[a-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if not n.endswith("_fn")))


def test_library_is_built_in_tree():
    from orz_amd import _native

    assert os.path.exists(_native.LIB_PATH), "run __graft_entry__.build() first"
    assert _native.LIB_PATH.startswith(ROOT)


def test_every_declared_symbol_is_exported_and_bound():
    from orz_amd import _native

    lib = _native.load()
    declared = _declared()
    assert len(declared) >= 14
    bound = {n for n, _, _ in _native.SYMBOLS}
    for name in declared:
        assert hasattr(lib, name), "liborz_hip.so does not export " + name
        assert name in bound, "orz_amd/_native.py does not bind " + name


def test_level_map_matches_reference_cli():
    # src/main.rs:97-102
    import orz_amd

    assert [(c.match_depth, c.lazy_match_depth1, c.lazy_match_depth2) for c in map(orz_amd.cfg_for_level, (0, 1, 2))] == [
        (5, 3, 2), (15, 9, 6), (45, 27, 18)]
    with pytest.raises(ValueError):
        orz_amd.cfg_for_level(3)


def test_no_silent_cpu_fallback():
    """Without a HIP device the product must refuse to encode (the oracle is never a fallback)."""
    import orz_amd
    from orz_amd import _native

    if _native.load().orz_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(Exception):
        orz_amd.StreamEncoder(device=0, level=1)
    with pytest.raises(Exception):
        orz_amd.LZEncoder(device=0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "orz_amd")
    for d, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(d, f), errors="replace").read()
                assert "orz_oracle" not in src and "_oracle" not in src, f


def test_symrank_reciprocal_division_is_exact():
    """orz_symrank_kernel replaces sum/16/cnt by a multiply-high with floor(2^32/cnt)+1 (orz_kernels.h):
    exact on the whole reachable domain (sum < 2^21 => n < 2^17, cnt <= 390)."""
    import numpy as np

    n = np.arange(0, 1 << 17, dtype=np.uint64)
    for d in range(2, 392):
        m = np.uint64((1 << 32) // d + 1)
        assert (((n * m) >> np.uint64(32)) == n // np.uint64(d)).all()


def test_parse_kernel_keeps_three_waves_per_simd():
    """the sweep window (3072 segments = 256 CUs x 4 SIMDs x 3) assumes every wave of a launch is resident at once:
    the build records each kernel's registers / occupancy next to the library"""
    import __graft_entry__ as ge

    ge.build()
    path = ge.LIB + ".resources.txt"
    if not os.path.exists(path):
        pytest.skip("library was built without the resource remarks")
    blocks = open(path).read().split("Function Name: ")
    parse = [b for b in blocks if b.strip() and "ParseWave" in b.splitlines()[0]]
    assert parse, "no resource record for the parse kernel"
    fields = dict(ln.split(": ", 1) for ln in parse[0].splitlines()[1:] if ": " in ln)
    assert int(fields["Occupancy [waves/SIMD]"]) >= 3, fields
    assert int(fields["ScratchSize [bytes/lane]"]) == 0 and int(fields["VGPRs Spill"]) == 0, fields


def test_the_hottest_parse_kernel_of_the_fast_mode_does_not_spill():
    """FastEval is built for an occupancy target (ORZ_EVAL_WAVES, __graft_entry__.build): whatever the target, the shipped build
    must not spill -- round 5's did (7 registers, 20 B of scratch per lane at six waves) while the build script said it did not"""
    import __graft_entry__ as ge

    ge.build()
    path = ge.LIB + ".resources.txt"
    if not os.path.exists(path):
        pytest.skip("library was built without the resource remarks")
    blocks = open(path).read().split("Function Name: ")
    ev = [b for b in blocks if b.strip() and "orz_thread_kernel_occ" in b.splitlines()[0] and "FastEval" in b.splitlines()[0]]
    assert ev, "no resource record for FastEval's occupancy-targeted kernel"
    fields = dict(ln.split(": ", 1) for ln in ev[0].splitlines()[1:] if ": " in ln)
    assert int(fields["ScratchSize [bytes/lane]"]) == 0 and int(fields["VGPRs Spill"]) == 0 and int(fields["SGPRs Spill"]) == 0, fields
    assert int(fields["Occupancy [waves/SIMD]"]) >= 5, fields


def test_bench_traffic_artefact_is_committed():
    """bench.py reports `roofline.traffic` from the newest committed PMC summary (tools/profile_round.sh) and labels it
    `from_profile: <file>`: the file must exist and hold the per-launch bytes of the kernels the roofline rows are about"""
    import importlib.util
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    name, by_name = bench.newest_pmc_profile()
    assert name and name.startswith("profiles/") and os.path.exists(os.path.join(root, name))
    data = json.load(open(os.path.join(root, name)))
    assert data["by_name"] == by_name
    assert by_name["orz_symrank_kernel"] > 0 and by_name["orz_thread_kernel<FastEval>"] > 0
