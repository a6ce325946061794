"""The reference's own benchmark file pins everything at once -- when it is there.  enwik8 (100,000,000 bytes) is
listed in the reference's .MISSING_LARGE_BLOBS and is nowhere on this image, so these tests skip;
the day `test/enwik8.xz` (or an ORZ_ENWIK8 path) is supplied they turn "parity unpinned" into a reference-held pin:
the sizes the reference publishes for its encoder (/root/reference/README.md:43-46, table of orz 1.6.2) must come
out of the oracle, byte-identically out of the GPU's exact mode, and within +-0.5 % out of the fast mode."""
import lzma
import os

import pytest

README_SIZES = {2: 26_892_825, 1: 27_217_825, 0: 27_898_433}  # README.md:43,45,46 (orz -l2 / -l1 / -l0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _enwik8(allow_reference_tree):
    cands = [os.environ.get("ORZ_ENWIK8", ""), os.path.join(ROOT, "tests", "data", "enwik8"), os.path.join(ROOT, "tests", "data", "enwik8.xz")]
    if allow_reference_tree:  # (the GPU box has no /root/reference: the gpu-marked test never looks there)
        cands += ["/root/reference/test/enwik8", "/root/reference/test/enwik8.xz"]
    for p in cands:
        if p and os.path.exists(p):
            raw = open(p, "rb").read()
            data = lzma.decompress(raw) if p.endswith(".xz") else raw
            if len(data) == 100_000_000:
                return data
    pytest.skip("enwik8 is not available (reference/.MISSING_LARGE_BLOBS); set ORZ_ENWIK8 or add tests/data/enwik8[.xz]")


@pytest.mark.parametrize("level", [0, 1, 2])
def test_oracle_reproduces_the_readme_sizes(oracle, level):
    data = _enwik8(True)
    out = oracle.encode(data, level)
    assert oracle.decode(out)[0] == data
    # the README's table was made with orz 1.6.2, this tree is 1.6.1: equality is expected, a few bytes would be a version note
    assert len(out) == README_SIZES[level]


@pytest.mark.gpu
@pytest.mark.parametrize("level", [0, 1, 2])
def test_gpu_modes_on_enwik8(oracle, level):
    import orz_amd

    data = _enwik8(False)
    exact = orz_amd.StreamEncoder(device=0, level=level, mode="exact")
    fast = orz_amd.StreamEncoder(device=0, level=level, mode="fast")
    try:
        a = exact.encode(data)
        b = fast.encode(data)
    finally:
        exact.close()
        fast.close()
    assert len(a) == README_SIZES[level]
    assert oracle.decode(b)[0] == data
    assert abs(len(b) - README_SIZES[level]) <= 0.005 * README_SIZES[level]
