"""Regenerates tests/golden/*.orz from the committed inputs with the CPU oracle.

The reference (Rust, nightly) cannot be built in this environment and ships no golden streams, so
these vectors pin the ORACLE's behaviour (regression guard) -- they are not outputs of the Rust
binary.  The hand-derived known answers of SURVEY.md A.8 (checked in tests/test_oracle.py) and the
reference's own unit test (src/coder.rs:224-265) are the independent anchors.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "tools"))
import _data  # noqa: E402
import _oracle  # noqa: E402

INPUTS = {
    "can": _data.SMALL_CASES["can"],
    "text20k": _data.text(20000, seed=21),
    "zeros20k": _data.zeros_noise(20000),
    "mixed30k": _data.mixed(30000, seed=22),
    "period3": _data.periodic(5000, 3),
}

if __name__ == "__main__":
    for name, data in INPUTS.items():
        with open(os.path.join(HERE, name + ".in"), "wb") as f:
            f.write(data)
        for level in (0, 1, 2):
            with open(os.path.join(HERE, "%s.l%d.orz" % (name, level)), "wb") as f:
                f.write(_oracle.encode(data, level))
    print("wrote", len(INPUTS) * 4, "files")
