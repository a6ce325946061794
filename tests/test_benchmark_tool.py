"""tools/benchmark_tool.py, this repo's counterpart of the reference's benchmark-tool
(/root/reference/benchmark-tool/src/main.rs:22-121): child processes, rounds, MD5 check, markdown table."""
import json
import os
import subprocess
import sys

import pytest

import _data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "benchmark_tool.py")


def _run(tmp_path, extra, mode=None):
    src = tmp_path / "in.bin"
    src.write_bytes(_data.mixed(400_000, seed=3))
    out = tmp_path / "rows.json"
    env = dict(os.environ)
    if mode:
        env["ORZ_MODE"] = mode
    r = subprocess.run([sys.executable, TOOL, str(src), "--rounds", "1", "--json", str(out)] + extra, capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout, json.load(open(out))


def test_cpu_rows(oracle, tmp_path):
    """oracle rows (the CPU baseline) + whatever gzip/bzip2/xz the box has; sizes shrink with the level"""
    text, res = _run(tmp_path, ["--skip-orz"])
    rows = {r["name"]: r for r in res["rows"]}
    sizes = [rows["oracle -l%d (CPU restatement, 1 thread)" % lv]["size"] for lv in (0, 1, 2)]
    assert sizes[0] >= sizes[1] >= sizes[2] > 0
    assert "| name | compressed size |" in text
    body = [ln for ln in text.splitlines() if ln.startswith("| ") and "compressed size" not in ln]
    got = [int(ln.split("|")[2].replace(",", "")) for ln in body]
    assert got == sorted(got)  # sorted by compressed size like the reference's table


@pytest.mark.gpu
def test_gpu_rows_match_the_oracle_sizes(oracle, tmp_path):
    """`orz -lN` children (HIP encoder, host decoder) round-trip; in exact mode they produce the oracle's sizes, in the
    default fast mode sizes close to them (the tool itself checks the MD5 of every round trip)"""
    text, res = _run(tmp_path, ["--skip-others"], mode="exact")
    rows = {r["name"]: r for r in res["rows"]}
    for lv in (0, 1, 2):
        assert rows["**orz -l%d** (MI355X)" % lv]["size"] == rows["oracle -l%d (CPU restatement, 1 thread)" % lv]["size"]
    text, res = _run(tmp_path, ["--skip-others"])
    rows = {r["name"]: r for r in res["rows"]}
    for lv in (0, 1, 2):
        a, b = rows["**orz -l%d** (MI355X)" % lv]["size"], rows["oracle -l%d (CPU restatement, 1 thread)" % lv]["size"]
        assert abs(a - b) <= 0.03 * b
