"""bench.py's members annex, host half (CPU tier): every member's stream through the oracle's decoder (a process each) and the
one-oracle-process-per-host-core baseline (/root/reference/benchmark-tool/src/main.rs:57-114 times an encoder
process and verifies its output by decoding it)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import corpus  # noqa: E402


def _case(oracle):
    base = corpus.enwik_like(1_500_000)
    members = [base[i:i + 500_000] for i in range(0, len(base), 500_000)]
    return members, [oracle.encode(m, 1) for m in members]


def test_members_annex_checks_and_times(oracle):
    members, streams = _case(oracle)
    r = bench.members_check_and_cpu(members, streams, 1, max_procs=4)
    assert r["members"]["roundtrip_ok"] and r["members"]["members_not_decoding"] == []
    assert r["members"]["size_delta_pct"] == 0.0  # (the streams ARE the oracle's here)
    c = r["cpu_baseline_members"]
    assert c["kind"] == "port" and 1 <= c["cores"] <= 4 and c["value"] > 0 and c["unit"] == "MB/s"


def test_members_annex_names_the_member_that_does_not_decode(oracle):
    members, streams = _case(oracle)
    bad = bytearray(streams[2])
    bad[len(bad) // 2] ^= 0x10
    streams[2] = bytes(bad)
    r = bench.members_check_and_cpu(members, streams, 1, max_procs=2)
    assert not r["members"]["roundtrip_ok"] and r["members"]["members_not_decoding"] == [2]
