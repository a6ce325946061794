"""Summarise an ORZ_TIMELINE dump (stderr of an encode run with ORZ_TIMELINE=<sweep>): per-wave wall-clock
stamps of one sweep of the parse kernel -> percentiles of each phase's end, re-walks, polls, per-64-block view.

  ORZ_TIMELINE=300 python ... 2> dump.txt ; python tools/timeline_summary.py dump.txt
"""
import sys

import numpy as np


def main(path):
    rows = [list(map(int, ln.split()[1:])) for ln in open(path) if ln.startswith("TL ")]
    if not rows:
        print("no TL lines in", path)
        return
    a = np.array(rows)
    a = a[np.argsort(a[:, 0])]
    blk, seg, st, p1, w1, we, en, pas, pol, ch = a.T
    print("waves %d, launch span %.1f us (first start to last end), all started within %.1f us" % (len(a), en.max() / 100, st.max() / 100))
    print("%-28s %7s %7s %7s %7s %7s" % ("microseconds after launch", "p10", "p50", "p90", "p99", "max"))
    for name, v in (("candidates collected", p1), ("first item walk done", w1), ("hand-off settled / given up", we), ("published, wave ends", en)):
        q = np.percentile(v, [10, 50, 90, 99, 100]) / 100
        print("%-28s %7.1f %7.1f %7.1f %7.1f %7.1f" % (name, *q))
    print("walks per wave (1 = no re-walk):", dict(zip(*np.unique(pas + 1, return_counts=True))))
    print("polls per wave: mean %.1f max %d; waves whose result changed: %d (first at block %s)" % (
        pol.mean(), pol.max(), int(ch.sum()), int(blk[ch > 0].min()) if ch.sum() else "-"))
    print("per 256 blocks from the front: last first-walk / median settle / last settle (us), re-walks, changed")
    for lo in range(0, len(a), 256):
        s = slice(lo, lo + 256)
        print("  blocks %4d-%4d: %6.1f %6.1f %6.1f   %3d %3d" % (lo, lo + len(a[s]) - 1, w1[s].max() / 100, np.median(we[s]) / 100,
                                                              we[s].max() / 100, int((pas[s] > 0).sum()), int(ch[s].sum())))


if __name__ == "__main__":
    main(sys.argv[1])
