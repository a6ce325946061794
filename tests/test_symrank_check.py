"""The arithmetic behind the unchecked groups of the symbol-ranking kernel (orz_amd/csrc/orz_symrank.h): a group of 32
items runs on the assumption that the quotient q = sum / 16 / count keeps its value, and the assumption is checked afterwards from
the group's 32 ranks -- count and sum after every item by a prefix sum, the one scaling by 9/10 (src/symrank.rs:63-66) at the item
that starts with count 390.  This restates the kernel's check lane by lane in numpy and holds it against the plain loop of
src/symrank.rs:61-74 on random rank sequences: the check passes exactly when the loop's quotient never moved, and then hands on
the loop's count and sum.  (CPU tier: the formula, not the kernel -- the kernel's ranks are held against the oracle in the GPU tier.)"""
import numpy as np

K_SYMS = 389
GROUP = 32


def loop(cnt, s, ranks):
    """src/symrank.rs:61-74 for one context: count, sum and the quotient used by every item"""
    qs = []
    for i in ranks:
        if cnt > K_SYMS:
            cnt = cnt * 9 // 10
            s = s * 9 // 10
        cnt += 1
        s += int(i)
        qs.append(s // 16 // cnt)
    return cnt, s, qs


def group_check(cnt, s, q, ranks):
    """the kernel's check: lane k = item k; returns (ok, count after the group, sum after the group)"""
    lane = np.arange(GROUP)
    ps = np.cumsum(ranks)                       # the DPP prefix sum
    r = K_SYMS + 1 - cnt                        # the item that starts with count 390 scales first (none: r >= 32)
    pr = int(ps[r - 1]) if 0 < r < GROUP else 0
    scaled = (s + pr) * 9 // 10
    after = lane >= r
    cnt_k = np.where(after, (K_SYMS + 1) * 9 // 10 + (lane - r) + 1, cnt + lane + 1)
    sum_k = np.where(after, scaled + (ps - pr), s + ps)
    lo = (q << 4) * cnt_k
    ok = (sum_k >= lo) & (sum_k - lo < (cnt_k << 4))
    return bool(ok.all()), int(cnt_k[-1]), int(sum_k[-1])


def test_the_groups_check_is_the_loops_quotient():
    rng = np.random.default_rng(6)
    passed = failed = 0
    for trial in range(4000):
        cnt = int(rng.integers(327, K_SYMS + 2))           # the steady state's counts, 390 (scale first) included
        q = int(rng.integers(0, 8))
        width = 16 * cnt
        # a sum that makes q the current quotient, near an edge of its interval half of the time
        if trial % 2:
            s = q * width + int(rng.integers(0, width))
        else:
            s = q * width + int(rng.choice([0, 1, 2, width - 3, width - 2, width - 1]))
        mean = int(rng.choice([1, 8, 16 * q + 8, 40, 120]))
        ranks = np.minimum(rng.poisson(mean, GROUP), K_SYMS - 1).astype(np.int64)
        c2, s2, qs = loop(cnt, s, ranks)
        ok, c3, s3 = group_check(cnt, s, q, ranks)
        assert ok == all(x == q for x in qs), (cnt, s, q, ranks.tolist(), qs)
        if ok:
            assert (c3, s3) == (c2, s2)
            passed += 1
        else:
            failed += 1
    assert passed > 500 and failed > 500  # (both outcomes are exercised)


def test_the_move_target_as_one_multiply_add():
    """i - i/16 - q == (15 i + 15 - 16 q) >> 4 (arithmetic shift), the unchecked item's form of src/symrank.rs:71-73"""
    i = np.arange(0, 512, dtype=np.int64)
    for q in range(0, 40):
        want = i - i // 16 - q
        got = (15 * i + 15 - 16 * q) >> 4
        assert (want == got).all()
        nxt = np.maximum(np.maximum(got, i // 2), 0)       # v_max3_i32 t, i >> 1, 0
        ref = np.maximum(i - np.minimum(i, i // 16 + q), i // 2)
        assert (nxt == ref).all()
        assert (((i + nxt) >> 1) == nxt + (i - nxt) // 2).all()   # ni1
