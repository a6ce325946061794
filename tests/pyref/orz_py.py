"""orz_py.py -- a SECOND, independent restatement of the reference encoder, in plain Python (TEST INFRASTRUCTURE).

Why: the Rust reference cannot be built or run in this environment, so the C oracle (oracle/) is pinned only by
two hand-derived vectors, the reference's one unit test and round trips -- a misreading of the source that still
round-trips would go unnoticed.  This file was written separately from the oracle, straight from the Rust source,
module by module, keeping the reference's control flow (not the oracle's); tests/test_pyref.py requires the two
restatements to produce identical streams.  Agreement of two independent readings is the strongest pin available
here; it is not an output of the reference binary, and DESIGN.md keeps saying "parity unpinned".

Scope: `orz::encode`, any LZCfg, any number of blocks (window slide and LZEncoder::forward included).  Slow (pure
Python loops): meant for inputs of a few tens of kilobytes, or larger ones with few items (long runs).

Source map (all /root/reference/src):
  lib.rs:31-34,54-92   constants, window layout, chunk loop, EOF chunk       -> encode()
  ioutil.rs:79-88      write_len                                            -> write_len()
  lz.rs:89-346         LZEncoder::encode                                    -> Encoder.encode_chunk()
  lz.rs:482-534        hash1, hash2, reduced-offset id table                -> hash1(), hash2(), ROID
  matcher.rs:62-87     Bucket::{update, forward}                            -> Bucket.update(), Bucket.forward()
  matcher.rs:115-228   BucketMatcher::{update, forward, find_match, has_lazy_match} -> Matcher, find_match(), ...
  matcher.rs:256-263   hash_dword                                           -> hash_dword()
  mem.rs:41-70         mem_fast_common_prefix, mem_fast_equal               -> common_prefix(), fast_equal()
  symrank.rs:22-97     SymRankCoder                                         -> SymRank
  huffman.rs:27-141    HuffmanTable::new_from_sym_weights, HuffmanEncoding  -> huffman_lengths(), huffman_codes()
  coder.rs:27-89,159-217  Encoder / BitBuffer                               -> BitWriter
"""
import heapq

BLOCK = (1 << 25) - 1          # lib.rs:31
CHUNK_ITEMS = 1 << 20          # lib.rs:32
MAX_LEN = 240                  # lib.rs:33
MIN_LEN = 4                    # lib.rs:34
SENTINEL = MAX_LEN * 2         # lib.rs:54
PREMATCH = BLOCK // 2          # lib.rs:55
RING = 4094                    # lz.rs:24
ROIDS, LENIDS = 22, 6          # lz.rs:27-28
NSYMS = 256 + ROIDS * LENIDS + 1  # lz.rs:25
WORD = NSYMS - 1               # lz.rs:29
HASHSIZE = int(RING * 1.13) | 1   # matcher.rs:18
M32 = 0xFFFFFFFF


def _roid_table():  # lz.rs:494-514
    encs, base, cur = [], 0, 0
    while base < RING:
        bit_len = cur // 2
        rest = 0
        while rest != (1 << bit_len):
            if base < RING:
                encs.append((cur, bit_len, rest))
                base += 1
            rest += 1
        cur += 1
    return encs


ROID = _roid_table()


def _alnum(c):  # u8::is_ascii_alphanumeric
    return 48 <= c <= 57 or 65 <= c <= 90 or 97 <= c <= 122


class BitWriter:  # coder.rs:12-89,159-217
    def __init__(self):
        self.out = bytearray()
        self.value = 0
        self.len = 0

    def _reserve(self):  # reserve_32bits -> save_u32
        if self.len >= 32:
            top = (self.value >> (self.len - 32)) & M32
            self.len -= 32
            self.value &= (1 << self.len) - 1  # (the u64 of the reference drops these bits by shifting; a Python int must mask)
            self.out += top.to_bytes(4, "big")

    def put(self, n, v):
        self._reserve()
        self.value = (self.value << n) ^ v
        self.len += n

    def varint(self, v):  # encode_varint
        while True:
            has_next = v > 1
            self.put(2, (v & 1) | (int(has_next) << 1))
            v >>= 1
            if not has_next:
                break

    def table(self, lens):  # encode_huffman_table
        mx = max(lens)
        self.varint(mx)
        last = None
        for sym, ln in enumerate(lens):
            if ln > 0:
                self.varint(sym + 1 if last is None else sym - last)
                self.varint(mx - ln)
                last = sym
        self.varint(0)

    def finish(self):  # finish_into_output_pos
        self._reserve()
        if self.len > 0:
            pad = 32 - self.len
            self.value <<= pad
            self.len += pad
            self.out += (self.value & M32).to_bytes(4, "big")
            self.len = 0
        return bytes(self.out)


def huffman_lengths(weights, max_code_len=15):  # huffman.rs:27-111
    n = len(weights)
    w = list(weights)  # leaf weights; shrunk cumulatively on retries
    while True:
        heap = [(w[i], i) for i in range(n) if weights[i] > 0]
        heapq.heapify(heap)  # unique (weight, index) keys: pop order is what the reference's BinaryHeap gives
        if len(heap) <= 1:
            lens = [0] * n
            if heap:
                lens[heap[0][1]] = 1
            return lens
        nodes = [(x, 0, 0) for x in w]
        while len(heap) > 1:
            w1, i1 = heapq.heappop(heap)
            w2, i2 = heapq.heappop(heap)
            nodes.append((w1 + w2, i1, i2))
            heapq.heappush(heap, (w1 + w2, len(nodes) - 1))
        depth = [0] * len(nodes)
        for i in range(len(nodes) - 1, n - 1, -1):
            depth[nodes[i][1]] = depth[i] + 1
            depth[nodes[i][2]] = depth[i] + 1
        lens = depth[:n]
        cur = max(lens)
        if cur > max_code_len:
            shrink = 1 << (cur - max_code_len)
            w = [max(x // shrink, 1) if x > 0 else x for x in w]
            continue
        return lens


def huffman_codes(lens):  # huffman.rs:118-141
    enc = [(0, 0)] * len(lens)
    bits, cur = 0, 1
    for sym in sorted((i for i in range(len(lens)) if lens[i] > 0), key=lambda s: (lens[s], s)):
        shift = lens[sym] - cur
        if shift > 0:
            bits <<= shift
            cur += shift
        enc[sym] = (bits, lens[sym])
        bits += 1
    return enc


class SymRank:  # symrank.rs:13-97
    def __init__(self, order=None):
        self.value = [0] * NSYMS
        self.index = [0] * NSYMS
        self.cnt = 0
        self.sum = 1000000
        if order is not None:
            for i, v in enumerate(order):
                self.value[i] = v
                self.index[v] = i

    def clone(self):
        c = SymRank()
        c.value, c.index, c.cnt, c.sum = list(self.value), list(self.index), self.cnt, self.sum
        return c

    def encode(self, v, vun):
        i = self.index[v]
        iu = self.index[vun]
        self._update(v, i)
        if i == iu:
            return NSYMS - 1
        return i - (1 if i > iu else 0)

    def _update(self, v, i):
        if self.cnt > NSYMS:
            self.cnt = self.cnt * 9 // 10
            self.sum = self.sum * 9 // 10
        self.cnt += 1
        self.sum += i
        dec = (i // 16 + ((self.sum // 16 // self.cnt) & 0xFFFF)) & 0xFFFF  # `as u16`, u16 arithmetic
        nxt = max(i - dec if i > dec else 0, i // 2)                          # saturating_sub(..).max(i / 2)
        n = i - nxt
        if n == 0:
            return
        if n == 1:
            nv1 = self.value[nxt]
            self.index[v] = nxt
            self.value[i] = nv1
            self.index[nv1] = i
            self.value[nxt] = v
            return
        ni2, ni1 = nxt, nxt + n // 2
        nv1, nv2 = self.value[ni1], self.value[ni2]
        self.value[i] = nv1
        self.index[nv1] = i
        self.value[ni1] = nv2
        self.index[nv2] = ni1
        self.value[ni2] = v
        self.index[v] = ni2


class Bucket:  # matcher.rs:28-100
    __slots__ = ("pos", "len_min", "len_exp", "head")

    def __init__(self):
        self.pos = [0] * RING
        self.len_min = [0] * RING
        self.len_exp = [0] * RING
        self.head = 0

    def update(self, pos, reduced_offset, match_len):
        new_head = (self.head + 1) % RING
        if match_len >= MIN_LEN:
            ni = (self.head + RING - reduced_offset) % RING
            if self.len_min[ni] <= match_len:
                self.len_min[ni] = min(match_len + 1, 127)
        self.pos[new_head] = pos
        self.len_min[new_head] = 0
        self.len_exp[new_head] = match_len
        self.head = new_head

    def forward(self, forward_len):  # matcher.rs:82-87
        self.pos = [p - forward_len if p > forward_len else 0 for p in self.pos]


class Matcher:  # matcher.rs:102-228
    __slots__ = ("heads", "nexts")

    def __init__(self):
        self.heads = [-1] * HASHSIZE
        self.nexts = [-1] * RING

    def forward(self, bucket):  # matcher.rs:123-133: entries that point at an out-of-date node
        self.heads = [-1 if h != -1 and bucket.pos[h] == 0 else h for h in self.heads]
        self.nexts = [-1 if x != -1 and bucket.pos[x] == 0 else x for x in self.nexts]


class Window:
    """sbvec_buf of lib.rs:67-69: index i of the reference's sbvec is self.b[SENTINEL + i].  One allocation for the
    whole stream: what a block leaves behind stays there (bytes past a short final block are the previous block's)."""

    def __init__(self):
        self.b = bytearray(BLOCK + SENTINEL * 2)
        self.n = PREMATCH

    def load(self, data):  # read_repeatedly into sbvec[SBVEC_PREMATCH_LEN..]
        self.b[SENTINEL + PREMATCH:SENTINEL + PREMATCH + len(data)] = data
        self.n = PREMATCH + len(data)  # sbuf.len() as LZEncoder::encode sees it (lib.rs:77)

    def slide(self):  # sbvec.copy_within(sbvec.len() - SBVEC_PREMATCH_LEN.., 0) on the FULL block slice (lib.rs:83)
        self.b[SENTINEL:SENTINEL + PREMATCH] = self.b[SENTINEL + BLOCK - PREMATCH:SENTINEL + BLOCK]

    def u8(self, i):
        return self.b[SENTINEL + i]

    def u32(self, i):  # little-endian dword, as ptr.get::<u32>
        return int.from_bytes(self.b[SENTINEL + i:SENTINEL + i + 4], "little")


def hash1(w, pos):  # lz.rs:482-486
    return (w.u8(pos) & 0x7F) | (int(_alnum(w.u8(pos - 1))) << 7)


def hash2(w, pos):  # lz.rs:489-492
    return (w.u8(pos) & 0x7F) | (hash1(w, pos - 1) << 7)


def hash_dword(w, pos):  # matcher.rs:256-263: lane-wise (byte * MUL) ^ ADD, wrapping u32, summed wrapping
    muls = (131313131, 1313131, 13131, 131)
    adds = (797, 79797, 7979797, 797979797)
    h = 0
    for k in range(4):
        h = (h + ((((w.u8(pos + k) * muls[k]) & M32) ^ adds[k]))) & M32
    return h


def common_prefix(w, p1, p2, max_len):  # mem.rs:41-51
    for l in range(0, max_len, 16):
        a = w.b[SENTINEL + p1 + l:SENTINEL + p1 + l + 16]
        c = w.b[SENTINEL + p2 + l:SENTINEL + p2 + l + 16]
        if a != c:
            k = 0
            while a[k] == c[k]:
                k += 1
            return l + k
    return max_len


def fast_equal(w, p1, p2, length, p2_last_dword):  # mem.rs:55-70
    if p2_last_dword != w.u32(p1 + length - 4):
        return False
    for l in reversed(range(0, length - 4, 4)):
        if w.u32(p1 + l) != w.u32(p2 + l):
            return False
    return True


def find_match(m, bucket, w, pos, depth):  # matcher.rs:135-192; returns (reduced_offset, len, len_expected, len_min)
    node = m.heads[hash_dword(w, pos) % HASHSIZE]
    if node == -1:
        return (0, 0, 0, 0)
    max_len = MIN_LEN - 1
    best_min, best_exp, best_node = MIN_LEN, MIN_LEN, 0
    node_pos = bucket.pos[node]
    max_len_dword = w.u32(pos + max_len - 3)
    for _ in range(depth):
        if w.u32(node_pos + max_len - 3) == max_len_dword:
            lcp = common_prefix(w, node_pos, pos, MAX_LEN)
            if lcp > max_len:
                best_min = bucket.len_min[node]
                best_exp = bucket.len_exp[node]
                max_len = lcp
                best_node = node
                max_len_dword = w.u32(pos + max_len - 3)
            if lcp == MAX_LEN:
                break
            if best_exp > 0 and lcp > best_exp:
                break
        node = m.nexts[node]
        if node == -1:
            break
        nxt = bucket.pos[node]
        if node_pos <= nxt:
            break
        node_pos = nxt
    if max_len >= MIN_LEN and pos + max_len < w.n:
        return ((bucket.head + RING - best_node) % RING, max_len, max(best_exp, MIN_LEN), max(best_min, MIN_LEN))
    return (0, 0, 0, 0)


def has_lazy_match(m, bucket, w, pos, min_len, depth):  # matcher.rs:194-228
    last = w.u32(pos + min_len - 4)
    node = m.heads[hash_dword(w, pos) % HASHSIZE]
    if node == -1:
        return False
    node_pos = bucket.pos[node]
    for _ in range(depth):
        if fast_equal(w, node_pos, pos, min_len, last):
            return True
        node = m.nexts[node]
        if node == -1:
            break
        nxt = bucket.pos[node]
        if node_pos <= nxt:
            break
        node_pos = nxt
    return False


class Encoder:  # LZEncoder + LZContext, lz.rs:49-80
    def __init__(self):
        self.buckets = [Bucket() for _ in range(256)]
        self.matchers = [Matcher() for _ in range(256)]
        self.symranks = [SymRank() for _ in range(512)]
        self.words = [(0, 0)] * 32768
        self.first_block = True
        self.after_literal = True

    def forward(self, forward_len):  # lz.rs:82-87
        for i in range(256):
            self.buckets[i].forward(forward_len)
            self.matchers[i].forward(self.buckets[i])

    def _insert(self, w, ctx, spos, reduced_offset, match_len):
        b, m = self.buckets[ctx], self.matchers[ctx]
        b.update(spos, reduced_offset, match_len)      # lz.rs:191-195 / 209
        entry = hash_dword(w, spos) % HASHSIZE          # matcher.rs:115-121
        m.nexts[b.head] = m.heads[entry]
        m.heads[entry] = b.head

    def encode_chunk(self, cfg, w, spos):  # lz.rs:89-346; returns (new spos, chunk bytes)
        depth, lazy1, lazy2 = cfg
        items = []  # (is_match, symbol, ctx, unlikely, robitlen, robits, enc_len, after_literal)
        while spos < w.n and len(items) < CHUNK_ITEMS:
            expected = self.words[hash2(w, spos - 1)]
            word_matched = (w.u8(spos), w.u8(spos + 1)) == expected
            c1 = hash1(w, spos - 1)
            ctx = c1 | (int(self.after_literal) << 8)
            unlikely = expected[0]
            lazy_id = 0
            ro, mlen, mexp, mmin = find_match(self.matchers[c1], self.buckets[c1], w, spos, depth)
            if mlen > 0:
                roid, robitlen, robits = ROID[ro]
                if mlen < MAX_LEN // 2:
                    l1 = mlen + 1 + int(robitlen < 8)
                    l2 = l1 - int(word_matched)
                    h = hash1(w, spos)
                    if has_lazy_match(self.matchers[h], self.buckets[h], w, spos + 1, l1, lazy1):
                        lazy_id = 1
                    else:
                        h = hash1(w, spos + 1)
                        if has_lazy_match(self.matchers[h], self.buckets[h], w, spos + 2, l2, lazy2):
                            lazy_id = 2
                if lazy_id == 0:
                    enc = mlen - mmin if mlen > mexp else (mlen - mmin + 1 if mlen < mexp else 0)
                    enc &= 0xFF
                    lenid = min(LENIDS - 1, enc)
                    items.append((True, 256 + roid * LENIDS + lenid, ctx, unlikely, robitlen, robits, enc, self.after_literal))
                    self._insert(w, c1, spos, ro, mlen)
                    spos += mlen
                    self.after_literal = False
                    self.words[hash2(w, spos - 3)] = (w.u8(spos - 2), w.u8(spos - 1))
                    continue
            self._insert(w, c1, spos, 0, 0)
            if spos + 1 < w.n and lazy_id != 1 and word_matched:
                items.append((False, WORD, ctx, unlikely, 0, 0, 0, self.after_literal))
                spos += 2
                self.after_literal = False
            else:
                items.append((False, w.u8(spos), ctx, unlikely, 0, 0, 0, self.after_literal))
                spos += 1
                self.after_literal = True
                self.words[hash2(w, spos - 3)] = (w.u8(spos - 2), w.u8(spos - 1))

        bw = BitWriter()
        if self.first_block:  # lz.rs:238-265
            counts = [0] * NSYMS
            for it in items:
                counts[it[1]] += 1
            counted = sum(1 for c in counts if c > 1)
            order = sorted(range(NSYMS), key=lambda i: -max(counts[i], 1))  # stable, like sort_by_key(Reverse(..))
            bw.varint(counted)
            for s in order[:counted]:
                bw.put(9, s)
            first = SymRank(order)
            self.symranks = [first.clone() for _ in range(512)]
            self.first_block = False
        bw.varint(min(spos, w.n))
        bw.varint(len(items))
        w1 = [[0] * NSYMS, [0] * NSYMS]
        w2 = [0] * MAX_LEN
        ranked = []
        for (is_match, sym, ctx, unlikely, robitlen, robits, enc, al) in items:  # lz.rs:274-305
            r = self.symranks[ctx].encode(sym, unlikely)
            w1[int(al)][r] += 1
            if is_match and enc >= LENIDS - 1:
                w2[enc] += 1
            ranked.append(r)
        lens = [huffman_lengths(w1[0]), huffman_lengths(w1[1]), huffman_lengths(w2)]
        for ln in lens:
            bw.table(ln)
        codes = [huffman_codes(ln) for ln in lens]
        for r, (is_match, sym, ctx, unlikely, robitlen, robits, enc, al) in zip(ranked, items):  # lz.rs:320-342
            code, ln = codes[int(al)][r]
            bw.put(ln, code)
            if is_match:
                bw.put(robitlen, robits)
                if enc >= LENIDS - 1:
                    code, ln = codes[2][enc]
                    bw.put(ln, code)
        return spos, bw.finish()


def write_len(n):  # ioutil.rs:79-88
    out = bytearray()
    while n >= 128:
        out.append(128 + n % 128)
        n //= 128
    out.append(n)
    return bytes(out)


def encode(data, cfg):
    """orz::encode (lib.rs:58-92); cfg = (match_depth, lazy_depth1, lazy_depth2)"""
    data = bytes(data)
    out = bytearray()
    enc = Encoder()
    w = Window()
    at = 0
    while at < len(data):
        piece = data[at:at + BLOCK - PREMATCH]
        at += len(piece)
        w.load(piece)
        spos = PREMATCH
        while spos < PREMATCH + len(piece):
            spos, chunk = enc.encode_chunk(cfg, w, spos)
            out += write_len(len(chunk))
            out += chunk
        w.slide()
        enc.forward(BLOCK - PREMATCH)
    out += write_len(0)
    return bytes(out)
