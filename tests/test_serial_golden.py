"""The serialized form (FmIndex.write, FM:948-975) pinned against DRIFT, and the HashMap key order of FM:956-960 replayed by a
third, independent restatement (CPU only).

tests/golden/ser/ was MINTED HERE, NOT BY A JVM (tools/make_golden_ser.py): the reference holds no golden serialized file and
this image has no JDK, so parity of the bytes with index4j stays unpinned.  These tests make sure the product's
writer and the oracle's keep producing the committed bytes, and that the writer never emits an order it does not model
without saying so (fmx_save_key_order_modelled)."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

import index4j_amd as ia
import orc
from common import GOLDEN, hdfs_text

SER = os.path.join(GOLDEN, "ser")
TEXTS = {"kat_fm": "This is a long string\0", "kat_wt": "aloha what a string this is string is eh"}


def _build(text, s):
    return ia.FmIndexBuilder().setSampleRate(s).setEnableExtraction(True).build(text, device=None)


@pytest.mark.parametrize("name", ["kat_fm", "kat_wt", "hdfs_fixture"])
@pytest.mark.parametrize("s", [1, 32])
def test_writers_reproduce_the_committed_streams(name, s):
    digests = json.load(open(os.path.join(SER, "digests.json")))["streams"]
    text = hdfs_text() if name == "hdfs_fixture" else TEXTS[name]
    fm, o = _build(text, s), orc.OracleFmIndex(text, s, True)
    for framed in (False, True):
        key = "%s_s%d_%s" % (name, s, "framed" if framed else "raw")
        b = fm.write(framed)
        assert b == o.write(framed), key
        assert len(b) == digests[key]["bytes"] and hashlib.sha256(b).hexdigest() == digests[key]["sha256"], key
        path = os.path.join(SER, key + ".ser")
        if os.path.exists(path):
            assert b == open(path, "rb").read(), key
        # ... and the committed stream reads back to an index that writes itself again
        again = ia.FmIndex.read(b, device=None)
        assert again.write(framed) == b
    assert fm.serialized_key_order_is_modelled() == digests[key]["key_order_modelled"] is True


def java_hashmap_key_order(keys):
    """java.util.HashMap<Integer, ?>.keySet() order after put(k) for k in keys (JDK 8+ putVal / resize / treeifyBin, restated
    here a third time, independently of fmx_serial.cpp and oracle/index4j_oracle.c).  -> (order, treeified)"""
    cap, table, size, treeified = 16, {}, 0, False

    def slot(k, c):
        h = k & 0xFFFFFFFF
        return (h ^ (h >> 16)) & (c - 1)

    def resize():
        nonlocal cap, table
        cap *= 2
        new = {}
        for b in sorted(table):
            for k in table[b]:
                new.setdefault(slot(k, cap), []).append(k)
        table = new

    for k in keys:
        chain = table.setdefault(slot(k, cap), [])
        before = len(chain)
        chain.append(k)
        if before >= 8:  # binCount >= TREEIFY_THRESHOLD - 1
            if cap < 64:
                resize()  # treeifyBin below MIN_TREEIFY_CAPACITY
            else:
                treeified = True
        size += 1
        if size > cap * 3 // 4:
            resize()
    return [k for b in sorted(table) for k in table[b]], treeified


def stream_keys(raw):
    """the character map's keys in the order a raw FmIndex.write stream holds them (FM:948-960)"""
    n = struct.unpack(">i", raw[18:22])[0]
    return [struct.unpack(">i", raw[22 + 6 * i: 26 + 6 * i])[0] for i in range(n)]


def _text_of(chars, rnd):
    body = list(chars) * 3
    rnd.shuffle(body)
    return np.array(list(chars) + body, dtype=np.uint16)  # first appearance = the order of `chars`


@pytest.mark.parametrize("case", ["ascii", "nine_in_one_slot_of_16", "tree_bin_at_256_slots", "cjk_block"])
def test_key_order_is_the_replayed_hashmap_order_and_tree_bins_are_reported(case):
    rnd = np.random.default_rng(3)
    if case == "ascii":
        chars = [ord(c) for c in "The quick brown fox jumps over the lazy dog 0123456789\n"]
        chars = list(dict.fromkeys(chars))
    elif case == "nine_in_one_slot_of_16":
        chars = [3 + 16 * k for k in range(1, 10)]  # 9 keys in slot 3 of 16: a JVM resizes to 32 slots at the ninth (no tree below 64)
    elif case == "tree_bin_at_256_slots":
        chars = list(range(0x41, 0x41 + 100)) + [0x105 + 0x100 * k for k in range(1, 10)]  # ... 9 keys = 5 mod 256 at 256 slots
    else:
        chars = [0x4E00 + 13 * k for k in range(1500)]  # 1,500 CJK ideographs (0x4E00 .. 0x9A1F)
    text = _text_of(chars, rnd)
    fm, o = _build(text, 4), orc.OracleFmIndex(text, 4, True)
    raw = fm.write(False)
    assert raw == o.write(False)
    # insertion order (FM:396-420): '\0' first, then first appearance
    want, tree = java_hashmap_key_order([0] + [c for c in chars if c != 0])
    assert stream_keys(raw) == want
    assert fm.serialized_key_order_is_modelled() == (not tree)
    assert tree == (case == "tree_bin_at_256_slots")
    # a reader does not depend on the order: the stream loads and answers
    again = ia.FmIndex.read(raw, device=None)
    assert again.getAlphabetLength() == fm.getAlphabetLength() and again.write(False) == raw
