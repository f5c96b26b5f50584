"""ObjectOutputStream framing of a serialized FmIndex (SER:67-79 writes, SER:89-100 reads), pinned on the Java Object
Serialization Stream Protocol — the one part of the byte format that can be pinned without a JVM.

A stream is: magic AC ED, version 00 05, then block-data records TC_BLOCKDATA 0x77 <u8 len> / TC_BLOCKDATALONG
0x7A <i32 len>, whose payloads form ONE byte sequence for DataInput: a primitive may straddle two records, records may be
empty, a TC_RESET 0x79 may stand between records, and a reader stops looking at the stream once FmIndex.read (FM:983-1025)
has what it asks for (java.io.ObjectInputStream.BlockDataInputStream).  These streams are built HERE by hand from the raw
payload — not by the library's writer — and fmx_load (the product's loader) and the oracle's reader must parse every one
of them to the same model: the same raw bytes on re-serialization, the same answers.

What stays unpinned (no JVM on any box): that the reference's writer emits exactly 1,024-byte records
(BlockDataOutputStream.MAX_BLOCK_SIZE) — the library's writer is checked against that published constant below — and the
HashMap key order of FM:956-960 (readers are order-agnostic)."""
import os
import random
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import orc  # noqa: E402

import index4j_amd as ia  # noqa: E402

MAGIC = b"\xac\xed\x00\x05"
TEXT = ("081109 203518 143 INFO dfs.DataNode$DataXceiver: Receiving block blk_-1608999687919862906 src: /10.250.19.102:54106\n"
        "081109 203518 35 INFO dfs.FSNamesystem: BLOCK* NameSystem.allocateBlock: /mnt/hadoop/mapred/system/job.jar\n" * 40)


def rec(payload, long_form=None):
    """one block-data record; long_form None = what a JVM writes (0x77 up to 255 bytes, else 0x7A)"""
    if long_form is None:
        long_form = len(payload) > 255
    if long_form:
        return b"\x7a" + struct.pack(">i", len(payload)) + payload
    assert len(payload) <= 255
    return b"\x77" + bytes([len(payload)]) + payload


def frame(raw, cuts, long_form=None, between=b""):
    """raw payload cut at the given offsets into records (`between` after every record)"""
    cuts = [0] + sorted(cuts) + [len(raw)]
    out = MAGIC
    for a, b in zip(cuts, cuts[1:]):
        out += rec(raw[a:b], long_form) + between
    return out


@pytest.fixture(scope="module")
def raw():
    o = orc.OracleFmIndex(TEXT, 4, True)
    r = o.write(False)
    assert len(r) > 4096
    return r


def both_parse_to(stream, raw):
    o = orc.OracleFmIndex.read(stream)
    assert o.write(False) == raw
    f = ia.FmIndex.read(stream, device=None)
    assert f.write(False) == raw
    assert f.getInputLength() == o.getInputLength() == len(TEXT) + 1
    f.close()


def test_jvm_shaped_stream_and_the_writers(raw):
    """records of exactly 1,024 bytes (BlockDataOutputStream.MAX_BLOCK_SIZE), the rest in a last record: what
    ObjectOutputStream emits for writeByte / writeInt / writeLong calls — and what both writers here emit"""
    jvm = frame(raw, list(range(1024, len(raw), 1024)))
    both_parse_to(jvm, raw)
    o = orc.OracleFmIndex.read(raw)
    assert o.write(True) == jvm
    f = ia.FmIndex.read(raw, device=None)
    assert f.write(True) == jvm
    f.close()


def test_a_record_boundary_at_every_offset_of_the_first_fields(raw):
    """version u8, sampleRate i32, flag u8, two i32 widths, length i32, the map's i32 size, its i32 / i16 entries: a record
    boundary at every byte offset of the first 96 bytes, i.e. inside every i16 / i32 there is; and at every offset of the
    first packed i64 word further in"""
    for cut in range(1, 96):
        both_parse_to(frame(raw, [cut]), raw)
    # the first long of the `suffixes` words: found by value — the raw form is big-endian DataOutput
    o = orc.OracleFmIndex.read(raw)
    n_keys = o.getAlphabetLength()
    head = 1 + 4 + 1 + 4 + 4 + 4 + 4 + 6 * n_keys
    for cut in range(head, head + 64):  # C counts, lookUp, the IntVector header and its first words
        both_parse_to(frame(raw, [cut, cut + 1, cut + 9]), raw)


def test_mixed_record_kinds_empty_records_resets_and_tiny_records(raw):
    rnd = random.Random(5)
    for trial in range(12):
        cuts = sorted(rnd.sample(range(1, len(raw)), rnd.randrange(1, 40)))
        parts = [0] + cuts + [len(raw)]
        out = MAGIC
        for a, b in zip(parts, parts[1:]):
            chunk = raw[a:b]
            while chunk:  # pieces of at most 255 bytes may take either form; longer ones must be long records
                take = chunk[: rnd.choice([1, 2, 3, 7, 255, 256, 1024, 4096])]
                chunk = chunk[len(take):]
                out += rec(take, long_form=True if len(take) > 255 else rnd.random() < 0.5)
                if rnd.random() < 0.3:
                    out += rec(b"", long_form=rnd.random() < 0.5)  # empty record
                if rnd.random() < 0.2:
                    out += b"\x79"  # TC_RESET between records
        both_parse_to(out, raw)
    both_parse_to(frame(raw, list(range(1, 300))), raw)  # 299 one-byte records first


def test_whatever_follows_the_index_is_not_looked_at(raw):
    """FmIndex.read stops after the wavelet tree; an ObjectInputStream never parses what is not asked for: further records,
    an object record (TC_OBJECT 0x73 ...), bytes that are no type code at all, a cut-off header"""
    jvm = frame(raw, list(range(1024, len(raw), 1024)))
    for tail in (rec(b"more data"), b"\x73\x72\x00\x03abc", b"\x00\x01\x02", b"\x7a\x00\x00", b"\x77", b"\x78", b"\x7a\xff\xff\xff\xff"):
        both_parse_to(jvm + tail, raw)


def test_streams_a_jvm_would_refuse(raw):
    """EOFException (the payload ends early), StreamCorruptedException (negative long length / a byte that is no type code
    where a header is needed), and the version check of SER:46-56 behind the framing"""
    cut = frame(raw[: len(raw) // 2], [700])
    for bad in (cut, cut + b"\x7a\xff\xff\xff\xf0", cut + b"\x05junk", frame(raw, [100])[:-5], MAGIC, MAGIC + b"\x77"):
        with pytest.raises(IOError):
            orc.OracleFmIndex.read(bad)
        with pytest.raises(Exception) as e:
            ia.FmIndex.read(bad, device=None)
        assert "fmx_load" in str(e.value)
    wrong_version = frame(b"\x07" + raw[1:], [1024])
    with pytest.raises(IOError, match="Incompatible serial versions"):
        orc.OracleFmIndex.read(wrong_version)
    with pytest.raises(Exception, match="Incompatible serial versions! Expected version 0 but was 7."):
        ia.FmIndex.read(wrong_version, device=None)
    # a declared record length beyond the buffer: the record holds what is there — too little for the index
    with pytest.raises(IOError):
        orc.OracleFmIndex.read(MAGIC + b"\x7a" + struct.pack(">i", len(raw) + 10) + raw[:-1])
    both_parse_to(MAGIC + b"\x7a" + struct.pack(">i", len(raw) + 10) + raw, raw)  # ... and enough when the bytes are there


def test_queries_on_an_index_loaded_from_a_hand_framed_stream(raw):
    rnd = random.Random(9)
    stream = frame(raw, sorted(rnd.sample(range(1, len(raw)), 25)), between=b"\x79")
    o = orc.OracleFmIndex.read(stream)
    assert o.count("INFO") == TEXT.count("INFO") and o.count("blk_") == 40
    n, locs = o.locate(ia.as_chars("NameSystem"), max_matches=100, cap=100)
    assert sorted(int(x) for x in locs[:n]) == [i for i in range(len(TEXT)) if TEXT.startswith("NameSystem", i)]
