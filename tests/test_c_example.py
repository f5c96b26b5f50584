"""examples/grep_log.c — the C ABI driven from plain C — compiles against include/fmx.h on CPU and, on the MI355X,
finds what a text scan finds."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "grep_log.c")
FIXTURE = os.path.join(ROOT, "tests", "golden", "HDFS_2k_multichar.log")


def build(tmp):
    exe = os.path.join(tmp, "grep_log")
    lib = os.path.join(ROOT, "index4j_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-L" + lib, "-lfmx",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_c_example_compiles_against_the_abi(tmp_path):
    assert os.path.exists(build(str(tmp_path)))


@pytest.mark.gpu
def test_c_example_greps_the_fixture(tmp_path):
    exe = build(str(tmp_path))
    out = subprocess.run([exe, FIXTURE, "WARN", "INFO", "no such thing"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    raw = open(FIXTURE, "rb").read()
    assert "'WARN': %d occurrence(s)" % raw.count(b"WARN") in out.stdout
    assert "'INFO': %d occurrence(s)" % raw.count(b"INFO") in out.stdout
    assert "'no such thing': 0 occurrence(s)" in out.stdout
    lines = [l for l in out.stdout.splitlines() if l.startswith("  @")]
    assert len(lines) == 6 and all(("WARN" in l or "INFO" in l) for l in lines)
