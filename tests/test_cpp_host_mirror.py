"""Builds tests/cpp/test_host_mirror.cpp against include/index4j/FmIndex.hpp + libfmx.so and runs it:
host mode on CPU (builder / serializer / error contract), gpu mode on the MI355X (queries)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")


def build():
    src = os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp")
    deps = [src, os.path.join(ROOT, "include", "index4j", "FmIndex.hpp"), os.path.join(ROOT, "include", "fmx.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        libdir = os.path.join(ROOT, "index4j_amd")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-o", EXE, src, "-L" + libdir, "-lfmx",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])


def run(mode):
    build()
    r = subprocess.run([EXE, mode, os.path.join(ROOT, "tests", "golden", "HDFS_2k_multichar.log")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


def test_cpp_host_mirror_host_side():
    run("host")


@pytest.mark.gpu
def test_cpp_host_mirror_queries_on_gpu():
    run("gpu")
