"""Host-side product code (builder, serializer, flattener) and the oracle under AddressSanitizer +
UndefinedBehaviorSanitizer (CPU build only; GPU ASan is not available on the pool)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_and_oracle_are_sanitizer_clean(tmp_path):
    exe = str(tmp_path / "san_main")
    csrc = os.path.join(ROOT, "index4j_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I" + csrc, "-I" + os.path.join(ROOT, "oracle"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "san_main.cpp")]
    cmd += [os.path.join(csrc, f) for f in ("fmx_build.cpp", "fmx_serial.cpp", "fmx_blob.cpp", "fmx_synth.cpp")]
    cmd += ["-x", "c", os.path.join(ROOT, "oracle", "index4j_oracle.c"), "-lpthread", "-o", exe]
    subprocess.check_call(cmd)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stdout + r.stderr
    assert r.stdout.count(" ok: ") == 4


def test_damaged_streams_and_images_never_leave_their_tables(tmp_path):
    """tests/cpp/fuzz_load.cpp: mutated index4j streams (bytes of the stream, fields of the model) and mutated images with
    a matching checksum go through the product's validators; whatever is accepted is queried with the DEVICE code compiled
    for the host under AddressSanitizer (count, locate, extract, extractUntilBoundary x 3 modes x 3 forms) with a watchdog —
    over the tree alone and, as a resident index is queried by default, over the window directory grown from that very image
    (entries of four and of six bytes: a directory made from a damaged tree holds arbitrary rows).
    An out-of-bounds read or an endless walk here would be a memory fault or a hung wave on the GPU."""
    csrc = os.path.join(ROOT, "index4j_amd", "csrc")
    # the expanded form (default images), and the same campaign over COMPACT images with the record-decoding device code
    # (sized for the CPU suite; longer campaigns with other seeds: run the two binaries by hand — docs/DESIGN_HISTORY.md 3)
    for name, defs, runs in (("fuzz_load", [], ((1, 1500),)), ("fuzz_load_compact", ["-DFMX_COMPACT=1"], ((3, 700),))):
        exe = str(tmp_path / name)
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer"] + defs + ["-I" + csrc,
               "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "fuzz_load.cpp")]
        cmd += [os.path.join(csrc, f) for f in ("fmx_build.cpp", "fmx_serial.cpp", "fmx_blob.cpp", "fmx_synth.cpp")]
        cmd += ["-lpthread", "-o", exe]
        subprocess.check_call(cmd)
        for seed, iters in runs:
            r = subprocess.run([exe, str(iters), str(seed)], capture_output=True, text=True, timeout=900)
            assert r.returncode == 0 and "ERROR" not in r.stderr and "HANG" not in r.stderr, r.stdout[-2000:] + r.stderr[-6000:]
            assert r.stdout.startswith("fuzz ok:"), r.stdout
