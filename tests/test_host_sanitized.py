"""The host side of the product under the sanitizers gcc has (CPU only).

`make -C pollen_amd/csrc host_check asan tsan` builds tests/host_check/host_check.cpp together with
flatgfa_core.cpp + synth.cpp three ways: plain, AddressSanitizer + UBSan (-fno-sanitize-recover), and
ThreadSanitizer.  The driver parses every golden fixture (both parser modes), hundreds of mutated
texts and damaged .flatgfa images (from an odd address: the reference's pools are align-1,
file.rs:163-167), round-trips them through the printer and both containers, and runs the threaded
step-list parse and the table formatter's threads.  All three builds must finish clean and print
the same digests."""
import glob
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT

CSRC = os.path.join(ROOT, "pollen_amd", "csrc")
BUILD = os.path.join(ROOT, "pollen_amd", "build")


@pytest.fixture(scope="module")
def binaries():
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no g++ / make")
    subprocess.run(["make", "-C", CSRC, "host_check", "asan", "tsan"], check=True, capture_output=True, timeout=600)
    return {k: os.path.join(BUILD, n) for k, n in (("plain", "host_check"), ("asan", "host_check_asan"), ("tsan", "host_check_tsan"))}


def run(exe):
    fixtures = sorted(glob.glob(os.path.join(GOLDEN, "*.gfa")))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
    return subprocess.run([exe] + fixtures, capture_output=True, text=True, timeout=600, env=env)


def test_sanitized_builds_are_clean_and_agree(binaries):
    plain = run(binaries["plain"])
    assert plain.returncode == 0, plain.stderr
    lines = plain.stdout.strip().splitlines()
    assert [ln.split()[0] for ln in lines] == ["fixtures", "mutated_texts", "damaged_images", "bed_and_floats", "threads", "all"]
    assert "accepted=" in lines[1] and "refused=" in lines[2]
    for kind in ("asan", "tsan"):
        r = run(binaries[kind])
        assert r.returncode == 0, f"{kind}: {r.stderr[-3000:]}"
        assert "runtime error" not in r.stderr and "Sanitizer" not in r.stderr, f"{kind}: {r.stderr[-3000:]}"
        assert r.stdout == plain.stdout, kind
