"""Host-side parsers (FIG database, PAD, packet mode, raw-file probe, TII detector) under AddressSanitizer and
UndefinedBehaviorSanitizer with random and mutated input: broadcast data is untrusted.  CPU only."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_host_parsers_survive_random_input_under_sanitizers(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "fuzz_host")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I" + os.path.join(HERE, "..", "abracadabra_amd", "csrc"), "-o", exe, os.path.join(HERE, "fuzz", "fuzz_host.cpp")]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "asan" in build.stderr.lower():
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "fuzz ok" in run.stdout, (run.stdout[-500:], run.stderr[-3000:])
