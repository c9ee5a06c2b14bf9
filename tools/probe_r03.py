#!/usr/bin/env python3
"""Round-3 timing probes (results may be wrong; timing only): builds variants of the product library into tests/debug/probe_libs/.
Run on the GPU box: for l in tests/debug/probe_libs/*.so; do DABX_LIBRARY=$PWD/$l python bench.py --no-cpu-baseline --no-pcie --no-legacy; done"""
import os, re, shutil, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "abracadabra_amd", "csrc")
OUT = os.path.join(HERE, "..", "tests", "debug", "probe_libs")


def build(name, text, defs=""):
    inc = os.path.join(CSRC, "dabx_acs32.inc")
    open(inc, "w").write(text)
    subprocess.check_call(f"hipcc -O3 --offload-arch=gfx950 {defs} -ffp-contract=off -fPIC -shared -std=c++17 "
                          f"-Wno-unused-function -pthread -o {OUT}/{name} dabx_api.hip dabsdr_shim.cpp", shell=True, cwd=CSRC)


def main():
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT, exist_ok=True)
    inc = os.path.join(CSRC, "dabx_acs32.inc")
    src = open(inc).read()
    shutil.copy(inc, inc + ".orig")
    try:
        build("lib0_reference.so", src, "-DDABX_PROBE_FORCE_MERGE")
        nonop = src.replace('    "s_nop 0\\n\\t" \\\n', '')
        assert nonop != src
        build("libV1_no_dpp_nop.so", nonop, "-DDABX_PROBE_FORCE_MERGE")
        vor = src.replace('"v_and_or_b32 %[pm], %[pm], %[m128], 63', '"v_or_b32 %[pm], %[m128], %[pm]')
        assert vor != src
        build("libV2_group_end_v_or.so", vor, "-DDABX_PROBE_FORCE_MERGE")
        nomw = re.sub(r'\s*"s_nop 7\\n\\t" \\\n\s*"s_nop 3\\n\\t" \\\n', '\n', src)
        assert nomw != src
        build("libV3_no_mfma_wait.so", nomw, "-DDABX_PROBE_FORCE_MERGE")
        for n in (8, 17, 35):
            build(f"libD{n}_demod_stagger.so", src, f"-DDABX_PROBE_FORCE_MERGE -DDABX_PROBE_STAGGER={n}")
    finally:
        shutil.move(inc + ".orig", inc)


if __name__ == "__main__":
    main()
