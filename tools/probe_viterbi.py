#!/usr/bin/env python3
"""Timing probes of k_viterbi: builds variants of the product library with one part of the kernel removed (their results
are wrong; DABX_PROBE_FORCE_MERGE keeps the merge-driven control flow on its normal path) into tests/debug/probe_libs/.
Run on the GPU box:   for l in tests/debug/probe_libs/*.so; do DABX_LIBRARY=$PWD/$l python bench.py --no-cpu-baseline --no-pcie; done
and read kernel_ms_per_step.viterbi.  Results of round 2: profiles/r02_notes.md."""
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "abracadabra_amd", "csrc")
OUT = os.path.join(HERE, "..", "tests", "debug", "probe_libs")
PAT4 = r'"ds_swizzle_b32 %\[D\], %\[S\] offset:swizzle\(SWAP,16\)\\n\\t" \\\n\s*"s_waitcnt lgkmcnt\(0\)\\n\\t" \\\n\s*"v_max_i32 %\[pm\], %\[K\], %\[D\]'
PAT5 = r'"ds_bpermute_b32 %\[D\], %\[ad\], %\[S\]\\n\\t" \\\n\s*"s_waitcnt lgkmcnt\(0\)\\n\\t" \\\n\s*"v_max_i32 %\[pm\], %\[K\], %\[D\]'


def build(name, text, defs=""):
    inc = os.path.join(CSRC, "dabx_acs32.inc")
    open(inc, "w").write(text)
    subprocess.check_call(f"hipcc -O3 --offload-arch=gfx950 -DDABX_PROBE_FORCE_MERGE {defs} -ffp-contract=off -fPIC -shared -std=c++17 "
                          f"-Wno-unused-function -pthread -o {OUT}/{name} dabx_api.hip dabsdr_shim.cpp", shell=True, cwd=CSRC)


def main():
    os.makedirs(OUT, exist_ok=True)
    inc = os.path.join(CSRC, "dabx_acs32.inc")
    src = open(inc).read()
    shutil.copy(inc, inc + ".orig")
    try:
        build("lib0_reference.so", src)
        dpp = re.sub(PAT4, '"s_nop 0\\\\n\\\\t" \\\n    "v_max_i32_dpp %[pm], %[S], %[K] row_ror:4 row_mask:0xf bank_mask:0xf', src)
        dpp = re.sub(PAT5, '"s_nop 0\\\\n\\\\t" \\\n    "v_max_i32_dpp %[pm], %[S], %[K] row_ror:12 row_mask:0xf bank_mask:0xf', dpp)
        build("libA_no_lds_exchange.so", dpp)
        swp = re.sub(PAT4, '"v_permlane16_swap_b32 %[S], %[K]\\\\n\\\\t" \\\n    "v_max_i32 %[pm], %[K], %[S]', src)
        swp = re.sub(PAT5, '"v_permlane32_swap_b32 %[S], %[K]\\\\n\\\\t" \\\n    "v_max_i32 %[pm], %[K], %[S]', swp)
        build("libB_permlane_swap.so", swp)
        nom = re.sub(r'\s*"ds_read_b32 [^"]*" \\\n', '\n', src)
        nom = re.sub(r'\s*"s_waitcnt lgkmcnt\([0-5]\)\\n\\t" \\\n\s*"v_mfma[^"]*" \\\n', '\n', nom)
        nom = re.sub(r'\s*"s_nop 7\\n\\t" \\\n\s*"s_nop 3\\n\\t" \\\n', '\n', nom)
        build("libP1_no_mfma.so", nom)
        noge = re.sub(r'\s*"ds_write_b8 [^"]*" \\\n', '\n', src)
        noge = re.sub(r'"v_and_or_b32 %\[pm\], %\[pm\], %\[m128\], 63', '"s_nop 0', noge)
        assert noge != src
        build("libP2_no_group_end.so", noge)
        build("libP3_no_gather.so", src, "-DDABX_PROBE_NOGATHER")
        build("libP4_no_merge_no_traceback.so", src, "-DDABX_PROBE_NOTRACE")
        build("libP5_core_only.so", nom, "-DDABX_PROBE_NOGATHER -DDABX_PROBE_NOTRACE")
    finally:
        shutil.move(inc + ".orig", inc)


if __name__ == "__main__":
    main()
