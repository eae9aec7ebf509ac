#!/usr/bin/env python3
"""Extra rounds of the differential fuzz (tests/helpers.fuzz_case) and of the ABI-level cases that
exercise the index / key-set / shared-group kernels, with seeds the suite does not use.
python tools/fuzz_more.py [first_seed] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from sdqlpy_amd import abi, build

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
hip = abi.Library(build.HIP_LIB).context(device=0)
cpu = abi.Library(os.path.join(ROOT, "oracle", "libsdqloracle.so")).context(threads=8)
rounds = 0
for seed in range(first, first + count):
    rounds += helpers.fuzz_case(hip, cpu, seed)
    if seed % 4 == 0:
        helpers.key_set_case(hip, n=1000 + 977 * (seed % 13), seed=seed)
        helpers.share_groups_case(hip, n_build=500 + 811 * (seed % 11), n_probe=3000 + 1013 * (seed % 7), seed=seed)
        helpers.groupby_key_case(hip, n=900 + 1999 * (seed % 9), seed=seed)
print("fuzz rounds without a difference:", rounds, "seeds", first, "..", first + count - 1)
