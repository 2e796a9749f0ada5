#!/usr/bin/env python
"""Build the WORKING TREE's kernels into a variant library next to the product one, for an A/B measurement on the GPU box.

    python profiles/ab_variant.py NAME [-DFLAG ...]      ->  pytorch3d_amd/libp3d_NAME.so   (git-ignored; travels with gpurun)

The product library pytorch3d_amd/libp3d_amd.so is not touched.  Typical round trip (what profiles/r03/call8.sh, call11.sh and
call12.sh did by hand):

    1. edit csrc/, `python profiles/ab_variant.py idea`, `git stash` (the product sources stay what the product library is)
    2. gpurun -- 'bash profiles/ab_call.sh idea'      exp_measure.py: product vs variant in ONE process, bit parity of the
                                                      forward, gradient deviation; then the parity suites ON the variant
    3. adopt (git stash pop, rebuild the product, re-run the suites) or drop; the record goes under profiles/rNN/exp_<idea>/
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) < 2 or sys.argv[1].startswith("-"):
        raise SystemExit(__doc__)
    name, flags = sys.argv[1], sys.argv[2:]
    lib = os.path.join(ROOT, "pytorch3d_amd", f"libp3d_{name}.so")
    env = dict(os.environ, P3D_LIB_PATH=lib, P3D_EXTRA_FLAGS=" ".join(flags))
    subprocess.check_call([sys.executable, "-m", "pytorch3d_amd.build", "--force"], cwd=ROOT, env=env, stdout=subprocess.DEVNULL)
    print(lib)


if __name__ == "__main__":
    main()
