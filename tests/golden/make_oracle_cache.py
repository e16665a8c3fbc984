"""Regenerates tests/golden/oracle/: the CPU oracle's answers for the LARGE cases of the GPU suite (tests/helpers.py,
"the oracle's answers for the LARGE cases").  The listed tests are run with JTK_WRITE_GOLDEN=1 and JTK_DEVICE_IS_ORACLE=1: the
oracle computes every answer once, stores it under the sha256 of what it was given, and also stands in for the device, so no GPU is
needed (the assertions compare the oracle with itself; the point of the run is its side effect).  ~15 min on 8 CPUs.

    python3 tests/golden/make_oracle_cache.py
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

# test ids whose oracle side is large (seconds x chunks x reads); everything else keeps the oracle in the loop
LARGE = [
    "tests/test_gpu_shapes.py::test_pileup_beyond_511_reads_matches_oracle",
    "tests/test_gpu_shapes.py::test_full_path_pileup_of_1100_reads_matches_oracle",
    "tests/test_gpu_shapes.py::test_random_shape_sweep_matches_oracle",
    "tests/test_gpu_shapes.py::test_random_chain_sweep_matches_oracle",
    "tests/test_gpu_shapes.py::test_cfg4_full_shape_matches_oracle",
    "tests/test_gpu_parity.py::test_pileups_beyond_1023_reads_match_oracle",
    "tests/test_gpu_parity.py::test_large_pileups_match_oracle",
    "tests/test_gpu_parity.py::test_recursive_split_matches_oracle",
    "tests/test_gpu_defining_shapes.py::test_copy_numbers_3_5_6_7_at_40_reads_per_copy_match_the_oracle",
    "tests/test_gpu_defining_shapes.py::test_256_chunks_of_the_headline_workload_match_the_oracle",
    "tests/test_gpu_defining_shapes.py::test_16_chunks_of_cfg4_match_the_oracle",
    "tests/test_gpu_defining_shapes.py::test_parity_campaign_seed_31_four_copy_slice",
]


def main():
    shutil.rmtree(os.path.join(HERE, "oracle"), ignore_errors=True)
    env = dict(os.environ, JTK_WRITE_GOLDEN="1", JTK_DEVICE_IS_ORACLE="1")
    rc = subprocess.call([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x"] + LARGE, cwd=ROOT, env=env)
    idx = os.path.join(HERE, "oracle", "INDEX.txt")   # (fixture -> the test that asks for it; every answer is computed twice)
    lines = sorted(set(open(idx).read().splitlines()))
    open(idx, "w").write("\n".join(lines) + "\n")
    n = len([f for f in os.listdir(os.path.join(HERE, "oracle")) if f.endswith(".npz")])
    print("wrote", n, "fixtures; pytest rc", rc)
    return rc


if __name__ == "__main__":
    sys.exit(main())
