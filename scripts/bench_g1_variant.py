"""bench_g1.py against another build of the library: python scripts/bench_g1_variant.py <lib file name> [N steps motion]"""
import runpy
import sys
sys.path.insert(0, ".")
import deepmimic_mujoco_amd._lib as _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libdeepmimic_hip.so", sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
runpy.run_path("scripts/bench_g1.py", run_name="__main__")
