"""One rank started by bench.py's own launcher (`python bench.py --gpus N` with SDQLPY_AMD_BENCH_CHILD naming this file): RANK /
WORLD_SIZE / MASTER_* come from the launcher's environment, the arguments are bench.py's own; gloo and the CPU implementation of the
ABI are injected through bench.main's hooks, as in bench_gloo_worker.py.  SDQLPY_TEST_FAIL_RANK=r makes rank r die at once (the
launcher must stop the others and report failure)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    if os.environ.get("SDQLPY_TEST_FAIL_RANK") == os.environ.get("RANK"):
        sys.exit(7)
    import bench
    from sdqlpy_amd import abi, engine
    eng = engine.Engine(abi.Library(os.path.join(ROOT, "oracle", "libsdqloracle.so")).context(threads=2))
    bench.main(sys.argv[1:], hooks={"backend": "gloo", "device": "cpu", "engine": eng})
