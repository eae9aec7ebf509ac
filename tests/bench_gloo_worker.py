"""One rank of `bench.py --gpus N` with the CPU implementation of the ABI injected (gloo, no GPU):
exercises bench.py's multi-process control flow — sharded generation, the distributed plan,
barriers, max-over-ranks timing, the single JSON line on rank 0 — exactly as torchrun drives it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(rank, world, port, sf):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    from sdqlpy_amd import abi, engine
    eng = engine.Engine(abi.Library(os.path.join(ROOT, "oracle", "libsdqloracle.so")).context(threads=2))
    bench.main(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--sf", str(sf), "--no-cpu-baseline"],
               hooks={"backend": "gloo", "device": "cpu", "engine": eng})


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]))
