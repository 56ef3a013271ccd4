"""Test helper: run bench.main() with the package's library path pointed at the TEST-ONLY lane emulator,
so the multi-rank control flow of bench.py (env rendezvous, unique-id broadcast, barrier, max-over-ranks,
statistics reduction, single JSON line from rank 0) can be exercised under torch.distributed.run on CPU.
Never used by the product; bench.py itself always loads libmpcq.so."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_quad_ros_amd import _lib  # noqa: E402

_lib.DEFAULT_LIB = os.path.join(ROOT, "tests", "wave_emu", "libmpcq_emu.so")
import bench  # noqa: E402

if __name__ == "__main__":
    bench.main()
