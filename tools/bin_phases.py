"""tools/bin_phases.py: per-phase cycles of the binning kernel (k_binj on the default path) on bench.py's default workload.
Needs a tuning build made with tools/build_variant.sh NAME -DCSM_BIN_TIMING and
CSM_HIP_LIB pointing at it."""
import ctypes
import os
import subprocess
import sys

sys.path[:0] = [".", "my-lidar-graph-slam-v2_amd"]
os.environ.setdefault("CSM_BENCH_SCANS", "256")
import torch  # noqa: E402,F401  (first: its bundled HIP runtime must be the one the library binds)
import bench  # noqa: E402
from csm_hip import _lib  # noqa: E402

lib = _lib.load()
fn = lib.csm_debug_bin_cycles
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 16)()
sys.argv = ["bench.py", "--no-configs", "--no-cpu-baseline", "--steps", "3", "--warmup", "1"]
bench.main()
rc = fn(buf)
wgs = buf[15]
names = ["init", "pass A (hash insert)", "pass B (count)", "scan", "records + cursors", "pass C (emit)"]
tot = sum(buf[k] for k in range(6)) + sum(buf[k] for k in range(8, 12))
print("rc", rc, "workgroups", wgs, file=sys.stderr)
for k, nm in enumerate(names):
    print("%-24s %9.0f cycles per workgroup  %5.1f %%" % (nm, buf[k] / max(wgs, 1), 100.0 * buf[k] / max(tot, 1)),
          file=sys.stderr)
print("total %.0f cycles per workgroup" % (tot / max(wgs, 1)), file=sys.stderr)
# k_binj accumulates 8..10 (the ticks 8..10 also advance the phase clock: pass A's own counter is the remainder)
for k, nm in enumerate(["A: key + ballots", "A: hash insert", "A: list append", "A: band test"]):
    print("  %-22s %9.0f cycles per workgroup" % (nm, buf[8 + k] / max(wgs, 1)), file=sys.stderr)
