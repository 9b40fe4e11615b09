"""Section timers of lookup_span_kernel (debug build: -DOVTK_SPAN_TIMERS, see tools/span_sections.sh): shader-clock ticks per section of
the kernel, summed over all waves and all launches of a `bench.py --sync` run on config 2.  Waves share a SIMD, so a section's ticks
include the time its wave waited for the others: the SHARES are what to read, not the absolute figures."""
import ctypes as C
import io
import sys
from contextlib import redirect_stdout

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from openvino_tokenizers_amd import _lib as L  # noqa: E402

NAMES = ["rows' headers, staging reservation", "block set-up: text to LDS, row-start flags", "span_flags", "piece list, first pieces of the rows",
         "sentinel, next block's load", "lookup rounds", "records, miss flush"]


def main():
    steps = 200
    sys.argv = ["bench.py", "--sync", "--steps", str(steps), "--warmup", "10", "--no-cpu-baseline", "--no-extras"] + sys.argv[1:]
    out = io.StringIO()
    with redirect_stdout(out):
        bench.main()
    lib = L.load()
    fn = lib.ovtk_debug_span_timers
    fn.argtypes, fn.restype = [C.POINTER(C.c_ulonglong), C.c_int], C.c_int
    buf = (C.c_ulonglong * 16)()
    assert fn(buf, 0) == 0
    t = np.array(list(buf)[:7], dtype=np.float64)
    for name, v in zip(NAMES, t):
        print(f"{v / t.sum() * 100:6.2f} %  {name}")


if __name__ == "__main__":
    main()
