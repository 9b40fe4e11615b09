"""rocprofv3 counter CSVs of tools/fetch_calib (gpurun_out/calib_fetch, calib_write) -> counter value per kernel launch next to
the bytes the kernel is known to move.   python tools/summarize_calib.py gpurun_out/calib > profiles/r03/fetch_calibration.txt"""
import glob
import re
import sys

import pandas as pd

root = sys.argv[1]
known = [ln.split() for ln in open(f"{root}/calib_fetch.log") if ln.startswith("calib_")]
for ctr, sub in (("FETCH_SIZE", "calib_fetch"), ("WRITE_SIZE", "calib_write")):
    f = glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    d = pd.read_csv(f[0])
    d = d[(d.Counter_Name == ctr) & d.Kernel_Name.str.contains("calib_")].sort_values("Dispatch_Id")
    print(f"# {ctr} (KB as reported) per launch, in launch order")
    for (_, r), k in zip(d.iterrows(), known):
        name = re.sub(r"\(.*", "", r.Kernel_Name)
        nbytes = int(k[k.index("probe_bytes") + 1]) if "probe_bytes" in k else int(k[2])
        extra = f" table {k[2]} MiB" if "table_MiB" in k else ""
        print(f"{name:22s}{extra:16s} known bytes {nbytes:>12d}   {ctr} {r.Counter_Value * 1024:>14.0f} B   ratio counter/known {r.Counter_Value * 1024 / nbytes:6.3f}")
