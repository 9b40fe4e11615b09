"""Turns the raw rocprofv3 output of a gpurun call (gpurun_out/prof_c*, pmc_fetch_c*, pmc_write_c*) into the small
summaries committed under profiles/ and into profiles/latest_pmc.json (read by bench.py for roofline.traffic).

    python tools/summarize_profiles.py r01/c            # writes profiles/r01/c_config{2,3,5}_*.csv + latest_pmc.json

HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes): MI355X_MICROARCH.md "HBM" -- on gfx950 FETCH_SIZE
reports half of the bytes of a wide coalesced streaming read; WRITE_SIZE is taken as reported.  Separate --pmc passes.
"""
import glob
import json
import sys
from pathlib import Path

import pandas as pd

ROOT = Path(__file__).resolve().parent.parent
TAG = {"lookup_span_kernel": "lookup_span", "lookup_rows_kernel": "lookup_rows", "regex_sparse_kernel": "regex_split", "regex_split_kernel<0>": "regex_count", "regex_split_kernel<1>": "regex_write",
       "ragged_to_dense_kernel": "ragged_to_dense", "vocab_encoder_kernel": "vocab_encoder",
       "lookup_kernel<0>": "lookup_fused", "lookup_kernel<1>": "lookup_pieces", "lookup_kernel<2>": "lookup_fused",
       "merge_kernel": "bpe_merge",
       "split_seq_kernel<0>": "split_count", "split_seq_kernel<1>": "split_write",
       "exact_kernel": "bpe_exact", "compact_kernel": "compact", "special_sparse_kernel": "special_split", "row_width_kernel": "row_width", "prep_rows_kernel": "prep_rows",
       "count_scan_kernel": "count_scan", "wordpiece_deferred_kernel": "wordpiece_deferred",
       "decode_count_kernel": "decode_count", "decode_write_kernel": "detokenize"}


def short(name):
    n = name.replace("void ", "").split("(")[0].replace("ovtk::", "")
    return n


def newest(pattern):
    files = glob.glob(pattern)
    return max(files, key=lambda f: Path(f).stat().st_mtime) if files else None


def tag_of(kernel):
    """bench.py's name of a kernel: template arguments matter only for the first one of lookup / split kernels."""
    base, _, args = kernel.partition("<")
    first = args.split(",")[0].rstrip(">").strip() if args else ""
    keyed = f"{base}<{first}>" if base in ("lookup_kernel", "split_seq_kernel", "split_kernel", "regex_split_kernel") and first else base
    return TAG.get(keyed, TAG.get(base, kernel))


def main(prefix):
    out_dir = ROOT / "profiles" / Path(prefix).parent
    out_dir.mkdir(parents=True, exist_ok=True)
    stem = Path(prefix).name
    pmc_json = {}
    for cfg in (2, 3, 4, 5, "r2d", "vocab_encoder", "pipeline"):
        st = newest(str(ROOT / f"gpurun_out/prof_c{cfg}/*/*kernel_stats.csv"))
        if st:
            d = pd.read_csv(st)
            d = d[d["Name"].str.contains("ovtk")]
            d.to_csv(out_dir / f"{stem}_config{cfg}_kernel_stats.csv", index=False)
        st1 = newest(str(ROOT / f"gpurun_out/prof1_c{cfg}/*/*kernel_stats.csv"))  # the one-stream pass (bench.py --streams 1)
        if st1:
            d = pd.read_csv(st1)
            d = d[d["Name"].str.contains("ovtk")]
            d.to_csv(out_dir / f"{stem}_config{cfg}_one_stream_kernel_stats.csv", index=False)
        rows = []
        for ctr in ("fetch", "write"):
            f = newest(str(ROOT / f"gpurun_out/pmc_{ctr}_c{cfg}/*/*counter_collection.csv"))
            if not f:
                continue
            d = pd.read_csv(f)
            d = d[d["Kernel_Name"].str.contains("ovtk")]
            d["kernel"] = d["Kernel_Name"].map(short)
            # the steady state: the last four launches of every kernel (bench.py's priming pass -- every distinct batch once, the
            # memo and the store still filling -- comes first and is not what the timed region runs)
            d = d.sort_values("Dispatch_Id").groupby(["kernel", "Counter_Name"]).tail(4)
            g = d.groupby(["kernel", "Counter_Name"]).agg(dispatches=("Counter_Value", "size"), mean_KB=("Counter_Value", "mean"),
                                                         vgpr=("VGPR_Count", "first"), sgpr=("SGPR_Count", "first"),
                                                         lds=("LDS_Block_Size", "first")).reset_index()
            rows.append(g)
        if rows:
            t = pd.concat(rows)
            t.to_csv(out_dir / f"{stem}_config{cfg}_pmc_summary.csv", index=False)
            per = {}
            for k, g in t.groupby("kernel"):
                fetch = float(g[g.Counter_Name == "FETCH_SIZE"].mean_KB.sum())
                write = float(g[g.Counter_Name == "WRITE_SIZE"].mean_KB.sum())
                per[tag_of(k)] = int((2 * fetch + write) * 1024)
            if cfg == 3:   # bench.py's name for the lookup kernel run with the BERT words scanner (lookup_rows_kernel since r03)
                per["lookup_words"] = per.pop("lookup_span") if "lookup_span" in per else (per.pop("lookup_rows") if "lookup_rows" in per else per.get("lookup_fused", 0))
            pmc_json[f"config{cfg}"] = per
    latest = ROOT / "profiles" / "latest_pmc.json"
    if latest.exists():   # a run over some of the configurations (CONFIGS=3 ...) keeps the others' entries
        pmc_json = {**json.loads(latest.read_text()), **pmc_json}
    latest.write_text(json.dumps(pmc_json, indent=1, sort_keys=True) + "\n")
    print(json.dumps(pmc_json, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01/x")
