import sys, json
for ln in sys.stdin:
    if ln.startswith("{"):
        d = json.loads(ln)
        r = d["roofline"]
        print(d["value"], d["ms_per_step"], "overlapped", d["kernel_ms"], "one-stream", r["one_stream_kernel_ms"], "sum", r["one_stream_kernel_sum_ms_per_step"], "frac", r["frac"], r["kernel"], "parity", d["parity_prefix_bit_exact"])
