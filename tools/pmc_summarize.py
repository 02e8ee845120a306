"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only)
of `bench.py --no-kernel-timing` into per-kernel HBM traffic per launch.

gfx950 correction (MI355X_MICROARCH.md, HBM section; re-checked for 8-byte-per-lane fp64 streams
with tools/pmc_calib.hip: a 512 MiB read reports FETCH_SIZE = 262 155 KB, a 512 MiB write reports
WRITE_SIZE = 524 288 KB): traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.

usage: python tools/pmc_summarize.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [label]
"""
import collections
import csv
import json
import sys


def avg(path, name):
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            kernel = r["Kernel_Name"].split("(")[0].replace("ssba::", "").replace("void ", "").split("<")[0]
            if kernel.endswith("_w"):      # window-layout variants report under the kernel class of bench.py
                kernel = kernel[:-2]
            if kernel.endswith("_mf"):     # matrix-core block factor / reduce (ssba_bcr_mfma.hip)
                kernel = kernel[:-3]
            a[kernel].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in a.items()}


def main():
    f, w = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
    out = {"label": sys.argv[4] if len(sys.argv) > 4 else "", "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE) KB; FETCH_SIZE halves "
           "coalesced reads on gfx950 (calibrated for 8- and 16-byte-per-lane reads with tools/pmc_calib.hip)", "kernels": {}}
    for k in sorted(f):
        fk, wk = f[k][1], w.get(k, (0, 0.0))[1]
        out["kernels"][k] = {"launches_sampled": f[k][0], "FETCH_SIZE_KB_avg": fk, "WRITE_SIZE_KB_avg": wk,
                             "traffic_bytes_per_launch": (2 * fk + wk) * 1024}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:28s} {v['traffic_bytes_per_launch'] / 1e6:10.2f} MB/launch")


if __name__ == "__main__":
    main()
