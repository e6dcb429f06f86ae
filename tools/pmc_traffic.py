"""Turns two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately as the
MI355X guide prescribes) into HBM bytes per launch for the kernels of interest.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ... --eager-body
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ... --eager-body
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r02_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled
for the streaming kernels listed below; WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import collections
import csv
import glob
import json
import sys

KERNELS = {  # bench op name -> substring of the kernel symbol
    "rowbn_bwd_apply": "rowbn_bwd_apply",        # plain and max variant
    "rowbn_bwd_reduce": "rowbn_bwd_reduce",
    "rowbn_fwd_stats": "rowbn_stats_kernel",
    "rowbn_fwd_apply": "rowbn_apply_kernel",
    "rowbn_fwd_apply_max": "rowbn_apply_max_kernel",
    "mlp_fwd": "mlp_fwd_kernel",
    "mlp_dgrad": "mlp_dgrad_kernel",
    "mlp_wgrad": "mlp_wgrad_kernel",
    "mlp_bn_bwd_apply": "mlp_bn_bwd_apply_",          # plain and row-sum form
    "small_tail_fwd": "small_tail_fwd_kernel",
    "small_tail_bwd": "small_tail_bwd_kernel",
    "knn_mfma": "knn_mfma_kernel",
    "knn_grid": "fg_knn_kernel",
    "frnn_grid": "fg_query_kernel",
    "rowcombine_fwd": "rowcombine_fwd_kernel",
    "rowcombine_bwd": "rowcombine_bwd_",             # thread- and wave-per-row forms
    "ball_query": "ball_query_kernel",
    "fps": "fps_kernel",
}


def collect(directory, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            name = row.get("Kernel_Name", "")
            for op, sub in KERNELS.items():
                if sub in name:
                    acc[op][0] += float(row["Counter_Value"])
                    acc[op][1] += 1
    return acc


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for op in KERNELS:
        if fetch[op][1] == 0 or write[op][1] == 0:
            continue
        f_kib = fetch[op][0] / fetch[op][1]
        w_kib = write[op][0] / write[op][1]
        out[op] = {"launches_sampled": fetch[op][1], "fetch_size_kib_raw": round(f_kib, 1),
                   "write_size_kib": round(w_kib, 1), "fetch_correction": 2.0,
                   "hbm_bytes_per_launch": int((2.0 * f_kib + w_kib) * 1024)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
