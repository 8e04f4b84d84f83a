"""HBM traffic of the GEMM family from two rocprofv3 --pmc passes of bench.py (FETCH_SIZE; WRITE_SIZE), as
MI355X_MICROARCH.md prescribes: separate passes, FETCH_SIZE doubled for 16-byte-per-lane streaming loads (gfx950 counts
128-byte requests at 64 bytes), WRITE_SIZE taken as is; both counters are in KiB.

    python tools/gemm_traffic.py <dir of the FETCH pass> <dir of the WRITE pass> <out.json>"""
import csv
import glob
import json
import sys


def collect(d, counter):
    tot, n = 0.0, 0
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if any(k in r["Kernel_Name"] for k in ("gemm_kernel", "gemm_rs_kernel", "gemm_group_kernel")) and r["Counter_Name"] == counter:
                    tot += float(r["Counter_Value"])
                    n += 1
    return tot, n


def main():
    fetch_kib, nf = collect(sys.argv[1], "FETCH_SIZE")
    write_kib, nw = collect(sys.argv[2], "WRITE_SIZE")
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_tree_hash
    out = {"kernel": "gemm_kernel<...> + gemm_rs_kernel<...> + gemm_group_kernel<...> (all instantiations)", "csrc_hash": csrc_tree_hash(), "launches_fetch_pass": nf, "launches_write_pass": nw,
           "fetch_bytes_per_launch": 2.0 * fetch_kib * 1024 / max(nf, 1), "write_bytes_per_launch": write_kib * 1024 / max(nw, 1),
           "correction": "FETCH_SIZE x2 (gfx950 wide streaming reads), WRITE_SIZE x1; KiB -> bytes"}
    out["hbm_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
