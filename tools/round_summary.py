"""profiles/<tag>_bench_summary.txt from one tools/profile_round.sh output directory: kernel families of the rocprofv3 --stats
pass, the GEMM family's launch count / average duration, the PMC traffic and the bench line measured in the same call.
    python tools/round_summary.py gpurun_out/<dir> <tag> [steps_in_trace]"""
import csv
import json
import os
import sys


def main():
    d, tag = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    rows = list(csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)

    def fam(n):
        if "gemm" in n:
            return "GEMM family (gemm_kernel + gemm_rs_kernel + gemm_group_kernel)"
        if "msda" in n:
            return "MSDA (fwd_rec, bwd_offw, bwd_value_fx / bwd_value)"
        if "add_ln" in n or "groupnorm" in n or "bn_relu" in n or "affine_act" in n:
            return "normalisation / folded-BN passes"
        if "attn" in n or "flash" in n:
            return "attention cores (flash_*, attn_*)"
        if "at::" in n or "rocclr" in n:
            return "framework kernels (at::*, rocclr copy / fill)"
        return "other library kernels (loss, optimizer, embeddings, GCN, sums)"

    fams = {}
    for r in rows:
        f = fams.setdefault(fam(r["Name"]), [0, 0])
        f[0] += int(r["Calls"]); f[1] += int(r["TotalDurationNs"])
    g = fams["GEMM family (gemm_kernel + gemm_rs_kernel + gemm_group_kernel)"]
    out = [f"# rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_decode --no_roofline`",
           f"# (tools/profile_round.sh), {steps} training steps in the trace (eager and hipGraph legs: warm-up + timed + host-cost probes)",
           f"all kernels: {calls} launches, {tot / 1e6:.1f} ms = {tot / steps / 1e6:.2f} ms of kernel time per step, {calls / steps:.0f} launches per step"]
    for k, (c, t) in sorted(fams.items(), key=lambda kv: -kv[1][1]):
        out.append(f"  {k}: {c} launches ({c / steps:.1f} per step), {t / steps / 1e6:.2f} ms per step, {100 * t / tot:.1f} % of kernel time, average {t / c / 1e3:.2f} us")
    out.append(f"GEMM family average launch duration in this trace: {g[1] / g[0] / 1e3:.2f} us (weight-gradient groups overlap the main chain on the side "
               "stream and stretch under the profiler; bench.py's serialised HIP-event figure is the one in roofline.avg_launch_us)")
    for r in rows:
        n = r["Name"]
        if "msda_bwd_value" in n or "flash_" in n:
            i = n.find("namespace)::")
            out.append(f"  {n[i + 12:i + 60] if i >= 0 else n[:48]}: {int(r['Calls']) / steps:.1f} per step, average {float(r['AverageNs']) / 1e3:.1f} us")
    tp = os.path.join(d, "gemm_traffic.json")
    if os.path.exists(tp):
        tr = json.load(open(tp))
        out.append(f"HBM traffic of the GEMM family (two --pmc passes, FETCH_SIZE x2 + WRITE_SIZE): {tr['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch "
                   f"({tr['launches_fetch_pass']} launches counted), csrc_hash {tr['csrc_hash']}")
    bp = os.path.join(d, "bench_line.json")
    if os.path.exists(bp):
        b = json.loads(open(bp).read().strip().splitlines()[-1])
        rf = b.get("roofline") or {}
        out.append(f"bench line of the same tree and box: {b['value']} {b['unit']} = {b['ms_per_step']} ms/step ({b.get('launch')}); "
                   f"launch_modes {json.dumps(b.get('launch_modes'))}")
        if rf:
            out.append(f"roofline: bound {rf.get('bound')}, frac {rf.get('frac')}, achieved {rf.get('achieved')} {rf.get('unit')}, algorithmic "
                       f"{rf.get('algorithmic_bytes_per_launch', 0) / 1e6:.1f} MB per launch, {rf.get('launches_per_step')} launches / "
                       f"{rf.get('products_per_step')} products per step, {rf.get('gemm_ms_per_step')} ms per step, avg {rf.get('avg_launch_us')} us; "
                       f"traffic_stale {rf.get('traffic_source', {}).get('traffic_stale')}")
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_bench_summary.txt")
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
