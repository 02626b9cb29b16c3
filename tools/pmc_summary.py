"""Summarises rocprofv3 passes of tools/roofline_run.py into profiles/:
    python tools/pmc_summary.py gpurun_out/rf_trace gpurun_out/rf_pmc_write gpurun_out/rf_pmc_fetch r01
writes profiles/<tag>_roofline_kernel_stats.csv (copy of the --stats summary) and
profiles/<tag>_pmc_traffic.json (per-kernel average FETCH_SIZE / WRITE_SIZE in bytes).
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  Per MI355X_MICROARCH.md,
WRITE_SIZE is exact for 16-B-per-lane streaming stores; FETCH_SIZE under-counts wide
(16 B/lane) coalesced reads by 2x and is uncalibrated for other widths -- the GEMM
reads 8 B/lane, so its fetch figure is kept raw and flagged."""
import glob
import json
import os
import shutil
import sys

import pandas as pd


def main():
    trace, pw, pf, tag = sys.argv[1:5]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stats = glob.glob(os.path.join(trace, "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(root, "profiles", "%s_roofline_kernel_stats.csv" % tag))
    st = pd.read_csv(stats)
    out = {"_source": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes) of "
                      "`python3 tools/roofline_run.py`; bytes = counter KiB x 1024; "
                      "FETCH_SIZE_bytes_avg is RAW; FETCH_SIZE_corrected_bytes_avg applies the guide's x2 "
                      "to the 16-B-per-lane share where that share is known",
           "kernels": {}}
    for path, name in ((pw, "WRITE_SIZE"), (pf, "FETCH_SIZE")):
        f = glob.glob(os.path.join(path, "**", "*_counter_collection.csv"), recursive=True)[0]
        t = pd.read_csv(f)
        t = t[t.Counter_Name == name]
        t["short"] = t.Kernel_Name.str.extract(r"(\w+_kernel(?:<[^>]*>)?)", expand=False)
        for k, g in t.groupby("short"):
            d = out["kernels"].setdefault(k, {})
            d[name + "_bytes_avg"] = float(g.Counter_Value.mean() * 1024)
            d[name + "_bytes_total"] = float(g.Counter_Value.sum() * 1024)
            d["launches"] = int(len(g))
    # the headline workload's launches alone: roofline_run.py runs its 20 C2 passes first (16
    # slab steps each); later slab steps belong to the tails of the N=16384 factorisations
    for path, name in ((pw, "WRITE_SIZE"), (pf, "FETCH_SIZE")):
        f = glob.glob(os.path.join(path, "**", "*_counter_collection.csv"), recursive=True)[0]
        t = pd.read_csv(f)
        t = t[(t.Counter_Name == name) & t.Kernel_Name.str.contains("slab_step_kernel<false>",
                                                                      regex=False)]
        t = t.sort_values("Dispatch_Id").head(320)
        d = out["kernels"].setdefault("slab_step_kernel<false> [C2 passes]", {})
        d[name + "_bytes_avg"] = float(t.Counter_Value.mean() * 1024)
        d["launches"] = int(len(t))
    # the trsv step kernels read the factor with fully coalesced 512-byte wave loads (8 B per
    # lane): FETCH_SIZE tallies them at half like the guide's wide reads -- 1.085 GB raw per
    # N=16384 solve against 2.147 GB that the two sweeps must read; corrected = 2 x raw
    for kn in ("trsv_fwd_step_kernel<8>", "trsv_bwd_step_kernel"):
        g = out["kernels"].get(kn)
        if g and "FETCH_SIZE_bytes_total" in g:
            g["FETCH_SIZE_corrected_bytes_total"] = 2.0 * g["FETCH_SIZE_bytes_total"]
            g["FETCH_SIZE_correction"] = "coalesced 512-B wave loads tallied at half: x2"
    # gemm_lds_kernel reads its C tile with 8-B-per-lane loads -- every tile once, exactly
    # what it writes -- and stages P / Q by LDS-DMA, 16 B per lane, which FETCH_SIZE tallies
    # at half (MI355X_MICROARCH.md, HBM): corrected fetch = C share + 2 x the rest.
    g = out["kernels"].get("gemm_lds_kernel")
    if g and "FETCH_SIZE_bytes_avg" in g and "WRITE_SIZE_bytes_avg" in g:
        cshare = min(g["WRITE_SIZE_bytes_avg"], g["FETCH_SIZE_bytes_avg"])
        g["FETCH_SIZE_corrected_bytes_avg"] = cshare + 2.0 * (g["FETCH_SIZE_bytes_avg"] - cshare)
        g["FETCH_SIZE_correction"] = ("C tile share (= WRITE_SIZE, 8 B/lane loads) kept, the "
                                      "LDS-DMA share (16 B/lane) doubled")
    st["short"] = st.Name.str.extract(r"(\w+_kernel(?:<[^>]*>)?)", expand=False)
    for _, r in st.iterrows():
        if r.short in out["kernels"]:
            out["kernels"][r.short]["avg_ns"] = float(r.AverageNs)
            out["kernels"][r.short]["calls_in_trace"] = int(r.Calls)
    with open(os.path.join(root, "profiles", "%s_pmc_traffic.json" % tag), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out["kernels"].get("gemm_lds_kernel"), indent=1))
    print(json.dumps(out["kernels"].get("gram_tri_kernel<2>"), indent=1))


if __name__ == "__main__":
    main()
