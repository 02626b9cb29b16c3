"""Condenses the rocprofv3 runs of tools/<tag>_profiles.sh into the small tables that are
committed under profiles/ (run on the GPU box; results come back through gpurun_out/):

    python3 tools/pmc_summary.py gpurun_out/r06p r06 [outdir]

Per pass P of tools/roofline_run.py (one rocprofv3 run each, never blended):
    <tag>_<P>_kernel_stats.csv     copy of the --stats summary of `t_<P>`
    <tag>_bench_kernel_stats.csv   the same for `python3 bench.py --steps 200`
and
    <tag>_trailing_dispatches.csv  one row per launch with trailing-update work of the sequential
                                   N=16384, tile-256 potrf (128-tile and 64-tile products, the
                                   one-launch steps of the last rows): dispatch, grid, m, ns, flop --
                                   1.4318e12 / sum(ns) is SURVEY 8(d)'s trailing-update line, the
                                   bulk launches' own sum(m^2 nb) / sum(ns) stays beside it
    <tag>_pmc_traffic.json         WRITE_SIZE / FETCH_SIZE per kernel and pass, RAW, plus clearly
                                   labelled estimates (see below)
    <tag>_mfma_util.json           SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_VALU_MFMA_MOPS_F64 of
                                   gemm_lds_kernel (potrf256 and c5 passes)

Counters: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  Per MI355X_MICROARCH.md WRITE_SIZE
is exact for 16-B-per-lane streaming stores; FETCH_SIZE tallies 16-B-per-lane coalesced reads
at half and is UNCALIBRATED for other widths.  The raw counter is always reported.  Two
estimates are added and labelled as such: (a) gemm_lds_kernel: C-tile share (8 B per lane,
= WRITE_SIZE) kept, LDS-DMA share (16 B per lane) doubled -- the guide's rule; (b) the trsv step
kernels (8 B per lane, 512 contiguous bytes per wave; since round 4 one launch per sweep,
trsv_*_flow_kernel): raw x the factor MEASURED in the `calib`
pass on probe_read8_kernel, which reads a known 1 GiB per launch with exactly that pattern."""
import csv
import glob
import json
import os
import re
import shutil
import sys

PASSES = ("c2", "gram", "potrf256", "potrf256_dense", "potrf_engine", "trsv", "solve256", "predict",
          "c5", "c5_nola", "c2x256", "c3", "calib", "fitpost_n1024", "fitpost_n2048", "fitpost_n4096",
          "fitpost_n16384")
PEAK = 78.6e12


def short(name):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return f[0] if f else None


def counters(d):
    """{kernel: {counter: [values per dispatch in dispatch order]}}"""
    f = find(d, "*_counter_collection.csv")
    out = {}
    if not f:
        return out
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        out.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], []).append(
            float(r["Counter_Value"]))
    return out


def main():
    O, tag = sys.argv[1], sys.argv[2]
    outdir = sys.argv[3] if len(sys.argv) > 3 else os.path.join(O, "profiles")
    os.makedirs(outdir, exist_ok=True)
    # ---- kernel stats of every pass ------------------------------------------------
    for p in ("bench",) + PASSES:
        f = find(os.path.join(O, "bench" if p == "bench" else "t_" + p), "*_kernel_stats.csv")
        if f:
            shutil.copy(f, os.path.join(outdir, "%s_%s_kernel_stats.csv" % (tag, p)))
    # ---- the trailing update, dispatch by dispatch -----------------------------------
    summary, summaries = {}, {}
    for tpass in ("potrf256", "potrf256_dense"):
      tr = find(os.path.join(O, "t_" + tpass), "*_kernel_trace.csv")
      summary = {}
      if tr:
          # EVERY launch that carries trailing-update work (1.4318e12 flop over the sum
          # of all trailing kernel time): the 128-tile and 64-tile LDS products and the one-launch
          # steps that take over for the last rows (slab_step_kernel: solve + update + next factor)
          names = ("gemm_lds_kernel", "gemm_lds64_kernel", "slab_step_kernel")
          rows = [r for r in csv.DictReader(open(tr)) if any(k in r["Kernel_Name"] for k in names)]
          rows.sort(key=lambda r: int(r["Start_Timestamp"]))
          nb = 256
          table, tot = [], {"bulk": [0, 0.0, 0], "small": [0, 0.0, 0], "tail_steps": [0, 0.0, 0]}
          for r in rows:
              wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
              ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
              kname = r["Kernel_Name"].split("(")[0].replace("void ", "")
              T = int(((8 * wgs + 1) ** 0.5 - 1) / 2 + 0.5)
              if "slab_step_kernel" in kname:
                  # one 64-column step over the m = 64 T rows below it: m^2 64 (+ m 64^2 of solve)
                  m, kk, cls = 64 * T, 64, "tail_steps"
                  fl = float(m) * m * 64 if T * (T + 1) // 2 == wgs else 0.0
              else:
                  tile = 64 if "gemm_lds64_kernel" in kname else 128
                  if T * (T + 1) // 2 != wgs:   # not a square lower update: a panel-internal product
                      continue
                  m, kk = tile * T, nb
                  fl = float(m) * m * nb
                  cls = "bulk" if (tile == 128 and wgs >= 256) else "small"
              tot[cls][0] += 1
              tot[cls][1] += fl
              tot[cls][2] += ns
              table.append((r["Dispatch_Id"], kname[:40], wgs, m, kk, ns, fl, cls))
          with open(os.path.join(outdir, "%s_trailing_dispatches%s.csv" % (tag, "" if tpass == "potrf256" else "_dense")), "w") as f:
              f.write("# every launch with trailing-update work of ONE sequential N=16384 potrf, outer "
                      "block 256 (python3 tools/roofline_run.py %s under rocprofv3 --kernel-trace); " % tpass +
                      "flop = m^2 k (the lower half of 2 m^2 k); tail_steps: the one-launch 64-column "
                      "steps over the last rows (their time includes the panel solve and the next "
                      "diagonal factor)\n")
              f.write("dispatch_id,kernel,workgroups,m,k,duration_ns,algorithmic_flop,class\n")
              for t in table:
                  f.write("%s,%s,%d,%d,%d,%d,%.0f,%s\n" % t)
              for cls in ("bulk", "small", "tail_steps"):
                  n, fl, ns = tot[cls]
                  if n:
                      f.write("# %s: %d launches, %.4e flop, %.3f ms -> %.2f TFLOP/s = %.3f of 78.6\n"
                              % (cls, n, fl, ns / 1e6, fl / ns / 1e3, fl / ns * 1e9 / PEAK))
              nall = sum(v[0] for v in tot.values())
              nsall = sum(v[2] for v in tot.values())
              if nsall:
                  f.write("# SURVEY 8(d): all %d trailing launches, 1.4318e12 flop / %.3f ms -> %.2f "
                          "TFLOP/s = %.3f of 78.6\n"
                          % (nall, nsall / 1e6, 1.4318e12 / nsall / 1e3, 1.4318e12 / nsall * 1e9 / PEAK))
                  summary["all_trailing_8d"] = {"launches": nall, "flop": 1.4318e12, "ms": nsall / 1e6,
                                                "tflops": 1.4318e12 / nsall / 1e3,
                                                "frac": 1.4318e12 / nsall * 1e9 / PEAK}
          for cls in ("bulk", "small"):
              n, fl, ns = tot[cls]
              if n:
                  summary[cls] = {"launches": n, "flop": fl, "ms": ns / 1e6,
                                  "tflops": fl / ns / 1e3, "frac": fl / ns * 1e9 / PEAK,
                                  "avg_launch_us": ns / n / 1e3}
      summaries[tpass] = summary
    summary = summaries.get("potrf256", {})
    # ---- traffic ----------------------------------------------------------------------
    out = {"_source": "tools/" + tag + "_profiles.sh: rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE, "
                      "separate runs, one per pass of tools/roofline_run.py; bytes = KiB x 1024; "
                      "*_bytes_* are RAW counters, *_estimate_* are labelled corrections "
                      "(tools/pmc_summary.py docstring)",
           "passes": {}, "kernels": {}, "trailing_update_from_trace": summary,
           "trailing_update_dense_from_trace": summaries.get("potrf256_dense", {})}
    calib = None
    for p in PASSES:
        w, f = counters(os.path.join(O, "w_" + p)), counters(os.path.join(O, "f_" + p))
        if not w and not f:
            continue
        d = out["passes"].setdefault(p, {})
        for k in sorted(set(w) | set(f)):
            e = d.setdefault(k, {})
            for src, name in ((w, "WRITE_SIZE"), (f, "FETCH_SIZE")):
                v = src.get(k, {}).get(name)
                if v:
                    e[name + "_bytes_avg"] = sum(v) / len(v) * 1024
                    e[name + "_bytes_total"] = sum(v) * 1024
                    e["launches"] = len(v)
    pc = out["passes"].get("calib", {}).get("probe_read8_kernel")
    if pc and pc.get("FETCH_SIZE_bytes_avg"):
        known = float(1 << 30)
        calib = known / pc["FETCH_SIZE_bytes_avg"]
        out["read8_calibration"] = {
            "known_bytes_per_launch": known, "FETCH_SIZE_raw_bytes_avg": pc["FETCH_SIZE_bytes_avg"],
            "factor": calib,
            "note": "probe_read8_kernel: 8 B per lane, 512 contiguous bytes per wave, every byte "
                    "of a 1 GiB buffer once per launch"}
    # flat view for bench.py: the pass that owns each roofline kernel
    own = {"gemm_lds_kernel": "potrf256", "gram_tri_kernel<2>": "gram", "gram_tri_kernel<1>": "gram",
           "trsv_fwd_flow_kernel<8>": "trsv", "trsv_bwd_flow_kernel": "trsv",
           "slab_step_kernel<false, 8>": "c2"}
    for k, p in own.items():
        e = out["passes"].get(p, {}).get(k)
        if not e:
            continue
        e = dict(e, **{"pass": p})
        if k == "gemm_lds_kernel" and "FETCH_SIZE_bytes_avg" in e and "WRITE_SIZE_bytes_avg" in e:
            c = min(e["WRITE_SIZE_bytes_avg"], e["FETCH_SIZE_bytes_avg"])
            e["FETCH_SIZE_estimate_bytes_avg"] = c + 2.0 * (e["FETCH_SIZE_bytes_avg"] - c)
            e["FETCH_SIZE_estimate_rule"] = ("ESTIMATE: C tile share (= WRITE_SIZE, 8 B/lane "
                                             "loads) kept, LDS-DMA share (16 B/lane) doubled")
        if k.startswith("trsv") and calib and "FETCH_SIZE_bytes_total" in e:
            e["FETCH_SIZE_estimate_bytes_total"] = calib * e["FETCH_SIZE_bytes_total"]
            e["FETCH_SIZE_estimate_rule"] = ("ESTIMATE: raw x %.3f, the factor measured on "
                                             "probe_read8_kernel (same access pattern, known bytes)"
                                             % calib)
        out["kernels"][k if p != "c2" else k + " [C2 passes]"] = e
    with open(os.path.join(outdir, "%s_pmc_traffic.json" % tag), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    # ---- MFMA utilisation ----------------------------------------------------------------
    mu = {"_source": "tools/" + tag + "_profiles.sh: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES "
                     "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE (own runs, "
                     "counters only); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 "
                     "XCDs x 256 CUs x 4 SIMDs)", "passes": {}}
    for p in ("potrf256", "potrf256_dense", "c5"):
        cs = counters(os.path.join(O, "m_" + p))
        # the kernel that carries the pass's trailing updates: the 128-tile LDS kernel for the
        # N = 16384 factorisation, the 64-tile one for the half-batches of a C5 shard
        kname = "gemm_lds_kernel" if p.startswith("potrf256") else next(
            (k for k in cs if k.startswith("gemm_lds64_kernel<false")), "gemm_lds_kernel")
        m = cs.get(kname)
        if not m:
            continue
        n = len(m["GRBM_GUI_ACTIVE"])
        sel = range(n)
        if p.startswith("potrf256"):   # the bulk launches only (>= 256 workgroups = the first ones)
            nbulk = summaries.get(p, {}).get("bulk", {}).get("launches", n)
            sel = range(min(n, nbulk))
        s = {c: sum(v[i] for i in sel) for c, v in m.items()}
        mu["passes"][p] = {
            "kernel": kname, "launches": len(sel), "sums": s,
            "mfma_util": s["SQ_VALU_MFMA_BUSY_CYCLES"] / (s["GRBM_GUI_ACTIVE"] / 8 * 256 * 4),
            "flop_from_MOPS": s["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0,
            "algorithmic_flop": (summaries.get(p, {}).get("bulk", {}).get("flop")
                                 if p.startswith("potrf256") else None)}
    with open(os.path.join(outdir, "%s_mfma_util.json" % tag), "w") as f:
        json.dump(mu, f, indent=1, sort_keys=True)
    print(json.dumps({"trailing": summary, "read8": out.get("read8_calibration"),
                      "mfma": {k: v["mfma_util"] for k, v in mu["passes"].items()}}, indent=1))


if __name__ == "__main__":
    main()
