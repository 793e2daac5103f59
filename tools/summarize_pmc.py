"""Turn rocprofv3 --pmc CSV output (tools/pmc_run.py runs) into the summaries kept under profiles/.

    python tools/summarize_pmc.py traffic <FETCH_SIZE dir> <WRITE_SIZE dir> <out.txt> <out.json>
    python tools/summarize_pmc.py mfma <counter dir> <out.txt>

Window = the dispatches of the LAST closure of the run (from after the previous loss_total_kernel up to and
including the last one).  FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derives them as requests x size / 1024); on gfx950 FETCH_SIZE tallies the 128-B requests of
wide streaming reads as 64 B (MI355X_MICROARCH.md, HBM section), so fetch bytes are doubled."""
import csv
import glob
import json
import os
import re
import sys
from collections import OrderedDict, defaultdict


def rows_of(directory):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {directory}")
    with open(files[0], newline="") as f:
        return list(csv.DictReader(f))


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("nst::", "").replace("(anonymous namespace)::", "")


def last_closure(rows):
    """dispatch ids of the last closure, in order"""
    disp = OrderedDict()
    for r in rows:
        disp.setdefault(int(r["Dispatch_Id"]), r["Kernel_Name"])
    ids = list(disp)
    marks = [i for i, d in enumerate(ids) if "loss_total" in disp[d] or "loss_assemble" in disp[d]]
    if len(marks) < 2:
        raise SystemExit("need at least two closures in the trace")
    return ids[marks[-2] + 1: marks[-1] + 1]


def per_dispatch(rows):
    out = defaultdict(dict)
    meta = {}
    for r in rows:
        d = int(r["Dispatch_Id"])
        out[d][r["Counter_Name"]] = out[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        meta[d] = r
    return out, meta


def traffic(fetch_dir, write_dir, out_txt, out_json):
    table = OrderedDict()
    for directory, counter in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
        rows = rows_of(directory)
        vals, meta = per_dispatch(rows)
        for d in last_closure(rows):
            k = short(meta[d]["Kernel_Name"])
            e = table.setdefault(k, {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
            if counter == "FETCH_SIZE":
                e["n"] += 1
            e[counter] += vals[d].get(counter, 0.0) * 1024.0
    lines = ["# HBM traffic per kernel over ONE L=2 closure (default schedule), rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE",
             "# separate passes: rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python tools/pmc_run.py 3 2",
             "# counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests of wide (16 B/lane) streaming reads as 64 B",
             "# (MI355X_MICROARCH.md, HBM), hence the x2 column. Window = the dispatches of the last closure.",
             f"{'kernel':<46}{'n':>5}{'FETCH GB raw':>14}{'FETCH GB x2':>13}{'WRITE GB':>10}"]
    tot = [0.0, 0.0]
    js = {}
    for k, e in sorted(table.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
        f, w = e["FETCH_SIZE"] / 1e9, e["WRITE_SIZE"] / 1e9
        lines.append(f"{k:<46}{e['n']:>5}{f:>14.3f}{2 * f:>13.3f}{w:>10.3f}")
        tot[0] += f
        tot[1] += w
        js[k] = {"launches": e["n"], "fetch_bytes_corrected": 2 * e["FETCH_SIZE"], "write_bytes": e["WRITE_SIZE"]}
    lines.append(f"{'TOTAL':<46}{'':>5}{tot[0]:>14.3f}{2 * tot[0]:>13.3f}{tot[1]:>10.3f}")
    open(out_txt, "w").write("\n".join(lines) + "\n")
    json.dump({"what": "HBM bytes per kernel over one L=2 closure, rocprofv3 PMC (FETCH_SIZE x2 correction, WRITE_SIZE)",
               "kernels": js}, open(out_json, "w"), indent=1)
    print("\n".join(lines))


def mfma(directory, out_txt):
    rows = rows_of(directory)
    vals, meta = per_dispatch(rows)
    lines = ["# MFMA utilisation of the batched 3x3 conv launches (conv_h2_batch_kernel / conv_wino_batch_kernel / conv_bf3_batch_kernel), one L=2 closure",
             "# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY",
             "#   --output-format csv -- python tools/pmc_run.py 3 2",
             "# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CU * 4 SIMD); clk ~ GRBM_GUI_ACTIVE/8/duration"]
    busy_sum = cyc_sum = 0.0
    for d in last_closure(rows):
        m = meta[d]
        if "conv_bf3" not in m["Kernel_Name"] and "conv_h2" not in m["Kernel_Name"] and "conv_wino" not in m["Kernel_Name"]:
            continue
        v = vals[d]
        dur = (int(m["End_Timestamp"]) - int(m["Start_Timestamp"])) / 1e3
        gui = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        util = busy / (gui * 256 * 4) if gui else 0.0
        wait = v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else 0.0
        busy_sum += busy
        cyc_sum += gui * 1024
        lines.append(f"{short(m['Kernel_Name']):<34} grid {int(m['Grid_Size']):>8} dur {dur:>6.0f} us "
                     f"clk~{gui / dur / 1e3:.2f} GHz mfma_util {util:.3f} wait_any/wave_cyc {wait:.2f}")
    if cyc_sum:
        lines.append(f"# all 3x3 conv launches of the closure: mfma_util {busy_sum / cyc_sum:.3f}")
    open(out_txt, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    if len(sys.argv) >= 6 and sys.argv[1] == "traffic":
        traffic(*sys.argv[2:6])
    elif len(sys.argv) >= 4 and sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3])
    else:
        raise SystemExit(__doc__)
