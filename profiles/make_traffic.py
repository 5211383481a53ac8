#!/usr/bin/env python3
"""Turns the output of profiles/collect.sh into the committed evidence of a round.

    python profiles/make_traffic.py gpurun_out/<tag> <round-name> [workload-key]

Writes profiles/<round-name>_kernel_stats.csv (the rocprofv3 --kernel-trace --stats summary of the bench command),
profiles/<round-name>_pmc/per_launch.txt (a per-launch table of the traversal, tail and shading kernels of one frame)
and updates profiles/traffic.json, which bench.py reads for roofline.traffic and the per-bounce figures.

Units.  TCC_EA0_RDREQ counts fabric read requests; one request is one 128-byte line, whether the kernel uses 64 or 128
bytes of it (profiles/r03_pmc/fetch_calib.txt: 1.00 requests per 128-byte line of a streaming read, 0.99 per
scattered 64-byte record, 0.99 per scattered 128-byte record; FETCH_SIZE = requests x 64 B / 1024 on gfx950, i.e. half
the bytes of the lines).  WRITE_SIZE is exact in KiB.  Memory-side bytes of a launch = requests x 128 + WRITE_SIZE x 1024.
valu_issue = SQ_INSTS_VALU / (cycles x 1024 SIMDs) / 0.431, the last being the wave instructions per SIMD and cycle a
stream of independent v_fma_f32 reaches under the same two counters (profiles/r03_pmc/pk_fma_probe.txt; on this part
SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU, so "x 4 cycles" says nothing about how busy the unit is);
lane_util = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64), cycles = GRBM_GUI_ACTIVE / 8 XCDs,
l2_hit = TCC_HIT / TCC_REQ, ta_busy = TA_TA_BUSY / (256 CUs x cycles).
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

VALU_PEAK_PER_SIMD_CYCLE = 0.431  # 4295180288 v_fma_f32 in 77877008 / 8 cycles on 1024 SIMDs

src, name = sys.argv[1], sys.argv[2]
key = sys.argv[3] if len(sys.argv) > 3 else "dragon_1920x1080_100spp"
here = os.path.dirname(os.path.abspath(__file__))


def short(kernel):
    return kernel.split("(")[0].replace("void ", "").replace("rtk::", "")


# counter -> kernel -> [(dispatch id, value)]
data = defaultdict(lambda: defaultdict(list))
files = glob.glob(os.path.join(src, "*", "*", "*counter_collection.csv"))
passes = [os.path.relpath(f, src).split(os.sep)[0] for f in files]
if len(set(passes)) != len(passes):
    raise SystemExit("more than one counter file per pass under %s (a previous run's files?): %s" % (src, sorted(files)))
for f in files:
    for r in csv.DictReader(open(f)):
        if "rtk::" not in r["Kernel_Name"]:
            continue
        data[r["Counter_Name"]][short(r["Kernel_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))


def launches(kernels):
    """Per launch of the named kernels, in dispatch order within their pass: {counter: value} plus the kernel name."""
    rows = None
    for counter, per_kernel in data.items():
        seq = sorted((d, v, k) for k in kernels for d, v in per_kernel.get(k, []))
        if rows is None:
            rows = [{"kernel": k} for _, _, k in seq]
        if len(seq) != len(rows):
            raise SystemExit("counter %s saw %d launches of %s, another saw %d" % (counter, len(seq), kernels, len(rows)))
        for row, (_, v, k) in zip(rows, seq):
            assert row["kernel"] == k
            row[counter] = v
    return rows or []


def derive(row):
    out = {"kernel": row["kernel"], "rdreq": row["TCC_EA0_RDREQ_sum"], "fetch_kib": row["FETCH_SIZE"],
           "write_kib": row["WRITE_SIZE"]}
    cyc = row["GRBM_GUI_ACTIVE"] / 8.0
    out["mcycles"] = cyc / 1e6
    out["l2_req"] = row["TCC_REQ_sum"]
    out["l2_hit"] = row["TCC_HIT_sum"] / max(row["TCC_REQ_sum"], 1.0)
    out["l1_accesses"] = row["TCP_TOTAL_CACHE_ACCESSES_sum"]
    out["ta_busy"] = row["TA_TA_BUSY_sum"] / (256.0 * cyc)
    out["valu_issue"] = row["SQ_INSTS_VALU"] / (cyc * 1024.0) / VALU_PEAK_PER_SIMD_CYCLE
    out["lane_util"] = row["SQ_THREAD_CYCLES_VALU"] / max(row["SQ_ACTIVE_INST_VALU"] * 64.0, 1.0)
    out["valu_insts"] = row["SQ_INSTS_VALU"]
    out["salu_insts"] = row["SQ_INSTS_SALU"]
    out["wait_any"] = row["SQ_WAIT_ANY"] / max(row["SQ_WAVE_CYCLES"], 1.0)
    out["wait_inst"] = row["SQ_WAIT_INST_ANY"] / max(row["SQ_WAVE_CYCLES"], 1.0)
    out["hbm_bytes"] = out["rdreq"] * 128.0 + out["write_kib"] * 1024.0
    return out


extend = [derive(r) for r in launches(["extend_kernel", "extend_queue_kernel"])]
for depth, r in enumerate(extend):  # one pass per frame on this workload: the launches of a frame are its bounces in order
    r["depth"] = depth
tail = [derive(r) for r in launches(["tail_kernel"])]
shade = [derive(r) for r in launches(["shade_kernel"])]
others = {k: [derive(r) for r in launches([k])] for k in ("raygen_kernel", "accumulate_kernel")}

stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(here, name + "_kernel_stats.csv"))
tb = os.path.join(src, "trace_bench.json")
if os.path.exists(tb):
    shutil.copy(tb, os.path.join(here, name + "_trace_bench.json"))
os.makedirs(os.path.join(here, name + "_pmc"), exist_ok=True)
with open(os.path.join(here, name + "_pmc", "per_launch.txt"), "w") as fh:
    fh.write("one frame of %s (rocprofv3 --pmc, one pass per counter set; sources %s)\n" %
             (key, open(os.path.join(src, "csrc_hash.txt")).read().strip()))
    hdr = "%-22s %5s %8s %9s %7s %9s %8s %8s %8s %9s %9s %8s %8s %8s\n" % (
        "kernel", "depth", "Mcycles", "lines(M)", "l2_hit", "l2_req(M)", "ta_busy", "hbm(GB)", "write(GB)", "valu_issue",
        "lane_util", "valu(G)", "wait_any", "wait_inst")
    fh.write(hdr)
    for group in (extend, tail, shade, others["raygen_kernel"], others["accumulate_kernel"]):
        for i, r in enumerate(group):
            fh.write("%-22s %5s %8.2f %9.1f %7.2f %9.1f %8.2f %8.2f %8.2f %9.2f %9.2f %8.2f %8.2f %8.2f\n" % (
                r["kernel"], r.get("depth", i), r["mcycles"], r["rdreq"] / 1e6, r["l2_hit"], r["l2_req"] / 1e6, r["ta_busy"],
                r["hbm_bytes"] / 1e9, r["write_kib"] * 1024 / 1e9, r["valu_issue"], r["lane_util"], r["valu_insts"] / 1e9,
                r["wait_any"], r["wait_inst"]))
    fh.write("frame: traversal launches %.1f GB memory-side (%.2f G lines), tail %.1f GB, shading %.1f GB\n" % (
        sum(r["hbm_bytes"] for r in extend) / 1e9, sum(r["rdreq"] for r in extend) / 1e9,
        sum(r["hbm_bytes"] for r in tail) / 1e9, sum(r["hbm_bytes"] for r in shade) / 1e9))

tfile = os.path.join(here, "traffic.json")
tj = json.load(open(tfile)) if os.path.exists(tfile) else {}
keep = ("kernel", "depth", "rdreq", "fetch_kib", "write_kib", "l2_hit", "valu_issue", "lane_util", "ta_busy", "hbm_bytes")
tj[key] = {
    "csrc_sha256": open(os.path.join(src, "csrc_hash.txt")).read().strip(),
    "from": "profiles/%s_pmc/per_launch.txt (profiles/collect.sh + profiles/make_traffic.py)" % name,
    "unit": "rdreq = fabric read requests = 128-byte lines; hbm_bytes = rdreq x 128 + write_kib x 1024",
    "extend": [{k: r[k] for k in keep if k in r} for r in extend],
    "tail": ({k: sum(r[k] for r in tail) if k in ("rdreq", "fetch_kib", "write_kib", "hbm_bytes") else tail[0][k]
              for k in keep if k in tail[0] and k != "depth"} if tail else None),
    "shade": [{k: r[k] for k in keep if k in r} for r in shade],
}
json.dump(tj, open(tfile, "w"), indent=1)
print(open(os.path.join(here, name + "_pmc", "per_launch.txt")).read())
