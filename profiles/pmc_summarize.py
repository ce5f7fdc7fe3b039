#!/usr/bin/env python3
"""Aggregates the counter CSVs of profiles/pmc_run.sh: summary.txt (per kernel and counter, per dispatch) and
pmc_traffic.json (HBM bytes per launch of the X-engine kernels, tied to the sources they were taken from).
usage: pmc_summarize.py <dir with pass*/>"""
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-56:]
        a = agg[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
with open(out + "/summary.txt", "w") as fh:
    for k, d in agg.items():
        fh.write(k + "\n")
        for c, (tot, n) in sorted(d.items()):
            fh.write("   %-32s per-dispatch %16.1f  (n=%d)\n" % (c, tot / max(n, 1), n))
print(open(out + "/summary.txt").read())
# HBM traffic per launch of the X-engine kernel for bench.py's roofline.traffic, tied to the binary it was taken from
import hashlib, json, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
srcs = ["xcorr_kernels.h", "xcorr_tiling.h", "xcorr.hip"]
h = hashlib.sha256()
for f in srcs:
    h.update(open(os.path.join(root, "caltech-bifrost-dsp_amd", "csrc", f), "rb").read())
res = {"xcorr_sources_sha256": h.hexdigest(), "xcorr_sources": srcs,
       "correction": "FETCH_SIZE x2 (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section); "
                     "WRITE_SIZE exact for 16-B-per-lane stores; separate --pmc passes (profiles/pmc_run.sh)"}
for k, d in agg.items():
    if ("xcorr_" in k or "fused_kernel<" in k) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        if "fused_kernel<" in k:      # template arguments <ABL, LACC, DESC, TAB>: long accumulation in the epilogue; gulps by descriptor (packet slabs); through offset tables
            targs = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
            name = ("xcorr_fused_kernel" + ("_lacc" if targs[1] == "true" else "") + ("_slabs" if len(targs) > 2 and targs[2] == "true" else "") +
                    ("_tables" if len(targs) > 3 and targs[3] == "true" else ""))
        else:
            name = k.strip()
        f, w = d["FETCH_SIZE"][0] / d["FETCH_SIZE"][1], d["WRITE_SIZE"][0] / d["WRITE_SIZE"][1]
        res[name + "_bytes_per_launch"] = int(round((2 * f + w) * 1024))
        res[name] = {"FETCH_SIZE_KB_per_dispatch": round(f, 1), "WRITE_SIZE_KB_per_dispatch": round(w, 1),
                     "TCC_HIT_sum": d.get("TCC_HIT_sum", [0, 1])[0] / max(d.get("TCC_HIT_sum", [0, 1])[1], 1),
                     "TCC_MISS_sum": d.get("TCC_MISS_sum", [0, 1])[0] / max(d.get("TCC_MISS_sum", [0, 1])[1], 1)}
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
