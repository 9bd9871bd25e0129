"""Average the SQ counters of a rocprofv3 --pmc pass per kernel:  python tools/gemm_pmc_summary.py <dir> <label> <out.json>
(the counter run: rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
 --output-format csv -d <dir> -- python3 tools/gemm_probe.py e1 bf16)"""
import collections, csv, glob, json, os, sys
d, label, out = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "mli::gemm" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"note": "rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -- python3 "
               "tools/gemm_probe.py e1 bf16: decode projection x[1024,2048].[Wk|Wq|Wv], averages per launch; counters are chip-wide "
               "sums (MFMA busy over the 1024 SIMDs, CU busy over the 256 CUs)", "kernels": {}}
for k, c in agg.items():
    e = {n: sum(v) / len(v) for n, v in c.items()}
    e["launches"] = max(len(v) for v in c.values())
    if e.get("SQ_BUSY_CU_CYCLES"):
        e["mfma_busy_fraction_of_cu_busy"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 4 / e["SQ_BUSY_CU_CYCLES"]
    res["kernels"][f"{label}: {k}"] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
