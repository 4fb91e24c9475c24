"""Condense rocprofv3 output (kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE passes of the
same bench.py command) into the small files kept under profiles/.
usage: python tools/profile_summary.py <gpurun_out/prof_dir> <tag>"""
import collections, csv, glob, importlib, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
source_digest = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd.build").source_digest()
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)

CLASS = [("block_rs_kernel", "block_tail"), ("pack_rs_weights", "pack_weights"), ("block_ws_kernel", "block_tail"), ("pack_tail_weights", "pack_weights"), ("mlp_kernel<true", "block_tail"), ("mlp_kernel", "mlp_fused"), ("attention_kernel", "attention"), ("pq_screen_kernel", "pq_assign"), ("pq_assign_kernel", "pq_assign"),
         ("pq_prep_codebook", "pq_prep"), ("gemm8_kernel<100,", "gemm_resid_ln(proj,fc2)"), ("gemm8_kernel<101,", "patch_gemm+pos+ln"),
         ("cls_rows_kernel", "patch_cls_rows"), ("probe_", "probes"), ("confusion_", "confusion"),
         ("layernorm_rowtile_kernel", "layernorm_tail(proj,fc2)"), ("layernorm_kernel", "layernorm"),
         ("gemm_sw_kernel<0>", "gemm_bf16out"), ("gemm_sw_kernel<1>", "gemm_fc1"), ("gemm_sw_kernel<2>", "head_gemm1"),
         ("gemm_ws_kernel<0,", "gemm_bf16out"), ("gemm_ws_kernel<1,", "gemm_fc1"), ("gemm_ws_kernel<2,", "head_gemm1"),
         ("gemm8_kernel<1,", "gemm_fc1"), ("gemm8_kernel<0,", "gemm_bf16out"), ("gemm8_kernel<2,", "head_gemm1"),
         ("gemm8_kernel<3, 256", "head_gemm2"), ("gemm8_kernel<3, 128", "patch_gemm"),
         ("gemm_bf16_kernel<4", "gemm_resid_tail(proj,fc2)"), ("gemm_bf16_kernel", "gemm_128"),
         ("patch_gather", "patch_gather"), ("assemble_tokens", "assemble_tokens"), ("pq_loss_reduce", "pq_loss_reduce")]


def classify(name):
    for key, cls in CLASS:
        if key in name:
            return cls
    return None


# trace  = `bench.py --streams 1` (every kernel alone on the device: the durations the roofline figures use)
# trace2 = `bench.py --streams 2` (batch spread over two streams: kernels of the two image ranges overlap)
for sub, suffix in (("trace", ""), ("trace2", "_2streams")):
    found = glob.glob(os.path.join(src, sub, "*", "*_kernel_stats.csv"))
    if not found:
        continue
    rows = list(csv.DictReader(open(max(found, key=os.path.getmtime))))   # (gpurun merges runs: keep the newest)
    with open(os.path.join(out, f"{tag}_kernel_stats{suffix}.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "class", "calls", "total_ms", "avg_us", "percent"])
        for r in rows:
            cls = classify(r["Name"])
            if cls is None and float(r["Percentage"]) < 0.05:
                continue
            w.writerow([r["Name"][:110], cls or "torch/runtime", r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 3),
                        round(float(r["AverageNs"]) / 1e3, 2), r["Percentage"]])

traffic = collections.defaultdict(lambda: {"fetch_raw_kb": 0.0, "write_raw_kb": 0.0, "launches_f": 0, "launches_w": 0})
for counter, sub, fld, cnt in (("FETCH_SIZE", "fetch", "fetch_raw_kb", "launches_f"), ("WRITE_SIZE", "write", "write_raw_kb", "launches_w")):
    for f in sorted(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            cls = classify(r["Kernel_Name"])
            if cls is None:
                continue
            traffic[cls][fld] += float(r["Counter_Value"])
            traffic[cls][cnt] += 1
summary = {}
for cls, t in traffic.items():
    if not t["launches_f"] or not t["launches_w"]:
        continue
    # rocprofv3 reports KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
    # stream, i.e. half the bytes read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
    fetch = 2.0 * t["fetch_raw_kb"] / t["launches_f"] * 1024
    write = t["write_raw_kb"] / t["launches_w"] * 1024
    summary[cls] = {"hbm_read_bytes_per_launch": round(fetch), "hbm_write_bytes_per_launch": round(write),
                    "hbm_bytes_per_launch": round(fetch + write), "launches_sampled": t["launches_f"]}
with open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w") as f:
    json.dump({"source_digest": source_digest, "note": "FETCH_SIZE and WRITE_SIZE from two separate rocprofv3 --pmc passes of bench.py; "
                       "FETCH_SIZE doubled (gfx950 counts half of a wide coalesced read); bytes per launch",
               "classes": summary}, f, indent=1, sort_keys=True)
flat = {"attention": summary.get("attention", {}).get("hbm_bytes_per_launch"),
        "pq_assign": summary.get("pq_assign", {}).get("hbm_bytes_per_launch"),
        "gemm_fc1": summary.get("gemm_fc1", {}).get("hbm_bytes_per_launch"),
        "mlp_fused": summary.get("mlp_fused", {}).get("hbm_bytes_per_launch"),
        "block_tail": summary.get("block_tail", {}).get("hbm_bytes_per_launch"),
        "gemm_qkv": summary.get("gemm_bf16out", {}).get("hbm_bytes_per_launch")}
flat = {k: v for k, v in flat.items() if v is not None}
# bench.py reports these numbers only while the kernel sources still are the ones they were measured on
flat["source_digest"] = source_digest
flat["measured_from"] = tag
with open(os.path.join(out, "pmc_traffic.json"), "w") as f:
    json.dump(flat, f, indent=1, sort_keys=True)
print(json.dumps(summary, indent=1))


# ---- matrix-pipe utilisation per kernel class (third PMC pass): SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs,
# GRBM_GUI_ACTIVE over the 8 XCDs; utilisation = busy cycles / (1024 SIMDs x kernel cycles)
mf = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(os.path.join(src, "mfma", "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        cls = classify(r["Kernel_Name"])
        if cls is None:
            continue
        mf[cls][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[cls].add(r["Dispatch_Id"])
    util = {}
    for cls, c in sorted(mf.items()):
        cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cycles <= 0 or c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) <= 0:
            continue
        n = len(seen[cls])
        util[cls] = {"mfma_busy_fraction": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles), 4),
                     "valu_issue_fraction": round(4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024.0 * cycles), 4),
                     "kernel_cycles_per_launch": round(cycles / n), "launches_sampled": n}
    if util:
        util["note"] = ("one rocprofv3 --pmc pass of bench.py --streams 1: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); "
                        "valu_issue_fraction = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) over the same denominator, MFMA issue included")
        util["source_digest"] = source_digest
        json.dump(util, open(os.path.join(out, f"{tag}_mfma_util.json"), "w"), indent=1, sort_keys=True)
        print(json.dumps(util, indent=1, sort_keys=True))
