"""Build profiles/rNN_pmc_traffic.json from the per-counter CSVs that tools/rocpd_summary.py (pmc) or rocprofv3's own
counter_collection CSV produce.  usage: pmc_traffic_json.py <out.json> <step>:<tag>=<counter_collection.csv> ...
<step> names the profiled driver (fwd, train_step, cnn1d_fwd, cae_score, cae_train_step, ...): besides the per-kernel means the
file gets "steps": {<step>: {"hbm_bytes_per_step": sum over every dfa kernel launch of the run / launches of the driver's loop}}.
gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE (KB, 64-B requests tallied for 128-B ones) is doubled;
WRITE_SIZE (KB) is exact; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch (mean over the launches seen)."""
import csv, json, sys, collections


def read_counter_csv(path):
    """rocprofv3 --output-format csv counter_collection: columns Kernel_Name, Counter_Name, Counter_Value (one row per dispatch)."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


STEPS_IN_RUN = {"fwd": 6, "train_step": 6, "cnn1d_fwd": 10, "cnn1d_train_step": 6, "cae_score": 10, "cae_train_step": 6}   # loops of tools/gpu_prof_*.py
STEP_ALGO = {   # algorithmic bytes per step at B = 256 (read x once + write the outputs)
    "cnn1d_fwd": 256 * 231_124,
    "cae_score": 256 * (321 * 180 * 2 + 4),          # bf16 features in, one float out
}

ALGO = {   # algorithmic bytes per launch at B = 256, T = 321, F = 180
    "cnn1d_fused_kernel<": 256 * 231_124,
    "cae_dec_fused_kernel": 256 * (20 * 11 * 256 * 2 + 321 * 180 * 2 + 7 * 4),        # latent in, x (bf16) in, partial sums out
    "conv12_fused_kernel": 256 * (321 * 180 * 2 + 80 * 180 * 64 * 2),
    "conv3_m16_meant_kernel": 256 * (80 * 180 * 64 * 2 + 128 * 180 * 4),
    "conv_split_kernel<32, 4, 0": 256 * (160 * 180 * 32 * 4 + 80 * 180 * 64 * 4),     # bf16x3 block 2: a1 split in, a2 split out
    "conv_split_kernel<64, 8, 1": 256 * (80 * 180 * 64 * 4 + 128 * 180 * 4),          # bf16x3 block 3: a2 split in, embedding out
    "conv_split_kernel<128, 4, 2": 256 * (80 * 180 * 128 * 2 + 80 * 180 * 64 * 2),    # train: dz3 in, da2 out
    "conv_split_kernel<64, 2, 2": 256 * (160 * 180 * 64 * 2 + 160 * 180 * 32 * 2),    # train: dz2 in, da1 out
    "conv3_m16_meant_kernel<true, true>": 256 * (80 * 180 * 64 * 2 + 80 * 180 * 128 * 2),   # train forward 3: a2 in, z3 out
    "conv1_bn_relu_poolh2_kernel": 256 * (321 * 180 * 4 + 160 * 180 * 32 * 4),
    "conv1_mfma_kernel<0>": 256 * (321 * 180 * 2),                                     # train statistics: x in
    "conv1_mfma_kernel<1>": 256 * (321 * 180 * 2 + 160 * 180 * 32 * 2),               # train forward 1: x in, a1 out
    "conv1_mfma_kernel<2>": 256 * (321 * 180 * 2 + 160 * 180 * 32 * 2),               # train backward 1: x, da1 in
    "wgrad3x3_bf16_v3_kernel<64, 128": 256 * (80 * 180 * 128 * 2 + 80 * 180 * 64 * 2),     # dz3, a2 in
    "wgrad3x3_bf16_v3_kernel<32, 64": 256 * (160 * 180 * 64 * 2 + 160 * 180 * 32 * 2),     # dz2, a1 in
    "conv3x3_mfma_kernel<dfa::bf16_t, 32, 2, 2, 2, 1, 3": 256 * (160 * 180 * 32 * 2 + 160 * 180 * 64 * 2),   # train forward 2: a1 in, z2 out
    "bn_relu_poolh2_drop_kernel": 256 * (160 * 180 * 64 * 2 + 80 * 180 * 64 * 2),
    "bn_relu_meant_kernel": 256 * (80 * 180 * 128 * 2),
    "bn_bwd_apply_meant_kernel": 256 * (80 * 180 * 128 * 2 * 2),
    "bn_bwd_reduce_pool_kernel": 256 * (160 * 180 * 64 * 2 + 80 * 180 * 64 * 2),
    "bn_bwd_apply_pool_kernel": 256 * (160 * 180 * 64 * 2 * 2 + 80 * 180 * 64 * 2),
}

if __name__ == "__main__":
    out, merged = sys.argv[1], collections.defaultdict(dict)
    step_tot = collections.defaultdict(lambda: collections.defaultdict(float))
    for spec in sys.argv[2:]:
        tag, path = spec.split("=", 1)
        step = tag.split(":", 1)[0] if ":" in tag else None
        for kern, counters in read_counter_csv(path).items():
            for c, vals in counters.items():
                merged[kern][c] = sum(vals) / len(vals)
                merged[kern][c + "_launches"] = len(vals)
                if step and kern.startswith(("dfa::", "void dfa::")):
                    step_tot[step][c] += sum(vals)
    blob = {"source": "rocprofv3 --kernel-trace --pmc <one counter group per pass> -- python3 tools/gpu_prof_fwd.py / gpu_prof_train.py "
                      "(B=256, T=321, F=180); per-kernel means over the launches of each pass",
            "correction": "gfx950: FETCH_SIZE (KB) under-reports wide streaming reads by exactly 2x (MI355X_MICROARCH.md, HBM); "
                          "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024", "kernels": {}}
    for kern, c in sorted(merged.items()):
        if not kern.startswith(("dfa::", "void dfa::")):
            continue
        rec = {k: round(v, 1) for k, v in c.items()}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rec["hbm_bytes_per_launch"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
        for key, nbytes in ALGO.items():         # the most specific (longest) matching key wins
            if key in kern and len(key) >= len(rec.get("_k", "")):
                rec["algorithmic_bytes_per_launch"], rec["_k"] = nbytes, key
        rec.pop("_k", None)
        if "hbm_bytes_per_launch" in rec and "algorithmic_bytes_per_launch" in rec:
            rec["hbm_over_algorithmic"] = round(rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"], 3)
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and c.get("GRBM_GUI_ACTIVE", 0) > 0:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs issue MFMAs
            rec["mfma_busy_fraction"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0), 3)
        blob["kernels"][kern] = rec
    blob["steps"] = {}
    for step, tot in sorted(step_tot.items()):
        n = STEPS_IN_RUN.get(step)
        if n and "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
            rec = {"launches_of_the_loop": n, "hbm_bytes_per_step": int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / n),
                   "what": "sum over every dfa:: kernel launch of the profiled run (weight packing and reductions included) / loop count"}
            if step in STEP_ALGO:
                rec["algorithmic_bytes_per_step"] = STEP_ALGO[step]
                rec["hbm_over_algorithmic"] = round(rec["hbm_bytes_per_step"] / STEP_ALGO[step], 3)
            blob["steps"][step] = rec
    json.dump(blob, open(out, "w"), indent=1)
    print("wrote", out, len(blob["kernels"]), "kernels")
