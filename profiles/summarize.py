#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof/...) into the small summaries committed under profiles/.

    python profiles/summarize.py gpurun_out/prof r01

kt/        : rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py ...
pmc_fetch/ : rocprofv3 --kernel-trace --pmc FETCH_SIZE ...      (separate pass, no other trace domains)
pmc_write/ : rocprofv3 --kernel-trace --pmc WRITE_SIZE ...
sq_a/, sq_b/ : rocprofv3 --kernel-trace --pmc <8 SQ counters> ... (tools/profile_round.sh)
Writes <tag>_sq.json (matrix-pipe busy, LDS bank conflicts, instruction mix per hand-written kernel),
<tag>_kernel_stats.csv (rocprof's own per-kernel stats for the whole process, MIOpen's first-call
solver search included), <tag>_step_breakdown.txt (the fastest bench step, from the kernel trace) and
<tag>_pmc.json (+ roofline_traffic.json, which bench.py reads for roofline.traffic).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

OURS = ("bin_head_kernel", "pixel_dot_kernel", "patch_embed_partial_kernel", "patch_embed_reduce_kernel",
        "attention_kernel", "linear_kernel", "linear_stream_kernel", "layernorm_kernel", "ffn_fused_kernel",
        "conv_igemm_kernel", "pointwise_kernel", "pointwise_smallk_kernel", "depthwise_kernel", "depthwise_nhwc_kernel",
        "channel_sum_kernel", "channel_mean_finish_kernel", "se_hidden_kernel", "se_gate_kernel",
        "conv_split_dma_kernel", "upsample_concat_split_kernel", "pw_rows_kernel", "pw_tile_kernel", "pw_stream_kernel",
        "bin_head_split_kernel", "bin_head_split3_kernel", "bin_head_combine_kernel", "wino_input_kernel", "wino_output_kernel", "conv_exact_kernel", "tap_interp_kernel", "lin3_kernel", "ffn3_kernel", "ffn3_finish_kernel", "layer_tail3_kernel", "pack3_kernel", "cross_attn_fused_kernel", "depth_metrics_partial_kernel", "depth_metrics_finish_kernel", "dw_slide_kernel", "se_hidden_partials_kernel", "se_gate_hid_kernel", "stem_conv_kernel",
        "mbconv_expand_dw_kernel", "pos_sample_kernel", "pw_big_kernel", "pw_hl_kernel", "se_gate_weights_kernel", "xattn_kv3_kernel", "xattn_main3_kernel", "encoder_stack_kernel", "upsample_concat_split8_kernel", "upsample_concat_split_2x2_kernel", "conv_splitk_finish_kernel", "ffn_finish_kernel", "upsample_concat_split_lds_kernel",
        "xattn_main_h2_kernel", "xattn_kv_h2_kernel", "pack_h2_kernel", "bin_head_h2_kernel", "conv_few_kernel", "attention_h2_kernel", "layer_tail_h2_kernel", "wino43_input_kernel", "wino43_output_kernel")


def is_step_end(name):
    """The launch that ends a bench step's forward: the bin head (its merge pass when it runs in two halves)."""
    return "bin_head_combine_kernel" in name or "bin_head_kernel<" in name or "bin_head_split_kernel" in name or "bin_head_h2_kernel" in name


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")[:110]


# kernels that run several problem shapes per step: one row per launch grid (the grid identifies the shape)
PER_SHAPE = ("tap_interp_kernel", "conv_split_dma_kernel", "conv_splitk_finish_kernel", "pw_tile_kernel", "pw_rows_kernel", "pw_stream_kernel", "pw_hl_kernel",
             "pw_big_kernel", "upsample_concat_split", "dw_slide_kernel", "mbconv_expand_dw_kernel")


def grid_of(r):
    if "Grid_Size_X" in r:
        return int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]), \
            int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    return int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0)


def shape_key(r):
    k = short(r["Kernel_Name"])
    if any(p in k for p in PER_SHAPE):
        g, w = grid_of(r)
        return f"{k.split('(')[0][:84]} [wgs={g // max(w, 1)}x{w}]"
    return k


SQ_KERNELS = ("tap_interp_kernel", "layer_tail3_kernel", "lin3_kernel", "conv_split_dma_kernel", "wino_input_kernel", "wino_output_kernel", "bin_head_split3_kernel", "bin_head_combine_kernel", "pw_tile_kernel", "pw_big_kernel", "pw_rows_kernel", "patch_embed_partial_kernel", "bin_head_kernel",
              "bin_head_split_kernel", "attention_kernel", "cross_attn_fused_kernel", "ffn_fused_kernel", "linear_stream_kernel",
              "dw_slide_kernel", "mbconv_expand_dw_kernel", "upsample_concat_split_lds_kernel", "encoder_stack_kernel", "pw_hl_kernel",
              "xattn_main3_kernel", "xattn_kv3_kernel", "se_gate_weights_kernel", "stem_conv_kernel", "se_fused_small_kernel", "pw_stream_kernel", "xattn_main_h2_kernel", "xattn_kv_h2_kernel", "bin_head_h2_kernel", "attention_h2_kernel", "layer_tail_h2_kernel")
N_SIMD = 1024            # 256 CUs x 4 SIMDs


def sq_summary(src, newest):
    """Per hand-written kernel (and per launch grid): counter sums over the launches of the run's LAST bench step, the
    launch duration from the same pass, and the derived ratios the north star asks for.  Units (MI355X_MICROARCH.md):
    SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over all SIMDs; SQ_*_CYCLES / SQ_WAIT / SQ_ACTIVE_INST
    count quad-cycles; SQ_LDS_BANK_CONFLICT = extra LDS cycles."""
    out = {}
    for d in ("sq_a", "sq_b"):
        fs = newest(os.path.join(src, d, "*", "*_counter_collection.csv"))
        if not fs:
            continue
        rows = [r for r in csv.DictReader(open(fs[0])) if any(k in r["Kernel_Name"] for k in SQ_KERNELS)]
        if not rows:
            continue
        # last step = launches after the second-to-last bin-head dispatch
        heads = sorted({int(r["Dispatch_Id"]) for r in rows if is_step_end(r["Kernel_Name"])})
        lo = heads[-2] if len(heads) >= 2 else 0
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        seen = collections.defaultdict(set)
        for r in rows:
            did = int(r["Dispatch_Id"])
            if did <= lo:
                continue
            key = shape_key(r)
            per[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if did not in seen[key]:
                seen[key].add(did)
                per[key]["_ns_" + d] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                per[key]["_launches_" + d] += 1
        for k, v in per.items():
            out.setdefault(k, {}).update(v)
    res = {}
    for k, v in out.items():
        e = {c: int(x) for c, x in v.items() if not c.startswith("_")}
        la = max(v.get("_launches_sq_a", 0), v.get("_launches_sq_b", 0))
        e["launches"] = int(la)
        ns = v.get("_ns_sq_a") or v.get("_ns_sq_b") or 0
        e["total_us_under_pmc"] = round(ns / 1e3, 1)
        if v.get("SQ_VALU_MFMA_BUSY_CYCLES") and v.get("_ns_sq_a"):
            e["mfma_busy_frac_at_2.1GHz"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["_ns_sq_a"] * 2.1 * N_SIMD), 4)
        if v.get("SQ_VALU_MFMA_BUSY_CYCLES") and v.get("SQ_BUSY_CYCLES"):
            e["mfma_busy_over_sq_busy"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_BUSY_CYCLES"], 4)
        if v.get("SQ_INSTS_MFMA"):
            e["valu_per_mfma"] = round(v.get("SQ_INSTS_VALU", 0) / v["SQ_INSTS_MFMA"], 2)
            e["salu_per_mfma"] = round(v.get("SQ_INSTS_SALU", 0) / v["SQ_INSTS_MFMA"], 2)
        if v.get("SQ_WAVE_CYCLES"):
            e["wait_inst_frac"] = round(v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"], 4)
            e["lds_wait_frac"] = round(v.get("SQ_WAIT_INST_LDS", 0) / v["SQ_WAVE_CYCLES"], 4)
        res[k] = e
    return res


def main(src, tag):
    out = os.path.dirname(os.path.abspath(__file__))
    def newest(pattern):
        """gpurun merges every call's output into the same local directory: take the most recent run's file"""
        return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]

    ks = newest(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
    ksd = newest(os.path.join(src, "kt_default", "*", "*_kernel_stats.csv"))     # the default (pipelined) command
    if ksd:
        shutil.copy(ksd[0], os.path.join(out, f"{tag}_kernel_stats_default_inflight.csv"))
    kt = newest(os.path.join(src, "kt", "*", "*_kernel_trace.csv"))
    if kt:
        rows = list(csv.DictReader(open(kt[0])))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        idx = [i for i, r in enumerate(rows) if is_step_end(r["Kernel_Name"])]
        # a bench step = the launches between two bin-head launches; take the step with the shortest wall time (the
        # first steps still load code objects lazily and the last one is disturbed by the profiler's buffer flush)
        steps = [rows[a + 1: b + 1] for a, b in zip(idx[:-1], idx[1:])]
        step = min(steps, key=lambda st: int(st[-1]["End_Timestamp"]) - int(st[0]["Start_Timestamp"]))
        t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
        agg = collections.defaultdict(lambda: [0, 0])
        for r in step:
            a = agg[shape_key(r)]
            a[0] += 1
            a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        tot = sum(v[1] for v in agg.values())
        ours = sum(v[1] for k, v in agg.items() if any(o in k for o in OURS))
        with open(os.path.join(out, f"{tag}_step_breakdown.txt"), "w") as f:
            f.write(f"# fastest bench step of {len(steps)} in {os.path.basename(kt[0])}: wall {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels, "
                    f"sum of kernel time {tot / 1e6:.3f} ms, hand-written kernels {ours / 1e6:.3f} ms\n")
            f.write(f"{'kernel':110s} {'n':>5s} {'total_ms':>9s} {'avg_us':>9s} {'pct':>6s}\n")
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{k:110s} {v[0]:5d} {v[1] / 1e6:9.3f} {v[1] / v[0] / 1e3:9.1f} {100 * v[1] / tot:6.2f}\n")
    pmc, series, grids = {}, {}, {}
    for d, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        fs = newest(os.path.join(src, d, "*", "*_counter_collection.csv"))
        if not fs:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] == cname and any(o in r["Kernel_Name"] for o in OURS):
                agg[short(r["Kernel_Name"]).split("(")[0]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
                grids.setdefault(short(r["Kernel_Name"]).split("(")[0], {}).setdefault(cname, {})[int(r["Dispatch_Id"])] = int(r.get("Grid_Size", 0) or 0)
        for k, byd in agg.items():
            v = [byd[i] for i in sorted(byd)]            # launches in dispatch order (same order in both passes)
            series.setdefault(k, {})[cname] = v
            pmc.setdefault(k, {})[cname + "_KB_max"] = max(v)
            pmc[k][cname + "_KB_mean"] = sum(v) / len(v)
            pmc[k]["launches"] = len(v)
    if pmc:
        json.dump(pmc, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
        traffic = {}
        for key, prefixes in (("bin_head", ("bin_head_h2_kernel", "bin_head_split3_kernel", "bin_head_split_kernel", "bin_head_kernel")), ("patch_embed", ("patch_embed_partial_kernel",)),
                              ("conv3x3", ("conv_split_dma_kernel", "conv_igemm_kernel"))):
            names = [n for n in series if n.startswith(prefixes) and len(series[n]) == 2
                     and len(series[n]["FETCH_SIZE"]) == len(series[n]["WRITE_SIZE"])]
            if names:
                f, w = series[names[0]]["FETCH_SIZE"], series[names[0]]["WRITE_SIZE"]
                # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> x2; WRITE_SIZE exact.
                # The two counters come from separate passes of the same deterministic launch sequence: pair launch i
                # with launch i.  The roofline kernel of bench.py is the launch that moves the most bytes; report the
                # MEDIAN over the launches within 5 % of that maximum (= the same shape on other steps), so one
                # cold-cache launch does not set the number.
                # The convolution kernel runs fourteen shapes per step.  bench.py's roofline launch is the direct 128 -> 128
                # convolution at half resolution, three launches per step on ONE grid (with the last stage's skip-part
                # convolution, a much lighter launch, on the same grid): take the grid that moves the most bytes in total,
                # drop its launches below 60 % of the heaviest, and report the MEAN of the rest -- bench.py's event timing is
                # the mean over the same three launches (one writes fp32, one the split layout, one both).
                g = grids.get(names[0], {}).get("FETCH_SIZE", {})
                gl = [g[i] for i in sorted(g)]
                tot = [2 * a + b for a, b in zip(f, w)]
                if key == "conv3x3" and len(gl) == len(f) and max(gl) > 0:
                    by_grid = collections.defaultdict(float)
                    for i, t in enumerate(tot):
                        by_grid[gl[i]] += t
                    best = max(by_grid, key=by_grid.get)
                    sel = [tot[i] for i in range(len(tot)) if gl[i] == best]
                    sel = [t for t in sel if t >= 0.6 * max(sel)]
                    traffic[key] = int(sum(sel) / len(sel) * 1024)
                    continue
                tot = sorted(tot)
                top = [t for t in tot if t >= 0.95 * tot[-1]]
                traffic[key] = int(top[len(top) // 2] * 1024)
        json.dump(traffic, open(os.path.join(out, "roofline_traffic.json"), "w"), indent=1, sort_keys=True)
    sq = sq_summary(src, newest)
    if sq:
        json.dump(sq, open(os.path.join(out, f"{tag}_sq.json"), "w"), indent=1, sort_keys=True)
    print("wrote summaries to", out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
