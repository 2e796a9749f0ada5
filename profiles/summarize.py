#!/usr/bin/env python
"""Fold the rocprofv3 CSVs written by profiles/run_rocprof.sh into a committed summary.

    python profiles/summarize.py gpurun_out/prof profiles/r01_v3 [torus_div]  ->  profiles/r01_v3_rocprof.md
                                                                        profiles/traffic.json (read by bench.py)

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are collected in
separate --pmc passes, are in KiB-units of 1024 B, and on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e.
reads up to 2x low for wide coalesced streams: both the raw and the x2-corrected figure are reported; WRITE_SIZE is
calibrated here against the fine kernel, whose only HBM writes are its outputs (an exactly known byte count).
"""
import collections
import csv
import glob
import json
import os
import sys

OURS = {"mesh_raster_kernel": "mesh_fine", "area_list_kernel": "mesh_backward_areas", "mesh_backward": "mesh_backward", "points_raster": "points_fine",
        "bin_count": "bin_count", "bin_fill": "bin_fill", "bin_scan_offsets": "bin_scan_offsets",
        "bin_scan_rows": "bin_scan_rows", "bin_scan_small": "bin_scan_small", "bin_plan": "bin_plan", "gather_faces": "gather_face_verts",
        "scatter_face": "scatter_face_grads", "point_raster_kernel": "points_fine (register queues, naive launch)", "point_tile_sorted": "points_fine (tile-sorted kernel, LDS queues)",
        "point_sorted_kernel": "points_fine (sorted kernel, LDS queues)", "point_backward": "points_backward",
        "splat_backward": "points_composite_bwd (compositor + rasterizer backward, one kernel)", "splat_composite": "points_composite (pass)",
        "composite_fwd": "composite_fwd", "composite_bwd": "composite_bwd", "transform_verts": "transform_verts",
        "interp_fwd": "interp_fwd", "interp_bwd": "interp_bwd", "softmax_blend": "softmax_blend", "phong": "phong", "soft_phong": "soft_phong"}


def short(name):
    for k, v in OURS.items():
        if k in name:
            return v
    return None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    # argv[4] (optional): the profiled command, when it is not bench.py's (then profiles/traffic.json is left alone: bench.py reads
    # it as the counters of ITS dominant kernel)
    cmd_text = sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-dropin --no-reference-device"
    lines = ["# rocprofv3 summary (" + os.path.basename(dst) + ")", "",
             "Command: `" + cmd_text + "` under `rocprofv3` (profiles/run_rocprof.sh: `--kernel-trace --stats`, then one run per `--pmc` set), "
             "MI355X / gfx950.", ""]
    f = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
    if f:
        rows = list(csv.DictReader(open(f[0])))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        lines += ["## `--kernel-trace --stats`: per-kernel time (all launches incl. warm-up)", "",
                  "| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|"]
        for r in rows[:16]:
            nm = short(r["Name"]) or r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-70:]
            lines.append(f"| {nm} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                         f"{float(r['MaxNs'])/1e3:.1f} | {100*float(r['TotalDurationNs'])/tot:.2f} |")
        lines.append("")
    traffic = {}
    pmc = collections.defaultdict(dict)
    meta = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
        if not f:
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f[0])):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = dict(vgpr=r["VGPR_Count"], agpr=r.get("Accum_VGPR_Count", "0"), sgpr=r["SGPR_Count"],
                           lds=r["LDS_Block_Size"], scratch=r["Scratch_Size"], grid=r["Grid_Size"], wg=r["Workgroup_Size"])
        for k, cs in acc.items():
            for c, v in cs.items():
                pmc[k][c] = sum(v) / len(v)
    if pmc:
        lines += ["## PMC passes (per-launch averages; each `--pmc` set collected in its own run)", "",
                  "rocprofv3's `VGPR_Count` column reads HALF the wave64 allocation on gfx950; the compiler's figures per kernel (VGPRs, "
                  "AGPRs, scratch, spills, occupancy) are recorded at build time in `pytorch3d_amd/libp3d_amd.so.resources.json`.", ""]
        for k in sorted(pmc, key=lambda k: -pmc[k].get("SQ_WAVE_CYCLES", 0)):
            m = meta[k]
            lines += [f"### {k}", "",
                      f"grid {m['grid']} threads, workgroup {m['wg']}, VGPR {m['vgpr']} (+{m['agpr']} acc), SGPR {m['sgpr']}, "
                      f"LDS {m['lds']} B, scratch {m['scratch']} B/lane", "", "| counter | per launch |", "|---|---|"]
            for c in sorted(pmc[k]):
                lines.append(f"| {c} | {pmc[k][c]:.4g} |")
            c = pmc[k]
            if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
                fetch = c.get("FETCH_SIZE", 0.0) * 1024
                write = c.get("WRITE_SIZE", 0.0) * 1024
                traffic[k] = {"fetch_bytes_raw": fetch, "fetch_bytes_x2": 2 * fetch, "write_bytes": write,
                              "hbm_bytes": 2 * fetch + write}
                lines += ["", f"HBM traffic per launch: FETCH_SIZE {fetch/1e6:.1f} MB raw (x2 gfx950 correction: "
                          f"{2*fetch/1e6:.1f} MB), WRITE_SIZE {write/1e6:.1f} MB -> {(2*fetch+write)/1e6:.1f} MB"]
            if "SQ_WAVE_CYCLES" in c and "SQ_INSTS_VALU" in c:
                lines += ["", f"VALU wave-instructions per wave: {c['SQ_INSTS_VALU']/max(c.get('SQ_WAVES',1),1):.0f}; "
                          f"busy {100*c.get('SQ_ACTIVE_INST_VALU',0)/c['SQ_WAVE_CYCLES']:.1f}% / waiting "
                          f"{100*c.get('SQ_WAIT_ANY',0)/c['SQ_WAVE_CYCLES']:.1f}% / issue-stalled "
                          f"{100*c.get('SQ_WAIT_INST_ANY',0)/c['SQ_WAVE_CYCLES']:.1f}% of wave cycles"]
            lines.append("")
    with open(dst + "_rocprof.md", "w") as fh:
        fh.write("\n".join(lines) + "\n")
    if len(sys.argv) > 4:
        print("\n".join(lines))
        return
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "traffic.json"), "w") as fh:
        # the issue-side counters bench.py quotes beside the HBM roofline ("fp32 VALU issue rate and LDS bandwidth
        # for the fine stage, reported alongside"): wave-instructions per launch, LDS-active and bank-conflict cycles
        issue = {k: {c: pmc[k][c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                                            "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
                                            "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if c in pmc[k]} for k in pmc if "SQ_INSTS_VALU" in pmc[k]}
        json.dump({k: v["hbm_bytes"] for k, v in traffic.items()} | {"_detail": traffic, "_issue": issue, "_source": os.path.basename(dst),
                                                                    "_torus_div": float(sys.argv[3]) if len(sys.argv) > 3 else 1.0},
                  fh, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
