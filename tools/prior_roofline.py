"""Bytes, time and fraction of the HBM roof for the kernels of the plane-prior path (cost evaluation, refinement, plane fit,
raster, sparse matcher, resampling) from a rocprofv3 kernel trace of `tools/bench_pipeline.py --single 2` (tools/prof_pipeline.sh).
Algorithmic bytes per unit: what the kernel must read and write once per point / pixel / triangle / candidate window, not what the
caches make of it.   usage: prior_roofline.py <kernel_stats.csv> [n_high_gradient n_support n_triangles n_left n_right]"""
import csv
import sys

ROWS, COLS = 368, 1232
n_hg, n_sup, n_tri, n_left, n_right = (int(v) for v in sys.argv[2:7]) if len(sys.argv) >= 7 else (285446, 27240, 54300, 52000, 160000)
px = ROWS * COLS
HBM = 8.0e12
# name prefix -> (unit count per launch, bytes per unit, what a unit is)
MODEL = {
    "cost_eval_kernel": (n_hg, 8 + 4 + 4 + 4 + 1, "point: xy 8 B, prior 4 B, two census words 8 B, cost 1 B"),
    "refine_points_kernel": (n_hg, 8 + 4 + 1 + 1 + 1 + 4, "point: xy 8 B, prior 4 B, cost 1 B, final cost read + write 2 B, final disparity 4 B"),
    "refine_cells_kernel": (2 * (ROWS // 64 + 1) * (COLS // 64 + 1), 8 + 8 + 4 + 1 + 8 + 1, "grid cell: two 64-bit keys, the winner's xy, disparity, cost"),
    "plane_fit_kernel": (n_tri, 24 + 12 + 12, "triangle: six corner coordinates 24 B, three disparities 12 B, plane 12 B"),
    "raster_owner_kernel": (px, 4, "pixel: one 4-byte owner (atomic minimum of the triangle index), plus 24 B per triangle"),
    "raster_eval_kernel": (px, 4 + 4, "pixel: owner 4 B read, prior 4 B written (plane 12 B per triangle through the cache)"),
    "sparse_calc_costs": (n_left, (40 + 1) * 30 * 4, "left feature: its window of 30 census words once, about 40 candidate windows of 30 words (cache traffic: the words of neighbouring windows overlap)"),
    "sparse_consistency_kernel": (n_sup, (70 + 1) * 30 * 4, "matched feature: its right window once, 70 left windows of 30 census words (cache traffic)"),
    "support_resampling_kernel": ((ROWS // 64 + 1) * (COLS // 64 + 1), 70 * 2 * 121 * 4, "grid cell with an invalid match: 70 disparities x 11 x 11 census words on either side (cache traffic)"),
    "census5x5_tile_kernel": (px, 1 + 4, "pixel: 1 B in, 4 B out"),
    "exfast_score_kernel": (px, 1 + 2, "pixel: 1 B in, 2-byte score out"),
}
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-34s %6s %9s %12s %9s %8s  %s" % ("kernel", "calls", "avg us", "MB / launch", "GB/s", "of 8 TB/s", "unit"))
for r in rows:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    for key, (units, per, what) in MODEL.items():
        if name.startswith(key):
            us = float(r["AverageNs"]) / 1e3
            b = units * per
            frac = "   cache" if "cache traffic" in what else "%8.3f" % (b / (us * 1e-6) / HBM)      # window gathers are served by L1/L2, not HBM
            print("%-34s %6s %9.1f %12.2f %9.0f %s  %s" % (key, r["Calls"], us, b / 1e6, b / us / 1e3, frac, what))
