#!/usr/bin/env python3
"""Roofline table of the headline solve from committed evidence: per-kernel average durations out of a rocprofv3
`--kernel-trace --stats` csv, algorithmic bytes / flops per launch from DESIGN.md section 4's formulas, HBM traffic from
profiles/hbm_traffic.json.  usage: roofline_report.py [kernel_stats.csv] [M] [nnz] [n]  (defaults: the round's headline files)"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05", "headline_kernel_stats.csv")
M = float(sys.argv[2]) if len(sys.argv) > 2 else 1e7
nnz = float(sys.argv[3]) if len(sys.argv) > 3 else 5e7
n = int(sys.argv[4]) if len(sys.argv) > 4 else 200
HBM, MFMA = 8000.0, 78.6  # GB/s, TFLOP/s (MI355X_MICROARCH.md: HBM3E ~8 TB/s, dense FP64 MFMA 78.6 TF)

avg_j = (n - 1) / 2.0  # mean number of rows already in the basis over a run of n steps
classes = {  # kernel name prefix -> (label, algorithmic bytes per launch, flops per launch, bound)
    "k_update_slice<true": ("update  V[j] = 2 v - sum c_i V_i", 8 * avg_j * M + 16 * M, 2 * (avg_j + 1) * M, "hbm"),
    "k_qtw_mfma4<2": ("Q.w     c = V r (4x4x4 MFMA)", 8 * avg_j * M + 8 * M, 2 * (avg_j + 1) * M, "hbm"),  # fused-norm mode: reads r, writes nothing
    "k_spmv_fixed": ("SpMV    r = A v_j, alpha partials", 12 * nnz + 4 * (M + 1) + 16 * M, 2 * nnz, "hbm"),
    "k_three_term": ("3-term  r = r - a v_j - b v_{j-1}", 32 * M, 6 * M, "hbm"),
    "k_gemm_tn_sreg": ("Ritz    Y = V S (FP64 MFMA)", 16 * n * M + 8 * n * n, 2 * M * n * n, "mfma"),
    "k_gram_sym": ("Gram    G = Y^T Y, symmetric half", 8 * n * M, M * n * (n + 1), "mfma"),
}
traffic = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get("lap2d_5pt_M1e7_k200", {})
tkey = {"k_update_slice<true": "update", "k_qtw_mfma4<2": "qtw", "k_spmv_fixed": "spmv", "k_three_term": "three_term"}
print(f"source: {os.path.relpath(stats, ROOT)}  (M = {M:.0f}, nnz = {nnz:.0f}, n = {n})")
print(f"{'kernel class':40s} {'calls':>6s} {'avg us':>10s} {'alg GB':>8s} {'GB/s | TF/s':>12s} {'frac':>6s} {'PMC traffic / alg':>18s}")
for row in csv.DictReader(open(stats)):
    name = row["Name"].replace("void ", "").replace("lz::", "")
    for pre, (label, by, fl, bound) in classes.items():
        if name.startswith(pre):
            us = float(row["AverageNs"]) / 1e3
            if bound == "hbm":
                rate = by / us / 1e3
                frac = rate / HBM
                tr = traffic.get(tkey.get(pre, ""), None)
                extra = f"{tr / by:.3f}" if tr else "-"
                print(f"{label:40s} {row['Calls']:>6s} {us:10.1f} {by / 1e9:8.3f} {rate:9.0f} GB/s {frac:6.3f} {extra:>18s}")
            else:
                rate = fl / us / 1e6
                print(f"{label:40s} {row['Calls']:>6s} {us:10.1f} {by / 1e9:8.3f} {rate:9.1f} TF/s {rate / MFMA:6.3f} {'-':>18s}")
