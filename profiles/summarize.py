#!/usr/bin/env python3
"""Turns gpurun_out/<tag>_prof (written by profiles/collect.sh) into the files kept under profiles/:
    <tag>_bench_n1.json                    the bench line
    <tag>_bench_n1_kernel_stats.csv        rocprofv3 --kernel-trace --stats summary of bench.py
    <tag>_bench_n1_pmc_summary.txt         mean counter values per kernel (FETCH_SIZE, WRITE_SIZE, TCC, SQ passes)
    <tag>_bench_dense_<path>_kernel_stats.csv, <tag>_dense_pmc_summary.txt, <tag>_bnf_pmc_summary.txt
    <tag>_bench_bnf.json, <tag>_bnf_autoencoder.txt, <tag>_bnf_autoencoder_kernel_stats.csv   brain-network step and autoencoder
    traffic.json                           HBM-side bytes per launch of the three sparse kernels, with the kernel symbols
                                           they were counted on (bench.py attaches a figure only to the same symbol)
usage: python profiles/summarize.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def kernel_key(name):
    """'void gatmi::k<1, 2>(gatmi::Args)' -> 'k<1, 2>'"""
    name = name.strip()
    if name.startswith("void "):
        name = name[5:]
    depth, end = 0, len(name)
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            end = i
            break
    name = name[:end]
    return name[len("gatmi::"):] if name.startswith("gatmi::") else name


def counters(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        # gpurun merges a call's outputs into the local directory without removing what an earlier call left there:
        # of several runs of one pass only the newest file counts
        found = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
        for f in found[-1:]:
            for r in csv.DictReader(open(f)):
                if "gatmi" not in r["Kernel_Name"]:
                    continue
                acc[kernel_key(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def write_summary(acc, path, header):
    with open(path, "w") as out:
        out.write(header + "\n")
        for k, cs in sorted(acc.items()):
            out.write(k + "\n")
            for c, v in sorted(cs.items()):
                out.write("    %-26s mean %.6g  (n=%d)\n" % (c, sum(v) / len(v), len(v)))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
                busy = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"])
                gui = sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"])
                # GRBM_GUI_ACTIVE sums the 8 XCDs' clocks; MFMA-busy cycles sum over the 1024 SIMDs
                out.write("    %-26s %.1f %%  (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8))\n"
                          % ("matrix-core busy", 100.0 * busy / (1024.0 * gui / 8.0)))


def copy_stats(src_dir, dst):
    files = sorted(glob.glob(os.path.join(src_dir, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if files:
        shutil.copy(files[-1], dst)                # the newest run of the part (see counters)
        return True
    return False


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag + "_prof")
    j = os.path.join(src, "bench_n1.json")
    if os.path.exists(j):
        line = [l for l in open(j).read().splitlines() if l.startswith("{")][-1]
        open(os.path.join(HERE, tag + "_bench_n1.json"), "w").write(line + "\n")
    copy_stats(os.path.join(src, "stats_bench"), os.path.join(HERE, tag + "_bench_n1_kernel_stats.csv"))
    for mp in ("lists", "dense", "csr", "prebuilt", "lists_drop", "prebuilt_drop"):
        copy_stats(os.path.join(src, "stats_dense_" + mp), os.path.join(HERE, "%s_bench_dense_%s_kernel_stats.csv" % (tag, mp)))
    copy_stats(os.path.join(src, "stats_bnf_ae"), os.path.join(HERE, tag + "_bnf_autoencoder_kernel_stats.csv"))
    for name in ("bench_dense.jsonl", "bench_two_layer.jsonl", "bench_bnf.json", "bnf_autoencoder.txt"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(HERE, "%s_%s" % (tag, name)))
    sparse = counters([os.path.join(src, "pmc_" + n) for n in ("fetch", "write", "tcc", "sq")])
    write_summary(sparse, os.path.join(HERE, tag + "_bench_n1_pmc_summary.txt"),
                  "# bench.py --steps 2 --warmup 1 --no-cpu-baseline; one rocprofv3 --pmc pass per counter group; FETCH_SIZE / WRITE_SIZE in KiB")
    dense = counters(glob.glob(os.path.join(src, "pmc_dense_*")))
    write_summary(dense, os.path.join(HERE, tag + "_dense_pmc_summary.txt"),
                  "# benchmarks/bench_dense.py --mask-path lists|dense|prebuilt --steps 2 --warmup 1 (configs[2]: 4096 graphs x 264 nodes)")
    bnf = counters([os.path.join(src, "pmc_bnf_sq")])
    write_summary(bnf, os.path.join(HERE, tag + "_bnf_pmc_summary.txt"), "# benchmarks/bench_bnf.py (brain-network step and inner-product decoder)")
    # HBM-side traffic per launch of the sparse kernels: 2 x FETCH_SIZE (gfx950 tallies the 128-byte requests of
    # 16-byte-per-lane reads at 64 bytes) + WRITE_SIZE, both in KiB
    entry_of = {"gat_fwd_kernel": "gatmi_gat_fwd", "gat_bwd_csc_kernel": "gatmi_gat_bwd_csc", "gat_bwd_node_kernel": "gatmi_gat_bwd_node"}
    kernels, symbols = {}, {}
    for k, cs in sparse.items():
        base = k.split("<")[0]
        if base in entry_of and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
            w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            kernels[entry_of[base]] = int(round((2.0 * f + w) * 1024.0))
            symbols[entry_of[base]] = k
    if kernels and os.path.exists(os.path.join(HERE, tag + "_bench_n1.json")):
        bench = json.loads(open(os.path.join(HERE, tag + "_bench_n1.json")).read())
        edges = int(bench["config"]["workload"].split(" nodes, ")[1].split(" edge entries")[0])
        json.dump({"_note": "HBM-side bytes per launch from separate rocprofv3 --pmc passes on the default bench.py workload: "
                            "2 x FETCH_SIZE KiB (gfx950 tallies the 128-byte requests of 16-byte-per-lane reads at 64 bytes) + WRITE_SIZE KiB",
                   "source": "profiles/%s_bench_n1_pmc_summary.txt" % tag, "workload_edges": edges,
                   "kernels": kernels, "kernel_symbols": symbols}, open(os.path.join(HERE, "traffic.json"), "w"), indent=1)
    print("summaries written for", tag)


if __name__ == "__main__":
    main()
