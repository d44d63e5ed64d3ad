"""README.md's results table from the committed bench lines:  python3 tools/readme_table.py [tag]
(config | one call | three in flight | every step from HBM | fraction of the 8 TB/s roofline)"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
rows = [("cfgL", "**cfg-L**: 1 M segments, 1000 paths x 100 k steps (the benchmark's random walks)"),
        ("cfgL-chrom", "... paths that run along the graph, half of them on the reverse strand, a jump per hundred steps"),
        ("cfgL-uniform", "... ids without any runs (adversarial)"),
        ("cfgL-short", "... as 100 k paths of 1 k steps"), ("tiny-paths", "... as a million paths of a hundred steps"),
        ("cfgL-4paths", "... as four paths of 25 M steps (split paths)"),
        ("cfgL-4Mseg", "4 M segments"), ("cfgL-16Mseg", "16 M segments"), ("cfgL-64Mseg", "64 M segments (two ranges, two walks of the steps)"),
        ("hap-chr20", "one chromosome's worth: 4 M segments, 2000 in-order contigs, 200 M steps"),
        ("rep-chr20", "... whose walks go back over a few segments every 6400 steps"),
        ("cfgL-x16", "1.6 G steps on 1 M segments"), ("hap-16M", "16 M segments, 16 000 haplotype walks, 1.6 G steps"),
        ("chr-like", "16 M segments, ninety paths of ten million steps that wrap around the graph"),
        ("x16-16Mseg", "16 M segments, 16 000 random walks of 100 k steps"), ("x16-16Mseg-contigs", "... as 160 000 contigs of 10 k steps"),
        ("cfgS", "cfg-S: 10 k segments, 1 M steps (launch-bound: the atomic kernels)")]
print("| graph (synthetic, `bench.py --workload`) | one call, ms | three in flight, ms per call | steps/s | step-scan kernel: fraction of 8 TB/s warm (cold) | whole call: kernels' sum / as timed |")
print("|---|---|---|---|---|---|")
for w, what in rows:
    f = os.path.join(here, f"{tag}_bench_{w}.json")
    if not os.path.exists(f):
        continue
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    wc = r["whole_call"]
    tr = wc["timed_region"]
    fc = r.get("frac_cold")
    dom = r["kernel"].replace("<uniq>", "")
    print(f"| {what} (`{w}`) | {d.get('one_call_ms', tr['ms_per_step_one_call_in_flight']):.4f} | **{d['ms_per_step']:.4f}** | {d['value']:.2e} | `{dom}` {r['frac']:.3f}" +
          (f" ({fc:.3f})" if fc else "") + f" | {wc['frac']:.3f} / {tr['frac']:.3f} |")
