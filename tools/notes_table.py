"""The per-workload table of profiles/NOTES.md (R5.7) from the committed bench lines:  python3 tools/notes_table.py [tag]"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
order = ("cfgL cfgL-chrom cfgL-uniform cfgL-short cfgL-medium chrom-10k chrom-1k tiny-paths hap-1k hap-10k hap-100 hap-chr20 rep-chr20 cfgL-fewlong cfgL-4paths "
         "cfgL-32k cfgL-100kseg cfgL-4Mseg cfgL-16Mseg cfgL-64Mseg cfgM cfgS cfgL-x16 hap-16M rep-16M hap-chr chr-like chr-like-2k chr-like-40M x16-16Mseg "
         "x16-16Mseg-chrom x16-16Mseg-contigs x16-16Mseg-fewlong").split()
r4 = {"cfgL": 0.13906, "cfgL-chrom": 0.19999, "cfgL-uniform": 0.34865, "cfgL-short": 0.1902, "cfgL-medium": 0.19927, "chrom-10k": 0.27208, "chrom-1k": 0.23958,
      "tiny-paths": 0.37728, "cfgL-fewlong": 0.146, "cfgL-4paths": 0.14025, "cfgL-32k": 0.15392, "cfgL-100kseg": 0.15042, "cfgL-4Mseg": 0.18758, "cfgL-16Mseg": 0.28679,
      "cfgL-64Mseg": 0.6985, "cfgM": 0.05047, "cfgS": 0.01967, "cfgL-x16": 2.00393, "hap-16M": 2.77308, "hap-chr": 1.81311, "chr-like": 2.30243, "chr-like-2k": 2.23795,
      "chr-like-40M": 7.18176, "x16-16Mseg": 3.79818, "x16-16Mseg-chrom": 4.81713, "x16-16Mseg-contigs": 7.4352, "x16-16Mseg-fewlong": 3.5846}
for w in order:
    f = os.path.join(here, f"{tag}_bench_{w}.json")
    if not os.path.exists(f):
        continue
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    wc = r["whole_call"]
    tr = wc["timed_region"]
    kk = " + ".join(f"{k.replace('<uniq>', '')} {v * 1e3:.0f}" if v * 1e3 >= 10 else f"{k.replace('<uniq>', '')} {v * 1e3:.1f}" for k, v in r["kernels_avg_ms"].items())
    fc = r.get("frac_cold")
    print(f"| `{w}` | {r4.get(w, '-')} | {tr['ms_per_step_one_call_in_flight']:.4f} | **{d['ms_per_step']:.4f}** | {kk} | {r['frac']:.3f}" + (f" ({fc:.3f})" if fc else "") +
          f" | {wc['frac']:.3f} / {tr['frac']:.3f} |")
