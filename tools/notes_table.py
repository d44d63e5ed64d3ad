"""The per-workload table of profiles/NOTES.md (R5.7, R6.9) from the committed bench lines, beside the round before:  python3 tools/notes_table.py [tag] [previous tag]"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
prev_tag = sys.argv[2] if len(sys.argv) > 2 else "r%02d" % (int(tag[1:]) - 1)
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
order = ("cfgL cfgL-chrom cfgL-uniform cfgL-short cfgL-medium chrom-10k chrom-1k tiny-paths hap-1k hap-10k hap-100 hap-chr20 rep-chr20 cfgL-fewlong cfgL-4paths "
         "cfgL-32k cfgL-100kseg cfgL-4Mseg cfgL-16Mseg cfgL-64Mseg cfgM cfgS cfgL-x16 hap-16M rep-16M hap-chr chr-like chr-like-2k chr-like-40M x16-16Mseg "
         "x16-16Mseg-chrom x16-16Mseg-contigs x16-16Mseg-fewlong").split()
def line_of(t, w):
    f = os.path.join(here, f"{t}_bench_{w}.json")
    if not os.path.exists(f):
        return None
    try:
        return json.loads(open(f).read().strip().splitlines()[-1])
    except (ValueError, IndexError):
        return None


for w in order:
    d = line_of(tag, w)
    if d is None:
        continue
    before = line_of(prev_tag, w)
    try:
        was = "-" if before is None else f"{before['roofline']['whole_call']['timed_region']['ms_per_step_one_call_in_flight']:.4f} / {before['ms_per_step']:.4f}"
    except (KeyError, TypeError):
        was = "-" if before is None else f"{before['ms_per_step']:.4f}"
    r = d["roofline"]
    wc = r["whole_call"]
    tr = wc["timed_region"]
    kk = " + ".join(f"{k.replace('<uniq>', '')} {v * 1e3:.0f}" if v * 1e3 >= 10 else f"{k.replace('<uniq>', '')} {v * 1e3:.1f}" for k, v in r["kernels_avg_ms"].items())
    fc = r.get("frac_cold")
    print(f"| `{w}` | {was} | {tr['ms_per_step_one_call_in_flight']:.4f} | **{d['ms_per_step']:.4f}** | {kk} | {r['frac']:.3f}" + (f" ({fc:.3f})" if fc else "") +
          f" | {wc['frac']:.3f} / {tr['frac']:.3f} |")
