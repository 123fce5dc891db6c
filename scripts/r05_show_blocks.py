#!/usr/bin/env python3
"""What a bench.py line says about balance: per block the rate, max / mean of the subdomains' particles, the re-cuts made."""
import json
import sys

for path in sys.argv[1:]:
    lines = [ln for ln in open(path) if ln.startswith("{")]
    if not lines:
        print(path, "no JSON line")
        continue
    d = json.loads(lines[-1])
    print(f"== {path}: {d['value']:.3e} {d['unit']}, {d['ms_per_step']:.4f} ms/step, link {d['config'].get('link', '')[:60]}")
    if d["config"].get("overlap"):
        print("   overlap:", d["config"]["overlap"])
    blocks = {"top level": d}
    for k in ("laser_plasma_interaction", "colliding_beams"):
        b = d.get(k) or d["config"].get(k)
        if isinstance(b, dict):
            blocks[k] = b
    for name, b in blocks.items():
        pr = b.get("per_rank") or b.get("config", {}).get("per_rank")
        print(f"   {name}: {b.get('value', 0):.3e}, {b.get('ms_per_step', 0):.4f} ms/step, link {b.get('link')}, imbalance max/mean "
              f"{b.get('imbalance_max_over_mean')}, recut {b.get('recut')}")
        if name == "colliding_beams":
            print("      subdomains holding a tenth of their share of the electrons:", b.get("subdomains_holding_a_tenth_of_their_share_of_the_electrons"))
        for p in pr or []:
            print("      ", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in p.items()})
