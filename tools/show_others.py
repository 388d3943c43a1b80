#!/usr/bin/env python3
"""Print the `others` table of a bench line: tools/show_others.py FILE.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
print("value", d["value"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
for e in d.get("others", []):
    print(f'{e["kernel_ms"]:9.4f} ms  frac {e["frac"]:.4f}  {e["config"][:90]}')
