#!/bin/bash
# Per-layer launch-plan alternatives in ONE gpurun call: tools/plan_sweep.sh "name=BM,BN,KG,ks;..." ["..." ...]
# Prints, for the default plan and for each alternative, the lines of tools/layer_table.py that differ from the default.
cd "$(dirname "$0")/.."
python tools/layer_table.py > /tmp/lt_base.txt 2>/dev/null
grep "^total" /tmp/lt_base.txt
for plan in "$@"; do
  echo "== VNECT_PLAN=$plan"
  VNECT_PLAN="$plan" python tools/layer_table.py > /tmp/lt_alt.txt 2>/dev/null
  python - "$plan" <<'PY'
import sys
base = {l.split()[0]: l for l in open('/tmp/lt_base.txt') if l.strip() and not l.startswith(('layer', 'total', '{'))}
for l in open('/tmp/lt_alt.txt'):
    if not l.strip() or l.startswith(('layer', '{')): continue
    k = l.split()[0]
    if l.startswith('total'): print(l.strip()); continue
    if (k + '=') in sys.argv[1]:
        b = base[k].split(); a = l.split()
        print("%-32s %s x %s x ks%s %5s WGs %7s us  ->  %s x %s x ks%s %5s WGs %7s us" % (k, b[4], b[5], b[6], b[7], b[8], a[4], a[5], a[6], a[7], a[8]))
PY
done
