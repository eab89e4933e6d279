"""python tools/layer_times_diff.py a.json b.json [batch]: per conv launch, time under setting a vs b, with the conv's shape."""
import json, sys
sys.path.insert(0, '.')
from eosvos_amd.topology import conv_infos
a, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
infos = conv_infos('resnet50')
ta = tb = tbest = 0.0
for k in a:
    if k not in b:
        continue
    ci, kind = map(int, k.split(':'))
    c = infos[ci]
    ta += a[k]; tb += b[k]; tbest += min(a[k], b[k])
    flag = '<' if b[k] < 0.97 * a[k] else ('>' if b[k] > 1.03 * a[k] else ' ')
    print(f"conv {ci:2d} {'fwd dgrad wgrad'.split()[kind]:5s} {c.name:38s} cin {c.cin:4d} cout {c.cout:4d} k {c.k} s {c.stride} d {c.dil}  a {a[k]:7.1f}  b {b[k]:7.1f} {flag}")
print(f'sum a {ta:.0f} us  b {tb:.0f} us  best-of {tbest:.0f} us')
