#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
V=${VARIANT:-sdiag}
O=gpurun_out/ab_${V}.txt
: > $O
cp $C/libapgp.so /tmp/ship.so
for v in ship $V; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== bits $v" >> $O
    timeout 300 python tools/ab_bits.py 2>&1 | grep -E "N=|Error|error" >> $O
done
for v in ship $V ship $V; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== $v" >> $O
    timeout 600 python tools/sweep_shapes.py --ring ${MODES:---inverse-only} 2>&1 | grep -E "N=" >> $O
    timeout 600 python tools/sweep_shapes.py --dsweep ${MODES:---inverse-only} 2>&1 | grep -E "N=" >> $O
done
cp /tmp/ship.so $C/libapgp.so
VARIANT=$V python3 - <<'PY'
import re,collections,os
d=collections.OrderedDict(); cur=None
V=os.environ.get("VARIANT","img")
for l in open('gpurun_out/ab_%s.txt'%V):
    if l.startswith('== bits'): cur=None; print(l.strip()); continue
    if l.startswith('=='): cur=l.split()[1]; continue
    if cur is None: print(l.strip()[:100]); continue
    m=re.match(r'N=\s*(\d+) D=(\d+) M=\s*(\d+) \w+\s+(\w+)\s+([\d.]+) ms',l)
    if m: d.setdefault(m.groups()[:4],{}).setdefault(cur,[]).append(float(m.group(5)))
for k,v in d.items():
    s=sum(v['ship'])/len(v['ship']); p=sum(v[V])/len(v[V])
    print("N=%s D=%s M=%s %s ship %.3f %s %.3f  %+.1f%%"%(k[0],k[1],k[2],k[3],s,V,p,(s/p-1)*100))
PY
