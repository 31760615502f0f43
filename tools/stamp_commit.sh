#!/bin/bash
# write the commit the working tree was cut from into .ivf_commit (travels with the gpurun snapshot, which has no .git): profiles/*_pmc_*.json record it
R=$(cd "$(dirname "$0")/.." && pwd)
c=$(git -C $R rev-parse --short HEAD 2>/dev/null || echo unknown)
[ -n "$(git -C $R status --porcelain 2>/dev/null | grep -v '^??')" ] && c="$c+dirty"
echo $c > $R/.ivf_commit; echo "stamped $c"
