#!/bin/bash
# Device assembly of one translation unit with the product's flags, and a per-kernel summary (registers, LDS, scratch,
# instruction counts by class).  usage: tools/kernel_isa.sh ivf_kernels.hip [kernel-name-substring]   (asm lands in /tmp/isa/)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$1; PAT=${2:-}
mkdir -p /tmp/isa
OUT=/tmp/isa/$(basename $SRC .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -S --cuda-device-only -o $OUT $R/iv_slam_amd/csrc/$SRC $EXTRA 2>/dev/null
python3 - "$OUT" "$PAT" <<'PY'
import re, sys
txt = open(sys.argv[1]).read(); pat = sys.argv[2]
# kernel bodies: from "<name>:" label to ".Lfunc_end"
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat and pat not in name: continue
    ins = [l.split()[0] for l in body.splitlines() if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    def cnt(p): return sum(1 for i in ins if re.match(p, i))
    meta = re.search(r'\.amdhsa_kernel %s\b(.*?)\.end_amdhsa_kernel' % re.escape(name), txt, re.S)
    def mv(k):
        mm = re.search(r'\.set %s\.%s, (\S+)' % (re.escape(name), k), txt)
        if mm: return mm.group(1)
        mm = re.search(r'\.amdhsa_%s\s+(\S+)' % k, meta.group(1)) if meta else None
        return mm.group(1) if mm else '?'
    print("%s\n   insts %d  valu %d  salu %d  mfma %d  ds %d  vmem %d  branch %d | vgpr %s agpr %s sgpr %s lds %s scratch %s" % (
        name[:110], len(ins), cnt(r'v_(?!mfma)'), cnt(r's_(?!waitcnt|nop|barrier|cbranch|branch)'), cnt(r'v_mfma'), cnt(r'ds_'),
        cnt(r'(global|buffer|flat|scratch)_'), cnt(r's_c?branch'), mv('num_vgpr'), mv('num_agpr'), mv('numbered_sgpr'),
        mv('group_segment_fixed_size'), mv('private_seg_size')))
PY
