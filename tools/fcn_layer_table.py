"""Per-layer FCN timing table from a rocprofv3 kernel trace CSV of tools/time_fcn.py (batch 32)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'ivffcn' in r['Kernel_Name']]
starts = [i for i, r in enumerate(rows) if 'k_fcn_prep' in r['Kernel_Name']]
seq = rows[starts[-1]:]
BLOCKS = [(32,16,1,1,1),(16,24,6,2,1),(24,24,6,1,1),(24,32,6,2,1),(32,32,6,1,1),(32,32,6,1,1),(32,64,6,1,1),(64,64,6,1,2),(64,64,6,1,2),(64,64,6,1,2),(64,96,6,1,2),(96,96,6,1,2),(96,96,6,1,2),(96,160,6,1,2),(160,160,6,1,4),(160,160,6,1,4),(160,320,6,1,4)]
layers = [('prep', 0, (375*1242*3 + 3*512*512*4)), ('conv0', 2*27*32*256*256, (3*512*512 + 32*256*256)*4)]
H = 256
for inp, oup, t, s, d in BLOCKS:
    hid = inp * t
    if t != 1: layers.append(('pw %d->%d @%d' % (inp, hid, H), 2*inp*hid*H*H, (inp+hid)*H*H*4))
    Ho = H // s
    layers.append(('dw %d s%d d%d @%d' % (hid, s, d, H), 2*9*hid*Ho*Ho, (hid*H*H + hid*Ho*Ho)*4))
    H = Ho
    layers.append(('pw %d->%d @%d' % (hid, oup, H), 2*hid*oup*H*H, (hid+oup)*H*H*4))
layers += [('cbr 3x3 320->80', 2*9*320*80*64*64, (320+80)*4096*4), ('last', 2*80*4096, 81*4096*4), ('out', 0, 64*64*4 + 375*1242)]
B = 32; tot = 0
for r, (name, fl, by) in zip(seq, layers):
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; tot += us
    print("%-26s %-22s %8.1f us %6.1f TF/s %5.2f TB/s" % (name, r['Kernel_Name'].split('(')[0][-22:], us, fl*B/us/1e6, by*B/us/1e6))
print("total %.1f us for %d images" % (tot, B))
