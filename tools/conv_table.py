"""Per-layer table of the last forward in a rocprofv3 kernel trace: time vs bytes/flops of each conv."""
import csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eioku_amd import weights as W
variant, batch, H = sys.argv[2] if len(sys.argv) > 2 else 'n', int(sys.argv[3]) if len(sys.argv) > 3 else 64, 640
tab = W.conv_table(variant, 80)
def level(name):
    m = re.match(r'model\.(\d+)', name); i = int(m.group(1))
    if i == 22: return 3 + int(name.split('.')[3])
    return {0:0,1:1,2:2,3:2,4:3,5:3,6:4,7:4,8:5,9:5,12:4,15:3,16:3,18:4,19:4,21:5}[i]
fs = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(fs[-1])) if ('k_conv' in r['Kernel_Name'] or 'stem_chain' in r['Kernel_Name']) and 'gather' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a launch whose kernel carries POST = true (5th template argument) also ran the NEXT layer (3x3 + 1x1 pair): walk the
# trace backwards from its end, one forward's worth of layers
def n_layers_of(r):
    if 'stem_chain' in r['Kernel_Name']: return 3
    if 'k_conv3x3_pair_rs' in r['Kernel_Name']: return 2  # 64-channel Bottleneck pair, weights in registers
    m = re.search(r'k_conv3x3_chain<([^>]*)>', r['Kernel_Name'])
    if m:  # <NF, DB, CAT, NF2>: CAT > 0 = the C2f's closing 1x1 ran in the launch too
        args = m.group(1).replace(' ', '').split(',')
        return 3 if len(args) > 2 and args[2] not in ('0', 'false') else 2
    return 2 if is_pair(r) else 1
def is_pair(r):
    if 'k_conv3x3_chain' in r['Kernel_Name']: return True  # Bottleneck: 3x3 -> 3x3 (+ x) in one launch
    m = re.search(r'k_conv3x3_persist<([^>]*)>', r['Kernel_Name'])
    args = m.group(1).replace(' ', '').split(',') if m else []
    return len(args) >= 5 and args[4] == 'true'  # <NF, S, NCH, DB, POST[, waves]>
pairs_layers = {'model.1.conv', 'model.3.conv'}
def take(n_layers):
    out, i, need = [], len(rows) - 1, n_layers
    while need > 0 and i >= 0:
        out.append(rows[i]); need -= n_layers_of(rows[i]); i -= 1
    out.reverse()
    return out
# detect(): the box branch's last 1x1 (model.22.cv2.l.2) is evaluated lazily by k_box_gather, and with few anchors
# passing so are its two 3x3 layers (k_conv3x3_gather): those rows have no conv launch of their own
lazy_names, last = set(), []
for pat in (r'cv2\.\d\.[012]', r'cv2\.\d\.2$', None):
    names = {n for n, *_ in tab if pat and re.search(pat, n)}
    cand = take(len(tab) - len(names))
    if cand and ('k_conv3x3_c8' in cand[0]['Kernel_Name'] or 'stem_chain' in cand[0]['Kernel_Name'] or pat is None):
        lazy_names, last = names, cand
        break
tot_t = tot_b = tot_f = 0
print(f"{'layer':28s} {'shape':22s} {'kernel':28s} {'grid':>9s} {'us':>7s} {'MB':>7s} {'GB/s':>6s} {'TF/s':>6s}")
it = iter(last)
skip = 0
for (name, cout, cin, k, s) in tab:
    lv = level(name); hin = H >> lv; hout = hin // s
    if name in lazy_names:
        print(f"{name:28s} {cin:4d}->{cout:4d} k{k}s{s} @{hout:3d} (lazy: evaluated only where decode needs it -- k_conv3x3_gather / k_box_gather)")
        continue
    if skip:  # ran inside the previous row's launch: its bytes are the output only, its time is already counted
        skip -= 1
        px_out = batch * hout * hout
        by = (px_out * cout * 2) if skip == 0 else 0; fl = 2.0 * px_out * cout * cin * k * k
        tot_b += by; tot_f += fl
        print(f"{name:28s} {cin:4d}->{cout:4d} k{k}s{s} @{hout:3d} (fused into the launch above; +{by/1e6:.1f} MB out, +{fl/1e9:.1f} GFLOP)")
        continue
    r = next(it)
    nl = n_layers_of(r)
    fused = nl > 1
    cin_eff = 8 if name == 'model.0.conv' else cin
    px_in, px_out = batch * hin * hin, batch * hout * hout
    obytes = 4 if re.search(r'cv[23]\.\d\.2$', name) else 2
    by = px_in * cin_eff * 2 + (0 if fused else px_out * cout * obytes)
    if name == 'model.0.conv' and 'stem_chain' in r['Kernel_Name']:
        by = px_in * 3
    elif name == 'model.0.conv' and 'k_conv3x3_c8' in r['Kernel_Name'] and not re.search(r'c8<\d+, \d+, 0>', r['Kernel_Name']):
        by = px_in * 3 + px_out * cout * obytes  # fused letterbox: the stem reads the BGR u8 frames
    skip = nl - 1
    fl = 2.0 * px_out * cout * cin * k * k
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    m_ = re.search(r'(k_conv\w+<[^>]*>)', r['Kernel_Name'])
    kn = m_.group(1) if m_ else ('k_conv3x3_pair_rs' if 'pair_rs' in r['Kernel_Name'] else 'k_conv_stem_chain')
    tot_t += us; tot_b += by; tot_f += fl
    print(f"{name:28s} {cin:4d}->{cout:4d} k{k}s{s} @{hout:3d} {kn:28s} {int(r['Grid_Size_X'])//256:5d}x{r['Grid_Size_Y']:>3s} {us:7.1f} {by/1e6:7.1f} {by/us/1e3:6.0f} {fl/us/1e6:6.1f}")
print(f"TOTAL {tot_t:.1f} us, {tot_b/1e9:.2f} GB -> {tot_b/tot_t/1e3:.0f} GB/s, {tot_f/1e9:.1f} GFLOP -> {tot_f/tot_t/1e6:.1f} TF/s")
