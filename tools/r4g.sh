set -e
mkdir -p gpurun_out/r4g
python tools/prof_shapes.py > gpurun_out/r4g/s_default.txt 2>&1
DM_GEMM_256=2 python tools/prof_shapes.py > gpurun_out/r4g/s_256.txt 2>&1
DM_GEMM_RING=2 python tools/prof_shapes.py > gpurun_out/r4g/s_ring.txt 2>&1
DM_GEMM_W4=3 python tools/prof_shapes.py > gpurun_out/r4g/s_w43.txt 2>&1
DM_GEMM_W4_TN=2 python tools/prof_shapes.py > gpurun_out/r4g/s_w4tn2.txt 2>&1
DM_GEMM_256=0 DM_GEMM_RING=0 DM_GEMM_W4=0 python tools/prof_shapes.py > gpurun_out/r4g/s_plain.txt 2>&1
DM_GEMM_T128_TOUCH=0 python tools/prof_shapes.py > gpurun_out/r4g/s_notouch.txt 2>&1
python - <<'PY'
import re, glob, collections
tabs = {}
for f in sorted(glob.glob('gpurun_out/r4g/s_*.txt')):
    name = f.split('s_')[1][:-4]
    for l in open(f):
        m = re.match(r'(gemm_\w+?_\d+x\d+x\d+_e\w+?)_t(\d+)\s+([\d.]+)\s+([\d.]+)', l)
        if m:
            tabs.setdefault(m.group(1), {})[name] = (float(m.group(3)), m.group(2))
    tot = [l for l in open(f) if l.startswith('total')]
    print(name, tot[-1].split()[-1] if tot else '?')
names = ['default', '256', 'ring', 'w43', 'w4tn2', 'plain', 'notouch']
print('%-44s' % 'shape' + ''.join('%14s' % n for n in names))
for k, v in sorted(tabs.items(), key=lambda kv: -kv[1].get('default', (0,))[0]):
    if v.get('default', (0,))[0] < 0.03: continue
    print('%-44s' % k + ''.join(('%8.3f/%-5s' % v[n]) if n in v else '%14s' % '-' for n in names))
PY
