"""Per-shape GEMM table of one bench run: DM_PROF_SHAPES=1 python tools/prof_shapes.py [bench.py args]."""
import json, os, subprocess, sys
env = dict(os.environ, DM_PROF_SHAPES="1")
out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "..", "bench.py"), "--no-cpu-baseline"] + sys.argv[1:],
                     env=env, capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
r = json.loads(line)["roofline"]
ms, tf = r["all_kernels_ms_per_step"], r["all_kernels_TFLOPs"]
tot = sum(ms.values())
print(f"{'kernel':64s} {'ms/step':>8s} {'TFLOP/s':>8s}")
for k in sorted(ms, key=lambda k: -ms[k]):
    print(f"{k:64s} {ms[k]:8.3f} {tf.get(k, 0):8.1f}")
print(f"{'total':64s} {tot:8.3f}")
