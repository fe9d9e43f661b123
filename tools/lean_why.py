import os, sys, subprocess
# run a few steps of a config under TRGT_WFA_DEBUG and sum the "[lean]" lines
cfg = sys.argv[1]
env = dict(os.environ, TRGT_WFA_DEBUG="1")
p = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--steps", "1", "--warmup", "1", "--no-streaming", "--no-cpu-baseline", "--contexts", "1"], env=env, capture_output=True, text=True)
lines = [l for l in p.stderr.splitlines() if "lean kernels" in l or "handed on" in l]
print("cfg", cfg, len(lines), "lines"); print(p.stderr[-600:] if not lines else "")
for l in lines[-16:]: print(l)
