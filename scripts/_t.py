import os, sys
sys.argv = ["x", "none"]
exec(open("scripts/engine_check.py").read().split('if mode in ("quick", "all"):')[0])
env = {k: v for k, v in os.environ.items() if k.startswith("RFLU_")}
_, _, _, info, t = factor(int(os.environ.get("TN", "16384")), "f64", 0, env, reps=3)
print(f"{env}: info {info} best {t:.2f} ms", flush=True)
