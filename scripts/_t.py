import os, sys
sys.argv = ["x", "none"]
exec(open("scripts/engine_check.py").read().split('if mode in ("quick", "all"):')[0])
base = {"RFLU_ENGINE": "1", "RFLU_ENGINE_ROWS": "0", "RFLU_ENGINE_X5": "1", "RFLU_ENGINE_X6": "16", "RFLU_ENGINE_X2": "1"}
for n in (16384, 8192):
    for extra in ({}, {"RFLU_ENGINE_ROWS": "4096"}, {"RFLU_ENGINE_ROWS": "2048"}, {"RFLU_ENGINE_X6": "4"}, {"RFLU_ENGINE_X6": "32"}, {"RFLU_ENGINE_X5": "0", "RFLU_ENGINE_X6": "0"}):
        env = dict(base); env.update(extra)
        _, _, _, info, t = factor(n, "f64", 0, env, reps=3)
        print(f"n={n} {extra}: info {info} best {t:.2f} ms", flush=True)
