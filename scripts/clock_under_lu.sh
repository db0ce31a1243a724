#!/bin/bash
# samples the shader clock / power (sysfs, every ~20 ms) while bench.py factors N=16384 matrices back to back: is the in-schedule
# GEMM rate clock- / power-limited?
cd $GRAFT_REPO_ROOT
python bench.py --steps ${1:-100} --warmup 2 --no-cpu-baseline --no-extras --no-check > gpurun_out/clk_lu.log 2>&1 &
PID=$!
python - $PID <<'PY'
import glob, os, sys, time
pid = int(sys.argv[1])
sclk = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
pw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
freq = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")
print("files:", sclk[:1], pw[:1], freq[:1])
rows = []
t0 = time.time()
while True:
    try:
        os.kill(pid, 0)
    except OSError:
        break
    r = [round(time.time() - t0, 2)]
    for fl in (freq[:1], pw[:1]):
        try:
            r.append(int(open(fl[0]).read().strip()))
        except Exception:
            r.append(-1)
    try:
        cur = [l for l in open(sclk[0]).read().splitlines() if l.endswith("*")]
        r.append(cur[0] if cur else "?")
    except Exception:
        r.append("?")
    rows.append(r)
    time.sleep(0.02)
# print a thinned series
for r in rows[::10]:
    print(r)
PY
wait $PID
tail -1 gpurun_out/clk_lu.log | grep -o '"ms_per_step": [0-9.]*'
