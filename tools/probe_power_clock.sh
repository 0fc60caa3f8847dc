#!/bin/bash
# Why do the GEMM kernels run at 1.5-1.9 GHz and not at 2.4?  Samples rocm-smi (power, shader / memory clock, temperature)
# every ~30 ms while (a) the default training + inference bench runs, (b) one GEMM layer loops on random operands, (c) the same
# layer loops on ZERO operands (same instruction stream, far fewer bit flips), (d) an HBM-streaming kernel loops.
#   tools/probe_power_clock.sh > gpurun_out/power_clock.txt
cd "$(dirname "$0")/.."
sample() {   # $1 = label, $2.. = command
  local label=$1; shift
  ( while true; do rocm-smi -P -c -t --json 2>/dev/null | tr -d '\n'; echo; sleep 0.03; done ) > /tmp/smi_$label.txt &
  local spid=$!
  "$@" > /tmp/out_$label.txt 2>&1
  kill $spid 2>/dev/null; wait $spid 2>/dev/null
  python3 - "$label" <<'PY'
import json, sys, statistics as st
label = sys.argv[1]
pw, sclk, mclk, temp = [], [], [], []
for ln in open(f"/tmp/smi_{label}.txt"):
    try:
        d = json.loads(ln)["card0"]
    except Exception:
        continue
    for k, v in d.items():
        kl = k.lower()
        try:
            if "power" in kl and "(w)" in kl: pw.append(float(v))
            elif kl.startswith("sclk clock speed"): sclk.append(float(str(v).strip("()Mhz ")))
            elif kl.startswith("mclk clock speed"): mclk.append(float(str(v).strip("()Mhz ")))
            elif "temperature" in kl and "junction" in kl: temp.append(float(v))
        except ValueError:
            pass
def s(x): return f"n {len(x):4d} median {st.median(x):7.1f} p10 {sorted(x)[len(x)//10]:7.1f} max {max(x):7.1f}" if x else "n/a"
print(f"[{label}] power W: {s(pw)} | sclk MHz: {s(sclk)} | mclk MHz: {s(mclk)} | T junction: {s(temp)}")
PY
  tail -3 /tmp/out_$label.txt | cut -c1-300
}
rocm-smi --showmaxpower 2>/dev/null | grep -i "max\|power" | head -3
sample idle sleep 2
sample bench python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline --no-input-stage
sample gemm_random python tools/probe_power_gemm.py random
sample gemm_zero python tools/probe_power_gemm.py zero
sample hbm_stream python tools/probe_power_gemm.py stream
