#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_input_stage.py tests/test_datamodule.py -q > $O/pytest_input.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_input.txt | tail -5 | tee -a $O/summary.txt
for rep in 1 2; do
timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timer --with-input-stage 2>/dev/null | tail -1 > $O/pcie_$rep.json
python - <<PY | tee -a $O/summary.txt
import json
d=json.loads(open("$O/pcie_$rep.json").read())
print("rep $rep: resident", d["value"], "tiles/s; PCIe-inclusive", d.get("pcie_inclusive"))
PY
done
