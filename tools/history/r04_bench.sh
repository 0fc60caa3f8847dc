#!/bin/bash
# default bench line (all extras, CPU baseline skipped unless $2 = cpu) -> gpurun_out/$1/bench.json + a short summary
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04bench}; mkdir -p $O; cd $R
EXTRA="--no-cpu-baseline"; [ "$2" = "cpu" ] && EXTRA=""
timeout 1500 python bench.py $EXTRA > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err | cut -c1-300
python - "$O/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r = d["roofline"]
print("train %.1f tiles/s (%.2f ms)  inference %.1f (%.2f ms)  MFU %s  executed %s" % (d["value"], d["ms_per_step"], d["inference_tiles_per_s"], d["inference_ms_per_step"], d["model_flops_utilisation"], d.get("executed_flops_utilisation")))
print("roofline: %s  %.1f TF/s frac %.3f share %.3f traffic %s" % (r["kernel"][:50], r["achieved"], r["frac"], r["share_of_step_time"], r["traffic"]))
for l in r["by_layer"]: print("   ", l)
print("other variants:", {k[:40]: v for k, v in r["other_conv_gemm_variants"].items()})
for k, v in d.get("step_roofline", {}).items():
    print(k, "frac_of_bound", v["frac_of_bound"], [(o["op"], o["ms"], o["frac_of_own_bound"]) for o in v["top_ops"]])
print("pcie", d.get("pcie_inclusive", {}).get("train_tiles_per_s"))
for b, v in d.get("by_batch", {}).items():
    print("batch", b, "default path %.1f / %.1f" % (v["train_tiles_per_s"], v["inference_tiles_per_s"]), "eager", v.get("eager"), v.get("default_path", "")[:20])
for m, v in d.get("other_models", {}).items():
    print(m, v.get("tile"), "batch", v["per_gpu_batch"], "train %.1f infer %.1f" % (v["train_tiles_per_s"], v["inference_tiles_per_s"]), "MFU", v["model_flops_utilisation"],
          {k: (x["frac_of_bound"], [(o["op"], o["ms"], o["frac_of_own_bound"]) for o in x["top_ops"]]) for k, x in v.get("roofline", {}).items()})
PY
