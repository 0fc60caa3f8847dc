#!/bin/bash
# UNet++ on the reference's shipped encoder (resnext101_32x8d, configs/unetplus_config_RGB.yaml:37): grouped 3x3 as batched
# super-groups (round 6) vs the block-diagonal dense filter of round 3 (GDL_GROUPED_DENSE=1); tools/bench_unetpp_encoder.py
cd "$(dirname "$0")/.."
for v in 1 0 1 0; do echo "== GDL_GROUPED_DENSE=$v"; GDL_GROUPED_DENSE=$v python tools/bench_unetpp_encoder.py resnext101_32x8d 8 2>&1 | grep -v amdgpu.ids | tail -3; done
