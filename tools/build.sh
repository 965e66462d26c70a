#!/bin/bash
# build the library from anywhere: bash /root/repo/tools/build.sh
cd /root/repo && python -c "
import importlib.util
spec=importlib.util.spec_from_file_location('b','3d-semantic-segmentation-amp-net_amd/build.py');m=importlib.util.module_from_spec(spec);spec.loader.exec_module(m);m.build()" 2>&1 | grep -v "^/opt/rocm/bin/hipcc\|BW_NW\|^ *24 \|\^~\|warning generated"
