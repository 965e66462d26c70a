#!/bin/bash
# builds tools/lib_stamps.so = the library with the split kernels' cycle stamps compiled in (-DAMPNET_PW_STAMPS); tools/x3_stamps.py reads them
cd /root/repo/3d-semantic-segmentation-amp-net_amd/csrc || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAMPNET_PW_STAMPS=1 -c pw_gemm.hip -o /tmp/pw_gemm_stamps.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAMPNET_PW_STAMPS=1 -c pw_bwd_x3.hip -o /tmp/pw_bwd_x3_stamps.o || exit 1
OBJS=$(ls _obj/*.o | grep -v 'pw_gemm.o\|pw_bwd_x3.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/lib_stamps.so $OBJS /tmp/pw_gemm_stamps.o /tmp/pw_bwd_x3_stamps.o && echo built tools/lib_stamps.so
