#!/bin/bash
# builds ../libcsd_hip_xwabl<N>.so for every N given: the tuning library with conv_xw.hip compiled with -DXW_ABL=N
cd /root/repo/conditional_score_diffusion_amd/csrc
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize -DCSD_TUNE -DCSD_FF_TUNE -DXW_ABL=$n -c conv_xw.hip -o /tmp/conv_xw_abl$n.o 2>/dev/null &
done
wait
for n in "$@"; do
  objs=$(ls *.tune.o | grep -v conv_xw.tune.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/conv_xw_abl$n.o -o ../libcsd_hip_xwabl$n.so
done
ls -la ../libcsd_hip_xwabl*.so
