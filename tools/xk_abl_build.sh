#!/bin/bash
# builds ../libcsd_hip_xkabl<N>.so for every N given: the tuning library with conv_xk.hip compiled with -DXK_ABL=N
cd /root/repo/conditional_score_diffusion_amd/csrc
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize -DCSD_TUNE -DCSD_FF_TUNE -DXK_ABL=$n -c conv_xk.hip -o /tmp/conv_xk_abl$n.o 2>/dev/null &
done
wait
for n in "$@"; do
  objs=$(ls *.tune.o | grep -v conv_xk.tune.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/conv_xk_abl$n.o -o ../libcsd_hip_xkabl$n.so
done
ls -la ../libcsd_hip_xkabl*.so
