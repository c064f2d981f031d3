#!/bin/bash
# developer tool: build agatha_amd/libagatha_amd_var.so with ONLY the (32,3) instantiation and extra compiler flags,
# for quick A/B runs:  tools/build_variant.sh "<extra flags>"  then  AGATHA_AMD_LIB=.../libagatha_amd_var.so bench.py
set -e
cd /root/repo/agatha_amd/csrc
mkdir -p _obj/var
python3 - <<'PY'
s=open('align_kernel.hip').read()
a=s.index('static const Cfg kCfgs[] = {'); b=s.index('};',a)
s=s[:a]+'static const Cfg kCfgs[] = {\n    {32, 3, launch_align_t<32, 3>},\n'+s[b:]
open('_obj/var/align_kernel_var.hip','w').write(s)
PY
cp kernels.h _obj/var/
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $1 -c _obj/var/align_kernel_var.hip -o _obj/var/align_kernel.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -I. -c capi.cpp -o _obj/var/capi.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libagatha_amd_var_$2.so _obj/var/align_kernel.o _obj/var/capi.o
ls -la ../libagatha_amd_var_$2.so
