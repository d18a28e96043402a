#!/bin/bash
# builds libarachne_amd.so variants for A/B runs of compile-time switches: tools/build_variants.sh NAME "-DFLAG=0 ..." ...
# the variants land in gpurun_out/variants/ (scratch) -- copy them under arachne_amd/variants/ to ship them to the GPU box
set -e
cd "$(dirname "$0")/.."
mkdir -p arachne_amd/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-value -c -O3 $flags arachne_amd/csrc/arx_api.hip -Rpass-analysis=kernel-resource-usage -o /tmp/arx_var_$name.o 2> /tmp/arx_var_$name.log &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/arx_var_$name.o arachne_amd/csrc/arx_cold.o arachne_amd/csrc/arx_index.o arachne_amd/csrc/arx_feeder.o arachne_amd/csrc/arx_bam.o arachne_amd/csrc/arx_multi.o -o arachne_amd/variants/lib_$name.so -Wl,-rpath,/opt/rocm/lib -lz && echo built $name ) &
done
wait
