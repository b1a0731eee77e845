#!/bin/bash
# several experimental libraries at once: tools/build_exp_many.sh "sfx1:-DA=1 -DB=2" "sfx2:-DC=3" ...   (product objects must be up to date: run make first)
cd "$(dirname "$0")/../tatajuba_amd/csrc" || exit 1
make -s || exit 1
for spec in "$@"; do
  sfx=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include $flags -c hopo_device.hip -o /tmp/tj_dev_$sfx.o 2>/tmp/tj_dev_$sfx.err \
    && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libtatajuba_amd_$sfx.so hopo_host.o context_host.o fastq_reader.o feeder.o tj_inflate.o synth.o version.o /tmp/tj_dev_$sfx.o -lz -lpthread -ldl \
    && echo "built $sfx" || { echo "FAILED $sfx"; grep -m5 error /tmp/tj_dev_$sfx.err; } ) &
done
wait
