#!/bin/sh
# diagnostic library with in-kernel stamps (never shipped / never loaded by tests or bench)
set -e
cd "$(dirname "$0")/../tatajuba_amd/csrc"
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTJ_STAMPS=1 -c hopo_device.hip -o /tmp/tj_dev_diag.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libtatajuba_amd_diag.so hopo_host.o context_host.o fastq_reader.o feeder.o tj_inflate.o synth.o version.o /tmp/tj_dev_diag.o -lz -lpthread -ldl
