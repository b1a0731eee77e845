#!/bin/sh
# experimental library: tools/build_exp.sh <suffix> <-D flags...>   (never loaded by tests or bench)
set -e
sfx=$1; shift
cd "$(dirname "$0")/../tatajuba_amd/csrc"
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include "$@" -c hopo_device.hip -o /tmp/tj_dev_$sfx.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libtatajuba_amd_$sfx.so hopo_host.o context_host.o fastq_reader.o feeder.o tj_inflate.o synth.o version.o /tmp/tj_dev_$sfx.o -lz -lpthread -ldl
