#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]   -> demuxlet_amd/libdmx_<name>.so   (kernel experiments only)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -fvisibility=default -Iinclude -Idemuxlet_amd/csrc "$@" -x hip demuxlet_amd/csrc/dmx_host.cpp demuxlet_amd/csrc/dmx_engine.hip -o demuxlet_amd/libdmx_$name.so
