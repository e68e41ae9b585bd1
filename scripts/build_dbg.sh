#!/bin/bash
# diagnostic builds (run in the build container; the .so files travel to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/dbg
hipcc --offload-arch=gfx950 -O3 -fPIC -shared -DSF_STAMPS -Iinclude -Itinyrecurrentunet_amd/csrc \
    tinyrecurrentunet_amd/csrc/stream_fwd.hip -o scripts/dbg/libsf_stamps.so
hipcc --offload-arch=gfx950 -O3 -fPIC -shared -DSF_STAMPS -DSF_ABL=1 -Iinclude -Itinyrecurrentunet_amd/csrc \
    tinyrecurrentunet_amd/csrc/stream_fwd.hip -o scripts/dbg/libsf_abl1.so
hipcc --offload-arch=gfx950 -O3 -fPIC -shared -DSF_STAMPS -DSF_ABL=3 -Iinclude -Itinyrecurrentunet_amd/csrc \
    tinyrecurrentunet_amd/csrc/stream_fwd.hip -o scripts/dbg/libsf_abl3.so
hipcc --offload-arch=gfx950 -O3 -fPIC -shared -DSF_STAMPS -DSF_ABL=9 -Iinclude -Itinyrecurrentunet_amd/csrc \
    tinyrecurrentunet_amd/csrc/stream_fwd.hip -o scripts/dbg/libsf_abl9.so
echo built scripts/dbg/libsf_stamps.so scripts/dbg/libsf_abl1.so scripts/dbg/libsf_abl3.so scripts/dbg/libsf_abl9.so
