#!/bin/bash
# Experiment builds: tools/build_variant.sh <name> "<extra compiler flags>" <file.hip>...  -> accurate_aprilgroup_tracking_amd/libagt_hip_exp_<name>.so
# (the knobs build with the named sources recompiled under the extra flags; objects under tools/_exp/<name>/, nothing tracked)
set -e
name=$1; extra=$2; shift 2
cd "$(dirname "$0")/../accurate_aprilgroup_tracking_amd/csrc"
make -s -j8 knobs
mkdir -p ../../tools/_exp/$name
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -DAGT_DEBUG_KNOBS $extra"
STEPFLAGS="${STEPFLAGS--mllvm -disable-machine-licm}"     # (as the Makefile: agt_step_nolicm.hip only)
objs=""
for f in agt_api agt_pyramid agt_lk agt_pnp agt_step agt_step_nolicm agt_step_dense agt_preproc agt_dense; do
  if [[ " $* " == *" $f.hip "* ]]; then
    sf=""; [[ $f == agt_step_nolicm ]] && sf=$STEPFLAGS
    /opt/rocm/bin/hipcc $FLAGS $sf -c $f.hip -o ../../tools/_exp/$name/$f.o &
    objs="$objs ../../tools/_exp/$name/$f.o"
  else
    objs="$objs $f.knobs.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libagt_hip_exp_$name.so $objs
echo built libagt_hip_exp_$name.so
