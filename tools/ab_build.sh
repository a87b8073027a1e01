#!/bin/bash
# tools/ab_build.sh NAME [extra hipcc flags]: builds the three tower-machine translation units with the extra flags and
# links kyber_amd/lib/libkyberhip_NAME.so (the other objects are reused) -- for same-box A/B runs through KYBER_HIP_LIB.
set -e
cd "$(dirname "$0")/../kyber_amd/csrc"
name=$1; shift
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value $*"
mkdir -p /tmp/ab_$name
for f in bls12381_pair bn256_pair bn254_pair; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o /tmp/ab_$name/$f.o & done; wait
objs=""; for o in context ed25519 bls12381 bls12381_prep bls12381_msm bls12381_h2c bn256 bn256_msm bn254 bn254_msm; do objs="$objs $o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/ab_$name/*.o -o ../lib/libkyberhip_$name.so
ls -la ../lib/libkyberhip_$name.so
