#!/bin/bash
# [AB_TUS="tu ..."] tools/ab_build.sh NAME [extra hipcc flags]: builds the named translation units (default: the three
# tower-machine ones) with the extra flags and
# links kyber_amd/lib/libkyberhip_NAME.so (the other objects are reused) -- for same-box A/B runs through KYBER_HIP_LIB.
set -e
cd "$(dirname "$0")/../kyber_amd/csrc"
name=$1; shift
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value $*"
rm -rf /tmp/ab_$name; mkdir -p /tmp/ab_$name
TUS=${AB_TUS:-bls12381_pair bn256_pair bn254_pair}
for f in $TUS; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o /tmp/ab_$name/$f.o & done; wait
objs=""; for o in context scalar_poly ed25519 bls12381 bls12381_g1split bls12381_unm2 bls12381_fb bls12381_pair bls12381_prep bls12381_msm bls12381_msm_plain bls12381_msm_gls bls12381_h2c bn256 bn256_pair bn256_msm bn256_msm_glv bn254 bn254_pair bn254_msm bn254_msm_glv; do case " $TUS " in *" $o "*) ;; *) objs="$objs $o.o";; esac; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/ab_$name/*.o -o ../lib/libkyberhip_$name.so
ls -la ../lib/libkyberhip_$name.so
