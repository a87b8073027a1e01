// pairing/bn254: Suite.Pair / ValidatePairing / pointGT.Mul entry points (bn_pair.inc over the bn254 device library
// and its generated tower-machine programs: xi = 9 + i, 27-bit limbs).
#include "bn254.cuh"
#include "tower_vm_bn254.inc"
#define KYB_BN_PFX bn254
#define KYB_BN_NS bn4
#define KYB_BN_VMNS bn4vm
#define KYB_BN_VM Bn254Vm
#define KYB_BN_TVM(x) TVM_BN254_##x
#include "bn_pair.inc"
