// pairing/bn256: Suite.Pair / ValidatePairing / pointGT.Mul entry points (bn_pair.inc over the bn256 device library
// and its generated tower-machine programs).
#include "bn256.cuh"
#include "tower_vm_bn256.inc"
#define KYB_BN_PFX bn256
#define KYB_BN_NS bn
#define KYB_BN_VMNS bnvm
#define KYB_BN_VM Bn256Vm
#define KYB_BN_TVM(x) TVM_BN256_##x
#define KYB_BN_HAS_CHECKP 1  // a product-form ValidatePairing program next to the literal one
#include "bn_pair.inc"
