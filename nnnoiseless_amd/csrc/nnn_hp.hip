// nnn_hp.hip -- the translation unit of the two high-pass kernels (k_hp, k_hp2): the same source as everywhere else (nnn_kernels.hip), compiled
// WITH the compiler's SLP pairing while the rest of the library is compiled without it (see the note at k_hp2 in nnn_kernels.hip; round 6).
#include <hip/hip_runtime.h>
#define NNN_ONLY_HP 1
#include "nnn_kernels.hip"
