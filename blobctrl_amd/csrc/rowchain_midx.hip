// BC_CHAIN_MIDX (the MID launch of the row-chain with the block's cross-attention inside: rowchain.hip) as its own translation unit.
// hipcc allocates the registers of the other row-chain kinds differently - and spills (1 - 11 registers in MID and OUT, all of which
// sit at 250 - 256 VGPRs) - as soon as this instantiation lives in the same module, so it is compiled apart from them: the same
// source, only the MIDX kernel, bc_rowchain_pack_kv and the bc_rowchain_midx entry point.
#define BC_ROWCHAIN_MIDX_TU 1
#include "rowchain.hip"
