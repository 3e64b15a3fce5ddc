// Unit 0 of the column-strip chain kernels (chain_t.hpp): the explicit instantiations of its share of the (gh, L) shapes.
#include "chain_t.hpp"
GC_CHAIN_T_UNIT(0)
