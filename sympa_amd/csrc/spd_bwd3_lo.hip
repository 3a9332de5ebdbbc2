// Three-kernel SPD backward, instantiations n = 9..12 (the other half of spd_bwd3.hip).
#define SYMPA_BWD3_LO 1
#include "spd_bwd3.hip"
