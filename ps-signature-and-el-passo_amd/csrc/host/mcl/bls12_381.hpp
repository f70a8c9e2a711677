// Stand-in for <mcl/bls12_381.hpp> (src/ps-encoding.h:5): the same include line resolves to the GPU-backed types.
#pragma once
#include "../elp_mcl_compat.h"
