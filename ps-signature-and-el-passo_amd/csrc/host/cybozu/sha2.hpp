// Stand-in for <cybozu/sha2.hpp> (src/ps-verifier.cc:4).
#pragma once
#include "../elp_mcl_compat.h"
