// hades252.hip -- the one translation unit of libhades252 (gfx950 only): arithmetic headers, constant tables, the kernel
// headers by domain (kernels_*.hpp) and, below, launch policy + the C ABI of include/hades252.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/hades252.h"
#include "fr32.hpp"
#include "hades_constants.inc"
#include "hades_literal.hpp"
#include "staging.hpp"
#include "hades_fast.hpp"
#include "k_perm_fast.hpp"
#include "hades_coop.hpp"
#include "hades_lanes.hpp"

using namespace hades;

#include "device_tables.hpp"
#include "kernels_perm.hpp"
#include "kernels_merkle.hpp"
#include "kernels_sponge.hpp"
#include "kernels_aux.hpp"


// ------------------------------------------------------------------------------------------
// host side: launch policy and the C ABI of include/hades252.h, by domain (one translation unit: -fno-gpu-rdc)
// ------------------------------------------------------------------------------------------
#include "host_fault.hpp"
#include "launch.hpp"
#include "abi_perm.hpp"
#include "abi_merkle.hpp"
#include "abi_sponge.hpp"
#include "abi_util.hpp"
#include "host_pin.hpp"
#include "host_pool.hpp"
#include "host_pipe.hpp"
#include "host_callers.hpp"
