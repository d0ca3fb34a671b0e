// Translation unit of one kernel-table slice of libmigan_hip.so:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DMIGAN_SLICE_G=<0|1|2> -DMIGAN_SLICE_S=<0|1|2> migan_k_slice.hip
#include "migan_rt_hip.h"
#define MIGAN_TEMPLATE_KERNELS_ONLY
#include "migan_kernels.hpp"
#include "migan_table.hpp"
#include "migan_k_slice.inc"
