// Translation unit of the 256-pixel x 256-channel persistent tile kernel of libmigan_hip.so:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c migan_wide2.hip
#include "migan_rt_hip.h"
#define MIGAN_TEMPLATE_KERNELS_ONLY
#include "migan_kernels.hpp"
#include "migan_table.hpp"
#include "migan_pipe.hpp"
#include "migan_wide2.hpp"
#include "migan_wide2_table.inc"
