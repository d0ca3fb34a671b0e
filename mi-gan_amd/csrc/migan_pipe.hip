// Translation unit of the software-pipelined SeparableConv2d kernels of libmigan_hip.so:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c migan_pipe.hip
#include "migan_rt_hip.h"
#define MIGAN_TEMPLATE_KERNELS_ONLY
#include "migan_kernels.hpp"
#include "migan_table.hpp"
#include "migan_pipe.hpp"
#include "migan_pipe_table.inc"
